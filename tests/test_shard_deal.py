"""CPU tier: the task -> rank deal of the sharded archiver (CSAMI_PlanShards / CSAMI_AddShardEncode, csa_archive.cpp).

The reference's workers take the next task of the size-sorted list whenever one is free (csarc.cpp:361-398); ranks cannot, so every
rank computes the same longest-processing-time-first deal over estimated costs (bytes x a data-kind factor).  Checked here on the
many-task workload of bench.py (csc_amd/treegen.py): every task dealt exactly once, the same deal from two calls, and the most
loaded rank within 5 % of the mean at world 2 / 4 / 8 -- with round 3's `i mod world` printed beside it."""
import os

import pytest

from csc_amd import corpus, csa, treegen


def _sparse_tree(root, spec):
    """the tree's names, sizes and FIRST 8 KiB (what the cost estimate looks at); the rest of every file is a hole"""
    for rel, kind, seed, size in treegen.files(spec):
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(corpus.fill(kind, seed, 0, min(size, 8192)).tobytes())
            f.truncate(size)


def _loads(rank_of, cost, world):
    load = [0.0] * world
    for r, c in zip(rank_of, cost):
        load[r] += c
    return load


@pytest.mark.parametrize("spec", ["tree_small", "tree"])
def test_deal_is_balanced_and_deterministic(tmp_path, spec, monkeypatch):
    root = str(tmp_path)
    _sparse_tree(root, spec)
    monkeypatch.chdir(root)
    monkeypatch.delenv("CSA_DEAL", raising=False)
    ntasks = None
    for world in (2, 4, 8):
        ro, co = csa.plan_shards(["t"], world, level=3, dict_size=64 << 20, recurse=True)
        ro2, co2 = csa.plan_shards(["t"], world, level=3, dict_size=64 << 20, recurse=True)
        assert ro == ro2 and co == co2                       # every rank computes the same deal
        ntasks = ntasks or len(ro)
        assert len(ro) == ntasks and all(0 <= r < world for r in ro)
        assert len({c for c in co}) > 1 and min(co) > 0      # kinds differ: text-like tasks cost less per byte
        load = _loads(ro, co, world)
        mean = sum(load) / world
        monkeypatch.setenv("CSA_DEAL", "mod")
        rm, cm = csa.plan_shards(["t"], world, level=3, dict_size=64 << 20, recurse=True)
        monkeypatch.delenv("CSA_DEAL")
        assert rm == [i % world for i in range(ntasks)]
        lm = _loads(rm, co, world)                           # the static deal, priced with the same cost estimate
        print(f"{spec} world {world}: {ntasks} tasks, LPT max/mean {max(load) / mean:.4f}, i mod world max/mean {max(lm) / mean:.4f}")
        assert max(load) / mean <= 1.05
        assert max(load) <= max(lm) + 1e-9
    assert ntasks == treegen.SPECS[spec][0] // treegen.SPECS[spec][1]


def test_one_rank_gets_everything(tmp_path, monkeypatch):
    _sparse_tree(str(tmp_path), "tree_small")
    monkeypatch.chdir(str(tmp_path))
    ro, _ = csa.plan_shards(["t"], 1, level=3, dict_size=64 << 20, recurse=True)
    assert set(ro) == {0}
