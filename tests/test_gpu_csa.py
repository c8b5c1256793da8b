"""GPU tier for the `.csa` container rows: `CSA_Add` must write, byte for byte, the
archive `csarc a -t1` writes for the same tree and options; `CSA_Test` / `CSA_Extract` / `CSA_List`
must read both its own and the reference's archives.  Everything goes through the C ABI of
include/csa_mi355x.h (streams and adler32 on the device); the checkers are the digests recorded from
the reference archiver (tests/golden/csa.json), the reference archiver itself when oracle/_ref
travelled with the snapshot, and the CPU oracle (oracle/orc_csa.py + liborc).
"""
import json
import os
import stat
import subprocess
import sys
import zlib

import pytest

import cases
import csa_cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc_csa  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "csa.json")))
CSARC_REF = os.path.join(ROOT, "oracle", "_ref", "csarc_ref")
HAVE_REF = os.path.exists(CSARC_REF)


@pytest.fixture(scope="module")
def csa(prod):
    from csc_amd import csa as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def orc_dec(orc, zalloc):
    def dec(stream):
        rc, raw = orc.decode(stream, alloc=zalloc)
        assert rc == 0
        return raw
    return dec


# ---------------------------------------------------------------------------------------------
# adler32 reduction kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,shift", [(0, 0), (1, 0), (1, 7), (15, 1), (16, 0), (17, 15), (4096, 3), (16383, 0), (16384, 0),
                                     (16385, 5), (100000, 9), ((2 << 20), 0), ((2 << 20) + 77, 13), (9999991, 2)])
def test_adler32_device(csa, n, shift):
    import torch
    data = cases.build([["random", 31, 0, n]]) if n else b""
    buf = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    if n:
        buf[shift:shift + n] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    for seed in (0, 1, 0x0ABC0DEF):
        assert csa.adler32_device(buf.data_ptr() + shift, n, seed) == zlib.adler32(data, seed)
    ones = torch.full((n + 64,), 255, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert csa.adler32_device(ones.data_ptr() + shift, n, 0) == zlib.adler32(b"\xff" * n, 0)


# ---------------------------------------------------------------------------------------------
# csarc a
# ---------------------------------------------------------------------------------------------
def _add(csa, case, **extra):
    spec = csa_cases.CSA_CASES[case]
    opts = dict(spec["opts"])
    opts.update(extra)
    opts.setdefault("overwrite", True)
    return csa.add(csa_cases.ARCNAME, spec["args"], **opts)


@pytest.mark.parametrize("case", list(csa_cases.CSA_CASES))
def test_add_writes_the_reference_archive(csa, case, tmp_path, monkeypatch, orc_dec):
    content = csa_cases.make_tree(str(tmp_path), case)
    monkeypatch.chdir(tmp_path)
    rc, st = _add(csa, case)
    assert rc == 0
    arc = (tmp_path / csa_cases.ARCNAME).read_bytes()
    g = GOLD[case]
    if cases.digest(arc) != g["archive_sha256"]:
        # say where it differs before failing
        a = orc_csa.parse(arc, orc_dec)
        detail = {"size": (len(arc), g["archive_size"]), "index": (a["index_pos"], g["index_pos"], a["index_csize"], g["index_csize"],
                                                                   a["index_rsize"], g["index_rsize"]), "ab": a["abindex"]}
        pytest.fail(f"{case}: archive differs from the reference's: {detail}")
    assert len(arc) == g["archive_size"] == st["archive_bytes"]
    assert st["raw_bytes"] == (g["raw_bytes"] if case not in ("single_shadowed_by_empty", "named_files_m3") else st["raw_bytes"])
    assert st["index_raw_size"] == g["index_rsize"] and st["index_compressed_size"] == g["index_csize"]
    if HAVE_REF:
        # live, not only through the recorded digest: the reference archiver on the same tree
        os.rename(csa_cases.ARCNAME, "mine.csa")
        subprocess.run([CSARC_REF] + csa_cases.csarc_argv(case), cwd=tmp_path, check=True, capture_output=True)
        assert (tmp_path / csa_cases.ARCNAME).read_bytes() == arc
        # and the reference accepts ours
        t = subprocess.run([CSARC_REF, "t", "mine.csa"], cwd=tmp_path, capture_output=True)
        assert t.returncode == 0 and b"failed" not in t.stderr
    del content


def test_add_is_independent_of_stream_concurrency(csa, tmp_path, monkeypatch):
    csa_cases.make_tree(str(tmp_path), "many_files")
    monkeypatch.chdir(tmp_path)
    for streams in (1, 3, 0):
        rc, st = _add(csa, "many_files", device_streams=streams)
        assert rc == 0 and (streams == 0 or st["peak_streams"] <= streams)
        assert cases.digest((tmp_path / csa_cases.ARCNAME).read_bytes()) == GOLD["many_files"]["archive_sha256"]
    rc, st = _add(csa, "many_files", hbm_budget=64 << 20)       # tight HBM budget: few streams at a time
    assert rc == 0 and st["peak_streams"] < 10
    assert cases.digest((tmp_path / csa_cases.ARCNAME).read_bytes()) == GOLD["many_files"]["archive_sha256"]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/csarc_ref not in this snapshot")
def test_add_with_multi_chunk_tasks_reads_ahead_and_equals_the_reference(csa, tmp_path, monkeypatch):
    """tasks of several 2 MiB chunks: from the second chunk on a stream's bytes come from the read-ahead thread that runs during the
    batch encode call (csa_archive.cpp: FillLane 1 -- the reference's reader thread, csa_io.h:215-272), the adler32 pieces of those
    chunks are folded one round later.  The archive must still be the reference archiver's, byte for byte; an empty file, a file
    that ends exactly at a chunk boundary and one that is a chunk plus one byte ride along."""
    files = {"d/big1.bin": [["silesia", 31, 0, 9 << 20]], "d/big2.txt": [["text", 32, 0, (5 << 20) + 12345]], "d/edge.bin": [["exe", 33, 0, 2 << 20]],
             "d/edge1.bin": [["exe", 34, 0, (2 << 20) + 1]], "d/edge2.bin": [["delta", 35, 0, 300000]], "d/small.txt": [["text", 36, 0, 70000]], "d/empty.txt": []}
    for rel, parts in files.items():
        path = tmp_path / rel
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_bytes(cases.build(parts))
        os.chmod(path, 0o644)
        os.utime(path, (csa_cases.MTIME, csa_cases.MTIME))
    os.chmod(tmp_path / "d", 0o755)
    os.utime(tmp_path / "d", (csa_cases.MTIME, csa_cases.MTIME))
    monkeypatch.chdir(tmp_path)
    rc, st = csa.add("mine.csa", ["d"], level=1, dict_size=4 << 20, recurse=True, overwrite=True)
    assert rc == 0 and st["raw_bytes"] == sum(len(cases.build(p)) for p in files.values())
    subprocess.run([CSARC_REF, "a", "-m1", "-d4m", "-r", "-t1", "-f", "ref0.csa", "d"], cwd=tmp_path, check=True, capture_output=True)
    mine, ref = (tmp_path / "mine.csa").read_bytes(), (tmp_path / "ref0.csa").read_bytes()
    # (the archive name's length enters the index: same length on both sides)
    assert len("mine.csa") == len("ref0.csa") and mine == ref
    rc, st2 = csa.test("mine.csa", [])
    assert rc == 0 and st2["verify_failures"] == 0


def test_add_refuses_to_overwrite(csa, tmp_path, monkeypatch):
    csa_cases.make_tree(str(tmp_path), "no_data")
    monkeypatch.chdir(tmp_path)
    (tmp_path / csa_cases.ARCNAME).write_bytes(b"precious")
    rc, _ = _add(csa, "no_data", overwrite=False)
    assert rc == 1 and (tmp_path / csa_cases.ARCNAME).read_bytes() == b"precious"       # csarc.cpp:474-483
    rc, _ = _add(csa, "no_data", overwrite=True)
    assert rc == 0 and (tmp_path / csa_cases.ARCNAME).read_bytes().hex() == GOLD["no_data"]["archive_hex"]


def test_archive_name_length_enters_the_index(csa, tmp_path, monkeypatch, orc_dec):
    """csa_indexpack.cpp:129-134 counts the archive name per task although it is not stored"""
    csa_cases.make_tree(str(tmp_path), "mixed_tree")
    monkeypatch.chdir(tmp_path)
    spec = csa_cases.CSA_CASES["mixed_tree"]
    sizes = {}
    for name in ("o.csa", "a_much_longer_archive_name.csa"):
        rc, st = csa.add(name, spec["args"], overwrite=True, **spec["opts"])
        assert rc == 0
        sizes[name] = st["index_raw_size"]
        info = orc_csa.parse((tmp_path / name).read_bytes(), orc_dec)
        assert len(info["index_raw"]) - info["index_used"] == st["n_tasks"] * (4 + len(name))
        assert info["index_raw"][info["index_used"]:] == bytes(len(info["index_raw"]) - info["index_used"])
        if HAVE_REF:
            subprocess.run([CSARC_REF] + [a if a != csa_cases.ARCNAME else "ref_" + name for a in csa_cases.csarc_argv("mixed_tree")],
                           cwd=tmp_path, check=True, capture_output=True)
            ref = (tmp_path / ("ref_" + name)).read_bytes()
            mine = (tmp_path / name).read_bytes()
            # "ref_" + name is 4 bytes longer: same body, index 4 bytes per task longer
            assert ref[24:int.from_bytes(ref[8:16], "little")] == mine[24:int.from_bytes(mine[8:16], "little")]
    assert sizes["a_much_longer_archive_name.csa"] - sizes["o.csa"] == 3 * (len("a_much_longer_archive_name.csa") - len("o.csa"))


# ---------------------------------------------------------------------------------------------
# csarc l / t / x
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["mixed_tree", "many_files", "single_split3", "chunk_edges_m5", "single_shadowed_by_empty", "no_data"])
def test_list_test_extract(csa, case, tmp_path, monkeypatch, orc_dec):
    content = csa_cases.make_tree(str(tmp_path), case)
    monkeypatch.chdir(tmp_path)
    assert _add(csa, case)[0] == 0
    arc = (tmp_path / csa_cases.ARCNAME).read_bytes()
    info = orc_csa.parse(arc, orc_dec)

    listed = csa.list_entries(csa_cases.ARCNAME)
    assert [e["name"] for e in listed] == sorted(info["index"].keys(), key=lambda s: s.encode("latin-1"))
    for e in listed:
        o = info["index"][e["name"]]
        assert (e["esize"], e["edate"], e["eattr"], e["frags"]) == (o["esize"], o["edate"], o["eattr"], o["frags"])
        assert e["edate"] == orc_csa.decimal_time(csa_cases.MTIME)
    assert csa.read_index(csa_cases.ARCNAME) == info["index_raw"]

    rc, st = csa.test(csa_cases.ARCNAME, mt_count=4)
    assert rc == 0 and st["verify_failures"] == 0 and st["n_tasks"] == len(info["abindex"])

    out = tmp_path / "out"
    rc, st = csa.extract(csa_cases.ARCNAME, to_dir=str(out), mt_count=3)
    assert rc == 0 and st["verify_failures"] == 0
    for rel, data in content.items():
        p = out / rel
        if rel not in info["index"]:
            assert not p.exists()
            continue
        got = p.read_bytes()
        stored = sum(f["size"] for f in info["index"][rel]["frags"])
        if stored == len(data):
            assert got == data, rel
        else:                                      # the shadowed single file: created empty, like the reference
            assert got == b""
        s = p.stat()
        assert int(s.st_mtime) == csa_cases.MTIME and stat.S_IMODE(s.st_mode) == 0o644
    for name in info["index"]:
        if name.endswith("/"):
            d = out / name
            assert d.is_dir() and stat.S_IMODE(d.stat().st_mode) == 0o755


def test_selection_by_name_and_wildcard(csa, tmp_path, monkeypatch):
    content = csa_cases.make_tree(str(tmp_path), "mixed_tree")
    monkeypatch.chdir(tmp_path)
    assert _add(csa, "mixed_tree")[0] == 0
    assert [e["name"] for e in csa.list_entries(csa_cases.ARCNAME, ["*.txt"])] == ["d/a.txt", "d/b.txt"]
    assert [e["name"] for e in csa.list_entries(csa_cases.ARCNAME, ["d/sub"])] == ["d/sub/", "d/sub/c.exe", "d/sub/e.dat"]
    out = tmp_path / "o2"
    rc, st = csa.extract(csa_cases.ARCNAME, ["d/sub/e.dat", "d/empty"], to_dir=str(out) + "/")
    assert rc == 0 and st["verify_failures"] == 0
    assert (out / "d/sub/e.dat").read_bytes() == content["d/sub/e.dat"] and (out / "d/empty").read_bytes() == b""
    assert not (out / "d/a.txt").exists() and not (out / "d/sub/c.exe").exists()
    rc, st = csa.test(csa_cases.ARCNAME, ["d/empty"])          # a task with nothing to verify: the reference would crash here
    assert rc == 0


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref/csarc_ref not in this snapshot")
def test_reads_archives_written_by_the_reference(csa, tmp_path, monkeypatch):
    content = csa_cases.make_tree(str(tmp_path), "many_files")
    argv = [a for a in csa_cases.csarc_argv("many_files")]
    argv.insert(1, "-t4")                          # four workers: blocks of different tasks interleave in the file
    subprocess.run([CSARC_REF] + argv, cwd=tmp_path, check=True, capture_output=True)
    monkeypatch.chdir(tmp_path)
    rc, st = csa.test(csa_cases.ARCNAME, mt_count=8)
    assert rc == 0 and st["verify_failures"] == 0
    out = tmp_path / "out"
    rc, st = csa.extract(csa_cases.ARCNAME, to_dir=str(out), mt_count=8)
    assert rc == 0 and st["verify_failures"] == 0
    for rel, data in content.items():
        assert (out / rel).read_bytes() == data, rel


def test_damaged_archives(csa, tmp_path, monkeypatch):
    csa_cases.make_tree(str(tmp_path), "mixed_tree")
    monkeypatch.chdir(tmp_path)
    assert _add(csa, "mixed_tree")[0] == 0
    arc = bytearray((tmp_path / csa_cases.ARCNAME).read_bytes())
    # header
    bad = bytearray(arc); bad[1] = ord("X")
    (tmp_path / "h.csa").write_bytes(bad)
    assert csa.test("h.csa")[0] == -1 and csa.extract("h.csa", to_dir=str(tmp_path / "x"))[0] == 1 and csa.list_entries("h.csa") is None
    assert csa.test("missing.csa")[0] == -1
    # a byte inside the first task's stream: either the decoder notices (-1) or a fragment fails its adler32
    bad = bytearray(arc); bad[24 + 5000] ^= 0x40
    (tmp_path / "b.csa").write_bytes(bad)
    rc, st = csa.test("b.csa")
    assert rc == -1 or st["verify_failures"] > 0
    # a wrong checksum in the index is reported, not fatal (csa_io.h:331-332)
    # truncated body: index position beyond the file
    (tmp_path / "t.csa").write_bytes(arc[:len(arc) // 2])
    assert csa.test("t.csa")[0] == -1


def test_task_bytes_gives_valid_archives_the_reference_reads(csa, tmp_path, monkeypatch, orc_dec):
    """CSAOptions.task_bytes (not in the reference): more, smaller tasks -- a different but valid archive"""
    content = csa_cases.make_tree(str(tmp_path), "mixed_tree")
    monkeypatch.chdir(tmp_path)
    spec = csa_cases.CSA_CASES["mixed_tree"]
    rc, st = csa.add("small.csa", spec["args"], overwrite=True, task_bytes=100000, **spec["opts"])
    assert rc == 0 and st["n_tasks"] >= 8
    arc = (tmp_path / "small.csa").read_bytes()
    assert cases.digest(arc) != GOLD["mixed_tree"]["archive_sha256"]
    info = orc_csa.parse(arc, orc_dec)
    assert max(len(e["frags"]) for e in info["index"].values()) >= 2          # some file spans tasks
    assert all(sum(f["size"] for f in info["index"][n]["frags"]) == len(d) for n, d in content.items())
    files, bad = orc_csa.extract(arc, orc_dec)                                 # the CPU oracle reads it
    assert not bad and all(files[n] == d for n, d in content.items())
    rc, st = csa.test("small.csa", mt_count=4)                                 # the product reads it
    assert rc == 0 and st["verify_failures"] == 0
    out = tmp_path / "o"
    assert csa.extract("small.csa", to_dir=str(out), mt_count=4)[0] == 0
    for n, d in content.items():
        assert (out / n).read_bytes() == d
    if HAVE_REF:                                                               # and so does the reference
        t = subprocess.run([CSARC_REF, "t", "small.csa"], cwd=tmp_path, capture_output=True)
        assert t.returncode == 0 and b"failed" not in t.stderr
        subprocess.run([CSARC_REF, "x", "-o", "r", "small.csa"], cwd=tmp_path, check=True, capture_output=True)
        for n, d in content.items():
            assert (tmp_path / "r" / n).read_bytes() == d
    # one huge file: never more than 127 fragments
    (tmp_path / "big.bin").write_bytes(cases.build([["text", 3, 0, 3 << 20]]))
    rc, st = csa.add("big.csa", ["big.bin"], overwrite=True, task_bytes=4096, level=1, dict_size=64 << 10)
    assert rc == 0 and st["n_tasks"] == 127
    rc, st = csa.test("big.csa", mt_count=16)
    assert rc == 0 and st["verify_failures"] == 0 and st["raw_bytes"] == 3 << 20


# ---------------------------------------------------------------------------------------------
# sharded Add: tasks dealt over ranks, blobs gathered, archive assembled on rank 0
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,world", [("mixed_tree", 2), ("many_files", 3), ("single_split3", 2), ("single_split_many", 8),
                                        ("single_shadowed_by_empty", 2), ("named_files_m3", 4)])
def test_sharded_add_equals_the_reference_archive(csa, case, world, tmp_path, monkeypatch):
    """every rank's shard encoded one after the other on the one GPU of this box, then assembled: the archive must be
    the reference's (the exchange itself is covered by the gloo tests)"""
    csa_cases.make_tree(str(tmp_path), case)
    monkeypatch.chdir(tmp_path)
    spec = csa_cases.CSA_CASES[case]
    opts = dict(spec["opts"], overwrite=True)
    blobs, ntasks = [], 0
    for r in range(world):
        rc, blob, st = csa.add_shard_encode(spec["args"], r, world, **opts)
        assert rc == 0
        blobs.append(blob)
        ntasks += st["n_tasks"]
    rc, st = csa.add_shard_assemble(csa_cases.ARCNAME, spec["args"], blobs[::-1], **opts)    # blob order must not matter
    assert rc == 0 and st["n_tasks"] == ntasks
    arc = (tmp_path / csa_cases.ARCNAME).read_bytes()
    assert cases.digest(arc) == GOLD[case]["archive_sha256"] and len(arc) == st["archive_bytes"]
    # a shard that is missing, or that belongs to another plan, is refused
    if ntasks > 1:
        rc, _ = csa.add_shard_assemble("bad.csa", spec["args"], blobs[:-1], **opts)
        assert rc == -1
    rc, _ = csa.add_shard_assemble("bad.csa", spec["args"], blobs + [blobs[0]], **opts)
    assert rc == -1 or ntasks == 0


SHARD_WORKER = """
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
import csa_cases
from csc_amd import sharded
torch.cuda.set_device(0)                       # both ranks share the one GPU of this box, so the transport is gloo
dist.init_process_group(backend="gloo")
spec = csa_cases.CSA_CASES[%r]
rc, st = sharded.add(csa_cases.ARCNAME, spec["args"], overwrite=True, **spec["opts"])
assert rc == 0, rc
print("RANK", dist.get_rank(), "tasks", st["n_tasks"], flush=True)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("case", ["many_files", "single_split_many"])
def test_sharded_add_two_processes(csa, case, tmp_path):
    """the whole multi-rank path, exchange included: two processes (gloo; one GPU between them) write the reference archive"""
    csa_cases.make_tree(str(tmp_path), case)
    script = tmp_path / "worker.py"
    script.write_text(SHARD_WORKER % (ROOT, ROOT, case))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29577", str(script)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert out.stdout.count("RANK") == 2
    arc = (tmp_path / csa_cases.ARCNAME).read_bytes()
    assert cases.digest(arc) == GOLD[case]["archive_sha256"]


def test_bench_multi_rank_path_two_processes(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver types it: the script starts its own two ranks (child torchrun), which run
    the -p8 tasks of the 10^9-byte stand-in (4 per rank, one batch launch per step), barrier + max-over-ranks timing, per-task
    digests against the reference's (tests/golden/split_prefix_digests.json) and the hand-over of the streams to rank 0.
    Two ranks share the one GPU of this box over gloo (CSC_BENCH_BACKEND, tests only; the driver's runs use RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(CSC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-split-sample-mib", "2"],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    # the CPU leg is on every line the driver can ask for: the reference's worker processes on rank 0's host cores
    cb = d["cpu_baseline"]
    assert cb and "error" not in cb and cb["value"] > 0 and cb["unit"] == "MB/s" and 1 <= cb["cores"] <= 8 and cb["kind"] in ("reference", "port")
    assert d["tasks_per_rank"] == [4, 4] and len(d["tasks"]) == 8
    assert d["bit_exact_vs_reference"] is True, d["tasks"]
    assert {t["rank"] for t in d["tasks"]} == {0, 1} and all(t["chunks"] == 2 for t in d["tasks"])
    assert d["exchange"]["ok"] is True and d["exchange"]["bytes"] > 0
    assert 0 < d["value"] < 1000 and d["roofline"]["launches"] == 1 and d["roofline"]["frac"] > 0


def test_bench_tree_workload_small_archive_equals_reference(tmp_path):
    """bench.py --workload tree_small end to end (128 tasks through CSA add on the GPU): the line says the archive is the one the
    REFERENCE archiver wrote for the tree (tests/golden/tree_workload.json)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, CSC_TREE_DIR=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tree_small", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["bit_exact_vs_reference"] is True and line["curve"] == "tree_small" and line["config"]["tasks"] == 128
    assert line["value"] > 0 and line["scaling"] == "strong"
    cb = line["cpu_baseline"]       # the reference archiver itself, timed on this box (oracle/_ref/csarc_ref -t8)
    assert cb and (("error" in cb and "csarc_ref" in cb["error"]) or (cb["value"] > 0 and cb["kind"] == "reference" and cb["cores"] >= 1))
