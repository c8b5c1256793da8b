"""CPU tier for the `.csa` container rows.

1. pins the container ORACLE (oracle/orc_csa.py + the liborc codec) against the reference archiver:
   byte-identical archives with oracle/_ref/csarc_ref when it is built, and with the digests that
   tools/make_golden_csa.py recorded from it (tests/golden/csa.json) always;
2. checks the product library's HOST pieces of the container (adler32, time stamps, header export)
   -- no GPU call is made here.
"""
import ctypes as C
import json
import os
import random
import re
import subprocess
import sys
import time
import zlib

import pytest

import cases
import csa_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc_csa  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "csa.json")))
CSARC_REF = os.path.join(ROOT, "oracle", "_ref", "csarc_ref")


@pytest.fixture(scope="module")
def codec(orc, zalloc):
    from csc_amd.capi import BytesWriter

    def enc(data, dict_size, level):
        w = BytesWriter()
        rc, s = orc.encode(data, props=orc.props_init(dict_size, level), alloc=zalloc, writer=w)
        assert rc == 0
        return s, [10] + w.sizes          # the caller's 10-byte header write comes first (csa_worker.cpp:42)

    def dec(stream):
        rc, raw = orc.decode(stream, alloc=zalloc)
        assert rc == 0
        return raw

    return enc, dec


# ---------------------------------------------------------------------------------------------
# adler32 / time stamps
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 5551, 5552, 5553, 65521, 200000, (2 << 20) + 3])
def test_adler32_oracle_and_product_host(prod, n):
    from csc_amd import csa
    data = cases.build([["random", 77, 0, n]]) if n else b""
    for seed in (0, 1, 0xFFF0FFF0 % (65521 << 16 | 65520), 0x12345678 & 0x7FFF7FFF):
        want = zlib.adler32(data, seed)             # zlib is what csa_adler32.cpp was cut from
        assert orc_csa.adler32(seed, data) == want
        assert csa.adler32(data, seed) == want
    assert orc_csa.adler32(0, b"\xff" * n) == zlib.adler32(b"\xff" * n, 0) == csa.adler32(b"\xff" * n, 0)


def test_adler32_is_foldable_piecewise():
    data = cases.build([["text", 5, 0, 100000]])
    a = 0
    for i in range(0, len(data), 7919):
        a = orc_csa.adler32(a, data[i:i + 7919])
    assert a == zlib.adler32(data, 0)


def test_decimal_time(prod):
    from csc_amd import csa
    L = csa.lib()
    rnd = random.Random(5)
    stamps = [0, 1, 59, 86399, 86400, 951782400, 951868800, 68169600, 68256000, csa_cases.MTIME, 4102444799] \
        + [rnd.randrange(0, 4102444800) for _ in range(3000)]
    for t in stamps:
        g = time.gmtime(t)                          # 1970..2099: every 4th year is leap, same calendar
        want = g.tm_year * 10**10 + g.tm_mon * 10**8 + g.tm_mday * 10**6 + g.tm_hour * 10**4 + g.tm_min * 100 + g.tm_sec
        assert orc_csa.decimal_time(t) == want
        assert L.CSA_DecimalTime(t) == want
        assert orc_csa.unix_time(want) == t
        assert L.CSA_UnixTime(want) == t
    assert orc_csa.decimal_time(-1) == L.CSA_DecimalTime(-1) == 19700101000000
    assert orc_csa.unix_time(0) == L.CSA_UnixTime(0) == -1
    # beyond 2099 both keep the reference's every-4th-year calendar
    for t in (4102444800 + 86400 * 59, 4102444800 + 86400 * 60, 5000000000, 2**33):
        assert orc_csa.decimal_time(t) == L.CSA_DecimalTime(t)


def test_exports_every_declared_container_symbol(prod):
    from csc_amd import csa
    hdr = open(os.path.join(ROOT, "include", "csa_mi355x.h")).read()
    names = set(re.findall(r"^[A-Za-z_][\w \*]*?\b(CSA(?:MI)?_\w+)\s*\(", hdr, flags=re.M))
    assert names == set(csa.SYMBOLS)
    for n in names:
        assert hasattr(prod.lib, n), n


def test_container_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "csa_mi355x.h"\nint main(void){CSAOptions o; CSA_OptionsInit(&o); return o.level == 2 ? 0 : 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)


def test_options_defaults(prod):
    from csc_amd import csa
    o = csa.CSAOptions()
    csa.lib().CSA_OptionsInit(C.byref(o))
    assert (o.level, o.dict_size, o.recurse, o.overwrite, o.mt_count, o.split_count, o.to_dir) == (2, 32000000, 0, 0, 1, 1, b"./")


def test_container_refuses_without_gpu(prod, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from csc_amd import csa
    (tmp_path / "f.txt").write_bytes(b"hello" * 100)
    rc, _ = csa.add(str(tmp_path / "o.csa"), [str(tmp_path / "f.txt")])
    assert rc == -95 and not (tmp_path / "o.csa").exists()      # CSCMI_DEVICE_ERROR, nothing written, no CPU path
    assert csa.test(str(tmp_path / "o.csa"))[0] == -95


# ---------------------------------------------------------------------------------------------
# std::sort restatement vs the real libstdc++
# ---------------------------------------------------------------------------------------------
SORT_PROBE = r"""
#include <algorithm>
#include <cstdio>
#include <vector>
struct E { long key; int id; };
int main() {
    int n, desc;
    while (scanf("%d %d", &n, &desc) == 2) {
        std::vector<E> v(n);
        for (int i = 0; i < n; i++) { scanf("%ld", &v[i].key); v[i].id = i; }
        if (desc) std::sort(v.begin(), v.end(), [](E a, E b) { return a.key > b.key; });
        else std::sort(v.begin(), v.end(), [](E a, E b) { return a.key < b.key; });
        for (int i = 0; i < n; i++) printf("%d ", v[i].id);
        printf("\n");
    }
}
"""


def test_std_sort_restatement_matches_libstdcxx(tmp_path):
    (tmp_path / "p.cpp").write_text(SORT_PROBE)
    subprocess.run(["g++", "-O2", "-o", str(tmp_path / "p"), str(tmp_path / "p.cpp")], check=True)
    rnd = random.Random(11)
    trials = []
    for n in [0, 1, 2, 15, 16, 17, 18, 31, 32, 33, 64, 100, 257, 954, 1000, 4096]:
        for nkeys in (1, 2, 3, 7, n + 1):
            for desc in (0, 1):
                trials.append((desc, [rnd.randrange(nkeys) for _ in range(n)]))
    # adversarial shapes for the depth limit / heapsort fallback
    for n in (200, 3000):
        trials += [(0, list(range(n))), (1, list(range(n))), (0, list(range(n, 0, -1))),
                   (0, [i % 2 * n + i for i in range(n)]), (0, ([0] * (n // 2) + list(range(n // 2))))]
        organ = list(range(n // 2)) + list(range(n // 2, 0, -1))
        trials.append((0, organ))
    inp = "".join(f"{len(k)} {d} " + " ".join(map(str, k)) + "\n" for d, k in trials)
    out = subprocess.run([str(tmp_path / "p")], input=inp, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(trials)
    for (desc, keys), line in zip(trials, out):
        ids = list(range(len(keys)))
        orc_csa.std_sort(ids, (lambda a, b: keys[a] > keys[b]) if desc else (lambda a, b: keys[a] < keys[b]))
        assert ids == [int(x) for x in line.split()], (desc, len(keys))


# ---------------------------------------------------------------------------------------------
# pieces
# ---------------------------------------------------------------------------------------------
def test_block_coalescer_rule():
    co = orc_csa.BlockCoalescer()
    for s in [10] + [1, 3, 65536] * 20:      # 10-byte header, then MemIO::WriteBlock's 1/[3]/payload triples
        co.write(s)
    blocks = co.finish()
    assert sum(blocks) == 10 + 20 * 65540
    assert blocks == [10 + 15 * 65540 + 1 + 3, 65536 + 4 * 65540]      # a block closes when the NEXT write would pass 1 MiB
    co = orc_csa.BlockCoalescer()
    co.write(1048576)
    co.write(1)
    assert co.finish() == [1048576, 1]
    assert orc_csa.BlockCoalescer().finish() == []


def test_index_round_trip():
    index = {"a/": {"edate": 20231114221320, "esize": 0, "eattr": ord("u") + (0o40755 << 8), "frags": []},
             "a/b.txt": {"edate": 20231114221320, "esize": 7, "eattr": ord("u") + (0o100644 << 8),
                         "frags": [{"bid": 0, "checksum": 0x1234, "posblock": 5, "size": 7, "posfile": 0}]}}
    ab = {0: [(24, 100), (124, 50)], 1: []}
    raw = orc_csa.pack_index(index, ab, "out.csa")
    i2, ab2, used = orc_csa.unpack_index(raw)
    assert ab2 == ab and {k: {f: v[f] for f in ("edate", "esize", "eattr", "frags")} for k, v in index.items()} == i2
    assert len(raw) - used == 2 * (4 + len("out.csa")) and raw[used:] == bytes(len(raw) - used)   # the over-counted, zero tail


def test_ispath():
    assert orc_csa.ispath("d", "d/a.txt") and orc_csa.ispath("d/", "d/a.txt") and orc_csa.ispath("d/a.txt", "d/a.txt")
    assert not orc_csa.ispath("d/a", "d/a.txt") and not orc_csa.ispath("d/a.txt", "d/a.txt2")
    assert orc_csa.ispath("*.txt", "d/a.txt") and orc_csa.ispath("d/?.txt", "d/a.txt") and not orc_csa.ispath("?", "")


# ---------------------------------------------------------------------------------------------
# whole archives
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", csa_cases.CPU_CASES)
def test_oracle_archive_matches_reference(case, codec, tmp_path, monkeypatch):
    enc, dec = codec
    content = csa_cases.make_tree(str(tmp_path), case)
    monkeypatch.chdir(tmp_path)
    spec = csa_cases.CSA_CASES[case]
    arc = orc_csa.create(csa_cases.ARCNAME, spec["args"], encode=enc, **spec["opts"])
    g = GOLD[case]
    assert len(arc) == g["archive_size"] and cases.digest(arc) == g["archive_sha256"]
    if "archive_hex" in g:
        assert arc.hex() == g["archive_hex"]
    if os.path.exists(CSARC_REF):
        subprocess.run([CSARC_REF] + csa_cases.csarc_argv(case), cwd=tmp_path, check=True, capture_output=True)
        assert (tmp_path / csa_cases.ARCNAME).read_bytes() == arc
    # and back: every selected file, verified fragment by fragment
    files, bad = orc_csa.extract(arc, dec)
    assert not bad
    info = orc_csa.parse(arc, dec)
    assert (info["index_pos"], info["index_csize"], info["index_rsize"]) == (g["index_pos"], g["index_csize"], g["index_rsize"])
    for name, data in files.items():
        if info["index"][name]["frags"] and sum(f["size"] for f in info["index"][name]["frags"]) == len(content[name]):
            assert data == content[name], name
    if case == "single_shadowed_by_empty":
        assert info["abindex"] == {} and info["index"]["q/a.bin"]["frags"] == []      # the reference stores nothing here


def test_tree_workload_is_deterministic_and_splits_per_extension():
    """csc_amd/treegen.py (bench.py --workload tree): names, sizes and kinds follow from the file number alone; two files per
    4-character extension, every group >= 64 KiB, so the multi-file split (csarc.cpp:545-557) makes one task per group"""
    from csc_amd import treegen
    for spec, (n, per, base, spread) in treegen.SPECS.items():
        fs = treegen.files(spec)
        assert len(fs) == n and fs == treegen.files(spec)
        exts = {}
        for rel, kind, seed, size in fs:
            assert base <= size < base + spread and kind in treegen.KINDS
            exts.setdefault(rel.rsplit(".", 1)[1], []).append(size)
        assert len(exts) == n // per and all(len(e) == 4 for e in exts) and all(sum(v) > 64 * 1024 for v in exts.values())
    assert treegen.total_bytes("tree") > 2 * 10 ** 9 and len(treegen.files("tree")) >= 4096


def test_oracle_archive_of_the_small_tree_matches_reference(codec, tmp_path, monkeypatch):
    """the container oracle over the 128-task tree == what csarc_ref a -r -m3 -d64m -t1 wrote (tests/golden/tree_workload.json)"""
    from csc_amd import treegen
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tree_workload.json")))["trees"]["tree_small"]
    enc, dec = codec
    assert treegen.materialize(str(tmp_path), "tree_small") == gold["input_bytes"]
    monkeypatch.chdir(tmp_path)
    arc = orc_csa.create("out.csa", ["t"], encode=enc, level=3, dict_size=64 << 20, recurse=True)
    import hashlib
    assert len(arc) == gold["archive_bytes"] and hashlib.sha256(arc).hexdigest() == gold["sha256"]
    info = orc_csa.parse(arc, dec)
    assert len(info["abindex"]) == 128          # one task per extension group
