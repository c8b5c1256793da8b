"""CPU tier: the hardened corners of the container layer (index pack / unpack, fragment bound, unsafe names), the
sanitizer build of the oracle, and bench.py's rank plan -- nothing here needs a GPU."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys

import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def raw_index(entries, blocks=()):
    """csa_indexpack.cpp:166-189 by hand: entries = [(name bytes, esize, [frag tuples (bid, checksum, posblock, size, posfile)])]"""
    out = struct.pack("<I", len(entries))
    for name, esize, frags in entries:
        out += struct.pack("<I", len(name)) + name + struct.pack("<qqq", 20240101120000, esize, 0o644) + struct.pack("<B", len(frags) & 0xFF)
        for f in frags:
            out += struct.pack("<IIQQQ", *f)
    out += struct.pack("<I", len(blocks))
    for bid, ext in blocks:
        out += struct.pack("<QI", bid, len(ext))
        for off, size in ext:
            out += struct.pack("<QQ", off, size)
    return out


def frags(n):
    return [(i, 0x1234 + i, 0, 10, 10 * i) for i in range(n)]


def test_index_round_trip_127_fragments_ok_128_refused():
    from csc_amd import csa
    ok = raw_index([(b"a/file.bin", 1270, frags(127))], [(0, [(24, 100)])])
    n, back = csa.index_round_trip(ok)
    assert n > 0 and back[:len(ok)] == ok                    # (re-packed buffer = same fields + the reference's zero tail)
    bad = raw_index([(b"a/file.bin", 1280, frags(128))], [(0, [(24, 100)])])
    assert csa.index_round_trip(bad)[0] == -1                # the count byte reads back as int8 -128: unreadable, and said so


@pytest.mark.parametrize("name", [b"", b"../evil", b"a/../../evil", b"a\\..\\evil", b".."])
def test_index_unsafe_names(name):
    from csc_amd import csa
    raw = raw_index([(name, 10, frags(1))])
    assert csa.index_round_trip(raw)[0] == csa.CSA_UNSAFE_NAME


@pytest.mark.parametrize("name", [b"a..b", b"...", b"/abs/path", b"c:/x", b"dir/", b"x/..y/z"])
def test_index_safe_names(name):
    from csc_amd import csa
    assert csa.index_round_trip(raw_index([(name, 10, frags(1))]))[0] > 0


def test_index_fuzz_truncations_and_flips():
    """every truncation and a few thousand byte flips of a valid index: parse or refuse, never crash"""
    import random
    from csc_amd import csa
    good = raw_index([(b"dir/", 0, []), (b"dir/one.txt", 100, frags(3)), (b"two", 5, frags(1))], [(0, [(24, 50), (74, 9)]), (1, [])])
    assert csa.index_round_trip(good)[0] > 0
    for cut in range(len(good)):
        assert csa.index_round_trip(good[:cut])[0] == -1
    rnd = random.Random(7)
    for _ in range(3000):
        b = bytearray(good)
        for _ in range(rnd.randint(1, 4)):
            b[rnd.randrange(len(b))] = rnd.randrange(256)
        csa.index_round_trip(bytes(b))       # any verdict; the point is that it returns


def test_plan_refuses_more_than_127_fragments(tmp_path):
    """split_count together with task_bytes used to allow 127 pieces per SLICE"""
    from csc_amd import csa
    f = tmp_path / "big.bin"
    with open(f, "wb") as fh:
        fh.truncate(40 << 20)                # sparse 40 MiB
    rc, nt, mf = csa.plan_info([str(f)], split_count=8)
    assert rc == 0 and nt == 8 and mf == 8
    rc, nt, mf = csa.plan_info([str(f)], split_count=8, task_bytes=1 << 16)      # wants 640 pieces
    assert rc == 0 and mf <= csa.CSA_MAX_FRAGMENTS, (rc, nt, mf)
    rc, nt, mf = csa.plan_info([str(f)], split_count=200)                        # 1 MiB + 4 slices -> 40 fragments
    assert rc == 0 and mf == 40
    g = tmp_path / "huge.bin"
    with open(g, "wb") as fh:
        fh.truncate(200 << 20)               # sparse 200 MiB, -p190 -> 190 slices > 127: must be refused, nothing written
    rc, nt, mf = csa.plan_info([str(g)], split_count=190)
    assert rc == csa.CSA_TOO_MANY_FRAGMENTS and mf > 127


def test_oracle_under_address_sanitizer(tmp_path):
    """SURVEY section 4.1 'sanitizers' tier, CPU only: the plain-C restatement built with -fsanitize=address,undefined
    encodes + decodes the small golden cases at every level in a child process (ASan must be the first DSO)"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liborc_asan.so"], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan):
        pytest.skip("libasan.so not installed")
    script = tmp_path / "run.py"
    script.write_text(f"""
import ctypes as C, json, os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
import cases
from csc_amd.capi import CscLib
orc = CscLib(os.path.join({ROOT!r}, "oracle", "liborc_asan.so"))
orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()
G = json.load(open(os.path.join({ROOT!r}, "tests", "golden", "streams.json")))
n = 0
for name in ("empty", "one_byte", "zeros_8k", "abcdefgh_64k", "random_64k", "text_20k", "mix_types", "dup_blocks", "ragged_tail_511",
             "short_reads_511", "window_wrap_32k", "delta_200k"):
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    for lv in cases.levels_for(name):
        rc, s = orc.encode(data, lv, dict_size, alloc=za, clamp_dict=clamp, max_read=max_read)
        assert rc == 0 and cases.digest(s) == G[f"{{name}}/m{{lv}}"]["stream_sha256"], (name, lv)
        rcd, back = orc.decode(s, alloc=za)
        assert rcd == 0 and back == data
        for cut in (len(s) // 2, len(s) - 3):            # damaged input through the sanitized decoder
            orc.decode(s[:cut], alloc=za)
        n += 1
print("ASAN_OK", n)
""")
    env = dict(os.environ, LD_PRELOAD=" ".join(p for p in (asan, ubsan) if os.path.exists(p)),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "ASAN_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr, out.stderr[-3000:]


# ---- bench.py: launch plan (the encode itself needs a GPU: tests/test_gpu_csa.py drives `python bench.py --gpus 2` there) ----
def test_bench_split_plan_is_the_same_at_every_n():
    from csc_amd import corpus, tasks
    sl = corpus.task_slices(10 ** 9, 8)
    assert len(sl) == 8 and sl[0] == (0, 125000004) and sl[7] == (875000028, 124999972)
    for world in (1, 2, 4, 8):
        a = tasks.assign(sl, world)
        assert sorted(t for r in a for t in r) == list(range(8)) and {len(r) for r in a} == {8 // world}
    sl = corpus.task_slices(10 ** 9, 127)
    assert [len(r) for r in tasks.assign(sl, 8)] == [16] * 7 + [15]


def test_bench_golden_prefix_digests_shape():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "split_prefix_digests.json")))
    assert g["level"] == 3 and g["dict"] == 64 << 20 and sorted(g["splits"]) == ["1", "8"]
    for t in range(8):
        rec = g["splits"]["8"][str(t)]
        assert len(rec) == 48 and rec["1"]["input_bytes"] == 2 << 20 and rec["48"]["input_bytes"] == 48 * (2 << 20)
        assert all(rec[str(c)]["stream_bytes"] < rec[str(c + 1)]["stream_bytes"] for c in range(1, 48))
    m = json.load(open(os.path.join(ROOT, "tests", "golden", "multi_stream_digests.json")))
    assert m["splits"]["1"]["tasks"][0]["input_bytes"] == 10 ** 9 and len(m["splits"]["8"]["tasks"]) == 8


def test_bench_spawns_its_own_ranks_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` with no RANK in the environment must start a child torchrun (not import torch, not exec).
    Here the child dies at once (no GPU in this container) -- what is checked is that the parent launched two ranks through
    torch.distributed.run, relayed the failure as a non-zero exit code, and never initialised anything itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0 and out.stdout.strip().startswith("{")
    else:
        assert out.returncode != 0
        assert "bench.py needs a GPU" in out.stderr and "2-rank child failed" in out.stderr, out.stderr[-2000:]
