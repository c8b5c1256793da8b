#!/usr/bin/env python3
"""Development aid: the hash-bucket inserter form (csc_kernels_hp.inc, levels 1 / 2) against the oracle.
usage: gpu_hp.py [lib]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib
import cases
lib = CscLib(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"))
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p; za = orc.lib.orc_zero_alloc()
bad = 0
def check(name, data, props_of):
    global bad
    t0 = time.time(); rc, s = lib.encode(data, props=props_of(lib)); dt = time.time() - t0
    rc2, want = orc.encode(data, props=props_of(orc), alloc=za)
    ok = (rc, s) == (rc2, want)
    bad += not ok
    k = next((i for i in range(min(len(s), len(want))) if s[i] != want[i]), None)
    print(f"{name:32s} {len(data):9d} B -> {len(s):8d} (want {len(want)}) rc={rc} {'OK ' if ok else 'MISMATCH at %s' % k} {len(data)/1e6/max(dt,1e-9):.3f} MB/s", flush=True)
for name in ("zeros_8k", "abcdefgh_64k", "text_20k", "text_300k", "exe_300k", "mix_types", "dup_blocks", "ragged_tail_511", "short_reads_511",
             "window_wrap_32k", "periodic_5000x200", "delta_200k", "zeros_5m"):
    spec, d, clamp, _ = cases.STREAM_CASES[name]
    data = cases.build(spec)
    dd = min(d, max(len(data), 1)) if clamp else d
    for lv in (1, 2):
        check(f"{name} m{lv}", data, lambda L: L.props_init(dd, lv))
data = cases.build([["text", 13, 0, 500000], ["exe", 14, 0, 200000], ["pattern", "00", 70000], ["delta", 3, 0, 100000], ["text", 13, 0, 100000]])
for width, bits, good, mode, dsz in ((1, 16, 32, 2, 1 << 20), (3, 12, 8, 2, 1 << 18), (8, 10, 24, 1, 40000), (5, 18, 200, 2, 1 << 21), (8, 20, 24, 2, 1 << 22)):
    def mk(L):
        p = L.props_init(dsz, 2); p.hash_width = width; p.hash_bits = bits; p.good_len = good; p.lz_mode = mode
        return p
    check(f"custom w{width} b{bits} good{good} mode{mode}", data, mk)
seg = 64 << 20
d5 = (corpus.fill("mix5", corpus.SEED_EXE, seg - (1 << 20), 2 << 20).tobytes() + corpus.fill("mix5", corpus.SEED_EXE, 2 * seg - (1 << 20), 2 << 20).tobytes())
check("config5 geometry 4 MiB", d5, lambda L: L.props_init(1 << 30, 2))
d5b = corpus.fill("mix5", corpus.SEED_EXE, 0, 4 << 20).tobytes()
check("config5 geometry, first 4 MiB", d5b, lambda L: L.props_init(1 << 30, 2))
d1 = corpus.fill("text", corpus.SEED_ENWIK9, 0, 8 << 20).tobytes()
check("text 8 MiB m1 -d64m", d1, lambda L: L.props_init(64 << 20, 1))
check("text 8 MiB m2 -d64m", d1, lambda L: L.props_init(64 << 20, 2))
print("FAILED" if bad else "ALL OK")
