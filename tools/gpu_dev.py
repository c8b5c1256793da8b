#!/usr/bin/env python3
"""Development aid: a -DCSCMI_DEV_M3ONLY build (only the level-3/4 hash-table kernels) against the oracle.
usage: gpu_dev.py [lib] [MiB]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib
import cases
lib = CscLib(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "csc_amd", "libcsc_mi355x_dev.so"))
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p; za = orc.lib.orc_zero_alloc()
bad = 0
def check(name, data, level, dict_size):
    global bad
    t0 = time.time(); rc, s = lib.encode(data, props=lib.props_init(dict_size, level)); dt = time.time() - t0
    rc2, want = orc.encode(data, props=orc.props_init(dict_size, level), alloc=za)
    ok = (rc, s) == (rc2, want)
    bad += not ok
    print(f"{name:28s} m{level} {len(data):9d} B -> {len(s):8d} rc={rc} {'OK ' if ok else 'MISMATCH'} {len(data)/1e6/max(dt,1e-9):.3f} MB/s", flush=True)
check("text", corpus.fill("text", corpus.SEED_ENWIK9, 0, mib << 20).tobytes(), 3, 64 << 20)
for name in ("empty", "one_byte", "zeros_8k", "abcdefgh_64k", "text_20k", "text_300k", "exe_300k", "mix_types", "dup_blocks", "ragged_tail_511",
             "window_wrap_32k", "periodic_5000x200", "delta_200k"):
    spec, d, clamp, _ = cases.STREAM_CASES[name]
    data = cases.build(spec)
    for lv in (3, 4):
        check(name, data, lv, min(d, max(len(data), 1)) if clamp else d)
print("FAILED" if bad else "ALL OK")
