#!/usr/bin/env python3
"""Development aid: the binary-tree inserter form (csc_kernels_bt.inc, level 5) against the oracle.
usage: gpu_bt.py [lib]"""
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
    print(f"{name:28s} {len(data):9d} B -> {len(s):8d} (want {len(want)}) rc={rc} {'OK ' if ok else 'MISMATCH at %s' % k} {len(data)/1e6/max(dt,1e-9):.3f} MB/s", flush=True)
for name in ("zeros_8k", "abcdefgh_64k", "text_20k", "text_300k", "exe_300k", "mix_types", "dup_blocks", "ragged_tail_511", "short_reads_511",
             "window_wrap_32k", "periodic_5000x200", "delta_200k", "silesia_like_3m", "zeros_5m"):
    spec, d, clamp, _ = cases.STREAM_CASES[name]
    data = cases.build(spec)
    dd = min(d, max(len(data), 1)) if clamp else d
    check(name, data, lambda L: L.props_init(dd, 5))
data = cases.build([["text", 13, 0, 500000], ["exe", 14, 0, 200000], ["silesia", 6, 1 << 20, 600000], ["pattern", "00", 70000], ["text", 13, 0, 100000]])
for bt_size, cyc, good, dsz in ((40000, 32, 48, 1 << 20), (100000, 4, 16, 1 << 18), (1 << 20, 32, 200, 1 << 21), (300000, 16, 8, 300000)):
    def mk(L):
        p = L.props_init(dsz, 5); p.bt_size = bt_size; p.bt_cyc = cyc; p.good_len = good
        return p
    check(f"custom bt{bt_size} cyc{cyc} good{good}", data, mk)
d3 = cases.build([["silesia", 6, 0, 2 << 20], ["silesia", 6, 90 << 20, 1 << 20]])
check("config3 geometry 3 MiB", d3, lambda L: L.props_init(211957760, 5))
print("FAILED" if bad else "ALL OK")
