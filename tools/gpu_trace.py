#!/usr/bin/env python3
"""Development aid: event times of the nodes of one DP window (-DCSCMI_TIMERS build).  gpurun -- python tools/gpu_trace.py <lib>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter
lib = CscLib(sys.argv[1])
data = corpus.fill("text", corpus.SEED_ENWIK9, 0, 2 << 20).tobytes()
p = lib.props_init(64 << 20, 3)
w = BytesWriter()
h = lib.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
lib.lib.CSCMI_EncodeHostChunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
lib.lib.CSCMI_EncodeHostChunk(h, data, len(data))
lib.lib.CSCMI_DebugTrace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
tr = (C.c_uint64 * 768)(); lib.lib.CSCMI_DebugTrace(h, tr)
rows = [[tr[k * 12 + e] for e in range(12)] for k in range(64)]
t0 = min(r[0] for r in rows if r[0])
names = ["start", "gather", "insdone", "labels", "reps", "final", "crit", "erep", "dec", "priced", "relaxed"]
print("node " + " ".join(f"{n:>8s}" for n in names))
for k, r in enumerate(rows):
    if not r[0]: continue
    print(f"{k:4d} " + " ".join(f"{(v - t0) if v else -1:8d}" for v in r[:11]))
