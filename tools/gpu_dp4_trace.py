#!/usr/bin/env python3
"""Development aid (-DCSCMI_TIMERS build): cycle stamps of the spine and the edge wavefronts over the nodes of ONE DP window of
the level-3 pipeline form.  gpurun -- python tools/gpu_dp4_trace.py [window numbers ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter
lib = CscLib(os.environ.get("CSC_DEV_LIB") or os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))      # (the node trace needs the -DCSCMI_TIMERS_FINE build: CSC_DEV_LIB=csc_amd/csrc/build/dev_fine/libcsc_mi355x_timers.so)
kind = os.environ.get("KIND", "text")
data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, 2 << 20).tobytes()
p = lib.props_init(64 << 20, 3)
lib.lib.CSCMI_EncodeHostChunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
lib.lib.CSCMI_DebugSetMask.argtypes = [C.c_void_p, C.c_uint64]
lib.lib.CSCMI_DebugTrace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
for win in [int(a) for a in sys.argv[1:]] or [3000]:
    w = BytesWriter()
    h = lib.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
    lib.lib.CSCMI_DebugSetMask(h, win << 32)
    lib.lib.CSCMI_EncodeHostChunk(h, data, len(data))
    tr = (C.c_uint64 * 768)(); lib.lib.CSCMI_DebugTrace(h, tr)
    rows = [[tr[(4 + k) * 12 + e] for e in range(12)] for k in range(50)]
    rows = [r for r in rows if r[0] or r[4]]
    if not rows: print(f"window {win}: nothing"); continue
    t0 = min(v for r in rows for v in (r[0], r[4]) if v)
    print(f"window {win}: {len(rows)} nodes.  spine (straight-line step of node k): step top, re-poll ended (- = the edges were not late), label k+1 out, own edge won | edge (node k): record looked at, label k seen, done, path (1 straight 2 rep 3 general) | step length, label k out -> edge sees it, edge work, edge done -> the step that needs it begins")
    f = lambda v: f"{v - t0:7d}" if v else "      -"
    for k, r in enumerate(rows):
        step = r[2] - r[0] if r[2] and r[0] else 0
        prev_out = rows[k - 1][2] if k else 0
        notice = r[5] - prev_out if r[5] and prev_out else 0
        work = r[6] - r[5] if r[6] and r[5] else 0
        slack = rows[k + 1][0] - r[6] if k + 1 < len(rows) and rows[k + 1][0] and r[6] else 0
        print(f"{k:3d}  {f(r[0])} {f(r[1])} {f(r[2])} {r[8]} | {f(r[4])} {f(r[5])} {f(r[6])} {r[7]} | {step:5d} {notice:5d} {work:5d} {slack:6d}")
