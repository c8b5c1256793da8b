#!/usr/bin/env python3
"""Development probe (the -DCSCMI_TIMERS build): find the first window the split parse gets wrong -- windows below a stream
position go through the split, the rest through the lone parse; bisect on the position.
gpurun -- python tools/gpu_dp4_bisect2.py MiB byte_offset lo hi [rounds]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter
mib, boff, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 10
lib = CscLib(os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so"))
orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()
data = corpus.fill("text", corpus.SEED_ENWIK9, boff, mib << 20).tobytes()
p = lib.props_init(64 << 20, 3)
rc2, want = orc.encode(data, props=p, alloc=za)
lib.lib.CSCMI_EncodeHostChunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
lib.lib.CSCMI_DebugSetMask.argtypes = [C.c_void_p, C.c_uint64]
def run(thr, extra=0):
    w = BytesWriter()
    h = lib.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
    for off in range(0, len(data), 2 << 20):
        lib.lib.CSCMI_DebugSetMask(h, ((thr // 16) << 8) | extra)
        lib.lib.CSCMI_EncodeHostChunk(h, data[off:off + (2 << 20)], min(2 << 20, len(data) - off))
    got = bytes(w.out)
    base = want.find(got[:64]); wn = want[base:] if base >= 0 else want
    return got == wn[:len(got)], h
for r in range(rounds):
    mid = (lo + hi) // 2
    ok, _ = run(mid)
    print(f"split below {mid}: {'exact' if ok else 'DIFF'}", flush=True)
    if ok: lo = mid
    else: hi = mid
print(f"first bad window in [{lo}, {hi})")
