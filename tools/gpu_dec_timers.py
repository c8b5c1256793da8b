#!/usr/bin/env python3
"""Development aid: decode one oracle-made stream with the -DCSCMI_TIMERS decode kernel and print where the
wavefront's cycles go.  gpurun -- python tools/gpu_dec_timers.py [level] [MiB] [kind]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesReader, BytesWriter, CSC_PROP_SIZE
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
lib = CscLib(os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
rc, s = orc.encode(data, level, 16 << 20, alloc=orc.lib.orc_zero_alloc())
props = lib.read_properties(s[:CSC_PROP_SIZE])
r = BytesReader(s[CSC_PROP_SIZE:]); w = BytesWriter()
h = lib.lib.CSCDec_Create(C.byref(props), C.cast(r.ptr(), C.c_void_p), None)
t0 = time.time(); rc = lib.lib.CSCDec_Decode(h, C.cast(w.ptr(), C.c_void_p), None); dt = time.time() - t0
lib.lib.CSCMI_DebugDecTimers.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
tm = (C.c_uint64 * 16)(); lib.lib.CSCMI_DebugDecTimers(h, tm)
print(f"level {level} {kind}: rc={rc} ok={bytes(w.out) == data} {len(data)/1e6/dt:.2f} MB/s, {len(s)} -> {len(data)} B, wall {dt:.2f}s")
names = ["literal packet decode", "match packet decode", "window write / copy", "run copy-out", "inverse filter (POST)", "", "", "whole kernel"]
for i, n in enumerate(names):
    if n: print(f"  {n:24s} {tm[i]/1e6:9.1f} Mcyc  n={tm[8+i]:>9d}  {tm[i]/max(1,tm[8+i]):8.0f} cyc each")
print(f"  launches {tm[15]}, kernel cycles/s if 2.4 GHz: {tm[7]/2.4e9:.2f}s")
