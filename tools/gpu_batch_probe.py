#!/usr/bin/env python3
"""Development probe: one CSCMI_EncodeDeviceChunkBatch launch over n small independent streams (the archiver's many-task regime)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, csc_amd
from csc_amd import corpus
from csc_amd.capi import BytesWriter
lib = csc_amd.load(); L = lib.lib
L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
L.CSCMI_EncodeDeviceChunkBatch.restype = C.c_int
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for kinds in (("text",), ("exe",), ("delta",), ("silesia",), ("text", "exe", "delta", "silesia")):
    for n in (8, 128):
        size = 262144
        hs, ws, devs = [], [], []
        for i in range(n):
            props = lib.props_init(size, level)
            w = BytesWriter(); h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
            hs.append(h); ws.append(w)
            devs.append(torch.from_numpy(corpus.fill(kinds[i % len(kinds)], 5000 + i, 0, size)).cuda())
        torch.cuda.synchronize()
        t0 = time.time()
        rc = L.CSCMI_EncodeDeviceChunkBatch(n, (C.c_void_p * n)(*hs), (C.c_void_p * n)(*[d.data_ptr() for d in devs]), (C.c_size_t * n)(*[size] * n))
        torch.cuda.synchronize()
        dt = time.time() - t0
        for h in hs:
            L.CSCEnc_Encode_Flush(h); L.CSCEnc_Destroy(h)
        print(f"m{level} {'+'.join(kinds):24s} n={n:4d} x {size} B: {dt:7.3f} s = {n*size/1e6/dt:8.2f} MB/s rc={rc}", flush=True)
