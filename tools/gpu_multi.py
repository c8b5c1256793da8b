#!/usr/bin/env python3
"""Many independent task streams on ONE GPU through CSCMI_EncodeDeviceChunkBatch.
usage: gpu_multi.py <streams> <chunks_per_stream> [level] [dict] [check]
The input is the enwik9 stand-in split like `csarc -p<streams>` (csarc.cpp:532-543)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import csc_amd
from csc_amd import corpus
from csc_amd.capi import BytesWriter, CscLib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dict_size = int(sys.argv[4]) << 20 if len(sys.argv) > 4 else 64 << 20
check = len(sys.argv) > 5
lib = csc_amd.load(); L = lib.lib
L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
slices = corpus.task_slices(10 ** 9, S)[:S]
chunk = 2 << 20
hs, ws, devs, datas = [], [], [], []
t0 = time.time()
for off, n in slices:
    props = lib.props_init(min(dict_size, n), level)
    w = BytesWriter()
    h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
    assert h, "create failed"
    w.out += lib.write_properties(props)
    nb = min(n, K * chunk)
    d = corpus.fill("text", corpus.SEED_ENWIK9, off, nb)
    hs.append(h); ws.append(w); datas.append(d); devs.append(torch.from_numpy(d).cuda())
torch.cuda.synchronize()
print(f"{S} streams created in {time.time()-t0:.1f}s, mem {torch.cuda.mem_get_info()[0]/2**30:.1f} GiB free", flush=True)
H = (C.c_void_p * S)(*hs)
t0 = time.time(); total = 0
for k in range(K):
    P = (C.c_void_p * S)(*[d.data_ptr() + k * chunk for d in devs])
    Z = (C.c_size_t * S)(*[max(0, min(chunk, len(d) - k * chunk)) for d in datas])
    rc = L.CSCMI_EncodeDeviceChunkBatch(S, H, P, Z)
    assert rc == 0, rc
    total += sum(Z)
    print(f"  step {k}: {total/1e6/(time.time()-t0):.2f} MB/s aggregate so far", flush=True)
dt = time.time() - t0
out = sum(len(w.out) for w in ws)
print(f"RESULT streams={S} level={level} chunks={K} bytes={total} time={dt:.2f}s aggregate={total/1e6/dt:.3f} MB/s ratio={out/total:.4f}")
for h in hs:
    L.CSCEnc_Encode_Flush(h); L.CSCEnc_Destroy(h)
if check:
    orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
    za = orc.lib.orc_zero_alloc()
    for i in (0, S // 2, S - 1):
        off, n = slices[i]
        rc, want = orc.encode(datas[i].tobytes(), props=orc.props_init(min(dict_size, n), level), alloc=za)
        print(f"  stream {i}: bit-exact vs oracle: {bytes(ws[i].out) == want}")
