#!/usr/bin/env python3
"""BASELINE.json configs[1] at FULL size, once: the whole 10^9-byte enwik9 stand-in as ONE `-m3 -d64m` stream through the HIP
encoder; {stream_bytes, sha256} against what the REFERENCE wrote for the same bytes (tests/golden/multi_stream_digests.json,
split "1", recorded from oracle/_ref by tools/make_golden_multi.py).  Writes gpurun_out/<dir>/fullsize.json; the committed
copy is profiles/r02_fullsize_single_stream.json.   gpurun --timeout 2400 -- python tools/gpu_fullsize.py gpurun_out/r02_full"""
import ctypes as C, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import csc_amd
from csc_amd import corpus
from csc_amd.capi import BytesWriter
out_dir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fullsize"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
os.makedirs(out_dir, exist_ok=True)
lib = csc_amd.load(); L = lib.lib
L.CSCMI_EncodeDeviceChunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.CSCMI_EncodeDeviceChunk.restype = C.c_int
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "multi_stream_digests.json")))["splits"]["1"]["tasks"][0]
props = lib.props_init(64 << 20, 3)
w = BytesWriter()
h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
w.out += lib.write_properties(props)
chunk = 2 << 20
piece = 256 << 20
t0 = time.time(); tenc = 0.0
marks = []
for base in range(0, total, piece):
    n = min(piece, total - base)
    dev = torch.from_numpy(corpus.fill("text", corpus.SEED_ENWIK9, base, n)).cuda()
    torch.cuda.synchronize()
    t1 = time.time()
    for off in range(0, n, chunk):
        rc = L.CSCMI_EncodeDeviceChunk(h, C.c_void_p(dev.data_ptr() + off), min(chunk, n - off))
        assert rc == 0, rc
    torch.cuda.synchronize()
    tenc += time.time() - t1
    marks.append({"input_bytes": base + n, "stream_bytes": len(w.out), "encode_seconds": round(tenc, 1)})
    print(marks[-1], flush=True)
    del dev
rc = L.CSCEnc_Encode_Flush(h); assert rc == 0
L.CSCEnc_Destroy(h)
s = bytes(w.out)
res = {"what": f"enwik9 stand-in, {total} B, -m3 -d64m, ONE stream, HIP encoder (chunks resident in HBM)",
       "input_bytes": total, "stream_bytes": len(s), "sha256": hashlib.sha256(s).hexdigest(),
       "encode_seconds": round(tenc, 1), "MBps": round(total / 1e6 / tenc, 4), "progress": marks,
       "reference": gold if total == 10 ** 9 else None,
       "bit_exact_vs_reference": (len(s) == gold["stream_bytes"] and hashlib.sha256(s).hexdigest() == gold["sha256"]) if total == 10 ** 9 else None}
json.dump(res, open(os.path.join(out_dir, "fullsize.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "progress"}))
sys.exit(0 if res["bit_exact_vs_reference"] in (True, None) else 1)
