#!/usr/bin/env python3
"""BASELINE configs[2] / task 0 of configs[4] at FULL size through the HIP encoder, once: {stream_bytes, sha256} against what the
REFERENCE wrote for the same bytes (tests/golden/fullsize_digests.json, recorded from oracle/_ref by tools/make_golden_fullsize.py).
gpurun --timeout 3000 -- python tools/gpu_fullsize_cfg.py silesia_m5_d256m gpurun_out/r03_full   (copy the json to profiles/)"""
import ctypes as C, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import csc_amd
from csc_amd import corpus
from csc_amd.capi import BytesWriter
name = sys.argv[1]
out_dir = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/fullsize"
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 0            # (development: stop after this many bytes, no comparison)
os.makedirs(out_dir, exist_ok=True)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))["cases"][name]
src = corpus.Source(gold["corpus"])
total = limit or gold["input_bytes"]
lib = csc_amd.load(); L = lib.lib
L.CSCMI_EncodeDeviceChunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.CSCMI_EncodeDeviceChunk.restype = C.c_int
props = lib.props_init(min(gold["dict"], gold["input_bytes"]), gold["level"])
assert int(props.dict_size) == gold["dict_size"] and int(props.bt_size) == gold["bt_size"]
w = BytesWriter()
h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
assert h
w.out += lib.write_properties(props)
chunk, piece = 2 << 20, 128 << 20
tenc, marks = 0.0, []
for base in range(0, total, piece):
    n = min(piece, total - base)
    dev = torch.from_numpy(src.read(gold["offset"] + base, n)).cuda()
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
full = total == gold["input_bytes"]
res = {"what": f"{gold['corpus']} stand-in bytes {gold['offset']}..{gold['offset'] + total}, -m{gold['level']} -d{gold['dict'] >> 20}m, ONE stream, HIP encoder (chunks resident in HBM)",
       "dict_size": gold["dict_size"], "bt_size": gold["bt_size"], "input_bytes": total, "stream_bytes": len(s), "sha256": hashlib.sha256(s).hexdigest(),
       "encode_seconds": round(tenc, 1), "MBps": round(total / 1e6 / tenc, 4), "progress": marks, "reference": gold if full else None,
       "bit_exact_vs_reference": (len(s) == gold["stream_bytes"] and hashlib.sha256(s).hexdigest() == gold["sha256"]) if full else None}
json.dump(res, open(os.path.join(out_dir, f"fullsize_{name}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "progress"}))
sys.exit(0 if res["bit_exact_vs_reference"] in (True, None) else 1)
