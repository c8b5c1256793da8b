#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE ITSELF (oracle/_ref/libcsc_ref.so, built by
oracle/Makefile from /root/reference/src).  Run in the development container only; the vectors
(data: sizes, digests, small streams in hex) are committed, the reference never travels.

  python tools/make_golden.py
"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from csc_amd.capi import CscLib, CSCProps

ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
orc = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
orc.orc_zero_alloc.restype = C.c_void_p
za = orc.orc_zero_alloc()            # zero-filling ISzAlloc: makes the reference deterministic
G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

# ---- whole streams -------------------------------------------------------------------------
streams = {}
for name, (spec, dict_size, clamp, max_read) in cases.STREAM_CASES.items():
    data = cases.build(spec)
    for level in cases.levels_for(name):
        rc, s = ref.encode(data, level, dict_size, alloc=za, clamp_dict=clamp, max_read=max_read)
        assert rc == 0
        rcd, back = ref.decode(s, alloc=za)
        assert rcd == 0 and back == data, (name, level)
        ent = {"input_size": len(data), "input_sha256": cases.digest(data), "stream_size": len(s), "stream_sha256": cases.digest(s)}
        if len(s) <= 1200:
            ent["stream_hex"] = s.hex()
        streams[f"{name}/m{level}"] = ent
        print(name, level, len(data), "->", len(s), flush=True)
json.dump(streams, open(os.path.join(G, "streams.json"), "w"), indent=1, sort_keys=True)

# ---- CSCEncProps_Init / EstMemUsage / WriteProperties ------------------------------------------
props = {}
for level in range(0, 7):
    for d in (0, 1000, 32768, 100000, (1 << 20) - 10240, 1 << 20, 4 << 20, 16 << 20, 64 << 20, 64000000, 125000004,
              211957760, 256 << 20, 1 << 30, 0xFFFFFFFF):
        p = CSCProps(); p.bt_cyc = 7
        ref.lib.CSCEncProps_Init(C.byref(p), d, level)
        props[f"{d}/{level}"] = {"props": p.as_dict(), "est_mem": ref.est_mem_usage(p), "header_hex": ref.write_properties(p).hex()}
json.dump(props, open(os.path.join(G, "props.json"), "w"), indent=1, sort_keys=True)

# ---- analyzer verdicts and filters (through oracle/ref_probe.cpp) ------------------------------
L = ref.lib
L.ref_analyze_block.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]; L.ref_analyze_block.restype = C.c_uint32
L.ref_dlt_bpb.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; L.ref_dlt_bpb.restype = C.c_uint32
for f in ("ref_forward_e89", "ref_inverse_e89", "ref_inverse_dict"):
    getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]; getattr(L, f).restype = None
L.ref_forward_dict.argtypes = [C.c_void_p, C.c_uint32]; L.ref_forward_dict.restype = C.c_uint32
for f in ("ref_forward_delta", "ref_inverse_delta"):
    getattr(L, f).argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]; getattr(L, f).restype = None

stages = {"analyze": {}, "filters": {}}
for kind, seed in (("text", 1), ("exe", 2), ("delta", 3), ("random", 4), ("entropy8", 5), ("silesia", 6)):
    buf = np.frombuffer(cases.build([[kind, seed, 0, 256 * 1024 + 300]]), dtype=np.uint8).copy()
    rows = []
    for i in range(0, len(buf), 8192):
        blk = buf[i:i + 8192].copy()
        bpb = C.c_uint32(0xFFFFFFFF)
        t = L.ref_analyze_block(blk.ctypes.data, len(blk), C.byref(bpb))
        row = [int(t), int(bpb.value)]
        if 0x10 <= t < 0x15 or t == 0x1E:
            row += [int(L.ref_dlt_bpb(blk.ctypes.data, len(blk), c)) for c in (1, 2, 3, 4, 8)]
        rows.append(row)
    stages["analyze"][f"{kind}/{seed}"] = rows
for kind, seed, n in (("exe", 2, 300000), ("text", 1, 300000), ("silesia", 6, 200000), ("random", 4, 70000), ("text", 1, 16383), ("text", 1, 16384)):
    src = cases.build([[kind, seed, 0, n]])
    a = np.frombuffer(src, dtype=np.uint8).copy(); L.ref_forward_e89(a.ctypes.data, n)
    b = np.frombuffer(src, dtype=np.uint8).copy(); r = L.ref_forward_dict(b.ctypes.data, n)
    ent = {"e89_sha256": cases.digest(a.tobytes()), "dict_ok": int(r), "dict_sha256": cases.digest(b.tobytes())}
    for chn in (1, 2, 3, 4, 8):
        d = np.frombuffer(src, dtype=np.uint8).copy(); L.ref_forward_delta(d.ctypes.data, n, chn)
        ent[f"delta{chn}_sha256"] = cases.digest(d.tobytes())
    stages["filters"][f"{kind}/{seed}/{n}"] = ent
json.dump(stages, open(os.path.join(G, "stages.json"), "w"), indent=1, sort_keys=True)
print("golden written to", G)
