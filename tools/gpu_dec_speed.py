#!/usr/bin/env python3
"""Development aid: decode speed of ONE stream through CSCDec_Decode for a few (level, data kind) pairs, best of two runs each,
every result compared with the input.   gpurun -- python3 tools/gpu_dec_speed.py [MiB]"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prod = csc_amd.load()
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
print("library sha256[:16] =", hashlib.sha256(open(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"), "rb").read()).hexdigest()[:16])
for level, kind in ((3, "text"), (2, "text"), (5, "text"), (1, "text"), (3, "exe"), (5, "exe"), (3, "silesia"), (2, "mix5")):
    try:
        data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
    except Exception as e:
        print(f"m{level} {kind}: no such corpus kind ({e})"); continue
    rc, s = orc.encode(data, level, 16 << 20, alloc=orc.lib.orc_zero_alloc())
    best = 0.0; ok = True
    for _ in range(2):
        t0 = time.time(); rcd, back = prod.decode(s); dt = time.time() - t0
        ok &= rcd == 0 and back == data
        best = max(best, len(data) / 1e6 / dt)
    print(f"decode m{level} {kind:8s} {mib} MiB ({len(s)} coded): ok={ok} {best:.2f} MB/s", flush=True)
