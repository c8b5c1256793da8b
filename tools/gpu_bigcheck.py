#!/usr/bin/env python3
"""One-off evidence run: a long single stream through the HIP encoder against the reference (oracle/_ref) or the
oracle, byte for byte, and decoded back on the GPU.  usage: gpu_bigcheck.py [MiB] [level] [dict MiB]"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dict_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 64
prod = csc_amd.load()
ref_path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
chk = CscLib(ref_path if os.path.exists(ref_path) else os.path.join(ROOT, "oracle", "liborc.so"))
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p; za = o.orc_zero_alloc()
data = corpus.fill("text", corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
t0 = time.time(); rc, s = prod.encode(data, props=prod.props_init(dict_mib << 20, level)); t1 = time.time()
rc2, want = chk.encode(data, props=chk.props_init(dict_mib << 20, level), alloc=za); t2 = time.time()
rcd, back = prod.decode(s); t3 = time.time()
print(f"m{level} d{dict_mib}m {mib} MiB: gpu {len(data)/1e6/(t1-t0):.3f} MB/s, checker {len(data)/1e6/(t2-t1):.2f} MB/s, gpu decode {len(data)/1e6/(t3-t2):.2f} MB/s")
print("stream bytes", len(s), "sha256", hashlib.sha256(s).hexdigest()[:16], "identical to checker:", (rc, s) == (rc2, want), "round trip:", rcd == 0 and back == data)
sys.exit(0 if (rc, s) == (rc2, want) and rcd == 0 and back == data else 1)
