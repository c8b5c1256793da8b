#!/usr/bin/env python3
"""Decode one oracle-made stream with the product library (for rocprofv3 runs).  args: [level] [MiB] [kind]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
prod = csc_amd.load()
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
rc, s = orc.encode(data, level, 16 << 20, alloc=orc.lib.orc_zero_alloc())
t0 = time.time(); rc, back = prod.decode(s); dt = time.time() - t0
print(f"decode m{level} {kind} {mib} MiB: rc={rc} ok={back == data} {len(data)/1e6/dt:.2f} MB/s")
