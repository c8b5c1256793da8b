#!/usr/bin/env python3
"""Encode <MiB> of the enwik9 stand-in at <level> with the product library (for rocprofv3 runs)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csc_amd
from csc_amd import corpus
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
lib = csc_amd.load()
data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
t0 = time.time()
rc, s = lib.encode(data, props=lib.props_init(64 << 20, level))
dt = time.time() - t0
print(f"level {level} {kind} {mib} MiB -> {len(s)} bytes rc={rc} {len(data)/1e6/dt:.3f} MB/s")
