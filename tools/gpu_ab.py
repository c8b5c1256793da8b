#!/usr/bin/env python3
"""Development aid: single-stream level-3 encode speed of several builds of the library on the same box, interleaved.
gpurun -- python tools/gpu_ab.py [MiB] name1 name2 ...   (names under csc_amd/csrc/build/ab/, 'cur' = the product library)"""
import ctypes as C, os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib
mib = int(sys.argv[1])
names = sys.argv[2:]
libs = {n: CscLib(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so") if n == "cur" else os.path.join(ROOT, "csc_amd", "csrc", "build", "ab", n + ".so")) for n in names}
for kind in ("text", "exe"):
    data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
    res = {n: [] for n in names}; dig = {}
    for rep in range(3):
        for n in names:
            p = libs[n].props_init(64 << 20, 3)
            t0 = time.time(); rc, got = libs[n].encode(data, props=p); dt = time.time() - t0
            res[n].append(len(data) / 1e6 / dt); dig[n] = hashlib.sha256(got).hexdigest()[:12]
    for n in names:
        print(f"{kind:5s} {n:12s} best {max(res[n]):.3f} MB/s  all {' '.join(f'{v:.3f}' for v in res[n])}  sha {dig[n]}", flush=True)
