#!/usr/bin/env python3
"""Development probe: a longer level-3 stream against the oracle, chunk by chunk.  gpurun -- python tools/gpu_dp4_long.py [MiB] [offset MiB] [offset bytes instead] [kind: text exe delta entropy8 mix]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 24
off = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prod = csc_amd.load()
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so"))
orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()
boff = int(sys.argv[3]) if len(sys.argv) > 3 else off << 20
kind = sys.argv[4] if len(sys.argv) > 4 else "text"
if kind == "mix":
    import numpy as np
    parts = [corpus.fill(k, corpus.SEED_ENWIK9 + i, boff, (mib << 20) // 6) for i, k in enumerate(("text", "exe", "delta", "text", "entropy8", "exe"))]
    data = np.concatenate(parts).tobytes()
else:
    data = corpus.fill(kind, corpus.SEED_ENWIK9, boff, mib << 20).tobytes()
p = prod.props_init(64 << 20, 3)
t0 = time.time(); rc, got = prod.encode(data, props=p); dt = time.time() - t0
rc2, want = orc.encode(data, props=p, alloc=za)
n = min(len(got), len(want))
first = next((i for i in range(n) if got[i] != want[i]), n)
print(f"{kind} {mib} MiB at +{boff} B: {len(data)/1e6/dt:.3f} MB/s rc={rc} out {len(got)} vs {len(want)}: {'bit-exact' if got == want else f'DIFF first at output byte {first} = {100.0*first/len(want):.1f} % of the stream'}")
