import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, csc_amd
from csc_amd import corpus
prod = csc_amd.load()
for kind in ("text", "exe", "delta", "silesia"):
    for n in (262144, 524288):
        data = corpus.fill(kind, 5000, 0, n).tobytes()
        for lvl in (3, 2):
            t0 = time.time(); rc, got = prod.encode(data, lvl, 64 << 20); dt = time.time() - t0
            print(f"{kind:8s} {n:7d} m{lvl} {n/1e6/dt:7.3f} MB/s  -> {len(got)} rc={rc}", flush=True)
