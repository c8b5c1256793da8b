#!/usr/bin/env python3
"""Development probe: encode assorted inputs with the HIP library and the oracle, report the
first differing byte.  Run on the GPU box:  gpurun -- python tools/gpu_probe.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib

prod = csc_amd.load()
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so"))
orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()

def case(name, data, level, dict_size=1 << 20, **kw):
    t0 = time.time()
    rc, got = prod.encode(data, level, dict_size, **kw)
    t1 = time.time()
    rc2, want = orc.encode(data, level, dict_size, alloc=za, **kw)
    t2 = time.time()
    ok = rc == 0 and got == want
    msg = f"{name:22s} m{level} {len(data):>8d} -> {len(got):>8d} (orc {len(want)}) rc={rc} gpu {t1-t0:.2f}s cpu {t2-t1:.2f}s {'OK' if ok else 'DIFF'}"
    if not ok:
        n = min(len(got), len(want))
        first = next((i for i in range(n) if got[i] != want[i]), n)
        msg += f" first diff @ {first}"
    print(msg, flush=True)
    return ok

T = lambda n, s=1: corpus.fill("text", s, 0, n).tobytes()
X = lambda n, s=2: corpus.fill("exe", s, 0, n).tobytes()
D = lambda n, s=3: corpus.fill("delta", s, 0, n).tobytes()
R = lambda n, s=4: corpus.fill("random", s, 0, n).tobytes()
E = lambda n, s=5: corpus.fill("entropy8", s, 0, n).tobytes()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
allok = True
levels = (3, 2, 1, 4, 5)
for lv in levels:
    allok &= case("empty", b"", lv)
    allok &= case("one", b"x", lv)
    allok &= case("zeros8k", bytes(8192), lv)
    allok &= case("abc64k", b"abcdefgh" * 8192, lv)
    allok &= case("rand64k", R(65536), lv)
    allok &= case("text20k", T(20000), lv)
    allok &= case("text300k", T(300000), lv)
    allok &= case("exe300k", X(300000), lv)
    allok &= case("delta200k", D(200000), lv)
    allok &= case("ent100k", E(100000), lv)
    allok &= case("mix", T(100000) + R(30000) + X(100000) + D(70000) + E(40000) + T(50000) + R(100), lv)
    if quick:
        break
if not quick:
    big = T(3_000_000, 7) + X(1_500_000, 8) + T(1_000_000, 9)
    for lv in (3, 2, 5):
        allok &= case("wrap100k", big, lv, 100_000, clamp_dict=False)
        allok &= case("shortread8191", big[:300_000], lv, max_read=8191)
        allok &= case("zeros5M", bytes(5_000_000), lv, 1 << 22)
        allok &= case("periodic", T(5000, 11) * 400, lv, 1 << 22)
print("ALL OK" if allok else "SOME DIFF")
sys.exit(0 if allok else 1)
