#!/usr/bin/env python3
"""Development probe for the pipeline form (csc_kernels_dp4.inc): level-3 cases against the oracle, then a timing.
Run on the GPU box:  gpurun -- python tools/gpu_dp4.py [quick]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import csc_amd, cases
from csc_amd import corpus
from csc_amd.capi import CscLib

prod = csc_amd.load()
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so"))
orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()
allok = True


def case(name, data, level=3, dict_size=1 << 20, **kw):
    global allok
    t0 = time.time()
    rc, got = prod.encode(data, level, dict_size, **kw)
    t1 = time.time()
    rc2, want = orc.encode(data, level, dict_size, alloc=za, **kw)
    ok = rc == 0 and got == want
    n = min(len(got), len(want))
    first = next((i for i in range(n) if got[i] != want[i]), n)
    print(f"{name:22s} m{level} {len(data):>9d} -> {len(got):>9d} (orc {len(want)}) rc={rc} gpu {t1-t0:.2f}s {'OK' if ok else f'DIFF first diff @ {first}'}", flush=True)
    allok &= ok
    return ok


names = ["empty", "one_byte", "zeros_8k", "text_20k", "abcdefgh_64k", "random_64k", "text_300k", "exe_300k", "mix_types", "dup_blocks",
         "ragged_tail_511", "short_reads_511", "window_wrap_32k", "periodic_5000x200"]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    names = names[:5]
for nm in names:
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[nm]
    if not case(nm, cases.build(spec), 3, dict_size, clamp_dict=clamp, max_read=max_read):
        break
if allok:
    data = corpus.fill("text", corpus.SEED_ENWIK9, 0, 8 << 20).tobytes()
    p = prod.props_init(64 << 20, 3)
    t0 = time.time(); rc, got = prod.encode(data, props=p); dt = time.time() - t0
    rc2, want = orc.encode(data, props=p, alloc=za)
    print(f"enwik9-like 8 MiB -m3 -d64m: {len(data)/1e6/dt:.3f} MB/s rc={rc} {'bit-exact' if got == want else 'DIFF'}", flush=True)
    allok &= got == want
print("ALL OK" if allok else "SOME DIFF")
sys.exit(0 if allok else 1)
