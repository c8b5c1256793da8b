#!/usr/bin/env python3
"""Development aid: ONE case of tools/gpu_soak.py again, through several builds of the library (names under csc_amd/csrc/build/ab/, 'cur' = the product), each
stream compared with the reference build's.  gpurun -- python tools/gpu_case.py <kind> <offset> <n> <level> <dict bytes> [name ...]"""
import ctypes as C, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib
kind, off, n, level, dsz = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
names = [a for a in sys.argv[6:] if not a.startswith("--")] or ["cur"]
flags = [a for a in sys.argv[6:] if a.startswith("--")]


def mk(L):
    p = L.props_init(dsz, level)
    if "--nofilters" in flags: p.DLTFilter = 0; p.TXTFilter = 0; p.EXEFilter = 0
    if "--nodlt" in flags: p.DLTFilter = 0
    if "--notxt" in flags: p.TXTFilter = 0
    if "--noexe" in flags: p.EXEFilter = 0
    return p

ref_path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
chk = CscLib(ref_path if os.path.exists(ref_path) else os.path.join(ROOT, "oracle", "liborc.so"))
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p; za = o.orc_zero_alloc()
data = corpus.fill(kind, corpus.SEED_ENWIK9, off, n).tobytes()
rc2, want = chk.encode(data, props=mk(chk), alloc=za)
rcr, ref_back = chk.decode(want, alloc=za)       # (zeroing allocator: csc_dec.cpp:525-527 can read a window byte the decoder never wrote)
print(f"reference: rc={rc2} {len(want)} B sha {hashlib.sha256(want).hexdigest()[:12]}; its own decoder: rc={rcr}, {'gives the input back' if ref_back == data else 'does NOT give the input back (%d bytes differ)' % sum(1 for a, b in zip(ref_back, data) if a != b)}", flush=True)
for nm in names:
    L = CscLib(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so") if nm == "cur" else os.path.join(ROOT, "csc_amd", "csrc", "build", "ab", nm + ".so"))
    rc, s = L.encode(data, props=mk(L))
    first = next((i for i in range(min(len(s), len(want))) if s[i] != want[i]), None)
    rcd, back = L.decode(s) if rc == 0 else (rc, b"")
    if "--save" in flags:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", f"case_{nm}.bin"), "wb").write(s); open(os.path.join(ROOT, "gpurun_out", "case_ref.bin"), "wb").write(want)
    if rc == 0 and back != data:
        bad = [i for i in range(min(len(back), len(data))) if back[i] != data[i]]
        print(f"    decoded {len(back)} B of {len(data)}; {len(bad)} bytes differ, first at {bad[0] if bad else None} (= {bad[0] / (1 << 20) if bad else 0:.4f} MiB), last at {bad[-1] if bad else None}; runs: "
              + ", ".join(f"{a}..{b}" for a, b in (lambda r: r[:12])([(x, y) for x, y in __import__('itertools').groupby(enumerate(bad), lambda t: t[1] - t[0]) for x, y in [(lambda g: (g[0][1], g[-1][1]))(list(y))]])), flush=True)
    if rc == 0 and back != data and "--show" in flags:
        for q in bad[:8]:
            print(f"      @{q}: want {data[q-8:q+8].hex(' ')} | got {back[q-8:q+8].hex(' ')}", flush=True)
    print(f"{nm:10s}: rc={rc} {len(s)} B sha {hashlib.sha256(s).hexdigest()[:12]} {'IDENTICAL' if (rc, s) == (rc2, want) else 'DIFFERS, first at byte %s' % first}; decode {'== the reference decoder' if (rcd, back) == (rcr, ref_back) else 'DIFFERS from the reference decoder rc=%d' % rcd}{'' if back == data else ' (not the input)'}", flush=True)
