#!/usr/bin/env python3
"""Development probe for the device decoder: decode oracle-produced streams on the GPU."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import csc_amd, cases
from csc_amd.capi import CscLib
prod = csc_amd.load()
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()
names = sys.argv[1:] or ["empty", "one_byte", "zeros_8k", "abcdefgh_64k", "random_64k", "text_20k", "text_300k", "exe_300k", "delta_200k",
                         "entropy8_100k", "mix_types", "ragged_tail_511", "window_wrap_32k", "periodic_5000x200", "zeros_5m", "text_4m_d16m"]
allok = True
for name in names:
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    for level in (3, 2, 5):
        rc, s = orc.encode(data, level, dict_size, alloc=za, clamp_dict=clamp, max_read=max_read)
        t0 = time.time()
        rcd, back = prod.decode(s)
        dt = time.time() - t0
        ok = rcd == 0 and back == data
        allok &= ok
        msg = f"{name:20s} m{level} {len(s):>8d} -> {len(back):>8d} rc={rcd} {len(data)/1e6/max(dt,1e-9):7.2f} MB/s {'OK' if ok else 'DIFF'}"
        if not ok:
            n = min(len(back), len(data)); first = next((i for i in range(n) if back[i] != data[i]), n)
            msg += f" first diff @ {first}"
        print(msg, flush=True)
print("ALL OK" if allok else "SOME DIFF")
