#!/usr/bin/env python3
"""Record what the REFERENCE (oracle/_ref) writes when its match finder's position counter starts 300 000 positions short of
0xFFFFFFF0, i.e. when MatchFinder::normalize (csc_mf.cpp:108-114) runs in the middle of a 1 MiB input -- levels 1-5, dictionary
1 MiB.  The start is set through oracle/ref_probe.cpp::ref_debug_set_pos right after CSCEnc_Create (tables zeroed: the same
situation as the ordinary start, csc_mf.cpp:56-57,81).  Writes tests/golden/renorm.json; tests/test_oracle_golden.py (oracle port)
and tests/test_gpu_stages.py (HIP path) compare with it."""
import ctypes as C, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from csc_amd.capi import CscLib
START = 0xFFFFFFF0 - 300000
SPEC = [["text", 77, 0, 500000], ["exe", 78, 0, 300000], ["text", 77, 100000, 248576]]
ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
ref.lib.ref_debug_set_pos.argtypes = [C.c_void_p, C.c_uint32]; ref.lib.ref_debug_set_pos.restype = None
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p
data = cases.build(SPEC)
out = {"what": "reference streams with MatchFinder::pos_ started at 0xFFFFFFF0 - 300000 (normalize() runs mid-input)", "start": START,
       "spec": SPEC, "dict": 1 << 20, "levels": {}}
for level in (1, 2, 3, 4, 5):
    props = ref.props_init(1 << 20, level)
    rc, s = ref.encode(data, props=props, alloc=o.orc_zero_alloc(), after_create=lambda h: ref.lib.ref_debug_set_pos(h, START))
    rc0, plain = ref.encode(data, props=props, alloc=o.orc_zero_alloc())
    assert rc == 0 and rc0 == 0
    out["levels"][str(level)] = {"stream_bytes": len(s), "sha256": hashlib.sha256(s).hexdigest(), "differs_from_ordinary_start": s != plain}
    print(level, out["levels"][str(level)], flush=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "renorm.json"), "w"), indent=1)
