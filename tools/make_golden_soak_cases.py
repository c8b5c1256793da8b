#!/usr/bin/env python3
"""Record the reference's streams for inputs on which a randomized soak (tools/gpu_soak.py, tools/gpu_soak_batch.py) once found the HIP path wrong, so that each such
case stays a test: tests/golden/soak_cases.json <- oracle/_ref (the reference's own sources, zeroing allocator).  tests/test_oracle_golden.py replays them through
the oracle, tests/test_gpu_parity.py through the HIP path.
  level3_short_windows_*: delta stand-in, -m3 -- hundreds of positions of literals between two- and three-byte matches: every DP window ends at its first node
  (csc_lz.cpp:213-217) or is skipped; the level-3 form dropped the re-based ids of its rep distances on that way out (csc_kernels_dp4.inc, DP_SEEN_EXIT) until a
  candidate entry was recycled under a live id (round 6)."""
import ctypes as C, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from csc_amd import corpus
from csc_amd.capi import CscLib
SEED = int(corpus.SEED_ENWIK9)
CASES = {
    "level3_short_windows_a": {"spec": [["delta", SEED, 65229574, 26254]], "level": 3, "dict": 32768},
    "level3_short_windows_b": {"spec": [["delta", SEED, 65237766, 12000]], "level": 3, "dict": 32768},
    "level3_short_windows_c": {"spec": [["delta", SEED, 65229574, 16384]], "level": 3, "dict": 64 << 20},
}
ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p; za = o.orc_zero_alloc()
out = {}
for name, cs in CASES.items():
    data = cases.build(cs["spec"])
    rc, s = ref.encode(data, props=ref.props_init(cs["dict"], cs["level"]), alloc=za)
    rcd, back = ref.decode(s, alloc=za)
    assert rc == 0 and rcd == 0 and back == data
    out[name] = dict(cs, input_sha256=hashlib.sha256(data).hexdigest(), stream_bytes=len(s), stream_sha256=hashlib.sha256(s).hexdigest())
    print(name, len(data), "->", len(s), out[name]["stream_sha256"][:16])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "soak_cases.json"), "w"), indent=1)
