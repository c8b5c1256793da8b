#!/usr/bin/env python3
"""Record a case in which the REFERENCE's decoder reads a byte it never wrote (found by tools/gpu_soak.py, round 6): 1 100 000 bytes of the silesia stand-in at -m5
with a 1 MiB dictionary (wnd_size_ = 1 058 816: the window wraps once).  CSCDecoder::lz_decode's one-byte rep match (csc_dec.cpp:525-527) computes its source with
`wnd_curpos_ > rep_dist_[0]` where every other copy uses `>=`: at wnd_curpos_ == rep_dist_[0] (here 8192, right after the wrap) it reads wnd_[wnd_size_] -- one byte
past the window, which the decoder never writes -- where the encoder meant wnd_[0].  What CSCDec_Decode returns for the reference's OWN stream therefore hangs on the
allocator: 0x00 from a zeroing ISzAlloc (the convention of every vector in tests/golden: oracle/zalloc.c), 0xAA from a 0xAA-filling one, and with glibc's malloc in a
process that has just encoded the same bytes the stale heap may even hold the right one.  Under the zeroing allocator nine bytes of the output differ from the input
(the first wrong byte is copied on by later matches), return code 0.  Parity is with the reference, so this is what the oracle and the HIP path must reproduce byte for
byte: the same stream from the encoder, the same nine wrong bytes from the decoder (its window and the 256 bytes behind it are zeroed when the handle is created).
Generated with oracle/_ref (the reference's own sources, zeroing allocator); writes tests/golden/ref_roundtrip_hazard.json -- digests and the differing (offset, input byte,
decoded byte) triples.  tests/test_oracle_golden.py (oracle) and tests/test_gpu_parity.py (HIP path) compare with it."""
import ctypes as C, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from csc_amd import corpus
from csc_amd.capi import CscLib
SPEC = [["silesia", int(corpus.SEED_ENWIK9), 141101860, 1100000]]
LEVEL, DICT = 5, 1 << 20
ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p
data = cases.build(SPEC)
za = o.orc_zero_alloc()
rc, s = ref.encode(data, props=ref.props_init(DICT, LEVEL), alloc=za)
rcd, back = ref.decode(s, alloc=za)
o.orc_aa_alloc.restype = C.c_void_p
rca, back_aa = ref.decode(s, alloc=o.orc_aa_alloc())
assert rca == 0 and back_aa != back, 'the decoded bytes do not hang on the allocator: not the case this file is about'
assert rc == 0 and rcd == 0 and len(back) == len(data)
diff = [[i, data[i], back[i]] for i in range(len(data)) if data[i] != back[i]]
out = {"what": "csc_dec.cpp:525-527 reads wnd_[wnd_size_] (never written) when wnd_curpos_ == rep_dist_[0]: the reference's decoded bytes under a zeroing allocator; parity is with the reference's stream AND with these bytes",
       "spec": SPEC, "level": LEVEL, "dict": DICT, "input_sha256": hashlib.sha256(data).hexdigest(),
       "stream_bytes": len(s), "stream_sha256": hashlib.sha256(s).hexdigest(),
       "decoded_rc": rcd, "decoded_sha256": hashlib.sha256(back).hexdigest(), "decoded_differs_from_input_at": diff}
print(json.dumps({k: v for k, v in out.items() if k != "what"})[:600])
assert diff, "the reference gives the input back: not the case this file is about"
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ref_roundtrip_hazard.json"), "w"), indent=1)
