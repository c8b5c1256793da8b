"""CPU tier: the oracle (plain-C restatement) against the vectors the REFERENCE produced
(tests/golden/*.json, written by tools/make_golden.py from oracle/_ref)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import cases

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAMS = json.load(open(os.path.join(G, "streams.json")))
PROPS = json.load(open(os.path.join(G, "props.json")))
STAGES = json.load(open(os.path.join(G, "stages.json")))


@pytest.mark.parametrize("key", sorted(STREAMS))
def test_oracle_stream_matches_reference_vector(orc, zalloc, key):
    name, lv = key.split("/m")
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    want = STREAMS[key]
    assert len(data) == want["input_size"] and cases.digest(data) == want["input_sha256"], "input generator drifted"
    rc, s = orc.encode(data, int(lv), dict_size, alloc=zalloc, clamp_dict=clamp, max_read=max_read)
    assert rc == 0
    assert len(s) == want["stream_size"]
    assert cases.digest(s) == want["stream_sha256"]
    if "stream_hex" in want:
        assert s.hex() == want["stream_hex"]
    rcd, back = orc.decode(s, alloc=zalloc)
    assert rcd == 0 and back == data


def test_known_answer_sizes():
    # sizes observed from the reference CLI
    assert STREAMS["empty/m3"]["stream_size"] == 27
    assert STREAMS["one_byte/m1"]["stream_size"] == 46
    assert STREAMS["zeros_8k/m3"]["stream_size"] == 54 and STREAMS["zeros_8k/m5"]["stream_size"] == 53
    assert STREAMS["abcdefgh_64k/m3"]["stream_size"] == 983
    assert STREAMS["random_64k/m2"]["stream_size"] > 65536       # stored via DT_BAD / 8-bit literals


@pytest.mark.parametrize("key", sorted(PROPS))
def test_props_init(orc, key):
    from csc_amd.capi import CSCProps
    d, lv = key.split("/")
    p = CSCProps()
    p.bt_cyc = 7
    orc.lib.CSCEncProps_Init(C.byref(p), int(d), int(lv))
    want = PROPS[key]
    assert p.as_dict() == want["props"]
    assert orc.est_mem_usage(p) == want["est_mem"]
    assert orc.write_properties(p).hex() == want["header_hex"]
    q = orc.read_properties(bytes.fromhex(want["header_hex"]))
    assert (q.dict_size, q.csc_blocksize, q.raw_blocksize) == (p.dict_size, p.csc_blocksize, p.raw_blocksize)


def test_analyzer_verdicts(orc):
    L = orc.lib
    L.orc_analyze_block.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.orc_dlt_bpb.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    seen = set()
    for key, rows in STAGES["analyze"].items():
        kind, seed = key.split("/")
        buf = np.frombuffer(cases.build([[kind, int(seed), 0, 256 * 1024 + 300]]), dtype=np.uint8).copy()
        for bi, i in enumerate(range(0, len(buf), 8192)):
            blk = buf[i:i + 8192].copy()
            bpb = C.c_uint32(0xFFFFFFFF)
            t = L.orc_analyze_block(blk.ctypes.data, len(blk), C.byref(bpb))
            row = [int(t), int(bpb.value)]
            if 0x10 <= t < 0x15 or t == 0x1E:
                row += [int(L.orc_dlt_bpb(blk.ctypes.data, len(blk), c)) for c in (1, 2, 3, 4, 8)]
            assert row == rows[bi], (key, bi)
            seen.add(t)
    # the ladder's main outcomes are all exercised (Appendix E)
    assert {1, 2, 3, 7, 8, 0x1E} <= seen and any(0x10 <= t < 0x15 for t in seen)


def test_filters(orc):
    L = orc.lib
    for f in ("orc_forward_e89", "orc_inverse_e89", "orc_inverse_dict"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]
        getattr(L, f).restype = None
    L.orc_forward_dict.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_forward_dict.restype = C.c_uint32
    for f in ("orc_forward_delta", "orc_inverse_delta"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        getattr(L, f).restype = None
    for key, want in STAGES["filters"].items():
        kind, seed, n = key.split("/")
        n = int(n)
        src = cases.build([[kind, int(seed), 0, n]])
        a = np.frombuffer(src, dtype=np.uint8).copy()
        L.orc_forward_e89(a.ctypes.data, n)
        assert cases.digest(a.tobytes()) == want["e89_sha256"]
        L.orc_inverse_e89(a.ctypes.data, n)
        assert a.tobytes() == src
        b = np.frombuffer(src, dtype=np.uint8).copy()
        assert L.orc_forward_dict(b.ctypes.data, n) == want["dict_ok"]
        assert cases.digest(b.tobytes()) == want["dict_sha256"]
        for chn in (1, 2, 3, 4, 8):
            d = np.frombuffer(src, dtype=np.uint8).copy()
            L.orc_forward_delta(d.ctypes.data, n, chn)
            assert cases.digest(d.tobytes()) == want[f"delta{chn}_sha256"]
            L.orc_inverse_delta(d.ctypes.data, n, chn)
            assert d.tobytes() == src
    # reject paths: < 16 KiB never transformed; > 82 % output rejected (random data)
    assert STAGES["filters"]["text/1/16383"]["dict_ok"] == 0
    assert STAGES["filters"]["text/1/16384"]["dict_ok"] == 1
    assert STAGES["filters"]["random/4/70000"]["dict_ok"] == 0


def test_multi_stream_digest_fixture_is_wellformed():
    """tests/golden/multi_stream_digests.json (reference digests of the whole 10^9-byte file as -p127 / -p954 task streams,
    tools/make_golden_multi.py): bench.py compares its GPU run with them; here only the shape, and that the split sizes
    are the archiver's (csarc.cpp:532-543)"""
    import json
    from csc_amd import corpus
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multi_stream_digests.json")))
    assert g["level"] == 3 and g["dict"] == 64 << 20 and g["total"] == 10 ** 9
    for want in (127, 954):
        assert len(corpus.task_slices(g["total"], want)) == want
        e = g["splits"][str(want)]
        assert len(e["sha256_of_stream_sha256s"]) == 64 and 0.2 < e["stream_bytes"] / g["total"] < 0.3


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5])
def test_oracle_pos_renormalisation_matches_the_reference(orc, zalloc, level):
    """MatchFinder::normalize (csc_mf.cpp:108-114): the restatement with its position counter started 300 000 positions short of
    0xFFFFFFF0 writes what the REFERENCE writes from the same start (tests/golden/renorm.json, tools/make_golden_renorm.py through
    oracle/ref_probe.cpp::ref_debug_set_pos) -- which is the stream of an ordinary start: a correct rebase is invisible."""
    import ctypes as C
    import hashlib
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "renorm.json")))
    orc.lib.orc_debug_set_pos.argtypes = [C.c_void_p, C.c_uint32]
    orc.lib.orc_debug_set_pos.restype = None
    orc.lib.orc_debug_get_pos.argtypes = [C.c_void_p]
    orc.lib.orc_debug_get_pos.restype = C.c_uint32
    data = cases.build(gold["spec"])
    props = orc.props_init(gold["dict"], level)
    ends = []
    rc, got = orc.encode(data, props=props, alloc=zalloc, after_create=lambda h: orc.lib.orc_debug_set_pos(h, gold["start"]))
    assert rc == 0
    want = gold["levels"][str(level)]
    assert len(got) == want["stream_bytes"] and hashlib.sha256(got).hexdigest() == want["sha256"]
    rc, plain = orc.encode(data, props=props, alloc=zalloc)
    assert rc == 0 and plain == got


def test_oracle_reproduces_a_stream_the_reference_does_not_turn_back_into_its_input(orc, zalloc):
    """tests/golden/ref_roundtrip_hazard.json (tools/make_golden_ref_roundtrip.py, found by tools/gpu_soak.py): the REFERENCE's one-byte rep match reads wnd_[wnd_size_] -- a byte the
    decoder never writes -- when wnd_curpos_ == rep_dist_[0] (csc_dec.cpp:525-527: `>` where every other copy has `>=`); under the zeroing allocator of all our
    vectors its decoder, fed the reference's own stream for this input, returns 0 and nine bytes that are not the input's.  Parity is with the reference: the
    restatement writes the same stream and decodes it into the same bytes -- wrong in the same nine places."""
    import hashlib
    gold = json.load(open(os.path.join(G, "ref_roundtrip_hazard.json")))
    data = cases.build(gold["spec"])
    assert hashlib.sha256(data).hexdigest() == gold["input_sha256"]
    rc, s = orc.encode(data, props=orc.props_init(gold["dict"], gold["level"]), alloc=zalloc)
    assert rc == 0 and len(s) == gold["stream_bytes"] and hashlib.sha256(s).hexdigest() == gold["stream_sha256"]
    rcd, back = orc.decode(s, alloc=zalloc)
    assert rcd == gold["decoded_rc"] and hashlib.sha256(back).hexdigest() == gold["decoded_sha256"]
    assert [[i, data[i], back[i]] for i in range(len(data)) if data[i] != back[i]] == gold["decoded_differs_from_input_at"]


@pytest.mark.parametrize("name", sorted(json.load(open(os.path.join(G, "soak_cases.json")))))
def test_oracle_on_the_cases_a_soak_once_found_the_hip_path_wrong_on(orc, zalloc, name):
    """tests/golden/soak_cases.json (tools/make_golden_soak_cases.py: the reference's streams): the restatement writes them too."""
    import hashlib
    gold = json.load(open(os.path.join(G, "soak_cases.json")))[name]
    data = cases.build(gold["spec"])
    assert hashlib.sha256(data).hexdigest() == gold["input_sha256"]
    rc, s = orc.encode(data, props=orc.props_init(gold["dict"], gold["level"]), alloc=zalloc)
    assert rc == 0 and len(s) == gold["stream_bytes"] and hashlib.sha256(s).hexdigest() == gold["stream_sha256"]
    assert orc.decode(s, alloc=zalloc) == (0, data)
