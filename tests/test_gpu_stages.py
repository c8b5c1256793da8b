"""GPU tier, stage level: the analyzer kernel and the three forward filters on their own -- not through stream bytes -- against
the vectors the REFERENCE produced (tests/golden/stages.json, tools/make_golden.py via oracle/_ref + oracle/ref_probe.cpp):
csc_analyzer.cpp:122-239, csc_filters.cpp:132-164, 256-335, 508-598.  The entry points live in tests/stage/libcsc_stage.so
(the product's sources built with -DCSCMI_STAGE_TEST); the product library does not export them."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = json.load(open(os.path.join(ROOT, "tests", "golden", "stages.json")))


@pytest.fixture(scope="module")
def stage():
    import torch  # noqa: F401  (one HIP runtime per process, see csc_amd.load)
    from csc_amd.capi import CscLib, BytesWriter
    lib = CscLib(os.path.join(ROOT, "tests", "stage", "libcsc_stage.so"))
    L = lib.lib
    L.CSCST_Analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    L.CSCST_Analyze.restype = C.c_int
    L.CSCST_Filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
    L.CSCST_Filter.restype = C.c_int
    props = lib.props_init(1 << 20, 3)
    w = BytesWriter()
    h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
    assert h
    yield lib, h, w
    L.CSCEnc_Destroy(h)


def test_product_library_has_no_stage_entry_points(prod):
    assert not hasattr(prod.lib, "CSCST_Analyze") and not hasattr(prod.lib, "CSCST_Filter") and not hasattr(prod.lib, "CSCST_SetPos")


@pytest.mark.parametrize("key", sorted(STAGES["analyze"]))
def test_analyzer_kernel_matches_reference_verdicts(stage, key):
    """k_analyze: type and bpb of every 8 KiB block; GetDltBpb for the verdict's channel (all five for a < 512 B tail)"""
    lib, h, _ = stage
    kind, seed = key.split("/")
    buf = np.frombuffer(cases.build([[kind, int(seed), 0, 256 * 1024 + 300]]), dtype=np.uint8).copy()
    rows = STAGES["analyze"][key]
    out = (C.c_uint32 * (7 * len(rows)))()
    assert lib.lib.CSCST_Analyze(h, buf.ctypes.data, len(buf), out) == 0
    for b, want in enumerate(rows):
        t, bpb = out[b * 7], out[b * 7 + 1]
        assert t == want[0], (key, b, t, want)
        if t == 0x1E:                                       # DT_SKIP: bpb untouched in the reference, all five channel figures
            assert [out[b * 7 + 2 + k] for k in range(5)] == want[2:7], (key, b)
            continue
        assert bpb == want[1], (key, b, bpb, want)
        if 0x10 <= t < 0x15:
            ch = t - 0x10
            assert out[b * 7 + 2 + ch] == want[2 + ch], (key, b)


@pytest.mark.parametrize("key", sorted(STAGES["filters"]))
def test_filter_kernels_match_reference_outputs(stage, key):
    """Forward_E89, Foward_Dict (incl. the 16 383 / 16 384-byte boundary and the > 82 % reject) and Forward_Delta x 5 channels"""
    lib, h, _ = stage
    kind, seed, n = key.split("/")
    n = int(n)
    want = STAGES["filters"][key]
    src = cases.build([[kind, int(seed), 0, n]])
    res = C.c_uint32(0)
    a = np.frombuffer(src, dtype=np.uint8).copy()
    assert lib.lib.CSCST_Filter(h, 0, a.ctypes.data, n, 0, C.byref(res)) == 0
    assert cases.digest(a.tobytes()) == want["e89_sha256"]
    b = np.frombuffer(src, dtype=np.uint8).copy()
    assert lib.lib.CSCST_Filter(h, 1, b.ctypes.data, n, 0, C.byref(res)) == 0
    assert res.value == want["dict_ok"]
    assert cases.digest(b.tobytes()) == want["dict_sha256"]
    for chn in (1, 2, 3, 4, 8):
        d = np.frombuffer(src, dtype=np.uint8).copy()
        assert lib.lib.CSCST_Filter(h, 2, d.ctypes.data, n, chn, C.byref(res)) == 0
        assert cases.digest(d.tobytes()) == want[f"delta{chn}_sha256"], (key, chn)


def test_dict_filter_reject_paths_are_the_references(stage):
    assert STAGES["filters"]["text/1/16383"]["dict_ok"] == 0 and STAGES["filters"]["text/1/16384"]["dict_ok"] == 1
    assert STAGES["filters"]["random/4/70000"]["dict_ok"] == 0


def test_enwik8_standin_level1_against_the_oracle(prod, orc, zalloc):
    """BASELINE.json configs[0] on the HIP path: the enwik8 stand-in (kind text, seed 0xC5C00001) at -m1 -d1m, 4 MiB + a ragged
    tail, encode == oracle, decode == input"""
    from csc_amd import corpus
    data = corpus.fill("text", corpus.SEED_ENWIK8, 0, (4 << 20) + 12345).tobytes()
    rc, got = prod.encode(data, 1, 1 << 20)
    rc2, want = orc.encode(data, 1, 1 << 20, alloc=zalloc)
    assert rc == 0 and rc2 == 0 and got == want
    rcd, back = prod.decode(got)
    assert rcd == 0 and back == data


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5])
def test_pos_renormalisation_is_reached_and_matches_the_oracle(level, orc, zalloc):
    """MatchFinder::normalize (csc_mf.cpp:108-114) runs when pos_ reaches 0xFFFFFFF0 -- 3.2 GB into a stream, beyond every BASELINE
    config.  Started 300 000 positions short of it (a development hook on both sides: orc_debug_set_pos in the oracle port,
    CSCST_SetPos in the test-only build of the product), the counter crosses the line in the middle of 1 MiB: every table is
    rebased there, and at level 3 the pipeline form hands over to its one-wavefront fallback from 0xFFFF0000 on.  Same bytes."""
    import torch  # noqa: F401
    from csc_amd.capi import CscLib
    lib = CscLib(os.path.join(ROOT, "tests", "stage", "libcsc_stage.so"))
    lib.lib.CSCST_SetPos.argtypes = [C.c_void_p, C.c_uint32]
    lib.lib.CSCST_SetPos.restype = C.c_int
    orc.lib.orc_debug_set_pos.argtypes = [C.c_void_p, C.c_uint32]
    orc.lib.orc_debug_set_pos.restype = None
    start = 0xFFFFFFF0 - 300000
    data = cases.build([["text", 77, 0, 500000], ["exe", 78, 0, 300000], ["text", 77, 100000, 248576]])      # (= renorm.json's spec)
    props = lib.props_init(1 << 20, level)
    rc, got = lib.encode(data, props=props, after_create=lambda h: lib.lib.CSCST_SetPos(h, start))
    rc2, want = orc.encode(data, props=props, alloc=zalloc, after_create=lambda h: orc.lib.orc_debug_set_pos(h, start))
    assert rc == 0 and rc2 == 0
    assert got == want, (level, len(got), len(want))
    # ... and what the REFERENCE writes from the same start (tests/golden/renorm.json, recorded through oracle/ref_probe.cpp)
    import hashlib
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "renorm.json")))
    assert gold["start"] == start and gold["dict"] == 1 << 20
    g = gold["levels"][str(level)]
    assert len(got) == g["stream_bytes"] and hashlib.sha256(got).hexdigest() == g["sha256"]
    rcd, back = lib.decode(got)
    assert rcd == 0 and back == data
