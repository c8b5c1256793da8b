"""GPU tier: the run-ahead inserter forms of round 4 on geometries CSCEncProps_Init never produces -- the paths their exactness rests on
(tests/model/bt_model.c, hp_model.c prove the arrangements on the CPU; here the kernels themselves against the oracle):
  * csc_kernels_bt.inc: a small tree ring that wraps many times (the sub-blocks around the wrap take the one-wavefront form; a descent
    that reaches the slot of a later position of its own batch reads the shadow copy), few / many tree steps, small / large good_len,
    zero runs (matches > 129: the inserter takes back what it inserted and replays with the skip rule);
  * csc_kernels_hp.inc: bucket widths 1 .. 8, greedy and lazy parser, good_len 8 .. 200, a window that wraps."""
import ctypes as C
import os

import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs():
    import csc_amd
    from csc_amd.capi import CscLib
    prod = csc_amd.load()
    orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so"))
    orc.lib.orc_zero_alloc.restype = C.c_void_p
    return prod, orc, orc.lib.orc_zero_alloc()


BT_DATA = [["text", 13, 0, 500000], ["exe", 14, 0, 200000], ["silesia", 6, 1 << 20, 600000], ["pattern", "00", 70000], ["text", 13, 0, 100000]]


@pytest.mark.parametrize("bt_size,cyc,good,dsz", [(40000, 32, 48, 1 << 20), (100000, 4, 16, 1 << 18), (1 << 20, 32, 200, 1 << 21), (300000, 16, 8, 300000)])
def test_binary_tree_inserter_custom_geometry(libs, bt_size, cyc, good, dsz):
    prod, orc, za = libs
    data = cases.build(BT_DATA)

    def mk(L):
        p = L.props_init(dsz, 5)
        p.bt_size = bt_size; p.bt_cyc = cyc; p.good_len = good
        return p
    assert prod.encode(data, props=mk(prod)) == orc.encode(data, props=mk(orc), alloc=za)


@pytest.mark.parametrize("raw", [8192, 4096 + 512, 24576])
def test_binary_tree_inserter_small_raw_blocksize(libs, raw):
    """round 4's advisor finding: the inserter's undo log (36 KiB) used to live in the filter scratch (4 x raw_blocksize), so a
    raw_blocksize below ~9 KiB let it run into the output arena.  It has a region of its own now (EncState::bt_undo)."""
    prod, orc, za = libs
    data = cases.build([["text", 21, 0, 150000], ["pattern", "00", 40000], ["exe", 22, 0, 60000], ["zeros", 30000]])

    def mk(L):
        p = L.props_init(1 << 20, 5)
        p.raw_blocksize = raw
        return p
    assert prod.encode(data, props=mk(prod)) == orc.encode(data, props=mk(orc), alloc=za)


HP_DATA = [["text", 13, 0, 500000], ["exe", 14, 0, 200000], ["pattern", "00", 70000], ["delta", 3, 0, 100000], ["text", 13, 0, 100000]]


@pytest.mark.parametrize("width,bits,good,mode,dsz", [(1, 16, 32, 2, 1 << 20), (3, 12, 8, 2, 1 << 18), (8, 10, 24, 1, 40000), (5, 18, 200, 2, 1 << 21), (8, 20, 24, 2, 1 << 22)])
def test_hash_bucket_inserter_custom_geometry(libs, width, bits, good, mode, dsz):
    prod, orc, za = libs
    data = cases.build(HP_DATA)

    def mk(L):
        p = L.props_init(dsz, 2)
        p.hash_width = width; p.hash_bits = bits; p.good_len = good; p.lz_mode = mode
        return p
    assert prod.encode(data, props=mk(prod)) == orc.encode(data, props=mk(orc), alloc=za)


def test_batch_of_level5_and_level2_streams_one_launch_each(libs):
    """the inserter forms under CSCMI_EncodeDeviceChunkBatch: six level-5 and six level-2 streams advanced together (one launch per
    kernel flavour and chunk round), every stream == the oracle's"""
    import torch
    from csc_amd.capi import BytesWriter
    prod, orc, za = libs
    L = prod.lib
    L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    specs = [(5, [["silesia", 40 + i, 0, 300000 + 7000 * i]]) for i in range(3)] + [(5, [["text", 50 + i, 0, 250000], ["zeros", 20000 + i]]) for i in range(3)] \
        + [(2, [["exe", 60 + i, 0, 400000 + 5000 * i]]) for i in range(3)] + [(2, [["text", 70 + i, 0, 300000], ["delta", 70 + i, 0, 90000]]) for i in range(3)]
    datas = [cases.build(sp) for _, sp in specs]
    hs, ws, devs = [], [], []
    for (lv, _), d in zip(specs, datas):
        p = prod.props_init(len(d), lv)
        w = BytesWriter()
        h = L.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
        assert h
        w.out += prod.write_properties(p)
        hs.append(h); ws.append(w)
        devs.append(torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda())
    torch.cuda.synchronize()
    n = len(hs)
    H = (C.c_void_p * n)(*hs)
    Z = [len(d) for d in datas]
    P = (C.c_void_p * n)(*[t.data_ptr() for t in devs])
    assert L.CSCMI_EncodeDeviceChunkBatch(n, H, P, (C.c_size_t * n)(*Z)) == 0
    for h in hs:
        assert L.CSCEnc_Encode_Flush(h) == 0
        L.CSCEnc_Destroy(h)
    for (lv, _), d, w in zip(specs, datas, ws):
        rc, want = orc.encode(d, lv, len(d), alloc=za)
        assert rc == 0 and bytes(w.out) == want, (lv, len(w.out), len(want))
