"""The synthetic corpora are deterministic, random-access, and hit the analyzer classes they are
meant to hit."""
import collections
import ctypes as C

import numpy as np

import cases
from csc_amd import corpus


def test_deterministic_and_random_access():
    a = corpus.fill("text", corpus.SEED_ENWIK9, 0, 300000)
    b = corpus.fill("text", corpus.SEED_ENWIK9, 0, 300000)
    assert (a == b).all()
    c = corpus.fill("text", corpus.SEED_ENWIK9, 123457, 70001)
    assert (c == a[123457:123457 + 70001]).all()
    # content is pinned through the input digests recorded in tests/golden/streams.json


def test_task_slices_follow_csarc():
    # csarc.cpp:532-543 with esize = 10^9, -p8: 7 x 125000004 + 1 x 124999972
    s = corpus.task_slices(10 ** 9, 8)
    assert [n for _, n in s] == [125000004] * 7 + [124999972]
    assert s[1][0] == 125000004 and sum(n for _, n in s) == 10 ** 9
    assert corpus.task_slices(3 * 1048576, 8) == [(0, 1048580), (1048580, 1048580), (2097160, 1048568)]   # min 1 MiB + 4
    assert corpus.task_slices(10, 1) == [(0, 10)]


def test_classes(orc):
    L = orc.lib
    L.orc_analyze_block.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    def hist(kind, seed):
        x = corpus.fill(kind, seed, 0, 1 << 20)
        cnt = collections.Counter()
        bpb = C.c_uint32()
        for i in range(0, len(x), 8192):
            blk = np.ascontiguousarray(x[i:i + 8192])
            cnt[L.orc_analyze_block(blk.ctypes.data, len(blk), C.byref(bpb))] += 1
        return cnt
    assert hist("text", corpus.SEED_ENWIK9)[2] >= 120           # DT_ENGTXT
    assert hist("exe", corpus.SEED_EXE)[3] >= 120               # DT_EXE
    d = hist("delta", corpus.SEED_DELTA)
    assert sum(v for k, v in d.items() if 0x10 <= k < 0x15) >= 100
    assert hist("random", 4)[8] >= 100                          # DT_BAD
    assert hist("entropy8", 5)[7] >= 120                        # DT_ENTROPY
