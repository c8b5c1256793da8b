"""CPU tier: tests/model/m3_model.c -- the ARRANGEMENT the HIP kernels give LZ::compress_advanced on the level-3 geometry
(csc_amd/csrc/csc_kernels_dp4.inc: speculative match-finder pre-pass with undo, parse-independent hash candidates, rep
lengths from equality masks that carry the reference's caps, DP in a ring relative to the current node) -- must produce the
oracle's bytes.  The model includes the oracle's encoder and replaces compress_advanced only; both are test infrastructure."""
import ctypes as C
import os
import subprocess
import sys

import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "model", "libm3model.so")


@pytest.fixture(scope="module")
def built():
    src = [os.path.join(ROOT, "tests", "model", "m3_model.c"), os.path.join(ROOT, "oracle", "orc_decoder.c"), os.path.join(ROOT, "oracle", "zalloc.c")]
    subprocess.run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-Wl,-Bsymbolic", "-o", SO] + src + ["-lm"], check=True)
    return SO


def run_cases(knobs):
    """child process: the model reads its knobs from the environment when the library loads"""
    code = f"""
import ctypes as C, os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
import cases
from csc_amd.capi import CscLib
orc = CscLib(os.path.join({ROOT!r}, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
mod = CscLib({SO!r}); mod.lib.orc_zero_alloc.restype = C.c_void_p
za, zb = orc.lib.orc_zero_alloc(), mod.lib.orc_zero_alloc()
n = 0
for name in ("empty", "one_byte", "zeros_8k", "abcdefgh_64k", "random_64k", "text_20k", "text_300k", "exe_300k", "mix_types", "dup_blocks",
             "ragged_tail_511", "short_reads_511", "window_wrap_32k", "periodic_5000x200", "delta_200k"):
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    for lv in (3, 4):
        rc, a = orc.encode(data, lv, dict_size, alloc=za, clamp_dict=clamp, max_read=max_read)
        rc2, b = mod.encode(data, lv, dict_size, alloc=zb, clamp_dict=clamp, max_read=max_read)
        assert rc == 0 and rc2 == 0 and a == b, (name, lv, len(a), len(b))
        n += 1
# custom geometry: bucket of 1, good_len 8 / 16 / 31, a window that wraps
from csc_amd.capi import CSCProps
data = cases.build([["text", 13, 0, 700000], ["exe", 14, 0, 200000]])
for width, good, dsz in ((1, 16, 65536), (2, 8, 1 << 20), (2, 31, 40000), (6, 24, 1 << 18)):
    p = orc.props_init(dsz, 3); p.hash_width = width; p.good_len = good
    rc, a = orc.encode(data, props=p, alloc=za); rc2, b = mod.encode(data, props=p, alloc=zb)
    assert rc == 0 and rc2 == 0 and a == b, (width, good, dsz)
    n += 1
print("MODEL_OK", n)
"""
    env = dict(os.environ, M3_STATS="1", **{k: str(v) for k, v in knobs.items()})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "MODEL_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    return out.stderr


@pytest.mark.parametrize("knobs", [
    {},                                                              # the model's defaults: look-ahead 192, masks re-based after 16 positions, 6 positions late
    {"M3_LA": 112, "M3_REFRESH_AT": 32, "M3_REFRESH_DELAY": 16},     # the kernel's constants (csc_kernels_dp4.inc): kD4LA 48 + one batch of 64 ahead, kD4RefreshAt 32,
                                                                     # a re-based mask taken up at the spine's next look at its ids (every 16th node)
    {"M3_LA": 1, "M3_REFRESH_DELAY": 0},                             # inserter in lock step with the parser
    {"M3_LA": 250, "M3_REFRESH_DELAY": 60, "M3_REFRESH_AT": 30},     # service far behind: rep lengths mostly by direct compare
])
def test_model_equals_oracle(built, knobs):
    err = run_cases(knobs)
    line = [l for l in err.splitlines() if l.startswith("m3_model: nodes")]
    assert line, err[-500:]
    # the paths the kernel's exactness rests on must have been walked
    import re
    s = line[-1]
    # the way out's literal trees, deferred and batched as the tree wavefront does it (d6_literal / d6_trees / d6_join), gave the oracle's
    # probabilities at every decision and the oracle's p_lit at every join (the model aborts otherwise): batches of 1..8 records, chains inside
    t = [l for l in err.splitlines() if l.startswith("m3_model: literal trees")][-1]
    assert int(re.search(r"joins (\d+)", t).group(1)) > 1000, t
    assert int(re.search(r"deferred batches (\d+)", t).group(1)) > 10000, t
    assert int(re.search(r"chained decisions (\d+)", t).group(1)) > 1000, t       # same context + common prefix inside a batch
    assert int(re.search(r"most records pending at a join (\d+)", t).group(1)) > 8, t
    assert int(re.search(r"deviations (\d+)", s).group(1)) > 0, s          # speculative inserts undone + replayed
    assert int(re.search(r"slide events (\d+)", s).group(1)) > 100, s
    assert int(re.search(r"mask refreshes (\d+)", s).group(1)) > 100, s
    # every node's label was also formed the way the kernel's spine forms it (two edge rings merged by price, then source; own edge last)
    assert int(re.search(r"spine/edge merges checked (\d+)", s).group(1)) > 100000, s
