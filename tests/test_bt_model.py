"""CPU tier: tests/model/bt_model.c -- the ARRANGEMENT the HIP kernels give the binary-tree match finder of level 5
(csc_amd/csrc/csc_kernels_bt.inc: an inserter that runs ahead of the parser, 64 positions per batch with interleaved descents,
same-hash positions following each other down their tree, shadow copies of reused ring slots, per-position records, an undo log
for the long-match skip rule, find_match's acceptance over a record) -- must produce the
oracle's bytes.  The model includes the oracle's encoder and replaces compress_advanced only; both are test infrastructure."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "model", "libbtmodel.so")


@pytest.fixture(scope="module")
def built():
    src = [os.path.join(ROOT, "tests", "model", "bt_model.c"), os.path.join(ROOT, "oracle", "orc_decoder.c"), os.path.join(ROOT, "oracle", "zalloc.c")]
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
             "ragged_tail_511", "short_reads_511", "window_wrap_32k", "periodic_5000x200", "delta_200k", "silesia_like_3m", "zeros_5m"):
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    rc, a = orc.encode(data, 5, dict_size, alloc=za, clamp_dict=clamp, max_read=max_read)
    rc2, b = mod.encode(data, 5, dict_size, alloc=zb, clamp_dict=clamp, max_read=max_read)
    assert rc == 0 and rc2 == 0 and a == b, (name, len(a), len(b))
    n += 1
# custom geometry: a small tree ring that wraps many times (the one-wavefront form takes the sub-blocks around the wrap), few
# and many tree steps, small and large good_len
data = cases.build([["text", 13, 0, 500000], ["exe", 14, 0, 200000], ["silesia", 6, 1 << 20, 600000], ["pattern", "00", 70000], ["text", 13, 0, 100000]])
for bt_size, cyc, good, dsz in ((40000, 32, 48, 1 << 20), (100000, 4, 16, 1 << 18), (1 << 20, 32, 200, 1 << 21), (300000, 16, 8, 300000)):
    p = orc.props_init(dsz, 5); p.bt_size = bt_size; p.bt_cyc = cyc; p.good_len = good
    rc, a = orc.encode(data, props=p, alloc=za); rc2, b = mod.encode(data, props=p, alloc=zb)
    assert rc == 0 and rc2 == 0 and a == b, (bt_size, cyc, good, dsz)
    n += 1
print("MODEL_OK", n)
"""
    env = dict(os.environ, BTM_STATS="1", **{k: str(v) for k, v in knobs.items()})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "MODEL_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    return out.stderr


@pytest.mark.parametrize("knobs", [
    {"BTM_LA": 64},      # the kernel's look-ahead: 64 positions + one batch
    {"BTM_LA": 0},       # inserter as close to the parser as batches allow
    {"BTM_LA": 190},     # far ahead: long undo ranges
])
def test_model_equals_oracle(built, knobs):
    err = run_cases(knobs)
    line = [l for l in err.splitlines() if l.startswith("bt_model: sub-blocks")]
    assert line, err[-500:]
    s = line[-1]
    # the paths the kernel's exactness rests on must have been walked
    assert int(re.search(r"sub-blocks pipe (\d+)", s).group(1)) > 1000, s
    assert int(re.search(r"fallback (\d+)", s).group(1)) > 10, s                 # tree ring wraps (custom geometry)
    assert int(re.search(r"same-hash lanes (\d+)", s).group(1)) > 1000, s         # serialised descents inside a batch
    assert int(re.search(r"long-match events (\d+)", s).group(1)) > 100, s        # skip rule: undo + replay
    assert int(re.search(r"positions undone (\d+)", s).group(1)) > 1000, s
    assert int(re.search(r"extensions (\d+)", s).group(1)) > 100, s               # capped lengths extended on demand
    assert int(re.search(r"shadowed steps (\d+)", s).group(1)) > 50, s           # ring slots of later positions of the batch (custom geometry)
    h = [l for l in err.splitlines() if l.startswith("bt_model: histograms")][-1]
    assert int(re.search(r"blocked lane-rounds (\d+)", h).group(1)) > 10000, h    # pipelined same-hash chains: lanes held back by an open slot
