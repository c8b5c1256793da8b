"""CPU tier: tests/model/hp_model.c -- the ARRANGEMENT the HIP kernels give the hash-bucket match finder under the lazy parser
(levels 1 / 2, csc_amd/csrc/csc_kernels_hp.inc: a speculative inserter ahead of the parser, 64 positions per batch, per-position
records of the hash candidates, undo + exact replay where SlidePos deviates from find_match's insert rule) -- must produce the
oracle's bytes.  The model includes the oracle's encoder and replaces compress_normal only; both are test infrastructure."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "model", "libhpmodel.so")


@pytest.fixture(scope="module")
def built():
    src = [os.path.join(ROOT, "tests", "model", "hp_model.c"), os.path.join(ROOT, "oracle", "orc_decoder.c"), os.path.join(ROOT, "oracle", "zalloc.c")]
    subprocess.run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-Wl,-Bsymbolic", "-o", SO] + src + ["-lm"], check=True)
    return SO


def run_cases(knobs):
    code = f"""
import ctypes as C, os, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tests"))
import cases
from csc_amd import corpus
from csc_amd.capi import CscLib
orc = CscLib(os.path.join({ROOT!r}, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
mod = CscLib({SO!r}); mod.lib.orc_zero_alloc.restype = C.c_void_p
za, zb = orc.lib.orc_zero_alloc(), mod.lib.orc_zero_alloc()
n = 0
for name in ("empty", "one_byte", "zeros_8k", "abcdefgh_64k", "random_64k", "text_20k", "text_300k", "exe_300k", "mix_types", "dup_blocks",
             "ragged_tail_511", "short_reads_511", "window_wrap_32k", "periodic_5000x200", "delta_200k", "zeros_5m"):
    spec, dict_size, clamp, max_read = cases.STREAM_CASES[name]
    data = cases.build(spec)
    for lv in (1, 2):
        rc, a = orc.encode(data, lv, dict_size, alloc=za, clamp_dict=clamp, max_read=max_read)
        rc2, b = mod.encode(data, lv, dict_size, alloc=zb, clamp_dict=clamp, max_read=max_read)
        assert rc == 0 and rc2 == 0 and a == b, (name, lv, len(a), len(b))
        n += 1
# the mix of BASELINE configs[4] around a segment seam, its geometry (-m2 -d1024m: 23-bit x 8 bucket)
seg = 64 << 20
data = corpus.fill("mix5", corpus.SEED_EXE, seg - (1 << 20), 2 << 20).tobytes()
rc, a = orc.encode(data, props=orc.props_init(1 << 30, 2), alloc=za); rc2, b = mod.encode(data, props=mod.props_init(1 << 30, 2), alloc=zb)
assert rc == 0 and rc2 == 0 and a == b
n += 1
# custom geometry: bucket widths 1 .. 8, greedy parser, small and large good_len, a window that wraps
data = cases.build([["text", 13, 0, 500000], ["exe", 14, 0, 200000], ["pattern", "00", 70000], ["delta", 3, 0, 100000], ["text", 13, 0, 100000]])
for width, bits, good, mode, dsz in ((1, 16, 32, 2, 1 << 20), (3, 12, 8, 2, 1 << 18), (8, 10, 24, 1, 40000), (5, 18, 200, 2, 1 << 21), (8, 20, 24, 2, 1 << 22)):
    p = orc.props_init(dsz, 2); p.hash_width = width; p.hash_bits = bits; p.good_len = good; p.lz_mode = mode
    rc, a = orc.encode(data, props=p, alloc=za); rc2, b = mod.encode(data, props=p, alloc=zb)
    assert rc == 0 and rc2 == 0 and a == b, (width, bits, good, mode, dsz)
    n += 1
print("MODEL_OK", n)
"""
    env = dict(os.environ, HPM_STATS="1", **{k: str(v) for k, v in knobs.items()})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "MODEL_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    return out.stderr


@pytest.mark.parametrize("knobs", [
    {"HPM_LA": 128},     # the kernel's look-ahead: 128 positions + one batch
    {"HPM_LA": 0},       # inserter as close to the parser as batches allow
    {"HPM_LA": 190},     # far ahead: long undo ranges
])
def test_model_equals_oracle(built, knobs):
    err = run_cases(knobs)
    line = [l for l in err.splitlines() if l.startswith("hp_model:")]
    assert line, err[-500:]
    s = line[-1]
    # the paths the kernel's exactness rests on must have been walked
    assert int(re.search(r"sub-blocks pipe (\d+)", s).group(1)) > 1000, s
    assert int(re.search(r"deviations (\d+)", s).group(1)) > 100, s               # speculative inserts undone + replayed
    assert int(re.search(r"positions undone (\d+)", s).group(1)) > 1000, s
    assert int(re.search(r"no event needed (\d+)", s).group(1)) > 10000, s        # the common case: the speculation was SlidePos
    assert int(re.search(r"same-key pairs (\d+)", s).group(1)) > 1000, s          # buckets resolved inside a batch
    assert int(re.search(r"extensions (\d+)", s).group(1)) > 100, s               # capped lengths extended on demand
