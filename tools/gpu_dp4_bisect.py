#!/usr/bin/env python3
"""Development probe (the -DCSCMI_TIMERS build): a level-3 stream with parts of the spine / edge paths switched off.
gpurun -- python tools/gpu_dp4_bisect.py MiB byte_offset mask[,mask...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter
mib, boff = int(sys.argv[1]), int(sys.argv[2])
masks = [int(x, 0) for x in sys.argv[3].split(",")]
thr = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lib = CscLib(os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so"))
orc.lib.orc_zero_alloc.restype = C.c_void_p
za = orc.lib.orc_zero_alloc()
# (KIND / NBYTES / DICT in the environment: another corpus kind, a byte count instead of MiB, another dictionary size -- a case of tools/gpu_soak.py)
data = corpus.fill(os.environ.get("KIND", "text"), corpus.SEED_ENWIK9, boff, int(os.environ.get("NBYTES", mib << 20))).tobytes()
p = lib.props_init(int(os.environ.get("DICT", 64 << 20)), 3)
if os.environ.get("NOFILTERS"): p.DLTFilter = 0; p.TXTFilter = 0; p.EXEFilter = 0
rc2, want = orc.encode(data, props=p, alloc=za)
lib.lib.CSCMI_EncodeHostChunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
lib.lib.CSCMI_DebugSetMask.argtypes = [C.c_void_p, C.c_uint64]
for mask in masks:
    w = BytesWriter()
    h = lib.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
    t0 = time.time()
    for off in range(0, len(data), 2 << 20):
        lib.lib.CSCMI_DebugSetMask(h, mask | ((thr // 16) << 8))
        lib.lib.CSCMI_EncodeHostChunk(h, data[off:off + (2 << 20)], min(2 << 20, len(data) - off))
    got = bytes(w.out)
    base = want.find(got[:64])          # (the oracle's wrapper writes the properties header first)
    wn = want[base:] if base >= 0 else want
    n = min(len(got), len(wn))
    first = next((i for i in range(n) if got[i] != wn[i]), n)
    lib.lib.CSCMI_DebugTrace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    tr = (C.c_uint64 * 768)(); lib.lib.CSCMI_DebugTrace(h, tr)
    if tr[58 * 12 + 11]:
        names = ["fwd hash", "ent hash", "ref_head", "rq head|tail", "lit req|done", "m_pos", "m_ids", "gen|ins_head", "fin ids", "kind|end", "a0 eid|slot", "pos"]
        for q in range(12):
            a, b = tr[58 * 12 + q], tr[59 * 12 + q]
            print(f"  {names[q]:14s} after split {a:#x}  after split+alone {b:#x} {'' if a == b else '  <-- differs'}")
    if tr[61 * 12]:
        for l in range(4):
            a, b, cc = tr[60 * 12 + 3 * l], tr[60 * 12 + 3 * l + 1], tr[60 * 12 + 3 * l + 2]
            if cc: print(f"  old id: rep {l} at sub-block position {a & 0xFFFFFFFF} (node {a >> 32}): id {b & 0xFFFFFFFF} age {cc} fwd {b >> 32:#x}")
    if tr[62 * 12] == 7:
        r = list(tr[62 * 12: 62 * 12 + 12]); u = list(tr[63 * 12: 63 * 12 + 12])
        print(f"  window at stream position {r[1]}: kind {r[2]} end {r[3]} a0l {r[4]} a0code {r[5]} a0slot {r[6]}; exit label's rep distances {r[7:11]} state-at-start {r[11]}; L->rep at window start {u[1:5]} rid {u[7] & 0xFFFFFFFF:#x} {u[7] >> 32:#x} {u[8] & 0xFFFFFFFF:#x} {u[8] >> 32:#x}; fwd[rid0] {u[9]:#x} base of rid0's mask {u[10]} rq head/tail {u[11] & 0xFFFFFFFF}/{u[11] >> 32}")
        for q in range(0, min(60, r[3] + 1)):
            dcode, bk = tr[q * 12], tr[q * 12 + 1]
            print(f"    node {q:2d}: dist code {dcode:6d} back {bk & 0xFFFF:3d} state {(bk >> 16) & 0x3F:2d}")
    elif tr[62 * 12] == 2:
        r = list(tr[62 * 12: 62 * 12 + 12])
        print(f"  length-price table differs after the window at {r[1]}: split {r[2]:#x} alone {r[3]:#x}; lp at open {r[4]} after {r[5]} end {r[6]}")
    elif tr[62 * 12]:
        r = list(tr[62 * 12: 62 * 12 + 12])
        for q in range(2, 30):
            a = tr[q * 12: q * 12 + 4]
            b = tr[q * 12: q * 12 + 12]
            print(f"    node {q}: alone price {b[4]} | split price {b[5]} (path {b[11] >> 32}) A {b[6]} D {b[7] & 0xFFFFF} back {(b[7] >> 20) & 0xFFF} B {b[8]} D {b[9] & 0xFFFFF} back {(b[9] >> 20) & 0xFFF} own edge {b[10]} lp {b[11] & 0xFFFFFFFF}")
            if (a[0], a[1]) != (a[2], a[3]): print(f"    node {q}: split label {a[0]} back {a[1]} | alone label {a[2]} back {a[3]}")
        u = list(tr[63 * 12: 63 * 12 + 12])
        print(f"  rep distances after: split {u[1:5]} slot {u[5]} | alone {u[6:10]} slot {u[10]}")
        print(f"  first differing window at stream position {r[1]}: split kind {r[2]} end {r[3]} lp {r[4]} a0l {r[5]} code {r[6]} | alone kind {r[7]} end {r[8]} lp {r[9]} a0l {r[10]} | lp at window open {r[11]}")
    print(f"mask {mask:#x}: {len(data)/1e6/(time.time()-t0):.3f} MB/s, {len(got)} B so far vs {len(want)}: {'prefix bit-exact' if first == n else f'DIFF at {first}'}", flush=True)
