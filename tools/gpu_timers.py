#!/usr/bin/env python3
"""Development aid: run a few chunks through the -DCSCMI_TIMERS build and print where the
wavefront's cycles go.  gpurun -- python tools/gpu_timers.py [level] [MiB]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = CscLib(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
PAIR = len(sys.argv) > 3
class St(C.Structure):
    _fields_ = [("chunks", C.c_uint64), ("input_bytes", C.c_uint64), ("output_bytes", C.c_uint64), ("encode_launches", C.c_uint64),
                ("encode_kernel_ms", C.c_double), ("analyze_kernel_ms", C.c_double), ("find", C.c_uint64), ("slide", C.c_uint64),
                ("bt", C.c_uint64), ("lit", C.c_uint64), ("match", C.c_uint64)]
KIND = os.environ.get("KIND", "text")
SEED = {"text": corpus.SEED_ENWIK9, "exe": corpus.SEED_EXE, "mix5": corpus.SEED_EXE, "silesia": 6, "delta": corpus.SEED_DELTA}.get(KIND, 1)
data = corpus.fill(KIND, SEED, int(os.environ.get("OFFSET", "0")), mib << 20).tobytes()
p = lib.props_init(int(os.environ.get("DICT_MIB", "64")) << 20, level)
w = BytesWriter()
h = lib.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
lib.lib.CSCMI_EncodeHostChunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
t0 = time.time()
for off in range(0, len(data), 2 << 20):
    lib.lib.CSCMI_EncodeHostChunk(h, data[off:off + (2 << 20)], min(2 << 20, len(data) - off))
dt = time.time() - t0
st = St(); lib.lib.CSCMI_GetStats.argtypes = [C.c_void_p, C.c_void_p]; lib.lib.CSCMI_GetStats(h, C.byref(st))
# raw timers live in the device KernelStats; fetch through a tiny extension
lib.lib.CSCMI_DebugTimers.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
tm = (C.c_uint64 * 16)(); lib.lib.CSCMI_DebugTimers(h, tm)
names = ["fm:hash+gather", "fm:slots+extend", "fm:replay+insert", "pricing", "dp:statefix", "dp:litprice+relax", "dp:exit(backward+encode+slide)",
         "slide_pos", "dict filter", "window memcpy", "lazy: symbol coding", "lazy: FindMatch pick", "-", "-", "-", "-"]
if PAIR:
    names = ["M pre:tables", "M pre:wait labels", "M pre:reps+lit", "M wait decision", "M crit", "M post:pricing", "M post:relax", "M way out",
             "H pre:tables", "H pre:wait labels", "H pre:reps+lit", "H wait decision", "H crit", "H post:pricing", "H post:relax", "-"]
if level == 5 and not PAIR:      # the inserter form (csc_kernels_bt.inc): parser wavefront 0-7, inserter wavefront 8-11
    names = ["P wait for the record", "P record + rep compare", "P acceptance", "P pricing", "P dp:statefix", "P dp:litprice+relax", "P dp:exit(backward+encode)", "P wait for the post wavefront",
             "I gather + same-key", "I HT2/HT3/far lengths", "I descents", "I skip events (undo + replay)", "Q pricing", "Q relaxation", "-", "-"]      # (slots 14 / 15: the finder wavefront of the round-5 experiment, not dispatched)
if level in (1, 2) and not PAIR:   # the inserter form of the lazy levels (csc_kernels_hp.inc): parser wavefront 0-11, inserter wavefront 12-15
    names = ["P wait for the record", "P record + rep compare", "P acceptance", "-", "-", "-", "-", "P slide check / event", "dict filter", "window memcpy",
             "P symbol -> token", "P FindMatch pick", "I gather + same-key + stores", "I lengths + record", "-", "I events (undo / exact slide)"]
tot = sum(tm[:8]) if level == 5 and not PAIR else sum(tm[:12]) if level in (1, 2) and not PAIR else sum(tm)
print(f"level {level}: {len(data)/1e6/dt:.3f} MB/s, kernel {st.encode_kernel_ms:.0f} ms, find_match {st.find}, slide {st.slide}, lit {st.lit}, match {st.match}, bt {st.bt} = redo_y {st.bt >> 16} redo_x {st.bt & 0xFFFF}")
for n, v in zip(names, tm):
    if v: print(f"  {n:34s} {v/1e6:10.1f} Mcyc  {100*v/tot:5.1f}%   {v/max(1,st.find):8.0f} cyc/find_match")
print(f"  total timed {tot/1e6:.1f} Mcyc -> {tot/max(1e-9,st.encode_kernel_ms*1e-3)/1e9:.2f} GHz-equivalent of kernel time")
