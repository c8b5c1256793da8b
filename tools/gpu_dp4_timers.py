#!/usr/bin/env python3
"""Development aid: section timers of the pipeline form's master wavefront (the -DCSCMI_TIMERS build).
make -C csc_amd/csrc dev;  gpurun -- python tools/gpu_dp4_timers.py [MiB]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = CscLib(os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
class St(C.Structure):
    _fields_ = [("chunks", C.c_uint64), ("input_bytes", C.c_uint64), ("output_bytes", C.c_uint64), ("encode_launches", C.c_uint64),
                ("encode_kernel_ms", C.c_double), ("analyze_kernel_ms", C.c_double), ("find", C.c_uint64), ("slide", C.c_uint64),
                ("bt", C.c_uint64), ("lit", C.c_uint64), ("match", C.c_uint64)]
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
data = corpus.fill(kind, corpus.SEED_ENWIK9, 0, mib << 20).tobytes()
p = lib.props_init(64 << 20, 3)
w = BytesWriter()
h = lib.lib.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
lib.lib.CSCMI_EncodeHostChunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
t0 = time.time()
for off in range(0, len(data), 2 << 20):
    lib.lib.CSCMI_EncodeHostChunk(h, data[off:off + (2 << 20)], min(2 << 20, len(data) - off))
dt = time.time() - t0
st = St(); lib.lib.CSCMI_GetStats.argtypes = [C.c_void_p, C.c_void_p]; lib.lib.CSCMI_GetStats(h, C.byref(st))
lib.lib.CSCMI_DebugTimers.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
tm = (C.c_uint64 * 16)(); lib.lib.CSCMI_DebugTimers(h, tm)
names = ["wait: slide acknowledged", "window function (DP nodes)", "exit: back-trace", "way out: packets into lanes", "literal flags (incl. the window-less literals)", "waiting for the tree wavefront (d6_join)", "exit: length, event, rebase", "way out: match / rep packets, exit packet",
         "# deviations (undo + replay)", "# rep lengths by compare", "wait record (cyc/16)", "# record waits", "wait literal price (cyc/16)", "# literal waits", "# nodes on the straight-line path", "# waits for a re-based mask (id guard)"]
tot = sum(tm[:8])
print(f"{len(data)/1e6/dt:.3f} MB/s, kernel {st.encode_kernel_ms:.0f} ms, nodes {st.find}, slid {st.slide}, lit {st.lit}, match {st.match}")
for i, (n, v) in enumerate(zip(names, tm)):
    if not v: continue
    if n.startswith("#"): print(f"  {n:34s} {v:10d}")
    elif "cyc/16" in n: print(f"  {n:34s} {16*v/1e6:10.1f} Mcyc   {16*v/max(1,st.find):8.0f} cyc/node")
    else: print(f"  {n:34s} {v/1e6:10.1f} Mcyc  {100*v/tot:5.1f}%   {v/max(1,st.find):8.0f} cyc/node")
print(f"  total timed {tot/1e6:.1f} Mcyc over {st.encode_kernel_ms:.0f} ms")
lib.lib.CSCMI_DebugTrace.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
tr = (C.c_uint64 * 768)(); lib.lib.CSCMI_DebugTrace(h, tr)
wn = ["general: record, fwd", "general: rep lengths", "general: acceptance", "general: tables + relax", "general: exits, literal wait", "general: literal edge", "general: requests + rotate", "top: label + log (all nodes)", "straight-line node body", "straight-line: requests + rotate"]
for n, v in zip(wn, tr[:10]):
    print(f"    window: {n:30s} {16*v/1e6:10.1f} Mcyc   {16*v/max(1,st.find):8.0f} cyc/node")

if tr[12]:
    print(f"    in-situ probes ({tr[12]}): dependent LDS read {tr[10]/tr[12]:.0f} cycles (incl. 2 s_memtime), 8 dependent VALU mads + readfirstlane {tr[11]/tr[12]:.0f} cycles")

if tr[15]:
    nd = max(1, st.find)
    print(f"    spine: {tr[15]} windows, {16*tr[13]/nd:.0f} cyc/node in the node loop, {tr[14]} straight-line steps")
    for b in (0, 1):
        n = max(1, tr[18 + 3*b])
        print(f"    edge {b}: {n} nodes, wait for label {16*tr[16+3*b]/n:.0f} cyc/node, work {16*tr[17+3*b]/n:.0f} cyc/node")

if tr[15]:
    nw = tr[15]
    print(f"    spine per window ({nw} windows, {st.find/nw:.1f} nodes each): prologue {16*tr[24]/nw:.0f}, first run of straight-line steps (loop start -> the outer loop's next entry with k >= 1: normally the 8 nodes whose literal prices the spine computed itself) {16*tr[25]/nw:.0f}, "
          f"between runs of steps (ids, literal horizon) {16*tr[26]/nw:.0f} (ids taken over {tr[33]}, d5_refresh_ids called {tr[34]}), general steps {16*tr[27]/nw:.0f} ({tr[28]/nw:.2f} of them, {16*tr[27]/max(1,tr[28]):.0f} each), last run of straight-line steps (the outer loop's last entry -> loop end) {16*tr[29]/nw:.0f}, epilogue {16*tr[30]/nw:.0f}, loop total {16*tr[13]/nw:.0f}")
    for i, n in enumerate(names[:8]):
        if tm[i]: print(f"    per window: {n:34s} {tm[i]/nw:8.0f}")

if tr[15] and any(tr[36:40]):
    print(f"    edge wavefronts' nodes by path: straight {tr[36]}, rep0 the only rep candidate {tr[39]}, several rep candidates {tr[37]}, general {tr[38]}")

if tr[15] and any(tr[0:5]):
    print(f"    edge wavefronts' general nodes because: record not there yet {tr[0]} (waited {16*tr[1]/max(1,tr[0]):.0f} cycles each), refresh bits in the label {tr[2]}, a rep mask that cannot answer {tr[3]}, other (several reps incl. one the masks cannot give, good_len) {tr[4]}")

if tr[15] and any(tr[32:36]):
    print(f"    straight-line steps left for a general step because: equality bits that cannot answer (mask too old / based ahead / straddling) {tr[32]}, refresh zone {tr[35]}   (general steps in all: {tr[28]})")

if tr[15] and tr[40]:
    nw = tr[15]
    print(f"    queue room asked for (d6_room's slow path) {tr[40]} times ({tr[40]/nw:.2f} a window): join {16*tr[41]/tr[40]:.0f} cycles each, publish + wait for the coder wavefront {16*tr[42]/tr[40]:.0f} each ({16*(tr[41]+tr[42])/nw:.0f} a window), room afterwards {16*tr[46]/tr[40]:.0f} entries")
if tr[15] and tr[44]:
    print(f"    coder wavefront: {tr[44]} queue entries in {tr[45]} batches, {16*tr[43]/tr[44]:.0f} cycles an entry, busy {16*tr[43]/tr[15]:.0f} cycles a window")
