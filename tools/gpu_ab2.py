#!/usr/bin/env python3
"""Development aid: several builds of the library on the SAME box, interleaved, on the workloads the round works on.
gpurun -- python tools/gpu_ab2.py m3,p8,m5,m2,dec name1 name2 ...
names: files under csc_amd/csrc/build/ab/<name>.so; 'cur' = the product library.  Every encode is checked against the first build's digest.
  m3  : 4 MiB of the enwik9 stand-in, -m3 -d64m, one stream (CSCMI_EncodeDeviceChunk, 2 MiB chunks)
  p8  : eight such streams at once (CSCMI_EncodeDeviceChunkBatch), 2 chunks each -> MB/s over all eight
  m5  : 4 MiB of the silesia stand-in, -m5 -d256m, one stream
  m2  : 4 MiB of mix5, -m2 -d1024m, one stream
  dec : decode of the m3 / m2 streams the first build wrote (CSCDec_Decode), MB/s of output"""
import ctypes as C, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter

what = sys.argv[1].split(",")
names = sys.argv[2:]
libs = {n: CscLib(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so") if n == "cur" else os.path.join(ROOT, "csc_amd", "csrc", "build", "ab", n + ".so")) for n in names}
for L in libs.values():
    L.lib.CSCMI_EncodeDeviceChunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.lib.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
CH = 2 << 20


def enc_streams(lib, datas, level, dict_size, batch=False):
    """-> (seconds, [stream bytes]) ; one stream: per-chunk calls; several: batch calls"""
    L = lib.lib
    hs, ws, devs = [], [], []
    for d in datas:
        p = lib.props_init(dict_size if dict_size else min(64 << 20, max(1, len(d))), level)
        w = BytesWriter()
        h = L.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
        assert h
        w.out += lib.write_properties(p)
        hs.append(h); ws.append(w)
        devs.append(torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda())
    torch.cuda.synchronize()
    n = len(hs)
    t0 = time.perf_counter()
    k = 0
    while True:
        Z = [max(0, min(CH, len(d) - k * CH)) for d in datas]
        if not any(Z):
            break
        if n == 1 and not batch:
            assert L.CSCMI_EncodeDeviceChunk(hs[0], C.c_void_p(devs[0].data_ptr() + k * CH), Z[0]) == 0
        else:
            P = (C.c_void_p * n)(*[t.data_ptr() + k * CH for t in devs])
            assert L.CSCMI_EncodeDeviceChunkBatch(n, (C.c_void_p * n)(*hs), P, (C.c_size_t * n)(*Z)) == 0
        k += 1
    dt = time.perf_counter() - t0
    for h in hs:
        assert L.CSCEnc_Encode_Flush(h) == 0
        L.CSCEnc_Destroy(h)
    return dt, [bytes(w.out) for w in ws]


WORK = {"m3": ("enwik9", 3, 64 << 20, 1), "p8": ("enwik9", 3, 64 << 20, 8), "m5": ("silesia.tar", 5, 256 << 20, 1), "m2": ("mix5", 2, 1 << 30, 1),
        "t2": ("enwik9", 2, 64 << 20, 1), "t1": ("enwik9", 1, 64 << 20, 1),          # text at the lazy / greedy level (the hp form)
        "s954": ("enwik9", 3, 64 << 20, 954),          # the whole stand-in as 954 task streams of ~1 MB, one batch launch (the one-wavefront form: > kD4MultiMax streams)
        "m3b": ("enwik9", 3, 64 << 20, 1), "m5b": ("silesia.tar", 5, 256 << 20, 1), "m2b": ("mix5", 2, 1 << 30, 1)}      # ..b: one stream through the batch call (the multi instance of the kernel)
mib = int(os.environ.get("AB_MIB", "4"))
streams_for_dec = {}
for wk in what:
    if wk == "dec":
        continue
    cname, level, dsz, ns = WORK[wk]
    src = corpus.Source(cname)
    if wk == "s954":
        from csc_amd import corpus as _c
        datas = [src.read(off, n).tobytes() for off, n in _c.task_slices(src.size, ns)]
    else:
        datas = [src.read(i * (100 << 20), mib << 20).tobytes() for i in range(ns)]
    res = {n: [] for n in names}; dig = {}
    for rep in range(int(os.environ.get("AB_REPS", "3"))):
        for n in names:
            dt, outs = enc_streams(libs[n], datas, level if wk != "s954" else level, dsz if wk != "s954" else None, batch=wk.endswith("b"))
            res[n].append(sum(len(d) for d in datas) / 1e6 / dt)
            dig[n] = hashlib.sha256(b"".join(outs)).hexdigest()[:12]
            if wk in ("m3", "m2") and n == names[0]:
                streams_for_dec[wk] = (outs[0], datas[0])
    for n in names:
        print(f"{wk:3s} {n:12s} best {max(res[n]):8.3f} MB/s  all {' '.join(f'{v:.3f}' for v in res[n])}  sha {dig[n]} {'' if dig[n] == dig[names[0]] else 'DIFFERS'}", flush=True)
if "dec" in what:
    for wk, (s, data) in streams_for_dec.items():
        for n in names:
            best = 0
            for rep in range(3):
                t0 = time.perf_counter(); rc, back = libs[n].decode(s); dt = time.perf_counter() - t0
                assert rc == 0 and back == data
                best = max(best, len(data) / 1e6 / dt)
            print(f"dec {wk} {n:12s} best {best:8.3f} MB/s", flush=True)
