#!/usr/bin/env python3
"""Development aid (round 5): the regime of MORE task streams than CUs, measured.  The whole enwik9 stand-in as S task
streams (-m3 -d64m, one batch call per chunk round) through the development build with
  * the product's choice: the pipeline form up to 256 streams, the one-wavefront form (four streams a CU) beyond;
  * CSCMI_D4_MULTI_MAX=4096: the pipeline form for every launch -- 256 streams resident (one twelve-wavefront workgroup a CU), the
    rest start as workgroups end;
  (round 5 also compared the turn-taking forms, CSCMI_DP_WAVES=2 / 4: retired in round 6.)
gpurun -- python tools/gpu_many_streams.py 954 2048   (stream counts)"""
import ctypes as C, hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    from csc_amd import corpus
    from csc_amd.capi import CscLib, BytesWriter
    S = int(sys.argv[2])
    lib = CscLib(os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
    L = lib.lib
    L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.CSCMI_FlushBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    src = corpus.Source("enwik9")
    total = int(os.environ.get("MS_BYTES", str(src.size)))
    slices = corpus.task_slices(total, S)
    whole = torch.from_numpy(src.read(0, total)).cuda()
    hs, ws = [], []
    for off, n in slices:
        p = lib.props_init(min(64 << 20, n), 3)
        w = BytesWriter(); h = L.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None); assert h
        w.out += lib.write_properties(p); hs.append(h); ws.append(w)
    n = len(hs); H = (C.c_void_p * n)(*hs)
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = 0
    while True:
        Z = [max(0, min(2 << 20, m - k * (2 << 20))) for _, m in slices]
        if not any(Z): break
        P = (C.c_void_p * n)(*[whole.data_ptr() + off + k * (2 << 20) for off, _ in slices])
        assert L.CSCMI_EncodeDeviceChunkBatch(n, H, P, (C.c_size_t * n)(*Z)) == 0
        k += 1
    assert L.CSCMI_FlushBatch(n, H) == 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    dig = hashlib.sha256(b"".join(hashlib.sha256(bytes(w.out)).digest() for w in ws)).hexdigest()[:16]
    print(f"{n} streams: {total / 1e6 / dt:8.1f} MB/s  ({dt:.2f} s, {k} chunk rounds)  digest {dig}", flush=True)
    sys.exit(0)
for S in [int(x) for x in sys.argv[1:]] or [954]:
    for label, env in (("product's choice", {}), ("pipeline form for every launch (256 resident)", {"CSCMI_D4_MULTI_MAX": "4096"}),
                       ):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(S)], env=dict(os.environ, **env), capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if "streams:" in l]
        print(f"{label:50s} {out[0] if out else 'FAILED ' + r.stderr[-300:]}", flush=True)
