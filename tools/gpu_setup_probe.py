"""What one task stream's set-up costs on the box, call by call (HIP runtime through ctypes; nothing of the product is loaded):
pinned allocations of the sizes CSCEnc_Create asks for, device allocations, stream and event creation, small copies + syncs.
Prints mean microseconds per call.  Usage: python3 tools/gpu_setup_probe.py [n]"""
import ctypes as C
import sys
import time

hip = C.CDLL("libamdhip64.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def chk(rc):
    assert rc == 0, rc


def timed(label, fn, reps=n):
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:52s} {dt * 1e6:10.1f} us")


chk(hip.hipSetDevice(0))
chk(hip.hipFree(None))
ptrs = [C.c_void_p() for _ in range(n)]
for mb in (0.02, 1, 4, 7):
    sz = int(mb * (1 << 20))
    timed(f"hipHostMalloc {mb} MiB", lambda i: chk(hip.hipHostMalloc(C.byref(ptrs[i]), C.c_size_t(sz), 0)))
    timed(f"hipHostFree   {mb} MiB", lambda i: chk(hip.hipHostFree(ptrs[i])))
for mb in (2, 8, 30, 60):
    sz = int(mb * (1 << 20))
    timed(f"hipMalloc {mb} MiB", lambda i: chk(hip.hipMalloc(C.byref(ptrs[i]), C.c_size_t(sz))))
    timed(f"hipFree   {mb} MiB", lambda i: chk(hip.hipFree(ptrs[i])))
streams = [C.c_void_p() for _ in range(n)]
timed("hipStreamCreateWithFlags(nonblocking)", lambda i: chk(hip.hipStreamCreateWithFlags(C.byref(streams[i]), 1)))
evs = [C.c_void_p() for _ in range(n * 8)]
timed("hipEventCreate", lambda i: chk(hip.hipEventCreate(C.byref(evs[i]))), n * 8)
timed("hipEventDestroy", lambda i: chk(hip.hipEventDestroy(evs[i])), n * 8)
d = C.c_void_p()
h = C.c_void_p()
chk(hip.hipMalloc(C.byref(d), C.c_size_t(64 << 20)))
chk(hip.hipHostMalloc(C.byref(h), C.c_size_t(64 << 20), 0))
st = streams[0]
timed("hipMemsetAsync 30 MiB + sync", lambda i: (chk(hip.hipMemsetAsync(d, 0, C.c_size_t(30 << 20), st)), chk(hip.hipStreamSynchronize(st))))
timed("hipMemcpyAsync D2H 8 B + sync", lambda i: (chk(hip.hipMemcpyAsync(h, d, C.c_size_t(8), 2, st)), chk(hip.hipStreamSynchronize(st))))
timed("hipMemcpyAsync D2H 8 B (queued, no sync)", lambda i: chk(hip.hipMemcpyAsync(h, d, C.c_size_t(8), 2, st)))
chk(hip.hipStreamSynchronize(st))
timed("hipMemcpyAsync D2H 400 KiB + sync", lambda i: (chk(hip.hipMemcpyAsync(h, d, C.c_size_t(400 << 10), 2, st)), chk(hip.hipStreamSynchronize(st))))
timed("hipMemcpyAsync D2H 400 KiB (queued)", lambda i: chk(hip.hipMemcpyAsync(h, d, C.c_size_t(400 << 10), 2, st)))
chk(hip.hipStreamSynchronize(st))
timed("hipMemcpyAsync D2H 64 MiB + sync", lambda i: (chk(hip.hipMemcpyAsync(h, d, C.c_size_t(64 << 20), 2, st)), chk(hip.hipStreamSynchronize(st))), 10)
timed("hipStreamDestroy", lambda i: chk(hip.hipStreamDestroy(streams[i])))
