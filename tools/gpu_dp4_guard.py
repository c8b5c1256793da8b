#!/usr/bin/env python3
"""Stress of the id guard (csc_kernels_dp4.inc: d5_refresh_ids) in the -DCSCMI_TIMERS build: the service wavefront is slowed by
~50 k cycles per round (debug mask bit 63), so re-based masks arrive dozens of nodes late; the spine must then WAIT for them
instead of letting an id outlive its entry.  The streams must still be the oracle's, with the guard's wait counter > 0.
make -C csc_amd/csrc dev;  gpurun -- python tools/gpu_dp4_guard.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import cases
from csc_amd import corpus
from csc_amd.capi import CscLib


def run(lib, orc, za, data, dict_size, mask):
    lib.lib.CSCMI_DebugSetMask.argtypes = [C.c_void_p, C.c_uint64]
    lib.lib.CSCMI_DebugTimers.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    waits = []

    def arm(h):
        lib.lib.CSCMI_DebugSetMask(h, mask)
        waits.append(h)
    props = lib.props_init(min(dict_size, len(data)), 3)
    from csc_amd.capi import BytesWriter, BytesReader
    w = BytesWriter(); r = BytesReader(data)
    h = lib.lib.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
    w.out += lib.write_properties(props)
    lib.lib.CSCMI_DebugSetMask(h, mask)
    t0 = time.time()
    rc = lib.lib.CSCEnc_Encode(h, C.cast(r.ptr(), C.c_void_p), None)
    rc2 = lib.lib.CSCEnc_Encode_Flush(h)
    dt = time.time() - t0
    tm = (C.c_uint64 * 16)(); lib.lib.CSCMI_DebugTimers(h, tm)
    lib.lib.CSCEnc_Destroy(h)
    rc3, want = orc.encode(data, props=props, alloc=za)
    return rc == 0 and rc2 == 0 and bytes(w.out) == want, int(tm[15]), dt


if __name__ == "__main__":
    lib = CscLib(os.path.join(ROOT, "csc_amd", "csrc", "build", "dev", "libcsc_mi355x_timers.so"))
    orc = CscLib(os.path.join(ROOT, "oracle", "liborc.so")); orc.lib.orc_zero_alloc.restype = C.c_void_p
    za = orc.lib.orc_zero_alloc()
    allok = True
    for name, data, dsz in (("text 1 MiB", corpus.fill("text", corpus.SEED_ENWIK9, 0, 1 << 20).tobytes(), 64 << 20),
                            ("exe 512 KiB", corpus.fill("exe", corpus.SEED_EXE, 0, 1 << 19).tobytes(), 64 << 20),
                            ("window_wrap_32k", cases.build(cases.STREAM_CASES["window_wrap_32k"][0]), cases.STREAM_CASES["window_wrap_32k"][1]),
                            ("periodic", cases.build(cases.STREAM_CASES["periodic_5000x200"][0]), 1 << 20)):
        for mask in (0, 1 << 63):
            ok, waits, dt = run(lib, orc, za, data, dsz, mask)
            print(f"{name:18s} service {'slowed' if mask else 'normal'}: {'bit-exact' if ok else 'DIFF'}, guard waits {waits}, {len(data)/1e6/dt:.3f} MB/s", flush=True)
            allok &= ok
    print("ALL OK" if allok else "SOME DIFF")
    sys.exit(0 if allok else 1)
