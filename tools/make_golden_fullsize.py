#!/usr/bin/env python3
"""Record what the REFERENCE (oracle/_ref, built from /root/reference/src) writes for the two big-geometry BASELINE configs at FULL
size: configs[2] -- the whole 211 957 760-byte silesia stand-in as ONE `-m5 -d256m` stream (binary tree; the BT ring of
78 157 824 positions wraps) -- and task 0 of configs[4] -- the first 1 250 000 004 bytes of the 10^10-byte mix as `-m2 -d1024m`
(the 1 GiB window fills and wraps).  CPU, one after the other, about eight minutes; writes tests/golden/fullsize_digests.json, which
tools/gpu_fullsize_cfg.py compares the HIP encoder with."""
import ctypes as C, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = {   # name -> (corpus name, offset, bytes, level, dict, task size the dictionary is clamped to)
    "silesia_m5_d256m": ("silesia.tar", 0, 211957760, 5, 256 << 20, 211957760),
    "mix5_m2_d1024m_task0": ("mix5", 0, 1250000004, 2, 1024 << 20, 1250000004),
}


def one(name):
    from csc_amd import corpus
    from csc_amd.capi import CscLib
    cname, off, n, level, dct, task = CASES[name]
    src = corpus.Source(cname)
    ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
    o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p
    data = src.read(off, n).tobytes()
    props = ref.props_init(min(dct, task), level)
    t0 = time.time()
    rc, s = ref.encode(data, props=props, alloc=o.orc_zero_alloc())
    assert rc == 0
    return name, {"corpus": cname, "offset": off, "input_bytes": n, "level": level, "dict": dct, "dict_size": int(props.dict_size),
                  "bt_size": int(props.bt_size), "stream_bytes": len(s), "sha256": hashlib.sha256(s).hexdigest(),
                  "reference_seconds": round(time.time() - t0, 1)}


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    path = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")
    out = json.load(open(path)) if os.path.exists(path) else {"what": "reference (oracle/_ref) streams of BASELINE configs[2] and task 0 of configs[4] at full size", "cases": {}}
    for nm in names:
        name, rec = one(nm)
        out["cases"][name] = rec
        print(name, rec, flush=True)
        json.dump(out, open(path, "w"), indent=1)
