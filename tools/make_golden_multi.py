#!/usr/bin/env python3
"""Record what the REFERENCE (oracle/_ref, built from /root/reference/src) produces for the whole 10^9-byte enwik9 stand-in as
`csarc -m3 -d64m` ONE stream (BASELINE configs[1]), `-p8` (configs[3]), `-p127` and `-p954` task streams: one digest per split = sha256 over the per-task sha256 digests, the figure
bench.py reports as `sha256_of_stream_sha256s`.  Runs on the CPU (8 processes, about a minute); writes
tests/golden/multi_stream_digests.json.  bench.py compares its GPU run with these at full size."""
import ctypes as C, hashlib, json, multiprocessing as mp, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOTAL, LEVEL, DICT = 10 ** 9, 3, 64 << 20


def one(args):
    off, n = args
    from csc_amd import corpus
    from csc_amd.capi import CscLib
    ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
    o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p
    data = corpus.fill("text", corpus.SEED_ENWIK9, off, n).tobytes()
    rc, s = ref.encode(data, props=ref.props_init(min(DICT, n), LEVEL), alloc=o.orc_zero_alloc())
    assert rc == 0
    return hashlib.sha256(s).digest(), len(s)


if __name__ == "__main__":
    from csc_amd import corpus
    out = {"what": "reference (oracle/_ref) over corpus kind=text seed=SEED_ENWIK9, 10^9 bytes, -m3 -d64m, split as csarc.cpp:532-543",
           "level": LEVEL, "dict": DICT, "total": TOTAL, "splits": {}}
    with mp.Pool(8) as pool:
        for S in (8, 1, 127, 954):
            sl = corpus.task_slices(TOTAL, S)
            res = pool.map(one, sl, chunksize=1 if S <= 8 else 4)
            out["splits"][str(len(sl))] = {"sha256_of_stream_sha256s": hashlib.sha256(b"".join(r[0] for r in res)).hexdigest(),
                                           "stream_bytes": sum(r[1] for r in res)}
            if S <= 8:   # few, big tasks: also one {size, sha256} per task stream (what each rank of bench.py --gpus N checks)
                out["splits"][str(len(sl))]["tasks"] = [{"offset": o, "input_bytes": n, "stream_bytes": r[1], "sha256": r[0].hex()}
                                                        for (o, n), r in zip(sl, res)]
            print(len(sl), out["splits"][str(len(sl))], flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "multi_stream_digests.json"), "w"), indent=1)
