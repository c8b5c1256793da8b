#!/usr/bin/env python3
"""Record what the REFERENCE (oracle/_ref, built from /root/reference/src) has written after each of the first CHUNKS chunks of
every task of the `-m3 -d64m -p8` split of the 10^9-byte enwik9 stand-in (task 0 of -p8 and the unsplit file share their first
59 chunks: same CSCProps, same bytes).  After chunk c the encoder has flushed its coder (csc_encoder_main.cpp:140-145), so the
bytes written so far are a prefix of the task's stream; `bench.py --gpus N` encodes W + K chunks of every task and compares
{stream_bytes, sha256} with these.  Runs on the CPU (8 processes, well under a minute); writes
tests/golden/split_prefix_digests.json."""
import ctypes as C, hashlib, json, multiprocessing as mp, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOTAL, LEVEL, DICT, CHUNKS, CHUNK = 10 ** 9, 3, 64 << 20, 48, 2 << 20


def one(args):
    off, n = args
    from csc_amd import corpus
    from csc_amd.capi import CscLib, BytesWriter
    ref = CscLib(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so"))
    o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p
    data = corpus.fill("text", corpus.SEED_ENWIK9, off, min(n, CHUNKS * CHUNK)).tobytes()
    w = BytesWriter()
    marks = []
    rc, s = ref.encode(data, props=ref.props_init(min(DICT, n), LEVEL), alloc=o.orc_zero_alloc(), writer=w,
                       progress=lambda a, b: marks.append((a, len(w.out))))      # Progress after every chunk, csc_enc.cpp:180-181
    assert rc == 0 and len(marks) == (len(data) + CHUNK - 1) // CHUNK
    return {str(c + 1): {"input_bytes": a, "stream_bytes": m, "sha256": hashlib.sha256(s[:m]).hexdigest()} for c, (a, m) in enumerate(marks)}


if __name__ == "__main__":
    from csc_amd import corpus
    out = {"what": "reference (oracle/_ref) over corpus kind=text seed=SEED_ENWIK9, 10^9 bytes, -m3 -d64m, split as csarc.cpp:532-543: "
                   "bytes written (10-byte header included) after chunk c of task t, splits[S][t][c]",
           "level": LEVEL, "dict": DICT, "total": TOTAL, "chunk": CHUNK, "splits": {}}
    with mp.Pool(8) as pool:
        for S in (8,):
            sl = corpus.task_slices(TOTAL, S)
            res = pool.map(one, sl, chunksize=1)
            out["splits"][str(len(sl))] = {str(t): r for t, r in enumerate(res)}
    out["splits"]["1"] = {"0": out["splits"]["8"]["0"]}      # the unsplit file: same props (dict 64 MiB), same first 125 MB
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "split_prefix_digests.json"), "w"), indent=0)
    print("ok", {k: len(v) for k, v in out["splits"].items()})
