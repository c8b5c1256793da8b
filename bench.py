#!/usr/bin/env python3
"""bench.py -- encode MB/s of the MI355X-native libcsc path on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload: the enwik9 stand-in (csc_amd/corpus.py, kind "text", seed 0xC5C00002, 10^9 bytes) at
`-m3 -d64m`.  N = 1 is BASELINE.json configs[1] (one libcsc stream: HT6 match finder + advanced
parser).  N > 1 is configs[3]: the archiver's single-file `-pN` split (csarc.cpp:532-543), task r on
rank r, one process per GPU -- tasks are independent streams, so there is NO data-path collective;
torch.distributed (RCCL) only carries the barrier and the max-over-ranks time.  A "step" is one
raw_blocksize (2 MiB) chunk = one CSCEncoder::Compress call (csc_enc.cpp:170-181) per rank, with the
chunk already resident in HBM (CSCMI_EncodeDeviceChunk).  Weak scaling: per-GPU work is fixed.

One JSON line on rank 0: the contract fields + `roofline` (dominant kernel k_encode_runs against the
8 TB/s HBM peak, algorithmic bytes per SURVEY.md section 8d) + `cpu_baseline` (the reference built as
oracle/_ref -- or the oracle port -- timed on this box's host cores over a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOTAL = 10 ** 9                     # enwik9
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
ALG_BYTES = {1: 33.0, 2: 96.0, 3: 42.0, 4: 96.0}   # B_HT(w) per input byte (+ ratio r)


class CSCMIStats(C.Structure):
    _fields_ = [("chunks", C.c_uint64), ("input_bytes", C.c_uint64), ("output_bytes", C.c_uint64),
                ("encode_launches", C.c_uint64), ("encode_kernel_ms", C.c_double),
                ("analyze_kernel_ms", C.c_double), ("find_match_calls", C.c_uint64),
                ("slide_positions", C.c_uint64), ("bt_steps", C.c_uint64), ("literals", C.c_uint64),
                ("matches", C.c_uint64)]


def parse_size(s):
    s = s.lower()
    mul = 1
    if s.endswith("k"):
        mul, s = 1024, s[:-1]
    elif s.endswith("m"):
        mul, s = 1 << 20, s[:-1]
    return int(s) * mul


def cpu_baseline(data, level, dict_size, task_size, sample_bytes):
    """the reference's own encoder (oracle/_ref) -- or the oracle port -- on this box's host cores"""
    from csc_amd.capi import CscLib
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
    orc_path = os.path.join(ROOT, "oracle", "liborc.so")
    kind = "reference" if os.path.exists(ref_path) else "port"
    lib = CscLib(ref_path if kind == "reference" else orc_path)
    za = None
    if os.path.exists(orc_path):
        o = C.CDLL(orc_path)
        o.orc_zero_alloc.restype = C.c_void_p
        za = o.orc_zero_alloc()      # deterministic flush byte
    props = lib.props_init(min(dict_size, task_size), level)
    sample = data[:sample_bytes]
    t0 = time.perf_counter()
    rc, stream = lib.encode(sample, props=props, alloc=za)
    dt = time.perf_counter() - t0
    assert rc == 0
    return {"value": round(len(sample) / 1e6 / dt, 3), "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": f"first {len(sample)} bytes of the same stream, same CSCProps, 1 thread, "
                      f"{'oracle/_ref (reference sources, g++ -O4)' if kind == 'reference' else 'oracle/liborc.so (C port)'}",
            "seconds": round(dt, 2), "ratio": round(len(stream) / max(1, len(sample)), 4)}, stream


def decode_all(lib, streams, whole, slices, width=256):
    """decode every task stream with CSCMI_DecodeBatch (waves of `width` handles) and compare with the input"""
    import numpy as np
    import torch
    from csc_amd.capi import BytesReader, BytesWriter, CSC_PROP_SIZE
    L = lib.lib
    L.CSCMI_DecodeBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.CSCMI_DecodeBatch.restype = C.c_int
    host = whole.cpu().numpy()
    ok, total = True, 0
    t0 = time.perf_counter()
    for a in range(0, len(streams), width):
        part = streams[a:a + width]
        rs, ws, hs = [], [], []
        for s in part:
            props = lib.read_properties(s[:CSC_PROP_SIZE])
            r = BytesReader(s[CSC_PROP_SIZE:]); w = BytesWriter()
            h = L.CSCDec_Create(C.byref(props), C.cast(r.ptr(), C.c_void_p), None)
            if not h:
                return {"error": "CSCDec_Create failed"}
            rs.append(r); ws.append(w); hs.append(h)
        n = len(hs)
        R = (C.c_int * n)()
        rc = L.CSCMI_DecodeBatch(n, (C.c_void_p * n)(*hs), (C.c_void_p * n)(*[C.cast(w.ptr(), C.c_void_p) for w in ws]), R)
        for h in hs:
            L.CSCDec_Destroy(h)
        if rc != 0 or any(R[i] != 0 for i in range(n)):
            return {"error": f"batch decode failed rc={rc}"}
        for i, w in enumerate(ws):
            off, size = slices[a + i]
            total += len(w.out)
            ok = ok and len(w.out) == size and bool(np.array_equal(np.frombuffer(w.out, dtype=np.uint8), host[off:off + size]))
    dt = time.perf_counter() - t0
    return {"value": round(total / 1e6 / dt, 3), "unit": "MB/s", "seconds": round(dt, 2), "roundtrip_ok": ok,
            "what": f"the same task streams decoded by CSCMI_DecodeBatch, {min(width, len(streams))} per launch, incl. compare"}


def multi_stream_job(lib, stream_counts, level, dict_size):
    """Secondary measurement (extra `multi_stream` field of the JSON line, never `value`): the archiver's
    task split is where this path shards, so one GPU can run every task of `csarc a -m3 -d64m -p<S>` at
    once -- one workgroup per task through CSCMI_EncodeDeviceChunkBatch.  Each entry encodes the WHOLE
    10^9-byte stand-in.  S = 127 is the largest -p the reference CLI accepts for one file (u8 nfrags)."""
    import hashlib
    import torch
    from csc_amd import corpus
    from csc_amd.capi import BytesWriter
    L = lib.lib
    L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    whole = torch.from_numpy(corpus.fill("text", corpus.SEED_ENWIK9, 0, TOTAL)).cuda()
    chunk = 2 << 20
    results = []
    for S in stream_counts:
        slices = corpus.task_slices(TOTAL, S)
        S = len(slices)
        hs, ws = [], []
        for off, n in slices:
            props = lib.props_init(min(dict_size, n), level)
            w = BytesWriter()
            h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
            if not h:
                return {"multi_stream": "CSCEnc_Create failed"}
            w.out += lib.write_properties(props)
            hs.append(h); ws.append(w)
        H = (C.c_void_p * S)(*hs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, k = 0, 0
        while True:
            Z = [max(0, min(chunk, n - k * chunk)) for _, n in slices]
            if not any(Z):
                break
            P = (C.c_void_p * S)(*[whole.data_ptr() + off + k * chunk for off, _ in slices])
            rc = L.CSCMI_EncodeDeviceChunkBatch(S, H, P, (C.c_size_t * S)(*Z))
            if rc != 0:
                return {"multi_stream": f"batch encode failed rc={rc}"}
            total += sum(Z)
            k += 1
        for h in hs:
            L.CSCEnc_Encode_Flush(h)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = sum(len(w.out) for w in ws)
        digest = hashlib.sha256(b"".join(hashlib.sha256(bytes(w.out)).digest() for w in ws)).hexdigest()
        for h in hs:
            L.CSCEnc_Destroy(h)
        # and back: every task stream through the HIP decoder, 256 streams per launch (one per CU), checked byte for byte
        dec = decode_all(lib, [bytes(w.out) for w in ws], whole, slices)
        balg = ALG_BYTES.get(level, 42.0) + out / total
        # what the REFERENCE produces for this split (tests/golden/multi_stream_digests.json, recorded by
        # tools/make_golden_multi.py from oracle/_ref): the whole 10^9 bytes, bit for bit
        bit_exact = None
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "multi_stream_digests.json")))
            if gold["level"] == level and gold["dict"] == dict_size and str(S) in gold["splits"]:
                bit_exact = bool(gold["splits"][str(S)]["sha256_of_stream_sha256s"] == digest)
        except (OSError, ValueError, KeyError):
            pass
        results.append({"what": f"whole enwik9 stand-in (10^9 B) as csarc -m{level} -d64m -p{S}: {S} independent task streams, one workgroup each, 1 GPU",
                        "value": round(total / 1e6 / dt, 3), "unit": "MB/s", "seconds": round(dt, 2), "ratio": round(out / total, 4),
                        "streams": S, "batch_launches": k, "hbm_roofline_frac": round(balg * total / dt / 1e9 / HBM_PEAK_GBS, 8),
                        "sha256_of_stream_sha256s": digest, "bit_exact_vs_reference_digest": bit_exact, "decode": dec})
    return {"multi_stream": results}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--dict", default="64m")
    ap.add_argument("--cpu-sample-mib", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N > 1: skip the RCCL hand-over of the encoded streams to rank 0 (outside the timed region)")
    ap.add_argument("--multi-streams", default="127,954",
                    help="extra (N=1 only): the WHOLE 10^9-byte file as -p<S> task splits, all tasks concurrently on this GPU; '' = skip")
    args = ap.parse_args()

    import numpy as np
    import torch
    import csc_amd
    from csc_amd import corpus
    from csc_amd.capi import BytesWriter

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # CSC_BENCH_BACKEND=gloo (tests only): the N > 1 code path with several ranks sharing one GPU; the driver's runs use RCCL
    backend = os.environ.get("CSC_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend=backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    level, dict_size = args.level, parse_size(args.dict)
    lib = csc_amd.load()
    L = lib.lib
    L.CSCMI_EncodeDeviceChunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.CSCMI_EncodeDeviceChunk.restype = C.c_int
    L.CSCMI_GetStats.argtypes = [C.c_void_p, C.POINTER(CSCMIStats)]

    # this rank's task = slice `rank` of the -p{world} split of the 10^9-byte file
    slices = corpus.task_slices(TOTAL, world)
    off, task_size = slices[rank % len(slices)]
    props = lib.props_init(min(dict_size, task_size), level)          # csa_worker.cpp:35
    chunk = int(props.raw_blocksize)
    nsteps = args.warmup + args.steps
    nbytes = min(task_size, nsteps * chunk)
    host = corpus.fill("text", corpus.SEED_ENWIK9, off, nbytes)
    dev = torch.from_numpy(host).cuda()                                # inputs resident in HBM
    torch.cuda.synchronize()

    writer = BytesWriter()
    h = L.CSCEnc_Create(C.byref(props), C.cast(writer.ptr(), C.c_void_p), None)
    if not h:
        raise SystemExit("CSCEnc_Create failed")
    writer.out += lib.write_properties(props)

    def step(i):
        n = min(chunk, nbytes - i * chunk)
        rc = L.CSCMI_EncodeDeviceChunk(h, C.c_void_p(dev.data_ptr() + i * chunk), n)
        if rc != 0:
            raise SystemExit(f"encode failed rc={rc}")
        return n

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    s0 = CSCMIStats()
    L.CSCMI_GetStats(h, C.byref(s0))
    barrier()
    t0 = time.perf_counter()
    timed_bytes = 0
    for i in range(args.warmup, nsteps):
        timed_bytes += step(i)
    barrier()
    dt = time.perf_counter() - t0
    s1 = CSCMIStats()
    L.CSCMI_GetStats(h, C.byref(s1))
    gpu_stream = bytes(writer.out)       # header + every finished chunk (no EOF yet): a prefix of the full stream

    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    tot_bytes = torch.tensor([float(timed_bytes)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_bytes, op=dist.ReduceOp.SUM)
    tmax, tot_bytes = float(tmax.item()), float(tot_bytes.item())

    # N > 1, outside the timed region: the one exchange of the sharded archiver (csc_amd/sharded.py) --
    # every rank's encoded stream so far goes to rank 0 as a device tensor over RCCL, rank 0 checks the digests
    exchange = None
    if world > 1 and not args.no_exchange:
        try:
            import hashlib
            from csc_amd import sharded
            meta = [None] * world
            dist.all_gather_object(meta, (len(gpu_stream), hashlib.sha256(gpu_stream).hexdigest()))
            barrier()
            te = time.perf_counter()
            got = sharded.gather_blobs(gpu_stream, 0)
            barrier()
            te = time.perf_counter() - te
            if rank == 0:
                ok = all(len(got[r]) == meta[r][0] and hashlib.sha256(got[r]).hexdigest() == meta[r][1] for r in range(world))
                exchange = {"what": "encoded task streams of all ranks -> rank 0 (all_gather of lengths + grouped RCCL send/recv of device tensors)",
                            "ok": bool(ok), "bytes": int(sum(m[0] for m in meta)), "seconds": round(te, 4)}
        except Exception as e:      # never lose the bench line to the hand-over
            exchange = {"ok": False, "error": repr(e)[:300]}

    if rank == 0:
        launches = s1.encode_launches - s0.encode_launches
        kern_ms = s1.encode_kernel_ms - s0.encode_kernel_ms
        in_b = s1.input_bytes - s0.input_bytes
        out_b = s1.output_bytes - s0.output_bytes
        ratio = out_b / max(1, in_b)
        balg = ALG_BYTES.get(level, 42.0) + ratio
        avg_ms = kern_ms / max(1, launches)
        bytes_per_launch = in_b / max(1, launches)
        achieved = balg * bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc runs of this very
        # command; counters cannot be read from inside the process).  The committed measurement is scaled by input bytes.
        traffic, traffic_note = None, "no PMC measurement committed for this configuration"
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ent = pmc.get(f"m{level}_d{args.dict}_single_stream")
            if ent and world == 1:
                traffic = round((ent["fetch_bytes_per_input_byte"] + ent["write_bytes_per_input_byte"]) * bytes_per_launch)
                traffic_note = ent["source"]
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "encode MB/s (10^6 input bytes / wall-clock) on the enwik9 stand-in, -m3 -d64m; stream bit-exact vs reference",
            "value": round(tot_bytes / 1e6 / tmax, 4),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(tmax * 1e3 / max(1, args.steps), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 (byte + 32-bit integer work; 12-bit probabilities, 64-bit range-coder low)",
            "data": "synthetic (seeded enwik9 stand-in, csc_amd/csrc/corpus.c kind=text seed=0xC5C00002)",
            "config": {"workload": f"enwik9-like 10^9 B, -m{level} -d{args.dict}"
                                   + (f" -p{world}: task r on GPU r (csarc.cpp:532-543), independent streams" if world > 1
                                      else " single stream (BASELINE.json configs[1])"),
                       "step": f"one {chunk}-byte chunk (CSCEncoder::Compress) per rank, input resident in HBM",
                       "dict_size": int(props.dict_size), "hash_bits": int(props.hash_bits),
                       "hash_width": int(props.hash_width), "good_len": int(props.good_len), "lz_mode": int(props.lz_mode)},
            "ratio": round(ratio, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "k_encode_runs", "launches": int(launches),
                         "avg_launch_ms": round(avg_ms, 3), "alg_bytes_per_input_byte": round(balg, 3),
                         "input_bytes_per_launch": round(bytes_per_launch, 1),
                         "note": "one stream = one workgroup of up to four parse wavefronts: the libcsc chain is latency/issue-bound, not bandwidth-bound"},
            "counters": {"find_match_calls": int(s1.find_match_calls - s0.find_match_calls),
                         "slide_positions": int(s1.slide_positions - s0.slide_positions),
                         "literals": int(s1.literals - s0.literals), "matches": int(s1.matches - s0.matches),
                         "analyze_kernel_ms": round(s1.analyze_kernel_ms - s0.analyze_kernel_ms, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = max(nbytes, args.cpu_sample_mib << 20)
            cpu_in = corpus.fill("text", corpus.SEED_ENWIK9, off, min(task_size, sample)).tobytes()
            base, cpu_stream = cpu_baseline(cpu_in, level, dict_size, task_size, len(cpu_in))
            line["cpu_baseline"] = base
            line["bit_exact_vs_cpu_baseline"] = bool(cpu_stream[:len(gpu_stream)] == gpu_stream)
        else:
            line["cpu_baseline"] = None
        if exchange is not None:
            line["exchange"] = exchange
    L.CSCEnc_Encode_Flush(h)
    L.CSCEnc_Destroy(h)
    if rank == 0:
        if world == 1 and args.multi_streams:
            # extra field, not `value`: every task of the -p<S> split at once on this one GPU
            line.update(multi_stream_job(lib, [int(x) for x in args.multi_streams.split(",")], level, dict_size))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
