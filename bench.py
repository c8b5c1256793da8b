#!/usr/bin/env python3
"""bench.py -- encode MB/s of the MI355X-native libcsc path on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W

N = 1  BASELINE.json configs[1]: the enwik9 stand-in (csc_amd/corpus.py "enwik9": kind text, seed 0xC5C00002, 10^9 bytes;
       the real file when $CSC_CORPUS_DIR/enwik9 exists) as ONE libcsc stream at `-m3 -d64m` (HT6 match finder + advanced
       parser).  A step is one raw_blocksize (2 MiB) chunk = one CSCEncoder::Compress call (csc_enc.cpp:170-181) with the
       chunk already resident in HBM (CSCMI_EncodeDeviceChunk).  `value` covers chunks W .. W+K-1; `steady` repeats the
       measurement after the 64 MiB window has filled (chunks >= 32).
N > 1  configs[3]: the archiver's `-p8` split of the same file (csarc.cpp:532-543) -- the SAME eight tasks at every N, dealt
       largest first, task i of the dispatch order -> rank i mod N (csarc.cpp:355), one process per GPU.  A step advances every
       task of the rank by one chunk with ONE launch (CSCMI_EncodeDeviceChunkBatch: one workgroup per stream).  Tasks are
       independent streams, so there is NO data-path collective; torch.distributed (RCCL) carries the barrier, the
       max-over-ranks time and -- outside the timed region -- the per-task digests and the hand-over of the streams to rank 0.
       Total work is the same at every N ("scaling": "strong"), and so is `ratio`.  `--split 127` gives the second curve
       (16 tasks per GPU at N = 8).  When launched plainly (`python bench.py --gpus N`, no RANK in the environment) the
       script starts its N ranks itself: `python -m torch.distributed.run` as a CHILD process, before anything here has
       touched the GPU, and relays rank 0's JSON line.

One JSON line on rank 0: the contract fields + `roofline` (dominant kernel k_encode_runs* against the 8 TB/s HBM peak,
algorithmic bytes per SURVEY.md section 8d) + `cpu_baseline` (the reference built as oracle/_ref -- or the oracle port --
timed on this box's host cores over a bounded sample: one thread for the single stream, the reference's eight worker processes for
the task splits (rank 0's host, after the timed region, at every N), the reference archiver `csarc_ref -t8` for `--workload tree`).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D4_MULTI_MAX = 768                  # csc_kernels_dp4.inc kD4MultiMax: launches of up to this many level-3 streams take the pipeline form (tests/test_product_host.py keeps the two equal)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
ALG_BYTES = {1: 33.0, 2: 96.0, 3: 42.0, 4: 96.0}   # B_HT(w) per input byte (+ ratio r)
CONFIGS = {   # BASELINE.json configs -> (corpus name, level, dict)
    "enwik9": ("enwik9", 3, "64m"),
    "silesia": ("silesia.tar", 5, "256m"),
    "mix5": ("mix5", 2, "1024m"),
}


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


def alg_bytes(level, ratio, stats=None):
    """algorithmic bytes per input byte of the LZ pass"""
    if level == 5:
        cbar = (stats.bt_steps / max(1, stats.find_match_calls + stats.slide_positions)) if stats else 4.6
        return 41.0 + 13.0 * cbar + ratio
    return ALG_BYTES.get(level, 42.0) + ratio


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torchrun (this process has not imported
    torch or touched HIP) and relay rank 0's line.  Never exec: the box refuses an exec from a process that initialised the GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # --standalone: torchrun picks a free rendezvous port itself (no bind-then-close probe that another launcher could race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if p.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank child failed (rc={p.returncode})", file=sys.stderr)
        sys.exit(p.returncode or 1)
    print(lines[-1], flush=True)
    sys.exit(0)


def host_info():
    """host model + core count for the cpu_baseline objects"""
    model = "unknown"
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return {"host_cpu": model, "nproc": os.cpu_count() or 1, "usable_cores": usable}


def _cpu_task(job):
    """one task of the split through the reference (a process of its own: the reference's worker threads share nothing either)"""
    name, seed_kind, off, nbytes, level, dict_size, task_size = job
    from csc_amd import corpus
    from csc_amd.capi import CscLib
    src = corpus.Source(name)
    data = src.read(off, nbytes).tobytes()
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
    orc_path = os.path.join(ROOT, "oracle", "liborc.so")
    lib = CscLib(ref_path if os.path.exists(ref_path) else orc_path)
    za = None
    if os.path.exists(orc_path):
        o = C.CDLL(orc_path)
        o.orc_zero_alloc.restype = C.c_void_p
        za = o.orc_zero_alloc()
    props = lib.props_init(min(dict_size, task_size), level)
    t0 = time.perf_counter()
    rc, stream = lib.encode(data, props=props, alloc=za)
    return rc, len(data), len(stream), time.perf_counter() - t0


def cpu_baseline_split(src, slices, level, dict_size, sample_bytes):
    """the -p<S> line's CPU leg: min(8, tasks) worker processes of the reference (csarc -t8: csarc.cpp:200-201, one libcsc
    stream per task, csa_worker.cpp:23-56), each over the first `sample_bytes` of its task, started together"""
    import multiprocessing as mp
    hi = host_info()
    workers = min(8, len(slices), hi["usable_cores"])
    jobs = [(src.name, None, off, min(n, sample_bytes), level, dict_size, n) for off, n in slices[:8]]
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")) else "port"
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_cpu_task, jobs)
    dt = time.perf_counter() - t0
    assert all(r[0] == 0 for r in res)
    tot = sum(r[1] for r in res)
    out = dict(value=round(tot / 1e6 / dt, 3), unit="MB/s", cores=workers, kind=kind,
               sample=f"first {jobs[0][3]} bytes of each of the first {len(jobs)} tasks of the split, {workers} worker processes at once "
                      f"(the reference runs min(8, tasks) worker threads, csarc.cpp:200-201), same CSCProps per task, "
                      f"{'oracle/_ref (reference sources, g++ -O4)' if kind == 'reference' else 'oracle/liborc.so (C port)'}; wall-clock incl. process start",
               seconds=round(dt, 2), ratio=round(sum(r[2] for r in res) / max(1, tot), 4),
               per_worker_MBps=[round(r[1] / 1e6 / r[3], 3) for r in res])
    out.update(hi)
    return out


def cpu_baseline(data, level, dict_size, task_size):
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
    t0 = time.perf_counter()
    rc, stream = lib.encode(data, props=props, alloc=za)
    dt = time.perf_counter() - t0
    assert rc == 0
    out = {"value": round(len(data) / 1e6 / dt, 3), "unit": "MB/s", "cores": 1, "kind": kind,
           "sample": f"first {len(data)} bytes of the same stream, same CSCProps, 1 thread, "
                     f"{'oracle/_ref (reference sources, g++ -O4)' if kind == 'reference' else 'oracle/liborc.so (C port)'}",
           "seconds": round(dt, 2), "ratio": round(len(stream) / max(1, len(data)), 4)}
    out.update(host_info())
    return out, stream


def decode_all(lib, streams, whole, slices, width=256):
    """decode every task stream with CSCMI_DecodeBatch (waves of `width` handles) and compare with the input"""
    import numpy as np
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


def golden(name):
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", name)))
    except (OSError, ValueError):
        return None


def multi_stream_job(lib, src, stream_counts, level, dict_size):
    """Secondary measurement (extra `multi_stream` field of the JSON line, never `value`): the archiver's
    task split is where this path shards, so one GPU can run every task of `csarc a -m3 -d64m -p<S>` at
    once -- one workgroup per task through CSCMI_EncodeDeviceChunkBatch.  Each entry encodes the WHOLE
    file.  S = 127 is the largest -p the reference CLI accepts for one file (u8 nfrags)."""
    import torch
    from csc_amd import corpus
    from csc_amd.capi import BytesWriter
    L = lib.lib
    total_size = src.size
    whole = torch.from_numpy(src.read(0, total_size)).cuda()
    chunk = 2 << 20
    results = []
    for S in stream_counts:
        slices = corpus.task_slices(total_size, S)
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
        # the EOF + last coder blocks of all S task streams in ONE round trip (CSCMI_FlushBatch == S x CSCEnc_Encode_Flush,
        # tests/test_gpu_parity.py); round 4 flushed handle by handle here: 954 launches + waits inside the timed region
        rc = L.CSCMI_FlushBatch(S, H)
        if rc != 0:
            return {"multi_stream": f"batch flush failed rc={rc}"}
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = sum(len(w.out) for w in ws)
        digest = hashlib.sha256(b"".join(hashlib.sha256(bytes(w.out)).digest() for w in ws)).hexdigest()
        for h in hs:
            L.CSCEnc_Destroy(h)
        # and back: every task stream through the HIP decoder, 256 streams per launch (one per CU), checked byte for byte
        dec = decode_all(lib, [bytes(w.out) for w in ws], whole, slices)
        balg = alg_bytes(level, out / total)
        # what the REFERENCE produces for this split (tests/golden/multi_stream_digests.json, recorded by
        # tools/make_golden_multi.py from oracle/_ref): the whole 10^9 bytes, bit for bit
        bit_exact = None
        gold = golden("multi_stream_digests.json")
        if src.synthetic and gold and gold["level"] == level and gold["dict"] == dict_size and str(S) in gold["splits"]:
            bit_exact = bool(gold["splits"][str(S)]["sha256_of_stream_sha256s"] == digest)
        results.append({"what": f"whole {src.name} ({total_size} B) as csarc -m{level} -d{dict_size >> 20}m -p{S}: {S} independent task streams, one workgroup each, 1 GPU",
                        "value": round(total / 1e6 / dt, 3), "unit": "MB/s", "seconds": round(dt, 2), "ratio": round(out / total, 4),
                        "streams": S, "batch_launches": k, "hbm_roofline_frac": round(balg * total / dt / 1e9 / HBM_PEAK_GBS, 8),
                        "traffic": (tr := traffic_from_profile(f"m{level}_d{dict_size >> 20}m_p{S}", total))[0], "traffic_note": tr[1],
                        "sha256_of_stream_sha256s": digest, "bit_exact_vs_reference_digest": bit_exact, "decode": dec})
    return {"multi_stream": results}


def traffic_from_profile(key, bytes_per_launch):
    """HBM bytes per launch from the committed PMC passes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc runs; counters
    cannot be read from inside the process).  Only used when the passes were recorded for the kernel code this run loaded."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = pmc.get(key)
        so = hashlib.sha256(open(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"), "rb").read()).hexdigest()[:16]
        if not ent:
            return None, "no PMC measurement committed for this configuration"
        if ent.get("library_sha256_16") != so:
            return None, f"profiles/pmc_traffic.json was recorded for another build of the kernels ({ent.get('library_sha256_16')} != {so}): stale, not reported"
        return (round((ent["fetch_bytes_per_input_byte"] + ent["write_bytes_per_input_byte"]) * bytes_per_launch), ent["source"])
    except (OSError, ValueError, KeyError) as e:
        return None, f"no usable PMC measurement ({e!r})"


class Rank:
    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        # CSC_BENCH_BACKEND=gloo (tests only): the N > 1 code path with several ranks sharing one GPU; the driver's runs use RCCL
        self.backend = os.environ.get("CSC_BENCH_BACKEND", "nccl")
        if self.backend == "nccl" and local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py --gpus {args.gpus}: rank {self.rank} has no GPU of its own (this node shows {torch.cuda.device_count()}); one process per GPU over RCCL")
        dev = local_rank if self.backend == "nccl" else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
            else:
                dist.init_process_group(backend=self.backend)
        self.red_dev = "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, x, op):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.red_dev)
        if self.dist:
            self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def gather_objects(self, obj):
        if not self.dist:
            return [obj]
        box = [None] * self.world
        self.dist.all_gather_object(box, obj)
        return box

    def done(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def base_line(args, R, src, level, value, tsec, scaling, workload, props, ratio):
    return {
        "metric": f"encode MB/s (10^6 input bytes / wall-clock) on {src.name}{'' if not src.synthetic else ' stand-in'}, -m{level} -d{args.dict}; stream bit-exact vs reference",
        "value": round(value, 4), "unit": "MB/s",
        "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(tsec * 1e3 / max(1, args.steps), 3),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8/u32 (byte + 32-bit integer work; 12-bit probabilities, 64-bit range-coder low)",
        "data": src.label,
        "config": {"workload": workload,
                   "dict_size": int(props.dict_size), "hash_bits": int(props.hash_bits), "hash_width": int(props.hash_width),
                   "good_len": int(props.good_len), "lz_mode": int(props.lz_mode), "bt_size": int(props.bt_size)},
        "ratio": round(ratio, 4),
    }


def roofline(level, ratio, launches, kern_ms, in_b, kernel, note, traffic=(None, None), stats=None):
    balg = alg_bytes(level, ratio, stats)
    avg_ms = kern_ms / max(1, launches)
    bpl = in_b / max(1, launches)
    achieved = balg * bpl / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    return {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic[0], "traffic_note": traffic[1],
            "kernel": kernel, "launches": int(launches), "avg_launch_ms": round(avg_ms, 3),
            "alg_bytes_per_input_byte": round(balg, 3), "input_bytes_per_launch": round(bpl, 1), "note": note}


# ------------------------------------------------------------------------------------------------------------------
def run_single(args, R, lib, src, level, dict_size):
    """N = 1: one libcsc stream (BASELINE configs[1] by default)"""
    torch = R.torch
    from csc_amd.capi import BytesWriter
    L = lib.lib
    task_size = src.size
    props = lib.props_init(min(dict_size, task_size), level)          # csa_worker.cpp:35
    chunk = int(props.raw_blocksize)
    nsteps = args.warmup + args.steps
    if args.config != "enwik9":
        args.steady_steps = 0        # (a 256 MiB / 1 GiB window takes minutes to fill at these levels; the headline config only)
    steady_from = max(nsteps, (int(props.dict_size) + chunk - 1) // chunk) if args.steady_steps > 0 else nsteps
    nchunks = steady_from + (args.steady_steps if args.steady_steps > 0 else 0)
    nbytes = min(task_size, nchunks * chunk)
    host = src.read(0, nbytes)
    dev = torch.from_numpy(host).cuda()                                # inputs resident in HBM
    torch.cuda.synchronize()
    writer = BytesWriter()
    h = L.CSCEnc_Create(C.byref(props), C.cast(writer.ptr(), C.c_void_p), None)
    if not h:
        raise SystemExit("CSCEnc_Create failed")
    writer.out += lib.write_properties(props)

    def step(i):
        n = min(chunk, nbytes - i * chunk)
        if n <= 0:
            return 0
        rc = L.CSCMI_EncodeDeviceChunk(h, C.c_void_p(dev.data_ptr() + i * chunk), n)
        if rc != 0:
            raise SystemExit(f"encode failed rc={rc}")
        return n

    def stats():
        s = CSCMIStats()
        L.CSCMI_GetStats(h, C.byref(s))
        return s

    def region(a, b):
        s0 = stats()
        R.barrier()
        t0 = time.perf_counter()
        nb = 0
        for i in range(a, b):
            nb += step(i)
        R.barrier()
        dt = time.perf_counter() - t0
        return nb, dt, s0, stats()

    for i in range(args.warmup):
        step(i)
    timed_bytes, dt, s0, s1 = region(args.warmup, nsteps)
    gpu_stream = bytes(writer.out)       # header + every finished chunk (no EOF yet): a prefix of the full stream
    launches = s1.encode_launches - s0.encode_launches
    kern_ms = s1.encode_kernel_ms - s0.encode_kernel_ms
    in_b, out_b = s1.input_bytes - s0.input_bytes, s1.output_bytes - s0.output_bytes
    ratio = out_b / max(1, in_b)
    line = base_line(args, R, src, level, timed_bytes / 1e6 / dt, dt, "weak",
                     f"{src.name} {src.size} B, -m{level} -d{args.dict} single stream"
                     + (" (BASELINE.json configs[1])" if args.config == "enwik9" else f" (BASELINE.json config '{args.config}')")
                     + f"; step = one {chunk}-byte chunk (CSCEncoder::Compress), input resident in HBM; timed chunks {args.warmup}..{nsteps - 1}",
                     props, ratio)
    # N = 1 default: ONE stream (BASELINE configs[1]).  The N > 1 lines belong to another curve ("p8": the archiver's task split,
    # the same eight tasks at every N); its N = 1 point is the `p8_on_one_gpu` object of this line / `--split 8`.
    line["curve"] = "single_stream"
    ds = CSCMIStats()
    for f, _ in CSCMIStats._fields_:
        setattr(ds, f, getattr(s1, f) - getattr(s0, f))
    bpl = in_b / max(1, launches)
    line["roofline"] = roofline(level, ratio, launches, kern_ms, in_b, {3: "k_encode_runs_dp4", 5: "k_encode_runs_bt", 1: "k_encode_runs_hp", 2: "k_encode_runs_hp"}.get(level, "k_encode_runs"),
                                "one stream = one workgroup: the libcsc chain is latency/issue-bound, not bandwidth-bound",
                                traffic_from_profile(f"m{level}_d{args.dict}_single_stream", bpl), ds)
    line["counters"] = {"find_match_calls": int(ds.find_match_calls), "slide_positions": int(ds.slide_positions),
                        "bt_steps": int(ds.bt_steps), "literals": int(ds.literals), "matches": int(ds.matches),
                        "analyze_kernel_ms": round(ds.analyze_kernel_ms, 3)}
    # the same stream once the window is full (chunks >= dict/chunk): what the 10^9-byte job runs at for most of its length
    if args.steady_steps > 0 and steady_from * chunk < nbytes:
        for i in range(nsteps, steady_from):
            step(i)
        sb, sdt, t0, t1 = region(steady_from, nchunks)
        sl, sk = t1.encode_launches - t0.encode_launches, t1.encode_kernel_ms - t0.encode_kernel_ms
        sr = (t1.output_bytes - t0.output_bytes) / max(1, t1.input_bytes - t0.input_bytes)
        line["steady"] = {"what": f"same stream, chunks {steady_from}..{nchunks - 1}: the {int(props.dict_size) >> 20} MiB window is full and wraps",
                          "value": round(sb / 1e6 / sdt, 4), "unit": "MB/s", "ratio": round(sr, 4),
                          "roofline_frac": round(alg_bytes(level, sr) * (t1.input_bytes - t0.input_bytes) / max(1e-9, sk * 1e-3) / 1e9 / HBM_PEAK_GBS, 8),
                          "avg_launch_ms": round(sk / max(1, sl), 3)}
        gpu_stream = bytes(writer.out)
    if not args.no_cpu_baseline:
        sample = max(nbytes, args.cpu_sample_mib << 20)
        cpu_in = src.read(0, min(task_size, sample)).tobytes()
        base, cpu_stream = cpu_baseline(cpu_in, level, dict_size, task_size)
        line["cpu_baseline"] = base
        line["bit_exact_vs_cpu_baseline"] = bool(cpu_stream[:len(gpu_stream)] == gpu_stream)
    else:
        line["cpu_baseline"] = None
    L.CSCEnc_Encode_Flush(h)
    L.CSCEnc_Destroy(h)
    return line


def run_split(args, R, lib, src, level, dict_size, split, steps, warmup, with_cpu=True):
    """the archiver's -p<split> tasks of the file, dealt over the ranks; one batch launch per step and rank"""
    torch = R.torch
    from csc_amd import corpus, tasks
    from csc_amd.capi import BytesWriter
    L = lib.lib
    slices = corpus.task_slices(src.size, split)
    mine = tasks.assign(slices, R.world)[R.rank]                     # dispatch order (largest first), i mod N
    chunk = 2 << 20
    nsteps = warmup + steps
    hs, ws, devs, sizes = [], [], [], []
    for tid in mine:
        off, n = slices[tid]
        props = lib.props_init(min(dict_size, n), level)
        chunk = int(props.raw_blocksize)
        w = BytesWriter()
        h = L.CSCEnc_Create(C.byref(props), C.cast(w.ptr(), C.c_void_p), None)
        if not h:
            raise SystemExit("CSCEnc_Create failed")
        w.out += lib.write_properties(props)
        nb = min(n, nsteps * chunk)
        hs.append(h); ws.append(w); sizes.append(nb)
        devs.append(torch.from_numpy(src.read(off, nb)).cuda())
    S = len(hs)
    H = (C.c_void_p * max(1, S))(*hs)
    torch.cuda.synchronize()

    def step(i):
        Z = [max(0, min(chunk, nb - i * chunk)) for nb in sizes]
        if S == 0 or not any(Z):
            return 0
        P = (C.c_void_p * S)(*[d.data_ptr() + i * chunk for d in devs])
        rc = L.CSCMI_EncodeDeviceChunkBatch(S, H, P, (C.c_size_t * S)(*Z))
        if rc != 0:
            raise SystemExit(f"batch encode failed rc={rc}")
        return sum(Z)

    def stats_sum():
        tot = CSCMIStats()
        for h in hs:
            s = CSCMIStats()
            L.CSCMI_GetStats(h, C.byref(s))
            for f, _ in CSCMIStats._fields_:
                setattr(tot, f, getattr(tot, f) + getattr(s, f))
        return tot

    for i in range(warmup):
        step(i)
    s0 = stats_sum()
    R.barrier()
    t0 = time.perf_counter()
    timed = 0
    for i in range(warmup, nsteps):
        timed += step(i)
    R.barrier()
    dt = time.perf_counter() - t0
    s1 = stats_sum()
    tmax = R.reduce(dt, "MAX")
    tot_bytes = R.reduce(timed, "SUM")
    in_b = R.reduce(s1.input_bytes - s0.input_bytes, "SUM")
    out_b = R.reduce(s1.output_bytes - s0.output_bytes, "SUM")
    # per-task check against what the REFERENCE wrote for the same task (tests/golden/split_prefix_digests.json, recorded by
    # tools/make_golden_prefix.py from oracle/_ref): the stream after c chunks is a prefix of the task's stream
    gold = golden("split_prefix_digests.json")
    recs = {}
    for tid, w, nb in zip(mine, ws, sizes):
        done = (min(nb, nsteps * chunk) + chunk - 1) // chunk
        s = bytes(w.out)
        rec = {"chunks": done, "stream_bytes": len(s), "sha256": hashlib.sha256(s).hexdigest(), "rank": R.rank}
        g = None
        if gold and src.synthetic and gold["level"] == level and gold["dict"] == dict_size:
            g = gold["splits"].get(str(len(slices)), {}).get(str(tid), {}).get(str(done))
        rec["bit_exact_vs_reference"] = None if g is None else bool(g["sha256"] == rec["sha256"] and g["stream_bytes"] == rec["stream_bytes"])
        recs[tid] = rec
    merged = {}
    for part in R.gather_objects(recs):
        merged.update(part)
    # N > 1, outside the timed region: the one exchange of the sharded archiver (csc_amd/sharded.py) -- every rank's
    # encoded streams so far go to rank 0 as a device tensor over RCCL, rank 0 checks the digests
    exchange = None
    if R.world > 1 and not args.no_exchange:
        try:
            from csc_amd import sharded
            blob = b"".join(bytes(w.out) for w in ws)
            meta = R.gather_objects((len(blob), hashlib.sha256(blob).hexdigest()))
            R.barrier()
            te = time.perf_counter()
            got = sharded.gather_blobs(blob, 0)
            R.barrier()
            te = time.perf_counter() - te
            if R.rank == 0:
                ok = all(len(got[r]) == meta[r][0] and hashlib.sha256(got[r]).hexdigest() == meta[r][1] for r in range(R.world))
                how = "grouped RCCL send/recv of device tensors over xGMI" if R.backend == "nccl" else f"{R.backend} send/recv of host tensors (test backend, ranks sharing a GPU)"
                exchange = {"what": f"encoded task streams of all ranks -> rank 0 (all_gather of lengths + {how})", "backend": R.backend,
                            "ok": bool(ok), "bytes": int(sum(m[0] for m in meta)), "seconds": round(te, 4)}
        except Exception as e:      # never lose the bench line to the hand-over
            exchange = {"ok": False, "error": repr(e)[:300]}
    line = None
    if R.rank == 0:
        ratio = out_b / max(1.0, in_b)
        props = lib.props_init(min(dict_size, slices[0][1]), level)
        line = base_line(args, R, src, level, tot_bytes / 1e6 / tmax, tmax, "strong",
                         f"{src.name} {src.size} B, -m{level} -d{args.dict} -p{split}: the same {len(slices)} tasks at every N (csarc.cpp:532-543), "
                         f"dealt largest first, task i -> rank i mod N (csarc.cpp:355); step = one {chunk}-byte chunk of every task of the rank, "
                         f"ONE launch per rank (one workgroup per stream), inputs resident in HBM; timed chunks {warmup}..{nsteps - 1} of every task",
                         props, ratio)
        line["steps"], line["warmup"] = steps, warmup
        line["curve"] = f"p{split}"            # which scaling curve this point belongs to: the same tasks at every N (strong scaling)
        line["ms_per_step"] = round(tmax * 1e3 / max(1, steps), 3)
        # launch count and HIP-event kernel time accrue on the batch's lead handle only, so the sums are the lead's
        nl = max(1, s1.encode_launches - s0.encode_launches)
        line["roofline"] = roofline(level, ratio, s1.encode_launches - s0.encode_launches, s1.encode_kernel_ms - s0.encode_kernel_ms,
                                    (s1.input_bytes - s0.input_bytes), # (which form a launch takes: csc_kernels_blocks.inc, launch_encode_runs_multi -- the pipeline form holds one stream a CU)
                                    ("k_encode_runs_multi_dp4" if S <= D4_MULTI_MAX else "k_encode_runs_multi_one (one parse wavefront a stream, four streams a CU)") if level == 3 else "k_encode_runs_multi*",
                                    f"rank 0's GPU: {S} streams per launch, one workgroup each; per-launch time from HIP events on the launch stream",
                                    traffic_from_profile(f"m{level}_d{args.dict}_p{split}", (s1.input_bytes - s0.input_bytes) / nl) if R.world == 1 else (None, "PMC passes are recorded for N = 1"))
        line["tasks_per_rank"] = [len(a) for a in tasks.assign(slices, R.world)]
        line["tasks"] = [dict(task=t, **merged[t]) for t in sorted(merged)]
        ex = [r["bit_exact_vs_reference"] for r in merged.values()]
        line["bit_exact_vs_reference"] = None if any(e is None for e in ex) else bool(all(ex))
        line["cpu_baseline"] = None
        if with_cpu and not args.no_cpu_baseline:
            # rank 0's host cores, after the timed region (the other ranks wait at the closing barrier): the reference's worker
            # threads as processes, min(8, tasks) of them, each over a bounded prefix of its task
            try:
                line["cpu_baseline"] = cpu_baseline_split(src, slices, level, dict_size, args.cpu_split_sample_mib << 20)
            except Exception as e:          # never lose the bench line to the CPU leg
                line["cpu_baseline"] = {"error": repr(e)[:300]}
        if exchange is not None:
            line["exchange"] = exchange
    if S:
        L.CSCMI_FlushBatch(S, H)
    for h in hs:
        L.CSCEnc_Destroy(h)
    return line


def other_config(args, R, lib, name):
    """The other single-GPU BASELINE.json configs under the same clock as the headline (extra `other_configs` object of the default
    N = 1 line, never `value`): one stream of the config's corpus at its level / dictionary, 1 warm-up + 2 timed chunks, the
    reference on one host core over a 16 MiB prefix, the GPU's stream so far checked as a prefix of the reference's."""
    import copy
    from csc_amd import corpus
    cname, lvl, dct = CONFIGS[name]
    a = copy.copy(args)
    a.config, a.dict, a.level, a.steps, a.warmup, a.steady_steps, a.cpu_sample_mib = name, dct, None, args.other_steps, 1, 0, 16
    line = run_single(a, R, lib, corpus.Source(cname), lvl, parse_size(dct))
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "ratio", "data", "roofline", "counters", "cpu_baseline", "bit_exact_vs_cpu_baseline")
    out = {k: line[k] for k in keep if k in line}
    out["what"] = line["config"]["workload"]
    out["config"] = {k: v for k, v in line["config"].items() if k != "workload"}
    return out


def cpu_baseline_tree(root, spec, level, dict_size, want_bytes=256 << 20):
    """the many-task line's CPU leg: the REFERENCE archiver (oracle/_ref/csarc_ref = archiver/*.cpp + libcsc/*.cpp, built by
    oracle/Makefile) with its own eight worker threads (`-t8`, csarc.cpp:200-201) over whole directories of the same tree until
    >= want_bytes of input.  A -t8 archive's block order depends on thread timing (csarc.cpp:361-398), so it is timed, not hashed."""
    import shutil
    import tempfile
    from csc_amd import treegen
    exe = os.path.join(ROOT, "oracle", "_ref", "csarc_ref")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/csarc_ref is not built on this box (oracle/Makefile needs /root/reference)"}
    by_dir = {}
    for rel, _kind, _seed, size in treegen.files(spec):
        by_dir.setdefault(os.path.dirname(rel), []).append(size)
    dirs, tot, nfiles = [], 0, 0
    for d in sorted(by_dir):
        dirs.append(d); tot += sum(by_dir[d]); nfiles += len(by_dir[d])
        if tot >= want_bytes:
            break
    hi = host_info()
    threads = min(8, hi["usable_cores"])
    out_dir = tempfile.mkdtemp(prefix="csc_cpu_tree_")
    try:
        arc = os.path.join(out_dir, "cpu.csa")
        t0 = time.perf_counter()
        p = subprocess.run([exe, "a", "-r", f"-m{level}", f"-d{dict_size >> 20}m", f"-t{threads}", arc] + dirs, cwd=root,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        if p.returncode != 0 or not os.path.exists(arc):
            return {"error": f"csarc_ref failed rc={p.returncode}: {p.stderr[-200:]}"}
        asize = os.path.getsize(arc)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    out = {"value": round(tot / 1e6 / dt, 3), "unit": "MB/s", "cores": threads, "kind": "reference",
           "sample": f"csarc_ref a -r -m{level} -d{dict_size >> 20}m -t{threads} over {len(dirs)} of the tree's {len(by_dir)} directories ({nfiles} files, {tot} bytes): "
                     f"the reference archiver itself (oracle/_ref, reference sources, g++ -O4) with its own worker threads, files from the page cache; "
                     f"timed, not hashed (-t{threads} block order depends on thread timing)",
           "seconds": round(dt, 2), "ratio": round(asize / max(1, tot), 4)}
    out.update(hi)
    return out


def run_tree(args, R, spec):
    """`--workload tree`: the many-task curve.  A seeded tree of thousands of files (csc_amd/treegen.py) goes through the archiver
    path end to end -- plan (csarc.cpp:490-557), one libcsc stream per extension group, container + index -- with the tasks dealt
    over the ranks (csc_amd/sharded.py: task i of the dispatch order -> rank i mod N, ONE exchange: the encoded tasks to rank 0).
    A step = one whole `csarc a -r -m3 -d64m` of the tree; the archive's SHA-256 is checked against what the REFERENCE archiver
    wrote for the same tree (tests/golden/tree_workload.json).  Inputs are files: host reads + H2D copies are inside the time."""
    import tempfile
    from csc_amd import sharded, treegen, csa
    root = os.environ.get("CSC_TREE_DIR") or os.path.join(tempfile.gettempdir(), f"csc_tree_{spec}_{os.getuid()}")
    if R.rank == 0:
        # the tree is benchmark INPUT: a private directory of this user, not a link somebody else could have planted, and stamped with
        # what generated it (a tree of an older generator is rebuilt instead of silently becoming the input)
        os.makedirs(root, mode=0o700, exist_ok=True)
        st = os.lstat(root)
        import stat as _stat
        if _stat.S_ISLNK(st.st_mode) or not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
            raise SystemExit(f"bench.py: {root} is not a directory owned by this user; set CSC_TREE_DIR")
        stamp = hashlib.sha256(repr((spec, treegen.SPECS[spec], treegen.files(spec)[:8], treegen.total_bytes(spec),
                                     open(os.path.join(ROOT, "csc_amd", "csrc", "corpus.c"), "rb").read())).encode()).hexdigest()
        marker = os.path.join(root, ".csc_tree_stamp")
        if not (os.path.isfile(marker) and open(marker).read().strip() == stamp):
            import shutil
            shutil.rmtree(os.path.join(root, "t"), ignore_errors=True)
        total = treegen.materialize(root, spec)
        with open(marker, "w") as f:
            f.write(stamp + "\n")
    R.barrier()
    total = treegen.total_bytes(spec)
    level, dict_size = 3, 64 << 20
    arc = os.path.join(root, "out.csa")
    cwd = os.getcwd()
    os.chdir(root)                                  # names in the archive are relative to the tree's root, as in the golden run
    try:
        def step():
            if R.rank == 0 and os.path.exists("out.csa"):
                os.unlink("out.csa")
            rc, st = sharded.add("out.csa", ["t"], level=level, dict_size=dict_size, recurse=True, overwrite=True)
            if rc != 0:
                raise SystemExit(f"sharded add failed rc={rc}")
            return st
        for _ in range(args.warmup):
            step()
        R.barrier()
        t0 = time.perf_counter()
        st = None
        for _ in range(args.steps):
            st = step()
        R.barrier()
        dt = time.perf_counter() - t0
    finally:
        os.chdir(cwd)
    tmax = R.reduce(dt, "MAX")
    line = None
    if R.rank == 0:
        h = hashlib.sha256()
        with open(arc, "rb") as f:
            for blk in iter(lambda: f.read(1 << 22), b""):
                h.update(blk)
        gold = (golden("tree_workload.json") or {}).get("trees", {}).get(spec)
        asize = os.path.getsize(arc)
        props = csc_props_for(level, dict_size)
        line = {
            "metric": f"archive-create MB/s (10^6 input bytes / wall-clock) on the seeded {spec} workload, csarc a -r -m{level} -d{dict_size >> 20}m; archive bit-exact vs reference (-t1 layout)",
            "value": round(total * args.steps / 1e6 / tmax, 4), "unit": "MB/s", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(tmax * 1e3 / max(1, args.steps), 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/u32 (byte + 32-bit integer work; 12-bit probabilities, 64-bit range-coder low)",
            "data": f"synthetic (csc_amd/treegen.py '{spec}': {len(treegen.files(spec))} files, {total} bytes, corpus kinds text/exe/delta/silesia)",
            "curve": spec,
            "config": {"workload": f"{spec}: {len(treegen.files(spec))} files in {st.get('n_tasks', '?')} tasks (one per 4-character extension, csarc.cpp:545-557), the same tasks at every N, "
                                   f"task i of the size-sorted dispatch order -> rank i mod N (csarc.cpp:355), every rank's tasks concurrently on its GPU (one workgroup per "
                                   f"stream), encoded tasks -> rank 0 over {'RCCL/xGMI' if R.backend == 'nccl' else R.backend}, rank 0 writes container + index; "
                                   f"step = one whole archive; files read from {root} (page cache), H2D inside the timed region",
                       "tasks": st.get("n_tasks"), "files": len(treegen.files(spec)), **props},
            "ratio": round(asize / max(1, total), 4), "archive_bytes": asize, "archive_sha256": h.hexdigest(),
            "bit_exact_vs_reference": None if not gold else bool(gold["sha256"] == h.hexdigest() and gold["archive_bytes"] == asize),
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes(level, asize / max(1, total)) * total * args.steps / tmax / 1e9, 4), "peak": HBM_PEAK_GBS * R.world,
                         "unit": "GB/s", "frac": round(alg_bytes(level, asize / max(1, total)) * total * args.steps / tmax / 1e9 / (HBM_PEAK_GBS * R.world), 8),
                         "traffic": (tr := traffic_from_profile(f"{spec}_m{level}_d{dict_size >> 20}m", total * args.steps))[0], "traffic_note": tr[1],
                         "note": "whole job incl. host I/O: algorithmic bytes of the LZ pass / wall-clock against N x 8 TB/s; kernels: k_encode_runs_multi* (one workgroup per task stream)"},
            "cpu_baseline": None,
        }
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_tree(root, spec, level, dict_size)
            except Exception as e:          # never lose the bench line to the CPU leg
                line["cpu_baseline"] = {"error": repr(e)[:300]}
        line["last_step_stats_rank0"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}
        if gold:
            line["cpu_reference_seconds_recorded"] = gold.get("reference_seconds")
    return line


def summary_of(line):
    """A compact object (<= 600 characters) emitted as the LAST key of the JSON line: a record that keeps only the tail of the line
    still shows the headline, the other single-GPU BASELINE configs (configs[2], the configs[4] task), the -p8 point and the
    many-stream legs.  Per entry: [MB/s, roofline frac, reference MB/s on the host cores (or None), bit-exact flag]."""
    def ent(o):
        if not isinstance(o, dict) or "value" not in o:
            return None
        cb = o.get("cpu_baseline") if isinstance(o.get("cpu_baseline"), dict) else {}
        ok = o.get("bit_exact_vs_cpu_baseline", o.get("bit_exact_vs_reference", o.get("bit_exact_vs_reference_digest")))
        fr = (o.get("roofline") or {}).get("frac", o.get("hbm_roofline_frac"))
        return [o["value"], fr, cb.get("value"), ok]
    out = {"fmt": "[MB/s, hbm_frac, cpu_ref_MB/s, bit_exact]", "headline": ent(line)}
    for k, v in (line.get("other_configs") or {}).items():
        lvl = "m5" if "_m5_" in k or k == "silesia" else "m2" if "_m2_" in k or k == "mix5" else k[:8]
        out[lvl] = ent(v) if ent(v) else str(v)[:60]
    if "p8_on_one_gpu" in line:
        out["p8"] = ent(line["p8_on_one_gpu"])
    ms = line.get("multi_stream")
    if isinstance(ms, list):
        for m in ms:
            out[f"p{m.get('streams')}"] = ent(m)
            if isinstance(m.get("decode"), dict):
                out[f"p{m.get('streams')}_dec"] = [m["decode"].get("value"), m["decode"].get("roundtrip_ok")]
    return out


def csc_props_for(level, dict_size):
    import csc_amd
    p = csc_amd.load().props_init(dict_size, level)
    return {"dict_size": int(p.dict_size), "hash_bits": int(p.hash_bits), "hash_width": int(p.hash_width), "good_len": int(p.good_len), "lz_mode": int(p.lz_mode), "bt_size": int(p.bt_size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="enwik9", choices=sorted(CONFIGS), help="which BASELINE.json workload (N = 1): enwik9 -m3 -d64m (headline), silesia -m5 -d256m, mix5 -m2 -d1024m")
    ap.add_argument("--level", type=int, default=None)
    ap.add_argument("--dict", default=None)
    ap.add_argument("--split", type=int, default=0, help="task split: 0 = one stream at N = 1 (curve 'single_stream'), -p8 at N > 1 (curve 'p8'); S > 0 = the -p<S> tasks of the "
                    "file at any N.  The N = 1 point of the p8 curve -- what a scaling sweep over --gpus 2/4/8 must be compared with -- is `--gpus 1 --split 8` "
                    "(also reported as `p8_on_one_gpu` in the default N = 1 line), NOT the default single-stream line")
    ap.add_argument("--steady-steps", type=int, default=3, help="N = 1 single stream: also time this many chunks after the window has filled (0 = skip)")
    ap.add_argument("--cpu-sample-mib", type=int, default=48)
    ap.add_argument("--cpu-split-sample-mib", type=int, default=24, help="-p8 line's CPU leg: bytes of each task every reference worker process encodes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange", action="store_true",
                    help="N > 1: skip the RCCL hand-over of the encoded streams to rank 0 (outside the timed region)")
    ap.add_argument("--multi-streams", default="127,954",
                    help="extra (N = 1, enwik9 only): the WHOLE file as -p<S> task splits, all tasks concurrently on this GPU; '' = skip")
    ap.add_argument("--workload", default="stream", choices=["stream", "tree", "tree_mid", "tree_small"],
                    help="stream: the libcsc stream workloads above (default).  tree: the many-task curve -- a seeded tree of 4096 files / 2048 extension "
                         "groups (2.1 GB) through the archiver path end to end, tasks dealt over the ranks (csc_amd/sharded.py); tree_small: 256 files (tests)")
    ap.add_argument("--other-steps", type=int, default=2, help="extra (N = 1, default config only): timed chunks of each of the other single-GPU BASELINE configs "
                    "(silesia -m5 -d256m, mix5 -m2 -d1024m) reported as `other_configs`; 0 = skip")
    ap.add_argument("--p8-steps", type=int, default=2, help="extra (N = 1, enwik9 only): the N > 1 workload (-p8) on this one GPU for this many steps; 0 = skip")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])          # does not return

    name, lvl, dct = CONFIGS[args.config]
    level = args.level if args.level is not None else lvl
    args.dict = args.dict or dct
    dict_size = parse_size(args.dict)
    R = Rank(args)
    import csc_amd
    from csc_amd import corpus
    lib = csc_amd.load()
    L = lib.lib
    L.CSCMI_EncodeDeviceChunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.CSCMI_EncodeDeviceChunk.restype = C.c_int
    L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.CSCMI_EncodeDeviceChunkBatch.restype = C.c_int
    L.CSCMI_GetStats.argtypes = [C.c_void_p, C.POINTER(CSCMIStats)]
    L.CSCMI_FlushBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.CSCMI_FlushBatch.restype = C.c_int
    src = corpus.Source(name)

    split = args.split if args.split > 0 else (8 if R.world > 1 else 0)
    if args.workload != "stream":
        line = run_tree(args, R, args.workload)
    elif split:
        line = run_split(args, R, lib, src, level, dict_size, split, args.steps, args.warmup)
    else:
        line = run_single(args, R, lib, src, level, dict_size)
        if args.config == "enwik9" and args.level is None:
            if args.p8_steps > 0:
                # the point N = 1 of the -p8 curve (what --gpus 2/4/8 run), as an extra field: `value` stays configs[1]
                p8 = run_split(args, R, lib, src, level, dict_size, 8, args.p8_steps, 1, with_cpu=False)
                line["p8_on_one_gpu"] = {k: p8[k] for k in ("value", "unit", "ratio", "ms_per_step", "steps", "warmup", "roofline", "bit_exact_vs_reference", "tasks_per_rank")}
                line["p8_on_one_gpu"]["what"] = p8["config"]["workload"]
                line["p8_on_one_gpu"]["curve"] = "p8"
                if not args.no_cpu_baseline:
                    try:
                        line["p8_on_one_gpu"]["cpu_baseline"] = cpu_baseline_split(src, corpus.task_slices(src.size, 8), level, dict_size, args.cpu_split_sample_mib << 20)
                    except Exception as e:          # never lose the bench line to the CPU leg
                        line["p8_on_one_gpu"]["cpu_baseline"] = {"error": repr(e)[:300]}
            if args.other_steps > 0:
                # BASELINE configs[2] and the configs[4] task on this GPU, same run, same clock
                line["other_configs"] = {}
                for oc in ("silesia", "mix5"):
                    try:
                        line["other_configs"][f"{CONFIGS[oc][0]}_m{CONFIGS[oc][1]}_d{CONFIGS[oc][2]}"] = other_config(args, R, lib, oc)
                    except (Exception, SystemExit) as e:          # never lose the headline to an extra (SystemExit included; Ctrl-C still ends the run)
                        line["other_configs"][oc] = {"error": repr(e)[:300]}
            if args.multi_streams:
                # extra field, not `value`: every task of the -p<S> split at once on this one GPU
                line.update(multi_stream_job(lib, src, [int(x) for x in args.multi_streams.split(",")], level, dict_size))
    if R.rank == 0:
        line.pop("summary", None)
        line["summary"] = summary_of(line)          # LAST in the line: what a reader of only the line's tail must see
        print(json.dumps(line), flush=True)
    R.done()


if __name__ == "__main__":
    main()
