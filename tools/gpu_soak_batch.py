#!/usr/bin/env python3
"""Evidence run (gpurun -- python tools/gpu_soak_batch.py [seconds] [seed]): the BATCH entry points -- CSCMI_EncodeDeviceChunkBatch / CSCMI_FlushBatch / CSCMI_DecodeBatch,
i.e. the k_encode_runs_multi_* kernels and k_decode_run_multi that the -pN split and the archiver use -- on random batches: 2 .. 900 streams a round (beyond 768 level-3
streams the one-wavefront form runs), every stream its own corpus kind / offset / size (1 byte .. a few chunks), its own level 1..5 and dictionary size (mixed in one
batch call: one launch per kernel flavour), inputs resident in device memory, 2 MiB chunk rounds.  Every stream is compared byte for byte with the reference build's
(oracle/_ref/libcsc_ref.so, zeroing allocator; the oracle where that is absent), every batch-decoded stream with the reference DECODER's bytes.  Exit code 1 on the
first difference."""
import ctypes as C, hashlib, os, random, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib, BytesWriter, BytesReader, CSC_PROP_SIZE

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261005
rng = random.Random(seed)
prod = csc_amd.load()
L = prod.lib
L.CSCMI_EncodeDeviceChunkBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
L.CSCMI_EncodeDeviceChunkBatch.restype = C.c_int
L.CSCMI_FlushBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
L.CSCMI_FlushBatch.restype = C.c_int
L.CSCMI_DecodeBatch.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
L.CSCMI_DecodeBatch.restype = C.c_int
ref_path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
chk = CscLib(ref_path if os.path.exists(ref_path) else os.path.join(ROOT, "oracle", "liborc.so"))
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p; za = o.orc_zero_alloc()
print(f"checker: {'reference build (oracle/_ref)' if os.path.exists(ref_path) else 'oracle'}; seed {seed}; budget {budget:.0f} s", flush=True)
kinds = ["text", "exe", "delta", "random", "entropy8", "silesia", "mix5"]
CH = 2 << 20


def one_size(cap):
    r = rng.random()
    if r < 0.2:
        return max(1, min(cap, (1 << rng.randrange(0, 22)) + rng.randrange(-64, 65)))
    if r < 0.8:
        return rng.randrange(1, min(cap, 300_000) + 1)
    return rng.randrange(1, cap + 1)


t_start = time.time()
rounds = n_streams = n_bytes = n_ref_rt = 0
while time.time() - t_start < budget:
    r = rng.random()
    S = rng.randrange(2, 41) if r < 0.6 else rng.randrange(41, 400) if r < 0.85 else rng.randrange(769, 901)
    byte_cap = 48_000_000 if S <= 400 else 30_000_000                 # per round: the checker encodes all of it on the host
    same_level = rng.choice([None, None, 1, 2, 3, 5]) if S <= 400 else 3      # > 768 streams: all level 3 (the one-wavefront form's regime)
    cap = max(1, min(5 << 20, byte_cap // S * 2))
    specs, datas = [], []
    for k in range(S):
        kind = rng.choice(kinds); off = rng.randrange(0, 900_000_000); n = one_size(cap)
        level = same_level or rng.randrange(1, 6)
        dsz = rng.choice([32 << 10, 256 << 10, 1 << 20, 4 << 20, 64 << 20])
        if rng.random() < 0.25 and n > (64 << 10):
            dsz = max(32 << 10, min(dsz, n // rng.choice([2, 3])))
        if S > 40:
            dsz = max(32 << 10, min(dsz, n))                  # many streams: the dictionary clamped to the input like csa_worker.cpp:35 does (device memory: a handle owns window + tables)
        specs.append((kind, off, n, level, dsz))
        datas.append(corpus.fill(kind, corpus.SEED_ENWIK9, off, n).tobytes())
    total = sum(len(d) for d in datas)
    # ---- the HIP path: one batch call per chunk round, one flush for all ----
    hs, ws, devs = [], [], []
    for (kind, off, n, level, dsz), d in zip(specs, datas):
        p = prod.props_init(dsz, level)
        w = BytesWriter()
        h = L.CSCEnc_Create(C.byref(p), C.cast(w.ptr(), C.c_void_p), None)
        assert h
        w.out += prod.write_properties(p)
        hs.append(h); ws.append(w)
        devs.append(torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda())
    torch.cuda.synchronize()
    t0 = time.time()
    k = 0
    while True:
        Z = [max(0, min(CH, len(d) - k * CH)) for d in datas]
        live = [i for i in range(S) if Z[i] > 0]
        if not live:
            break
        H = (C.c_void_p * len(live))(*[hs[i] for i in live])
        P = (C.c_void_p * len(live))(*[devs[i].data_ptr() + k * CH for i in live])
        assert L.CSCMI_EncodeDeviceChunkBatch(len(live), H, P, (C.c_size_t * len(live))(*[Z[i] for i in live])) == 0
        k += 1
    assert L.CSCMI_FlushBatch(S, (C.c_void_p * S)(*hs)) == 0
    t_gpu = time.time() - t0
    for h in hs:
        L.CSCEnc_Destroy(h)
    got = [bytes(w.out) for w in ws]
    # ---- the checker, eight host threads (ctypes releases the GIL) ----
    def ref_one(i):
        kind, off, n, level, dsz = specs[i]
        rc, s = chk.encode(datas[i], props=chk.props_init(dsz, level), alloc=za)
        rcd, back = chk.decode(s, alloc=za)
        return rc, s, rcd, back
    first = ref_one(0)                                        # (alone: whatever the checker sets up on first use is set up by one thread)
    with ThreadPoolExecutor(8) as ex:
        want = [first] + list(ex.map(ref_one, range(1, S)))
    bad = [i for i in range(S) if want[i][0] != 0 or got[i] != want[i][1]]
    # ---- batch decode of the HIP streams, 256 handles a call ----
    dec_bad = []
    for a in range(0, S, 256):
        part = list(range(a, min(S, a + 256)))
        rs, dws, dhs = [], [], []
        for i in part:
            props = prod.read_properties(got[i][:CSC_PROP_SIZE])
            rr = BytesReader(got[i][CSC_PROP_SIZE:]); dw = BytesWriter()
            dh = L.CSCDec_Create(C.byref(props), C.cast(rr.ptr(), C.c_void_p), None)
            assert dh
            rs.append(rr); dws.append(dw); dhs.append(dh)
        R = (C.c_int * len(part))()
        rc = L.CSCMI_DecodeBatch(len(part), (C.c_void_p * len(part))(*dhs), (C.c_void_p * len(part))(*[C.cast(w.ptr(), C.c_void_p) for w in dws]), R)
        for dh in dhs:
            L.CSCDec_Destroy(dh)
        assert rc == 0
        for j, i in enumerate(part):
            if (R[j], bytes(dws[j].out)) != (want[i][2], want[i][3]):
                dec_bad.append(i)
            elif want[i][3] != datas[i]:
                n_ref_rt += 1
    rounds += 1; n_streams += S; n_bytes += total
    lv = sorted(set(sp[3] for sp in specs))
    print(f"round {rounds:3d}: {S:4d} streams, {total:10d} B, levels {lv}, {k} chunk rounds, HIP encode {total / 1e6 / t_gpu:8.2f} MB/s -> "
          f"{'all streams == reference, all batch decodes == reference decoder' if not bad and not dec_bad else 'DIFFERS: encode %s decode %s' % ([specs[i] for i in bad[:4]], [specs[i] for i in dec_bad[:4]])}", flush=True)
    if bad or dec_bad:
        print("FAILED", flush=True)
        sys.exit(1)
print(f"ALL OK: {rounds} rounds, {n_streams} streams ({n_ref_rt} of them streams the reference's own decoder does not turn back into the input: reproduced), {n_bytes} input bytes, "
      f"{time.time() - t_start:.0f} s; library sha256[:16] {hashlib.sha256(open(os.path.join(ROOT, 'csc_amd', 'libcsc_mi355x.so'), 'rb').read()).hexdigest()[:16]}")
