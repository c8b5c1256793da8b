#!/usr/bin/env python3
"""Evidence run (gpurun -- python tools/gpu_soak.py [seconds] [seed]): random inputs x random CSCProps through the HIP library, every stream compared byte for
byte with the reference build's (oracle/_ref/libcsc_ref.so, or the oracle where that is absent) and decoded back on the GPU -- the decoded bytes compared with the
reference DECODER's output for the same stream.  A case = a corpus kind (or a
splice of two), a random offset and size (1 byte .. 3 MiB, a third of them within 64 bytes of a power of two), a level 1..5, a dictionary size 32 KiB .. 64 MiB
(a third of them SMALLER than the input: window wrap), one time in five a custom hash geometry / good_len / parser mode, one time in four filters switched off.
Prints one line per case and a summary; exit code 1 on the first difference (the case's parameters are in its line: the run is reproducible by seed)."""
import ctypes as C, hashlib, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import csc_amd
from csc_amd import corpus
from csc_amd.capi import CscLib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
rng = random.Random(seed)
prod = csc_amd.load()
ref_path = os.path.join(ROOT, "oracle", "_ref", "libcsc_ref.so")
chk = CscLib(ref_path if os.path.exists(ref_path) else os.path.join(ROOT, "oracle", "liborc.so"))
o = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so")); o.orc_zero_alloc.restype = C.c_void_p; za = o.orc_zero_alloc()
print(f"checker: {'reference build (oracle/_ref)' if os.path.exists(ref_path) else 'oracle'}; seed {seed}; budget {budget:.0f} s", flush=True)
kinds = ["text", "exe", "delta", "random", "entropy8", "silesia", "mix5"]


def size():
    r = rng.random()
    if r < 0.33:
        p = 1 << rng.randrange(0, 22)
        return max(1, p + rng.randrange(-64, 65))
    if r < 0.66:
        return rng.randrange(1, 200_000)
    return rng.randrange(200_000, 3 << 20)


def props_of(L, n, desc):
    level = rng.randrange(1, 6)
    dsz = rng.choice([32 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 32 << 20, 64 << 20])
    if rng.random() < 0.33 and n > (64 << 10):
        dsz = max(32 << 10, min(dsz, n // rng.choice([2, 3, 5])))
    p = L.props_init(dsz, level)
    d = [f"m{level}", f"d{dsz}"]
    return p, d


t_start = time.time()
n_cases = n_bytes = n_ref_rt = 0
by_level = {}
while time.time() - t_start < budget:
    n = size()
    k1 = rng.choice(kinds)
    off = rng.randrange(0, 900_000_000)
    data = corpus.fill(k1, corpus.SEED_ENWIK9, off, n).tobytes()
    desc = [k1, f"off{off}", f"n{n}"]
    if rng.random() < 0.25 and n > 4096:
        k2 = rng.choice(kinds); cut = rng.randrange(1, n)
        data = data[:cut] + corpus.fill(k2, corpus.SEED_ENWIK9, off + 12345, n - cut).tobytes()
        desc += [f"splice {k2}@{cut}"]
    st = rng.getstate()
    outs = []
    for L in (prod, chk):
        rng.setstate(st)
        p, d = props_of(L, n, desc)
        custom = rng.random()
        if custom < 0.2 and p.bt_size == 0:            # hash-table levels: bucket width / table size / good_len; the lazy levels also the parser mode
            p.hash_width = rng.choice([1, 2, 3, 4, 5, 8, 9]); p.hash_bits = rng.choice([10, 12, 15, 16, 18, 20])
            p.good_len = rng.choice([8, 16, 24, 32, 48, 64, 65, 100, 200])
            if p.lz_mode != 3:
                p.lz_mode = rng.choice([1, 2])
            d += [f"w{p.hash_width} b{p.hash_bits} good{p.good_len} mode{p.lz_mode}"]
        elif custom < 0.2:                               # binary-tree level: tree size / cycles / good_len
            p.bt_size = rng.choice([40000, 100000, 300000, 1 << 20]); p.bt_cyc = rng.choice([4, 16, 32]); p.good_len = rng.choice([8, 16, 48, 200])
            d += [f"bt{p.bt_size} cyc{p.bt_cyc} good{p.good_len}"]
        if rng.random() < 0.25:
            p.DLTFilter = 0; p.TXTFilter = 0; p.EXEFilter = 0
            d += ["nofilters"]
        outs.append(L.encode(data, props=p, alloc=za) if L is chk else L.encode(data, props=p))
    level = d[0]
    (rc, s), (rc2, want) = outs
    # the decoder's parity is with the REFERENCE decoder on the same stream (which, rarely, does not give the input back -- tests/golden/ref_roundtrip_hazard.json --:
    # such cases are counted, and the device decoder must reproduce them byte for byte)
    rcd, back = prod.decode(s) if rc == 0 else (rc, b"")
    rcr, ref_back = chk.decode(want, alloc=za) if rc2 == 0 else (rc2, b"")
    ok = (rc, s) == (rc2, want) and (rcd, back) == (rcr, ref_back)
    if ok and back != data:
        n_ref_rt += 1
        desc = desc + ["[reference round trip != input, reproduced]"]
    n_cases += 1; n_bytes += n
    by_level[level] = by_level.get(level, 0) + 1
    print(f"{n_cases:4d} {' '.join(desc + d):90s} -> {len(s):8d} B sha {hashlib.sha256(s).hexdigest()[:10]} {'OK' if ok else 'DIFFERS rc=%d/%d dec=%d' % (rc, rc2, rcd)}", flush=True)
    if not ok:
        print("FAILED", flush=True)
        sys.exit(1)
print(f"ALL OK: {n_cases} cases ({n_ref_rt} of them streams the reference's own decoder does not turn back into the input: reproduced byte for byte), {n_bytes} input bytes, by level {dict(sorted(by_level.items()))}, {time.time() - t_start:.0f} s; library sha256[:16] {hashlib.sha256(open(os.path.join(ROOT, 'csc_amd', 'libcsc_mi355x.so'), 'rb').read()).hexdigest()[:16]}")
