#!/usr/bin/env python3
"""Time a whole-archive `csarc a` on the GPU box: product (CSA_Add) vs the reference archiver
(oracle/_ref/csarc_ref -t8) on the same file, and check the archives are identical.
  python tools/gpu_csa_probe.py [size_MB] [split] [level] [dict_MB]"""
import os, subprocess, sys, tempfile, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from csc_amd import corpus, csa
size = int(sys.argv[1]) * 10**6 if len(sys.argv) > 1 else 200 * 10**6
split = int(sys.argv[2]) if len(sys.argv) > 2 else 190
level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dmb = int(sys.argv[4]) if len(sys.argv) > 4 else 64
td = tempfile.mkdtemp(prefix="csa_probe_")
os.chdir(td)
with open("enwik.xml", "wb") as f:
    for off in range(0, size, 64 << 20):
        f.write(corpus.fill("text", corpus.SEED_ENWIK9, off, min(64 << 20, size - off)).tobytes())
os.utime("enwik.xml", (1700000000, 1700000000))
csa.lib()
for rep in range(2):
    t0 = time.time()
    rc, st = csa.add("out.csa", ["enwik.xml"], level=level, dict_size=dmb << 20, split_count=split, overwrite=True)
    dt = time.time() - t0
    print(f"product rep{rep}: rc={rc} {size/dt/1e6:.1f} MB/s total {dt:.2f}s  encode {st['seconds_encode']:.2f}s io {st['seconds_io']:.2f}s "
          f"setup {st['seconds_setup']:.2f}s tasks {st['n_tasks']} peak {st['peak_streams']} ratio {st['archive_bytes']/size:.4f}", flush=True)
mine = hashlib.sha256(open("out.csa", "rb").read()).hexdigest()
t0 = time.time(); rc, st = csa.test("out.csa", mt_count=int(os.environ.get("MT", "32"))); dt = time.time() - t0
print(f"product test: rc={rc} failures={st['verify_failures']} {size/dt/1e6:.1f} MB/s ({dt:.1f}s) {st}", flush=True)
ref = os.path.join(ROOT, "oracle", "_ref", "csarc_ref")
if os.path.exists(ref) and not os.environ.get("NOREF"):
    for t in (8,):
        t0 = time.time()
        subprocess.run([ref, "a", f"-m{level}", f"-d{dmb}m", f"-p{split}", f"-t{t}", "-f", "ref.csa", "enwik.xml"], check=True, capture_output=True)
        dt = time.time() - t0
        print(f"reference -t{t}: {size/dt/1e6:.1f} MB/s ({dt:.1f}s)", flush=True)
    t0 = time.time()
    subprocess.run([ref, "a", f"-m{level}", f"-d{dmb}m", f"-p{split}", "-t1", "-f", "out.csa", "enwik.xml"], check=True, capture_output=True)
    print(f"reference -t1: {size/(time.time()-t0)/1e6:.1f} MB/s; identical: {hashlib.sha256(open('out.csa','rb').read()).hexdigest() == mine}")
subprocess.run(["rm", "-rf", td])
