#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/prof_dec.sh) -> profiles/<tag>_kernel_stats.{csv,md}, profiles/<tag>_pmc.md: the decode kernel's time, HBM traffic and
instruction mix per OUTPUT byte of one 4 MiB level-3 text stream.   usage: prof_collect_dec.py <tag>"""
import csv, glob, hashlib, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
ks = os.path.join(src, "prof", "dec_kernel_stats.csv")
shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), ks, os.path.join(dst, f"{tag}_kernel_stats.md"),
                f"{tag}: python3 tools/gpu_dec_one.py 3 4 text (one 4 MiB level-3 text stream through CSCDec_Decode) under rocprofv3 --kernel-trace --stats"], check=True, stdout=subprocess.DEVNULL)
out_bytes = 4 << 20
tot, launches = {}, {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_run" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            launches[r["Counter_Name"]] = launches.get(r["Counter_Name"], 0) + 1
so = hashlib.sha256(open(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"), "rb").read()).hexdigest()[:16]
per = lambda k: tot.get(k, 0) / out_bytes
plain = [l for l in open(os.path.join(src, "plain.txt")) if l.startswith("decode")]
with open(os.path.join(dst, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: counters of `k_decode_run` (one 4 MiB level-3 text stream, resumable kernel: a launch per batch of input blocks)\n\n"
            "rocprofv3 --pmc <set> --kernel-trace, one pass per set, `python3 tools/gpu_dec_one.py 3 4 text` directly after `--` (tools/prof_dec.sh).\n"
            "FETCH_SIZE / WRITE_SIZE are KiB.  Per OUTPUT byte (algorithmic: ratio + 1 B/B -- coded bytes in, decoded bytes out -- plus the 128 KiB literal table\n"
            "in and out of LDS once per launch).\n\n| counter | launches | sum | per output byte |\n|---|---|---|---|\n")
    for k in sorted(tot):
        f.write(f"| {k} | {launches[k]} | {tot[k]:.0f} | {per(k) * (1024 if k in ('FETCH_SIZE', 'WRITE_SIZE') else 1):.3f}{' B' if k in ('FETCH_SIZE', 'WRITE_SIZE') else ''} |\n")
    f.write(f"\nHBM traffic: {per('FETCH_SIZE') * 1024:.2f} B read + {per('WRITE_SIZE') * 1024:.2f} B written per output byte.  "
            f"L2: {100 * tot.get('TCC_HIT_sum', 0) / max(1, tot.get('TCC_HIT_sum', 0) + tot.get('TCC_MISS_sum', 0)):.1f} % hits.\n"
            f"Wave-instructions per output byte (ONE wavefront): SALU {per('SQ_INSTS_SALU'):.0f}, VALU {per('SQ_INSTS_VALU'):.0f}, LDS {per('SQ_INSTS_LDS'):.0f}, "
            f"branches {per('SQ_INSTS_BRANCH'):.0f}, VMEM reads {per('SQ_INSTS_VMEM_RD'):.1f}, VMEM writes {per('SQ_INSTS_VMEM_WR'):.1f}.  "
            f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {tot.get('SQ_WAIT_INST_ANY', 0) / max(1, tot.get('SQ_WAVE_CYCLES', 1)):.3f}, "
            f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {tot.get('SQ_ACTIVE_INST_ANY', 0) / max(1, tot.get('SQ_WAVE_CYCLES', 1)):.3f}.\n"
            f"{plain[0].strip() if plain else ''}\nlibrary sha256[:16] = {so}\n")
print(open(os.path.join(dst, f"{tag}_pmc.md")).read())
