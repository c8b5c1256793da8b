#!/usr/bin/env python3
"""Turn what tools/prof_round.sh left under gpurun_out/<tag>/ into the files committed under profiles/:
  <tag>_bench.json, <tag>_bench_under_rocprof.json, <tag>_kernel_stats.{csv,md}, <tag>_pmc.md (HBM traffic + instruction mix of the
  encode kernel per input byte) and profiles/pmc_traffic.json (what bench.py reports as roofline.traffic -- stamped with the hash
  of the library it was measured on, so that bench.py can refuse it after the kernels change).
usage: prof_collect.py <tag>"""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for name in ("bench.json", "bench_under_rocprof.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
ks = os.path.join(src, "prof", "bench_kernel_stats.csv")
shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), ks, os.path.join(dst, f"{tag}_kernel_stats.md"),
                f"{tag}: python3 bench.py --steps 4 --warmup 1 (single stream, -m3 -d64m) under rocprofv3"], check=True, stdout=subprocess.DEVNULL)
tot, launches = {}, {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "encode_runs" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            launches[r["Counter_Name"]] = launches.get(r["Counter_Name"], 0) + 1
kernel = sorted({r["Kernel_Name"].split("(")[0] for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)
                 for r in csv.DictReader(open(f)) if "encode_runs" in r["Kernel_Name"]})
inp = {k: v * (2 << 20) for k, v in launches.items()}      # every launch is one 2 MiB chunk
fetch = tot.get("FETCH_SIZE", 0) * 1024 / max(1, inp.get("FETCH_SIZE", 1))      # KiB units (MI355X_MICROARCH.md, HBM / rocprofv3 section)
write = tot.get("WRITE_SIZE", 0) * 1024 / max(1, inp.get("WRITE_SIZE", 1))
so = hashlib.sha256(open(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"), "rb").read()).hexdigest()[:16]
with open(os.path.join(dst, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: counters of `{', '.join(kernel)}` (one stream, -m3 -d64m, 2 launches of 2 MiB per pass)\n\n"
            "rocprofv3 --pmc <set> --kernel-trace, one pass per set, `python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline ...` directly after `--`.\n"
            "FETCH_SIZE / WRITE_SIZE are KiB; the gathers are 4-16 bytes wide, so no wide-read correction is applied.\n\n"
            "| counter | sum over the launches | per input byte |\n|---|---|---|\n")
    for k in sorted(tot):
        per = tot[k] / max(1, inp[k])
        f.write(f"| {k} | {tot[k]:.0f} | {per * (1024 if k in ('FETCH_SIZE', 'WRITE_SIZE') else 1):.3f}{' B' if k in ('FETCH_SIZE', 'WRITE_SIZE') else ''} |\n")
    f.write(f"\nHBM traffic: {fetch:.1f} B read + {write:.1f} B written per input byte (algorithmic: 42.24 B/B).  "
            f"L2: {100 * tot.get('TCC_HIT_sum', 0) / max(1, tot.get('TCC_HIT_sum', 0) + tot.get('TCC_MISS_sum', 0)):.1f} % hits.\n"
            f"Wave-instructions per input byte (all nine working wavefronts of the stream's workgroup, their polling loops included): "
            f"SALU {tot.get('SQ_INSTS_SALU', 0) / max(1, inp.get('SQ_INSTS_SALU', 1)):.0f}, VALU {tot.get('SQ_INSTS_VALU', 0) / max(1, inp.get('SQ_INSTS_VALU', 1)):.0f}, "
            f"LDS {tot.get('SQ_INSTS_LDS', 0) / max(1, inp.get('SQ_INSTS_LDS', 1)):.0f}, branches {tot.get('SQ_INSTS_BRANCH', 0) / max(1, inp.get('SQ_INSTS_BRANCH', 1)):.0f}.\n"
            f"library sha256[:16] = {so}\n")
path = os.path.join(dst, "pmc_traffic.json")
allk = json.load(open(path)) if os.path.exists(path) else {}
allk["m3_d64m_single_stream"] = {"fetch_bytes_per_input_byte": round(fetch, 2), "write_bytes_per_input_byte": round(write, 2), "library_sha256_16": so,
                                 "source": f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `python3 bench.py --steps 2 --warmup 0 "
                                           "--no-cpu-baseline --multi-streams '' --p8-steps 0 --steady-steps 0`; KiB units; narrow 4-16 byte gathers, so no gfx950 wide-read doubling applied"}
json.dump(allk, open(path, "w"), indent=2)
print(open(os.path.join(dst, f"{tag}_pmc.md")).read())
