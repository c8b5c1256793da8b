#!/usr/bin/env python3
"""Turn what tools/prof_cfg.sh left under gpurun_out/<tag>/ into the files committed under profiles/: <tag>_bench.json,
<tag>_kernel_stats.{csv,md}, <tag>_pmc.md (HBM traffic, L2 hits and instruction mix of the encode kernel per input byte) and the
configuration's key in profiles/pmc_traffic.json (what bench.py reports as roofline.traffic -- stamped with the hash of the library
it was measured on, so that bench.py can refuse it after the kernels change).
usage: prof_collect_cfg.py <tag> <traffic key, e.g. m5_d256m_single_stream> [what]"""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else key
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
for name in ("bench.json", "bench_under_rocprof.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
ks = os.path.join(src, "prof", "bench_kernel_stats.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), ks, os.path.join(dst, f"{tag}_kernel_stats.md"),
                    f"{tag}: bench.py ({what}) under rocprofv3 --kernel-trace --stats"], check=True, stdout=subprocess.DEVNULL)
line = json.load(open(os.path.join(src, "bench_under_rocprof.json")))
bpl = line["roofline"]["input_bytes_per_launch"]
files = glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)
tot, launches, kernels = {}, {}, set()
for f in files:
    for r in csv.DictReader(open(f)):
        if "encode_runs" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            launches[r["Counter_Name"]] = launches.get(r["Counter_Name"], 0) + 1
            kernels.add(r["Kernel_Name"].split("(")[0])
# every pass ran the same command: its input bytes are launches x bytes per launch of the bench line
inp = {k: v * bpl for k, v in launches.items()}
fetch = tot.get("FETCH_SIZE", 0) * 1024 / max(1, inp.get("FETCH_SIZE", 1))      # KiB units (MI355X_MICROARCH.md, HBM / rocprofv3 section)
write = tot.get("WRITE_SIZE", 0) * 1024 / max(1, inp.get("WRITE_SIZE", 1))
so = hashlib.sha256(open(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"), "rb").read()).hexdigest()[:16]
hit = 100 * tot.get('TCC_HIT_sum', 0) / max(1, tot.get('TCC_HIT_sum', 0) + tot.get('TCC_MISS_sum', 0))
per = lambda k: tot.get(k, 0) / max(1, inp.get(k, 1))
with open(os.path.join(dst, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: counters of `{', '.join(sorted(kernels))}` ({what})\n\n"
            "rocprofv3 --pmc <set> --kernel-trace, one pass per set, `python3 bench.py --config ... --steps N --warmup 0 --no-cpu-baseline ...` directly after `--` (tools/prof_cfg.sh).\n"
            "FETCH_SIZE / WRITE_SIZE are KiB; the gathers are 4-16 bytes wide, so no wide-read correction is applied.\n\n"
            "| counter | launches | sum over the launches | per input byte |\n|---|---|---|---|\n")
    for k in sorted(tot):
        f.write(f"| {k} | {launches[k]} | {tot[k]:.0f} | {per(k) * (1024 if k in ('FETCH_SIZE', 'WRITE_SIZE') else 1):.3f}{' B' if k in ('FETCH_SIZE', 'WRITE_SIZE') else ''} |\n")
    f.write(f"\nHBM traffic: {fetch:.1f} B read + {write:.1f} B written per input byte (algorithmic: {line['roofline']['alg_bytes_per_input_byte']} B/B).  "
            f"L2: {hit:.1f} % hits ({tot.get('TCC_HIT_sum', 0):.0f} / {tot.get('TCC_MISS_sum', 0):.0f}).\n"
            f"Wave-instructions per input byte (every wavefront of the stream's workgroup, polling loops included): "
            f"SALU {per('SQ_INSTS_SALU'):.0f}, VALU {per('SQ_INSTS_VALU'):.0f}, LDS {per('SQ_INSTS_LDS'):.0f}, branches {per('SQ_INSTS_BRANCH'):.0f}, "
            f"VMEM reads {per('SQ_INSTS_VMEM_RD'):.1f}, VMEM writes {per('SQ_INSTS_VMEM_WR'):.1f}.  "
            f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {tot.get('SQ_WAIT_INST_ANY', 0) / max(1, tot.get('SQ_WAVE_CYCLES', 1)):.3f}, "
            f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {tot.get('SQ_ACTIVE_INST_ANY', 0) / max(1, tot.get('SQ_WAVE_CYCLES', 1)):.3f}.\n"
            f"bench line of the FETCH_SIZE pass: {line['value']} MB/s, {line['roofline']['launches']} launches, {line['roofline']['avg_launch_ms']} ms per launch.\n"
            f"library sha256[:16] = {so}\n")
path = os.path.join(dst, "pmc_traffic.json")
allk = json.load(open(path)) if os.path.exists(path) else {}
allk[key] = {"fetch_bytes_per_input_byte": round(fetch, 2), "write_bytes_per_input_byte": round(write, 2), "library_sha256_16": so,
             "source": f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over bench.py ({what}); KiB units; "
                       "narrow 4-16 byte gathers, so no gfx950 wide-read doubling applied"}
json.dump(allk, open(path, "w"), indent=2)
print(open(os.path.join(dst, f"{tag}_pmc.md")).read())
