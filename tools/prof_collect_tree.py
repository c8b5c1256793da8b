#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/prof_tree.sh) -> profiles/<tag>_bench.json, <tag>_bench_cold.json, <tag>_pmc.md and the `tree_m3_d64m` key of
profiles/pmc_traffic.json (what `bench.py --workload tree` reports as roofline.traffic, stamped with the library's hash).
usage: prof_collect_tree.py <tag>"""
import csv, glob, hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for name in ("bench.json", "bench_cold.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
line = json.load(open(os.path.join(src, "pmc_FETCH_SIZE.json")))
total = line["last_step_stats_rank0"]["raw_bytes"] * line["steps"]
tot, launches, kernels = {}, {}, set()
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "encode_runs" in r["Kernel_Name"] or "adler" in r["Kernel_Name"] or "analyze" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            launches[r["Counter_Name"]] = launches.get(r["Counter_Name"], 0) + 1
            kernels.add(r["Kernel_Name"].split("(")[0])
fetch = tot.get("FETCH_SIZE", 0) * 1024 / max(1, total)
write = tot.get("WRITE_SIZE", 0) * 1024 / max(1, total)
hit = 100 * tot.get("TCC_HIT_sum", 0) / max(1, tot.get("TCC_HIT_sum", 0) + tot.get("TCC_MISS_sum", 0))
so = hashlib.sha256(open(os.path.join(ROOT, "csc_amd", "libcsc_mi355x.so"), "rb").read()).hexdigest()[:16]
with open(os.path.join(dst, f"{tag}_pmc.md"), "w") as f:
    f.write(f"# {tag}: HBM traffic of `bench.py --workload tree --steps 1 --warmup 0` ({line['config']['files']} files, {line['config']['tasks']} tasks, {total} input bytes)\n\n"
            "rocprofv3 --pmc <set> --kernel-trace, one pass per set, the program directly after `--` (tools/prof_tree.sh); kernels counted: "
            f"{', '.join(sorted(kernels))}.  FETCH_SIZE / WRITE_SIZE are KiB; narrow gathers, no wide-read correction.\n\n"
            "| counter | launches | sum | per input byte |\n|---|---|---|---|\n")
    for k in sorted(tot):
        per = tot[k] / max(1, total)
        f.write(f"| {k} | {launches[k]} | {tot[k]:.0f} | {per * (1024 if k in ('FETCH_SIZE', 'WRITE_SIZE') else 1):.3f}{' B' if k in ('FETCH_SIZE', 'WRITE_SIZE') else ''} |\n")
    f.write(f"\nHBM traffic: {fetch:.1f} B read + {write:.1f} B written per input byte (algorithmic 42.2 + ratio).  L2: {hit:.1f} % hits.\n"
            f"bench line of the FETCH_SIZE pass: {line['value']} MB/s.\nlibrary sha256[:16] = {so}\n")
path = os.path.join(dst, "pmc_traffic.json")
allk = json.load(open(path)) if os.path.exists(path) else {}
allk["tree_m3_d64m"] = {"fetch_bytes_per_input_byte": round(fetch, 2), "write_bytes_per_input_byte": round(write, 2), "library_sha256_16": so,
                        "source": f"profiles/{tag}_pmc.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `python3 bench.py --workload tree --steps 1 --warmup 0`; KiB units"}
json.dump(allk, open(path, "w"), indent=2)
print(open(os.path.join(dst, f"{tag}_pmc.md")).read())
