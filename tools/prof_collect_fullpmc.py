#!/usr/bin/env python3
"""L2 hit rate of the encode kernel launch by launch over a FULL-SIZE single stream (the binary tree and the window grow far beyond the
256 MiB Infinity Cache on the way): reads the counter csv of
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/<tag> -o p -- python3 tools/gpu_fullsize_cfg.py <case> <dir>
and writes profiles/<tag>.md: hits / misses / hit rate / launch duration for the first launches, every tenth one, the last ones.
Usage: python3 tools/prof_collect_fullpmc.py gpurun_out/<tag> profiles/<tag>.md <kernel name substring>"""
import csv
import glob
import os
import sys

src, dst, kname = sys.argv[1], sys.argv[2], sys.argv[3]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
assert files, src
rows = {}
for f in files:
    for r in csv.DictReader(open(f)):
        if kname not in r["Kernel_Name"]:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
assert ids, kname
out = [f"# L2 (TCC) hits and misses of `{kname}`, launch by launch over a full-size single stream\n",
       "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace, the program directly after `--` (tools/prof_collect_fullpmc.py); "
       "one launch = one 2 MiB chunk of the stream.\n",
       "| launch | input so far (MiB) | ms | TCC hits | TCC misses | hit rate |", "|---|---|---|---|---|---|"]


def line(n, i):
    d = rows[i]
    h, m = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
    return f"| {n + 1} | {2 * (n + 1)} | {(d['t1'] - d['t0']) / 1e6:.0f} | {h:.3e} | {m:.3e} | {100.0 * h / max(h + m, 1.0):.1f} % |"


N = len(ids)
pick = sorted(set(list(range(0, min(4, N))) + list(range(9, N, 10)) + list(range(max(0, N - 3), N))))
for n in pick:
    out.append(line(n, ids[n]))
H = sum(rows[i].get("TCC_HIT_sum", 0.0) for i in ids)
M = sum(rows[i].get("TCC_MISS_sum", 0.0) for i in ids)
half = ids[N // 2:]
H2 = sum(rows[i].get("TCC_HIT_sum", 0.0) for i in half)
M2 = sum(rows[i].get("TCC_MISS_sum", 0.0) for i in half)
T = sum(rows[i]["t1"] - rows[i]["t0"] for i in ids) / 1e9
out.append(f"\n{N} launches, {T:.1f} s of kernel time; all launches: {100.0 * H / max(H + M, 1.0):.1f} % hits; "
           f"second half of the stream: {100.0 * H2 / max(H2 + M2, 1.0):.1f} % hits.\n")
open(dst, "w").write("\n".join(out))
print("\n".join(out[-12:]))
