#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result (rocpd .db or *_kernel_stats.csv) into the small
text summary that is committed under profiles/.   usage: prof_summary.py <db-or-csv> <out.md> [title]"""
import csv, sqlite3, sys
src, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else src
rows = []
if src.endswith(".db"):
    c = sqlite3.connect(src)
    tot_all = c.execute("select sum(duration) from kernels").fetchone()[0] or 1
    for name, calls, total in c.execute("select name, count(*), sum(duration) from kernels group by name order by sum(duration) desc"):
        rows.append((name, int(calls), float(total), float(total) / calls, 100.0 * float(total) / tot_all))
    extra = list(c.execute("select name, count(*), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name"))
else:
    for r in csv.DictReader(open(src)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    extra = []
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats (durations in ns)\n\n")
    f.write("| kernel | calls | total ns | average ns | % |\n|---|---|---|---|---|\n")
    for name, calls, total, avg, pct in rows:
        f.write(f"| `{name}` | {calls} | {total:.0f} | {avg:.0f} | {pct:.4f} |\n")
    if extra:
        f.write("\n| kernel | dispatches | min ns | max ns | VGPR | SGPR | LDS B | grid.x | wg.x |\n|---|---|---|---|---|---|---|---|---|\n")
        for e in extra:
            f.write("| `%s` | %d | %d | %d | %s | %s | %s | %s | %s |\n" % e)
print(open(out).read())
