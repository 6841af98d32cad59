#!/usr/bin/env python3
"""Per (kernel, grid) average duration from a rocprofv3 --kernel-trace CSV (one kernel template serves several call sites of a decode step:
the grid tells them apart).  usage: trace_by_grid.py <dir or csv> [skip_first_n_dispatches]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

src = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
paths = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(list)
for p in paths:
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows[skip:]:
        n = re.sub(r"oavsr::\(anonymous namespace\)::|\(anonymous namespace\)::|oavsr::|void ", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:60]
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))
        acc[(n, g, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0))))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
print(f"{'kernel':60s} {'wgs':>6s} {'thr':>4s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'%':>6s}")
for (n, g, t), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:60s} {g:6d} {t:4d} {len(v):6d} {sum(v) / len(v) / 1e3:8.2f} {min(v) / 1e3:8.2f} {100 * sum(v) / tot:6.2f}")
print(f"total {tot / 1e6:.3f} ms over {sum(len(v) for v in acc.values())} dispatches")
