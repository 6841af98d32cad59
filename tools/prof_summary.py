#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: per kernel calls / total ms / avg us, normalised per bench step."""
import csv
import re
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':90s} {'calls/step':>10s} {'ms/step':>9s} {'avg_us':>9s} {'%':>6s}")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r"oavsr::\(anonymous namespace\)::|\(anonymous namespace\)::|oavsr::", "", r["Name"])
    n = re.sub(r"_ZN5oavsr12_GLOBAL__N_1\d+", "", n)[:90]
    print(f"{n:90s} {float(r['Calls']) / steps:10.1f} {float(r['TotalDurationNs']) / 1e6 / steps:9.3f} {float(r['AverageNs']) / 1e3:9.2f} {100 * float(r['TotalDurationNs']) / tot:6.2f}")
print(f"total kernel time per step: {tot / 1e6 / steps:.2f} ms")
