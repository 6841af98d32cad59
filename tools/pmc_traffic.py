#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, MI355X_MICROARCH.md §HBM) into the HBM-side
traffic per launch of every kernel.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide (16 B / lane) streaming reads at 64 B, so the read side is doubled.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\s*\[clone.*", "", r["Kernel_Name"])
        acc[n][0] += float(r["Counter_Value"])
        acc[n][1] += 1
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(f, key=lambda k: -f[k][0]):
    fk, n = f[k]
    wk = w.get(k, [0.0, 1])[0]
    out[k] = dict(launches=n, fetch_kib_per_launch=round(fk / n, 1), write_kib_per_launch=round(wk / max(w.get(k, [0, 1])[1], 1), 1),
                  traffic_bytes_per_launch=round((2.0 * fk / n + wk / max(w.get(k, [0, 1])[1], 1)) * 1024))
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --streams 1`; "
                    "traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (gfx950 read-side correction)", kernels=out), open(sys.argv[3], "w"), indent=1)
for k in list(out)[:12]:
    print(k[:100], out[k])
