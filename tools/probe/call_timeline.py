#!/usr/bin/env python3
"""Where the wall time of ONE call goes (rocprofv3 --kernel-trace CSV of tools/latency_profile.py): the last call of the trace (calls are
one stem_pool launch each, cut at the longest idle gap between two of them), split into encoders (up to the first rmsnorm = prefill), prefill (up to the first
rms_prep = decode) and decode; per phase the wall span, the union of kernel busy time (two streams may overlap) and the per-kernel sums.
usage: call_timeline.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", re.sub(r"oavsr::\(anonymous namespace\)::|\(anonymous namespace\)::|oavsr::|void ", "", r["Kernel_Name"]))[:44]) for r in rows]
# one stem_pool launch per call; the boundary between two calls is the longest idle gap between their stem launches (the host's synchronisation)
stems = [i for i, e in enumerate(ev) if "stem_pool" in e[2]]
cuts = []
for a, b in zip(stems, stems[1:]):
    best, arg, end = -1, a + 1, ev[a][1]
    for i in range(a + 1, b + 1):
        if ev[i][0] - end > best:
            best, arg = ev[i][0] - end, i
        end = max(end, ev[i][1])
    cuts.append(arg)
calls = [ev[a:b] for a, b in zip(cuts, cuts[1:] + [len(ev)])]
call = calls[-1]
t0, t1 = call[0][0], max(e[1] for e in call)
print(f"{len(calls)} calls in the trace; the last one: {len(call)} dispatches, {(t1 - t0) / 1e6:.2f} ms from first kernel start to last kernel end")


def first(pred):
    for e in call:
        if pred(e[2]):
            return e[0]
    return t1


# phases by name: prefill begins at the first copy_rows (prompt splice), decode at the first rms_prep
t_pref = first(lambda n: "copy_rows" in n)
t_dec = first(lambda n: "rms_prep" in n)
for name, a, b in (("encoders + projector", t0, t_pref), ("prompt + prefill", t_pref, t_dec), ("decode", t_dec, t1)):
    seg = [e for e in call if a <= e[0] < b]
    if not seg:
        continue
    iv = sorted((e[0], e[1]) for e in seg)
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    acc = defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        acc[n][0] += e - s
        acc[n][1] += 1
    print(f"\n== {name}: wall {(b - a) / 1e6:.2f} ms, {len(seg)} dispatches, chip busy (union) {busy / 1e6:.2f} ms, sum of kernel times {sum(e[1] - e[0] for e in seg) / 1e6:.2f} ms")
    for n, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:9]:
        print(f"   {n:46s} {c:5d} x {t / c / 1e3:7.2f} us = {t / 1e6:6.2f} ms")
