#!/usr/bin/env python3
"""Occupancy summary of a rocprofv3 kernel trace (r_kernel_trace.csv) of a pipelined bench run: how much wall time has an
encoder-side kernel running, a decode-side kernel running, both, or nothing; per-queue phase lengths."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3      # ignore the first fraction of the trace (setup + warmup)
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "at::native" in n or "rocclr" in n:
        continue
    dec = any(k in n for k in ("decode_attn", "argmax", "greedy_update", "sched_", "set_ints", "beam_", "rms_prep", "Li128ELi64E", "128, 64"))
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), dec, r.get("Queue_Id", "0"), n))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + (t1 - t0) * skip
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, dec, q, n in ev:
    pts.append((s, 1, dec)); pts.append((e, -1, dec))
pts.sort()
cnt = [0, 0]
last = t0
acc = defaultdict(int)
for t, d, dec in pts:
    key = ("enc" if cnt[0] else "") + ("+dec" if cnt[1] else "")
    acc[key or "idle"] += t - last
    last = t
    cnt[1 if dec else 0] += d
tot = t1 - t0
print(f"window {tot / 1e6:.1f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:10s} {v / 1e6:9.2f} ms  {100 * v / tot:5.1f} %")
ksum = defaultdict(lambda: [0, 0])
for s, e, dec, q, n in ev:
    ksum["decode" if dec else "encode"][0] += e - s
    ksum["decode" if dec else "encode"][1] += 1
for k, (ns, c) in ksum.items():
    print(f"  sum of {k} kernel durations {ns / 1e6:9.2f} ms over {c} kernels (avg {ns / c / 1e3:.1f} us)")
# decode kernel gaps per queue
byq = defaultdict(list)
for s, e, dec, q, n in ev:
    byq[q].append((s, e, dec))
for q, lst in byq.items():
    gaps = [b[0] - a[1] for a, b in zip(lst, lst[1:]) if a[2] and b[2]]
    if gaps:
        gaps.sort()
        print(f"  queue {q}: {len(lst)} kernels, decode->decode gap median {gaps[len(gaps) // 2] / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f} us")
