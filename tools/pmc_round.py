#!/usr/bin/env python3
"""Aggregate the rocprofv3 counter passes of one round into per-kernel-family JSON files under profiles/.

  pmc_round.py <round-tag> <fetch_dir> <write_dir> <sq_dir> [<out_dir>]

Each dir holds one `rocprofv3 --pmc ... --kernel-trace --output-format csv` pass over the SAME short bench command
(`bench.py --streams 1 --steps 1 --warmup 1 --calls-per-step 2 --no-cpu-baseline --no-roofline`): FETCH_SIZE and WRITE_SIZE in separate
passes (MI355X_MICROARCH.md §HBM; KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 B, so reads are
doubled), and one SQ pass: SQ_INSTS_VALU_MFMA_MOPS_BF16 (512 flop units), SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE.
Per family: HBM-side traffic per launch, MFMA flops per launch from the counter, average duration from the pass's kernel trace, and
MFMA utilisation = counter flops / duration / 2.5 PFLOP/s."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

FAMILY = [  # rocprof kernel-name pattern -> bench.py family name
    (r"gemm_8pp_kernel", "gemm_8pp_256x256_bf16"),
    (r"gemm_8p_kernel", "gemm_8p_256x256_bf16"),
    (r"gemm_dma_kernelIDF16bLi0ELi256ELi256E|gemm_dma_kernel<.*256, 256, 2", "gemm_dma_256x256_bf16"),
    (r"gemm_dma_kernelIDF16bLi0ELi256ELi128E|gemm_dma_kernel<[^>]*0, 256, 128", "gemm_dma_256x128_bf16"),
    (r"gemm_dma_kernelIDF16bLi0ELi128ELi64E|gemm_dma_kernel<[^>]*0, 128, 64", "decode_gemm_dma_128x64_bf16"),        # (tools/decode_mix.py passes: split-K launches, reduce inside)
    (r"gemm_dma_kernelIDF16bLi0ELi128ELi256E|gemm_dma_kernel<[^>]*0, 128, 256", "decode_gemm_dma_128x256_bf16"),     # lm_head
    (r"decode_strip_pair_kernel", "decode_gemm_strip16_bf16"),                                                        # ... two workgroups per strip (few long strips)
    (r"decode_strip_kernel", "decode_gemm_strip16_bf16"),                                                             # <= 16 rows: the weight-streaming strips (decode_strip.hip)
    (r"conv3x3_c128_kernel", "conv3x3_c128_bf16"),
    (r"conv3x3_c64_kernel", "conv3x3_c64_bf16"),
    (r"decode_attn_beam4_kernel", "decode_attention_beam4"),
    (r"splitk_reduce_kernel", "decode_splitk_reduce"),
    (r"decode_attn_kernel", "decode_attention"),
    (r"attn_dma_kernel", "attention_bf16"),
]


def fam(name):
    for pat, f in FAMILY:
        if re.search(pat, name):
            return f
    return None


def counters(d):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            f = fam(r["Kernel_Name"])
            if f:
                acc[f][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[f][r["Counter_Name"]] += 1
    return acc, cnt


def durations(d):
    tot, n, names = defaultdict(float), defaultdict(int), {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            f = fam(r["Kernel_Name"])
            if f:
                tot[f] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                n[f] += 1
                names[f] = r["Kernel_Name"]
    return {f: tot[f] / n[f] for f in tot}, names


tag, dfetch, dwrite, dsq = sys.argv[1:5]
out_dir = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
(fa, fc), (wa, wc), (sa, sc) = counters(dfetch), counters(dwrite), counters(dsq)
dur, names = durations(dsq)
for f in sorted(set(fa) | set(sa)):
    nl = max(fc[f].get("FETCH_SIZE", 0), 1)
    fetch = fa[f].get("FETCH_SIZE", 0.0) / nl
    write = wa[f].get("WRITE_SIZE", 0.0) / max(wc[f].get("WRITE_SIZE", 0), 1)
    ns = max(sc[f].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0), 1)
    mops = sa[f].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / ns
    what = "bench.py --gpus 1 --steps 1 --warmup 1 --calls-per-step 2 --no-cpu-baseline --no-roofline --no-grouped (the JOB's own launches: scheduler on, engine-owned caller streams)" if "job" in tag else \
        "tools/decode_mix.py 1 1b (one row: the weight-streaming strips)" if "decode_r1" in tag else "tools/decode_mix.py 8 1b (8 rows: the weight-streaming strips; lm_head on the 128 x 256 tile)" if "decode_r8" in tag else "tools/decode_mix.py 97 8b (97 rows, Llama-3.1-8B widths)" if "decode8b" in tag else "tools/decode_mix.py 97 1b (97 rows, Llama-3.2-1B widths)" if "decode" in tag else "tools/gemm_mix.py" if "gemm" in tag else \
        "bench.py --streams 1 --steps 1 --warmup 1 --calls-per-step 2 --no-cpu-baseline --no-roofline"
    d = dict(note="rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*; separate runs, each with --kernel-trace) over `" + what + "`; traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (gfx950 read-side "
                  "correction; the counters sit on the L2's memory side, Infinity-Cache hits included); mfma flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512; "
                  "mfma_util = mfma flops / avg duration / 2.5 PFLOP/s (profiled passes run ~3 % below the un-profiled clock)",
             kernel=f, rocprof_name=names.get(f), launches=nl, fetch_kib_per_launch=round(fetch, 1), write_kib_per_launch=round(write, 1),
             traffic_bytes_per_launch=round((2.0 * fetch + write) * 1024), avg_duration_us=round(dur.get(f, 0.0) / 1e3, 2),
             mfma_mops_bf16_per_launch=round(mops, 1), mfma_flops_per_launch=mops * 512,
             sq_valu_mfma_busy_cycles_per_launch=round(sa[f].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(sc[f].get("SQ_VALU_MFMA_BUSY_CYCLES", 0), 1), 1),
             sq_busy_cycles_per_launch=round(sa[f].get("SQ_BUSY_CYCLES", 0.0) / max(sc[f].get("SQ_BUSY_CYCLES", 0), 1), 1),
             grbm_gui_active_per_launch=round(sa[f].get("GRBM_GUI_ACTIVE", 0.0) / max(sc[f].get("GRBM_GUI_ACTIVE", 0), 1), 1))
    if dur.get(f):
        d["mfma_util"] = round(mops * 512 / (dur[f] * 1e-9) / 2.5e15, 4)
        d["hbm_side_gbs"] = round(d["traffic_bytes_per_launch"] / (dur[f] * 1e-9) / 1e9, 1)
    json.dump(d, open(os.path.join(out_dir, f"{tag}_pmc_traffic_{f}.json"), "w"), indent=1)
    print(f, {k: d[k] for k in ("launches", "traffic_bytes_per_launch", "avg_duration_us", "mfma_util", "hbm_side_gbs") if k in d})
