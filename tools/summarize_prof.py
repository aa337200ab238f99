#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short markdown table (library kernels by name, all other
kernels lumped) for profiles/.  usage: summarize_prof.py <kernel_stats.csv> <title> > profiles/xxx.md"""
import csv
import sys

OURS = ("bsfwd_kernel", "bsfwd64_kernel", "select_mask_kernel", "compensation_kernel", "pool_stats_kernel", "pooled_scores_kernel",
        "gapr_compare_kernel", "bsfwd_fp8_kernel", "fp8_blocks_kernel", "kmean_sample_kernel", "text_combine", "tail_combine", "dense_masked_kernel",
        "permute_tokens_kernel", "qk_norm_rope_kernel", "norm_rope_heads", "rel_l1_", "p2p_")
rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {title}\n")
print("Source: `rocprofv3 --kernel-trace --stats --output-format csv` (kernel_stats.csv), durations in microseconds.\n")
print("| kernel | calls | avg us | min us | max us | total ms | % of GPU time |")
print("|---|---|---|---|---|---|---|")
other = [0, 0.0]
for r in rows:
    name = r["Name"]
    if any(o in name for o in OURS):
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        print(f"| `{short[:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
              f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | "
              f"{100*float(r['TotalDurationNs'])/tot:.2f} |")
    else:
        other[0] += int(r["Calls"])
        other[1] += float(r["TotalDurationNs"])
print(f"| (PyTorch data-generation / copy kernels, outside the timed region) | {other[0]} | | | | "
      f"{other[1]/1e6:.2f} | {100*other[1]/tot:.2f} |")
