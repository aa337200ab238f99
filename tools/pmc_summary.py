#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean per dispatch per (kernel, counter)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if not any(k in name for k in ("bsfwd", "select_mask", "pool_stats", "pooled_scores", "compensation", "quant", "amax")):
                continue
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            if "bsfwd" in short:
                short += " grid=" + r.get("Grid_Size", "?")
            a = acc[(short, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {c:32s} {v/n:18.1f}  (n={n})")
