#!/usr/bin/env python3
"""L2 hits / misses of the K5 launches in rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum output directories (one line per launch).
usage: python3 tools/l2_hits.py <dir> [<dir> ...]"""
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "bsfwd" in r["Kernel_Name"]:
                key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
                acc[key] = acc.get(key, 0) + float(r["Counter_Value"])
        for disp, kern in sorted({(k[0], k[1]) for k in acc}):
            h, m = acc.get((disp, kern, "TCC_HIT_sum"), 0), acc.get((disp, kern, "TCC_MISS_sum"), 0)
            kind = "sparse" if h + m > 5e8 else "dense 16k"
            print(f"{d}: {kern} {kind} launch: L2 hits {h / 1e6:.1f} M, misses {m / 1e6:.1f} M, hit rate {h / (h + m + 1e-9):.3f}, "
                  f"fabric reads ~{m * 128 / 1e9:.1f} GB")
