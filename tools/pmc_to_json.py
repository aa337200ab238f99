#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -> profiles/<name>.json with the gfx950 correction of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams -> x2;
WRITE_SIZE is exact; both are in KiB.  usage: pmc_to_json.py <out.json> <kernel-substring> <csv glob>..."""
import csv
import glob
import json
import sys
from collections import defaultdict

out, kern = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for pat in sys.argv[3:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha  # noqa: E402
res = {"kernel": kern, "kernel_source_sha": kernel_source_sha("fp8" in kern), "regime": os.environ.get("RSA_PERF_REGIME", "r2")}
for k, v in acc.items():
    res[k] = {"mean": sum(v) / len(v), "n": len(v)}
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    f = res["FETCH_SIZE"]["mean"] * 1024.0
    w = res["WRITE_SIZE"]["mean"] * 1024.0
    res["fetch_bytes_raw"] = f
    res["fetch_bytes_corrected_x2"] = 2.0 * f
    res["write_bytes"] = w
    res["traffic_bytes_per_launch"] = 2.0 * f + w
    res["note"] = ("L2 memory-side (fabric) bytes per launch; Infinity-Cache hits are included in these counters, so "
                   "this is an upper bound on HBM bytes")
if "TCC_HIT_sum" in acc:
    h, m = res["TCC_HIT_sum"]["mean"], res["TCC_MISS_sum"]["mean"]
    res["l2_hit_rate"] = h / (h + m)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
