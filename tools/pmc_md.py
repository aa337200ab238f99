#!/usr/bin/env python3
"""tools/pmc_summary.py output (one line per kernel x counter) -> markdown tables with the derived ratios.
usage: python3 tools/pmc_md.py <title> <summary.txt> [<summary.txt> ...] > profiles/rNN_pmc_summary.md"""
import re
import sys
from collections import OrderedDict

title, files = sys.argv[1], sys.argv[2:]
data = OrderedDict()
for f in files:
    for line in open(f):
        m = re.match(r"^(.*?)\s{2,}(\S+)\s+([0-9.]+)\s+\(n=(\d+)\)\s*$", line.rstrip("\n"))
        if not m:
            continue
        data.setdefault(m.group(1).strip(), OrderedDict())[m.group(2)] = float(m.group(3))

print(f"# {title}\n")
print("Collected with `tools/pmc_passes.sh` / `tools/pmc_select.sh` (separate `rocprofv3 --pmc` passes, counters only), aggregated with")
print("`tools/pmc_summary.py` (mean per launch), formatted by `tools/pmc_md.py`.  SQ_ACTIVE_*/SQ_WAIT_*/SQ_WAVE_CYCLES count quad-cycles;")
print("SQ_VALU_MFMA_BUSY_CYCLES counts cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x2 on")
print("gfx950 for wide streaming reads).  `grid=5750784` = the sparse K5 call (22 464 workgroups x 256 threads), `grid=786432` = the dense")
print("16k call of the same kernel.  Per-regime memory-side traffic of K5: `profiles/r04_k5_traffic_{r2,r1,locality}.json`.\n")
for k, c in data.items():
    print(f"## `{k}`\n")
    print("| counter | per launch |\n|---|---|")
    for name, v in sorted(c.items()):
        print(f"| {name} | {v:,.0f} |")
    d = []
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        d.append(f"kernel cycles (GRBM/8) {cyc:,.0f}")
        simd = cyc * 1024
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            d.append(f"MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:.1f} % of SIMD-cycles "
                     f"({c['SQ_INSTS_MFMA']:,.0f} MFMAs x {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_INSTS_MFMA']:.0f} cycles)")
            if "SQ_INSTS_VALU" in c:
                d.append(f"VALU instructions per MFMA {(c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA']:.1f}")
        if "SQ_ACTIVE_INST_VALU" in c:
            d.append(f"vector-issue busy (SQ_ACTIVE_INST_VALU x4 / SIMD-cycles) {100 * 4 * c['SQ_ACTIVE_INST_VALU'] / simd:.0f} %")
        if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
            d.append(f"wave lifetime {4 * c['SQ_WAVE_CYCLES'] / c['SQ_WAVES']:,.0f} cycles, {4 * c['SQ_WAVE_CYCLES'] / simd:.2f} waves per SIMD on average")
            d.append(f"wave time: issuing {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.0f} %, "
                     f"waiting {100 * c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.0f} %")
    if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
        d.append(f"L2 hit rate {100 * c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.0f} %")
    if "FETCH_SIZE" in c:
        d.append(f"memory-side traffic {(2 * c['FETCH_SIZE'] + c.get('WRITE_SIZE', 0)) * 1024 / 1e9:.2f} GB per launch")
    if d:
        print("\nDerived: " + "; ".join(d) + ".\n")
