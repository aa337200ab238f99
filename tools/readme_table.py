#!/usr/bin/env python3
"""Prints README.md's results table from profiles/r06_bench*.json (the final set of the round: tools/final_run_r5.sh), so that the
table and the committed bench lines cannot drift apart.  usage: python tools/readme_table.py"""
import json
import os

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def L(n):
    return json.load(open(os.path.join(P, n)))


def ms(x):
    return f"{x:.2f}"


def tf(x):
    return f"{x:,.0f}".replace(",", " ")


def pc(x):
    return f"{100 * x:.1f} %"


def main():
    b, f, p = L("r06_bench.json"), L("r06_bench_fp8.json"), L("r06_bench_pv.json")
    rb, rf, rp = b["roofline"], f["roofline"], p["roofline"]
    print("| workload (24 heads unless noted, D = 128) | K5 operands | layer | algorithmic | block-sparse kernel (fraction of the dense MFMA peak of its operands) |")
    print("|---|---|---|---|---|")
    print(f"| HunyuanVideo 720p, S = 115 456, 10.2 % kept (bench default, regime R2: random lists) | bf16 | {ms(b['ms_per_step'])} ms | {tf(b['value'])} TFLOP/s | "
          f"{ms(rb['k5_ms'])} ms = {tf(rb['achieved'])} TFLOP/s = **{pc(rb['frac'])}** at the power cap, {rb['traffic'] / 1e9:.0f} GB over the fabric |")
    print(f"| same | e4m3 (`--qkv-fp8`) | {ms(f['ms_per_step'])} ms | {tf(f['value'])} TFLOP/s | {ms(rf['k5_ms'])} ms = **{pc(rf['frac'])}** of the dense fp8 peak (round 4: 44.9 %); "
          f"{rf['traffic'] / 1e9:.0f} GB |")
    print(f"| same | **pv** (`--qkv-fp8 pv`: 2-byte Q·Kᵀ, e4m3 P·V; relative L1 0.040 against 0.117) | {ms(p['ms_per_step'])} ms | {tf(p['value'])} TFLOP/s | "
          f"{ms(rp['k5_ms'])} ms = {pc(rp['frac'])} of the mixed peak (3.33 PF) = {100 * (1 - rp['k5_ms'] / rb['k5_ms']):.0f} % under the 2-byte kernel; {rp['traffic'] / 1e9:.0f} GB |")
    names = {"script": "same, **the reference scripts' operating point** (top_k 180, p 0.3, Gilbert neighbours: 20.2 % kept, `regimes.script`)",
             "r1": "same, Gilbert neighbours + p = 0.05 (R1, 11.9 % kept)",
             "locality": "same, spatially smooth centroids (`locality`: 78 % list overlap between neighbouring blocks)"}
    for rg in ("script", "r1", "locality"):
        x, y, z = b["regimes"][rg], p["regimes"][rg], f["regimes"][rg]
        print(f"| {names[rg]} | bf16 / pv / e4m3 | {ms(x['ms_per_step'])} / {ms(y['ms_per_step'])} / {ms(z['ms_per_step'])} ms | {tf(x['value'])} / {tf(y['value'])} / {tf(z['value'])} TFLOP/s | "
              f"{pc(x['k5_frac'])} / {pc(y['k5_frac'])} / {pc(z['k5_frac'])}; {x['traffic'] / 1e9:.0f} GB at {100 * x['l2_hit_rate']:.0f} % L2 hits (bf16) |")
    wls = [("wan22", "Wan2.2-TI2V 720p 121f, S = 27 280, 24.8 % kept (BASELINE config 5)"), ("flux", "Flux 4096², S = 66 048, 10.6 % kept"),
           ("wan21", "Wan2.1-T2V 720p 81f, 40 heads, S = 75 600, 25 % kept"),
           ("cogvideox", "CogVideoX1.5 768p 81f, 48 heads, **D = 64**, S = 42 466, 25 % kept (`--workload cogvideox_768p_81f`)")]
    for k, name in wls:
        x, y, z = L(f"r06_bench_{k}.json"), L(f"r06_bench_{k}_pv.json"), L(f"r06_bench_{k}_fp8.json")
        print(f"| {name} | bf16 / pv / e4m3 | {ms(x['ms_per_step'])} / {ms(y['ms_per_step'])} / {ms(z['ms_per_step'])} ms | {tf(x['value'])} / {tf(y['value'])} / {tf(z['value'])} TFLOP/s | "
              f"{pc(x['roofline']['frac'])} / {pc(y['roofline']['frac'])} / {pc(z['roofline']['frac'])} |")
    print(f"| dense attention 16k × 16k (`box_ref`, the same kernel in dense mode) | bf16 | {ms(b['box_ref']['ms'])} ms | {tf(b['box_ref']['tflops'])} TFLOP/s | "
          f"**{pc(b['box_ref']['tflops'] / 2500)}** |")
    c5 = b["fp8"]["config5_wan22_ti2v"]
    print(f"\nconfig 5 (default line's sub-record): bf16 {c5['bf16']['ms_per_layer']:.2f} / pv {c5['pv']['ms_per_layer']:.2f} / e4m3 {c5['e4m3']['ms_per_layer']:.2f} ms per layer; "
          f"box_ref {b['box_ref']['ms']:.2f} ms; clock {rb['clock']['sclk_mhz']['mean']:.0f} MHz at {rb['clock']['power_w']['mean']:.0f} W; select pass {rb['select_pass_ms']:.2f} ms; "
          f"L2 hits {rb['traffic_note'].split('l2_hit_rate ')[1].split(';')[0]}")


if __name__ == "__main__":
    main()
