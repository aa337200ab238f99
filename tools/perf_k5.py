#!/usr/bin/env python3
"""K5 micro-benchmarks on the GPU box: (a) block-sparse pass at the bench shape (mask from the selection
kernels), (b) the same kernel in dense mode at a few sequence lengths.  Prints TFLOP/s; used for A/B of kernel
variants in one process (env RSA_PERF_* select cases)."""
import os
import sys
import time

os.environ.setdefault("RSA_TUNING", "1")  # this tool flips kernel variants through rsa_set_tuning

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import REGIMES, WORKLOADS, gen_inputs, gen_qkv, make_neighbors, make_spec, regime_top_k  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def regime_call(regime, H, dev, fp8=False):
    """StagedCall of a bench workload (RSA_PERF_WORKLOAD, default the Hunyuan one) in one of bench.REGIMES (inputs generated on
    the device)."""
    wl = WORKLOADS[os.environ.get("RSA_PERF_WORKLOAD", "hunyuan_720p_128f")]
    spec = make_spec(wl)
    cent, nbr_kind, p = REGIMES[regime]
    q, k, v = gen_inputs(wl, H if H else wl["H"], 0, dev, cent, D=wl.get("D", 128))
    call = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, make_neighbors(wl, spec, nbr_kind), qkv_fp8=fp8)
    return call, spec


def call_flops(call, spec, H, D=128):
    pairs = call.bufs["counts"].sum().item()
    return 4.0 * D * 128 * 128 * pairs + 4.0 * D * spec.q_text_valid * spec.kv_text_valid * H, pairs


def main():
    dev = torch.device("cuda:0")
    what = sys.argv[1:] or ["sparse", "dense"]
    D = 128
    if "regimes" in what:  # K5 + select pass in every regime; kept-list overlap of adjacent query blocks
        H = int(os.environ.get("RSA_PERF_H", "24"))
        for regime in os.environ.get("RSA_PERF_REGIMES", "r2,r1,locality").split(","):
            call, spec = regime_call(regime, H, dev)
            call.select()
            torch.cuda.synchronize()
            flops, pairs = call_flops(call, spec, H)
            med, mn = timeit(call.attend, n=7, warm=2)
            msel, _ = timeit(call.select, n=5, warm=1)
            bm = _core.unpack_bitmask(call.bufs["bitmask"][0], spec.NB_total)[:, :spec.NBv]
            inter = (bm[:-1] & bm[1:]).sum(1).float()
            uni = (bm[:-1] | bm[1:]).sum(1).float()
            cnt = call.bufs["counts"].float()
            print(f"regime {regime}: K5 median {med:.3f} ms (min {mn:.3f}) {flops/med/1e9:.1f} TFLOP/s pairs={pairs} "
                  f"kept={pairs/(H*spec.NBv*spec.NB_total):.4f} counts min/mean/max {cnt.min().item():.0f}/"
                  f"{cnt.mean().item():.1f}/{cnt.max().item():.0f} | select {msel:.3f} ms | adjacent-list overlap "
                  f"(head 0) inter/union {(inter/uni).mean().item():.3f}, inter/own {(inter/bm[:-1].sum(1).float()).mean().item():.3f}",
                  flush=True)
            del call
            torch.cuda.empty_cache()
        return
    if "sparse" in what:
        H = int(os.environ.get("RSA_PERF_H", "24"))
        wl = WORKLOADS["hunyuan_720p_128f"]
        S = wl["S_vis"] + wl["text"]
        spec = _core.LayoutSpec.hunyuan(S, wl["S_vis"] + wl["text_valid"])
        q, k, v = gen_qkv(H, 0, S, wl["S_vis"], D, dev)
        call = _core.StagedCall(q, k, v, spec, wl["top_k"], 0.0, None)
        call.select()
        torch.cuda.synchronize()
        pairs = call.bufs["counts"].sum().item()
        flops = 4.0 * D * 128 * 128 * pairs + 4.0 * D * spec.q_text_valid * spec.kv_text_valid * H
        med, mn = timeit(call.attend)
        print(f"sparse hunyuan H={H}: K5 median {med:.3f} ms (min {mn:.3f})  {flops/med/1e9:.1f} TFLOP/s  pairs={pairs}")
        msel, _ = timeit(call.select)
        print(f"select pass: {msel:.3f} ms")
        del q, k, v, call
    if "pmcsel" in what:  # mask-selection pass only, for rocprofv3 --pmc passes over K1..K4
        call, spec = regime_call(os.environ.get("RSA_PERF_REGIME", "r2"), int(os.environ.get("RSA_PERF_H", "0")), dev)
        for _ in range(3):
            call.select()
        torch.cuda.synchronize()
        return
    if "pmc" in what:  # few launches, for rocprofv3 --pmc passes (env RSA_PERF_REGIME selects the regime)
        H = int(os.environ.get("RSA_PERF_H", "0"))   # 0 = the workload's head count
        fp8 = {"1": True, "pv": "pv"}.get(os.environ.get("RSA_PERF_FP8", "0"), False)
        call, spec = regime_call(os.environ.get("RSA_PERF_REGIME", "r2"), H, dev, fp8=fp8)
        if os.environ.get("RSA_PERF_NODENSE", "0") == "1":
            for _ in range(3):
                call.select()
                call.attend()
            torch.cuda.synchronize()
            return
        for _ in range(2):
            call.select()
            call.attend()
        if fp8:
            torch.cuda.synchronize()
            return
        Sd = 16384
        qd = torch.randn(1, 24, Sd, D, device=dev).to(torch.bfloat16)
        for _ in range(2):
            _core.dense_attention(qd, qd, qd)
        torch.cuda.synchronize()
        return
    if "fp8variants" in what:
        from rectified_spaattn_amd import _lib
        L = _lib.lib()
        H = 24
        wl = WORKLOADS["hunyuan_720p_128f"]
        S = wl["S_vis"] + wl["text"]
        spec = _core.LayoutSpec.hunyuan(S, wl["S_vis"] + wl["text_valid"])
        q, k, v = gen_qkv(H, 0, S, wl["S_vis"], D, dev)
        call = _core.StagedCall(q, k, v, spec, wl["top_k"], 0.0, None, qkv_fp8=True)
        call.select()
        pairs = call.bufs["counts"].sum().item()
        flops = 4.0 * D * 128 * 128 * pairs + 4.0 * D * spec.q_text_valid * spec.kv_text_valid * H
        ref = None
        for rnd in range(2):
            for opt in [int(x) for x in os.environ.get("RSA_PERF_OPTS", "0,1").split(",")]:
                assert L.rsa_set_tuning(b"fp8_variant", opt) == 0
                med, mn = timeit(call.attend, n=4, warm=1)
                o = call.out.float()
                if ref is None:
                    ref = o.clone()
                print(f"round {rnd} fp8 variant {opt}: {med:7.3f} ms {flops/med/1e9:7.1f} TF/s | max|d| {(o-ref).abs().max().item():.2e}", flush=True)
        L.rsa_set_tuning(b"fp8_variant", 0)
        return
    if "dense" in what:
        for S in (8192, 16384, 32768, 65536):
            H = 24
            q = torch.randn(1, H, S, D, device=dev).to(torch.bfloat16)
            k = torch.randn(1, H, S, D, device=dev).to(torch.bfloat16)
            v = torch.randn(1, H, S, D, device=dev).to(torch.bfloat16)
            med, mn = timeit(lambda: _core.dense_attention(q, k, v), n=3, warm=1)
            fl = 4.0 * S * S * D * H
            print(f"dense S={S} H={H}: median {med:.3f} ms  {fl/med/1e9:.1f} TFLOP/s")
            med8, _ = timeit(lambda: _core.dense_attention(q, k, v, qkv_fp8=True), n=3, warm=1)
            print(f"dense fp8 (incl. quantisation) S={S} H={H}: median {med8:.3f} ms  {fl/med8/1e9:.1f} TFLOP/s")
            medp, _ = timeit(lambda: _core.dense_attention(q, k, v, qkv_fp8="pv"), n=3, warm=1)
            print(f"dense pv form (incl. the V image pass) S={S} H={H}: median {medp:.3f} ms  {fl/medp/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
