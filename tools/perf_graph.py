#!/usr/bin/env python3
"""One layer step (select pass + K5) launched stage by stage (what bench.py times) against the same step replayed from one captured
HIP graph, at RSA_PERF_H heads of the bench workload: what the launch gaps between the step's seven or eight kernels cost."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import REGIMES, WORKLOADS, gen_inputs, make_neighbors, make_spec, regime_top_k  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402
from perf_k5 import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = WORKLOADS[os.environ.get("RSA_PERF_WORKLOAD", "hunyuan_720p_128f")]
    for H in [int(x) for x in os.environ.get("RSA_PERF_HEADS", "24,12,6,3").split(",")]:
        regime = os.environ.get("RSA_PERF_REGIME", "r2")
        cent, nbk, p = REGIMES[regime]
        spec = make_spec(wl)
        q, k, v = gen_inputs(wl, H, 0, dev, cent, D=wl.get("D", 128))
        call = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, make_neighbors(wl, spec, nbk))

        def step():
            call.select()
            call.attend()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step(); step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ref = call.out.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(call.out, ref)
        res = []
        for rnd in range(3):
            e, _ = timeit(step, n=15, warm=3)
            r, _ = timeit(g.replay, n=15, warm=3)
            res.append((e, r))
        e = sorted(x[0] for x in res)[1]; r = sorted(x[1] for x in res)[1]
        print(f"H={H}: stage by stage {e:.3f} ms | one graph {r:.3f} ms | {100 * (1 - r / e):+.1f} % | output identical: {same}", flush=True)
        del call, q, k, v, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
