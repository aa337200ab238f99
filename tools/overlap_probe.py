#!/usr/bin/env python3
"""Do kernels of the mask-selection pass overlap when issued on two HIP streams?  K1 (HBM-bound) of one head group beside
K2..K4 (matrix / vector bound) of another, against the same work issued back to back on one stream."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import REGIMES, WORKLOADS, gen_inputs, make_neighbors, make_spec  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    wl = WORKLOADS["hunyuan_720p_128f"]
    spec = make_spec(wl)
    H = int(os.environ.get("RSA_PERF_H", "12"))
    calls = []
    for grp in range(2):
        q, k, v = gen_inputs(wl, H, grp * H, dev, REGIMES["r2"][0])
        c = _core.StagedCall(q, k, v, spec, wl["top_k"], 0.0, None)
        c.select()
        calls.append(c)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def serial():
        calls[0].select_pool()
        calls[1].select_rest()

    def overlapped():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            calls[0].select_pool()
        with torch.cuda.stream(s2):
            calls[1].select_rest()
        cur.wait_stream(s1); cur.wait_stream(s2)

    t_pool = timed(calls[0].select_pool)
    t_rest = timed(calls[1].select_rest)
    print(f"H={H}: K1 alone {t_pool:.3f} ms, K2..K4 alone {t_rest:.3f} ms, back to back {timed(serial):.3f} ms, "
          f"on two streams {timed(overlapped):.3f} ms")


if __name__ == "__main__":
    main()
