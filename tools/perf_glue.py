#!/usr/bin/env python3
"""Bandwidth of the producer kernels at HunyuanVideo 720p sizes, next to the PyTorch ops
they replace.  Run on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from rectified_spaattn_amd import _operator as op  # noqa: E402
from rectified_spaattn_amd import glue  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


@torch.no_grad()
def main():
    dev = "cuda:0"
    S, H, D = 115200, 24, 128
    x = torch.randn(1, S, H * D, device=dev).to(torch.bfloat16)
    order = torch.randperm(S, device=dev)
    ms = timeit(lambda: glue.permute_tokens(x, order))
    ms_t = timeit(lambda: x[:, order])
    gb = 2 * x.numel() * 2 / 1e9
    print(f"permute_tokens [1,{S},{H*D}] bf16: {ms:.3f} ms = {gb/ms*1e3:.0f} GB/s   (torch index: {ms_t:.3f} ms)")
    norm = helpers.RMS(D).to(dev, torch.bfloat16)
    cos, sin = (t.to(dev) for t in helpers.rope_tables(S, D))
    out = torch.empty(1, S, H, D, dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: glue.qk_norm_rope(x, H, glue.norm_params(norm), (cos, sin), S, out=out))

    def unfused():
        q = op.split_heads(x, H)
        return op.rotary(norm(q), (cos, sin))

    ms_t = timeit(unfused, n=3, warm=1)
    gb = (2 * x.numel() * 2 + 2 * S * D * 4) / 1e9
    print(f"qk_norm_rope   [1,{S},{H}x{D}] bf16: {ms:.3f} ms = {gb/ms*1e3:.0f} GB/s   (RMSNorm module + "
          f"apply_rotary_emb in torch: {ms_t:.3f} ms)")

    # the Wan producers (norm across heads + rotation + head split), Wan2.1-T2V-14B 720p 81f shape
    from rectified_spaattn_amd import rectified_wan21_attn as w21, rectified_wan22_attn as w22
    del x, out
    torch.cuda.empty_cache()
    S, H = 75600, 40
    x = torch.randn(1, S, H * D, device=dev).to(torch.bfloat16)
    norm = helpers.RMS(H * D).to(dev, torch.bfloat16)
    fr = helpers.wan_freqs(S, D).to(dev)
    cs = tuple(t.to(dev) for t in helpers.wan22_rope(S, D))
    for name, rot, unf in (
            ("complex128 (Wan2.1)", fr, lambda: w21._complex_rope(norm(x).unflatten(2, (H, -1)).transpose(1, 2), fr)),
            ("cos/sin fp32 (Wan2.2)", cs, lambda: w22._cos_sin_rope(norm(x).unflatten(2, (H, -1)), *cs).transpose(1, 2))):
        ms = timeit(lambda: glue.norm_rope_across_heads(x, H, glue.norm_params(norm), rot))
        ms_t = timeit(unf, n=3, warm=1)
        tb = S * D // 2 * 16 if torch.is_tensor(rot) else 2 * S * D * 4
        gb = (2 * x.numel() * 2 + tb) / 1e9
        print(f"norm_rope_across_heads [1,{S},{H}x{D}] bf16, {name}: {ms:.3f} ms = {gb/ms*1e3:.0f} GB/s   "
              f"(module calls in torch: {ms_t:.3f} ms)")


if __name__ == "__main__":
    main()
