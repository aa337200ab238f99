#!/usr/bin/env python3
"""A/B of K5's split-KV for the text blocks (tuning key k5_tsplit) at several head counts, one process."""
import os
import sys

os.environ.setdefault("RSA_TUNING", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.perf_k5 import regime_call, call_flops, timeit  # noqa: E402
from rectified_spaattn_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
for H in (24, 12, 6, 3):
    call, spec = regime_call("r2", H, dev)
    call.select()
    torch.cuda.synchronize()
    flops, pairs = call_flops(call, spec, H)
    for rnd in range(2):
        for flag in (0, 1):
            assert L.rsa_set_tuning(b"k5_tsplit", flag) == 0
            med, mn = timeit(call.attend, n=7, warm=2)
            print(f"H={H} round {rnd} k5_tsplit={flag}: {med:7.3f} ms (min {mn:7.3f}) {flops/med/1e9:7.1f} TF/s", flush=True)
    L.rsa_set_tuning(b"k5_tsplit", 1)
    del call
    torch.cuda.empty_cache()
