#!/usr/bin/env python3
"""A/B of K4's tile height (tuning key k4_rows: 16 / 32 query blocks per workgroup) and timing of the select pass."""
import os
import sys

os.environ.setdefault("RSA_TUNING", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.perf_k5 import regime_call, timeit  # noqa: E402
from rectified_spaattn_amd import _lib  # noqa: E402

L = _lib.lib()
call, spec = regime_call("r2", 24, torch.device("cuda:0"))
call.select()
ref = call.bufs["comp"].clone()
for rnd in range(2):
    for rows in (32, 16):
        assert L.rsa_set_tuning(b"k4_rows", rows) == 0
        med, mn = timeit(call.select, n=9, warm=2)
        same = torch.equal(call.bufs["comp"], ref)
        print(f"round {rnd} k4_rows={rows}: select pass {med:.4f} ms (min {mn:.4f}) comp identical {same}", flush=True)
