#!/usr/bin/env python3
"""In-kernel s_memtime stamps of K5 (diagnostic instantiation, tuning k5_prio = 64): per workgroup the cycles spent
before the main loop (work mapping, row plan, Q fragments, kept list, first tiles), in it, and in the epilogue."""
import os
import sys

os.environ.setdefault("RSA_TUNING", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.perf_k5 import regime_call, timeit  # noqa: E402
from rectified_spaattn_amd import _lib  # noqa: E402

L = _lib.lib()
call, spec = regime_call(os.environ.get("RSA_PERF_REGIME", "r2"), 24, torch.device("cuda:0"))
call.select()
nwg = 24 * ((spec.NBv + 7) // 8 * 8) + 4096
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
p = buf.data_ptr()


def sgn(x):
    return x - (1 << 32) if x >= (1 << 31) else x


assert L.rsa_set_tuning(b"dbg_lo", sgn(p & 0xFFFFFFFF)) == 0 and L.rsa_set_tuning(b"dbg_hi", sgn(p >> 32)) == 0
assert L.rsa_set_tuning(b"k5_prio", 64) == 0
call.attend(); torch.cuda.synchronize()
buf.zero_()
call.attend(); torch.cuda.synchronize()
med, _ = timeit(call.attend, n=3, warm=0)
d = buf.view(nwg, 8).cpu()
d = d[d[:, 3] != 0]
T0, T1, T2, T3, n = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
sparse = d[:, 7] < spec.NBv
print(f"K5 (stamped build) {med:.3f} ms; workgroups stamped {len(d)} (sparse {int(sparse.sum())})")
for name, m in (("sparse", sparse), ("text", ~sparse)):
    if m.sum() == 0:
        continue
    pro, main, epi = (T1 - T0)[m].float(), (T2 - T1)[m].float(), (T3 - T2)[m].float()
    print(f"{name}: items mean {n[m].float().mean():.1f} | prologue {pro.mean():.0f} (p10 {pro.quantile(0.1):.0f}, p90 {pro.quantile(0.9):.0f}) "
          f"| main {main.mean():.0f} ({(main / n[m].float().clamp(min=1)).mean():.0f} per kept block) | epilogue {epi.mean():.0f} "
          f"(p10 {epi.quantile(0.1):.0f}, p90 {epi.quantile(0.9):.0f}) cycles")
# what the launch spends outside the stamped intervals: kernel cycles x resident slots - sum of the workgroups' stamped cycles
import subprocess  # noqa: E402,F401
tot = (T3 - T0).float().sum().item()
print(f"sum of stamped workgroup cycles {tot:.3e}; at 512 resident workgroups that is {tot / 512:.3e} cycles of the launch "
      f"({med:.3f} ms: the difference to the kernel's cycle count is dispatch gaps, ramp and tail)")
