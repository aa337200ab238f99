"""K5 at head_dim 64 (CogVideoX shape: 48 heads, S = 42 496): dense mode and the rectified operator."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectified_spaattn_amd import _core
from tools.perf_k5 import timeit
dev = torch.device("cuda:0")
H, S, D = 48, 42496, 64
q, k, v = (torch.randn(1, H, S, D, device=dev).to(torch.bfloat16) for _ in range(3))
med, _ = timeit(lambda: _core.dense_attention(q, k, v), n=3, warm=1)
print(f"dense D=64 H={H} S={S}: {med:.3f} ms {4.0*S*S*D*H/med/1e9:.0f} TFLOP/s")
spec = _core.LayoutSpec.cogvideo(S - 30, 226)  # 42 240 visual + 226 text (+30 pad rows not passed)
S2 = S - 30
q2, k2, v2 = q[:, :, :S2].contiguous(), k[:, :, :S2].contiguous(), v[:, :, :S2].contiguous()
call = _core.StagedCall(q2, k2, v2, spec, 82, 0.0, None)
call.select(); torch.cuda.synchronize()
pairs = call.bufs["counts"].sum().item()
med, _ = timeit(call.attend, n=3, warm=1)
fl = 4.0 * D * 128 * 128 * pairs + 4.0 * D * spec.q_text_valid * spec.kv_text_valid * H
print(f"sparse D=64 cogvideo layout: K5 {med:.3f} ms {fl/med/1e9:.0f} TFLOP/s (kept pairs {pairs})")
msel, _ = timeit(call.select, n=3, warm=1)
print(f"select pass {msel:.3f} ms")
