"""K5 at head_dim 64 (CogVideoX shape: 48 heads, S = 42 496): dense mode and the rectified operator.
RSA_PERF_AB=1: the A/B library (make ab) with its three loop forms (k5_form 0 = block as hipcc schedules it, 1 = its
hand-placed twin, 2 = the product's -m form), interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
AB = os.environ.get("RSA_PERF_AB", "0") == "1"
if os.environ.get("RSA_PERF_LIB") and not AB:
    from rectified_spaattn_amd import _lib as _l0
    _l0.LIB_PATH = os.path.abspath(os.environ["RSA_PERF_LIB"])
if AB:
    os.environ["RSA_TUNING"] = "1"
    from rectified_spaattn_amd import _lib
    _lib.LIB_PATH = os.environ.get("RSA_PERF_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rectified_spaattn_amd", "librsa_hip_ab.so")
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
if AB:
    L = _lib.lib()
    for rnd in range(2):
        for form in (0, 1, 2):
            assert L.rsa_set_tuning(b"k5_form", form) == 0
            ms, _ = timeit(call.attend, n=5, warm=1)
            md, _ = timeit(lambda: _core.dense_attention(q, k, v), n=3, warm=1)
            print(f"round {rnd} k5_form {form}: sparse {ms:.3f} ms {fl/ms/1e9:.0f} TFLOP/s | dense {md:.3f} ms {4.0*S*S*D*H/md/1e9:.0f} TFLOP/s", flush=True)
    L.rsa_set_tuning(b"k5_form", -1)
    # forms 0 and 1 share their arithmetic: bit-identical output at head dim 64 too
    outs = {}
    for form in (0, 1):
        L.rsa_set_tuning(b"k5_form", form)
        outs[form] = call.attend().clone()
    L.rsa_set_tuning(b"k5_form", -1)
    print("forms 0 and 1 bit-identical at head dim 64:", bool(torch.equal(outs[0], outs[1])))
if os.environ.get("RSA_PERF_FP8", "0") == "1":   # the e4m3 operand path at this shape (K1 writing the images + fp8 K5)
    call8 = _core.StagedCall(q2, k2, v2, spec, 82, 0.0, None, qkv_fp8=True)
    call8.select(); torch.cuda.synchronize()
    for rnd in range(3):
        m8, _ = timeit(call8.attend, n=7, warm=2)
        ms8, _ = timeit(call8.select, n=5, warm=1)
        print(f"round {rnd} fp8 sparse: K5 {m8:.3f} ms {fl/m8/1e9:.0f} TFLOP/s | select pass (with images) {ms8:.3f} ms", flush=True)
