import sys, torch, time
sys.path.insert(0, ".")
from rectified_spaattn_amd.teacache import rel_l1_distance
a = torch.randn(1, 115456, 3072, device="cuda:0").to(torch.bfloat16)
b = (a.float() * 1.01).to(torch.bfloat16)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ours = t(lambda: rel_l1_distance(a, b))
ref = t(lambda: ((a - b).abs().mean() / b.abs().mean()).cpu().item())
gb = 2 * a.numel() * 2 / 1e9
print(f"rel_l1 [1,115456,3072] bf16: HIP one-pass {ours:.3f} ms ({gb/ours:.2f} TB/s incl. host sync) vs torch expression {ref:.3f} ms")
