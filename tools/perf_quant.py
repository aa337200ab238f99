"""Time the fp8 producer kernels at the bench shape (run under rocprofv3 --kernel-trace --stats)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, gen_qkv
from rectified_spaattn_amd import _core
wl = WORKLOADS["hunyuan_720p_128f"]
S = wl["S_vis"] + wl["text"]
spec = _core.LayoutSpec.hunyuan(S, wl["S_vis"] + wl["text_valid"])
q, k, v = gen_qkv(24, 0, S, wl["S_vis"], 128, torch.device("cuda:0"))
call = _core.StagedCall(q, k, v, spec, wl["top_k"], 0.0, None, qkv_fp8=True)
for _ in range(5):
    call.select(); call.quantize()   # K1 writing the images, then the stand-alone producer
torch.cuda.synchronize()
