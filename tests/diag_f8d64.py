import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as orc
from rectified_spaattn_amd import _core, synth
DEV = "cuda:0"
for mk in (lambda: orc.layout_cogvideo(5 * 128 + 226, 226), lambda: orc.layout_wan(700, 1), lambda: orc.layout_hunyuan(4 * 128 + 256, 4 * 128 + 77)):
    lay = mk()
    D, H = 64, 2
    q, k, v = synth.structured_qkv(808, 1, H, lay.S, D, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(DEV, torch.bfloat16) for x in (q, k, v))
    spec = _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid, lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)
    call = _core.StagedCall(tq, tk, tv, spec, 2, 0.3, None, qkv_fp8=True)
    call.select_pool()
    fused = {n: t.clone() for n, t in call.fp8.items()}
    for t in call.fp8.values(): t.zero_()
    call.quantize()
    alone = call.fp8
    qf, kf, vf = (t.float().cpu().numpy() for t in (tq, tk, tv))
    want = orc.fp8_operands(qf, kf, vf, lay)
    for got, tag in ((fused, "fused"), (alone, "alone")):
        ex = _core.fp8_exps(got["scales"], H, lay.NB_total).cpu().numpy().astype(np.uint32)
        km = _core.fp8_kmean(got["scales"], H, lay.NB_total, D).cpu().numpy()
        print(lay.S, tag, "exps", np.array_equal(ex, want["exps"]), "kmean", np.array_equal(km, want["kmean"]),
              *[f"{n} {np.array_equal(got[n].cpu().numpy(), want[n])}" for n in ("q8", "k8", "v8t")])
