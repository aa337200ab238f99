"""The C-ABI without Python in the loop: examples/c_host/rsa_host_demo (plain C, HIP runtime C API + include/rsa.h)
reads Q/K/V from a file, calls rsa_rectified_attention[_fp8] once and writes O.  Its output must equal the Python
binding's byte for byte (same library, same kernels) and sit within tolerance of the oracle."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "examples", "c_host", "rsa_host_demo")


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8"])
def test_c_host_matches_python_binding(tmp_path, fp8):
    from rectified_spaattn_amd import _core, synth
    if not os.path.exists(DEMO):
        pytest.fail(f"{DEMO} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    B, H, S, D, top_k, p, ffb = 1, 2, 5 * 128 - 19, 128, 2, 0.35, 1
    lay = orc.layout_wan(S, ffb)
    q, k, v = synth.structured_qkv(606, B, H, S, D, smooth=0.0)
    tq, tk, tv = (torch.from_numpy(x).to(torch.bfloat16) for x in (q, k, v))
    raw = torch.cat([t.contiguous().view(torch.int16).reshape(-1) for t in (tq, tk, tv)]).numpy()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    raw.tofile(fin)
    args = [DEMO, str(fin), str(fout), str(B), str(H), str(S), str(D), str(top_k), str(p), str(ffb)]
    if fp8:
        args.append("fp8")
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = torch.from_numpy(np.fromfile(fout, dtype=np.int16)).view(torch.bfloat16).reshape(B, S, H * D)
    spec = _core.LayoutSpec.wan(S, ffb)
    want = _core.rectified_attention(tq.cuda(), tk.cuda(), tv.cuda(), spec, top_k, p, None, qkv_fp8=fp8).cpu()
    assert torch.equal(got, want), "C host and Python binding disagree"
    qf, kf, vf = (t.float().numpy() for t in (tq, tk, tv))
    ref = (orc.rectified_attention_fp8 if fp8 else orc.rectified_attention)(qf, kf, vf, lay, top_k, p, None)
    err = np.abs(got.float().numpy() - ref)
    assert err.max() <= (4e-2 if fp8 else 2e-2) and err.mean() <= (4e-3 if fp8 else 2e-3)


def test_c_host_takes_the_tail_split_on_a_layout_without_text_rows(tmp_path):
    """rsa_carve_workspace hands the partial buffer out for EVERY layout since 0.5.0 (before: NULL when there are no text rows,
    so a C host never got the tail split on Wan layouts).  8 heads x 72 blocks = 576 workgroups: the second
    generation's 64 walks are split.  The C host's bytes equal the Python binding's (which allocates tpart itself) and differ
    from the unsplit kernel's only within rounding -- i.e. the C path DID split."""
    from rectified_spaattn_amd import _core, _lib
    if not os.path.exists(DEMO):
        pytest.fail(f"{DEMO} is not built (python -c 'import __graft_entry__ as g; g.build()')")
    B, H, S, D, top_k, p, ffb = 1, 8, 72 * 128, 128, 12, 0.05, 2
    g = torch.Generator().manual_seed(5)
    cent = torch.randn(H, S // 128, D, generator=g).repeat_interleave(128, 1)
    tq, tk = ((cent + 0.7 * torch.randn(H, S, D, generator=g)).to(torch.bfloat16)[None] for _ in range(2))
    tv = torch.randn(1, H, S, D, generator=g).to(torch.bfloat16)
    raw = torch.cat([t.contiguous().view(torch.int16).reshape(-1) for t in (tq, tk, tv)]).numpy()
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    raw.tofile(fin)
    r = subprocess.run([DEMO, str(fin), str(fout), str(B), str(H), str(S), str(D), str(top_k), str(p), str(ffb)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = torch.from_numpy(np.fromfile(fout, dtype=np.int16)).view(torch.bfloat16).reshape(B, S, H * D)
    spec = _core.LayoutSpec.wan(S, ffb)
    want = _core.rectified_attention(tq.cuda(), tk.cuda(), tv.cuda(), spec, top_k, p, None).cpu()
    assert torch.equal(got, want), "C host and Python binding disagree"
    L = _lib.lib()
    try:
        assert L.rsa_set_tuning(b"k5_tail_split", 0) == 0
        whole = _core.rectified_attention(tq.cuda(), tk.cuda(), tv.cuda(), spec, top_k, p, None).cpu()
    finally:
        L.rsa_set_tuning(b"k5_tail_split", 1)
    d = (got.float() - whole.float()).abs()
    assert 0 < float(d.max()) <= 2 * 2.0 ** -7 * max(1.0, float(whole.float().abs().max()))
