#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle report (diagnostic; run on the GPU box).  Prints one line per stage per case
and never stops at the first mismatch, so one gpurun call localises a bug."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import OP_CASES, case_inputs, load_op_case  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from rectified_spaattn_amd import _core  # noqa: E402


def spec_from_oracle(lay):
    return _core.LayoutSpec(lay.S, lay.NB_total, lay.NBv, lay.n_txt, lay.kv_valid, lay.pool_valid,
                            lay.text_end_block, lay.ffb, lay.q_text_valid, lay.kv_text_valid)


def report(name, a, b, exact=False):
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        print(f"    {name:10s} SHAPE {a.shape} vs {b.shape}")
        return False
    if exact:
        neq = int((a != b).sum())
        print(f"    {name:10s} exact: {'OK' if neq == 0 else 'MISMATCH'}  differing={neq}/{a.size}"
              + ("" if neq == 0 else f"  max|d|={np.abs(a.astype(np.float64)-b.astype(np.float64)).max():.3e}"))
        return neq == 0
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    print(f"    {name:10s} max|d|={d.max():.3e} mean|d|={d.mean():.3e}")
    return True


def main():
    dev = torch.device("cuda:0")
    cases = sys.argv[1:] or OP_CASES
    for name in cases:
        meta, gold = load_op_case(name)
        q, k, v, lay, nbr = case_inputs(meta)
        print(f"== {name}: {meta}")
        B, H, S, D = q.shape
        q0, k0, v0 = q, k, v
        for dt in (torch.bfloat16, torch.float16):
            tq, tk, tv = (torch.from_numpy(x).to(dev, dt) for x in (q0, k0, v0))
            q, k, v = (x.float().cpu().numpy() for x in (tq, tk, tv))  # the values the kernels actually see
            o_ref, parts = orc.rectified_attention(q, k, v, lay, meta["top_k"], meta["p"], nbr, want_parts=True)
            tn = torch.from_numpy(nbr) if nbr is not None else None
            out, bufs = _core.rectified_attention(tq, tk, tv, spec_from_oracle(lay), meta["top_k"], meta["p"], tn,
                                                  return_parts=True)
            torch.cuda.synchronize()
            print(f"  dtype {dt}")
            g = {n: t.cpu().numpy() for n, t in bufs.items()}
            for bh in range(B * H):
                sel = parts[bh]
                st = sel["stats"]
                print(f"   bh={bh}")
                report("qbar", g["qbar"][bh], st.qbar, True)
                report("aq", g["aq"][bh], st.aq, True)
                report("kbar", g["kbar"][bh], st.kbar, True)
                report("ak", g["ak"][bh], st.ak, True)
                report("vbar", g["vbar"][bh], st.vbar, True)
                report("s_vis", g["scores"][bh][:, : lay.NBv], sel["s_vis"], True)
                if lay.n_txt:
                    report("s_txt", g["scores"][bh][:, lay.NBv:], sel["s_txt"], True)
                report("unrel", g["unrel"][bh], sel["unrel"], True)
                report("probs", g["probs"][bh], sel["probs"], True)
                kept = orc.unpack_bits(g["bitmask"][bh].view(np.uint32), lay.NB_total)
                report("kept", kept, sel["kept"], True)
                report("kept_gold", kept, gold["one_hot"][bh // H, bh % H], True)
                report("counts", g["counts"][bh], sel["kept"].sum(-1).astype(np.int32), True)
                report("R", g["R"][bh], sel["R"], True)
                report("w", g["w"][bh], sel["w"], True)
                report("comp", g["comp"][bh], sel["comp"])
            report("O/oracle", out.float().cpu().numpy(), o_ref)
            report("O/golden", out.float().cpu().numpy(), gold["out"])
            # dense kernel
            od = _core.dense_attention(tq, tk, tv).float().cpu().numpy()
            dref = np.stack([np.stack([orc.dense_attention(q[b, h], k[b, h], v[b, h]) for h in range(H)], 1)
                             for b in range(B)])
            report("dense", od, dref)


if __name__ == "__main__":
    main()
