#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU (build container only).

    python tests/golden/make_golden.py            # needs /root/reference; never run on the GPU box

Nothing of the reference is copied: it is imported from /root/reference, executed on small deterministic
inputs (rectified_spaattn_amd.synth -- counter-based, so tests regenerate identical inputs), and only the
numeric outputs are stored.  Oracle-side shims needed to run it without a GPU / diffusers / flash-attn:

  * stub modules `diffusers...` (names used only as type hints by the reference)
  * TRITON_INTERPRET=1  -> the reference's Triton kernel runs on CPU (fp16; the interpreter has no bf16)
  * torch.cuda.device -> nullcontext (the wrapper enters `with torch.cuda.device(q.device)`)
  * flash_attn_varlen_func -> an SDPA-based varlen equivalent (exact softmax attention per segment)
  * the module's `_triton_block_sparse_attention_onehot` is wrapped to cast q/k/v to fp16 on the way in, so
    the whole operator can run with fp32 statistics (the reference code is dtype-generic) -- this is the
    "fp32 statistics" regime the numeric contract (oracle/rsa_oracle.c) pins.

Every case is also checked here against the oracle: masks must agree bit-for-bit, otherwise the seed is
rejected (a near-tie between the reference's reduction order and the contract's) and the next one is tried.
"""
import contextlib
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FIRST_SEED = 20251212   # the op_* cases try FIRST_SEED, FIRST_SEED + 1, ...: a fixture whose meta says seed == FIRST_SEED had no rejected seed
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np
import torch
import torch.nn.functional as F


def _install_stubs():
    class _Dummy:
        pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("diffusers")
    mod("diffusers.models")
    mod("diffusers.models.attention_processor", Attention=_Dummy, AttentionProcessor=_Dummy)
    mod("diffusers.models.transformers")
    mod("diffusers.models.transformers.transformer_wan", _get_qkv_projections=None,
        _get_added_kv_projections=None)

    def apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1):
        cos, sin = freqs_cis
        cos, sin = cos[None, None], sin[None, None]
        x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
        x_rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
        return (x.float() * cos + x_rot.float() * sin).to(x.dtype)

    mod("diffusers.models.embeddings", apply_rotary_emb=apply_rotary_emb)
    sys.path.insert(0, REF)
    # this repository ships an alias package of the same name (a regular package beats the reference's namespace
    # package wherever it sits on sys.path): pin `rectified_spaattn` to the REFERENCE's directory for this script
    ref_pkg = types.ModuleType("rectified_spaattn")
    ref_pkg.__path__ = [os.path.join(REF, "rectified_spaattn")]
    sys.modules["rectified_spaattn"] = ref_pkg
    torch.cuda.device = lambda d: contextlib.nullcontext()


def _varlen_sdpa(q, k, v, cu_q, cu_kv, max_q, max_kv):
    """q [(B*Sq), H, D]; segments s: q rows [cu_q[s],cu_q[s+1]) attend kv rows [cu_kv[s],cu_kv[s+1])."""
    out = torch.zeros_like(q)
    cu_q = [int(x) for x in cu_q]
    cu_kv = [int(x) for x in cu_kv]
    for s in range(len(cu_q) - 1):
        a, b = cu_q[s], cu_q[s + 1]
        c, d = cu_kv[s], cu_kv[s + 1]
        if b <= a or d <= c:
            continue
        o = F.scaled_dot_product_attention(q[a:b].transpose(0, 1).float(), k[c:d].transpose(0, 1).float(),
                                           v[c:d].transpose(0, 1).float())
        out[a:b] = o.transpose(0, 1).to(q.dtype)
    return out


def _wrap_kernel(module):
    orig = module._triton_block_sparse_attention_onehot

    def wrapped(q, k, v, seqlens, block_mask, sm_scale, bm=128, bn=128):
        o = orig(q.half(), k.half(), v.half(), seqlens, block_mask, sm_scale, bm, bn)
        return o.float()

    module._triton_block_sparse_attention_onehot = wrapped


def main():
    _install_stubs()
    import rectified_spaattn.attn as ref_attn
    import rectified_spaattn.gapr_mask as ref_gapr
    import rectified_spaattn.rectified_hunyuan_attn as ref_hy
    import rectified_spaattn.rectified_flux_attn as ref_fx
    import rectified_spaattn.rectified_wan21_attn as ref_wan
    import rectified_spaattn.rectified_cogvideo_attn as ref_cog
    from oracle import oracle as orc
    from rectified_spaattn_amd import synth

    ref_attn.flash_attn_varlen_func = _varlen_sdpa
    for m in (ref_hy, ref_fx, ref_wan, ref_cog):
        _wrap_kernel(m)
    torch.set_num_threads(8)
    outdir = os.path.dirname(os.path.abspath(__file__))

    def run_case(name, variant, B, H, S, D, top_k, p, nb_width, seed, smooth=0.0, rejected=None, **kw):
        """Runs the reference builder + whole operator; returns dict of arrays or None if oracle disagrees."""
        q, k, v = synth.structured_qkv(seed, B, H, S, D, smooth=smooth)
        tq, tk, tv = (torch.from_numpy(x.copy()) for x in (q, k, v))
        if variant == "hunyuan":
            num_true = kw["num_true"]
            lay = orc.layout_hunyuan(S, num_true)
            mask = torch.zeros(B, 1, 1, S, dtype=torch.bool)
            mask[..., :num_true] = True
            cu = torch.tensor([0, num_true, S], dtype=torch.int32)
            mod = ref_hy
            extra = {}
            call = dict(attn_mask=mask, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S)
        elif variant == "flux":
            tl_ = kw["text_length"]
            lay = orc.layout_flux(S, tl_)
            cu = torch.tensor([0, S, S], dtype=torch.int32)
            mod = ref_fx
            extra = dict(text_length=tl_)
            call = dict(attn_mask=None, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S)
        elif variant == "cogvideo":
            tl_ = kw["text_length"]
            lay = orc.layout_cogvideo(S, tl_)
            cuq = torch.tensor([0, S, S * B], dtype=torch.int32)
            mod = ref_cog
            extra = dict(text_length=tl_)
            call = dict(attn_mask=None, cu_seqlens_q=cuq, cu_seqlens_kv=cuq, max_seqlen_q=S, max_seqlen_kv=S)
        elif variant == "wan":
            ffb = kw.get("ffb", 0)
            lay = orc.layout_wan(S, ffb)
            cu = torch.tensor([0, S, S], dtype=torch.int32)
            mod = ref_wan
            extra = dict(first_frame_blocks=ffb)
            call = dict(attn_mask=None, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S)
        else:
            raise ValueError(variant)
        nbr = synth.banded_neighbors(lay.NBv, nb_width) if nb_width >= 0 else None
        tnbr = torch.from_numpy(nbr) if nbr is not None else None

        # capture the builder outputs of the actual operator run
        captured = {}
        orig_builder = mod._build_block_index_with_importance_optimized

        def spy(*a, **k_):
            r = orig_builder(*a, **k_)
            captured["one_hot"], captured["probs"], captured["nogapr"] = (x.clone() for x in r)
            return r

        mod._build_block_index_with_importance_optimized = spy
        try:
            out = mod.rectified_block_sparse_attention(tq.clone(), tk.clone(), tv.clone(), top_k=top_k,
                                                       block_neighbor_list=tnbr, p_remain_rates=p,
                                                       **call, **extra)
        finally:
            mod._build_block_index_with_importance_optimized = orig_builder
        one_hot = captured["one_hot"].numpy().astype(np.uint8)      # [B,H,NBv,NB_total]
        probs = captured["probs"].numpy().astype(np.float32)        # [B,H,NBv,L]
        nogapr = captured["nogapr"].numpy().astype(np.uint8)        # [B,H,NBv,NBv]
        out = out.float().numpy()

        # oracle agreement on the discrete outputs
        o_out, parts = orc.rectified_attention(q, k, v, lay, top_k, p, nbr, want_parts=True)
        ok = True
        margin = []
        for b in range(B):
            for h in range(H):
                sel = parts[b * H + h]
                if not np.array_equal(sel["kept"], one_hot[b, h]):
                    ok = False
                if not np.array_equal(sel["unrel"], nogapr[b, h]):
                    ok = False
                margin.append(float(np.abs(sel["probs"] - probs[b, h]).max()))
        err_o = float(np.abs(o_out - out).max())
        print(f"{name}: seed {seed} mask/gapr equal={ok} max|dprobs|={max(margin):.2e} max|dO|={err_o:.2e} "
              f"kept={one_hot.mean():.3f} unrel={nogapr.mean():.3f}")
        if not ok:
            # a seed on which reference and oracle disagree is NOT silently skipped: the rows that differ and how close their
            # decisions were go into the accepted fixture's meta (the op_* fixtures were made by trying seeds;
            # every committed one was accepted at the FIRST seed, tests/test_oracle_golden.py pins that)
            rows = []
            for b in range(B):
                for h in range(H):
                    sel = parts[b * H + h]
                    bad = np.nonzero((sel["kept"] != one_hot[b, h]).any(1) | (sel["unrel"] != nogapr[b, h]).any(1))[0]
                    for i in bad[:8]:
                        pr = np.sort(probs[b, h, i].astype(np.float64))[::-1]
                        cs = np.cumsum(pr.astype(np.float32), dtype=np.float32)
                        n = int(sel["n_needed"][i]) if "n_needed" in sel else 0
                        rows.append(dict(b=b, h=h, row=int(i),
                                         kept_bits_differ=int((sel["kept"][i] != one_hot[b, h, i]).sum()),
                                         gapr_bits_differ=int((sel["unrel"][i] != nogapr[b, h, i]).sum()),
                                         cumsum_minus_p_around_n=[float(cs[j] - p) for j in range(max(0, n - 2), min(len(cs), n + 1))],
                                         adjacent_sorted_prob_gaps_around_n=[float(pr[j] - pr[j + 1]) for j in
                                                                             range(max(0, n - 2), min(len(pr) - 1, n + 1))],
                                         max_dprobs=float(np.abs(sel["probs"][i] - probs[b, h, i]).max())))
            return dict(rejected=dict(seed=seed, rows=rows))
        meta = dict(variant=variant, B=B, H=H, S=S, D=D, top_k=top_k, p=p, nb_width=nb_width, seed=seed,
                    smooth=smooth, rejected_seeds=list(rejected or []), **kw)
        out_store = out.astype(np.float16) if S > 8192 else out.astype(np.float32)   # big cases: fp16 storage
        return dict(meta=np.array(repr(meta)), one_hot=np.packbits(one_hot, axis=-1), probs=probs,
                    nogapr=np.packbits(nogapr, axis=-1), out=out_store,
                    one_hot_shape=np.array(one_hot.shape), nogapr_shape=np.array(nogapr.shape))

    cases = [
        # name, variant, B, H, S, D, top_k, p, neighbour band (-1 = None), kwargs
        ("wan_640", "wan", 1, 2, 640, 128, 2, 0.3, 1, dict(ffb=2)),
        ("wan_pad_1450", "wan", 1, 2, 1450, 128, 3, 0.3, 1, dict(ffb=2)),
        ("wan_d64_1100", "wan", 1, 2, 1100, 64, 2, 0.5, 1, dict(ffb=0)),
        ("wan_nonbr_1024", "wan", 1, 1, 1024, 128, 2, 0.3, -1, dict(ffb=1)),
        ("hunyuan_1280", "hunyuan", 1, 2, 1280, 128, 2, 0.3, 1, dict(num_true=1024 + 200)),
        ("hunyuan_3328", "hunyuan", 1, 2, 3328, 128, 5, 0.3, 1, dict(num_true=3072 + 77)),
        ("hunyuan_full_1536", "hunyuan", 1, 1, 1536, 128, 3, 0.2, 1, dict(num_true=1536)),
        ("flux_1536", "flux", 1, 2, 1536, 128, 2, 0.3, 1, dict(text_length=512)),
        ("cogvideo_994", "cogvideo", 1, 2, 994, 64, 2, 0.3, 1, dict(text_length=226)),
        ("wan_smooth_2048", "wan", 1, 2, 2048, 128, 4, 0.6, 2, dict(ffb=3, smooth=0.8)),
        # round 2: batch of two per layout (the reference kernel reads seqlens[0] for every batch item, SURVEY B-1)
        ("b2_wan_900", "wan", 2, 2, 900, 128, 2, 0.3, 1, dict(ffb=1)),
        ("b2_hunyuan_1280", "hunyuan", 2, 1, 1280, 128, 2, 0.3, 1, dict(num_true=1024 + 150)),
        ("b2_flux_1280", "flux", 2, 1, 1280, 128, 2, 0.3, 1, dict(text_length=256)),
        ("b2_cogvideo_994", "cogvideo", 2, 2, 994, 64, 2, 0.3, 1, dict(text_length=226)),
        # round 2: a row longer than 256 columns (K3's sorted-head path) through the reference, 260 blocks
        ("big_wan_33280", "wan", 1, 1, 33280, 64, 20, 0.3, 1, dict(ffb=3)),
        # round 3: the text-tail layouts with rows longer than 128 visual blocks (IPAR + text columns on K3's sorted-head
        # path) through the reference: 136 visual blocks + 256 text (200 valid) / 132 visual blocks + 512 text tokens
        ("big_hunyuan_17664", "hunyuan", 1, 1, 17664, 128, 14, 0.3, 1, dict(num_true=17408 + 200)),
        ("big_flux_17408", "flux", 1, 1, 17408, 128, 13, 0.3, 1, dict(text_length=512)),
        # ... and CogVideoX's layout (head dim 64, 226 text tokens, padded to x128): 136 visual blocks
        ("big_cogvideo_17634", "cogvideo", 1, 1, 17634, 64, 14, 0.3, 1, dict(text_length=226)),
    ]
    only = os.environ.get("RSA_GOLDEN_ONLY")
    if only:
        cases = [c for c in cases if c[0].startswith(only)]
    main_only_ops = only is not None
    for name, variant, B, H, S, D, top_k, p, nbw, kw in cases:
        kw = dict(kw)
        smooth = kw.pop("smooth", 0.0)
        rejected = []
        for seed in range(FIRST_SEED, FIRST_SEED + 20):
            try:
                res = run_case(name, variant, B, H, S, D, top_k, p, nbw, seed, smooth=smooth, rejected=rejected, **kw)
            except (TypeError, RuntimeError, ValueError) as e:
                if B == 1:
                    raise
                # B > 1 is outside what some reference operators can run (wan21 :329 indexes with a per-batch
                # tensor); such layouts get no B = 2 vector and the tests use the oracle alone for them
                print(f"{name}: the reference cannot run this layout with B={B}: {type(e).__name__}: {str(e)[:100]}")
                res = "unsupported"
                break
            if "rejected" in res:
                rejected.append(res["rejected"])
                continue
            np.savez_compressed(os.path.join(outdir, f"op_{name}.npz"), **res)
            break
        else:
            raise SystemExit(f"no agreeing seed for {name}")
        if isinstance(res, str):
            continue

    if main_only_ops:
        return
    # ---- estimate_pr_gain stand-alone (gapr_mask.py:4) ------------------------------------------------
    q, k, _ = synth.structured_qkv(7, 1, 2, 1024, 128)
    Qb = torch.from_numpy(q).reshape(1, 2, 8, 128, 128)
    Kb = torch.from_numpy(k).reshape(1, 2, 8, 128, 128)
    qp, kp = Qb.mean(-2), Kb.mean(-2)
    sc = torch.matmul(qp, kp.transpose(-1, -2))
    g = ref_gapr.estimate_pr_gain(Qb, Kb, qp, kp, sc)
    np.savez_compressed(os.path.join(outdir, "gapr_1024.npz"), seed=7, mask=np.packbits(g.numpy(), axis=-1),
                        shape=np.array(g.shape), q_pools=qp.numpy(), k_pools=kp.numpy(), scores=sc.numpy())
    print("gapr_1024: unreliable fraction", float(g.float().mean()))

    dense()


def dense():
    """fullattn(mode="torch" | "vanilla" | "flash") of the reference (attn.py:60-154) on the config-1 shape reduced to one
    head; round 3: also causal=True through "torch" and "vanilla"."""
    import rectified_spaattn.attn as ref_attn
    from rectified_spaattn_amd import synth
    ref_attn.flash_attn_varlen_func = _varlen_sdpa
    outdir = os.path.dirname(os.path.abspath(__file__))
    q, k, v = synth.structured_qkv(11, 1, 1, 1536, 128)
    tq, tk, tv = (torch.from_numpy(x) for x in (q, k, v))
    am = torch.zeros(1, 1, 1, 1536, dtype=torch.bool)
    am[..., :1400] = True
    o_t = ref_attn.fullattn(tq, tk, tv, mode="torch")
    o_v = ref_attn.fullattn(tq, tk, tv, mode="vanilla")
    o_tm = ref_attn.fullattn(tq, tk, tv, mode="torch", attn_mask=am)
    o_vm = ref_attn.fullattn(tq, tk, tv, mode="vanilla", attn_mask=am)
    cu = torch.tensor([0, 1400, 1536], dtype=torch.int32)
    o_f = ref_attn.fullattn(tq, tk, tv, mode="flash", cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=1536,
                            max_seqlen_kv=1536, batch_size=1)
    o_tc = ref_attn.fullattn(tq, tk, tv, mode="torch", causal=True)
    o_vc = ref_attn.fullattn(tq, tk, tv, mode="vanilla", causal=True)
    assert float((o_tm - o_vm).abs().max()) < 1e-5 and float((o_tc - o_vc).abs().max()) < 1e-5
    np.savez_compressed(os.path.join(outdir, "dense_1536.npz"), seed=11, torch=o_t.numpy(),
                        vanilla_masked=o_vm.numpy(), flash_varlen=o_f.numpy(), n_valid=1400, torch_causal=o_tc.numpy())
    print("dense_1536 done; |torch-vanilla| =", float((o_t - o_v).abs().max()))


def processors():
    """Processor-level vectors: the reference processors' __call__ on CPU (dense modes) through fake
    `Attention` modules (tests/helpers.py) -- pins projection / norm / RoPE / concat order / split logic."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import rectified_spaattn.attn as ref_attn
    import rectified_spaattn.rectified_hunyuan_attn as ref_hy
    import rectified_spaattn.rectified_flux_attn as ref_fx
    import rectified_spaattn.rectified_wan21_attn as ref_wan
    import rectified_spaattn.rectified_cogvideo_attn as ref_cog
    ref_attn.flash_attn_varlen_func = _varlen_sdpa
    outdir = os.path.dirname(os.path.abspath(__file__))
    heads, hd = 2, 128
    dim = heads * hd
    res = {}
    with torch.no_grad():
        # HunyuanVideo dual-stream block, mode "torch", padded text (200 of 256 valid)
        a = helpers.fake_attn(101, heads, hd, added=True)
        hs, enc = helpers.hidden(101, 20, 1, 1024, dim), helpers.hidden(101, 21, 1, 256, dim)
        mask = torch.zeros(1, 1, 1, 1280, dtype=torch.bool)
        mask[..., :1224] = True
        rope = helpers.rope_tables(1024, hd)
        p = ref_hy.RectifiedHunyuanVideoSpaAttnProcessor2_0("torch", 2, None, 0.3, 0)
        o, e = p(a, hs, enc, mask, rope)
        res["hy_dual_out"], res["hy_dual_enc"] = o.numpy(), e.numpy()
        # HunyuanVideo single-stream block (add_q_proj None), mode "vanilla"
        a = helpers.fake_attn(102, heads, hd, added=False)
        p = ref_hy.RectifiedHunyuanVideoSpaAttnProcessor2_0("vanilla", 2, None, 0.3, 25)
        o, e = p(a, hs, enc, mask, rope)
        res["hy_single_out"], res["hy_single_enc"] = o.numpy(), e.numpy()
        # Flux double-stream block at the config-1 shape (1024 image + 512 text), dense
        a = helpers.fake_attn(103, heads, hd, added=True)
        hs_f, enc_f = helpers.hidden(103, 20, 1, 1024, dim), helpers.hidden(103, 21, 1, 512, dim)
        rope_f = helpers.rope_tables(1536, hd)
        p = ref_fx.RectifiedFluxSpaAttnProcessor2_0("torch", 2, None, 0.3, 0, 512)
        o, e = p(a, hs_f, enc_f, None, rope_f)
        res["fx_dual_out"], res["fx_dual_enc"] = o.numpy(), e.numpy()
        # Flux single-stream block (no encoder states): [image | text] already concatenated
        a = helpers.fake_attn(104, heads, hd, added=False)
        p = ref_fx.RectifiedFluxSpaAttnProcessor2_0("torch", 2, None, 0.3, 40, 512)
        res["fx_single_out"] = p(a, helpers.hidden(104, 22, 1, 1536, dim), None, None, rope_f).numpy()
        # Wan2.1 T2V self-attention, mode "torch", complex RoPE, S not a multiple of 128
        a = helpers.fake_attn(105, heads, hd, wan=True)
        p = ref_wan.RectifiedWanT2VSpaAttnProcessor2_0("torch", 2, None, 0.3, 5, 1)
        res["wan_self_out"] = p(a, helpers.hidden(105, 20, 1, 900, dim), None, None,
                                helpers.wan_freqs(900, hd)).numpy()
        # Wan2.1 cross-attention (attn2 processors run mode "flash"): S_q != S_k
        p = ref_wan.RectifiedWanT2VSpaAttnProcessor2_0("flash", 2, None, 0.3, 5, 1)
        res["wan_cross_out"] = p(a, helpers.hidden(105, 20, 1, 900, dim), helpers.hidden(105, 23, 1, 512, dim),
                                 None, None).numpy()
        # CogVideoX, head_dim 64, dense (step counter < 5)
        a = helpers.fake_attn(106, 4, 64, added=False)
        p = ref_cog.RectifiedCogVideoXVideoSpaAttnProcessor2_0("sparse", 2, None, 0.3, 0)
        o, e = p(a, helpers.hidden(106, 20, 1, 768, 256), helpers.hidden(106, 21, 1, 226, 256), None,
                 helpers.rope_tables(768, 64))
        res["cog_out"], res["cog_enc"] = o.numpy(), e.numpy()
    np.savez_compressed(os.path.join(outdir, "processors.npz"), **{k: v.astype(np.float16) for k, v in res.items()})
    print("processors.npz:", {k: v.shape for k, v in res.items()})


def processors_round2():
    """Round-2 processor vectors: the three Wan2.2 processors (dense AND sparse branches), the CogVideoX sparse
    branch, and B = 2 through a processor.  Wan2.2 imports `_get_qkv_projections` / `_get_added_kv_projections` from
    diffusers.models.transformers.transformer_wan; the stubs installed here are the documented NON-FUSED behaviour of
    those helpers (to_q / to_k / to_v of hidden_states or encoder_hidden_states; add_k_proj / add_v_proj of the image
    context) -- the fused variant only concatenates the same weight matrices."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import rectified_spaattn.attn as ref_attn
    import rectified_spaattn.rectified_wan21_attn as ref_wan
    import rectified_spaattn.rectified_cogvideo_attn as ref_cog
    ref_attn.flash_attn_varlen_func = _varlen_sdpa
    tw = sys.modules["diffusers.models.transformers.transformer_wan"]

    def _get_qkv_projections(attn, hidden_states, encoder_hidden_states):
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        return attn.to_q(hidden_states), attn.to_k(encoder_hidden_states), attn.to_v(encoder_hidden_states)

    def _get_added_kv_projections(attn, enc_img):
        return attn.add_k_proj(enc_img), attn.add_v_proj(enc_img)

    tw._get_qkv_projections, tw._get_added_kv_projections = _get_qkv_projections, _get_added_kv_projections
    import rectified_spaattn.rectified_wan22_attn as ref_w22
    for m in (ref_wan, ref_cog):
        if not getattr(m, "_rsa_wrapped", False):
            _wrap_kernel(m)
            m._rsa_wrapped = True
    # wan22 imported rectified_block_sparse_attention from wan21: the wrapped kernel is looked up in ref_wan at call time
    outdir = os.path.dirname(os.path.abspath(__file__))
    heads, hd = 2, 128
    dim = heads * hd
    res = {}
    from rectified_spaattn_amd import synth
    with torch.no_grad():
        S = 900
        a = helpers.fake_attn(111, heads, hd, wan=True)
        hs = helpers.hidden(111, 20, 1, S, dim)
        rope = helpers.wan22_rope(S, hd)
        nbr = torch.from_numpy(synth.banded_neighbors((S + 127) // 128, 1))
        # TI2V: dense ("torch"), sparse warm-up (step < 10 -> flash), sparse proper (layer 3, step 10)
        p = ref_w22.RectifiedWanTI2VSpaAttnProcessor2_0("torch", 2, None, 0.3, 3, 1)
        res["w22_ti2v_dense"] = p(a, hs, None, None, rope).numpy()
        p = ref_w22.RectifiedWanTI2VSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, 3, 1)
        res["w22_ti2v_warm"] = p(a, hs, None, None, rope).numpy()
        p.current_step = 10
        res["w22_ti2v_sparse"] = p(a, hs, None, None, rope).numpy()
        assert p.current_step == 11
        # T2V (A14B): warm_steps, dense on layers 0/1/40/41
        p = ref_w22.RectifiedWanT2VSpaAttnProcessor2_0("sparse", 3, nbr, 0.4, 5, 0, warm_steps=2)
        p.current_step = 2
        res["w22_t2v_sparse"] = p(a, hs, None, None, rope).numpy()
        p = ref_w22.RectifiedWanT2VSpaAttnProcessor2_0("sparse", 3, nbr, 0.4, 40, 0, warm_steps=0)
        res["w22_t2v_layer40_dense"] = p(a, hs, None, None, rope).numpy()
        # I2V (A14B) without an image context (the A14B I2V model feeds the image through the latent channels)
        p = ref_w22.RectifiedWanI2VSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, 7, 2, warm_steps=0)
        res["w22_i2v_sparse"] = p(a, hs, None, None, rope).numpy()
        # cross attention of a Wan2.2 block (mode "flash", text keys only)
        p = ref_w22.RectifiedWanT2VSpaAttnProcessor2_0("flash", 2, None, 0.3, 5, 0)
        res["w22_cross"] = p(a, hs, helpers.hidden(111, 23, 1, 512, dim), None, None).numpy()
        # the reference's image-context branch (:94-97) feeds query as [B,S,H,D] to SDPA against [B,H,S_img,D] keys:
        # record what it does with it (an error unless S == H)
        a_img = helpers.fake_attn_wan_i2v(112, heads, hd)
        try:
            p = ref_w22.RectifiedWanI2VSpaAttnProcessor2_0("torch", 2, None, 0.3, 7, 0)
            p(a_img, hs, helpers.hidden(112, 24, 1, 512 + 17, dim), None, rope)
            res["w22_imgctx_reference_runs"] = np.array(1)
        except Exception as e:  # noqa: BLE001
            print("reference Wan2.2 image-context branch raises:", type(e).__name__, str(e)[:120])
            res["w22_imgctx_reference_runs"] = np.array(0)

        # CogVideoX sparse branch (step counter >= 5), head_dim 64, S = 768 + 226 (padded to x128 inside)
        a = helpers.fake_attn(106, 4, 64, added=False)
        nbr_c = torch.from_numpy(synth.banded_neighbors(6, 1))
        p = ref_cog.RectifiedCogVideoXVideoSpaAttnProcessor2_0("sparse", 2, nbr_c, 0.3, 0)
        p.current_step = 5
        o, e = p(a, helpers.hidden(106, 20, 1, 768, 256), helpers.hidden(106, 21, 1, 226, 256), None,
                 helpers.rope_tables(768, 64))
        res["cog_sparse_out"], res["cog_sparse_enc"] = o.numpy(), e.numpy()
        # CogVideoX dense warm-up with B = 2 (classifier-free guidance batches cond / uncond): cu_seqlens [0, S, 2S]
        p = ref_cog.RectifiedCogVideoXVideoSpaAttnProcessor2_0("sparse", 2, nbr_c, 0.3, 0)
        o, e = p(a, helpers.hidden(106, 30, 2, 768, 256), helpers.hidden(106, 31, 2, 226, 256), None,
                 helpers.rope_tables(768, 64))
        res["cog_b2_out"], res["cog_b2_enc"] = o.numpy(), e.numpy()
    np.savez_compressed(os.path.join(outdir, "processors_r2.npz"),
                        **{k: (v.astype(np.float16) if v.ndim else v) for k, v in res.items()})
    print("processors_r2.npz:", {k: v.shape for k, v in res.items()})


def processors_round3():
    """Round-3 vectors for the processors' SPARSE branches: besides the processor output (processors_r2.npz) the kept-block
    mask the reference's builder produced and the operator's own output [B, S, H*D] of the same call, so that the device
    test can compare mask against mask first and then bound the output on the query blocks whose kept set agrees
    (the device processor projects in bf16, the reference in fp32: a few rows near a threshold may flip)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import rectified_spaattn.attn as ref_attn
    import rectified_spaattn.rectified_wan21_attn as ref_wan
    import rectified_spaattn.rectified_cogvideo_attn as ref_cog
    ref_attn.flash_attn_varlen_func = _varlen_sdpa
    tw = sys.modules["diffusers.models.transformers.transformer_wan"]
    tw._get_qkv_projections = lambda attn, hs, enc: (attn.to_q(hs), attn.to_k(hs if enc is None else enc),
                                                     attn.to_v(hs if enc is None else enc))
    tw._get_added_kv_projections = lambda attn, enc_img: (attn.add_k_proj(enc_img), attn.add_v_proj(enc_img))
    import rectified_spaattn.rectified_wan22_attn as ref_w22
    for m in (ref_wan, ref_cog):
        if not getattr(m, "_rsa_wrapped", False):
            _wrap_kernel(m)
            m._rsa_wrapped = True
    from rectified_spaattn_amd import synth
    outdir = os.path.dirname(os.path.abspath(__file__))
    res = {}

    def spied(mod, key, fn):
        cap = {}
        ob, oc = mod._build_block_index_with_importance_optimized, mod.block_sparse_attention_combined

        def sb(*a, **k):
            r = ob(*a, **k)
            cap["one_hot"] = r[0].clone()
            return r

        def sc(*a, **k):
            r = oc(*a, **k)
            cap["op_out"] = r.clone()
            return r
        mod._build_block_index_with_importance_optimized, mod.block_sparse_attention_combined = sb, sc
        try:
            fn()
        finally:
            mod._build_block_index_with_importance_optimized, mod.block_sparse_attention_combined = ob, oc
        oh = cap["one_hot"].numpy().astype(np.uint8)
        res[key + "_mask"] = np.packbits(oh, axis=-1)
        res[key + "_mask_shape"] = np.array(oh.shape)
        res[key + "_op_out"] = cap["op_out"].float().numpy().astype(np.float16)

    heads, hd = 2, 128
    dim = heads * hd
    with torch.no_grad():
        S = 900
        a = helpers.fake_attn(111, heads, hd, wan=True)
        hs = helpers.hidden(111, 20, 1, S, dim)
        rope = helpers.wan22_rope(S, hd)
        nbr = torch.from_numpy(synth.banded_neighbors((S + 127) // 128, 1))
        p = ref_w22.RectifiedWanTI2VSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, 3, 1)
        p.current_step = 10
        spied(ref_wan, "w22_ti2v_sparse", lambda: p(a, hs, None, None, rope))
        p2 = ref_w22.RectifiedWanT2VSpaAttnProcessor2_0("sparse", 3, nbr, 0.4, 5, 0, warm_steps=2)
        p2.current_step = 2
        spied(ref_wan, "w22_t2v_sparse", lambda: p2(a, hs, None, None, rope))
        p3 = ref_w22.RectifiedWanI2VSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, 7, 2, warm_steps=0)
        spied(ref_wan, "w22_i2v_sparse", lambda: p3(a, hs, None, None, rope))
        ac = helpers.fake_attn(106, 4, 64, added=False)
        nbr_c = torch.from_numpy(synth.banded_neighbors(6, 1))
        pc = ref_cog.RectifiedCogVideoXVideoSpaAttnProcessor2_0("sparse", 2, nbr_c, 0.3, 0)
        pc.current_step = 5
        spied(ref_cog, "cog_sparse", lambda: pc(ac, helpers.hidden(106, 20, 1, 768, 256), helpers.hidden(106, 21, 1, 226, 256),
                                                None, helpers.rope_tables(768, 64)))
    np.savez_compressed(os.path.join(outdir, "processors_r3.npz"), **res)
    print("processors_r3.npz:", {k: v.shape for k, v in res.items()})


def processors_round4():
    """Round-4 vectors: (i) the Flux processor's SPARSE branch through the reference (double-stream block, 1024 image + 512
    text tokens, processor ids 0 and 57 -- the gate `processor_id < 37 or >= 57`, rectified_flux_attn.py:493) with the
    reference's kept mask and operator output of the same call, and the dense warm-up layer id 40; (ii) the Wan2.1 I2V
    processor with an IMAGE CONTEXT (257 CLIP tokens + 512 text tokens through add_k_proj / add_v_proj / norm_added_k,
    rectified_wan21_attn.py:512-632).  Shim for (ii): `sdpa_kernel(FLASH_ATTENTION)` has no CPU fp32 kernel -> null context."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import rectified_spaattn.attn as ref_attn
    import rectified_spaattn.rectified_flux_attn as ref_fx
    import rectified_spaattn.rectified_wan21_attn as ref_wan
    ref_attn.flash_attn_varlen_func = _varlen_sdpa
    for m in (ref_fx, ref_wan):
        if hasattr(m, "flash_attn_varlen_func"):
            m.flash_attn_varlen_func = _varlen_sdpa
    if not getattr(ref_fx, "_rsa_wrapped", False):
        _wrap_kernel(ref_fx)
        ref_fx._rsa_wrapped = True
    ref_wan.sdpa_kernel = lambda *a, **k: contextlib.nullcontext()
    from rectified_spaattn_amd import synth
    outdir = os.path.dirname(os.path.abspath(__file__))
    res = {}
    heads, hd = 2, 128
    dim = heads * hd

    def spied(mod, key, fn):
        cap = {}
        ob, oc = mod._build_block_index_with_importance_optimized, mod.block_sparse_attention_combined

        def sb(*a, **k):
            r = ob(*a, **k)
            cap["one_hot"] = r[0].clone()
            return r

        def sc(*a, **k):
            r = oc(*a, **k)
            cap["op_out"] = r.clone()
            return r
        mod._build_block_index_with_importance_optimized, mod.block_sparse_attention_combined = sb, sc
        try:
            out = fn()
        finally:
            mod._build_block_index_with_importance_optimized, mod.block_sparse_attention_combined = ob, oc
        oh = cap["one_hot"].numpy().astype(np.uint8)
        res[key + "_mask"] = np.packbits(oh, axis=-1)
        res[key + "_mask_shape"] = np.array(oh.shape)
        res[key + "_op_out"] = cap["op_out"].float().numpy().astype(np.float16)
        return out

    with torch.no_grad():
        a = helpers.fake_attn(103, heads, hd, added=True)
        hs_f, enc_f = helpers.hidden(103, 20, 1, 1024, dim), helpers.hidden(103, 21, 1, 512, dim)
        rope_f = helpers.rope_tables(1536, hd)
        nbr = torch.from_numpy(synth.banded_neighbors(8, 1))
        for pid in (0, 57):
            pr = ref_fx.RectifiedFluxSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, pid, 512)
            o, e = spied(ref_fx, f"fx_sparse_id{pid}", lambda: pr(a, hs_f, enc_f, None, rope_f))
            res[f"fx_sparse_id{pid}_out"], res[f"fx_sparse_id{pid}_enc"] = o.numpy().astype(np.float16), e.numpy().astype(np.float16)
            assert pr.current_step == 1
        pr = ref_fx.RectifiedFluxSpaAttnProcessor2_0("sparse", 2, nbr, 0.3, 40, 512)     # warm-up layers 37..56: dense
        o, e = pr(a, hs_f, enc_f, None, rope_f)
        res["fx_sparse_id40_out"], res["fx_sparse_id40_enc"] = o.numpy().astype(np.float16), e.numpy().astype(np.float16)
        # Wan2.1 I2V cross attention with an image context: [257 image tokens | 512 text tokens]
        ai = helpers.fake_attn_wan_i2v(112, heads, hd)
        hs, enc = helpers.hidden(111, 20, 1, 900, dim), helpers.hidden(112, 24, 1, 257 + 512, dim)
        pw = ref_wan.RectifiedWanI2VSpaAttnProcessor2_0("flash", 2, None, 0.3, 5, 0)
        res["wan21_i2v_imgctx"] = pw(ai, hs, enc, None, None).numpy().astype(np.float16)
    # ids 0 and 57 see the same inputs: the second call is stored as "equal to the first" (asserted here), not as arrays
    same = all(np.array_equal(res[f"fx_sparse_id0{k}"], res[f"fx_sparse_id57{k}"]) for k in ("_mask", "_op_out", "_out", "_enc"))
    assert same
    for k in ("_mask", "_mask_shape", "_op_out", "_out", "_enc"):
        del res[f"fx_sparse_id57{k}"]
    res["fx_sparse_id57_equals_id0"] = np.array(1)
    np.savez_compressed(os.path.join(outdir, "processors_r4.npz"), **res)
    print("processors_r4.npz:", {k: v.shape for k, v in res.items()})


def headline():
    """Round 4: the reference's OWN mask builder at BASELINE's full sizes, one head, fp32 statistics.

    `_build_block_index_with_importance_optimized` (hunyuan :171-280, flux :170-279, wan21 :171-273) is reached through the
    whole operator (so key / value zeroing, padding, `attenable`, text block range are the operator's own), with the Triton
    kernel and the flash call replaced by zero stubs -- only the mask-selection outputs are taken.  Inputs =
    synth.structured_qkv (the CPU twin of the device generator, independent block centroids = bench regime R2), the
    HunyuanVideo case with the true Gilbert neighbour matrix.  Stored: bit-packed one-hot mask, bit-packed GAPR mask, and
    num_blocks_needed recomputed from the reference's returned probabilities with the reference's own torch expression
    (sort descending, cumsum, <= p, +1, max with top_k: hunyuan :226-238) -- all < 1 MB per layout.

    NO reseeding: if a row of the oracle differs from the reference, its margins are printed and stored
    (`mismatch_rows`, `mismatch_margin`) and the tests report them.
    """
    _install_stubs()
    import rectified_spaattn.attn as ref_attn
    import rectified_spaattn.rectified_hunyuan_attn as ref_hy
    import rectified_spaattn.rectified_flux_attn as ref_fx
    import rectified_spaattn.rectified_wan21_attn as ref_wan
    from oracle import oracle as orc
    from rectified_spaattn_amd import synth
    from rectified_spaattn_amd.utils import jenga_gilbert

    torch.set_num_threads(8)
    outdir = os.path.dirname(os.path.abspath(__file__))
    ref_attn.flash_attn_varlen_func = lambda q, k, v, *a, **kw: torch.zeros_like(q)
    for m in (ref_hy, ref_fx, ref_wan):
        m._triton_block_sparse_attention_onehot = lambda q, k, v, *a, **kw: torch.zeros_like(q)
        if hasattr(m, "flash_attn_varlen_func"):
            m.flash_attn_varlen_func = ref_attn.flash_attn_varlen_func

    cases = [
        # name, variant, S, top_k, p, seed, kwargs   (test_gpu_fullsize.py's configurations)
        ("hunyuan_115456", "hunyuan", 115456, 90, 0.05, 20251301, dict(num_true=115400, gilbert=(32, 45, 80))),
        ("flux_66048", "flux", 66048, 51, 0.3, 20251302, dict(text_length=512)),
        ("wan_75600", "wan", 75600, 147, 0.3, 20251303, dict(ffb=28)),
    ]
    only = os.environ.get("RSA_GOLDEN_ONLY")
    for name, variant, S, top_k, p, seed, kw in cases:
        if only and not name.startswith(only):
            continue
        D, B, H = 128, 1, 1
        q, k, v = synth.structured_qkv(seed, B, H, S, D)
        tq, tk, tv = (torch.from_numpy(x.copy()) for x in (q, k, v))
        nbr = None
        if variant == "hunyuan":
            nt = kw["num_true"]
            lay = orc.layout_hunyuan(S, nt)
            mask = torch.zeros(B, 1, 1, S, dtype=torch.bool)
            mask[..., :nt] = True
            cu = torch.tensor([0, nt, S], dtype=torch.int32)
            mod, extra = ref_hy, {}
            call = dict(attn_mask=mask, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S)
            nbr = jenga_gilbert.gilbert_block_neighbor_mapping(*kw["gilbert"]).numpy()
        elif variant == "flux":
            lay = orc.layout_flux(S, kw["text_length"])
            cu = torch.tensor([0, S, S], dtype=torch.int32)
            mod, extra = ref_fx, dict(text_length=kw["text_length"])
            call = dict(attn_mask=None, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S)
        else:
            lay = orc.layout_wan(S, kw["ffb"])
            cu = torch.tensor([0, S, S], dtype=torch.int32)
            mod, extra = ref_wan, dict(first_frame_blocks=kw["ffb"])
            call = dict(attn_mask=None, cu_seqlens_q=cu, cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S)
        tnbr = torch.from_numpy(nbr) if nbr is not None else None
        captured = {}
        orig_builder = mod._build_block_index_with_importance_optimized

        def spy(*a, **k_):
            r = orig_builder(*a, **k_)
            captured["one_hot"], captured["probs"], captured["nogapr"] = (x.clone() for x in r)
            return r

        mod._build_block_index_with_importance_optimized = spy
        try:
            mod.rectified_block_sparse_attention(tq, tk, tv, top_k=top_k, block_neighbor_list=tnbr, p_remain_rates=p,
                                                 **call, **extra)
        finally:
            mod._build_block_index_with_importance_optimized = orig_builder
        one_hot = captured["one_hot"][0, 0].numpy().astype(np.uint8)     # [NBv, NB_total]
        nogapr = captured["nogapr"][0, 0].numpy().astype(np.uint8)       # [NBv, NBv]
        rprobs = captured["probs"][0, 0]                                 # [NBv, L] fp32
        sp, _ = torch.sort(rprobs, dim=-1, descending=True)
        nbn = torch.maximum((torch.cumsum(sp, dim=-1) <= p).sum(-1) + 1, torch.tensor(top_k)).numpy().astype(np.int32)
        # ---- the oracle on the same head (test infrastructure talking to test infrastructure) ----
        qh, kh, vh = q[0, 0], k[0, 0].copy(), v[0, 0].copy()
        if lay.pool_valid < lay.S:
            kh[lay.pool_valid:] = 0
            vh[lay.pool_valid:] = 0
        sel = orc.select_head(qh, kh, vh, lay, top_k, p, nbr)
        bad_rows = [i for i in range(lay.NBv) if not (np.array_equal(sel["kept"][i], one_hot[i])
                                                      and np.array_equal(sel["unrel"][i], nogapr[i]))]
        dpr = float(np.abs(sel["probs"] - rprobs.numpy()).max())
        nn_eq = int((np.maximum(sel["n_needed"], top_k) == nbn).sum())
        print(f"headline {name}: NBv={lay.NBv} L={lay.L} kept={one_hot.mean():.4f} unrel={nogapr.mean():.4f} "
              f"rows differing from the oracle: {len(bad_rows)}  max|dprobs|={dpr:.2e}  n_needed equal on {nn_eq}/{lay.NBv}")
        margins = []
        for i in bad_rows:
            # margins of the row: smallest distance of the sorted cumulative sum to p, smallest gap between adjacent sorted
            # probabilities around the cut, smallest |gain - err| of a flipped GAPR bit
            po = np.sort(sel["probs"][i])[::-1].astype(np.float64)
            cs = np.cumsum(po)
            cut = int(nbn[i])
            gap = float(po[max(cut - 2, 0): cut + 1][:-1].min() - po[max(cut - 2, 0): cut + 1][1:].max()) if cut >= 2 else 0.0
            margins.append((i, float(np.abs(cs - p).min()), gap, int((sel["kept"][i] != one_hot[i]).sum()),
                            int((sel["unrel"][i] != nogapr[i]).sum())))
            print(f"   row {i}: min|cumsum - p| = {margins[-1][1]:.3e}, prob gap at the cut = {gap:.3e}, "
                  f"mask bits differing {margins[-1][3]}, GAPR bits differing {margins[-1][4]}")
        meta = dict(variant=variant, S=S, D=D, top_k=top_k, p=p, seed=seed, **kw)
        np.savez_compressed(os.path.join(outdir, f"headline_{name}.npz"), meta=np.array(repr(meta)),
                            one_hot=np.packbits(one_hot, axis=-1), one_hot_shape=np.array(one_hot.shape),
                            nogapr=np.packbits(nogapr, axis=-1), nogapr_shape=np.array(nogapr.shape),
                            num_blocks_needed=nbn.astype(np.int16),
                            probs_rowsum=rprobs.sum(-1).numpy().astype(np.float32),
                            probs_sample=rprobs[:: max(1, lay.NBv // 8)].numpy().astype(np.float32),
                            mismatch_rows=np.array(bad_rows, np.int32),
                            mismatch_margin=np.array(margins, np.float64).reshape(-1, 5))



def _load_script(name, extra_stubs=()):
    """Execute /root/reference/scripts/<name>.py as a module (its __main__ block does not run) with the diffusers /
    torchvision-dependent imports replaced by inert stand-ins."""
    import importlib.util

    def mod(name_, **attrs):
        m = types.ModuleType(name_)
        m.__dict__.update(attrs)
        sys.modules[name_] = m
        return m

    class _Any:
        pass

    class _Out:
        def __init__(self, sample):
            self.sample = sample

    d = sys.modules.get("diffusers") or mod("diffusers")
    for n in ("HunyuanVideoPipeline", "HunyuanVideoTransformer3DModel", "AutoencoderKLWan", "WanPipeline",
              "CogVideoXPipeline", "CogVideoXImageToVideoPipeline", "FluxPipeline", "FluxControlNetModel",
              "WanImageToVideoPipeline"):
        setattr(d, n, _Any)
    mod("diffusers.pipelines", FluxControlNetPipeline=_Any)
    mod("diffusers.schedulers")
    mod("diffusers.schedulers.scheduling_unipc_multistep", UniPCMultistepScheduler=_Any)
    log = types.SimpleNamespace(get_logger=lambda n: types.SimpleNamespace(warning=lambda *a, **k: None))
    mod("diffusers.utils", USE_PEFT_BACKEND=False, logging=log, scale_lora_layers=lambda *a: None,
        unscale_lora_layers=lambda *a: None, is_torch_version=lambda *a: True, export_to_video=lambda *a, **k: None,
        load_image=lambda *a, **k: None)
    mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=_Out)
    mod("utils.seed", set_seed=lambda s: None)
    mod("utils.save_video", save_videos_grid=lambda *a, **k: None)
    torch.cuda.synchronize = lambda *a, **k: None
    spec = importlib.util.spec_from_file_location("ref_script_" + name, os.path.join(REF, "scripts", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    cvd = os.environ.get("CUDA_VISIBLE_DEVICES")
    spec.loader.exec_module(m)
    if cvd is None:
        os.environ.pop("CUDA_VISIBLE_DEVICES", None)
    else:
        os.environ["CUDA_VISIBLE_DEVICES"] = cvd
    return m


def teacache():
    """TeaCache decision sequences produced by the REFERENCE's own patched forwards (scripts/main_hunyuan.py:47-210,
    scripts/main_wan21t2v.py:47-203), run on a stand-in transformer whose modulated input is a known drifting tensor
    sequence (tests/helpers.py::teacache_sequence).  Stored: per forward call, whether the blocks ran."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import matplotlib
    matplotlib.use("Agg")
    outdir = os.path.dirname(os.path.abspath(__file__))
    res = {}
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        hy = _load_script("main_hunyuan")
        wan = _load_script("main_wan21t2v")

    class Block(torch.nn.Module):
        def __init__(self, counter):
            super().__init__()
            self.counter = counter

        def norm1(self, inp, emb=None):
            return inp, None, None, None, None

    class HyBlock(Block):
        def forward(self, hs, enc, temb, mask, rope, tre, ffnt):
            self.counter[0] += 1
            return hs * 1.01 + 0.1, enc

    class WanBlock(Block):
        def forward(self, hs, enc, tproj, rope):
            self.counter[0] += 1
            return hs * 1.01 + 0.1

    # ---- HunyuanVideo: single stream --------------------------------------------------------------------------
    for ci, (seed, drift, thresh, num_steps) in enumerate(helpers.TEACACHE_HUNYUAN_CASES):
        C, T, Hh, W = 8, 2, 4, 4
        N = T * Hh * W
        counter = [0]
        order = torch.arange(N)
        me = types.SimpleNamespace(
            config=types.SimpleNamespace(patch_size=1, patch_size_t=1),
            rope=lambda x: (torch.ones(N, 4), torch.zeros(N, 4)),
            time_text_embed=lambda t, pp, g: (torch.zeros(1, 4), None),
            x_embedder=lambda x: x.flatten(2).transpose(1, 2),
            context_embedder=lambda e, t, m: e,
            hilbert_order=order, linear_to_hilbert=order,
            transformer_blocks=[HyBlock(counter)], single_transformer_blocks=[],
            norm_out=lambda h, t: h, proj_out=lambda h: h,
            enable_teacache=True, cnt=0, num_steps=num_steps, rel_l1_thresh=thresh,
            accumulated_rel_l1_distance=0, previous_modulated_input=None, previous_residual=None)
        seq = helpers.teacache_sequence(seed, 2 * num_steps + 3, drift, (1, N, C))
        dec = []
        with contextlib.redirect_stdout(io.StringIO()):
            for x in seq:
                before = counter[0]
                x5 = x.transpose(1, 2).reshape(1, C, T, Hh, W).clone()
                hy.teacache_forward(me, x5, torch.zeros(1), torch.zeros(1, 3, 4), torch.ones(1, 3), torch.zeros(1, 4),
                                    None, None, True)
                dec.append(counter[0] > before)
        res[f"hunyuan_{ci}"] = np.array(dec, np.uint8)
        print("teacache hunyuan", ci, "computed", sum(dec), "of", len(dec))

    # ---- Wan2.1: even / odd streams ---------------------------------------------------------------------------
    coeff = {  # main_wan21t2v.py:273-286 (keyed as the script keys them: model size, use_ret_steps)
        ("1.3B", True): [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
        ("14B", True): [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
        ("1.3B", False): [2.39676752e+03, -1.31110545e+03, 2.01331979e+02, -8.29855975e+00, 1.37887774e-01],
        ("14B", False): [-5784.54975374, 5449.50911966, -1811.16591783, 256.27178429, -13.02252404],
    }
    for ci, (seed, drift, thresh, steps, size, use_ret) in enumerate(helpers.TEACACHE_WAN_CASES):
        C, T, Hh, W = 8, 2, 4, 4
        N = T * Hh * W
        counter = [0]
        order = torch.arange(N)
        seq = helpers.teacache_sequence(seed, 2 * steps + 5, drift, (1, 6 * 16))
        it = iter(seq)
        me = types.SimpleNamespace(
            config=types.SimpleNamespace(patch_size=(1, 1, 1)),
            rope=lambda x: torch.ones(1, 1, N, 4),
            patch_embedding=lambda x: x,
            condition_embedder=None,
            hilbert_order=order, linear_to_hilbert=order, blocks=torch.nn.ModuleList([WanBlock(counter)]),
            scale_shift_table=torch.zeros(1, 2, C), norm_out=lambda h: h, proj_out=lambda h: h,
            enable_teacache=True, cnt=0, num_steps=2 * steps, teacache_thresh=thresh,
            accumulated_rel_l1_distance_even=0, accumulated_rel_l1_distance_odd=0, previous_e0_even=None,
            previous_e0_odd=None, previous_residual_even=None, previous_residual_odd=None, use_ref_steps=use_ret,
            coefficients=coeff[(size, use_ret)], ret_steps=(5 * 2 if use_ret else 1 * 2),
            cutoff_steps=(2 * steps if use_ret else 2 * steps - 2))

        def cond(t, enc, enc_img, _it=it):
            tp = next(_it)
            # use_ret_steps: modulated input = timestep_proj; else temb (both carry the sequence element here)
            return tp[:, :C].clone(), tp.clone(), enc, enc_img

        me.condition_embedder = cond
        dec = []
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(len(seq)):
                before = counter[0]
                wan.teacache_forward(me, torch.zeros(1, C, T, Hh, W), torch.zeros(1), torch.zeros(1, 3, 4), None, True,
                                     None)
                dec.append(counter[0] > before)
        res[f"wan_{ci}"] = np.array(dec, np.uint8)
        print("teacache wan", ci, size, "ret" if use_ret else "noret", "computed", sum(dec), "of", len(dec))
    # ---- CogVideoX: statistic on the time embedding, two cached residuals (main_cogvideox.py:45-196) -----------
    with contextlib.redirect_stdout(io.StringIO()):
        cog = _load_script("main_cogvideox")
        flux = _load_script("main_upflux")
        w22 = _load_script("main_wan22ti2v")   # main_wan22t2v.py:12 imports its teacache_forward from here

    class CogBlock(Block):
        def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb):
            self.counter[0] += 1
            return hidden_states * 1.01 + 0.1, encoder_hidden_states * 0.99 + 0.2

    for ci, (seed, drift, thresh, num_steps, model) in enumerate(helpers.TEACACHE_COG_CASES):
        C, F, Hh, W, NT = 8, 2, 4, 4, 3
        N = F * Hh * W
        counter = [0]
        order = torch.arange(N)
        seq = helpers.teacache_sequence(seed, 2 * num_steps + 3, drift, (1, 16))
        it = iter(seq)
        me = types.SimpleNamespace(
            config=types.SimpleNamespace(patch_size=1, patch_size_t=None, use_rotary_positional_embeddings=True),
            time_proj=lambda t: torch.zeros(1, 4), time_embedding=lambda t_emb, cond, _it=it: next(_it).clone(),
            ofs_embedding=None,
            patch_embed=lambda enc, hs: torch.cat([enc, hs.flatten(3).permute(0, 1, 3, 2).reshape(1, N, C)], 1),
            embedding_dropout=lambda x: x, hilbert_order=order, linear_to_hilbert=order,
            transformer_blocks=[CogBlock(counter)], attn_processors={},
            norm_final=lambda h: h, norm_out=lambda h, temb: h, proj_out=lambda h: h,
            enable_teacache=True, cnt=0, num_steps=num_steps, rel_l1_thresh=thresh,
            coefficients=cog.coefficients_dict[model], accumulated_rel_l1_distance=0, previous_modulated_input=None,
            previous_residual=None, previous_residual_encoder=None)
        dec = []
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(len(seq)):
                before = counter[0]
                cog.teacache_forward(me, torch.zeros(1, F, C, Hh, W), torch.zeros(1, NT, C), torch.zeros(1), None, None,
                                     (torch.ones(N, 4), torch.zeros(N, 4)), None, True)
                dec.append(counter[0] > before)
        res[f"cog_{ci}"] = np.array(dec, np.uint8)
        print("teacache cogvideox", ci, model, "computed", sum(dec), "of", len(dec))

    # ---- Flux (main_upflux.py:44-262): statistic on norm1's modulated input, controlnet samples required -------
    class FluxBlock(Block):
        def __init__(self, counter, it):
            super().__init__(counter)
            self.it = it

        def norm1(self, inp, emb=None):
            return next(self.it).clone(), None, None, None, None

        def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb, joint_attention_kwargs):
            self.counter[0] += 1
            return encoder_hidden_states, hidden_states * 1.01 + 0.1

    for ci, (seed, drift, thresh, num_steps) in enumerate(helpers.TEACACHE_FLUX_CASES):
        C, N, NT = 8, 32, 3
        counter = [0]
        order = torch.arange(N)
        seq = helpers.teacache_sequence(seed, 2 * num_steps + 3, drift, (1, N, C))
        it = iter(seq)
        me = types.SimpleNamespace(
            x_embedder=lambda x: x, time_text_embed=lambda *a: torch.zeros(1, 4), context_embedder=lambda e: e,
            pos_embed=lambda ids: (torch.ones(NT + N, 4), torch.zeros(NT + N, 4)),
            hilbert_order=order, linear_to_hilbert=order,
            transformer_blocks=[FluxBlock(counter, it)], single_transformer_blocks=[],
            norm_out=lambda h, t: h, proj_out=lambda h: h,
            enable_teacache=True, cnt=0, num_steps=num_steps, rel_l1_thresh=thresh,
            accumulated_rel_l1_distance=0, previous_modulated_input=None, previous_residual=None)
        dec = []
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(len(seq)):
                before = counter[0]
                flux.teacache_forward(me, torch.zeros(1, N, C), torch.zeros(1, NT, C), torch.zeros(1, 4), torch.zeros(1),
                                      torch.zeros(N, 3), torch.zeros(NT, 3), None, None, [torch.zeros(1, N, C)], None, True)
                dec.append(counter[0] > before)
        res[f"flux_{ci}"] = np.array(dec, np.uint8)
        print("teacache flux", ci, "computed", sum(dec), "of", len(dec))

    # ---- Wan2.2 T2V: two transformers over one schedule (main_wan22t2v.py:82-127, use_ret_steps) ---------------
    for ci, (seed, drift, thresh, steps, tsteps, size) in enumerate(helpers.TEACACHE_WAN22_CASES):
        C, T, Hh, W = 8, 2, 4, 4
        N = T * Hh * W
        order = torch.arange(N)
        seq = helpers.teacache_sequence(seed, 2 * (2 * steps) + 6, drift, (1, 6 * 16))
        it = iter(seq)
        counters = [[0], [0]]

        def make(counter, cnt0, num, ret, cutoff):
            def cond(t, enc, enc_img, timestep_seq_len=None, _it=it):
                tp = next(_it)
                return tp[:, :C].clone(), tp.clone(), enc, enc_img
            return types.SimpleNamespace(
                config=types.SimpleNamespace(patch_size=(1, 1, 1)), rope=lambda x: (torch.ones(1, N, 4), torch.ones(1, N, 4)),
                patch_embedding=lambda x: x, condition_embedder=cond, hilbert_order=order, linear_to_hilbert=order,
                blocks=torch.nn.ModuleList([WanBlock(counter)]), scale_shift_table=torch.zeros(1, 2, C),
                norm_out=lambda h: h, proj_out=lambda h: h, enable_teacache=True, cnt=cnt0, num_steps=num,
                teacache_thresh=thresh, accumulated_rel_l1_distance_even=0, accumulated_rel_l1_distance_odd=0,
                previous_e0_even=None, previous_e0_odd=None, previous_residual_even=None, previous_residual_odd=None,
                use_ref_steps=True, coefficients=coeff[(size, True)], ret_steps=ret, cutoff_steps=cutoff)

        m1 = make(counters[0], 0, 2 * tsteps, 3 * 2, 2 * tsteps)                       # :85-93, :113-114
        m2 = make(counters[1], 2 * tsteps, 2 * steps, 2 * tsteps + 1 * 2, 2 * steps)   # :96-105, :115-116
        dec = []
        with contextlib.redirect_stdout(io.StringIO()):
            for gen in range(2):               # two generations with the same pipe: the counters wrap (m2 wraps to 0)
                for call in range(2 * steps):
                    me, cn = (m1, counters[0]) if call < 2 * tsteps else (m2, counters[1])
                    before = cn[0]
                    w22.teacache_forward(me, torch.zeros(1, C, T, Hh, W), torch.zeros(1), torch.zeros(1, 3, 4), None,
                                         True, None)
                    dec.append(cn[0] > before)
        res[f"wan22_{ci}"] = np.array(dec, np.uint8)
        print("teacache wan2.2 pair", ci, size, "computed", sum(dec), "of", len(dec))
    np.savez_compressed(os.path.join(outdir, "teacache.npz"), **res)


def gilbert():
    """utils/jenga_gilbert.py vectors: permutations and block-neighbour matrices for small cuboids (full
    arrays) and the HunyuanVideo 32x45x80 latent (sha256 digests only)."""
    import contextlib
    import hashlib
    import io
    import matplotlib
    matplotlib.use("Agg")
    with contextlib.redirect_stdout(io.StringIO()):
        import utils.jenga_gilbert as ref
    outdir = os.path.dirname(os.path.abspath(__file__))
    res = {}
    shapes = [(4, 12, 16), (1, 32, 32), (2, 6, 10), (3, 5, 7), (5, 4, 3)]
    orders = [("w", "h", "t"), None, ("t", "h", "w")]
    for (t, h, w) in shapes:
        for ao in orders:
            tag = f"{t}x{h}x{w}_{''.join(ao) if ao else 'auto'}"
            with contextlib.redirect_stdout(io.StringIO()):
                l2h, h2l = ref.gilbert_mapping(t, h, w, axis_order=ao) if ao else ref.gilbert_mapping(t, h, w, axis_order=None)
                nb = ref.gilbert_block_neighbor_mapping(t, h, w, block_size=16, axis_order=ao)
            res[f"l2h_{tag}"] = np.asarray(l2h, np.int32)
            res[f"h2l_{tag}"] = np.asarray(h2l, np.int32)
            res[f"nbr16_{tag}"] = np.packbits(nb.numpy(), axis=-1)
            res[f"nbr16n_{tag}"] = np.array(nb.shape[0])
    # round 4: transpose_order (reference :290-346, :458-504)
    for (t, h, w) in [(2, 6, 10), (3, 5, 7), (4, 12, 16)]:
        for to in ([2, 1, 0], [1, 0, 2], [0, 2, 1]):
            with contextlib.redirect_stdout(io.StringIO()):
                l2h, h2l = ref.gilbert_mapping(t, h, w, transpose_order=to)
                nbt = ref.gilbert_block_neighbor_mapping(t, h, w, block_size=16, transpose_order=to)
                nb0 = ref.gilbert_block_neighbor_mapping(t, h, w, block_size=16)
            assert torch.equal(nbt, nb0)      # the reference ignores transpose_order in the neighbour function
            res[f"tr_l2h_{t}x{h}x{w}_{''.join(map(str, to))}"] = np.asarray(l2h, np.int32)
            res[f"tr_h2l_{t}x{h}x{w}_{''.join(map(str, to))}"] = np.asarray(h2l, np.int32)
    with contextlib.redirect_stdout(io.StringIO()):
        l2h, h2l = ref.gilbert_mapping(32, 45, 80, axis_order=("w", "h", "t"))
        nb = ref.gilbert_block_neighbor_mapping(32, 45, 80, axis_order=("w", "h", "t"))
    res["hunyuan_l2h_sha256"] = np.array(hashlib.sha256(np.asarray(l2h, np.int32).tobytes()).hexdigest())
    res["hunyuan_nbr_sha256"] = np.array(hashlib.sha256(nb.numpy().astype(np.uint8).tobytes()).hexdigest())
    res["hunyuan_nbr_rowsum"] = nb.sum(1).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(outdir, "gilbert.npz"), **res)
    print("gilbert.npz:", len(res), "arrays; hunyuan neighbour density", float(nb.float().mean()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "headline":
        headline()
    elif len(sys.argv) > 1 and sys.argv[1] == "processors_r4":
        _install_stubs()
        processors_round4()
    elif len(sys.argv) > 1 and sys.argv[1] == "gilbert":
        _install_stubs()
        gilbert()
    elif len(sys.argv) > 1 and sys.argv[1] == "processors_r2":
        _install_stubs()
        processors_round2()
    elif len(sys.argv) > 1 and sys.argv[1] == "dense":
        _install_stubs()
        dense()
    elif len(sys.argv) > 1 and sys.argv[1] == "processors_r3":
        _install_stubs()
        processors_round3()
    elif len(sys.argv) > 1 and sys.argv[1] == "teacache":
        _install_stubs()
        teacache()
    elif len(sys.argv) > 1 and sys.argv[1] == "processors":
        _install_stubs()
        processors()
    elif len(sys.argv) > 1 and sys.argv[1] == "ops":
        main()
    else:
        main()
        processors()
        processors_round2()
        processors_round3()
        teacache()
        gilbert()
