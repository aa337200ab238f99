"""CPU oracle for the Rectified-SpaAttn hot path (numpy + oracle/librsa_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  Each function cites the reference lines (under /root/reference) it follows.
The arithmetic contract for the mask statistics is written in oracle/rsa_oracle.c (C1..C8).

Pinned against the reference itself: tests/golden/*.npz were produced by tests/golden/make_golden.py, which
imports the reference's own Python on CPU in the build container; tests/test_oracle_golden.py checks this
file against those vectors (masks bit-exact, floats <= 1e-5, O within the fp16 kernel tolerance).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
BLOCK = 128


def build_oracle_lib(force: bool = False) -> str:
    so = os.path.join(_HERE, "librsa_oracle.so")
    src = os.path.join(_HERE, "rsa_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librsa_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build_oracle_lib())
        f32p = ctypes.POINTER(ctypes.c_float)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.orc_exp.restype = ctypes.c_float
        L.orc_exp.argtypes = [ctypes.c_float]
        L.orc_row_sum.restype = ctypes.c_float
        L.orc_row_sum.argtypes = [f32p, ctypes.c_int]
        L.orc_pool.restype = None
        L.orc_pool.argtypes = [f32p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, f32p]
        L.orc_dots.restype = None
        L.orc_dots.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p]
        L.orc_select_row.restype = None
        L.orc_select_row.argtypes = [f32p, f32p, f32p, f32p, u8p] + [ctypes.c_int] * 7 + [
            ctypes.c_float, ctypes.c_float, u8p, u8p, f32p, f32p, ctypes.POINTER(ctypes.c_int), f32p]
        _LIB = L
    return _LIB


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


# ------------------------------------------------------------------------------------------------------
# rounding helpers (exact emulation of bf16 / fp16 storage)
# ------------------------------------------------------------------------------------------------------
def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(np.shape(x))


def round_fp16(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


# ------------------------------------------------------------------------------------------------------
# layout: the four reference variants expressed as data
# ------------------------------------------------------------------------------------------------------
@dataclass
class Layout:
    """One self-attention call's geometry (all lengths in tokens / 128-token blocks).

    S              true sequence length (q and kv)
    NB_total       ceil(S / 128) -- the reference zero-pads to this (wan21 :299-302, cogvideo pad)
    NBv            visual blocks = sparse query blocks
    n_txt          valid text tokens scored individually ("attenable", hunyuan :315 / flux :139); 0 = no text tail
    kv_valid       kv columns < kv_valid are attended by the sparse kernel ("seqlens", hunyuan :314)
    pool_valid     rows >= pool_valid are zero when pooling K/V (hunyuan zeroes masked K/V in place :307-308)
    text_end_block text blocks [NBv, text_end_block) are kept by every row (hunyuan :277,:331)
    ffb            first_frame_blocks (wan21 :270-271)
    q_text_valid   text query rows [NBv*128, NBv*128+q_text_valid) get dense attention over kv [0, kv_text_valid)
                   (hunyuan :371-380); remaining rows up to S come out 0 (they only see zeroed K/V)
    """
    S: int
    NB_total: int
    NBv: int
    n_txt: int
    kv_valid: int
    pool_valid: int
    text_end_block: int
    ffb: int
    q_text_valid: int
    kv_text_valid: int
    name: str = ""

    @property
    def L(self) -> int:  # length of the sorted probability row
        return self.NBv + (1 if self.n_txt > 0 else 0)


def layout_hunyuan(S: int, num_true: int) -> Layout:
    """rectified_hunyuan_attn.py:313-332 (text tail padded to 256; num_true = attention_mask.sum())."""
    assert S % BLOCK == 0, "Hunyuan path has no padding branch (reference :326-327 with cu_seqlens given)"
    NB = S // BLOCK
    NBv = NB - 256 // BLOCK
    n_txt = 256 - (S - num_true)
    return Layout(S, NB, NBv, n_txt, num_true, num_true, (num_true + BLOCK - 1) // BLOCK, 0,
                  num_true - NBv * BLOCK, num_true, "hunyuan")


def layout_flux(S: int, text_length: int, s_k: Optional[int] = None) -> Layout:
    """rectified_flux_attn.py:307-320 (no KV zeroing; seqlens = cu_seqlens_kv[1])."""
    assert S % BLOCK == 0
    s_k = S if s_k is None else s_k
    NB = S // BLOCK
    NBv = NB - text_length // BLOCK
    return Layout(S, NB, NBv, text_length, s_k, S, (s_k + BLOCK - 1) // BLOCK, 0, S - NBv * BLOCK, s_k, "flux")


def layout_cogvideo(S: int, text_length: int) -> Layout:
    """rectified_cogvideo_attn.py:306-322 (pad to x128; all text blocks kept)."""
    NB = (S + BLOCK - 1) // BLOCK
    pad = NB * BLOCK - S
    NBv = NB - (text_length + pad) // BLOCK
    return Layout(S, NB, NBv, text_length, S, S, NB, 0, text_length, S, "cogvideo")


def layout_wan(S: int, first_frame_blocks: int = 0) -> Layout:
    """rectified_wan21_attn.py:297-313 (visual only, pad to x128, first-frame square)."""
    NB = (S + BLOCK - 1) // BLOCK
    return Layout(S, NB, NB, 0, S, S, NB, int(first_frame_blocks or 0), 0, S, "wan")


# ------------------------------------------------------------------------------------------------------
# statistics
# ------------------------------------------------------------------------------------------------------
def pool(x: np.ndarray, s_valid: int, nb: int, want_mad: bool):
    """x [S?, D] fp32 -> (mean [nb, D], mad [nb, D] | None).  Contract C2/C3."""
    x = np.ascontiguousarray(x[:s_valid], dtype=np.float32)
    D = x.shape[1]
    mean = np.empty((nb, D), np.float32)
    mad = np.empty((nb, D), np.float32) if want_mad else None
    _lib().orc_pool(_f32p(x), min(s_valid, x.shape[0]), nb, BLOCK, D, _f32p(mean),
                    _f32p(mad) if want_mad else None)
    return mean, mad


def dots(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """out[i, j] = fmaf-chain dot(a[i], b[j]).  Contract C4."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    _lib().orc_dots(_f32p(a), _f32p(b), a.shape[0], b.shape[0], a.shape[1], _f32p(out))
    return out


def orc_exp(x: float) -> float:
    return float(_lib().orc_exp(ctypes.c_float(x)))


def softmax_scale(D: int) -> np.float32:
    """head_dim ** -0.5 as the fp32 the reference multiplies with (hunyuan :208)."""
    return np.float32(float(D) ** -0.5)


@dataclass
class HeadStats:
    qbar: np.ndarray
    kbar: np.ndarray
    vbar: np.ndarray
    aq: np.ndarray
    ak: np.ndarray


def head_stats(q, k, v, lay: Layout) -> HeadStats:
    """Pooling of one (b, h) slice; q, k, v are [S, D] fp32 (hunyuan :189-194, :356; gapr_mask.py:19-23,:30)."""
    nq = lay.NBv
    qbar, aq = pool(q[: nq * BLOCK], min(lay.S, nq * BLOCK), nq, True)
    kbar, ak = pool(k[: nq * BLOCK], min(lay.pool_valid, nq * BLOCK), nq, True)
    vbar, _ = pool(v, lay.pool_valid, lay.NB_total, False)
    return HeadStats(qbar, kbar, vbar, aq, ak)


def select_head(q, k, v, lay: Layout, top_k: int, p: float, neighbor: Optional[np.ndarray],
                rows: Optional[Sequence[int]] = None, stats: Optional[HeadStats] = None):
    """Mask selection + rectification terms for one (b, h) slice.

    Returns dict with kept [nrows, NB_total] u8, unrel [nrows, NBv] u8, probs [nrows, L], w [nrows, L],
    n_needed [nrows], R [nrows], comp [nrows, D], rows.
    Follows _build_block_index_with_importance_optimized (hunyuan :171-280 / wan21 :171-273) and the inline
    rectification (hunyuan :348-357).
    """
    D = q.shape[1]
    st = stats or head_stats(q, k, v, lay)
    rows = list(range(lay.NBv)) if rows is None else list(rows)
    nr = len(rows)
    qb = st.qbar[rows]
    s_vis = dots(qb, st.kbar)
    eq = dots(st.aq[rows], st.kbar)
    ek = dots(qb, st.ak)
    if lay.n_txt > 0:
        ktxt = np.ascontiguousarray(k[lay.NBv * BLOCK: lay.NBv * BLOCK + lay.n_txt], np.float32)
        s_txt = dots(qb, ktxt)
    else:
        s_txt = np.zeros((nr, 1), np.float32)
    L = lay.L
    kept = np.zeros((nr, lay.NB_total), np.uint8)
    unrel = np.zeros((nr, lay.NBv), np.uint8)
    probs = np.zeros((nr, L), np.float32)
    w = np.zeros((nr, L), np.float32)
    n_needed = np.zeros(nr, np.int32)
    R = np.zeros(nr, np.float32)
    lib = _lib()
    scale = softmax_scale(D)
    thr = np.float32(p)
    nbr_u8 = None
    if neighbor is not None:
        nbr_u8 = np.ascontiguousarray(np.asarray(neighbor)[: lay.NBv, : lay.NBv], np.uint8)
    nn = ctypes.c_int(0)
    rr = ctypes.c_float(0)
    for a, i in enumerate(rows):
        lib.orc_select_row(_f32p(s_vis[a]), _f32p(s_txt[a]), _f32p(eq[a]), _f32p(ek[a]),
                           _u8p(nbr_u8[i]) if nbr_u8 is not None else None,
                           lay.NBv, lay.n_txt, lay.NB_total, lay.text_end_block, int(i), lay.ffb, int(top_k),
                           ctypes.c_float(thr), ctypes.c_float(scale), _u8p(kept[a]), _u8p(unrel[a]),
                           _f32p(probs[a]), _f32p(w[a]), ctypes.byref(nn), ctypes.byref(rr))
        n_needed[a] = nn.value
        R[a] = rr.value
    comp = (w.astype(np.float64) @ st.vbar[:L].astype(np.float64)).astype(np.float32)
    return dict(kept=kept, unrel=unrel, probs=probs, w=w, n_needed=n_needed, R=R, comp=comp, rows=rows,
                s_vis=s_vis, s_txt=s_txt, eq=eq, ek=ek, stats=st)


# ------------------------------------------------------------------------------------------------------
# attention proper
# ------------------------------------------------------------------------------------------------------
def _masked_attention_rows(qr, k, v, col_ok, sm_scale):
    """softmax(qr k^T * sm_scale) v restricted to columns where col_ok; float64 math."""
    s = (qr.astype(np.float64) @ k.astype(np.float64).T) * sm_scale
    s = np.where(col_ok[None, :], s, -np.inf)
    m = s.max(axis=1, keepdims=True)
    e = np.exp(s - m)
    return (e @ v.astype(np.float64)) / e.sum(axis=1, keepdims=True)


def sparse_attention_head(q, k, v, lay: Layout, kept_rows: np.ndarray, rows: Sequence[int]):
    """Block-sparse attention for q-blocks `rows` of one head (semantics of the reference Triton kernel,
    hunyuan :15-105: kept blocks only, kv columns >= seqlen masked, q rows >= S not produced)."""
    D = q.shape[1]
    sm = float(D) ** -0.5
    out = np.zeros((len(rows), BLOCK, D), np.float64)
    Spad = lay.NB_total * BLOCK
    kp = np.zeros((Spad, D), np.float32)
    vp = np.zeros((Spad, D), np.float32)
    kp[: k.shape[0]] = k
    vp[: v.shape[0]] = v
    col_tok = np.arange(Spad)
    for a, i in enumerate(rows):
        r0 = i * BLOCK
        nrow = max(0, min(BLOCK, lay.S - r0))
        if nrow == 0:
            continue
        col_ok = (np.repeat(kept_rows[a].astype(bool), BLOCK)) & (col_tok < lay.kv_valid)
        out[a, :nrow] = _masked_attention_rows(q[r0: r0 + nrow], kp, vp, col_ok, sm)
    return out


def dense_attention(q, k, v, kv_valid: Optional[int] = None, causal: bool = False):
    """Exact softmax attention for one head; fullattn(mode='torch'/'vanilla') semantics (attn.py:101-149).  causal: key j
    visible to row i iff j <= i + (keys - rows) (flash-attn's alignment, attn.py:108-116; == the tril of :105 / :129-133
    when there are as many keys as rows); rows that see no key give zeros."""
    D = q.shape[1]
    n = k.shape[0] if kv_valid is None else kv_valid
    ok = np.arange(k.shape[0]) < n
    if not causal:
        return _masked_attention_rows(q, k, v, ok, float(D) ** -0.5)
    off = n - q.shape[0]
    vis = ok[None, :] & (np.arange(k.shape[0])[None, :] <= np.arange(q.shape[0])[:, None] + off)
    s = (q.astype(np.float64) @ k.astype(np.float64).T) * float(D) ** -0.5
    s = np.where(vis, s, -np.inf)
    m = np.where(vis.any(1, keepdims=True), s.max(axis=1, keepdims=True), 0.0)
    e = np.where(vis, np.exp(s - m), 0.0)
    den = e.sum(axis=1, keepdims=True)
    return np.where(den > 0, (e @ v.astype(np.float64)) / np.where(den > 0, den, 1.0), 0.0)


def rectified_attention(q, k, v, lay: Layout, top_k: int, p: float, neighbor=None, want_parts: bool = False):
    """Whole operator: q, k, v [B, H, S, D] fp32 (bf16/fp16-representable values) -> [B, S, H*D] fp32.

    Follows block_sparse_attention_combined (hunyuan :283-389, flux :282-376, cogvideo :282-378,
    wan21 :276-357), with all mask statistics in the fp32 contract.
    """
    B, H, S, D = q.shape
    assert S == lay.S
    out = np.zeros((B, S, H, D), np.float32)
    parts = []
    nvis_tok = lay.NBv * BLOCK
    for b in range(B):
        for h in range(H):
            qq, kk, vv = q[b, h], k[b, h], v[b, h]
            if lay.pool_valid < S:  # hunyuan zeroes masked K/V rows (:307-308)
                kk = kk.copy()
                vv = vv.copy()
                kk[lay.pool_valid:] = 0
                vv[lay.pool_valid:] = 0
            sel = select_head(qq, kk, vv, lay, top_k, p, neighbor)
            o = sparse_attention_head(qq, kk, vv, lay, sel["kept"], sel["rows"])
            o = o * sel["R"][:, None, None].astype(np.float64) + sel["comp"][:, None, :].astype(np.float64)
            o = o.reshape(-1, D)[: min(S, nvis_tok)]
            out[b, : o.shape[0], h] = o
            if lay.q_text_valid > 0:
                r0 = nvis_tok
                ot = dense_attention(qq[r0: r0 + lay.q_text_valid], kk, vv, lay.kv_text_valid)
                out[b, r0: r0 + lay.q_text_valid, h] = ot
            if want_parts:
                parts.append(sel)
    res = out.reshape(B, S, H * D)
    return (res, parts) if want_parts else res


# ---------------------------------------------------------------------------------------------------------
# fp8 operands (BASELINE config "fp8 Q/K/V on CDNA4 fp8 MFMA").  The reference has no fp8 code; its rule
# "operands rounded to the input dtype, fp32 statistics" (rectified_hunyuan_attn.py:61-62, :97) is applied to
# e4m3 (OCP e4m3fn: 4 exponent bits, bias 7, 3 mantissa bits, max 448, no infinity).
# ---------------------------------------------------------------------------------------------------------
E4M3_MAX = np.float32(448.0)


def quantize_e4m3(y: np.ndarray) -> np.ndarray:
    """fp32 -> e4m3 bytes: clamp to +-448, round to nearest even, subnormals (multiples of 2^-9) kept."""
    y = np.clip(np.asarray(y, np.float32), -E4M3_MAX, E4M3_MAX)
    sign = (np.signbit(y).astype(np.uint8)) << 7
    a = np.abs(y).astype(np.float64)
    m, e = np.frexp(a)                      # a = m * 2^e, m in [0.5, 1)
    e = e - 1                               # a = (2m) * 2^e, 2m in [1, 2)
    sub = e < -6                            # below the smallest normal 2^-6: fixed step 2^-9
    step = np.where(sub, 2.0 ** -9, np.exp2(np.maximum(e, -6).astype(np.float64) - 3))
    qv = np.rint(a / step)                  # np.rint = round half to even; a/step is exact (power-of-two step)
    val = qv * step                         # representable magnitude (may have carried to the next binade)
    m2, e2 = np.frexp(val)
    e2 = e2 - 1
    normal = val >= 2.0 ** -6
    mant = np.where(normal, np.rint((m2 * 2 - 1) * 8), np.rint(val * 2.0 ** 9)).astype(np.int64)
    expo = np.where(normal, e2 + 7, 0).astype(np.int64)
    byte = ((expo << 3) | mant).astype(np.uint8)
    byte = np.where(val == 0, 0, byte).astype(np.uint8)
    return (byte | sign).astype(np.uint8)


def dequantize_e4m3(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, np.uint8)
    sign = np.where(b & 0x80, -1.0, 1.0)
    expo = ((b >> 3) & 0xF).astype(np.int64)
    mant = (b & 7).astype(np.float64)
    val = np.where(expo == 0, mant * 2.0 ** -9, (1 + mant / 8) * np.exp2(expo.astype(np.float64) - 7))
    return (sign * val).astype(np.float32)


def fp8_kslot_key(p: np.ndarray) -> np.ndarray:
    """Key (0..63 inside a 64-key tile) stored at byte p of a V^T row (layout of rsa_fp8.hip)."""
    p = np.asarray(p)
    h, j = p >> 5, p & 31
    return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h


def fp8_qk_const(D: int) -> np.float32:
    """sm_scale * log2(e) as the library rounds it to fp32 (rsa_fp8.hip: (float)((1 / sqrt(D)) * 1.44269504))."""
    return np.float32((1.0 / np.sqrt(float(D))) * 1.44269504)


def fp8_block_exponent(amax: np.float32) -> int:
    """E8M0 byte (127 + e) of a block's power-of-two scale: the smallest e with amax * 2^-e <= 448, clamped to +-120;
    127 for an all-zero block (rsa_fp8_emit.h::e8m0_of_amax: exponent field and one mantissa compare, no division)."""
    a = np.float32(amax)
    if not a > 0:
        return 127
    bits = int(np.array(a, np.float32).view(np.uint32))
    field, frac = (bits >> 23) & 0xFF, bits & 0x7FFFFF
    if field == 0:                               # subnormal maximum: far below any scale that matters
        return 127 - 120
    e = (field - 126) - 9 + (1 if frac > 0x600000 else 0)   # amax = m 2^x, m in [0.5, 1): m > 0.875 needs one more
    return 127 + max(-120, min(120, e))


def fp8_kmean(k, valid: int):
    """"Smooth K" vector of one head (rsa_fp8.hip::kmean_sample_kernel): the mean of up to 8 evenly spaced 128-row blocks
    of K -- block j of the nb = max(valid // 128, 1) full blocks for j = floor(i nb / n), i < n = min(nb, 8) -- each block's mean in the pooling
    pass's arithmetic (contract C2: rows >= valid count as zero, divide by 128), the block means added in order, divided
    by n; fp32 throughout.  Any vector works in exact arithmetic (q.(k - mu) shifts every score of a query row alike); this
    one needs 8 blocks of K instead of a pass over it, which is what lets K1 quantise K in the pass that pools it."""
    kk = np.asarray(k, np.float32)
    D = kk.shape[1]
    if valid <= 0:
        return np.zeros(D, np.float32)
    nb = max(valid // BLOCK, 1)                  # full blocks only (a lone partial block 0 when there is none)
    n = min(nb, 8)
    acc = np.zeros(D, np.float32)
    for i in range(n):
        j = (i * nb) // n
        rows = kk[j * BLOCK: min((j + 1) * BLOCK, valid)]
        m, _ = pool(rows, rows.shape[0], 1, False)
        acc = (acc + m[0]).astype(np.float32)
    return (acc / np.float32(n)).astype(np.float32)


def fp8_block_images(x, valid: int, pad: int, pre: str, kmean=None):
    """One tensor of one head [S, D] fp32 -> (image [pad, D] uint8, E8M0 bytes [pad / 128]) in the block-scaled format:
    y = x * qk_const (pre = "q": one fp32 multiply), x - mu (pre = "k", one fp32 subtract) or x ("v"); per 128-row block
    amax over its valid rows, power-of-two scale 2^e (fp8_block_exponent), bytes = e4m3(y * 2^-e) round-to-nearest-even;
    rows >= valid are zero."""
    D = x.shape[1]
    nblk = pad // BLOCK
    img = np.zeros((pad, D), np.uint8)
    ex = np.full(nblk, 127, np.uint8)
    xv = np.asarray(x[:valid], np.float32)
    if pre == "q":
        y = (xv * fp8_qk_const(D)).astype(np.float32)
    elif pre == "k" and kmean is not None:
        y = (xv - kmean[None, :]).astype(np.float32)
    else:
        y = xv
    for j in range(nblk):
        blk = y[j * BLOCK: (j + 1) * BLOCK]
        if blk.size == 0:
            continue
        eb = fp8_block_exponent(np.float32(np.max(np.abs(blk))))
        ex[j] = eb
        img[j * BLOCK: j * BLOCK + blk.shape[0]] = quantize_e4m3(np.ldexp(blk, -(eb - 127)).astype(np.float32))
    return img, ex


def _v8t_from_image(v8):
    BH, SP, D = v8.shape
    v8 = v8.reshape(BH, SP // 64, 64, D)
    return np.ascontiguousarray(v8[:, :, fp8_kslot_key(np.arange(64)), :].transpose(0, 1, 3, 2))


def _fp8_operands_rows(xs, valid, pads, smooth_k: bool):
    """xs: three [BH, S_i, D] fp32 arrays -> dict(q8, k8, v8t, exps [BH, nblk] uint32 = eq | ek << 8 | ev << 16, kmean)."""
    BH, D = xs[0].shape[0], xs[0].shape[2]
    nblk = max(pads) // BLOCK
    exps = np.full((BH, nblk), 127 | (127 << 8) | (127 << 16), np.uint32)
    kmean = np.zeros((BH, D), np.float32)
    imgs = [np.zeros((BH, pads[i], D), np.uint8) for i in range(3)]
    for bh in range(BH):
        if smooth_k:
            kmean[bh] = fp8_kmean(xs[1][bh], valid[1])
        word = np.zeros(nblk, np.uint32)
        for i, pre in enumerate("qkv"):
            img, ex = fp8_block_images(xs[i][bh], valid[i], pads[i], pre, kmean[bh] if (smooth_k and i == 1) else None)
            imgs[i][bh] = img
            full = np.full(nblk, 127, np.uint32)
            full[: ex.shape[0]] = ex
            word |= full << np.uint32(8 * i)
        exps[bh] = word
    return dict(exps=exps, kmean=kmean, q8=imgs[0], k8=imgs[1], v8t=_v8t_from_image(imgs[2]), v8=imgs[2])


def fp8_operands(q, k, v, lay: Layout, smooth_k: bool = True):
    """The e4m3 operands of the fp8 K5 exactly as rsa_pool_stats_fp8 (fused into K1) and rsa_quantize_fp8 (stand-alone)
    write them -- the two are bit-identical: block-scaled images (fp8_block_images) of Q * qk_const, K - mu, V, one E8M0
    byte per tensor and 128-row block, mu = fp8_kmean (smooth_k = False: mu = 0, kept for the accuracy study).
    q, k, v: [B, H, S, D] fp32 -> dict(exps [BH, NB_total] uint32, kmean [BH, D], q8 / k8 [BH, S_pad, D],
    v8t [BH, S_pad / 64, D, 64])."""
    B, H, S, D = q.shape
    BH, SP = B * H, lay.NB_total * BLOCK
    assert lay.pool_valid >= max(lay.kv_valid, lay.kv_text_valid)
    valid = (S, lay.pool_valid, lay.pool_valid)  # the rows the pooling pass counts
    xs = [np.asarray(x, np.float32).reshape(BH, S, D) for x in (q, k, v)]
    return _fp8_operands_rows(xs, valid, (SP, SP, SP), smooth_k)


def _fp8_dequant(ops, bh, i, rows, D):
    """Dequantised rows [0, rows) of tensor i (0 q in ORIGINAL units, 1 k minus mu, 2 v) of head bh, float64."""
    if i == 2:
        img = ops["v8"][bh]
    else:
        img = ops["q8" if i == 0 else "k8"][bh]
    ex = ((ops["exps"][bh] >> np.uint32(8 * i)) & np.uint32(0xFF)).astype(np.int64) - 127
    val = dequantize_e4m3(img).astype(np.float64) * np.exp2(np.repeat(ex, BLOCK)[: img.shape[0]].astype(np.float64))[:, None]
    if i == 0:
        val = val / float(fp8_qk_const(D))
    return val[:rows]


def dense_attention_fp8(q, k, v, q_split: Optional[int] = None, kv_split: Optional[int] = None, causal: bool = False,
                        qk: str = "e4m3"):
    """Dense attention of ONE head on e4m3 operands as rsa_dense_fwd_fp8 quantises them (block-scaled images over all Sq /
    Sk rows, K minus fp8_kmean over its Sk rows), two-segment semantics of attn.py:107-120.  q [Sq, D], k/v [Sk, D] fp32
    -> [Sq, D].  (With two segments the shift q.mu is still one constant per query row, so both softmaxes are unchanged.)"""
    Sq, D = q.shape
    Sk = k.shape[0]
    pad = lambda n: (n + BLOCK - 1) // BLOCK * BLOCK
    ops = _fp8_operands_rows([q[None], k[None], v[None]], (Sq, Sk, Sk), (pad(Sq), pad(Sk), pad(Sk)), True)
    qd, kd, vd = (_fp8_dequant(ops, 0, i, n, D) for i, n in ((0, Sq), (1, Sk), (2, Sk)))
    if qk == "2byte":    # the pv form (rsa_dense_fwd_fp8pv): the scores from q and k as they are, only V from its e4m3 image
        qd, kd = np.asarray(q, np.float64), np.asarray(k, np.float64)
    else:
        assert qk == "e4m3", qk
    q_split = Sq if q_split is None else q_split
    kv_split = Sk if kv_split is None else kv_split
    out = np.zeros((Sq, D), np.float64)
    sm = float(D) ** -0.5
    cols = np.arange(Sk)
    if causal:   # per segment, flash-attn's bottom-right alignment (dense_attention)
        if q_split > 0 and kv_split > 0:
            out[:q_split] = dense_attention(qd[:q_split], kd[:kv_split], vd[:kv_split], causal=True)
        if q_split < Sq and kv_split < Sk:
            out[q_split:] = dense_attention(qd[q_split:], kd[kv_split:], vd[kv_split:], causal=True)
        return out
    if q_split > 0:
        out[:q_split] = _masked_attention_rows(qd[:q_split], kd, vd, cols < kv_split, sm)
    if q_split < Sq:
        out[q_split:] = _masked_attention_rows(qd[q_split:], kd, vd, cols >= kv_split, sm)
    return out


def fp8_dequantized_qkv(q, k, v, lay: Layout, smooth_k: bool = True):
    """The values the fp8 K5 multiplies, cropped back to [B, H, S, D] fp32: q in its original units (the image holds
    q * qk_const), k minus its mu, v."""
    B, H, S, D = q.shape
    ops = fp8_operands(q, k, v, lay, smooth_k)
    res = [np.stack([_fp8_dequant(ops, bh, i, S, D) for bh in range(B * H)]).reshape(B, H, S, D).astype(np.float32)
           for i in range(3)]
    return res[0], res[1], res[2], ops


# The e4m3 kernel's P ("code map", rsa_attn_fp8_kernel.hip PMap<true>): with x = log2(e) sm_scale (q . k) - m + 6.5 the byte
# stored for P is code = rint(8 x + 56) clamped to [0, 255] (round half to even), i.e. P = 2^e (1 + m3 / 8) for
# code = 8 (e + 7) + m3 -- the format's own piecewise-linear log2 instead of an exponential followed by a rounding.  m is
# the row's running reference: the first finite tile maximum, afterwards moved to the current tile maximum whenever SOME
# row of the same wave (32 consecutive query rows) sees a tile maximum more than 2 above its reference; O and l are
# rescaled by 2^(m_old - m_new) then.  Tiles are 64 keys, in the order the kernel walks the kept list.
PCODE_U, PCODE_BIAS, PCODE_OFFSET, PCODE_THRESH = 8.0, 56.0, 6.5, 2.0
_E4M3_CODE_VALUE = None


def e4m3_code_values() -> np.ndarray:
    global _E4M3_CODE_VALUE
    if _E4M3_CODE_VALUE is None:
        _E4M3_CODE_VALUE = dequantize_e4m3(np.arange(128, dtype=np.uint8)).astype(np.float64)
    return _E4M3_CODE_VALUE


def attention_rows_pcode(qr, k, v, tile_keys: Sequence[int], lo, hi, sm_scale, wave: int = 32):
    """softmax(qr k^T sm_scale) v over the 64-key tiles starting at `tile_keys` (in that order), keys outside
    [lo[row], hi[row]) masked, P through the code map above with the kernel's deferred reference.  fp64 accumulation.
    Returns (O, l) un-normalised in V's units plus the final reference m: [rows, D], [rows], [rows]."""
    n, D = qr.shape
    val = e4m3_code_values()
    lo = np.broadcast_to(np.asarray(lo), (n,))
    hi = np.broadcast_to(np.asarray(hi), (n,))
    c = float(sm_scale) * math.log2(math.e)
    q64 = qr.astype(np.float64)
    O = np.zeros((n, D), np.float64)
    l = np.zeros(n, np.float64)
    m = np.full(n, -np.inf)
    npad = (-n) % wave
    for key0 in tile_keys:
        kk = np.arange(key0, key0 + 64)
        kt = np.zeros((64, D), np.float64)
        vt = np.zeros((64, D), np.float64)
        ok = kk < k.shape[0]
        kt[ok] = k[kk[ok]]
        vt[ok] = v[kk[ok]]
        s = (q64 @ kt.T) * c
        s = np.where((kk[None, :] >= lo[:, None]) & (kk[None, :] < hi[:, None]), s, -np.inf)
        mx = s.max(axis=1)
        grow = np.where(np.isneginf(m), mx > -np.inf, mx > m + PCODE_THRESH)
        g = np.concatenate([grow, np.zeros(npad, bool)]).reshape(-1, wave).any(axis=1)
        g = np.repeat(g, wave)[:n]
        m_new = np.where(g, np.maximum(m, mx), m)
        with np.errstate(invalid="ignore"):
            alpha = np.where(np.isneginf(m_new) | (m_new == m), 1.0, np.exp2(np.where(np.isneginf(m), -np.inf, m) - np.where(np.isneginf(m_new), 0.0, m_new)))
        O *= alpha[:, None]
        l *= alpha
        m = m_new
        m_eff = np.where(np.isneginf(m), 0.0, m)
        with np.errstate(invalid="ignore"):
            code = np.rint(PCODE_U * (s - m_eff[:, None] + PCODE_OFFSET) + PCODE_BIAS)
        code = np.clip(np.where(np.isnan(code), 0, code), 0, 127).astype(np.int64)
        P = val[code]
        O += P @ vt
        l += P.sum(axis=1)
    return O, l, m


def sparse_attention_head_pcode(q, k, v, lay: Layout, kept_rows: np.ndarray, rows: Sequence[int]):
    """sparse_attention_head with the e4m3 kernel's P (same arguments; q, k, v = the dequantised images)."""
    D = q.shape[1]
    sm = float(D) ** -0.5
    out = np.zeros((len(rows), BLOCK, D), np.float64)
    for a, i in enumerate(rows):
        r0 = i * BLOCK
        nrow = max(0, min(BLOCK, lay.S - r0))
        if nrow == 0:
            continue
        qb = np.zeros((BLOCK, D), np.float32)          # the kernel always runs four full waves (rows past S: zero queries)
        qb[:nrow] = q[r0: r0 + nrow]
        tiles = [int(b) * BLOCK + t for b in np.nonzero(kept_rows[a])[0] for t in (0, 64)]
        O, l, _ = attention_rows_pcode(qb, k, v, tiles, 0, lay.kv_valid, sm)
        out[a, :nrow] = (O / np.where(l > 0, l, 1.0)[:, None])[:nrow]
    return out


def dense_rows_pcode(qr, k, v, kv_valid: int, row0: int):
    """Dense text rows with the e4m3 kernel's P: query rows `qr` start at row `row0` of their 128-row block grid (the wave
    grouping follows the block), keys [0, kv_valid), one workgroup per query block (no split-KV: < 32 key blocks)."""
    D = qr.shape[1]
    assert (kv_valid + BLOCK - 1) // BLOCK < 32, "split-KV text rows are not modelled here"
    out = np.zeros((qr.shape[0], D), np.float64)
    tiles = list(range(0, kv_valid, 64))
    first = row0
    while first < row0 + qr.shape[0]:
        blk0 = (first // BLOCK) * BLOCK
        n_here = min(blk0 + BLOCK, row0 + qr.shape[0]) - first
        qb = np.zeros((BLOCK, D), np.float32)
        qb[first - blk0: first - blk0 + n_here] = qr[first - row0: first - row0 + n_here]
        O, l, _ = attention_rows_pcode(qb, k, v, tiles, 0, kv_valid, float(D) ** -0.5)
        res = O / np.where(l > 0, l, 1.0)[:, None]
        out[first - row0: first - row0 + n_here] = res[first - blk0: first - blk0 + n_here]
        first += n_here
    return out


def rectified_attention_fp8(q, k, v, lay: Layout, top_k: int, p: float, neighbor=None, want_parts: bool = False,
                            smooth_k: bool = True, p_form: str = "exact", qk: str = "e4m3"):
    """Operator with fp8 K5 operands: mask statistics, R and comp from the 2-byte inputs (unchanged contract), the
    sparse / text-row attention itself on the dequantised e4m3 values.  p_form = "exact": P kept in fp64 (what the e4m3
    rounding of P is measured against: the stated fp8 tolerance); "code": P exactly as the kernel forms it (code map and
    deferred reference above) -- the kernel then differs only by fp32 accumulation and by codes on a rounding boundary.
    qk = "2byte": the pv form's scores (q, k as given; its kernel additionally rounds q * sm_scale * log2(e) * 8 to the 2-byte type,
    as the 2-byte kernels round their scaled q)."""
    assert p_form in ("exact", "code") and qk in ("e4m3", "2byte")
    B, H, S, D = q.shape
    q8, k8, v8, ops = fp8_dequantized_qkv(q, k, v, lay, smooth_k)
    if qk == "2byte":      # the "pv" form (rsa_block_sparse_fwd_fp8pv): the scores from the 2-byte q and k themselves, e4m3 only P and V
        q8, k8 = q, k
    out = np.zeros((B, S, H, D), np.float32)
    parts = []
    nvis_tok = lay.NBv * BLOCK
    for b in range(B):
        for h in range(H):
            qq, kk, vv = q[b, h], k[b, h], v[b, h]
            if lay.pool_valid < S:
                kk = kk.copy()
                vv = vv.copy()
                kk[lay.pool_valid:] = 0
                vv[lay.pool_valid:] = 0
            sel = select_head(qq, kk, vv, lay, top_k, p, neighbor)
            head = sparse_attention_head_pcode if p_form == "code" else sparse_attention_head
            o = head(q8[b, h], k8[b, h], v8[b, h], lay, sel["kept"], sel["rows"])
            o = o * sel["R"][:, None, None].astype(np.float64) + sel["comp"][:, None, :].astype(np.float64)
            o = o.reshape(-1, D)[: min(S, nvis_tok)]
            out[b, : o.shape[0], h] = o
            if lay.q_text_valid > 0:
                r0 = nvis_tok
                if p_form == "code":
                    ot = dense_rows_pcode(q8[b, h][r0: r0 + lay.q_text_valid], k8[b, h], v8[b, h], lay.kv_text_valid, r0)
                else:
                    ot = dense_attention(q8[b, h][r0: r0 + lay.q_text_valid], k8[b, h], v8[b, h], lay.kv_text_valid)
                out[b, r0: r0 + lay.q_text_valid, h] = ot
            if want_parts:
                parts.append(sel)
    res = out.reshape(B, S, H * D)
    return (res, parts, ops) if want_parts else res


def pack_bits(mask_u8: np.ndarray) -> np.ndarray:
    """[..., N] 0/1 -> [..., ceil(N/32)] uint32, bit j%32 of word j//32 (the library's bitmask format)."""
    n = mask_u8.shape[-1]
    nw = (n + 31) // 32
    padded = np.zeros(mask_u8.shape[:-1] + (nw * 32,), np.uint8)
    padded[..., :n] = mask_u8 != 0
    bits = padded.reshape(mask_u8.shape[:-1] + (nw, 32)).astype(np.uint32)
    return (bits << np.arange(32, dtype=np.uint32)).sum(axis=-1).astype(np.uint32)


def unpack_bits(words: np.ndarray, n: int) -> np.ndarray:
    b = ((words[..., :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8)
    return b.reshape(words.shape[:-1] + (-1,))[..., :n]
