"""Host side of the rectified block-sparse attention path: layout bookkeeping + calls into librsa_hip.so.

PyTorch is used only for device memory and the current HIP stream.  The four reference operator variants
(rectified_{hunyuan,flux,cogvideo,wan21}_attn.py::block_sparse_attention_combined) differ only in the
numbers of `LayoutSpec`; one code path serves them all.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import BLOCK, BUFFER_NAMES, RsaBuffers, RsaFp8Operands, RsaLayout, RsaOut4, RsaTensor4


@dataclass
class LayoutSpec:
    """Geometry of one call; see include/rsa.h::rsa_layout for the field meanings."""
    S: int
    NB_total: int
    NBv: int
    n_txt: int
    kv_valid: int
    pool_valid: int
    text_end_block: int
    first_frame_blocks: int
    q_text_valid: int
    kv_text_valid: int

    # -- the reference's variants (file:line under the reference root) ---------------------------------
    @staticmethod
    def hunyuan(S: int, num_true: int) -> "LayoutSpec":
        """rectified_hunyuan_attn.py:313-332: text tail padded to 256, num_true = attention_mask.sum()."""
        if S % BLOCK:
            raise ValueError("HunyuanVideo layout needs S % 128 == 0 (the reference reshapes without padding)")
        NB = S // BLOCK
        NBv = NB - 256 // BLOCK
        n_txt = 256 - (S - num_true)
        if NBv < 0 or n_txt <= 0 or n_txt > 256:
            # reference: attenable == 0 makes scores[..., :-0] empty and crashes
            raise ValueError(f"HunyuanVideo layout needs 1..256 valid text tokens, got {n_txt}")
        return LayoutSpec(S, NB, NBv, n_txt, num_true, num_true, (num_true + BLOCK - 1) // BLOCK, 0,
                          num_true - NBv * BLOCK, num_true)

    @staticmethod
    def flux(S: int, text_length: int, s_k: Optional[int] = None) -> "LayoutSpec":
        """rectified_flux_attn.py:307-320."""
        if S % BLOCK:
            raise ValueError("Flux layout needs S % 128 == 0")
        s_k = S if s_k is None else int(s_k)
        NB = S // BLOCK
        NBv = NB - text_length // BLOCK
        return LayoutSpec(S, NB, NBv, text_length, s_k, S, (s_k + BLOCK - 1) // BLOCK, 0, S - NBv * BLOCK, s_k)

    @staticmethod
    def cogvideo(S: int, text_length: int, s_k: Optional[int] = None) -> "LayoutSpec":
        """rectified_cogvideo_attn.py:306-322 (zero-pad to x128, every text block kept)."""
        NB = (S + BLOCK - 1) // BLOCK
        pad = NB * BLOCK - S
        NBv = NB - (text_length + pad) // BLOCK
        if NBv < 0 or text_length > S - NBv * BLOCK:
            # the reference cuts the text rows at normal_blocks * 128 and hands flash-attn cu_seqlens_q = [0, text_length]
            # (rectified_cogvideo_attn.py:318-320,:359-366): with fewer rows than text_length behind the cut that reads past the
            # tensor -- the layout only exists when the visual tokens fill whole blocks
            raise ValueError(f"CogVideoX layout: the {S - text_length} visual tokens of S = {S}, text_length = {text_length} must be a "
                             f"multiple of {BLOCK}")
        s_k = S if s_k is None else int(s_k)
        return LayoutSpec(S, NB, NBv, text_length, S, S, NB, 0, text_length, s_k)

    @staticmethod
    def wan(S: int, first_frame_blocks: Optional[int] = 0) -> "LayoutSpec":
        """rectified_wan21_attn.py:297-313 (visual only)."""
        NB = (S + BLOCK - 1) // BLOCK
        return LayoutSpec(S, NB, NB, 0, S, S, NB, int(first_frame_blocks or 0), 0, S)

    @property
    def L(self) -> int:
        return self.NBv + (1 if self.n_txt > 0 else 0)

    def to_c(self, B: int, H: int, D: int, dtype: torch.dtype) -> RsaLayout:
        return RsaLayout(B, H, D, self.S, self.NB_total, self.NBv, self.n_txt, self.kv_valid, self.pool_valid,
                         self.text_end_block, self.first_frame_blocks, self.q_text_valid, self.kv_text_valid,
                         dtype_code(dtype))


def check_head_dim(D: int) -> None:
    """The reference asserts head_dim in {16, 32, 64, 128} (rectified_hunyuan_attn.py:119-121); the gfx950 kernels are
    built for 64 and 128 (every model the reference ships: 128, CogVideoX 64).  16 / 32 reach them zero-padded
    (pad_small_head_dim) through rectified_attention / dense_attention; a StagedCall takes 64 / 128 only."""
    assert D in (16, 32, 64, 128), "head_dim must be in {16, 32, 64, 128}"  # reference :121
    assert D in (64, 128), (f"head_dim {D}: the MI355X kernels are built for head_dim 64 and 128; pass head_dim 16 / 32 "
                            "through rectified_attention / dense_attention, which zero-pad them")


# head dims the reference's assert admits but no kernel is built for (no reference pipeline uses them) -> the padded head
# dim that serves them EXACTLY: D' = 4 D, so (D') ** -0.5 is exactly half of D ** -0.5 (also after rounding to fp32) and
# doubling Q -- exact in a binary format -- restores every score: pooled scores, GAPR comparison, softmax statistics and
# hence the block mask are bit-identical to the native-D contract; zero columns add exact zeros to every dot product,
# mean and deviation.  Costs 4x the MFMA work of a native kernel: a compatibility path, not a fast one.
_PAD_HEAD_DIM = {16: 64, 32: 128}


def pad_small_head_dim(q, k, v):
    """[B, H, S, D] with D in {16, 32} -> (2 q | 0, k | 0, v | 0) with D' = 4 D columns (see _PAD_HEAD_DIM).

    The doubling is exact unless it overflows: bfloat16 has fp32's range, float16 does not -- a float16 |q| above 32 752 would
    become inf where the native head-dim contract is finite, so that case is refused (one device reduction + host sync, on this
    compatibility path only).  With return_parts the statistics come back in the PADDED problem's shape and scale: qbar / aq /
    kbar / ak / vbar / comp have 4 D columns (the last 3 D zero), qbar and aq are those of 2 q; scores, probabilities, GAPR
    bytes, R, w, the mask and the lists are the native problem's, bit for bit; the fp8 images are those of the padded tensors."""
    D = q.shape[-1]
    if q.dtype == torch.float16 and bool((q.abs() > 32752).any()):
        raise _lib.RsaError(f"head_dim {D} runs zero-padded with Q doubled, which overflows float16 for |q| > 32752; "
                            "use bfloat16 inputs or scale Q down")
    pad = (0, _PAD_HEAD_DIM[D] - D)
    return torch.nn.functional.pad(q * 2, pad), torch.nn.functional.pad(k, pad), torch.nn.functional.pad(v, pad)


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return _lib.RSA_BF16
    if dtype == torch.float16:
        return _lib.RSA_FP16
    raise AssertionError(f"rectified_spaattn_amd supports bfloat16 / float16 device tensors, got {dtype}")


def _require_device(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.RsaError("this operator runs only on the HIP extension (device tensors); "
                                "CPU tensors are supported by fullattn(mode='torch'|'vanilla') alone")


def _as_bhsd(t: torch.Tensor) -> torch.Tensor:
    """Accept any [B,H,S,D] view whose head dim is contiguous and whose strides keep 16-byte row chunks."""
    if t.stride(-1) != 1 or any(s % 8 for s in t.stride()[:-1]) or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _k_for_pv(k: torch.Tensor) -> torch.Tensor:
    """The pv kernel addresses a key row as a 32-bit byte offset from its head's base (the library refuses longer spans with
    RSA_ERR_UNSUPPORTED): a strided [B,S,H,D] view of more than 4 GiB per head span is made head-contiguous first."""
    if k.shape[2] * k.stride(2) * k.element_size() >= 1 << 32:
        k = k.contiguous()
    return k


def _t4(t: torch.Tensor) -> RsaTensor4:
    return RsaTensor4(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_BUF_DTYPES = dict(qbar=torch.float32, aq=torch.float32, kbar=torch.float32, ak=torch.float32, vbar=torch.float32,
                   scores=torch.float32, unrel=torch.uint8, probs=torch.float32, w=torch.float32, R=torch.float32,
                   comp=torch.float32, bitmask=torch.int32, cols=torch.int32, counts=torch.int32,
                   tpart=torch.float32)


def buffer_shapes(spec: LayoutSpec, B: int, H: int, D: int) -> Dict[str, tuple]:
    BH, NBv, NB = B * H, spec.NBv, spec.NB_total
    NS, L, NW = NBv + spec.n_txt, spec.L, (NB + 31) // 32
    return dict(qbar=(BH, NBv, D), aq=(BH, NBv, D), kbar=(BH, NBv, D), ak=(BH, NBv, D), vbar=(BH, NB, D),
                scores=(BH, NBv, NS), unrel=(BH, NBv, NBv), probs=(BH, NBv, L), w=(BH, NBv, L), R=(BH, NBv),
                comp=(BH, NBv, D), bitmask=(BH, NBv, NW), cols=(BH, NBv, NB), counts=(BH, NBv),
                tpart=(BH * (NB - NBv) * _lib.TEXT_SPLIT + _lib.TAIL_PIECES, BLOCK, D + 2))


def alloc_buffers(spec: LayoutSpec, B: int, H: int, D: int, device) -> Dict[str, torch.Tensor]:
    return {n: torch.empty(s, dtype=_BUF_DTYPES[n], device=device) for n, s in buffer_shapes(spec, B, H, D).items()}


# Statistics / mask buffers are reused across calls of the same geometry on the same stream (a pipeline calls the
# operator once per layer with identical shapes: 14 allocations + a ctypes struct per call otherwise).  Only
# intermediates are cached -- the output tensor is always fresh, as in the reference.  Stream-ordered reuse is safe;
# a different stream gets its own set.  A few entries at most (0.4 GB each at the HunyuanVideo shape).
_BUF_CACHE: "Dict[tuple, Dict[str, torch.Tensor]]" = {}
BUFFER_CACHE = True                    # set False to allocate per call (e.g. while debugging memory)
BUFFER_CACHE_MAX_BYTES = 2 << 30       # least recently used sets are dropped beyond this (0.45 GB per set at the Hunyuan shape)


def _set_bytes(bufs: Dict[str, torch.Tensor]) -> int:
    return sum(t.numel() * t.element_size() for t in bufs.values())


def cached_buffers(spec: LayoutSpec, B: int, H: int, D: int, device) -> Dict[str, torch.Tensor]:
    """Intermediate buffers reused across calls of one geometry on one stream.  Never during a HIP-graph capture: a
    captured launch bakes the pointers in, and an entry evicted later would hand that memory to someone else while the
    graph still writes to it -- captured calls get their own buffers (owned by the capture's allocator pool)."""
    dev = torch.device(device)
    if not BUFFER_CACHE or (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        return alloc_buffers(spec, B, H, D, device)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    key = (dev.type, dev.index, stream, B, H, D, spec.S, spec.NB_total, spec.NBv, spec.n_txt)
    hit = _BUF_CACHE.pop(key, None)
    if hit is None:
        hit = alloc_buffers(spec, B, H, D, device)
        total = _set_bytes(hit) + sum(_set_bytes(b_) for b_ in _BUF_CACHE.values())
        while _BUF_CACHE and total > BUFFER_CACHE_MAX_BYTES:
            total -= _set_bytes(_BUF_CACHE.pop(next(iter(_BUF_CACHE))))
    _BUF_CACHE[key] = hit  # most recently used last
    return hit


def clear_buffer_cache() -> None:
    """Drops every cached buffer set (they return to PyTorch's allocator)."""
    _BUF_CACHE.clear()


def _c_buffers(bufs: Dict[str, torch.Tensor]) -> RsaBuffers:
    tp = bufs["tpart"]
    return RsaBuffers(*[bufs[n].data_ptr() if bufs[n].numel() else None for n in BUFFER_NAMES],
                      tp.numel() * tp.element_size())


def neighbor_on_device(block_neighbor_list, NBv: int, device) -> Optional[torch.Tensor]:
    """uint8 [NBv, NBv] device copy of block_neighbor_list[:NBv, :NBv] (the reference re-uploads the CPU bool
    matrix on every call, rectified_hunyuan_attn.py:267-268).  The copy is cached ON the source tensor object
    (so it lives and dies with it) and is refreshed when the tensor was modified in place."""
    if block_neighbor_list is None:
        return None
    t = block_neighbor_list
    if t.shape[0] < NBv or t.shape[1] < NBv:
        raise ValueError(f"block_neighbor_list {tuple(t.shape)} smaller than [{NBv},{NBv}]")
    key = (t._version, NBv, str(device))
    cache = getattr(t, "_rsa_device_copies", None)
    if cache is None:
        cache = {}
        try:
            t._rsa_device_copies = cache
        except AttributeError:  # exotic tensor subclass without a __dict__: no caching
            pass
    hit = cache.get(key)
    if hit is None:
        cache.clear()
        hit = t[:NBv, :NBv].to(device=device, dtype=torch.uint8).contiguous()
        cache[key] = hit
    return hit


def alloc_fp8_operands(spec: LayoutSpec, B: int, H: int, D: int, device, v_only: bool = False) -> Dict[str, torch.Tensor]:
    """e4m3 images of Q, K, V for the fp8 K5 (include/rsa.h::rsa_fp8_operands).  `scales` is ONE int32 buffer:
    [BH, NB_total] block-exponent words (byte 0 / 1 / 2 = E8M0 of the Q / K / V block) followed by the K mean, [BH, D]
    fp32 bit patterns (fp8_exps / fp8_kmean view it)."""
    assert D in (64, 128), "the fp8 block-sparse kernels are built for head_dim 64 and 128"
    BH, SP = B * H, spec.NB_total * BLOCK
    if v_only:     # the pv form reads the V image and the V exponents only
        return dict(q8=torch.empty((0,), dtype=torch.uint8, device=device), k8=torch.empty((0,), dtype=torch.uint8, device=device),
                    v8t=torch.empty((BH, SP // 64, D, 64), dtype=torch.uint8, device=device),
                    scales=torch.zeros((BH * (spec.NB_total + D),), dtype=torch.int32, device=device))
    return dict(q8=torch.empty((BH, SP, D), dtype=torch.uint8, device=device),
                k8=torch.empty((BH, SP, D), dtype=torch.uint8, device=device),
                v8t=torch.empty((BH, SP // 64, D, 64), dtype=torch.uint8, device=device),
                scales=torch.zeros((BH * (spec.NB_total + D),), dtype=torch.int32, device=device))


def fp8_exps(scales: torch.Tensor, BH: int, NB_total: int) -> torch.Tensor:
    """[BH, NB_total] int32 block-exponent words of an fp8 operand set (byte 3 masked off)."""
    return scales[: BH * NB_total].view(BH, NB_total) & 0xFFFFFF


def fp8_kmean(scales: torch.Tensor, BH: int, NB_total: int, D: int = 128) -> torch.Tensor:
    """[BH, D] fp32 "smooth K" vector of an fp8 operand set."""
    return scales[BH * NB_total:].view(torch.float32).view(BH, D)


class StagedCall:
    """One rectified-attention call with its buffers: select() runs K1..K4 (mask-selection pass), attend() runs
    K5.  Both are asynchronous on the current stream.  q, k, v: [B, H, S, D] device tensors.
    qkv_fp8: K5 runs on e4m3 images of Q, K, V (written by K1 in its own pass) on the fp8 MFMA; the mask-selection
    statistics are unchanged, so the kept lists are the 2-byte path's bit for bit."""

    def __init__(self, q, k, v, spec: LayoutSpec, top_k: int, p_remain: float, block_neighbor_list=None,
                 qkv_fp8: bool = False, reuse_buffers: bool = False):
        _require_device(q, k, v)
        self.L = _lib.lib()
        B, H, S, D = q.shape
        assert k.shape == q.shape and v.shape == q.shape, "q, k, v must have equal shapes (self-attention)"
        check_head_dim(D)
        assert k.dtype == q.dtype and v.dtype == q.dtype
        if S != spec.S:
            raise ValueError(f"layout S={spec.S} does not match tensors S={S}")
        self.q, self.k, self.v = _as_bhsd(q), _as_bhsd(k), _as_bhsd(v)
        if isinstance(qkv_fp8, str) and qkv_fp8 == "pv":
            self.k = _k_for_pv(self.k)
        self.spec, self.top_k, self.p = spec, int(top_k), float(p_remain)
        self.lay = spec.to_c(B, H, D, q.dtype)
        self.bufs = cached_buffers(spec, B, H, D, q.device) if reuse_buffers else alloc_buffers(spec, B, H, D, q.device)
        self.cb = _c_buffers(self.bufs)
        self.out = torch.empty((B, S, H, D), dtype=q.dtype, device=q.device)
        self.o4 = RsaOut4(self.out.data_ptr(), self.out.stride(0), self.out.stride(2), self.out.stride(1))
        self.nbr = neighbor_on_device(block_neighbor_list, spec.NBv, q.device)
        self.t = (_t4(self.q), _t4(self.k), _t4(self.v))
        self.fp8 = None
        # qkv_fp8: False | True (e4m3 Q, K, V and P) | "pv" (2-byte Q . K^T, e4m3 P . V)
        self.fp8_pv = isinstance(qkv_fp8, str) and qkv_fp8 == "pv"
        if isinstance(qkv_fp8, str) and not self.fp8_pv:
            raise ValueError(f"qkv_fp8 must be False, True or 'pv', got {qkv_fp8!r}")
        if self.fp8_pv and D not in (64, 128):
            raise NotImplementedError("qkv_fp8='pv' (2-byte Q.K^T + e4m3 P.V) is built for head dims 64 and 128")
        if qkv_fp8:
            self.fp8 = alloc_fp8_operands(spec, B, H, D, q.device, v_only=self.fp8_pv)
            self.cf = RsaFp8Operands(*[self.fp8[n].data_ptr() if self.fp8[n].numel() else None for n in ("q8", "k8", "v8t", "scales")])

    def quantize(self):
        """The stand-alone producer of the e4m3 images (rsa_quantize_fp8: one pass over Q, K, V).  select() does not need
        it: with qkv_fp8 K1 writes the same bytes in the pass that pools the blocks."""
        tq, tk, tv = self.t
        with torch.cuda.device(self.q.device):
            _lib.check(self.L.rsa_quantize_fp8(ctypes.byref(self.lay), tq, tk, tv, ctypes.byref(self.cf), _stream()),
                       "rsa_quantize_fp8")

    def select_pool(self):
        """K1 (writing the e4m3 images of Q, K, V as it goes when qkv_fp8)."""
        L, lay, cb, st = self.L, ctypes.byref(self.lay), ctypes.byref(self.cb), _stream()
        tq, tk, tv = self.t
        with torch.cuda.device(self.q.device):
            if self.fp8 is not None:
                _lib.check(L.rsa_pool_stats_fp8(lay, tq, tk, tv, cb, ctypes.byref(self.cf), st), "rsa_pool_stats_fp8")
            else:
                _lib.check(L.rsa_pool_stats(lay, tq, tk, tv, cb, st), "rsa_pool_stats")

    def select_rest(self):
        """K2..K4 (need only K1's statistics)."""
        L, lay, cb, st = self.L, ctypes.byref(self.lay), ctypes.byref(self.cb), _stream()
        tk = self.t[1]
        with torch.cuda.device(self.q.device):
            _lib.check(L.rsa_pooled_scores(lay, tk, cb, st), "rsa_pooled_scores")
            _lib.check(L.rsa_select_mask(lay, self.nbr.data_ptr() if self.nbr is not None else None, self.top_k,
                                         self.p, cb, st), "rsa_select_mask")
            _lib.check(L.rsa_compensation(lay, cb, st), "rsa_compensation")

    def select(self):
        self.select_pool()
        self.select_rest()

    def attend(self):
        tq, tk, tv = self.t
        if self.fp8 is not None and self.fp8_pv:
            with torch.cuda.device(self.q.device):
                _lib.check(self.L.rsa_block_sparse_fwd_fp8pv(ctypes.byref(self.lay), tq, tk, ctypes.byref(self.cf),
                                                             ctypes.byref(self.cb), self.o4, _stream()),
                           "rsa_block_sparse_fwd_fp8pv")
            return self.out
        if self.fp8 is not None:
            with torch.cuda.device(self.q.device):
                _lib.check(self.L.rsa_block_sparse_fwd_fp8(ctypes.byref(self.lay), ctypes.byref(self.cf),
                                                           ctypes.byref(self.cb), self.o4, _stream()),
                           "rsa_block_sparse_fwd_fp8")
            return self.out
        with torch.cuda.device(self.q.device):
            _lib.check(self.L.rsa_block_sparse_fwd(ctypes.byref(self.lay), tq, tk, tv, ctypes.byref(self.cb),
                                                   self.o4, _stream()), "rsa_block_sparse_fwd")
        return self.out


def rectified_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, spec: LayoutSpec, top_k: int,
                        p_remain: float, block_neighbor_list=None, return_parts: bool = False,
                        shape_xfuse: bool = False, qkv_fp8: bool = False):
    """q, k, v: [B, H, S, D] device tensors -> [B, S, H*D] (or [B, S, H, D] if shape_xfuse).

    K1 pool_stats -> K2 pooled_scores -> K3 select_mask -> K4 compensation -> K5 block_sparse_fwd on the
    current stream; no host synchronisation, no K/V mutation (the reference zeroes masked K/V rows in place,
    hunyuan :307-308; here they are treated as zero by predication)."""
    if q.shape[-1] in _PAD_HEAD_DIM:   # head dim 16 / 32: served zero-padded (exactly; see _PAD_HEAD_DIM)
        B, H, S, D = q.shape
        r = rectified_attention(*pad_small_head_dim(q, k, v), spec, top_k, p_remain, block_neighbor_list, return_parts,
                                True, qkv_fp8)
        o = (r[0] if return_parts else r)[..., :D]
        o = o.contiguous() if shape_xfuse else o.reshape(B, S, H * D)
        return (o, r[1]) if return_parts else o
    # return_parts hands the buffers to the caller, so those calls get their own set
    call = StagedCall(q, k, v, spec, top_k, p_remain, block_neighbor_list, qkv_fp8=qkv_fp8,
                      reuse_buffers=not return_parts)
    call.select()
    out = call.attend()
    B, H, S, D = q.shape
    res = out if shape_xfuse else out.view(B, S, H * D)
    if return_parts:
        parts = dict(call.bufs)
        if qkv_fp8:
            parts.update(call.fp8)
            parts["exps"] = fp8_exps(call.fp8["scales"], B * H, spec.NB_total)
            parts["kmean"] = fp8_kmean(call.fp8["scales"], B * H, spec.NB_total, D)
        return res, parts
    return res


def dense_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_split: Optional[int] = None,
                    kv_split: Optional[int] = None, qkv_fp8=False, causal: bool = False) -> torch.Tensor:
    """Exact attention on the HIP kernel.  q [B,H,Sq,D], k/v [B,H,Sk,D] -> [B,Sq,H,D].
    Rows < q_split attend kv [0, kv_split); rows >= q_split attend kv [kv_split, Sk) (attn.py:107-120); causal: inside a
    segment key j is visible to row i iff j <= i + (keys - rows) (2-byte kernel only)."""
    _require_device(q, k, v)
    if q.shape[-1] in _PAD_HEAD_DIM and k.shape[-1] == q.shape[-1] == v.shape[-1]:   # head dim 16 / 32: zero-padded, exact
        return dense_attention(*pad_small_head_dim(q, k, v), q_split, kv_split, qkv_fp8, causal)[..., :q.shape[-1]].contiguous()
    L = _lib.lib()
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    assert k.shape == v.shape and k.shape[0] == B and k.shape[1] == H and k.shape[3] == D
    check_head_dim(D)
    q, k, v = _as_bhsd(q), _as_bhsd(k), _as_bhsd(v)
    q_split = Sq if q_split is None else int(q_split)
    kv_split = Sk if kv_split is None else int(kv_split)
    out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
    o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    if isinstance(qkv_fp8, str) and qkv_fp8 != "pv":
        raise ValueError(f"qkv_fp8: False, True or 'pv', got {qkv_fp8!r}")
    if qkv_fp8:  # block-scaled e4m3 images of q, k, v + the fp8 MFMA kernel (head_dim 64 / 128)
        total = ctypes.c_size_t()
        _lib.check(L.rsa_dense_fp8_bytes(B, H, Sq, Sk, D, ctypes.byref(total)), "rsa_dense_fp8_bytes")
        ws = torch.empty(total.value, dtype=torch.uint8, device=q.device)
        if qkv_fp8 == "pv":   # scores from the 2-byte q and k, e4m3 only for P and the V image
            if D not in (64, 128):
                raise NotImplementedError("the pv form of the fp8 kernel serves head dims 64 and 128")
            k = _k_for_pv(k)
            with torch.cuda.device(q.device):
                _lib.check(L.rsa_dense_fwd_fp8pv(B, H, Sq, Sk, D, dtype_code(q.dtype), _t4(q), _t4(k), _t4(v), q_split, kv_split,
                                                 int(bool(causal)), ws.data_ptr(), ws.numel(), o4, _stream()), "rsa_dense_fwd_fp8pv")
            return out
        with torch.cuda.device(q.device):
            fn8, name8 = ((L.rsa_dense_causal_fwd_fp8, "rsa_dense_causal_fwd_fp8") if causal
                          else (L.rsa_dense_fwd_fp8, "rsa_dense_fwd_fp8"))
            _lib.check(fn8(B, H, Sq, Sk, D, dtype_code(q.dtype), _t4(q), _t4(k), _t4(v), q_split,
                           kv_split, ws.data_ptr(), ws.numel(), o4, _stream()), name8)
        return out
    fn, name = (L.rsa_dense_causal_fwd, "rsa_dense_causal_fwd") if causal else (L.rsa_dense_fwd, "rsa_dense_fwd")
    with torch.cuda.device(q.device):
        _lib.check(fn(B, H, Sq, Sk, D, dtype_code(q.dtype), _t4(q), _t4(k), _t4(v), q_split, kv_split, o4, _stream()), name)
    return out


MASK_BOOL, MASK_ADD_2BYTE, MASK_ADD_F32 = 1, 2, 3


def dense_attention_masked(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: torch.Tensor,
                           empty_rows_nan: bool = True) -> torch.Tensor:
    """softmax(q k^T / sqrt(d) + mask) v on the device for a mask that depends on the query row (rsa_dense_masked_fwd).
    q [B,H,Sq,D], k/v [B,H,Sk,D]; mask broadcastable to [B,H,Sq,Sk]: bool (False = not attended) or additive float (the dtype
    of q, or float32) -> [B,Sq,H,D].  A row without attended keys is NaN (explicit softmax) or, with empty_rows_nan=False, zeros
    (torch's fused SDPA)."""
    _require_device(q, k, v)
    L = _lib.lib()
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    assert k.shape == v.shape and k.shape[0] == B and k.shape[1] == H and k.shape[3] == D
    if D not in (64, 128):
        raise NotImplementedError(f"device fullattn with a row-dependent mask: head dim {D} (64 and 128 are built)")
    if mask.device != q.device:
        raise _lib.RsaError("attn_mask must live on the device of q")
    if mask.dtype == torch.bool:
        kind = MASK_BOOL
    elif mask.dtype == q.dtype:
        kind = MASK_ADD_2BYTE
    elif mask.dtype == torch.float32:
        kind = MASK_ADD_F32
    else:
        raise NotImplementedError(f"attn_mask dtype {mask.dtype}: bool, {q.dtype} or float32")
    m = mask
    while m.dim() < 4:
        m = m.unsqueeze(0)
    if m.dim() != 4 or any(a != 1 and a != b for a, b in zip(m.shape, (B, H, Sq, Sk))):
        raise ValueError(f"attn_mask of shape {tuple(mask.shape)} does not broadcast to {(B, H, Sq, Sk)}")
    strides = [0 if m.shape[i] == 1 else m.stride(i) for i in range(4)]
    q, k, v = _as_bhsd(q), _as_bhsd(k), _as_bhsd(v)
    out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
    o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    with torch.cuda.device(q.device):
        _lib.check(L.rsa_dense_masked_fwd(B, H, Sq, Sk, D, dtype_code(q.dtype), _t4(q), _t4(k), _t4(v), m.data_ptr(), kind,
                                          *strides, int(bool(empty_rows_nan)), o4, _stream()), "rsa_dense_masked_fwd")
    return out


def dense_attention_dropout(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, drop_rate: float, seed: int,
                            mask: Optional[torch.Tensor] = None, causal: bool = False, empty_rows_nan: bool = True) -> torch.Tensor:
    """softmax(q k^T / sqrt(d) [+ mask] [causal]) with dropout on the attention weights, times v (rsa_dense_dropout_fwd): every
    weight is kept with probability 1 - drop_rate and scaled by 1 / (1 - drop_rate) (torch.dropout / SDPA's dropout_p); the keep
    decisions are a counter-based hash of (seed, head, row, key).  Shapes and mask forms as dense_attention_masked; -> [B,Sq,H,D]."""
    _require_device(q, k, v)
    L = _lib.lib()
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    assert k.shape == v.shape and k.shape[0] == B and k.shape[1] == H and k.shape[3] == D
    if D not in (64, 128):
        raise NotImplementedError(f"device fullattn with dropout: head dim {D} (64 and 128 are built)")
    kind, mptr, strides = 0, None, [0, 0, 0, 0]
    if mask is not None:
        if mask.device != q.device:
            raise _lib.RsaError("attn_mask must live on the device of q")
        kind = MASK_BOOL if mask.dtype == torch.bool else (MASK_ADD_2BYTE if mask.dtype == q.dtype else MASK_ADD_F32)
        if kind == MASK_ADD_F32 and mask.dtype != torch.float32:
            raise NotImplementedError(f"attn_mask dtype {mask.dtype}: bool, {q.dtype} or float32")
        m = mask
        while m.dim() < 4:
            m = m.unsqueeze(0)
        if m.dim() != 4 or any(a_ != 1 and a_ != b_ for a_, b_ in zip(m.shape, (B, H, Sq, Sk))):
            raise ValueError(f"attn_mask of shape {tuple(mask.shape)} does not broadcast to {(B, H, Sq, Sk)}")
        strides = [0 if m.shape[i] == 1 else m.stride(i) for i in range(4)]
        mptr = m.data_ptr()
    q, k, v = _as_bhsd(q), _as_bhsd(k), _as_bhsd(v)
    out = torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
    o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    with torch.cuda.device(q.device):
        _lib.check(L.rsa_dense_dropout_fwd(B, H, Sq, Sk, D, dtype_code(q.dtype), _t4(q), _t4(k), _t4(v), mptr, kind, *strides,
                                           int(bool(causal)), int(bool(empty_rows_nan)), float(drop_rate),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF, o4, _stream()), "rsa_dense_dropout_fwd")
    return out


def unpack_bitmask(bitmask: torch.Tensor, n: int) -> torch.Tensor:
    """[..., NW] int32 words -> [..., n] bool (bit j%32 of word j//32)."""
    w = bitmask.to(torch.int64) & 0xFFFFFFFF
    bits = (w.unsqueeze(-1) >> torch.arange(32, device=bitmask.device)) & 1
    return bits.flatten(-2)[..., :n].bool()


def rectified_attention_onecall(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, spec: LayoutSpec, top_k: int,
                                p_remain: float, block_neighbor_list=None, workspace: Optional[torch.Tensor] = None,
                                qkv_fp8: bool = False):
    """Same operator through the single C entry point rsa_rectified_attention with one caller-provided workspace
    (what a non-Python host would call).  Returns ([B, S, H*D], workspace)."""
    _require_device(q, k, v)
    if q.shape[-1] in _PAD_HEAD_DIM:   # head dim 16 / 32: zero-padded, exact
        B, H, S, D = q.shape
        o, workspace = rectified_attention_onecall(*pad_small_head_dim(q, k, v), spec, top_k, p_remain, block_neighbor_list,
                                                   workspace, qkv_fp8)
        return o.view(B, S, H, -1)[..., :D].reshape(B, S, H * D), workspace
    if isinstance(qkv_fp8, str) and (qkv_fp8 != "pv" or q.shape[-1] not in (64, 128)):
        raise (ValueError(f"qkv_fp8: False, True or 'pv', got {qkv_fp8!r}") if qkv_fp8 != "pv"
               else NotImplementedError("the pv form of the fp8 kernel serves head dims 64 and 128"))
    L = _lib.lib()
    B, H, S, D = q.shape
    q, k, v = _as_bhsd(q), _as_bhsd(k), _as_bhsd(v)
    lay = spec.to_c(B, H, D, q.dtype)
    sizes = (ctypes.c_size_t * _lib.NUM_BUFFERS)()
    total = ctypes.c_size_t()
    _lib.check(L.rsa_buffer_bytes(ctypes.byref(lay), ctypes.byref(sizes), ctypes.byref(total)), "rsa_buffer_bytes")
    if workspace is None or workspace.numel() < total.value:
        workspace = torch.empty(total.value, dtype=torch.uint8, device=q.device)
    out = torch.empty((B, S, H, D), dtype=q.dtype, device=q.device)
    o4 = RsaOut4(out.data_ptr(), out.stride(0), out.stride(2), out.stride(1))
    nbr = neighbor_on_device(block_neighbor_list, spec.NBv, q.device)
    if qkv_fp8:
        if qkv_fp8 == "pv":
            k = _k_for_pv(k)
        s4 = (ctypes.c_size_t * 4)()
        t8 = ctypes.c_size_t()
        _lib.check(L.rsa_fp8_operand_bytes(ctypes.byref(lay), ctypes.byref(s4), ctypes.byref(t8)),
                   "rsa_fp8_operand_bytes")
        ws8 = torch.empty(t8.value, dtype=torch.uint8, device=q.device)
        fn8, name8 = ((L.rsa_rectified_attention_fp8pv, "rsa_rectified_attention_fp8pv") if qkv_fp8 == "pv"
                      else (L.rsa_rectified_attention_fp8, "rsa_rectified_attention_fp8"))
        with torch.cuda.device(q.device):
            _lib.check(fn8(ctypes.byref(lay), _t4(q), _t4(k), _t4(v), nbr.data_ptr() if nbr is not None else None, int(top_k),
                           float(p_remain), workspace.data_ptr(), workspace.numel(), ws8.data_ptr(), ws8.numel(), o4, _stream()),
                       name8)
        return out.view(B, S, H * D), workspace
    with torch.cuda.device(q.device):
        _lib.check(L.rsa_rectified_attention(ctypes.byref(lay), _t4(q), _t4(k), _t4(v),
                                             nbr.data_ptr() if nbr is not None else None, int(top_k),
                                             float(p_remain), workspace.data_ptr(), workspace.numel(), o4, _stream()),
                   "rsa_rectified_attention")
    return out.view(B, S, H * D), workspace
