"""TeaCache step skipping: the control logic of the reference's patched transformer forwards
(scripts/main_hunyuan.py:110-157, main_upflux.py:129-170, main_cogvideox.py, main_wan21t2v.py:101-164) as a small
model-agnostic controller, with the per-step statistic -- mean|x - prev| / mean|prev| of the modulated input -- taken in
one HBM pass on the HIP reduction `rsa_rel_l1` (the reference spends five elementwise / reduction launches and two
full-size temporaries on it).

Usage inside a transformer forward, mirroring the reference (single stream, HunyuanVideo style):

    tc = TeaCache.hunyuan(num_steps=50, rel_l1_thresh=0.15)          # once per pipeline
    ...
    if tc.should_compute(modulated_inp):                              # one host sync, as the reference's .cpu().item()
        x_in = hidden_states.clone()
        hidden_states = run_all_blocks(hidden_states)
        tc.store_residual(hidden_states, x_in)
    else:
        hidden_states = tc.apply_residual(hidden_states)

Wan runs two forwards per denoising step (conditional, unconditional); `streams=2` keeps separate state for even and odd
calls exactly like the reference's *_even / *_odd attributes.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _core, _lib

# rescaling polynomials of the reference scripts (highest power first, as numpy.poly1d takes them)
COEFFICIENTS = {
    "hunyuan": [7.33226126e+02, -4.01131952e+02, 6.75869174e+01, -3.14987800e+00, 9.61237896e-02],   # main_hunyuan.py:118
    "flux": [4.98651651e+02, -2.83781631e+02, 5.58554382e+01, -3.82021401e+00, 2.64230861e-01],      # main_upflux.py:137
    # main_wan21t2v.py:273-286: the script keys the tables by model size ('1.3B' / '14B' in model_id) and use_ret_steps
    "wan21_1_3b_ret": [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],  # :275
    "wan21_14b_ret": [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],   # :277
    "wan21_1_3b": [2.39676752e+03, -1.31110545e+03, 2.01331979e+02, -8.29855975e+00, 1.37887774e-01],       # :282
    "wan21_14b": [-5784.54975374, 5449.50911966, -1811.16591783, 256.27178429, -13.02252404],               # :284
    # main_cogvideox.py:20-26, keyed by the model directory name as the script does (:247)
    "CogVideoX-2b": [-3.10658903e+01, 2.54732368e+01, -5.92380459e+00, 1.75769064e+00, -3.61568434e-03],
    "CogVideoX-5b": [-1.53880483e+03, 8.43202495e+02, -1.34363087e+02, 7.97131516e+00, -5.23162339e-02],
    "CogVideoX-5b-I2V": [-1.53880483e+03, 8.43202495e+02, -1.34363087e+02, 7.97131516e+00, -5.23162339e-02],
    "CogVideoX1.5-5B": [2.50210439e+02, -1.65061612e+02, 3.57804877e+01, -7.81551492e-01, 3.58559703e-02],
    "CogVideoX1.5-5B-I2V": [1.22842302e+02, -1.04088754e+02, 2.62981677e+01, -3.06009921e-01, 3.71213220e-02],
}


def rel_l1_distance(x: torch.Tensor, prev: torch.Tensor) -> float:
    """mean|x - prev| / mean|prev| as a python float (one host sync, like the reference's `.cpu().item()`).
    Device bf16 / fp16 tensors go through the one-pass HIP reduction; anything else (CPU plumbing runs, fp32 timestep
    embeddings of a few KB) uses the reference's own expression."""
    if x.shape != prev.shape:
        raise ValueError(f"shape mismatch {tuple(x.shape)} vs {tuple(prev.shape)}")
    if x.is_cuda and prev.is_cuda and x.dtype == prev.dtype and x.dtype in (torch.bfloat16, torch.float16):
        a, b = x.contiguous(), prev.contiguous()
        if a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0:
            out = torch.empty(2 + 2048, dtype=torch.float32, device=x.device)
            vp = ctypes.c_void_p
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().rsa_rel_l1(vp(a.data_ptr()), vp(b.data_ptr()), a.numel(), _core.dtype_code(a.dtype),
                                                 vp(out.data_ptr()), vp(out[2:].data_ptr()), _core._stream()),
                           "rsa_rel_l1")
            sd, sb = out[:2].tolist()
            if sb == 0.0:  # all-zero previous input: the reference's tensor expression gives inf / nan -> recompute
                return float("inf") if sd > 0.0 else float("nan")
            return sd / sb   # the two means share the element count
    return ((x - prev).abs().mean() / prev.abs().mean()).cpu().item()


class TeaCache:
    """Accumulated-relative-L1 step skipping.  `cnt` counts forward calls and wraps at `total_calls`; call index c
    belongs to stream c % streams; a call always computes when cnt < ret_calls or cnt >= cutoff_calls."""

    def __init__(self, total_calls: int, thresh: float, coefficients: Sequence[float], streams: int = 1,
                 ret_calls: int = 1, cutoff_calls: Optional[int] = None, start_cnt: int = 0):
        self.start_cnt = int(start_cnt)
        self.total_calls = int(total_calls)
        self.thresh = float(thresh)
        self.poly = np.poly1d(list(coefficients))
        self.streams = int(streams)
        self.ret_calls = int(ret_calls)
        self.cutoff_calls = self.total_calls - 1 if cutoff_calls is None else int(cutoff_calls)
        self.reset()

    # -- the reference's configurations ----------------------------------------------------------------------
    @classmethod
    def hunyuan(cls, num_steps: int = 50, rel_l1_thresh: float = 0.15) -> "TeaCache":
        """main_hunyuan.py:114: always compute on the first and the last step."""
        return cls(num_steps, rel_l1_thresh, COEFFICIENTS["hunyuan"], 1, 1, num_steps - 1)

    @classmethod
    def flux(cls, num_steps: int, rel_l1_thresh: float) -> "TeaCache":
        return cls(num_steps, rel_l1_thresh, COEFFICIENTS["flux"], 1, 1, num_steps - 1)

    @classmethod
    def cogvideox(cls, model: str, num_steps: int = 50, rel_l1_thresh: float = 0.2) -> "TeaCache":
        """main_cogvideox.py:106-118: the statistic is taken on the time embedding `emb`; always compute on the first and
        the last step; `model` is the checkpoint directory name ('CogVideoX1.5-5B', ...).  The forward caches TWO
        residuals (video and text tokens, :133-134): use store_residual / apply_residual with slot=0 and slot=1."""
        return cls(num_steps, rel_l1_thresh, COEFFICIENTS[model], 1, 1, num_steps - 1)

    @classmethod
    def wan22_pair(cls, num_steps: int, transformer_steps: int, teacache_thresh: float = 0.2, use_ret_steps: bool = True,
                   big: bool = True):
        """Wan2.2 T2V / I2V run two transformers over one schedule (main_wan22t2v.py:82-127, main_wan22i2v.py:90-135):
        `transformer` serves the first `transformer_steps` denoising steps, `transformer_2` the rest, each with its own
        controller.  The script starts transformer_2's call counter at 2*transformer_steps and gives it the windows
        below; its counter wraps to 0 (not to its start value) after 2*num_steps calls, which is kept.  In the
        use_ret_steps = False branch the script assigns `transformer.ret_steps` twice and `transformer_2.cutoff_steps`
        twice (:124-127), leaving transformer.cutoff_steps and transformer_2.ret_steps unset -- that branch cannot run
        upstream, so only use_ret_steps = True is offered here.  (Wan2.2 TI2V has one transformer: TeaCache.wan.)"""
        if not use_ret_steps:
            raise NotImplementedError("the reference's use_ret_steps=False branch for two transformers leaves "
                                      "cutoff_steps / ret_steps unset (main_wan22t2v.py:124-127)")
        coeff = COEFFICIENTS[("wan21_14b" if big else "wan21_1_3b") + "_ret"]
        t1 = cls(2 * transformer_steps, teacache_thresh, coeff, 2, 3 * 2, 2 * transformer_steps)
        t2 = cls(2 * num_steps, teacache_thresh, coeff, 2, 2 * transformer_steps + 1 * 2, 2 * num_steps,
                 start_cnt=2 * transformer_steps)
        return t1, t2

    @classmethod
    def wan(cls, num_steps: int, teacache_thresh: float = 0.2, use_ret_steps: bool = True, big: bool = True) -> "TeaCache":
        """main_wan21t2v.py:273-286: two calls per step; ret_steps = 5*2 / 1*2, cutoff = 2*steps / 2*steps - 2."""
        key = ("wan21_14b" if big else "wan21_1_3b") + ("_ret" if use_ret_steps else "")
        if use_ret_steps:
            return cls(2 * num_steps, teacache_thresh, COEFFICIENTS[key], 2, 5 * 2, 2 * num_steps)
        return cls(2 * num_steps, teacache_thresh, COEFFICIENTS[key], 2, 1 * 2, 2 * num_steps - 2)

    # -- state ---------------------------------------------------------------------------------------------------
    def reset(self):
        self.cnt = self.start_cnt
        self.accumulated = [0.0] * self.streams
        self.previous_input: List[Optional[torch.Tensor]] = [None] * self.streams
        self.previous_residual: List[dict] = [{} for _ in range(self.streams)]   # per stream: slot -> tensor
        self._stream = 0

    @property
    def stream(self) -> int:
        """stream (0 = even / conditional, 1 = odd / unconditional) of the call should_compute() was last asked about"""
        return self._stream

    def should_compute(self, modulated_inp: torch.Tensor) -> bool:
        s = self._stream = self.cnt % self.streams
        if self.cnt < self.ret_calls or self.cnt >= self.cutoff_calls or self.previous_input[s] is None:
            calc = True
            self.accumulated[s] = 0.0
        else:
            self.accumulated[s] += float(self.poly(rel_l1_distance(modulated_inp, self.previous_input[s])))
            if self.accumulated[s] < self.thresh:
                calc = False
            else:
                calc = True
                self.accumulated[s] = 0.0
        self.previous_input[s] = modulated_inp.clone()
        self.cnt += 1
        if self.cnt == self.total_calls:
            self.cnt = 0
        if not calc and not self.previous_residual[s]:  # nothing cached yet (fresh controller mid-run)
            calc = True
        return calc

    def store_residual(self, hidden_out: torch.Tensor, hidden_in: torch.Tensor, slot: int = 0):
        """slot: which of the forward's cached residuals (CogVideoX keeps two: 0 = video tokens, 1 = text tokens)"""
        self.previous_residual[self._stream][slot] = hidden_out - hidden_in

    def apply_residual(self, hidden_states: torch.Tensor, slot: int = 0) -> torch.Tensor:
        hidden_states += self.previous_residual[self._stream][slot]
        return hidden_states
