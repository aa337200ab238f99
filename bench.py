#!/usr/bin/env python3
"""Attention-layer benchmark for the rectified block-sparse attention path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of the whole hot path (K1 pool_stats .. K5 block_sparse_fwd) over one synthetic
HunyuanVideo-720p attention call: B=1, H=24, S=115 200 visual + 256 text (200 valid), D=128, bf16, top_k = 90
of 900 visual blocks (10 % kept), inputs already resident in HBM, generated on the device by the counter-based
generator of rectified_spaattn_amd/synth_device.py.

--gpus N > 1: the 24 heads are sharded contiguously (24/N per rank, strong scaling, no data-path collective);
time = max over ranks.  Launched without an outer launcher, this process never touches the GPU: it starts
`python -m torch.distributed.run --nproc-per-node N` as a CHILD and exits with its code; launched under
torch.distributed.run (WORLD_SIZE set) it is a rank.  Rank 0 prints ONE JSON line.

The line's `value` is regime R2 of SURVEY.md 8(d) (exactly top_k kept visual blocks + the text blocks, independent
block centroids).  Sub-records measured in the same run (N = 1): `regimes.r1` (Gilbert neighbours, p = 0.05),
`regimes.locality` (spatially smooth centroids -> neighbouring query blocks keep overlapping lists, Gilbert
neighbours, p = 0.05), `api` (the public operator and a processor __call__ end to end), `sustained` (>= 2 s of
back-to-back steps), `cpu_baseline` (reference dense path + the rectified sparse path on the same mask, host cores).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
MFMA_FP8_PEAK_TFLOPS = 5000.0   # MI355X dense fp8 (same table)
SEED = 20251212                 # seed + global head index

WORKLOADS = {
    # top_k 90 = the north star's "10 % kept blocks"; script_top_k = what scripts/main_hunyuan.py ships (sa_drop_rate 0.8:
    # int(0.2 * 900), :219, :251-254).  The other workloads' top_k ARE their scripts' (main_upflux.py:274 0.9, main_wan21t2v.py:218
    # / main_wan22ti2v.py:236 / main_cogvideox.py:313 0.75).
    "hunyuan_720p_128f": dict(H=24, S_vis=115200, text=256, text_valid=200, top_k=90, script_top_k=180, variant="hunyuan",
                              latent=(32, 45, 80)),
    "flux_4096": dict(H=24, S_vis=65536, text=512, text_valid=512, top_k=51, variant="flux", latent=(1, 256, 256)),
    "wan21_720p_81f": dict(H=40, S_vis=75600, text=0, text_valid=0, top_k=147, variant="wan", ffb=28,
                           latent=(21, 45, 80)),
    "wan22_ti2v_720p_121f": dict(H=24, S_vis=27280, text=0, text_valid=0, top_k=53, variant="wan", ffb=6,
                                 latent=(31, 22, 40)),
    # CogVideoX1.5 81f 768x1280 (scripts/main_cogvideox.py:226-235; not one of BASELINE's configs): head dim 64
    "cogvideox_768p_81f": dict(H=48, S_vis=42240, text=226, text_valid=226, top_k=82, variant="cogvideo", D=64,
                               latent=(11, 48, 80)),
    # eight heads, one per rank of an 8-process functional run on one device (tests/test_gpu_shard_invariance.py)
    "tiny8": dict(H=8, S_vis=8192, text=256, text_valid=200, top_k=6, variant="hunyuan", latent=(8, 32, 32)),
    "tiny": dict(H=4, S_vis=4096, text=256, text_valid=200, top_k=6, variant="hunyuan", latent=(4, 32, 32)),
}

# regime -> (centroid model, neighbour matrix, p_remain)
REGIMES = {
    "r2": ("iid", "none", 0.0),            # controlled: exactly top_k kept visual blocks per row (+ text blocks)
    "r1": ("iid", "gilbert", 0.05),        # algorithmic: cumulative-probability rule + true Gilbert neighbours
    "locality": ("spatial", "gilbert", 0.05),  # as r1 on spatially smooth centroids (overlapping kept lists)
    # the reference scripts' shipped operating point: top_k = script_top_k of the workload, p_remain_rates 0.3 (every script's
    # default, e.g. main_hunyuan.py:220), true Gilbert neighbours (:245-246), spatially smooth centroids
    "script": ("spatial", "gilbert", 0.3),
}


def regime_top_k(wl, regime):
    return wl.get("script_top_k", wl["top_k"]) if regime == "script" else wl["top_k"]
K5_SOURCES = {False: ("rsa_attn_kernel64.hip", "gen_k5_block64.py", "rsa_attn_kernel.hip", "rsa_attn.h", "rsa_attn.hip", "gen_k5_block.py"),
              True: ("rsa_attn_fp8_kernel.hip", "rsa_attn.h", "rsa_attn.hip", "gen_k5_block.py")}


def k5_peak(qkv_fp8) -> float:
    """Dense MFMA peak the K5 of this operand form is priced against; the pv form runs half its FLOPs (Q.K^T) on the 2-byte
    pipe and half (P.V) on the fp8 pipe: 1 / (0.5 / 2500 + 0.5 / 5000) = 3 333 TFLOP/s."""
    if qkv_fp8 == "pv":
        return round(1.0 / (0.5 / MFMA_BF16_PEAK_TFLOPS + 0.5 / MFMA_FP8_PEAK_TFLOPS), 1)
    return MFMA_FP8_PEAK_TFLOPS if qkv_fp8 else MFMA_BF16_PEAK_TFLOPS


def traffic_suffix(qkv_fp8) -> str:
    return "_pv" if qkv_fp8 == "pv" else ("_fp8" if qkv_fp8 else "")


def kernel_source_sha(fp8: bool = False) -> str:
    """sha256 over the CODE of one K5 kernel's sources (the 2-byte kernel or the e4m3 one; `//` and `#` comments, blank
    lines and indentation do not count): profiles/*traffic*.json carry the value they were collected with."""
    h = hashlib.sha256()
    for n in K5_SOURCES[bool(fp8)]:
        mark = "#" if n.endswith(".py") else "//"
        with open(os.path.join(ROOT, "rectified_spaattn_amd", "csrc", n), "r") as f:
            for line in f:
                code = line.split(mark, 1)[0].strip()
                if code:
                    h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def make_spec(wl):
    from rectified_spaattn_amd import _core
    S = wl["S_vis"] + wl["text"]
    if wl["variant"] == "hunyuan":
        return _core.LayoutSpec.hunyuan(S, wl["S_vis"] + wl["text_valid"])
    if wl["variant"] == "flux":
        return _core.LayoutSpec.flux(S, wl["text"])
    if wl["variant"] == "cogvideo":
        return _core.LayoutSpec.cogvideo(S, wl["text"])
    return _core.LayoutSpec.wan(S, wl.get("ffb", 0))


def make_neighbors(wl, spec, kind):
    import torch
    if kind == "none":
        return None
    if kind == "gilbert":
        from rectified_spaattn_amd.utils import jenga_gilbert
        return jenga_gilbert.gilbert_block_neighbor_mapping(*wl["latent"], axis_order=("w", "h", "t"))
    from rectified_spaattn_amd import synth
    return torch.from_numpy(synth.banded_neighbors(spec.NBv, int(kind)))


def gen_qkv(H_local, head0, S, S_vis, D, dev, seed=SEED, centroid_fn=None):
    """Structured synthetic Q/K/V from the counter-based generator, on the device, per head
    (seed + global head index): per 128-token block a centroid shared by Q and K, V ~ N(0,1)."""
    from rectified_spaattn_amd import synth_device
    return synth_device.structured_qkv_device(seed, H_local, head0, S, D, dev, centroid_fn=centroid_fn)


def gen_inputs(wl, H_local, head0, dev, centroids="iid", D=128, corr_len=6.0):
    """Inputs of one regime: 'iid' block centroids (R2 / R1) or the spatially smooth field ('spatial', locality)."""
    from rectified_spaattn_amd import synth_device
    S = wl["S_vis"] + wl["text"]
    fn = None
    if centroids == "spatial":
        field = synth_device.SpatialField(wl["latent"], (S + 127) // 128, corr_len=corr_len)
        fn = lambda seed: field.centroids(seed, D)  # noqa: E731
    return gen_qkv(H_local, head0, S, wl["S_vis"], D, dev, centroid_fn=fn)


def cpu_baseline_dense(S, D, budget_s=20.0):
    """The reference's CPU dense path (fullattn mode='torch' = SDPA, attn.py:101-106) restated in
    oracle/dense_cpu.py, timed on this host's cores on a bounded sample: 1 head, as many query rows as fit the
    time budget, all S keys, bf16."""
    import torch
    from oracle import dense_cpu
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    g = torch.Generator().manual_seed(1)
    k = torch.randn(1, 1, S, D, generator=g).to(torch.bfloat16)
    v = torch.randn(1, 1, S, D, generator=g).to(torch.bfloat16)
    rows = 2048
    q = torch.randn(1, 1, rows, D, generator=g).to(torch.bfloat16)
    dense_cpu.fullattn_torch_cpu(q[:, :, :256], k, v)  # warm-up
    t0 = time.perf_counter()
    dense_cpu.fullattn_torch_cpu(q, k, v)
    dt = time.perf_counter() - t0
    rows2 = int(min(S, max(rows, rows * budget_s * 0.5 / max(dt, 1e-3))))
    if rows2 > rows * 2:
        q = torch.randn(1, 1, rows2, D, generator=g).to(torch.bfloat16)
        t0 = time.perf_counter()
        dense_cpu.fullattn_torch_cpu(q, k, v)
        dt = time.perf_counter() - t0
        rows = rows2
    flops = 4.0 * rows * S * D
    return dict(value=flops / dt / 1e12, unit="TFLOP/s", cores=ncores, kind="port",
                sample=f"dense SDPA bf16 (reference fullattn mode='torch'), 1 head, {rows} query rows x {S} keys, "
                       f"D={D}: {dt:.2f} s")


def capture_sparse_sample(call):
    """Head 0 of a finished call, copied to the host: inputs + the mask / rectification terms the GPU produced."""
    return tuple(t[0, 0].cpu() for t in (call.q, call.k, call.v)) + tuple(
        call.bufs[n][0].cpu() for n in ("cols", "counts", "R", "comp"))


def cpu_baseline_sparse(sample, spec, D, budget_s=10.0):
    """The rectified SPARSE path on the same mask the GPU used (oracle/sparse_cpu.py: per query block exact attention
    over its kept blocks, then *R + comp), head 0 of this rank, a bounded sample of query blocks, bf16, host cores."""
    import torch
    from oracle import sparse_cpu
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    q, k, v, cols, counts, R, comp = sample
    nb = min(spec.NBv, 96)   # many query blocks per call: they are the parallel axis of the CPU SDPA kernel
    step = max(1, spec.NBv // nb)
    blocks = list(range(0, spec.NBv, step))[:nb]
    sparse_cpu.rectified_sparse_blocks_cpu(q, k, v, cols, counts, R, comp, spec.kv_valid, blocks[:8])  # warm-up
    t0 = time.perf_counter()
    sparse_cpu.rectified_sparse_blocks_cpu(q, k, v, cols, counts, R, comp, spec.kv_valid, blocks)
    dt = time.perf_counter() - t0
    nb2 = int(min(spec.NBv, max(nb, nb * budget_s * 0.6 / max(dt, 1e-3))))
    if nb2 > nb * 2:
        step = max(1, spec.NBv // nb2)
        blocks = list(range(0, spec.NBv, step))[:nb2]
        t0 = time.perf_counter()
        sparse_cpu.rectified_sparse_blocks_cpu(q, k, v, cols, counts, R, comp, spec.kv_valid, blocks)
        dt = time.perf_counter() - t0
    pairs = int(counts[torch.tensor(blocks)].sum().item())
    flops = 4.0 * D * 128 * 128 * pairs
    return dict(value=flops / dt / 1e12, unit="TFLOP/s", cores=ncores, kind="port",
                sample=f"rectified sparse path, same mask as the GPU run, head 0, {len(blocks)} query blocks "
                       f"({pairs} kept block pairs), bf16 SDPA over gathered kept blocks + R/comp: {dt:.2f} s")


# --------------------------------------------------------------------------------------------------------------
# multi-process plumbing
# --------------------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv) -> int:
    """Parent of an N-rank run.  Nothing here imports torch.cuda or touches the GPU; the ranks are children of a
    `torch.distributed.run` CHILD process (never an exec of this one)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


class Comm:
    """torch.distributed in one place: nccl (= RCCL) on GPUs, gloo for --dry."""

    def __init__(self, dry):
        import torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = dry
        self.dist = None
        # RSA_BENCH_ONE_DEVICE=1 (tests only): every rank computes on cuda:0 and the control plane runs over gloo, so the
        # N > 1 code path (spawn, head shards, per-rank timing, exchange over IPC) can be exercised on a 1-GPU box.
        # The numbers of such a run are functional evidence, not a measurement (the ranks time-slice one GPU).
        self.one_device = (not dry) and os.environ.get("RSA_BENCH_ONE_DEVICE") == "1"
        if dry:
            self.dev = torch.device("cpu")
        else:
            self.dev = torch.device("cuda", 0 if self.one_device else self.local_rank)
            torch.cuda.set_device(self.dev)
        self.stat_dev = torch.device("cpu") if (dry or self.one_device) else self.dev
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if dry or self.one_device:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
            self.dist = dist

    def local_sync(self):
        import torch
        if not self.dry:
            torch.cuda.synchronize()

    def sync(self):
        self.local_sync()
        if self.dist is not None:
            self.dist.barrier()
        self.local_sync()

    def all_ok(self, ok: bool) -> bool:
        """True only if EVERY rank says ok (a rank that failed inside a guarded stage must not leave its peers inside
        that stage's collectives: all ranks agree here first, then skip the stage together)."""
        if self.dist is None:
            return ok
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.stat_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


class SmiSampler:
    """Clock / power telemetry over a timed window: `rocm-smi --showclocks --showpower --json` polled from a thread (about
    4 samples per second).  The kernels run at the board's power cap, so the clock the chip holds is part of the result."""

    def __init__(self, period=0.25):
        import threading
        self.period, self.rows, self._stop = period, [], False
        self._t = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _num(x):
        import re
        m = re.search(r"[-+]?\d+(\.\d+)?", str(x))
        return float(m.group(0)) if m else None

    def _run(self):
        while not self._stop:
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                                   timeout=5)
                c = json.loads(r.stdout).get("card0", {})
                row = {}
                for k_, v_ in c.items():
                    kl = k_.lower()
                    if "sclk clock speed" in kl: row["sclk_mhz"] = self._num(v_)
                    elif "mclk clock speed" in kl: row["mclk_mhz"] = self._num(v_)
                    elif "power" in kl and "(w)" in kl: row["power_w"] = self._num(v_)
                if row:
                    self.rows.append(row)
            except Exception:  # noqa: BLE001  (telemetry is best effort)
                pass
            time.sleep(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._t.join(timeout=6)

    def summary(self):
        rows = self.rows[2:] if len(self.rows) > 4 else self.rows   # the first samples still see the ramp
        out = {"samples": len(rows)}
        for key in ("sclk_mhz", "mclk_mhz", "power_w"):
            vals = [r_[key] for r_ in rows if r_.get(key) is not None]
            if vals:
                out[key] = {"min": min(vals), "max": max(vals), "mean": round(sum(vals) / len(vals), 1)}
        return out


def timed_steps(comm, step, steps, warmup, events=None, want_busy=False):
    """W untimed warm-up steps, then EXACTLY `steps` steps between barrier + synchronize on both sides.  Returns the
    bracketed time (and, with want_busy, this rank's own time up to its local synchronize, before the closing
    barrier -- what shows the imbalance between ranks)."""
    for _ in range(warmup):
        step(None)
    comm.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(events[i] if events is not None else None)
    comm.local_sync()
    busy = time.perf_counter() - t0
    comm.sync()
    total = time.perf_counter() - t0
    return (total, busy) if want_busy else total


# (max, mean) |dO| of the output check below.  2-byte operands: against the dense-masked fp32 reference on the inputs themselves (the
# operator tests' bound).  fp8 forms: against the same reference on the operands the kernel MULTIPLIES (the dequantised e4m3 images;
# pv: the 2-byte q, k and the dequantised V) -- what is left is the e4m3 rounding of P, fp32 accumulation and the output's rounding
# (tests/test_gpu_fp8.py: 4e-2 / 4e-3 against that oracle on full blocks, measured 1.6e-2..2.3e-2 / 2.6e-3..2.9e-3; 3e-2..5e-2 /
# 6e-3 on the 16-row last block of Wan2.2-TI2V, whose rows hang on a handful of keys and average less of it).  The distance to the
# UN-quantised reference is the number format's, not the kernel's (tests/diag/diag_fp8_ragged_block.py: the oracle on the e4m3
# operands alone is 0.35 away on the 16-row last block of Wan2.2-TI2V, whose rows hang on a handful of keys, and 0.04-0.07
# elsewhere); it is reported in `vs_unquantised` and only its mean is bounded.
CHECK_TOL = {False: (2e-2, 2e-3), True: (8e-2, 5e-3), "pv": (8e-2, 5e-3)}
CHECK_TOL_UNQUANTISED_MEAN = {True: 2e-2, "pv": 1e-2}


def _e4m3_lut(dev):
    """fp32 value of every e4m3 byte (OCP e4m3fn: bias 7, 3 mantissa bits, subnormals in steps of 2^-9)."""
    import torch
    b = torch.arange(256, device=dev)
    expo, mant = (b >> 3) & 15, (b & 7).float()
    val = torch.where(expo == 0, mant * 2.0 ** -9, (1.0 + mant / 8.0) * torch.exp2(expo.float() - 7.0))
    return torch.where((b & 0x80) != 0, -val, val)


def fp8_multiplied_operands(call, spec, h, pv):
    """fp32 [S, D] q, k, v of local head h AS THE fp8 K5 MULTIPLIES THEM, rebuilt from the call's own images with plain torch ops
    (include/rsa.h::rsa_fp8_operands): block-scaled e4m3 bytes x 2^(E8M0 - 127); q back in its own units (the image holds
    q * sm_scale * log2 e), k = K minus the head's smooth-K vector (a shift of every score of a row alike: the softmax does not
    see it), V from its transposed tiles ([S_pad / 64, D, 64], key order of a tile = rsa_fp8.hip's k-slot order).  pv form: q and
    k are the 2-byte inputs themselves, only V comes from its image."""
    import torch
    from rectified_spaattn_amd import _core
    B, H, S, D = call.q.shape
    dev = call.q.device
    BH, NBt = B * H, spec.NB_total
    lut = _e4m3_lut(dev)
    ex = _core.fp8_exps(call.fp8["scales"], BH, NBt)[h]
    blk_scale = lambda byte: torch.exp2(((ex >> (8 * byte)) & 0xFF).float() - 127.0).repeat_interleave(128)[:, None]   # noqa: E731
    v8t = call.fp8["v8t"].view(BH, NBt * 2, D, 64)[h]
    p = torch.arange(64, device=dev)
    j = p & 31
    key_of_slot = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * (p >> 5)
    vt = torch.empty((NBt * 2, 64, D), dtype=torch.float32, device=dev)
    vt[:, key_of_slot, :] = lut[v8t.long()].permute(0, 2, 1)
    vd = (vt.reshape(NBt * 128, D) * blk_scale(2))[:S]
    if pv:
        return call.q[0, h].float(), call.k[0, h].float(), vd
    qk_const = float(D) ** -0.5 * 1.44269504
    qd = (lut[call.fp8["q8"].view(BH, NBt * 128, D)[h].long()] * blk_scale(0))[:S] / qk_const
    kd = (lut[call.fp8["k8"].view(BH, NBt * 128, D)[h].long()] * blk_scale(1))[:S]
    return qd, kd, vd


def check_output(call, spec, qkv_fp8=False, n_heads=3, n_blocks=8):
    """What the timed steps PRODUCED, checked after the timed region (the launch that is timed -- all local heads, aligned starts,
    tail split as planned from its size -- is the launch that is checked): every element of O finite; `n_blocks` query blocks
    (first, last and evenly spaced ones) of `n_heads` heads (first, middle, last) and every text row of those heads against a
    dense-masked fp32 reference computed here with plain torch ops on the device from the call's OWN kept lists, R and comp
    (softmax over the kept keys below kv_valid, x R + comp; text rows: exact attention over the valid keys; padded text rows
    zero).  With fp8 operands the reference that decides `ok` runs on the operands the kernel multiplies
    (fp8_multiplied_operands); the same reference on the un-quantised inputs is reported beside it (CHECK_TOL).  The selection
    pass that made the lists is checked bit for bit elsewhere (tests/test_gpu_fullsize.py); this is the check of K5 at the
    launch's own size.  Returns the record of the bench line's `check` field."""
    import torch
    q, k, v, out = call.q, call.k, call.v, call.out          # [B,H,S,D] x 3, [B,S,H,D]
    B, H, S, D = q.shape
    dev = q.device
    finite = bool(torch.isfinite(out).all())
    heads = sorted({0, H // 2, H - 1})[:max(1, n_heads)]
    NBv = spec.NBv
    blocks = sorted({int(round(i * (NBv - 1) / max(1, n_blocks - 1))) for i in range(n_blocks)}) if NBv > 0 else []
    cols, counts, Rb, compb = (call.bufs[n] for n in ("cols", "counts", "R", "comp"))
    scale = float(D) ** -0.5
    ar = torch.arange(128, device=dev)
    mode = "pv" if qkv_fp8 == "pv" else bool(qkv_fp8)
    tol_max, tol_mean = CHECK_TOL[mode]

    class Dist:
        def __init__(self):
            self.worst, self.mean_sum, self.n, self.at = 0.0, 0.0, 0, None

        def add(self, err, where):
            e = float(err.max())
            if e > self.worst:
                self.worst, self.at = e, where
            self.mean_sum += float(err.mean()); self.n += 1

        def mean(self):
            return self.mean_sum / max(1, self.n)

    def reference(qf, kf, vf, h, dist):
        for i in blocks:
            n = int(counts[h, i].item())
            sel = cols[h, i, :n].long()
            key_idx = (sel[:, None] * 128 + ar[None]).reshape(-1)
            valid = key_idx < spec.kv_valid
            key_idx = key_idx.clamp(max=S - 1)
            rows = slice(i * 128, min(S, (i + 1) * 128))
            sc = (qf[rows] @ kf[key_idx].t()) * scale
            sc = sc.masked_fill(~valid[None, :], float("-inf"))
            ref = torch.softmax(sc, dim=-1) @ vf[key_idx] * Rb[h, i] + compb[h, i][None, :]
            dist.add((out[0, rows, h].float() - ref).abs(), (h, i))
        if spec.q_text_valid > 0:
            r0 = NBv * 128
            sc = (qf[r0:r0 + spec.q_text_valid] @ kf[:spec.kv_text_valid].t()) * scale
            ref = torch.softmax(sc, dim=-1) @ vf[:spec.kv_text_valid]
            dist.add((out[0, r0:r0 + spec.q_text_valid, h].float() - ref).abs(), (h, "text"))

    own, plain, text_rows, pad_bad = Dist(), Dist(), 0, None
    for h in heads:
        qf, kf, vf = (x[0, h].float() for x in (q, k, v))
        reference(qf, kf, vf, h, plain)
        if mode is not False:
            reference(*fp8_multiplied_operands(call, spec, h, mode == "pv"), h, own)
        if spec.q_text_valid > 0:
            r0 = NBv * 128
            text_rows += spec.q_text_valid
            if r0 + spec.q_text_valid < S and float(out[0, r0 + spec.q_text_valid:, h].float().abs().max()) != 0.0:
                pad_bad = (h, "padded text rows not zero")
    gate = plain if mode is False else own
    ok = finite and pad_bad is None and gate.worst <= tol_max and gate.mean() <= tol_mean
    rec = dict(ok=False, finite=finite, heads=heads, blocks=len(blocks) * len(heads), text_rows=text_rows,
               max_abs=round(gate.worst, 6), mean_abs=round(gate.mean(), 6),
               worst_at=list(pad_bad or gate.at) if (pad_bad or gate.at) else None,
               tol=dict(max_abs=tol_max, mean_abs=tol_mean))
    what = ("O of the last timed step (all local heads in one launch): finite everywhere; sampled query blocks + all text rows "
            "of the listed heads vs a dense-masked fp32 torch reference on the device built from the call's own kept lists, R, comp")
    if mode is not False:
        tol_u = CHECK_TOL_UNQUANTISED_MEAN[mode]
        ok = ok and plain.mean() <= tol_u
        rec["vs_unquantised"] = dict(max_abs=round(plain.worst, 6), mean_abs=round(plain.mean(), 6),
                                     worst_at=list(plain.at) if plain.at else None, tol_mean_abs=tol_u)
        what += ("; fp8 operands: max_abs / mean_abs are against that reference on the operands the kernel multiplies (dequantised "
                 "e4m3 images" + ("" if mode is True else " of V, the 2-byte q and k") + "), vs_unquantised against it on the inputs "
                 "themselves (the number format's distance: reported, only its mean bounded)")
    rec["ok"] = bool(ok)
    rec["what"] = what
    return rec


# --------------------------------------------------------------------------------------------------------------
def run_regime(comm, args, wl, regime, q, k, v, spec, steps, warmup, want_call=False):
    """Times the staged hot path (select -> [quantize] -> attend) for one regime; returns the record (+ the call)."""
    import torch
    from rectified_spaattn_amd import _core, parallel
    D = q.shape[-1]
    _, nbr_kind, p = REGIMES[regime]
    if args.neighbors is not None and regime == "r2":
        nbr_kind = args.neighbors
    if args.p_remain is not None and regime == "r2":
        p = args.p_remain
    nbr = make_neighbors(wl, spec, nbr_kind)
    call = _core.StagedCall(q, k, v, spec, regime_top_k(wl, regime), p, nbr, qkv_fp8=args.qkv_fp8)

    def step(ev):
        call.select()
        if ev is not None:
            ev[0].record()
        call.attend()
        if ev is not None:
            ev[1].record()

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    elapsed, busy = timed_steps(comm, step, steps, warmup, evs, want_busy=True)
    k5_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, steps)
    H_local = q.shape[1]
    counts = float(call.bufs["counts"].sum().item())  # kept (q-block, k-block) pairs over local heads
    local_flops = 4.0 * D * 128 * 128 * counts + 4.0 * D * spec.q_text_valid * spec.kv_text_valid * H_local
    el_max, fl_sum, pairs_sum, k5_max, per_rank = parallel.reduce_step_stats(elapsed, local_flops, counts, k5_ms,
                                                                            comm.stat_dev, busy_s=busy)
    H = wl["H"]
    rec = dict(regime=regime, neighbors=nbr_kind, p_remain=p, top_k=regime_top_k(wl, regime),
               ms_per_step=el_max / steps * 1e3, value=fl_sum / (el_max / steps) / 1e12,
               kept_block_fraction=pairs_sum / (H * spec.NBv * spec.NB_total),
               k5_ms=k5_ms, k5_tflops=local_flops / (k5_ms * 1e-3) / 1e12,
               select_pass_ms=el_max / steps * 1e3 - k5_max, per_rank_ms=[x / steps * 1e3 for x in per_rank],
               flops=fl_sum, elapsed=el_max)
    return (rec, call) if want_call else rec


def api_record(comm, wl, spec, q, k, v, steps, warmup, staged_ms, with_processor):
    """The PUBLIC entry points end to end (what a diffusers pipeline calls), regime r2: the reference-shaped operator
    rectified_hunyuan_attn.rectified_block_sparse_attention and, optionally, a processor __call__ on a stand-in
    attention module (random-init projections of the HunyuanVideo width)."""
    import torch
    from rectified_spaattn_amd import rectified_hunyuan_attn as rh
    S = q.shape[2]
    num_true = wl["S_vis"] + wl["text_valid"]
    cu = [0, num_true, S]

    def op_step(_):
        rh.rectified_block_sparse_attention(q, k, v, attn_mask=None, top_k=wl["top_k"], cu_seqlens_q=cu,
                                            cu_seqlens_kv=cu, max_seqlen_q=S, max_seqlen_kv=S,
                                            block_neighbor_list=None, p_remain_rates=0.0)

    el = timed_steps(comm, op_step, steps, warmup)
    rec = dict(operator_ms=round(el / steps * 1e3, 4), staged_ms=round(staged_ms, 4),
               operator_over_staged=round(el / steps * 1e3 / staged_ms, 4))
    if with_processor:
        import types
        H, D = q.shape[1], q.shape[3]
        dim = H * D
        dev, dt = q.device, q.dtype
        g = torch.Generator(device=dev).manual_seed(7)

        def lin():
            m = torch.nn.Linear(dim, dim, bias=True, device=dev, dtype=dt)
            with torch.no_grad():
                m.weight.copy_(torch.randn(dim, dim, generator=g, device=dev, dtype=torch.float32) * dim ** -0.5)
                m.bias.zero_()
            return m

        class RMS(torch.nn.Module):   # diffusers' RMSNorm over the head dim (qk_norm="rms_norm"), as the HunyuanVideo blocks carry
            def __init__(self, d, eps=1e-6):
                super().__init__()
                self.weight = torch.nn.Parameter(torch.ones(d, device=dev, dtype=dt))
                self.eps, self.bias = eps, None

            def forward(self, x):
                dt_in = x.dtype
                var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
                x = x * torch.rsqrt(var + self.eps)
                return (x.to(self.weight.dtype) * self.weight).to(dt_in)

        attn = types.SimpleNamespace(heads=H, to_q=lin(), to_k=lin(), to_v=lin(),
                                     to_out=torch.nn.ModuleList([lin(), torch.nn.Identity()]), norm_q=RMS(D),
                                     norm_k=RMS(D), add_q_proj=None, add_k_proj=None, add_v_proj=None,
                                     norm_added_q=None, norm_added_k=None, to_add_out=None)
        proc = rh.RectifiedHunyuanVideoSpaAttnProcessor2_0("sparse", wl["top_k"], None, 0.0, 0)
        hs = torch.randn(1, wl["S_vis"], dim, generator=g, device=dev, dtype=torch.float32).to(dt)
        ehs = torch.randn(1, wl["text"], dim, generator=g, device=dev, dtype=torch.float32).to(dt)
        mask = (torch.arange(S, device=dev) < num_true)[None, None, None, :]
        ang = torch.rand(wl["S_vis"], D, generator=g, device=dev, dtype=torch.float32) * 6.2831853
        rope = (torch.cos(ang), torch.sin(ang))   # (cos, sin) tables of the visual tokens, diffusers' real form

        def proc_step(_):
            with torch.no_grad():
                proc(attn, hs, encoder_hidden_states=ehs, attention_mask=mask, image_rotary_emb=rope)

        def gemm_step(_):  # the projections alone (what the processor adds around the operator)
            with torch.no_grad():
                x = torch.cat([hs, ehs], dim=1)
                attn.to_q(x), attn.to_k(x), attn.to_v(x)
                attn.to_out[0](x[:, :wl["S_vis"]])

        from rectified_spaattn_amd import _operator as _op
        n = max(3, steps // 4)
        elp = timed_steps(comm, proc_step, n, 2)          # the fused QK-norm + RoPE producer is what runs
        elg = timed_steps(comm, gemm_step, n, 2)
        _op.FUSED_PRODUCER = False                        # the same call through the module-by-module path (PyTorch ops)
        try:
            elu = timed_steps(comm, proc_step, max(2, n // 2), 1)
            nu = max(2, n // 2)
        finally:
            _op.FUSED_PRODUCER = True
        rec.update(processor_ms=round(elp / n * 1e3, 4), projections_ms=round(elg / n * 1e3, 4),
                   processor_minus_projections_ms=round((elp - elg) / n * 1e3, 4),
                   processor_unfused_ms=round(elu / nu * 1e3, 4),
                   fused_producer_saves_ms=round(elu / nu * 1e3 - elp / n * 1e3, 4),
                   processor_what="RectifiedHunyuanVideoSpaAttnProcessor2_0 (single-stream block: projections of the "
                                  "[visual | text] sequence, RMSNorm on q / k heads, RoPE on the visual tokens, sparse operator, "
                                  "to_out) on a stand-in attention module with random weights")
    return rec


def fp8_records(comm, args, call_bf16, q, k, v, spec, wl, dev):
    """e4m3 operands beside the 2-byte ones, in the default line (BASELINE config 5 has no row of its own otherwise):
    (i) quality -- the SAME layer (headline workload, locality regime: the data with the structure of real attention maps) through
    K5-fp8 and through the 2-byte K5, relative L1 / max distance of the two outputs (same kept lists: the selection pass reads
    the 2-byte inputs in both); (ii) speed -- that layer with qkv_fp8, and BASELINE config 5 (Wan2.2-TI2V 720p 121f) with
    2-byte and with e4m3 operands."""
    import torch
    from rectified_spaattn_amd import _core
    out = {}
    del call_bf16          # (the last regime's call: may be another operating point than the locality one compared here)
    _, nbr_kind, p = REGIMES["locality"]
    nbr = make_neighbors(wl, spec, nbr_kind)
    cb = _core.StagedCall(q, k, v, spec, wl["top_k"], p, nbr)
    cb.select(); cb.attend()
    torch.cuda.synchronize()
    ref = cb.out.float()
    del cb
    for mode, key, what in ((True, "fp8_vs_bf16", "K5 on e4m3 Q/K/V/P"),
                            ("pv", "pv_vs_bf16", "the pv form: Q.K^T on the bf16 operands, only P.V on e4m3")):
        c8 = _core.StagedCall(q, k, v, spec, wl["top_k"], p, nbr, qkv_fp8=mode)

        def st8(_):
            c8.select()
            c8.attend()
        el = timed_steps(comm, st8, 5, 2)
        d = (c8.out.float() - ref).abs()
        pairs = float(c8.bufs["counts"].sum().item())
        fl = 4.0 * 128 * 128 * 128 * pairs + 4.0 * 128 * spec.q_text_valid * spec.kv_text_valid * q.shape[1]
        out[key] = dict(what=f"HunyuanVideo 720p layer, locality regime: {what} vs K5 on the bf16 operands, same kept lists",
                        rel_l1=round(float(d.sum() / ref.abs().sum()), 5), max_abs=round(float(d.max()), 4),
                        mean_abs=round(float(d.mean()), 5), out_rms=round(float(ref.pow(2).mean().sqrt()), 4),
                        fp8_layer_ms=round(el / 5 * 1e3, 4), fp8_layer_tflops=round(fl / (el / 5) / 1e12, 1))
        del c8, d
    del ref
    torch.cuda.empty_cache()
    w5 = WORKLOADS["wan22_ti2v_720p_121f"]
    s5 = make_spec(w5)
    q5, k5, v5 = gen_inputs(w5, w5["H"], 0, dev, "iid")
    rec5, calls5 = {}, {}
    for f8 in (False, "pv", True):     # the three forms timed back to back (the checks, host-bound, come after: they would let the clocks drop)
        c5 = _core.StagedCall(q5, k5, v5, s5, w5["top_k"], 0.0, None, qkv_fp8=f8, reuse_buffers=False)

        def st5(_):
            c5.select()
            c5.attend()
        el5 = timed_steps(comm, st5, 10, 3)
        pr5 = float(c5.bufs["counts"].sum().item())
        fl5 = 4.0 * 128 * 128 * 128 * pr5
        name5 = "pv" if f8 == "pv" else ("e4m3" if f8 else "bf16")
        rec5[name5] = dict(ms_per_layer=round(el5 / 10 * 1e3, 4), tflops=round(fl5 / (el5 / 10) / 1e12, 1))
        calls5[name5] = (c5, f8)
    for name5, (c5, f8) in calls5.items():
        try:      # the output of the launch just timed (check_output: the ragged last query block is one of the sampled ones)
            ck5 = check_output(c5, s5, f8)
            ck5 = {n: ck5[n] for n in ("ok", "blocks", "max_abs", "mean_abs", "vs_unquantised") if n in ck5}
        except Exception as e:  # noqa: BLE001
            ck5 = dict(ok=False, error=repr(e)[:200])
        rec5[name5]["check"] = ck5
    del calls5, c5
    rec5["what"] = ("BASELINE config 5: Wan2.2-TI2V 720p 121f, S = 27 280, 24 heads, top_k 53 (regime r2), whole layer (select + K5); `check` = "
                    "check_output of that launch (fp8 forms: against the operands the kernel multiplies, vs_unquantised beside it)")
    out["config5_wan22_ti2v"] = rec5
    return out


def load_traffic(name):
    """PMC-derived memory-side bytes per launch of the dominant kernel, only if the json was collected from the kernel
    sources that are built now (otherwise the number would silently go stale)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, f"profiles/{name} not collected yet"
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, f"profiles/{name} unreadable"
    sha = kernel_source_sha("fp8" in str(d.get("kernel", "")))
    if d.get("kernel_source_sha") != sha:
        return None, (f"profiles/{name} was collected from other kernel sources (sha {d.get('kernel_source_sha')} != "
                      f"{sha}); re-run tools/pmc_traffic.sh")
    return d.get("traffic_bytes_per_launch"), (f"L2 memory-side bytes/launch from rocprofv3 PMC passes "
                                               f"(profiles/{name}, FETCH_SIZE x2 + WRITE_SIZE); includes Infinity-Cache "
                                               f"hits; l2_hit_rate {d.get('l2_hit_rate')}")


def profiler_attached() -> bool:
    """True when this process runs under rocprofv3 / rocprofiler (tools/prof_quick.sh, the kernel-stats runs): children started
    from here would inherit the outer tool's LD_PRELOAD and ROCP_* variables."""
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
        return True
    return any(("rocprof" in os.environ.get(k, "")) for k in ("LD_PRELOAD", "HSA_TOOLS_LIB"))


def measure_traffic_live(regime: str, fp8: bool, timeout_s: int = 150, workload: str = "hunyuan_720p_128f"):
    """The dominant kernel's memory-side bytes per launch, measured NOW: three rocprofv3 counter-only passes (child processes;
    counters in passes of their own, no trace domains, the program itself behind `--`: MI355X_MICROARCH.md's recipe) over
    `tools/perf_k5.py pmc` = three launches of the same call on the same synthetic inputs.  FETCH_SIZE x2 (gfx950: it counts 64 B
    per 128-B request of a wide stream) + WRITE_SIZE, both KiB.  Returns (bytes, note) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # K5 = the walk kernel AND the combine passes that read the split-KV partials back (text rows, split tail)
    kerns = ("bsfwd_fp8" if fp8 else "bsfwd", "text_combine_kernel", "tail_combine_kernel")
    if profiler_attached():
        return None, "this process already runs under a profiler (rocprofv3 children would inherit its preload): not nested"
    env = dict(os.environ, RSA_PERF_REGIME=regime, RSA_PERF_NODENSE="1", TMPDIR="/tmp", RSA_PERF_WORKLOAD=workload)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if fp8:
        env["RSA_PERF_FP8"] = "pv" if fp8 == "pv" else "1"
    acc, walks = {}, {}
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for n, counters in enumerate((("TCC_HIT_sum", "TCC_MISS_sum"), ("FETCH_SIZE",), ("WRITE_SIZE",))):
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(td, f"p{n}"), "--",
                   sys.executable, os.path.join(ROOT, "tools", "perf_k5.py"), "pmc"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, f"rocprofv3 pass {n} did not finish: {e!r}"[:200]
            if r.returncode != 0:
                return None, f"rocprofv3 pass {n} exit {r.returncode}: {r.stderr.decode(errors='replace')[-160:]}"
            for f in glob.glob(os.path.join(td, f"p{n}", "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    which = next((i for i, kn in enumerate(kerns) if kn in name), None)
                    if which is None:
                        continue
                    # per counter: the walk kernel's dispatches, and the combine passes' totals (folded into the per-launch mean)
                    acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                    acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
                    if which == 0:
                        walks.setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
    # bytes per K5 launch = everything the walk kernel and its combine passes moved / number of walk launches
    mean = {k: sum(v.values()) / max(1, len(walks.get(k, ()))) for k, v in acc.items() if v}
    if "FETCH_SIZE" not in mean or "WRITE_SIZE" not in mean:
        return None, f"no counters for kernels {kerns} in the rocprofv3 output ({sorted(mean)})"
    tb = 2.0 * mean["FETCH_SIZE"] * 1024.0 + mean["WRITE_SIZE"] * 1024.0
    hit = mean.get("TCC_HIT_sum", 0.0) / max(mean.get("TCC_HIT_sum", 0.0) + mean.get("TCC_MISS_sum", 0.0), 1.0)
    n_l = len(walks.get("FETCH_SIZE", ()))
    return tb, (f"measured in this run: rocprofv3 counter-only passes (TCC_HIT/MISS | FETCH_SIZE | WRITE_SIZE, child processes, "
                f"{time.time() - t0:.0f} s) over {n_l} launches of the same call (the walk kernel + its text / tail combine passes); L2 "
                f"memory-side (fabric) bytes per launch = FETCH_SIZE x2 + WRITE_SIZE; includes Infinity-Cache hits; l2_hit_rate {hit:.4f}")


def dry_worker(args, comm):
    """Host logic only (gloo, no GPU): sharding, barriers, max-over-ranks timing, the JSON line."""
    from rectified_spaattn_amd import parallel
    wl = WORKLOADS[args.workload]
    H = wl["H"]
    _, H_local = parallel.head_shard(H, comm.world, comm.rank)
    spec = make_spec(wl)
    pairs = float(H_local * spec.NBv * (wl["top_k"] + spec.NB_total - spec.NBv))
    flops = 4.0 * 128 * 128 * 128 * pairs
    elapsed, busy = timed_steps(comm, lambda ev: time.sleep(0.001 * (1 + comm.rank)), args.steps, args.warmup,
                                want_busy=True)
    el_max, fl_sum, _, _, per_rank = parallel.reduce_step_stats(elapsed, flops, pairs, 0.0, comm.stat_dev, busy_s=busy)
    if comm.rank == 0:
        print(json.dumps({"metric": "dry run (host logic only, no GPU work)", "value": fl_sum / (el_max / args.steps) / 1e12,
                          "unit": "TFLOP/s", "n_gpus": comm.world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": el_max / args.steps * 1e3, "dry": True, "heads_per_gpu": H_local,
                          "per_rank_ms": [x / args.steps * 1e3 for x in per_rank],
                          "imbalance": max(per_rank) / (sum(per_rank) / len(per_rank))}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hunyuan_720p_128f", choices=list(WORKLOADS))
    ap.add_argument("--regime", default="r2", choices=list(REGIMES), help="regime of the headline value")
    ap.add_argument("--p-remain", type=float, default=None, help="override the cumulative-probability threshold (r2)")
    ap.add_argument("--neighbors", default=None, help="override the block-neighbour matrix (r2): none | gilbert | <band>")
    ap.add_argument("--qkv-fp8", nargs="?", const=True, default=False, type=lambda x: x if x == "pv" else bool(int(x)),
                    help="K5 on e4m3 images of Q/K/V (fp8 MFMA); `--qkv-fp8 pv`: the pv form (2-byte Q.K^T, e4m3 P.V); "
                         "the quantisation pass is inside the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline regime (no r1/locality/api/sustained)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC json instead of three rocprofv3 counter passes in this run")
    ap.add_argument("--via-api", action="store_true", help="(kept for old command lines: the processor timing is on by default)")
    ap.add_argument("--no-processor", action="store_true", help="skip the processor __call__ timing of the api record")
    ap.add_argument("--gather-output", action="store_true",
                    help="N > 1: report only the variant with the all-gather of O along heads inside the timed region "
                         "(default: both variants are measured and reported)")
    ap.add_argument("--gather-transports", default="rccl,p2p",
                    help="N > 1: the exchange is also timed through the library's own transports: comma list of rccl (rsa_"
                         "allgather_heads) and p2p (rsa_allgather_heads_p2p); each is reported beside the torch.distributed one "
                         "('' = none)")
    ap.add_argument("--dry", action="store_true", help="host logic only (gloo, CPU): for the multi-process tests")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    comm = Comm(args.dry)
    if args.gpus != comm.world:
        if comm.rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={comm.world}; using {comm.world}", file=sys.stderr)
        args.gpus = comm.world
    if args.dry:
        dry_worker(args, comm)
        comm.close()
        return

    import torch
    from rectified_spaattn_amd import parallel
    world, rank, dev = comm.world, comm.rank, comm.dev
    wl = WORKLOADS[args.workload]
    D, H = wl.get("D", 128), wl["H"]
    head0, H_local = parallel.head_shard(H, world, rank)
    S = wl["S_vis"] + wl["text"]
    spec = make_spec(wl)

    main_regime = args.regime
    q, k, v = gen_inputs(wl, H_local, head0, dev, REGIMES[main_regime][0], D=D)
    rec, call = run_regime(comm, args, wl, main_regime, q, k, v, spec, args.steps, args.warmup, want_call=True)
    # the headline launch's own output, checked on every rank (a rank with a wrong O fails the whole line)
    comm.local_sync()
    try:
        check = check_output(call, spec, args.qkv_fp8)
    except Exception as e:  # noqa: BLE001
        check = dict(ok=False, error=repr(e)[:300])
    check["ok"] = comm.all_ok(bool(check.get("ok")))
    sample = capture_sparse_sample(call) if (world == 1 and not args.no_cpu_baseline) else None

    extras = {}
    gather = None
    if world > 1:  # the optional exchange step at the layer boundary: all-gather of O along the head axis
        # Self-check of every exchange (the first real N-GPU run is the driver's, with no one watching): each rank publishes
        # per-head fingerprints of its LOCAL O (fp64 sum, sum of squares, 8 whole rows: parallel.head_checksums); after each
        # transport's timed steps every rank recomputes them from the buffer that transport gathered; a transport whose buffer
        # does not reproduce them on EVERY rank is reported as {"error": ...}, never as a time.
        call.select(); call.attend()
        comm.local_sync()
        want = parallel.exchange_checksums(parallel.head_checksums(call.out, D), comm.stat_dev)
        last = {}

        def verified(full):
            msg = parallel.verify_gathered(full, want, D)
            ok = comm.all_ok(msg is None)
            return ok, (msg or ("the gathered buffer is wrong on another rank" if not ok else None))

        def gstep(_):
            call.select()
            call.attend()
            last["torch"] = parallel.gather_heads(call.out)
        # (a failure inside a collective cannot be caught rank by rank -- the peers would hang in it --, so these timed
        # stages run unguarded: an exception takes the job down through torch.distributed.run.  What CAN fail on one rank
        # alone, the set-up of the library's own transports below, is agreed on across ranks before anyone enters it.)
        elg = timed_steps(comm, gstep, args.steps, max(1, args.warmup))
        elg, _, _, _, per_rank_g = parallel.reduce_step_stats(elg, 0.0, 0.0, 0.0, comm.stat_dev)
        try:
            rccl_ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl_ver = "unknown"
        gather = dict(ranks=world, rccl_version=rccl_ver, ms_per_step=round(elg / args.steps * 1e3, 4),
                      value=round(rec["flops"] / (elg / args.steps) / 1e12, 3),
                      bytes_per_rank=int(call.out.numel() * call.out.element_size()),
                      transport="torch.distributed all_gather (RCCL)")
        comm.local_sync()
        ok_t, why_t = verified(last.pop("torch"))
        gather["verified"] = ok_t
        if not ok_t:
            gather = dict(ranks=world, rccl_version=rccl_ver, error=f"torch.distributed all_gather: {why_t}",
                          bytes_per_rank=gather["bytes_per_rank"], transport=gather["transport"], verified=False)
        for tr in [t for t in args.gather_transports.split(",") if t]:
            B_, S_, Hl_, D_ = call.out.shape
            hg, err = None, None
            try:   # the library's own transports (C-ABI): the set-up may fail on one rank (librccl missing, IPC refused)
                hg = parallel.HeadGather(B_, S_, Hl_, D_, call.out.dtype, dev, transport=tr)
            except Exception as e:  # noqa: BLE001
                err = repr(e)[:300]
            if not comm.all_ok(hg is not None):   # all ranks skip the stage together
                if hg is not None:
                    hg.close()
                gather[tr] = {"error": err or "set-up failed on another rank"}
                continue

            def hstep(_):
                call.select()
                call.attend()
                last[tr] = hg.gather(call.out)
            elh = timed_steps(comm, hstep, args.steps, max(1, args.warmup))
            elh, _, _, _, _ = parallel.reduce_step_stats(elh, 0.0, 0.0, 0.0, comm.stat_dev)
            # a p2p wait that gave up (a peer's slab did not arrive within 4 s) leaves stale data behind a valid-looking time:
            # the time-out word is read on every rank and the stage is reported as failed if any rank saw one
            terr = None
            try:
                hg.check()
            except Exception as e:  # noqa: BLE001
                terr = repr(e)[:200]
            if comm.all_ok(terr is None):
                ok_h, why_h = verified(last.pop(tr))       # (synchronised by hg.check() / the closing barrier of timed_steps)
                if ok_h:
                    gather[tr] = dict(ms_per_step=round(elh / args.steps * 1e3, 4),
                                      value=round(rec["flops"] / (elh / args.steps) / 1e12, 3), verified=True)
                    cr = None
                    try:
                        cr = hg.comm_ranks()
                    except Exception:  # noqa: BLE001  (an RCCL without ncclCommCount: the record just lacks the field)
                        pass
                    if cr is not None:
                        gather[tr]["comm_ranks"] = cr
                else:
                    gather[tr] = {"error": why_h, "verified": False}
            else:
                gather[tr] = {"error": terr or "a wait timed out on another rank"}
            hg.close()
    if world == 1 and not args.no_extras:
        # sustained: >= 6 s of back-to-back steps of the headline regime, with the clock / power the chip holds meanwhile
        n_sus = max(args.steps, int(6.5 / max(rec["ms_per_step"] * 1e-3, 1e-4)))

        def sstep(_):
            call.select()
            call.attend()
        with SmiSampler() as smi:
            els = timed_steps(comm, sstep, n_sus, 0)
        extras["sustained"] = dict(steps=n_sus, seconds=round(els, 3), ms_per_step=round(els / n_sus * 1e3, 4),
                                   value=round(rec["flops"] / (els / n_sus) / 1e12, 3), smi=smi.summary())
        # box calibration: the SAME kernel in dense mode on a fixed 16k x 16k x 24-head problem (L2-resident K/V): devices
        # differ by several percent on this loop, so every line carries its own reference point
        try:
            from rectified_spaattn_amd import _core as _c
            gq = torch.Generator(device=dev).manual_seed(11)
            qd = torch.randn(1, 24, 16384, D, device=dev, generator=gq).to(torch.bfloat16)
            for _ in range(2):
                _c.dense_attention(qd, qd, qd, qkv_fp8=args.qkv_fp8)
            evd = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            for a_, b_ in evd:
                a_.record(); _c.dense_attention(qd, qd, qd, qkv_fp8=args.qkv_fp8); b_.record()
            torch.cuda.synchronize()
            msd = sorted(a_.elapsed_time(b_) for a_, b_ in evd)[len(evd) // 2]
            extras["box_ref"] = dict(what="dense attention 1 x 24 x 16384 x 128 bf16 inputs through the same K5 kernel "
                                          "(fp8 line: incl. its quantisation pass), median of 8",
                                     ms=round(msd, 4), tflops=round(4.0 * 16384 * 16384 * D * 24 / (msd * 1e-3) / 1e12, 1))
            del qd
        except Exception as e:  # noqa: BLE001
            extras["box_ref"] = {"error": repr(e)[:200]}
        if not args.qkv_fp8 and wl["variant"] == "hunyuan" and main_regime == "r2":
            extras["api"] = api_record(comm, wl, spec, q, k, v, args.steps, args.warmup, rec["ms_per_step"],
                                       not args.no_processor)
        regs = {}
        cur_cent = REGIMES[main_regime][0]     # which centroid model q, k, v currently hold
        for rg in REGIMES:
            if rg == main_regime:
                continue
            if REGIMES[rg][0] != cur_cent:
                cur_cent = REGIMES[rg][0]
                del call
                q = k = v = None
                torch.cuda.empty_cache()
                q, k, v = gen_inputs(wl, H_local, head0, dev, REGIMES[rg][0], D=D)
                r2, call = run_regime(comm, args, wl, rg, q, k, v, spec, max(5, args.steps // 2), 2, want_call=True)
            else:
                r2 = run_regime(comm, args, wl, rg, q, k, v, spec, max(5, args.steps // 2), 2)
            peak_ = k5_peak(args.qkv_fp8)
            regs[rg] = dict(neighbors=r2["neighbors"], p_remain=r2["p_remain"], top_k=r2["top_k"],
                            kept_block_fraction=round(r2["kept_block_fraction"], 4),
                            ms_per_step=round(r2["ms_per_step"], 4), value=round(r2["value"], 3),
                            k5_ms=round(r2["k5_ms"], 4), k5_tflops=round(r2["k5_tflops"], 2),
                            k5_frac=round(r2["k5_tflops"] / peak_, 4), select_pass_ms=round(r2["select_pass_ms"], 4))
            tb, _ = load_traffic(f"r06_k5_traffic_{rg}{traffic_suffix(args.qkv_fp8)}.json")
            regs[rg]["traffic"] = tb
        extras["regimes"] = regs
        if not args.qkv_fp8 and args.workload == "hunyuan_720p_128f":
            # (q, k, v now hold the spatially smooth inputs of the last regimes; fp8_records builds its own locality call)
            try:
                extras["fp8"] = fp8_records(comm, args, call, q, k, v, spec, wl, dev)
            except Exception as e:  # noqa: BLE001
                extras["fp8"] = {"error": repr(e)[:300]}

    if rank != 0:
        comm.close()
        if not check.get("ok"):
            sys.exit(3)
        return

    peak = k5_peak(args.qkv_fp8)
    tname = f"r06_k5_traffic_{main_regime}{traffic_suffix(args.qkv_fp8)}.json"
    traffic, tnote = (None, "N > 1: traffic is collected at N = 1")
    if world == 1:
        traffic, tnote = load_traffic(tname) if args.workload == "hunyuan_720p_128f" else (None, "no committed traffic json for this workload")
        if not args.no_extras and not args.no_live_traffic and not comm.one_device:
            torch.cuda.synchronize()
            live, lnote = measure_traffic_live(main_regime, args.qkv_fp8, workload=args.workload)
            if live is not None:
                committed = traffic
                traffic, tnote = live, lnote + (f"; committed profiles/{tname}: {committed:.4g}" if committed else "")
                for rg, rr in (extras.get("regimes") or {}).items():      # the other regimes' launches, the same way
                    lv, ln = measure_traffic_live(rg, args.qkv_fp8, workload=args.workload)
                    if lv is not None:
                        rr["traffic"], rr["traffic_note"] = lv, "measured in this run; " + ln.split("; ")[-1]
                        rr["l2_hit_rate"] = float(ln.rsplit("l2_hit_rate ", 1)[-1])
            else:
                tnote = f"live measurement failed ({lnote}); " + tnote
    per_rank_ms = [round(x, 3) for x in rec["per_rank_ms"]]
    res = {
        "metric": "attention-layer TFLOPs/sec (rectified block-sparse attention, HunyuanVideo seq~120k d=128 bf16)",
        "value": round(rec["value"], 3), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(rec["ms_per_step"], 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16 Q.K^T + fp8_e4m3 P.V" if args.qkv_fp8 == "pv" else ("fp8_e4m3" if args.qkv_fp8 else "bf16"),
        "data": "synthetic (counter-based SplitMix64/Box-Muller generator on the device, seed 20251212 + head)",
        "config": {"workload": f"{args.workload}: B=1 H={H} S={S} ({wl['S_vis']} visual + {wl['text']} text, "
                               f"{wl['text_valid']} valid) D={D}, top_k={regime_top_k(wl, main_regime)}, regime={main_regime} "
                               f"(p_remain={rec['p_remain']}, neighbors={rec['neighbors']})",
                   "kept_block_fraction": round(rec["kept_block_fraction"], 4), "heads_per_gpu": H_local,
                   "dense_equivalent_tflops": round(4.0 * S * S * D * H / (rec["elapsed"] / args.steps) / 1e12, 1),
                   "parallelism": f"head-shard x{world}", "world_size": (comm.dist.get_world_size() if comm.dist is not None else 1),
                   "per_rank_ms": per_rank_ms,
                   "imbalance": round(max(per_rank_ms) / (sum(per_rank_ms) / len(per_rank_ms)), 4),
                   "gather_output": gather},
        "roofline": {"kernel": ("bsfwd_fp8_kernel<..., HYB> (K5 block_sparse_fwd_fp8pv; peak = both products at their own dense peaks: "
                                "half the FLOPs at 2.5, half at 5 PFLOP/s)") if args.qkv_fp8 == "pv" else
                     "bsfwd_fp8_kernel (K5 block_sparse_fwd_fp8)" if args.qkv_fp8 else
                     f"bsfwd64_kernel<bf16_tag,..., D = {D}> (K5 block_sparse_fwd, 64 rows per wave)", "bound": "mfma",
                     "achieved": round(rec["k5_tflops"], 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(rec["k5_tflops"] / peak, 4), "traffic": traffic, "traffic_note": tnote,
                     "compulsory_bytes": 4 * 2 * H_local * S * D,
                     "k5_ms": round(rec["k5_ms"], 4), "select_pass_ms": round(rec["select_pass_ms"], 4),
                     "select_pass_tbps": round(2.0 * H_local * D * (wl["S_vis"] + 2 * S) /
                                               max(rec["select_pass_ms"], 1e-6) / 1e9, 3)},
    }
    if comm.one_device:
        res["config"]["one_device_test"] = "all ranks time-slice cuda:0 over gloo: functional evidence, not a measurement"
    if args.gather_output and gather is not None and "value" in gather:  # headline = the gather-inclusive variant
        res["value"], res["ms_per_step"] = gather["value"], gather["ms_per_step"]
    res["check"] = check
    res.update(extras)
    if traffic is not None:   # the second wall: memory-side (Infinity Fabric / Infinity Cache) bytes per second under K5
        res["roofline"]["traffic_tbps"] = round(traffic / max(rec["k5_ms"], 1e-6) / 1e9, 3)
    if "sustained" in extras:
        res["roofline"]["clock"] = extras["sustained"].get("smi")
    if "box_ref" in extras:
        res["roofline"]["box_ref"] = extras["box_ref"]
    if world == 1:
        if args.no_cpu_baseline:
            res["cpu_baseline"] = None
        else:
            cb = cpu_baseline_dense(S, D)
            try:
                cb["sparse"] = cpu_baseline_sparse(sample, spec, D)
            except Exception as e:  # noqa: BLE001  (the dense column must survive a failure of the sampled port)
                cb["sparse"] = {"error": repr(e)}
            res["cpu_baseline"] = cb
    print(json.dumps(res))
    comm.close()
    if not check.get("ok"):
        print(f"bench.py: the output check FAILED: {check}", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
