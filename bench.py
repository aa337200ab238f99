#!/usr/bin/env python3
"""Attention-layer benchmark for the rectified block-sparse attention path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the whole hot path (K1 pool_stats .. K5 block_sparse_fwd) over one synthetic
HunyuanVideo-720p attention call: B=1, H=24, S=115 200 visual + 256 text (200 valid), D=128, bf16, top_k = 90
of 900 visual blocks (10 % kept), inputs already resident in HBM.  With N GPUs the 24 heads are sharded
contiguously (24/N per rank, strong scaling, no data-path collective); time = max over ranks.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
MFMA_FP8_PEAK_TFLOPS = 5000.0   # MI355X dense fp8 (same table)

WORKLOADS = {
    # name: (H, S_visual, text_pad, text_valid, top_k, variant)
    "hunyuan_720p_128f": dict(H=24, S_vis=115200, text=256, text_valid=200, top_k=90, variant="hunyuan",
                              latent=(32, 45, 80)),
    "flux_4096": dict(H=24, S_vis=65536, text=512, text_valid=512, top_k=51, variant="flux", latent=(1, 256, 256)),
    "wan21_720p_81f": dict(H=40, S_vis=75600, text=0, text_valid=0, top_k=147, variant="wan", ffb=28,
                           latent=(21, 45, 80)),
    "wan22_ti2v_720p_121f": dict(H=24, S_vis=27280, text=0, text_valid=0, top_k=53, variant="wan", ffb=6,
                                 latent=(31, 22, 40)),
    "tiny": dict(H=4, S_vis=4096, text=256, text_valid=200, top_k=6, variant="hunyuan", latent=(4, 32, 32)),
}


def gen_qkv(H_local, head0, S, S_vis, D, dev, seed=20251212, c=1.5, sigma=0.5):
    """Structured synthetic Q/K/V: per 128-token block a shared centroid for Q and K, V ~ N(0,1);
    generated on device per head (seed + global head index)."""
    q = torch.empty(1, H_local, S, D, dtype=torch.bfloat16, device=dev)
    k = torch.empty_like(q)
    v = torch.empty_like(q)
    nb = (S + 127) // 128
    for hl in range(H_local):
        g = torch.Generator(device=dev)
        g.manual_seed(seed + head0 + hl)
        u = torch.randn(nb, D, generator=g, device=dev)
        cent = u.repeat_interleave(128, dim=0)[:S] * c
        q[0, hl] = (cent + sigma * torch.randn(S, D, generator=g, device=dev)).to(torch.bfloat16)
        k[0, hl] = (cent + sigma * torch.randn(S, D, generator=g, device=dev)).to(torch.bfloat16)
        v[0, hl] = torch.randn(S, D, generator=g, device=dev).to(torch.bfloat16)
    return q, k, v


def cpu_baseline(S, D, budget_s=25.0):
    """The reference's CPU dense path (fullattn mode='torch' = SDPA, attn.py:101-106) restated in
    oracle/dense_cpu.py, timed on this host's cores on a bounded sample: 1 head, as many query rows as fit the
    time budget, all S keys, bf16."""
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
    # scale the sample towards the budget (bounded by the full S rows)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="hunyuan_720p_128f", choices=list(WORKLOADS))
    ap.add_argument("--p-remain", type=float, default=0.0,
                    help="cumulative-probability threshold; 0 keeps exactly top_k visual blocks per row")
    ap.add_argument("--neighbors", default="none",
                    help="block-neighbour matrix: 'none' (exactly top_k kept: the 10 %% regime), 'gilbert' (true "
                         "26-neighbourhood along the Gilbert curve of the workload's latent), or an int band width")
    ap.add_argument("--qkv-fp8", action="store_true",
                    help="K5 on e4m3 images of Q/K/V (fp8 MFMA); the quantisation pass is inside the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-output", action="store_true", help="all-gather O along heads inside the timed region")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        args.gpus = world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from rectified_spaattn_amd import _core, parallel, synth

    wl = WORKLOADS[args.workload]
    D = 128
    H = wl["H"]
    head0, H_local = parallel.head_shard(H, world, rank)
    S = wl["S_vis"] + wl["text"]
    if wl["variant"] == "hunyuan":
        num_true = wl["S_vis"] + wl["text_valid"]
        spec = _core.LayoutSpec.hunyuan(S, num_true)
    elif wl["variant"] == "flux":
        spec = _core.LayoutSpec.flux(S, wl["text"])
    else:
        spec = _core.LayoutSpec.wan(S, wl.get("ffb", 0))
    q, k, v = gen_qkv(H_local, head0, S, wl["S_vis"], D, dev)
    if args.neighbors == "none":
        nbr = None
    elif args.neighbors == "gilbert":
        from rectified_spaattn_amd.utils import jenga_gilbert
        nbr = jenga_gilbert.gilbert_block_neighbor_mapping(*wl["latent"], axis_order=("w", "h", "t"))
    else:
        nbr = torch.from_numpy(synth.banded_neighbors(spec.NBv, int(args.neighbors)))
    top_k = wl["top_k"]

    stages = _core.StagedCall(q, k, v, spec, top_k, args.p_remain, nbr, qkv_fp8=args.qkv_fp8)

    def step(ev=None):
        stages.select()
        if args.qkv_fp8:
            stages.quantize()
        if ev is not None:
            ev[0].record()
        stages.attend()
        if ev is not None:
            ev[1].record()
        if args.gather_output and world > 1:
            parallel.gather_heads(stages.out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(evs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k5_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, args.steps)

    # work actually done (from the mask the selection kernels produced)
    counts = stages.bufs["counts"].sum().item()  # kept (q-block, k-block) pairs over local heads
    pair_flops = 4.0 * D * 128 * 128
    text_flops = 4.0 * D * spec.q_text_valid * spec.kv_text_valid * H_local
    local_flops = pair_flops * counts + text_flops
    k5_ms_local = k5_ms
    elapsed, total_flops, total_pairs, k5_ms, per_rank_s = parallel.reduce_step_stats(
        elapsed, local_flops, float(counts), k5_ms, dev)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_flops / (elapsed / args.steps) / 1e12
    kept_frac = total_pairs / (H * spec.NBv * spec.NB_total)
    k5_flops_local = local_flops  # rank-0 launch
    achieved = k5_flops_local / (k5_ms_local * 1e-3) / 1e12
    # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE with
    # the gfx950 x2 correction; tools/pmc_to_json.py) of this same command; it cannot be collected inside this process.
    traffic = None
    tname = "r01_k5_fp8_traffic.json" if args.qkv_fp8 else "r01_k5_traffic.json"
    tfile = os.path.join(ROOT, "profiles", tname)
    if args.workload == "hunyuan_720p_128f" and args.neighbors == "none" and world == 1 and os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    peak = MFMA_FP8_PEAK_TFLOPS if args.qkv_fp8 else MFMA_BF16_PEAK_TFLOPS
    res = {
        "metric": "attention-layer TFLOPs/sec (rectified block-sparse attention, HunyuanVideo seq~120k d=128 bf16)",
        "value": round(value, 3), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "fp8_e4m3" if args.qkv_fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"{args.workload}: B=1 H={H} S={S} ({wl['S_vis']} visual + {wl['text']} text, "
                               f"{wl['text_valid']} valid) D={D}, top_k={top_k}, p_remain={args.p_remain}, "
                               f"neighbors={args.neighbors}",
                   "kept_block_fraction": round(kept_frac, 4), "heads_per_gpu": H_local,
                   "dense_equivalent_tflops": round(4.0 * S * S * D * H / (elapsed / args.steps) / 1e12, 1),
                   "gather_output": bool(args.gather_output), "parallelism": f"head-shard x{world}",
                   "per_rank_ms": [round(x / args.steps * 1e3, 3) for x in per_rank_s]},
        "roofline": {"kernel": "bsfwd_fp8_kernel<2> (K5 block_sparse_fwd_fp8)" if args.qkv_fp8 else
                     "bsfwd_kernel<128,bf16_tag,4,1,258> (K5 block_sparse_fwd)", "bound": "mfma",
                     "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "traffic_note": f"L2 memory-side bytes/launch from rocprofv3 PMC (profiles/{tname}); "
                                     "includes Infinity-Cache hits; compulsory Q+K+V+O = 2.84e9",
                     "k5_ms": round(k5_ms_local, 4), "select_pass_ms": round(ms_per_step - k5_ms, 4)},  # + the fp8 quantisation pass with --qkv-fp8
    }
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(S, D)
    elif world == 1:
        res["cpu_baseline"] = None
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
