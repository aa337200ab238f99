"""Head sharding: a rank that holds heads [h0, h0+n) computes, for those heads, exactly the bytes the
unsharded call computes -- bitmask, kept lists, R, compensation and O.  (No statistic crosses heads: softmax, sort and
the cumulative rule are per (b, h, query block); top_k, p and the neighbour matrix are head-independent.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("layout", ["hunyuan", "wan"])
def test_head_shard_equals_full_run(layout):
    from bench import gen_qkv
    from rectified_spaattn_amd import _core, parallel, synth
    H, D, world = 8, 128, 4
    S = 48 * 128 + (256 if layout == "hunyuan" else 37)
    dev = torch.device(DEV)
    q, k, v = gen_qkv(H, 0, S, S, D, dev, seed=11)
    spec = _core.LayoutSpec.hunyuan(S, S - 56) if layout == "hunyuan" else _core.LayoutSpec.wan(S, 3)
    nbr = torch.from_numpy(synth.banded_neighbors(spec.NBv, 2))
    full, fb = _core.rectified_attention(q, k, v, spec, 6, 0.3, nbr, return_parts=True, shape_xfuse=True)
    for rank in range(world):
        h0, hl = parallel.head_shard(H, world, rank)
        # the rank generates ITS heads itself (seed + global head index) and never sees the others
        ql, kl, vl = gen_qkv(hl, h0, S, S, D, dev, seed=11)
        assert torch.equal(ql, q[:, h0:h0 + hl])
        part, pb = _core.rectified_attention(ql, kl, vl, spec, 6, 0.3, nbr, return_parts=True, shape_xfuse=True)
        torch.cuda.synchronize()
        assert torch.equal(part, full[:, :, h0:h0 + hl]), f"rank {rank}: O differs"
        for name in ("bitmask", "counts", "R", "comp", "probs", "w"):
            assert torch.equal(pb[name], fb[name][h0:h0 + hl]), f"rank {rank}: {name} differs"
        cnt = pb["counts"]
        valid = torch.arange(spec.NB_total, device=dev)[None, None, :] < cnt[..., None]
        assert torch.equal(torch.where(valid, pb["cols"], -1), torch.where(valid, fb["cols"][h0:h0 + hl], -1))


@pytest.mark.parametrize("transport", ["rccl", "p2p"])
def test_head_gather_transports_single_rank(transport):
    """rsa_allgather_heads (RCCL) / rsa_allgather_heads_p2p through the C-ABI on the one GPU of this box: a world of one
    rank must reproduce the local tensor in the [B, S, H*D] layout (ncclAllGather + unpack kernel, resp. the 2-D copy)."""
    from rectified_spaattn_amd import parallel
    B, S, Hl, D = 1, 1000, 3, 128
    x = torch.randn(B, S, Hl, D, device=DEV).to(torch.bfloat16)
    g = parallel.HeadGather(B, S, Hl, D, torch.bfloat16, torch.device(DEV), transport=transport)
    try:
        full = g.gather(x)
        torch.cuda.synchronize()
        assert full.shape == (B, S, Hl * D) and torch.equal(full, x.reshape(B, S, Hl * D))
    finally:
        g.close()


def test_bench_two_ranks_on_one_device():
    """The N > 1 bench path end to end on hardware: `python bench.py --gpus 2` starts its own two ranks (as the driver
    runs it), both compute their 12-head shard on cuda:0 (RSA_BENCH_ONE_DEVICE test hook, gloo control plane), rank 0
    prints the one JSON line; the p2p exchange transport runs across the two processes through HIP IPC."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSA_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline", "--gather-transports", "p2p"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["heads_per_gpu"] == 12
    assert len(rec["config"]["per_rank_ms"]) == 2 and all(ms > 0 for ms in rec["config"]["per_rank_ms"])
    assert rec["value"] > 0 and "one_device_test" in rec["config"]
    g = rec["config"]["gather_output"]
    assert g is not None and "p2p" in g and "ms_per_step" in g["p2p"], g


def test_p2p_gather_two_processes_one_device():
    """rsa_allgather_heads_p2p across two PROCESSES (HIP IPC handles exchanged over gloo), both on cuda:0: content check
    of the gathered rows on every rank, three consecutive gathers (tests/_ipc_gather_worker.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(here, "_ipc_gather_worker.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "IPC_GATHER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
