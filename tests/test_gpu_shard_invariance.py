"""Head sharding: a rank that holds heads [h0, h0+n) computes, for those heads, exactly the unsharded call's
bitmask, kept lists, R and compensation -- always (no statistic crosses heads: softmax, sort and the cumulative rule are per
(b, h, query block); top_k, p and the neighbour matrix are head-independent) -- and exactly its O wherever K5 planned the row's
walk the same way in both launches.  Two plans depend on the size of the launch (rsa_attn.hip::launch_attn): the tail split
(which query blocks of the last, partial generation are walked in pieces) and the piece count of the dense text rows (32 on
short grids, 16 otherwise); the rows they touch agree within rounding.  `parallel.set_shard_invariant(True)` (C:
rsa_set_shard_invariant) plans both per head: byte-identical O everywhere, which the second test pins at a shape where one
side splits.  The first test's shape (8 x 48 = 384 workgroups) is below one generation: nothing is ever split there."""
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


def _tail_blocks(H, spec):
    """[H, NBv] bool: the query blocks a launch of H heads walks in pieces (the plan of rsa_plan_tail_split)."""
    NBp = (spec.NBv + 7) // 8 * 8
    n_sparse = H * NBp
    full, T = divmod(n_sparse, 512)
    touched = torch.zeros(H, spec.NBv, dtype=torch.bool)
    if full < 1 or T == 0 or min(512 // T, 4) < 2:
        return touched
    for vv in range(full * 512, n_sparse):
        h, j = divmod(vv, NBp)
        qb = (j & 7) * (NBp // 8) + (j >> 3)
        if qb < spec.NBv:
            touched[h, qb] = True
    return touched


def test_head_shard_where_one_side_splits_its_tail():
    """8 heads x 72 query blocks = 576 workgroups: the unsharded launch splits the 64 walks of its second generation 4 ways;
    a rank with 4 heads (288 workgroups, less than a generation) splits nothing.  Default: the selection results and every
    block neither launch split are byte-identical, the split blocks agree within two output ulps.  With
    set_shard_invariant(True): byte-identical everywhere (the invariant the first test documents, at a production-like
    shape)."""
    from bench import gen_qkv
    from rectified_spaattn_amd import _core, parallel
    H, D, world, nbv = 8, 128, 2, 72
    S = nbv * 128
    dev = torch.device(DEV)
    q, k, v = gen_qkv(H, 0, S, S, D, dev, seed=23)
    spec = _core.LayoutSpec.wan(S, 2)
    touched_full = _tail_blocks(H, spec)
    assert int(touched_full.sum()) == 64 and int(_tail_blocks(H // world, spec).sum()) == 0
    for invariant in (False, True):
        prev = parallel.set_shard_invariant(invariant)
        try:
            full, fb = _core.rectified_attention(q, k, v, spec, 12, 0.05, None, return_parts=True, shape_xfuse=True)
            for rank in range(world):
                h0, hl = parallel.head_shard(H, world, rank)
                part, pb = _core.rectified_attention(q[:, h0:h0 + hl], k[:, h0:h0 + hl], v[:, h0:h0 + hl], spec, 12, 0.05,
                                                     None, return_parts=True, shape_xfuse=True)
                torch.cuda.synchronize()
                for name in ("bitmask", "counts", "R", "comp"):
                    assert torch.equal(pb[name], fb[name][h0:h0 + hl]), f"rank {rank}: {name} differs"
                want = full[:, :, h0:h0 + hl]
                if invariant:
                    assert torch.equal(part, want), f"rank {rank}: O differs with shard invariance on"
                    continue
                diff = (part.float() - want.float()).abs()[0].reshape(nbv, 128, hl, D).amax(dim=(1, 3)).t().cpu()   # [hl, NBv]
                tf = touched_full[h0:h0 + hl]
                assert (diff[~tf] == 0).all(), f"rank {rank}: a block neither launch split differs"
                assert float(diff.max()) <= 2 * 2.0 ** -7 * max(1.0, float(want.float().abs().max()))
                if tf.any():
                    assert (diff[tf] > 0).any(), "the split blocks came out byte-identical: is the tail split still planned?"
        finally:
            parallel.set_shard_invariant(prev)


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
    # every exchange checked itself: the fingerprints each rank published for its heads reappear in the gathered buffers
    assert g.get("verified") is True and g["p2p"].get("verified") is True, g
    assert rec["config"]["world_size"] == 2


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


def test_p2p_gather_eight_processes_one_device():
    """Software pre-flight of the first 8-GPU run: the same worker with EIGHT processes on cuda:0 -- 7 IPC peers
    per rank, 8 flags per state block, six stream-ordered gathers alternating between the two result buffers under uneven load,
    the exchange object constructed and destroyed twice per process."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29641",
                        os.path.join(here, "_ipc_gather_worker.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "IPC_GATHER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_eight_ranks_on_one_device():
    """`python bench.py --gpus 8` as the driver starts it, at a reduced shape (8 heads x 8 448 tokens, one head per rank), every
    rank on cuda:0: spawn, head shards, per-rank timing, the output check on every rank, the torch exchange and BOTH library
    transports.  p2p must run and verify across the eight processes; RCCL refuses eight ranks on one device -- its set-up failure must
    be agreed on by all ranks and reported as an error record, never as a hang or a time."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSA_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "tiny8", "--steps", "4",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--gather-transports", "rccl,p2p"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 8 and rec["config"]["heads_per_gpu"] == 1 and rec["config"]["world_size"] == 8
    assert len(rec["config"]["per_rank_ms"]) == 8 and all(ms > 0 for ms in rec["config"]["per_rank_ms"])
    assert rec["check"]["ok"] is True, rec["check"]
    g = rec["config"]["gather_output"]
    assert g.get("verified") is True, g
    assert g["p2p"].get("verified") is True and g["p2p"]["ms_per_step"] > 0, g
    assert ("error" in g["rccl"]) or g["rccl"].get("verified") is True, g


@pytest.mark.parametrize("transport", ["rccl", "p2p"])
def test_head_gather_twice_in_one_process(transport):
    """Construct, use and destroy the exchange object twice (a pipeline that rebuilds its processors does): nothing of the first one
    -- the RCCL communicator, the fine-grained state block, its flags -- may leak into the second."""
    from rectified_spaattn_amd import parallel
    B, S, Hl, D = 1, 777, 2, 128
    for n in range(2):
        x = torch.randn(B, S, Hl, D, device=DEV).to(torch.bfloat16)
        g = parallel.HeadGather(B, S, Hl, D, torch.bfloat16, torch.device(DEV), transport=transport)
        try:
            for i in range(3):
                full = g.gather(x + i).clone()
            torch.cuda.synchronize()
            g.check()
            assert torch.equal(full, (x + 2).reshape(B, S, Hl * D)), (transport, n)
        finally:
            g.close()
