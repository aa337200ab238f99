"""world_size-2 gloo test (CPU) of the multi-GPU host logic: head sharding, output all-gather order, and the
max-over-ranks / sum-of-work reduction bench.py reports."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rectified_spaattn_amd import parallel


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, B, S, D = 6, 1, 5, 4
        full = torch.arange(B * S * H * D, dtype=torch.float32).view(B, S, H, D)
        h0, hl = parallel.head_shard(H, world, rank)
        local = full[:, :, h0:h0 + hl].contiguous()
        got = parallel.gather_heads(local)
        ok_gather = torch.equal(got, full.reshape(B, S, H * D))
        # the self-check bench.py --gpus N runs on every exchange: fingerprints of the LOCAL heads, published over the control
        # plane, must reappear in the gathered buffer -- and a buffer with a stale / swapped / partly written slab must not pass
        g = torch.Generator().manual_seed(3)
        big = torch.randn(2, 64, H, 8, generator=g).to(torch.bfloat16)
        mine = big[:, :, h0:h0 + hl].contiguous()
        want = parallel.exchange_checksums(parallel.head_checksums(mine, 8), "cpu")
        gathered = parallel.gather_heads(mine)
        ok_chk = parallel.verify_gathered(gathered, want, 8) is None
        swapped = gathered.view(2, 64, H, 8)[:, :, [3, 4, 5, 0, 1, 2]].reshape(2, 64, H * 8)      # the ranks' slabs exchanged
        stale = gathered.clone(); stale.view(2, 64, H, 8)[1, 40:, 4] = 0                          # the tail of one head's rows missing
        onebit = gathered.clone().view(torch.int16); onebit[0, 7, 5 * 8 + 3] ^= 1                   # one flipped bit outside the sampled rows
        ok_neg = all(isinstance(parallel.verify_gathered(x, want, 8), str)
                     for x in (swapped, stale, onebit.view(torch.bfloat16), gathered[:, :, :-8]))
        el, fl, pr, k5, per = parallel.reduce_step_stats(1.0 + rank, 10.0 * (rank + 1), 3.0, 0.5 + rank, "cpu")
        q.put((rank, ok_gather and ok_chk and ok_neg, el, fl, pr, k5, per))
    finally:
        dist.destroy_process_group()


def test_head_shard_gather_and_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, el, fl, pr, k5, per in res:
        assert ok, "all-gather must reproduce the unsharded [B,S,H*D] layout"
        assert el == 2.0 and fl == 30.0 and pr == 6.0 and k5 == 1.5 and per == [1.0, 2.0]


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, B, S, D = 24, 1, 9, 8                      # the headline's 24 heads over 8 ranks: 3 per rank
        g = torch.Generator().manual_seed(5)
        full = torch.randn(B, S, H, D, generator=g).to(torch.bfloat16)
        h0, hl = parallel.head_shard(H, world, rank)
        mine = full[:, :, h0:h0 + hl].contiguous()
        got = parallel.gather_heads(mine)
        ok = torch.equal(got, full.reshape(B, S, H * D))
        want = parallel.exchange_checksums(parallel.head_checksums(mine, D), "cpu")
        ok = ok and tuple(want.shape)[0] == H and parallel.verify_gathered(got, want, D) is None
        # two ranks' slabs exchanged (ranks 2 and 5): the fingerprints must notice
        perm = list(range(H)); perm[6:9], perm[15:18] = perm[15:18], perm[6:9]
        ok = ok and isinstance(parallel.verify_gathered(got.view(B, S, H, D)[:, :, perm].reshape(B, S, H * D), want, D), str)
        el, fl, pr, k5, per = parallel.reduce_step_stats(1.0 + 0.5 * rank, 2.0, 1.0, 0.25 * (rank + 1), "cpu", busy_s=0.1 * (rank + 1))
        q.put((rank, ok, (h0, hl), el, fl, pr, k5, per))
    finally:
        dist.destroy_process_group()


def test_head_shard_gather_and_reduce_world8():
    """The host logic of the 8-GPU run (world 2 above): shards of 3 heads, gather order, the exchange self-check
    and the max-over-ranks / sum-of-work reduction with eight gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [(3 * i, 3) for i in range(8)]
    for rank, ok, _, el, fl, pr, k5, per in res:
        assert ok, f"rank {rank}"
        assert el == 4.5 and fl == 16.0 and pr == 8.0 and k5 == 2.0
        assert len(per) == 8 and abs(per[7] - 0.8) < 1e-6 and abs(per[0] - 0.1) < 1e-6


def test_bench_dry_world8():
    """`python bench.py --gpus 8 --dry`: the parent starts eight gloo ranks as the driver's launcher would; one JSON line, 3 heads
    per rank, eight per-rank times, the slowest rank sets the step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["heads_per_gpu"] == 3 and len(rec["per_rank_ms"]) == 8
    assert rec["per_rank_ms"][7] > rec["per_rank_ms"][0] and rec["ms_per_step"] >= max(rec["per_rank_ms"]) * 0.99


def test_head_shard_rules():
    assert parallel.head_shard(24, 8, 3) == (9, 3)
    assert parallel.head_shard(40, 8, 7) == (35, 5)
    with pytest.raises(ValueError):
        parallel.head_shard(24, 5, 0)
    x = torch.zeros(1, 3, 2, 4)
    assert parallel.gather_heads(x).shape == (1, 3, 8)  # world of one: reshape only


def test_bench_starts_its_own_ranks_dry():
    """`python bench.py --gpus 2` with no outer launcher: the parent starts a torch.distributed.run child with two
    ranks (gloo, --dry: host logic only) and rank 0 prints one JSON line for n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1 and rec["heads_per_gpu"] == 12
    assert len(rec["per_rank_ms"]) == 2 and rec["per_rank_ms"][1] > rec["per_rank_ms"][0]  # rank 1 sleeps longer
    assert rec["imbalance"] > 1.05 and rec["ms_per_step"] >= max(rec["per_rank_ms"]) * 0.99


def test_comm_entry_points_validate_arguments_on_the_host():
    """The exchange entry points of the C-ABI reject bad arguments before touching RCCL / HIP (runs without a GPU)."""
    import ctypes
    from rectified_spaattn_amd import _lib
    L = _lib.lib()
    vp = ctypes.c_void_p
    assert L.rsa_comm_unique_id(None) == -1
    assert L.rsa_comm_create(2, 5, vp(1), ctypes.byref(vp())) == -1          # rank outside the world
    assert L.rsa_comm_destroy(None) == -1
    assert L.rsa_allgather_heads(None, 2, vp(16), vp(16), vp(16), 4, 32, None) == -1
    assert L.rsa_allgather_heads(vp(1), 2, vp(16), vp(16), vp(16), 4, 24, None) == -1   # rows not 16-byte multiples
    peers = (vp * 2)()
    states = (vp * 2)()
    assert L.rsa_allgather_heads_p2p(2, 0, vp(16), peers, states, 4, 32, None) == -1    # null peer pointers
    assert L.rsa_allgather_heads_p2p(2, 3, vp(16), peers, states, 4, 32, None) == -1    # rank outside the world
    assert L.rsa_allgather_heads_p2p(2, 0, vp(16), peers, None, 4, 32, None) == -1      # no state buffers
    assert L.rsa_allgather_heads_p2p(65, 0, vp(16), peers, states, 4, 32, None) == -1   # more ranks than flag slots
    peers[0], peers[1], states[0], states[1] = 16, 32, 4096, 8192
    assert L.rsa_allgather_heads_p2p(2, 0, vp(16), peers, states, 4, 24, None) == -1    # rows not 16-byte multiples
    assert L.rsa_p2p_state_bytes() == 1024
    assert L.rsa_p2p_state_alloc(None) == -1 and L.rsa_p2p_state_timeout(None, None) == -1 and L.rsa_p2p_state_free(None) == 0
    assert L.rsa_ipc_export(None, None) == -1 and L.rsa_ipc_open(None, 0, None) == -1 and L.rsa_ipc_close(None) == -1
    assert L.rsa_ipc_offset(None, None) == -1


class _FakeLib:
    """Stand-in for the exchange entry points of the C-ABI (host logic only): every call succeeds, except `fail_call` on
    `fail_rank`, which returns a HIP-launch style error code."""

    def __init__(self, rank, fail_rank, fail_call):
        self.rank, self.fail_rank, self.fail_call, self.calls = rank, fail_rank, fail_call, []

    def __getattr__(self, name):
        def fn(*a):
            self.calls.append(name)
            if name == self.fail_call and self.rank == self.fail_rank:
                return -5
            if name == "rsa_p2p_state_alloc":
                a[0]._obj.value = 0x1000
            if name == "rsa_ipc_open":
                a[2]._obj.value = 0x2000
            if name == "rsa_comm_create":
                a[3]._obj.value = 0x3000
            return 0
        return fn


def _setup_failure_worker(rank, world, port, q, transport, fail_call):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rectified_spaattn_amd._lib import RsaError
        fake = _FakeLib(rank, 1, fail_call)
        try:
            parallel.HeadGather(1, 8, 2, 16, torch.bfloat16, "cpu", transport=transport, lib=fake)
            q.put((rank, "constructed", fake.calls))
        except RsaError as e:
            q.put((rank, "RsaError: " + str(e), fake.calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport,fail_call", [("p2p", "rsa_ipc_export"), ("p2p", "rsa_ipc_open"), ("p2p", "rsa_p2p_state_alloc"),
                                                 ("rccl", "rsa_comm_unique_id")])
def test_head_gather_setup_failure_on_one_rank_raises_on_all(transport, fail_call):
    """A set-up step of the library's own transports that fails on ONE rank (IPC refused, allocation failed, librccl missing)
    must raise on EVERY rank before the next collective -- nobody is left waiting inside one -- and must release what the
    failed construction had already opened.  (For rccl only rank 0 creates the id, so that is where it can fail alone.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() * 7 + hash(fail_call)) % 2000)
    procs = [ctx.Process(target=_setup_failure_worker, args=(r, 2, port, q, transport, fail_call if transport == "p2p" else fail_call))
             for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if transport == "rccl":   # the id is created on rank 0 only: a rank-1 stub failure never triggers -> both construct
        assert all(r[1] == "constructed" for r in res)
        return
    for rank, what, calls in res:
        assert what.startswith("RsaError"), (rank, what)
        assert ("this rank" in what) == (rank == 1) and ("another rank" in what) == (rank == 0)
        if "rsa_ipc_open" in calls and fail_call == "rsa_ipc_open" and rank == 0:
            assert "rsa_ipc_close" in calls           # what was opened before the agreement is closed again
        if "rsa_p2p_state_alloc" in calls and not (rank == 1 and fail_call == "rsa_p2p_state_alloc"):
            assert "rsa_p2p_state_free" in calls
