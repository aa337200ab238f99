"""Worker of tests/test_gpu_shard_invariance.py::test_p2p_gather_{two,eight}_processes_one_device: N processes on cuda:0,
gloo control plane; each contributes its head slab through rsa_allgather_heads_p2p (HIP IPC peer copies) and checks
the gathered [B, S, H*D] rows against the layout the reference's unsharded output has (hunyuan :383-387).  World 8 is the
software pre-flight of the first real 8-GPU run: 8 IPC peers, 8 flags per state block, the two result buffers alternating over
six gathers, and the whole exchange object built, used and destroyed TWICE in one process."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectified_spaattn_amd import parallel  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, S, Hl, D = 1, 1000, 3, 128
    g = torch.Generator(device="cpu").manual_seed(7)
    full_ref = torch.randn(B, S, world * Hl, D, generator=g).to(torch.bfloat16)
    mine = full_ref[:, :, rank * Hl:(rank + 1) * Hl].contiguous().to(dev)
    # construct / use / destroy once before the run proper: a second HeadGather in the same process must find nothing left over
    # (IPC handles closed, state block freed, flags of the new block zero)
    first = parallel.HeadGather(B, S, Hl, D, torch.bfloat16, dev, transport="p2p")
    warm = first.gather(mine - 3).clone()
    torch.cuda.synchronize()
    first.check()
    assert torch.equal(warm, (full_ref.to(dev) - 3).reshape(B, S, world * Hl * D)), f"rank {rank}: first exchange object"
    first.close()
    dist.barrier()
    hg = parallel.HeadGather(B, S, Hl, D, torch.bfloat16, dev, transport="p2p")
    # six back-to-back gathers with NO host synchronisation between them: the exchange is stream-ordered (device flags),
    # the results alternate between the two full buffers, each is consumed (copied) on the issuing stream
    sync_calls = []
    real_sync, real_barrier = torch.cuda.Stream.synchronize, dist.barrier
    torch.cuda.Stream.synchronize = lambda self_: (sync_calls.append("stream"), real_sync(self_))[1]
    dist.barrier = lambda *a, **k: (sync_calls.append("barrier"), real_barrier(*a, **k))[1]
    outs = []
    for step in range(6):
        outs.append(hg.gather(mine + step).clone())
        if rank == world - 1 and step == 2:
            torch.cuda._sleep(200_000_000)   # uneven load: this rank falls ~0.1 s behind; the peer's wait kernel covers it
    torch.cuda.Stream.synchronize, dist.barrier = real_sync, real_barrier
    assert not sync_calls, f"gather() synchronised with the host: {sync_calls}"
    torch.cuda.synchronize()
    hg.check()
    for step, out in enumerate(outs):
        want = (full_ref.to(dev) + step).reshape(B, S, world * Hl * D)
        assert torch.equal(out, want), f"rank {rank} step {step}: gathered rows differ"
    # the self-check of bench.py --gpus N on the same exchange: fingerprints of the local heads over the control plane (gloo),
    # recomputed from the gathered buffer on every rank; a buffer missing a peer's slab must fail it
    sums = parallel.exchange_checksums(parallel.head_checksums(mine + 5, D), torch.device("cpu"))
    assert sums.shape == (world * Hl, 2 + parallel.CHECK_ROWS * D)
    assert parallel.verify_gathered(outs[5], sums, D) is None
    msg = parallel.verify_gathered(outs[4], sums, D)          # the previous gather's buffer: every slab is stale by 1.0
    assert isinstance(msg, str) and "head 0" in msg, msg
    broken = outs[5].clone()
    peer = (rank + 1) % world
    broken.view(B, S, world * Hl, D)[:, :, peer * Hl:(peer + 1) * Hl] = 0      # a peer's slab never arrived
    msg = parallel.verify_gathered(broken, sums, D)
    assert isinstance(msg, str) and f"head {peer * Hl}" in msg, msg
    hg.close()
    dist.barrier()
    if rank == 0:
        print("IPC_GATHER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
