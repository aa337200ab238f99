"""Worker of tests/test_gpu_shard_invariance.py::test_p2p_gather_two_processes_one_device: two processes on cuda:0,
gloo control plane; each contributes its head slab through rsa_allgather_heads_p2p (HIP IPC peer copies) and checks
the gathered [B, S, H*D] rows against the layout the reference's unsharded output has (hunyuan :383-387)."""
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
    hg = parallel.HeadGather(B, S, Hl, D, torch.bfloat16, dev, transport="p2p")
    for step in range(3):   # repeated gathers reuse the peers' buffers
        out = hg.gather(mine + step)
        torch.cuda.synchronize()
        want = (full_ref.to(dev) + step).reshape(B, S, world * Hl * D)
        assert torch.equal(out, want), f"rank {rank} step {step}: gathered rows differ"
    hg.close()
    dist.barrier()
    if rank == 0:
        print("IPC_GATHER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
