"""Head sharding across the GPUs of one node (one process per GPU, torch.distributed: "nccl" == RCCL over xGMI
on ROCm, "gloo" in the CPU tests).

The algorithm has no cross-head dependency (softmax / sort / cumsum are per (b, h, q-block) row; top_k, p and
the neighbour matrix are head-independent), so mask selection and the sparse pass need no collective.  The
only exchange is optional and sits at the layer boundary: an all-gather of O along the head axis, needed iff
the consumer (the to_out GEMM) is not head-sharded itself."""
import ctypes
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def head_shard(num_heads: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous head blocks: returns (first_head, heads_on_this_rank).  Requires world_size | num_heads."""
    if num_heads % world_size:
        raise ValueError(f"{num_heads} heads do not split evenly over {world_size} ranks")
    per = num_heads // world_size
    return rank * per, per


def gather_heads(out_local: torch.Tensor, group=None) -> torch.Tensor:
    """out_local [B, S, H_local, D] (or [B, S, H_local*D]) on every rank -> [B, S, H*D] on every rank, heads
    in rank order (== the unsharded layout of the reference's output, hunyuan :383-387)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    B, S = out_local.shape[:2]
    flat = out_local.reshape(B, S, -1).contiguous()
    if world == 1:
        return flat
    parts: List[torch.Tensor] = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat, group=group)
    return torch.cat(parts, dim=-1)


def reduce_step_stats(elapsed_s: float, flops: float, pairs: float, k5_ms: float, device, group=None,
                      busy_s: float = None):
    """(max elapsed, sum flops, sum kept pairs, max K5 ms, per-rank list) over all ranks.  The per-rank list holds
    each rank's own busy time `busy_s` (measured before the closing barrier; default: elapsed_s)."""
    busy_s = elapsed_s if busy_s is None else busy_s
    t = torch.tensor([busy_s, flops, pairs, k5_ms, elapsed_s], dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return elapsed_s, flops, pairs, k5_ms, [busy_s]
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM, group=group)
    per_rank = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(per_rank, t, group=group)
    return tmax[4].item(), tsum[1].item(), tsum[2].item(), tmax[3].item(), [p[0].item() for p in per_rank]


class HeadGather:
    """The optional exchange step through the library's own transports (include/rsa.h, "multi-GPU"):

        transport="rccl"  rsa_allgather_heads      ncclAllGather into a rank-major staging buffer + one unpack kernel
        transport="p2p"   rsa_allgather_heads_p2p  world 2-D peer copies straight into every rank's full buffer (xGMI is
                                                   point to point: one link per peer, no ring, no staging)

    Both need one process per GPU with torch.distributed initialised (any backend): it carries the 128-byte RCCL id /
    the 64-byte IPC handles between the ranks once, at construction.  gather(out_local) -> [B, S, H*D] on every rank.
    world_size 1 (or no process group) degenerates to a local copy through the same entry points."""

    def __init__(self, B: int, S: int, H_local: int, D: int, dtype: torch.dtype, device, transport: str = "rccl",
                 group=None):
        from . import _lib
        if transport not in ("rccl", "p2p"):
            raise ValueError(f"unknown transport {transport!r}")
        self.L = _lib.lib()
        self._check = _lib.check
        self.transport, self.group = transport, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.rows, self.row_bytes = B * S, H_local * D * torch.empty((), dtype=dtype).element_size()
        if self.row_bytes % 16:
            raise ValueError("H_local * D * itemsize must be a multiple of 16 bytes")
        self.full = torch.empty((B, S, self.world * H_local * D), dtype=dtype, device=self.device)
        self.comm = ctypes.c_void_p()
        self.staging: Optional[torch.Tensor] = None
        self.peers = None
        self._opened: List[ctypes.c_void_p] = []
        vp = ctypes.c_void_p
        with torch.cuda.device(self.device):
            if transport == "rccl":
                idbuf = (ctypes.c_ubyte * 128)()
                if self.rank == 0:
                    self._check(self.L.rsa_comm_unique_id(idbuf), "rsa_comm_unique_id")
                ids = [bytes(idbuf)]
                if self.world > 1:
                    dist.broadcast_object_list(ids, src=0, group=group)
                idbuf = (ctypes.c_ubyte * 128).from_buffer_copy(ids[0])
                self._check(self.L.rsa_comm_create(self.world, self.rank, idbuf, ctypes.byref(self.comm)),
                            "rsa_comm_create")
                self.staging = torch.empty((self.world, B * S, H_local * D), dtype=dtype, device=self.device)
            else:
                h = (ctypes.c_ubyte * 64)()
                self._check(self.L.rsa_ipc_export(vp(self.full.data_ptr()), h), "rsa_ipc_export")
                mine = (bytes(h), self.device.index if self.device.index is not None else torch.cuda.current_device())
                allh = [None] * self.world
                if self.world > 1:
                    dist.all_gather_object(allh, mine, group=group)
                else:
                    allh = [mine]
                self.peers = (vp * self.world)()
                for r, (hb, dev_idx) in enumerate(allh):
                    if r == self.rank:
                        self.peers[r] = self.full.data_ptr()
                    else:
                        ptr = vp()
                        self._check(self.L.rsa_ipc_open((ctypes.c_ubyte * 64).from_buffer_copy(hb), int(dev_idx),
                                                        ctypes.byref(ptr)), "rsa_ipc_open")
                        self.peers[r] = ptr.value
                        self._opened.append(ptr)

    def gather(self, out_local: torch.Tensor) -> torch.Tensor:
        flat = out_local.reshape(self.rows, -1)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        assert flat.shape[1] * flat.element_size() == self.row_bytes and flat.device == self.device
        vp = ctypes.c_void_p
        st = vp(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            if self.transport == "rccl":
                self._check(self.L.rsa_allgather_heads(self.comm, self.world, vp(flat.data_ptr()),
                                                       vp(self.staging.data_ptr()), vp(self.full.data_ptr()), self.rows,
                                                       self.row_bytes, st), "rsa_allgather_heads")
            else:
                if self.world > 1:   # every peer must be done READING its full buffer of the previous step
                    torch.cuda.current_stream(self.device).synchronize()
                    dist.barrier(group=self.group)
                self._check(self.L.rsa_allgather_heads_p2p(self.world, self.rank, vp(flat.data_ptr()), self.peers,
                                                           self.rows, self.row_bytes, st), "rsa_allgather_heads_p2p")
                if self.world > 1:   # ... and every peer's slab must have landed here before this rank reads
                    torch.cuda.current_stream(self.device).synchronize()
                    dist.barrier(group=self.group)
        return self.full

    def close(self):
        for ptr in self._opened:
            self.L.rsa_ipc_close(ptr)
        self._opened = []
        if self.comm:
            self.L.rsa_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()
