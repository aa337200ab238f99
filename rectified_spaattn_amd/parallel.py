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


def set_shard_invariant(on: bool = True) -> bool:
    """Process-global (rsa_set_shard_invariant, include/rsa.h).  K5 plans the split of the dense text rows (32 pieces on short
    grids, else 16) and the split of the last, partial generation's walks from the SIZE OF THE LAUNCH, so a rank that holds 3
    heads and the unsharded 24-head call sum the rows those plans touch in different orders: equal within rounding, not byte
    for byte.  With the switch on both are planned per head and every rank produces exactly the unsharded call's bytes for its
    heads (3-5 % of K5 on short grids).  Returns the previous setting."""
    from . import _lib
    return bool(_lib.lib().rsa_set_shard_invariant(1 if on else 0))


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


CHECK_ROWS = 8   # sampled rows per head in head_checksums


def head_checksums(o: torch.Tensor, D: int) -> torch.Tensor:
    """Per-head fingerprints of an attention output, fp64 [H, 2 + CHECK_ROWS * D]: sum, sum of squares and CHECK_ROWS whole
    rows (positions spread over the sequence, the same for every head).  o: [B, S, H, D] or [B, S, H*D].  What a rank computes
    from its LOCAL heads must reappear in every rank's gathered buffer: sums within 1e-9 relative (the reduction order of the
    two layouts may differ), the sampled rows exactly."""
    B, S = o.shape[:2]
    x = o.reshape(B, S, -1, D)
    H = x.shape[2]
    xf = x.to(torch.float64)
    rows = torch.tensor([(S * (2 * i + 1)) // (2 * CHECK_ROWS) for i in range(CHECK_ROWS)], device=o.device)
    samp = xf[B - 1].index_select(0, rows).permute(1, 0, 2).reshape(H, CHECK_ROWS * D)
    return torch.cat([xf.sum(dim=(0, 1, 3))[:, None], (xf * xf).sum(dim=(0, 1, 3))[:, None], samp], dim=1)


def exchange_checksums(local: torch.Tensor, stat_device, group=None) -> torch.Tensor:
    """[H_local, C] on every rank -> [H, C] on every rank, heads in rank order (a small all_gather on the control plane)."""
    t = local.to(stat_device)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    return torch.cat(parts, dim=0)


def verify_gathered(full: torch.Tensor, want: torch.Tensor, D: int) -> Optional[str]:
    """Checks a gathered [B, S, H*D] buffer against the fingerprints every rank published for its heads.  Returns None if every
    head matches, else a short description of the first mismatch (which head -- i.e. which source rank's slab -- and how)."""
    got = head_checksums(full, D).to(want.device)
    if got.shape != want.shape:
        return f"gathered buffer has {got.shape[0]} heads, expected {want.shape[0]}"
    scale = want[:, :2].abs().clamp_min(1e-30)
    bad_sum = ((got[:, :2] - want[:, :2]).abs() / scale > 1e-9).any(dim=1)
    bad_row = (got[:, 2:] != want[:, 2:]).any(dim=1)
    bad = bad_sum | bad_row | ~torch.isfinite(got).all(dim=1)
    if not bool(bad.any()):
        return None
    h = int(torch.nonzero(bad)[0])
    return (f"head {h} of the gathered buffer differs from what its rank computed (sum {got[h, 0].item():.6e} vs "
            f"{want[h, 0].item():.6e}, sampled rows {'differ' if bool(bad_row[h]) else 'equal'}); "
            f"{int(bad.sum())} of {bad.numel()} heads wrong")


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
        transport="p2p"   rsa_allgather_heads_p2p  one copy kernel writing this rank's slab into every peer's full buffer
                                                   over all xGMI links at once + device flags + a wait kernel: stream-
                                                   ordered, NO host synchronisation or barrier inside gather()

    Both need one process per GPU with torch.distributed initialised (any backend): it carries the 128-byte RCCL id /
    the 64-byte IPC handles between the ranks once, at construction.  gather(out_local) -> [B, S, H*D] on every rank.
    world_size 1 (or no process group) degenerates to a local copy through the same entry points.

    Construction is collective.  Every step that can fail on one rank alone (library missing, IPC refused) is local and
    is followed by an agreement (all ranks learn whether anyone failed) BEFORE the next collective, so a failure raises on
    every rank instead of leaving the others inside a collective.

    p2p: the result alternates between two buffers, and a peer may start overwriting the buffer of gather n as soon as THIS
    rank's copy kernel of gather n + 1 has run (that lets the peer pass its wait n + 1 and issue copy n + 2 into it): consume a
    gather's result on the issuing stream BEFORE the next gather() is issued (include/rsa.h says the same).  `check()` reads
    the time-out word (synchronises); a timed-out wait leaves stale slabs in the result, so call it before trusting a run."""

    def __init__(self, B: int, S: int, H_local: int, D: int, dtype: torch.dtype, device, transport: str = "rccl",
                 group=None, lib=None):
        from . import _lib
        if transport not in ("rccl", "p2p"):
            raise ValueError(f"unknown transport {transport!r}")
        self.L = lib if lib is not None else _lib.lib()   # (lib: a stand-in with the same entry points, for the host-logic tests)
        self._check = _lib.check
        self.transport, self.group = transport, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        self.rows, self.row_bytes = B * S, H_local * D * torch.empty((), dtype=dtype).element_size()
        if self.row_bytes % 16:
            raise ValueError("H_local * D * itemsize must be a multiple of 16 bytes")
        self.comm = ctypes.c_void_p()
        self.staging: Optional[torch.Tensor] = None
        self._opened: List[ctypes.c_void_p] = []
        self._calls = 0
        vp = ctypes.c_void_p
        try:
            with self._devctx():
                if transport == "rccl":
                    self.full = torch.empty((B, S, self.world * H_local * D), dtype=dtype, device=self.device)
                    idbuf = (ctypes.c_ubyte * 128)()
                    err = self._local(lambda: self._check(self.L.rsa_comm_unique_id(idbuf), "rsa_comm_unique_id")
                                      if self.rank == 0 else None)
                    self._agree(err, "rsa_comm_unique_id")
                    ids = [bytes(idbuf)]
                    if self.world > 1:
                        dist.broadcast_object_list(ids, src=0, group=group)
                    idbuf = (ctypes.c_ubyte * 128).from_buffer_copy(ids[0])
                    # (ncclCommInitRank is itself collective: every rank enters it, or none does -- agreed above)
                    self._check(self.L.rsa_comm_create(self.world, self.rank, idbuf, ctypes.byref(self.comm)),
                                "rsa_comm_create")
                    self.staging = torch.empty((self.world, B * S, H_local * D), dtype=dtype, device=self.device)
                else:
                    # two allocations, two IPC handles per rank: the exchange state (FINE-GRAINED memory from the library: peers
                    # write its flags over xGMI while this GPU polls them -- cross-agent visibility of that is only defined for
                    # fine-grained allocations) and one ordinary block [full buffer 0 | full buffer 1]
                    nfull = self.rows * self.world * self.row_bytes
                    nfull_pad = (nfull + 255) // 256 * 256
                    self._state = vp()
                    self._blob = torch.empty(2 * nfull_pad, dtype=torch.uint8, device=self.device)
                    self._fulls = [self._blob[i * nfull_pad: i * nfull_pad + nfull].view(dtype)
                                   .view(B, S, self.world * H_local * D) for i in range(2)]
                    self.full = self._fulls[0]
                    hs, hd = (ctypes.c_ubyte * 64)(), (ctypes.c_ubyte * 64)()
                    off = ctypes.c_int64()

                    def export():
                        self._check(self.L.rsa_p2p_state_alloc(ctypes.byref(self._state)), "rsa_p2p_state_alloc")   # zeroed, synchronised
                        self._check(self.L.rsa_ipc_export(self._state, hs), "rsa_ipc_export(state)")
                        self._check(self.L.rsa_ipc_export(vp(self._blob.data_ptr()), hd), "rsa_ipc_export")
                        self._check(self.L.rsa_ipc_offset(vp(self._blob.data_ptr()), ctypes.byref(off)), "rsa_ipc_offset")
                    self._agree(self._local(export), "rsa_ipc_export")
                    dev_idx = self.device.index if self.device.index is not None else (
                        torch.cuda.current_device() if self.device.type == "cuda" else 0)
                    mine = (bytes(hs), bytes(hd), int(off.value), dev_idx)
                    allh = [None] * self.world
                    if self.world > 1:
                        dist.all_gather_object(allh, mine, group=group)
                    else:
                        allh = [mine]
                    sbases, dbases = [0] * self.world, [0] * self.world

                    def open_peers():
                        for r, (hsb, hdb, off_r, dev_idx) in enumerate(allh):
                            if r == self.rank:
                                sbases[r], dbases[r] = self._state.value, self._blob.data_ptr()
                                continue
                            for hb, dst, add in ((hsb, sbases, 0), (hdb, dbases, off_r)):
                                ptr = vp()
                                self._check(self.L.rsa_ipc_open((ctypes.c_ubyte * 64).from_buffer_copy(hb), int(dev_idx),
                                                                ctypes.byref(ptr)), "rsa_ipc_open")
                                self._opened.append(ptr)
                                dst[r] = ptr.value + add
                    self._agree(self._local(open_peers), "rsa_ipc_open")
                    self._states = (vp * self.world)(*sbases)
                    self._peer_fulls = [(vp * self.world)(*[b + i * nfull_pad for b in dbases]) for i in range(2)]
        except BaseException:
            self.close()   # handles already opened / the communicator do not outlive a failed construction
            raise

    def _devctx(self):
        import contextlib
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    @staticmethod
    def _local(fn):
        """Runs a step that may fail on this rank alone; returns the error text or None."""
        try:
            fn()
            return None
        except Exception as e:  # noqa: BLE001
            return repr(e)[:300]

    def _agree(self, err, what):
        """All ranks learn whether the local step failed anywhere; raises on EVERY rank if it did."""
        bad = 1 if err else 0
        if self.world > 1:
            t = torch.tensor([bad], dtype=torch.int32)
            if dist.get_backend(self.group) == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            bad = int(t.item())
        if bad:
            from ._lib import RsaError
            raise RsaError(f"HeadGather({self.transport}): {what} failed on " + (f"this rank: {err}" if err else "another rank"))

    def gather(self, out_local: torch.Tensor) -> torch.Tensor:
        flat = out_local.reshape(self.rows, -1)
        if not flat.is_contiguous():
            flat = flat.contiguous()
        assert flat.shape[1] * flat.element_size() == self.row_bytes and flat.device == self.device
        vp = ctypes.c_void_p
        st = vp(torch.cuda.current_stream(self.device).cuda_stream)
        with self._devctx():
            if self.transport == "rccl":
                self._check(self.L.rsa_allgather_heads(self.comm, self.world, vp(flat.data_ptr()),
                                                       vp(self.staging.data_ptr()), vp(self.full.data_ptr()), self.rows,
                                                       self.row_bytes, st), "rsa_allgather_heads")
            else:
                par = self._calls & 1
                self._calls += 1
                self._check(self.L.rsa_allgather_heads_p2p(self.world, self.rank, vp(flat.data_ptr()),
                                                           self._peer_fulls[par], self._states, self.rows, self.row_bytes,
                                                           st), "rsa_allgather_heads_p2p")
                self.full = self._fulls[par]
        return self.full

    def comm_ranks(self) -> Optional[int]:
        """rccl transport: the rank count the RCCL communicator itself reports (ncclCommCount); None for p2p."""
        if self.transport != "rccl" or not self.comm:
            return None
        n = ctypes.c_int32(0)
        self._check(self.L.rsa_comm_count(self.comm, ctypes.byref(n)), "rsa_comm_count")
        return int(n.value)

    def check(self) -> None:
        """p2p: raises if a wait inside an exchange gave up (a peer did not deliver within 4 s).  Synchronises."""
        if self.transport != "p2p":
            return
        torch.cuda.current_stream(self.device).synchronize()
        w = ctypes.c_int32(0)
        with self._devctx():
            self._check(self.L.rsa_p2p_state_timeout(self._state, ctypes.byref(w)), "rsa_p2p_state_timeout")
        word = int(w.value)
        if word:
            from ._lib import RsaError
            raise RsaError(f"HeadGather(p2p): rank {self.rank} timed out waiting for rank {word - 1}")

    def close(self):
        for ptr in getattr(self, "_opened", []):
            self.L.rsa_ipc_close(ptr)
        self._opened = []
        if getattr(self, "_state", None):
            self.L.rsa_p2p_state_free(self._state)
            self._state = ctypes.c_void_p()
        if getattr(self, "comm", None):
            self.L.rsa_comm_destroy(self.comm)
            self.comm = ctypes.c_void_p()
