"""ctypes binding of librsa_hip.so (C-ABI declared in include/rsa.h).

The HIP library is the product: there is no CPU or PyTorch fallback for device tensors.  If the shared
object is missing or cannot be loaded this module raises, and every operator that needs it fails loudly.
"""
from __future__ import annotations

import ctypes
import os

# PyTorch-ROCm bundles its own HIP runtime (soname libamdhip64.so.7).  It must be loaded BEFORE librsa_hip.so so
# that the library binds to the same runtime instance that owns torch's device context and streams; loading
# the system copy first leaves torch without a device on the GPU box.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librsa_hip.so")

RSA_BF16, RSA_FP16 = 0, 1
BLOCK = 128
HEADER_VERSION = 600   # RSA_HEADER_VERSION of include/rsa.h this ctypes mirror follows (rsa_abi_check)


class RsaError(RuntimeError):
    pass


class RsaLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "H", "D", "S", "NB_total", "NBv", "n_txt", "kv_valid", "pool_valid", "text_end_block",
        "first_frame_blocks", "q_text_valid", "kv_text_valid", "dtype")]


class RsaTensor4(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("stride_b", ctypes.c_int64), ("stride_h", ctypes.c_int64),
                ("stride_s", ctypes.c_int64)]


class RsaOut4(ctypes.Structure):
    _fields_ = RsaTensor4._fields_


BUFFER_NAMES = ("qbar", "aq", "kbar", "ak", "vbar", "scores", "unrel", "probs", "w", "R", "comp", "bitmask",
                "cols", "counts", "tpart")
TEXT_SPLIT = 32
TAIL_PIECES = 512
NUM_BUFFERS = len(BUFFER_NAMES)


class RsaBuffers(ctypes.Structure):
    # the 15 buffers, then the capacity of the last one in bytes (rsa.h 0.5.0: a non-NULL tpart must declare it)
    _fields_ = [(n, ctypes.c_void_p) for n in BUFFER_NAMES] + [("tpart_bytes", ctypes.c_size_t)]


class RsaFp8Operands(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("q8", "k8", "v8t", "scales")]


_lib = None


def lib():
    """Load librsa_hip.so once; raise RsaError if it is not there (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RsaError(f"{LIB_PATH} not found: the HIP extension is not built "
                       f"(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
    L = ctypes.CDLL(LIB_PATH)
    P = ctypes.POINTER
    vp, i32, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.rsa_version.restype = i32
    L.rsa_abi_check.argtypes = [i32, sz, sz]
    L.rsa_abi_check.restype = i32
    if L.rsa_abi_check(HEADER_VERSION, ctypes.sizeof(RsaBuffers), ctypes.sizeof(RsaLayout)) != 0:
        raise RsaError(f"{LIB_PATH} (version {L.rsa_version()}) was built from another include/rsa.h than this package's "
                       f"ctypes mirror (header {HEADER_VERSION}, rsa_buffers {ctypes.sizeof(RsaBuffers)} B): rebuild it")
    L.rsa_status_string.restype = ctypes.c_char_p
    L.rsa_status_string.argtypes = [i32]
    L.rsa_last_hip_error.restype = ctypes.c_char_p
    L.rsa_gilbert_mapping.argtypes = [i32, i32, i32, ctypes.c_char_p, vp, vp]
    L.rsa_gilbert_mapping.restype = i32
    L.rsa_gilbert_block_neighbors.argtypes = [i32, i32, i32, i32, ctypes.c_char_p, vp]
    L.rsa_gilbert_block_neighbors.restype = i32
    i64 = ctypes.c_int64
    L.rsa_permute_tokens.argtypes = [i32, i32, i32, vp, i64, i64, vp, vp, i64, i64, vp]
    L.rsa_permute_tokens.restype = i32
    L.rsa_qk_norm_rope.argtypes = [i32, i32, i32, i32, i32, RsaTensor4, vp, f32, i32, vp, vp, i32, RsaOut4, vp]
    L.rsa_qk_norm_rope.restype = i32
    L.rsa_qk_layernorm_rope.argtypes = [i32, i32, i32, i32, i32, RsaTensor4, vp, vp, f32, vp, vp, i32, RsaOut4, vp]
    L.rsa_qk_layernorm_rope.restype = i32
    L.rsa_norm_rope_heads.argtypes = [i32, i32, i32, i32, i32, vp, i64, i64, vp, f32, i32, i32, vp, vp, RsaOut4, vp]
    L.rsa_norm_rope_heads.restype = i32
    L.rsa_rel_l1.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.rsa_rel_l1.restype = i32
    L.rsa_set_tuning.argtypes = [ctypes.c_char_p, i32]
    L.rsa_set_tuning.restype = i32
    L.rsa_set_shard_invariant.argtypes = [i32]
    L.rsa_set_shard_invariant.restype = i32
    L.rsa_buffer_bytes.argtypes = [P(RsaLayout), P(sz * NUM_BUFFERS), P(sz)]
    L.rsa_carve_workspace.argtypes = [P(RsaLayout), vp, sz, P(RsaBuffers)]
    L.rsa_pool_stats.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, P(RsaBuffers), vp]
    L.rsa_pooled_scores.argtypes = [P(RsaLayout), RsaTensor4, P(RsaBuffers), vp]
    L.rsa_select_mask.argtypes = [P(RsaLayout), vp, i32, f32, P(RsaBuffers), vp]
    L.rsa_compensation.argtypes = [P(RsaLayout), P(RsaBuffers), vp]
    L.rsa_block_sparse_fwd.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, P(RsaBuffers), RsaOut4, vp]
    L.rsa_rectified_attention.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, vp, i32, f32, vp, sz,
                                          RsaOut4, vp]
    L.rsa_dense_fwd.argtypes = [i32] * 6 + [RsaTensor4, RsaTensor4, RsaTensor4, i32, i32, RsaOut4, vp]
    L.rsa_dense_causal_fwd.argtypes = L.rsa_dense_fwd.argtypes
    L.rsa_dense_causal_fwd.restype = i32
    L.rsa_dense_masked_fwd.argtypes = [i32] * 6 + [RsaTensor4, RsaTensor4, RsaTensor4, vp, i32] + [ctypes.c_int64] * 4 + [i32, RsaOut4, vp]
    L.rsa_dense_dropout_fwd.argtypes = ([i32] * 6 + [RsaTensor4, RsaTensor4, RsaTensor4, vp, i32] + [ctypes.c_int64] * 4 +
                                        [i32, i32, f32, ctypes.c_uint64, RsaOut4, vp])
    L.rsa_dense_dropout_fwd.restype = i32
    L.rsa_estimate_pr_gain.argtypes = [i32] * 5 + [vp] * 9
    L.rsa_fp8_operand_bytes.argtypes = [P(RsaLayout), P(sz * 4), P(sz)]
    L.rsa_carve_fp8_operands.argtypes = [P(RsaLayout), vp, sz, P(RsaFp8Operands)]
    L.rsa_quantize_fp8.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, P(RsaFp8Operands), vp]
    L.rsa_pool_stats_fp8.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, P(RsaBuffers), P(RsaFp8Operands),
                                     vp]
    L.rsa_dense_fp8_bytes.argtypes = [i32] * 5 + [P(sz)]
    L.rsa_dense_fwd_fp8.argtypes = [i32] * 6 + [RsaTensor4, RsaTensor4, RsaTensor4, i32, i32, vp, sz, RsaOut4, vp]
    L.rsa_dense_causal_fwd_fp8.argtypes = L.rsa_dense_fwd_fp8.argtypes
    L.rsa_dense_causal_fwd_fp8.restype = i32
    L.rsa_dense_fwd_fp8pv.argtypes = [i32] * 6 + [RsaTensor4, RsaTensor4, RsaTensor4, i32, i32, i32, vp, sz, RsaOut4, vp]
    L.rsa_dense_fwd_fp8pv.restype = i32
    L.rsa_block_sparse_fwd_fp8.argtypes = [P(RsaLayout), P(RsaFp8Operands), P(RsaBuffers), RsaOut4, vp]
    L.rsa_block_sparse_fwd_fp8pv.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, P(RsaFp8Operands), P(RsaBuffers), RsaOut4, vp]
    L.rsa_block_sparse_fwd_fp8pv.restype = i32
    L.rsa_rectified_attention_fp8.argtypes = [P(RsaLayout), RsaTensor4, RsaTensor4, RsaTensor4, vp, i32, f32, vp, sz,
                                              vp, sz, RsaOut4, vp]
    L.rsa_rectified_attention_fp8pv.argtypes = L.rsa_rectified_attention_fp8.argtypes
    L.rsa_rectified_attention_fp8pv.restype = i32
    L.rsa_comm_unique_id.argtypes = [vp]
    L.rsa_comm_create.argtypes = [i32, i32, vp, P(vp)]
    L.rsa_comm_destroy.argtypes = [vp]
    L.rsa_comm_count.argtypes = [vp, P(i32)]
    L.rsa_allgather_heads.argtypes = [vp, i32, vp, vp, vp, i64, i64, vp]
    L.rsa_allgather_heads_p2p.argtypes = [i32, i32, vp, P(vp), P(vp), i64, i64, vp]
    L.rsa_p2p_state_bytes.restype = i32
    L.rsa_p2p_state_alloc.argtypes = [P(vp)]
    L.rsa_p2p_state_free.argtypes = [vp]
    L.rsa_p2p_state_timeout.argtypes = [vp, P(i32)]
    L.rsa_ipc_offset.argtypes = [vp, P(i64)]
    L.rsa_ipc_export.argtypes = [vp, vp]
    L.rsa_ipc_open.argtypes = [vp, i32, P(vp)]
    L.rsa_ipc_close.argtypes = [vp]
    for name in ("rsa_comm_unique_id", "rsa_comm_create", "rsa_comm_destroy", "rsa_comm_count", "rsa_allgather_heads",
                 "rsa_allgather_heads_p2p", "rsa_ipc_export", "rsa_ipc_open", "rsa_ipc_close", "rsa_ipc_offset",
                 "rsa_p2p_state_alloc", "rsa_p2p_state_free", "rsa_p2p_state_timeout"):
        getattr(L, name).restype = i32
    for name in ("rsa_fp8_operand_bytes", "rsa_carve_fp8_operands", "rsa_quantize_fp8", "rsa_block_sparse_fwd_fp8",
                 "rsa_rectified_attention_fp8", "rsa_rectified_attention_fp8pv", "rsa_pool_stats_fp8", "rsa_dense_fp8_bytes",
                 "rsa_dense_fwd_fp8"):
        getattr(L, name).restype = i32
    for name in ("rsa_buffer_bytes", "rsa_carve_workspace", "rsa_pool_stats", "rsa_pooled_scores",
                 "rsa_select_mask", "rsa_compensation", "rsa_block_sparse_fwd", "rsa_rectified_attention",
                 "rsa_dense_fwd", "rsa_dense_masked_fwd", "rsa_estimate_pr_gain"):
        getattr(L, name).restype = i32
    # kernel-variant switches for A/B runs and the variant tests; rsa_set_tuning works only under RSA_TUNING=1
    for key in ("k5_tsplit", "k3_prefix", "k3_long", "k4_split", "k5_rows256", "k5_static", "k5_w64", "k5_gsync", "k5_gsync_ratio", "k5_text_last", "k5_tail_split"):
        val = os.environ.get("RSA_" + key.upper())
        if val is not None:
            check_rc = L.rsa_set_tuning(key.encode(), int(val))
            if check_rc != 0:
                raise RsaError(f"RSA_{key.upper()} is set but rsa_set_tuning refused it (export RSA_TUNING=1)")
    _lib = L
    return L


EXPORTED = ("rsa_version", "rsa_abi_check", "rsa_buffer_bytes", "rsa_carve_workspace", "rsa_pool_stats", "rsa_pooled_scores",
            "rsa_select_mask", "rsa_compensation", "rsa_block_sparse_fwd", "rsa_rectified_attention",
            "rsa_dense_fwd", "rsa_dense_causal_fwd", "rsa_dense_masked_fwd", "rsa_dense_dropout_fwd", "rsa_estimate_pr_gain", "rsa_status_string", "rsa_last_hip_error", "rsa_set_tuning", "rsa_set_shard_invariant", "rsa_gilbert_mapping",
            "rsa_gilbert_block_neighbors", "rsa_permute_tokens", "rsa_qk_norm_rope", "rsa_qk_layernorm_rope", "rsa_norm_rope_heads", "rsa_fp8_operand_bytes",
            "rsa_carve_fp8_operands", "rsa_quantize_fp8", "rsa_block_sparse_fwd_fp8", "rsa_block_sparse_fwd_fp8pv", "rsa_rectified_attention_fp8", "rsa_rectified_attention_fp8pv",
            "rsa_pool_stats_fp8", "rsa_dense_fp8_bytes", "rsa_dense_fwd_fp8", "rsa_dense_causal_fwd_fp8", "rsa_dense_fwd_fp8pv", "rsa_rel_l1",
            "rsa_comm_unique_id", "rsa_comm_create", "rsa_comm_destroy", "rsa_comm_count", "rsa_allgather_heads",
            "rsa_allgather_heads_p2p", "rsa_p2p_state_bytes", "rsa_p2p_state_alloc", "rsa_p2p_state_free", "rsa_p2p_state_timeout", "rsa_ipc_export", "rsa_ipc_open", "rsa_ipc_close",
            "rsa_ipc_offset")


def check(status: int, what: str):
    """Status code -> the exception types the reference raises: bad head_dim / dtype are
    AssertionError there (hunyuan :119-121); everything else is a RuntimeError subclass."""
    if status == 0:
        return
    msg = f"{what}: {lib().rsa_status_string(status).decode()} (rsa_status {status})"
    if status == -4:
        msg += f": {lib().rsa_last_hip_error().decode()}"
    if status == -2:
        raise AssertionError(msg)
    raise RsaError(msg)
