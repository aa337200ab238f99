"""rectified_spaattn_amd -- MI355X-native Rectified SpaAttn attention path.

Module names mirror the reference package `rectified_spaattn`:
    attn                      fullattn, get_cu_seqlens, get_attn_mask, get_flash_attn_params, MEMORY_LAYOUT
    gapr_mask                 estimate_pr_gain
    rectified_hunyuan_attn    rectified_block_sparse_attention, RectifiedHunyuanVideoSpaAttnProcessor2_0
    rectified_flux_attn       rectified_block_sparse_attention, RectifiedFluxSpaAttnProcessor2_0
    rectified_wan21_attn      rectified_block_sparse_attention, RectifiedWan{T2V,I2V}SpaAttnProcessor2_0
    rectified_wan22_attn      RectifiedWan{TI2V,T2V,I2V}SpaAttnProcessor2_0
    rectified_cogvideo_attn   rectified_block_sparse_attention, RectifiedCogVideoXVideoSpaAttnProcessor2_0
    attn_processor            get_attn_processors, set_attn_processor
    teacache                  TeaCache step-skipping controller (scripts' teacache_forward bookkeeping), rel_l1_distance
Device work goes through librsa_hip.so (C-ABI in include/rsa.h): the attention operators and fullattn never fall back to
PyTorch kernels for device tensors (they raise).  Two deliberate uses of plain PyTorch expressions remain and are the
reference's own code paths: fullattn(mode="torch" | "vanilla") on CPU tensors (BASELINE config 1) and
teacache.rel_l1_distance for CPU tensors / small fp32 inputs such as timestep embeddings (bf16 / fp16 device tensors take
the one-pass HIP reduction rsa_rel_l1).
"""
__version__ = "0.5.0"


def set_qkv_fp8(enabled):
    """Run the block-sparse kernel of every sparse operator / processor call on e4m3 operands (fp8 MFMA): True = e4m3 images of
    Q, K, V (head dim 64 / 128; relative L1 distance 0.12 of the layer output from the 2-byte path), "pv" = Q . K^T on the 2-byte
    inputs and only P . V on e4m3 (head dims 64 and 128; relative L1 0.04, within SURVEY 8(d)'s 8e-2 of the bf16 oracle); returns the
    previous setting.  Default off = the reference's input-dtype behaviour."""
    from . import _operator
    return _operator.set_qkv_fp8(enabled)


def set_dense_fp8(enabled) -> bool:
    """Run fullattn's device path (dense attention, head dims 64 / 128) on e4m3 images of Q, K, V (True), or -- "pv" -- take the
    scores from the 2-byte q and k and use e4m3 only for P and V (relative L1 0.04 instead of 0.12); returns the previous setting.
    Default off."""
    from . import _operator
    return _operator.set_dense_fp8(enabled)


def clear_buffer_cache() -> None:
    """Drops the intermediate buffers the operators keep per (geometry, device, stream) between calls (about 0.45 GB per
    set at the HunyuanVideo shape; `_core.BUFFER_CACHE_MAX_BYTES` bounds the total, `_core.BUFFER_CACHE = False` turns the
    cache off).  Calls inside a HIP-graph capture never use the cache."""
    from . import _core
    _core.clear_buffer_cache()
