/*
 * rsa.h -- C-ABI of librsa_hip.so: the MI355X (gfx950) rectified block-sparse attention path.
 *
 * This is the drop-in boundary for the reference's hot path (BienLuky/Rectified-SpaAttn).  The reference has
 * no FFI of its own (it is pure Python + one Triton kernel); each entry point below names the reference
 * function (file:line under the reference root) whose work it replaces, and INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers unless marked host
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates, never
 *     synchronises, never throws; returns RSA_OK or a negative rsa_status
 *   - q/k/v are [B, H, S, D] views described by element strides (stride_b, stride_h, stride_s); the head
 *     dimension is contiguous.  dtype: RSA_BF16 or RSA_FP16 (2-byte elements)
 *   - "bh" below means the flattened index b*H + h
 *   - statistics follow the fp32 numeric contract written in oracle/rsa_oracle.c (C1..C8)
 */
#ifndef RSA_H_
#define RSA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSA_BLOCK 128 /* tokens per block (block_size_M = block_size_N = 128 in every reference script) */

typedef enum rsa_status {
    RSA_OK = 0,
    RSA_ERR_BAD_ARG = -1,       /* null pointer, negative size, inconsistent layout */
    RSA_ERR_UNSUPPORTED = -2,   /* head_dim not in {64,128}, dtype unknown, row too long for the select kernel */
    RSA_ERR_WORKSPACE = -3,     /* workspace too small */
    RSA_ERR_LAUNCH = -4         /* hipGetLastError() after a launch was not hipSuccess */
} rsa_status;

typedef enum rsa_dtype { RSA_BF16 = 0, RSA_FP16 = 1 } rsa_dtype;

/* Geometry of one self-attention call.  The four reference variants differ only in these numbers
 * (hunyuan :313-332, flux :307-320, cogvideo :306-322, wan21 :297-313). */
typedef struct rsa_layout {
    int32_t B, H, D;
    int32_t S;               /* true sequence length (q and kv) */
    int32_t NB_total;        /* ceil(S / 128): the reference zero-pads up to this */
    int32_t NBv;             /* visual blocks == sparse query blocks */
    int32_t n_txt;           /* valid text tokens scored individually ("attenable"); 0 = no text tail, no IPAR */
    int32_t kv_valid;        /* kv columns >= kv_valid are masked in the sparse pass ("seqlens") */
    int32_t pool_valid;      /* K/V rows >= pool_valid count as zero when pooling (hunyuan zeroes them :307-308) */
    int32_t text_end_block;  /* text blocks [NBv, text_end_block) are kept by every query block */
    int32_t first_frame_blocks; /* wan: rows < ffb keep cols < ffb (wan21 :270-271) */
    int32_t q_text_valid;    /* text query rows [NBv*128, NBv*128+q_text_valid): dense attention */
    int32_t kv_text_valid;   /*   ... over kv [0, kv_text_valid); later rows up to S are written as 0 */
    int32_t dtype;           /* rsa_dtype */
} rsa_layout;

typedef struct rsa_tensor4 { /* a [B, H, S, D] view */
    const void* ptr;
    int64_t stride_b, stride_h, stride_s; /* in elements */
} rsa_tensor4;

typedef struct rsa_out4 {
    void* ptr;
    int64_t stride_b, stride_h, stride_s;
} rsa_out4;

/* Sizes (in elements) of the intermediate buffers for a layout; all are per call.  Fill the struct with
 * rsa_carve_workspace, or zero it first (memset) and set the members by hand: a member this header adds in a later
 * version is then NULL = "feature not used" (check rsa_version() >= 300 for this layout: 15 members; >= 310 for the
 * block-scaled rsa_fp8_operands). */
typedef struct rsa_buffers {
    float* qbar;      /* [BH, NBv, D]       block means of visual Q            */
    float* aq;        /* [BH, NBv, D]       mean |Q - qbar|                    */
    float* kbar;      /* [BH, NBv, D]                                          */
    float* ak;        /* [BH, NBv, D]                                          */
    float* vbar;      /* [BH, NB_total, D]                                     */
    float* scores;    /* [BH, NBv, NBv + n_txt]  unscaled pooled scores        */
    uint8_t* unrel;   /* [BH, NBv, NBv]     GAPR: 1 = pooled compensation NOT trustworthy */
    float* probs;     /* [BH, NBv, L]       L = NBv + (n_txt > 0): implicit full attention row */
    float* w;         /* [BH, NBv, L]       compensation weights (probs where dropped & reliable, else 0) */
    float* R;         /* [BH, NBv]          rectification factor               */
    float* comp;      /* [BH, NBv, D]       sum_j w_ij * vbar_j                 */
    uint32_t* bitmask;/* [BH, NBv, ceil(NB_total/32)]  kept blocks, bit j%32 of word j/32 */
    int32_t* cols;    /* [BH, NBv, NB_total] kept block indices ascending (first counts[] entries valid) */
    int32_t* counts;  /* [BH, NBv]                                              */
    /* split-KV partials (may be NULL: then nothing is split).  First region: the dense TEXT query blocks -- K5 splits each
     * text block's key range over up to RSA_TEXT_SPLIT workgroups and a small combine kernel merges them (NULL: one workgroup
     * walks all keys of a text block).  Second region, RSA_TAIL_PIECES more [128, D + 2] blocks behind it (since 0.4.0): the
     * walks of the LAST, partial generation of sparse query blocks, split over the workgroup slots that generation would
     * leave idle (head dim 128; merged, rectified and stored by a second combine kernel): */
    float* tpart;     /* [BH * (NB_total - NBv) * RSA_TEXT_SPLIT + RSA_TAIL_PIECES, 128, D + 2]  unnormalised O, then (m, l) per query row */
    /* (since 0.5.0; not a buffer: RSA_NUM_BUFFERS stays 15) bytes the caller allocated behind tpart.  rsa_carve_workspace
     * fills it; a host that sets tpart by hand must set it too: K5 clamps the text split and the tail split to what fits,
     * and a non-NULL tpart with tpart_bytes == 0 is refused (RSA_ERR_WORKSPACE) instead of trusted -- the formula above grew
     * in 0.4.0, and a buffer sized by an older header would otherwise be overrun. */
    size_t tpart_bytes;
} rsa_buffers;
#define RSA_NUM_BUFFERS 15
#define RSA_TEXT_SPLIT 32
#define RSA_TAIL_PIECES 512

/* Library identification: returns 10000*major + 100*minor + patch. */
int rsa_version(void);

/* ABI guard (since 0.6.0).  rsa_buffers GREW in 0.5.0 (tpart_bytes behind the 15 pointers) and carries no size member of
 * its own: a host compiled against an older header hands the library a shorter struct, and what lies behind it is read as a
 * capacity.  Call this ONCE after loading the library, with the macros of the header the host was compiled against:
 *     if (rsa_abi_check(RSA_HEADER_VERSION, sizeof(rsa_buffers), sizeof(rsa_layout)) != RSA_OK) refuse to run;
 * RSA_OK iff the library was built from a header with the same struct sizes and the same major.minor; RSA_ERR_UNSUPPORTED
 * otherwise.  (The Python host calls it when it loads the library; examples/c_host does too.) */
#define RSA_HEADER_VERSION 600
int rsa_abi_check(int header_version, size_t sizeof_rsa_buffers, size_t sizeof_rsa_layout);

/* Process-global switch (default 0).  K5 plans two things from the SIZE OF THE LAUNCH: how many pieces the dense text rows
 * are split into (32 on grids of fewer than 8 generations, else 16) and whether the walks of the last, partial generation are
 * split over its idle slots.  Both change the summation order of the rows they touch, so a head-sharded run (3 heads per
 * rank) and the unsharded one (24 heads) agree on those rows within rounding, not byte for byte.  on != 0 plans both per
 * head (16 text pieces, no tail split): every rank then produces exactly the bytes the unsharded call produces for its
 * heads, at 3-5 % of K5 on short grids.  Returns the previous value. */
int rsa_set_shard_invariant(int on);

/* Bytes needed for each rsa_buffers member, written in member order into sizes[RSA_NUM_BUFFERS], and their sum
 * (each rounded up to 256 B) into *total.  Lets a caller carve one workspace.  Replaces the ~25 temporaries
 * the reference allocates per call (hunyuan :189-262, :348-357). */
int rsa_buffer_bytes(const rsa_layout* lay, size_t sizes[RSA_NUM_BUFFERS], size_t* total);

/* Carve `ws` (>= total bytes from rsa_buffer_bytes, 256-B aligned) into an rsa_buffers. */
int rsa_carve_workspace(const rsa_layout* lay, void* ws, size_t ws_bytes, rsa_buffers* out);

/* K1 -- one HBM pass over Q (visual rows), K, V: block means and mean-absolute-deviations.
 * Replaces: Q/K pooling hunyuan :189-194 (wan21 :189-192), V pooling :356, and the |Q - qbar| / |K - kbar|
 * statistics of estimate_pr_gain, gapr_mask.py:19-23,:30. */
int rsa_pool_stats(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v, const rsa_buffers* buf,
                   void* stream);

/* K2 -- pooled scores qbar.kbar^T (+ one column per valid text token), and the GAPR bit.
 * Replaces: bmm hunyuan :198-205 and estimate_pr_gain gapr_mask.py:26-42. */
int rsa_pooled_scores(const rsa_layout* lay, rsa_tensor4 k, const rsa_buffers* buf, void* stream);

/* K3 -- per (b,h,q-block) row: softmax, IPAR, stable descending sort, cumulative threshold, top-k,
 * neighbour / text / first-frame union, bitmask + ascending column list, R and compensation weights.
 * neighbor: device uint8 [NBv, NBv] (row-major, nonzero = neighbour) or NULL.
 * Replaces: hunyuan :208-277 (wan21 :206-271) and the rectification masks :348-355. */
int rsa_select_mask(const rsa_layout* lay, const uint8_t* neighbor, int top_k, float p_remain,
                    const rsa_buffers* buf, void* stream);

/* K4 -- comp = w @ vbar.  Replaces torch.matmul(attn_pool_novalid, value_pool), hunyuan :357. */
int rsa_compensation(const rsa_layout* lay, const rsa_buffers* buf, void* stream);

/* K5 -- block-sparse flash attention over the kept lists with the fused rectification epilogue
 *   O = (acc / l) * R + comp   for visual query blocks, exact dense attention for text query rows,
 * written straight into `out` ([B, S, H, D] element strides: the reference's final permute+reshape,
 * hunyuan :383-387, becomes a strided store).
 * Replaces: _triton_block_sparse_attention_onehot + kernel hunyuan :15-168, the R/comp combine :365,
 * the text-row flash call :371-380 and the concat :383. */
int rsa_block_sparse_fwd(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                         const rsa_buffers* buf, rsa_out4 out, void* stream);

/* The whole operator = K1..K5 on one stream.  Replaces rectified_block_sparse_attention /
 * block_sparse_attention_combined (hunyuan :283-417, flux :282-405, cogvideo :282-407, wan21 :276-386). */
int rsa_rectified_attention(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                            const uint8_t* neighbor, int top_k, float p_remain, void* workspace,
                            size_t workspace_bytes, rsa_out4 out, void* stream);

/* Dense attention with the reference's two-segment varlen semantics (attn.py:107-120 as called from
 * hunyuan :503-524): query rows < q_split attend kv [0, kv_split); rows >= q_split attend kv [kv_split, Sk).
 * q: [B,H,Sq,D], k/v: [B,H,Sk,D]; out: [B,Sq,H,D]-strided.  With q_split = Sq, kv_split = Sk it is plain
 * softmax attention (fullattn mode "torch"/"vanilla", attn.py:101-106, :121-149, without bias). */
int rsa_dense_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                  int q_split, int kv_split, rsa_out4 out, void* stream);
/* The same with causal = True (attn.py:60-73 forwards it to flash_attn_varlen_func, :108-116): inside a segment key j
 * is visible to row i iff j <= i + (keys - rows) -- flash-attn's bottom-right alignment, which is the top-left mask of
 * the reference's "torch" / "vanilla" modes (attn.py:105, :129-133) whenever a segment has as many keys as rows. */
int rsa_dense_causal_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                         int q_split, int kv_split, rsa_out4 out, void* stream);

/* Dense attention with an arbitrary mask: fullattn's "torch" / "vanilla" modes when attn_mask depends on the query row
 * (attn.py:101-106: SDPA takes any mask broadcastable to [b, a, s, s1]; :134-147: boolean masks become -inf, float masks are
 * added to the scores).  mask: DEVICE pointer; mask_kind: RSA_MASK_BOOL (1 byte per element, 0 = not attended),
 * RSA_MASK_ADD_2BYTE (additive, in the dtype of q) or RSA_MASK_ADD_F32; mask_stride_*: ELEMENT strides of its [b, a, s, s1]
 * view, 0 for a broadcast dimension.  A query row without any attended key gives NaN when empty_rows_nan != 0 (an explicit
 * softmax over an all -inf row: the reference's "vanilla" mode) and zeros otherwise (torch's fused SDPA since 2.5: its "torch" mode).
 * The plain kernel of the family (no pipeline of the reference builds such a mask; key masks go through rsa_dense_fwd's key
 * limit): expect a fraction of rsa_dense_fwd's rate.  D = 64 or 128; out as rsa_dense_fwd. */
#define RSA_MASK_BOOL 1
#define RSA_MASK_ADD_2BYTE 2
#define RSA_MASK_ADD_F32 3
int rsa_dense_masked_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                         const void* mask, int mask_kind, int64_t mask_stride_b, int64_t mask_stride_h,
                         int64_t mask_stride_q, int64_t mask_stride_k, int empty_rows_nan, rsa_out4 out, void* stream);

/* The same kernel with DROPOUT on the attention weights (fullattn's drop_rate: attn.py:104-106 hands it to SDPA as dropout_p,
 * :148 applies torch.dropout to the softmax's output): every weight is kept with probability 1 - drop_rate and scaled by
 * 1 / (1 - drop_rate), the softmax's denominator keeps all of them.  The keep decisions are a counter-based hash of (seed, batch *
 * head, query row, key): reproducible for a seed, not torch's Philox stream (which cannot be reproduced from outside) -- what the
 * reference's semantics fix is the distribution.  mask may be NULL with mask_kind = 0; causal != 0: key j is attended by row i
 * only if j <= i (the reference's "torch" / "vanilla" triangle).  drop_rate in [0, 1].  Since 0.5.0. */
int rsa_dense_dropout_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                          const void* mask, int mask_kind, int64_t mask_stride_b, int64_t mask_stride_h,
                          int64_t mask_stride_q, int64_t mask_stride_k, int causal, int empty_rows_nan, float drop_rate,
                          uint64_t seed, rsa_out4 out, void* stream);

/* Stand-alone GAPR for callers of estimate_pr_gain (gapr_mask.py:4): blocks are [BH, N, 128, D] contiguous
 * 2-byte elements, pools [BH, N, D] fp32, scores [BH, NQ, NK] fp32 -> mask [BH, NQ, NK] uint8 (1 = ~gapr_mask). */
int rsa_estimate_pr_gain(int BH, int NQ, int NK, int D, int dtype, const void* q_blocks, const void* k_blocks,
                         const float* q_pools, const float* k_pools, const float* scores, float* scratch_aq,
                         float* scratch_ak, uint8_t* mask_out, void* stream);

/* ---- host-side geometry: produces the hot path's `neighbor` input and the token permutation ---- */

/* Generalized Hilbert ("Gilbert") curve over a t x h x w cuboid (x spans w, y spans h, z spans t; linear index =
 * z*h*w + y*w + x).  axis_order: 3 chars out of "w","h","t" = major, middle, minor axis (the scripts pass "wht",
 * main_hunyuan.py:245) or NULL for the size-ordered default.  HOST pointers; linear_to_hilbert may be NULL.
 * Replaces gilbert_mapping / gilbert_xyz2d (utils/jenga_gilbert.py:12-288, :458-504). */
int rsa_gilbert_mapping(int t, int h, int w, const char* axis_order, int32_t* linear_to_hilbert,
                        int32_t* hilbert_to_linear);

/* 26-neighbourhood relation between `block_size`-token blocks along the curve: neighbor[i*NB + j] = 1 iff some
 * point of block i touches (Chebyshev distance <= 1) a point of block j; NB = ceil(t*h*w / block_size).  HOST
 * pointer, NB*NB bytes.  Replaces gilbert_block_neighbor_mapping (utils/jenga_gilbert.py:613-693). */
int rsa_gilbert_block_neighbors(int t, int h, int w, int block_size, const char* axis_order, uint8_t* neighbor);

/* ---- producers / consumers either side of the path; DEVICE pointers ---- */

/* Row gather out[b, i, :] = x[b, order[i], :] for i < S; rows are C 2-byte elements (C % 8 == 0), strides in
 * elements.  Replaces hidden_states[:, hilbert_order] and hidden_states[:, linear_to_hilbert]
 * (scripts/main_hunyuan.py:88, :183; main_wan21t2v.py:92-93, :167). */
int rsa_permute_tokens(int B, int S, int C, const void* x, int64_t x_stride_b, int64_t x_stride_s,
                       const int32_t* order, void* out, int64_t o_stride_b, int64_t o_stride_s, void* stream);

/* Fused per-head RMSNorm (if apply_norm; weight fp32 [D] or NULL) + rotary embedding (if cos/sin non-NULL: fp32
 * [S_rope, D] tables in the interleaved-pair convention of diffusers' apply_rotary_emb, applied to tokens < S_rope)
 * over a [B,H,S,D] view, written to a strided [B,H,S,D] destination (e.g. a slice of the [visual | text] concat).
 * Replaces attn.norm_q / attn.norm_k + apply_rotary_emb + torch.cat in the processors
 * (rectified_hunyuan_attn.py:452-498, rectified_flux_attn.py:443-484). */
int rsa_qk_norm_rope(int B, int H, int S, int D, int dtype, rsa_tensor4 x, const float* weight, float eps,
                     int apply_norm, const float* cos, const float* sin, int S_rope, rsa_out4 y, void* stream);

/* Same pass with torch.nn.LayerNorm over the head dim instead of RMSNorm (CogVideoX's qk_norm = "layer_norm",
 * rectified_cogvideo_attn.py:455-466): fp32 mean / biased variance, (x - mean) * rstd * weight + bias, one rounding; weight
 * and bias [D] fp32 copies or null; tokens [0, S_rope) are rotated (the visual tokens come first). */
int rsa_qk_layernorm_rope(int B, int H, int S, int D, int dtype, rsa_tensor4 x, const float* weight, const float* bias,
                          float eps, const float* cos, const float* sin, int S_rope, rsa_out4 y, void* stream);

/* The Wan producers in one pass: RMSNorm ACROSS heads (one variance per token over all H*D channels; weight [H*D] as an
 * fp32 copy, or null; skipped when apply_norm = 0) and the rotary embedding per head, x [B, S, H*D] (row stride
 * x_stride_s elements) -> y strided [B,H,S,D].  rope_kind 0: none; 1: freqs_a = complex128 [S, D/2] (Wan2.1, the rotation
 * runs in fp64 like the reference); 2: freqs_a = cos fp32 [S, D], freqs_b = sin fp32 [S, D] with per-pair values
 * duplicated (Wan2.2).  Replaces attn.norm_q / norm_k + unflatten + apply_rotary_emb in
 * rectified_wan21_attn.py:424-438 and rectified_wan22_attn.py:54-76. */
int rsa_norm_rope_heads(int B, int H, int S, int D, int dtype, const void* x, int64_t x_stride_b, int64_t x_stride_s,
                        const float* weight, float eps, int apply_norm, int rope_kind, const void* freqs_a,
                        const void* freqs_b, rsa_out4 y, void* stream);

/* ---- fp8 operands for K5 (BASELINE config "fp8 Q/K/V on CDNA4 fp8 MFMA"; the reference has no fp8 path, its P/Q
 * rounding rule "operands in the input dtype, fp32 statistics" (rectified_hunyuan_attn.py:61-62, :97) is kept) ---- */

/* e4m3 (OCP e4m3fn) images of one call's Q, K, V in the BLOCK-SCALED format, written by rsa_pool_stats_fp8 /
 * rsa_quantize_fp8 and read by rsa_block_sparse_fwd_fp8.  Per tensor and 128-row block: y = q * sm_scale*log2(e) | k - mu
 * | v in fp32, A = max |y| over the block's valid rows, scale 2^e with the smallest e such that A * 2^-e <= 448, bytes =
 * e4m3(y * 2^-e) rounded to nearest even.  mu ("smooth K", exact under softmax) = the mean of up to 8 evenly spaced full
 * 128-row blocks of K, so nothing needs a pass of its own over a tensor.  The kernel applies the scales through the fp8
 * MFMA's E8M0 block-scale operands. */
typedef struct rsa_fp8_operands {
    uint8_t* q8;      /* [BH, NB_total*128, D]     rows >= S are zero                                                       */
    uint8_t* k8;      /* [BH, NB_total*128, D]     rows >= pool_valid are zero (needs pool_valid >= kv_valid, kv_text_valid) */
    uint8_t* v8t;     /* [BH, NB_total*2, D, 64]   V^T per 64-key tile, keys in the MFMA k-slot order (rsa_fp8_emit.h)       */
    uint32_t* scales; /* [BH, NB_total] words: byte 0 / 1 / 2 = E8M0 exponent (127 + e) of the Q / K / V block, byte 3 unused;
                       * followed by the K mean mu, [BH, D] fp32                                                            */
} rsa_fp8_operands;

/* Bytes of the four members (member order) and their 256-B-rounded sum.  D = 128 or 64. */
int rsa_fp8_operand_bytes(const rsa_layout* lay, size_t sizes[4], size_t* total);
int rsa_carve_fp8_operands(const rsa_layout* lay, void* ws, size_t ws_bytes, rsa_fp8_operands* out);

/* Stand-alone producer: mu (one small launch over 8 blocks of K), then ONE pass over Q, K, V that writes the three images
 * and the block exponents. */
int rsa_quantize_fp8(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                     const rsa_fp8_operands* ops, void* stream);

/* The same fused into the mask-selection pass: K1 (rsa_pool_stats) writes the images of every block it pools in the pass
 * that pools it (Q and K visual blocks, every V block: the tensors are read from HBM once), a small launch covers the
 * text-tail blocks of Q and K.  buf gets K1's statistics as rsa_pool_stats would write them; ops is bit-identical to
 * rsa_quantize_fp8's.  With ops->q8 == ops->k8 == NULL only the V image and the V bytes of the exponent words are written (the
 * operands of rsa_block_sparse_fwd_fp8pv; since 0.5.0). */
int rsa_pool_stats_fp8(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v, const rsa_buffers* buf,
                       const rsa_fp8_operands* ops, void* stream);

/* K5 on v_mfma_f32_32x32x64_f8f6f4: same lists, R, comp, text rows and output layout as rsa_block_sparse_fwd;
 * lay->dtype selects the OUTPUT element type. */
int rsa_block_sparse_fwd_fp8(const rsa_layout* lay, const rsa_fp8_operands* ops, const rsa_buffers* buf,
                             rsa_out4 out, void* stream);

/* The "pv" form (since 0.5.0): Q . K^T on the 2-byte q and k themselves (v_mfma_f32_32x32x16), e4m3 only for P and V
 * (ops->v8t and the V bytes of ops->scales; q8 / k8 are not read).  The scores are then the 2-byte path's -- the e4m3 rounding of
 * Q and K is nine tenths of rsa_block_sparse_fwd_fp8's error -- while P . V still runs at the fp8 rate: 800 matrix cycles per 64
 * keys and 32 rows against 1 024 (2-byte) and 544 (e4m3) at head dim 128.  Head dims 64 and 128. */
int rsa_block_sparse_fwd_fp8pv(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, const rsa_fp8_operands* ops,
                               const rsa_buffers* buf, rsa_out4 out, void* stream);

/* The whole operator with fp8 K5: K1..K4 on the 2-byte inputs (the mask is the bf16 path's, bit for bit; K1 in its
 * rsa_pool_stats_fp8 form, which leaves the images behind), then rsa_block_sparse_fwd_fp8. */
int rsa_rectified_attention_fp8(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                const uint8_t* neighbor, int top_k, float p_remain, void* workspace,
                                size_t workspace_bytes, void* fp8_workspace, size_t fp8_workspace_bytes,
                                rsa_out4 out, void* stream);
/* ... in the pv form (head dims 64, 128): K1 writes only the V image and its exponents, K5 = rsa_block_sparse_fwd_fp8pv.  Same workspaces
 * (the Q / K images' share of fp8_workspace stays untouched). */
int rsa_rectified_attention_fp8pv(const rsa_layout* lay, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                  const uint8_t* neighbor, int top_k, float p_remain, void* workspace,
                                  size_t workspace_bytes, void* fp8_workspace, size_t fp8_workspace_bytes,
                                  rsa_out4 out, void* stream);

/* TeaCache's step-skipping statistic (scripts/main_hunyuan.py:120, main_wan21t2v.py:112): out2[0] =
 * sum |a - b|, out2[1] = sum |b| over two equal-length 2-byte tensors (n elements, 16-B aligned) in one HBM pass
 * instead of the reference's five elementwise/reduction launches; scratch: >= 2048 floats.  DEVICE pointers. */
int rsa_rel_l1(const void* a, const void* b, int64_t n, int dtype, float* out2, float* scratch, void* stream);

/* rsa_dense_fwd with e4m3 operands (block-scaled images produced inside): quantisation pass + the fp8 kernel in dense
 * mode.  workspace: >= *total of rsa_dense_fp8_bytes, 256-B aligned.  D = 128 or 64. */
int rsa_dense_fp8_bytes(int B, int H, int Sq, int Sk, int D, size_t* total);
int rsa_dense_fwd_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                      int q_split, int kv_split, void* workspace, size_t workspace_bytes, rsa_out4 out, void* stream);
/* ... with causal = True inside each segment (rsa_dense_causal_fwd's meaning: bottom-right aligned, attn.py:108-116). */
int rsa_dense_causal_fwd_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                             int q_split, int kv_split, void* workspace, size_t workspace_bytes, rsa_out4 out,
                             void* stream);
/* The "pv" form of the dense fp8 kernel (round 5; head dims 64, 128): Q . K^T on the 2-byte q and k as they are, e4m3 only for P and the
 * V image (rsa_block_sparse_fwd_fp8pv's kernel in its dense mode).  causal != 0: as rsa_dense_causal_fwd.  Same workspace. */
int rsa_dense_fwd_fp8pv(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                        int q_split, int kv_split, int causal, void* workspace, size_t workspace_bytes, rsa_out4 out, void* stream);

/* ---- multi-GPU: the exchange step at the layer boundary.  The path itself shards by (batch, head) with
 * NO collective (reference: no cross-head dependency anywhere, rectified_hunyuan_attn.py:211-277, gapr_mask.py:15-42);
 * only a consumer that is not head-sharded (the to_out GEMM of an unsharded model) needs every rank's O.  The
 * reference's only multi-GPU code is prompt-level replication (eval/video/experiments/multigpu_hunyuan.py:290-299). ---- */

/* RCCL communicator over the ranks of one node (librccl is bound at run time).  rsa_comm_unique_id: rank 0 fills
 * id128 (HOST, 128 bytes) and distributes it by any host channel; every rank then calls rsa_comm_create. */
int rsa_comm_unique_id(void* id128);
int rsa_comm_create(int world, int rank, const void* id128, void** comm);
int rsa_comm_destroy(void* comm);
/* The rank count the communicator itself reports (ncclCommCount): lets a launcher record that the collective really spans the
 * ranks it believes it does (since 0.5.0). */
int rsa_comm_count(void* comm, int* ranks);

/* All-gather of O along the head axis: local [rows][local_row_bytes] on every rank -> full [rows][world*local_row_bytes]
 * on every rank (rank r's bytes at column offset r*local_row_bytes) = ncclAllGather into `staging`
 * ([world][rows][local_row_bytes]) + one unpack kernel.  DEVICE pointers, 16-byte aligned, local_row_bytes % 16 == 0. */
int rsa_allgather_heads(void* comm, int world, const void* local, void* staging, void* full, int64_t rows,
                        int64_t local_row_bytes, void* stream);

/* The same exchange written straight into the peers: ONE copy kernel whose workgroups are dealt over the peers (all xGMI
 * links carry their slab at once; 16-byte stores into column range [rank*local_row_bytes, +local_row_bytes) of every
 * rank's full buffer), arrival signalled through IPC-shared device flags and awaited by a one-workgroup kernel on the same
 * stream: stream-ordered end to end, no host barrier, no host synchronisation.
 *   full_of_rank / state_of_rank: HOST arrays of `world` device pointers (own buffers at index rank; the others opened with
 *   rsa_ipc_open + rsa_ipc_offset).  state = the block rsa_p2p_state_alloc() returns (FINE-GRAINED device memory, zeroed: peers
 *   write its flags over xGMI while this GPU polls them, which the memory model only defines for fine-grained allocations;
 *   rsa_p2p_state_bytes() of it are used); its word 1 turns nonzero (missing rank + 1) if a wait gave up after 4 s.
 * Every rank must call this the same number of times.  A rank may overwrite a peer's full buffer as soon as that peer has
 * issued its NEXT exchange, so alternate between two full buffers and consume each on the stream that issued the exchange. */
int rsa_p2p_state_bytes(void);
int rsa_p2p_state_alloc(void** state);   /* fine-grained, zeroed, IPC-exportable; one per rank and exchange object */
int rsa_p2p_state_free(void* state);
int rsa_p2p_state_timeout(const void* state, int* missing_rank_plus_1);   /* synchronous read of the time-out word */
int rsa_allgather_heads_p2p(int world, int rank, const void* local, void* const* full_of_rank, void* const* state_of_rank,
                            int64_t rows, int64_t local_row_bytes, void* stream);
int rsa_ipc_export(const void* dev_ptr, void* handle64);                    /* 64-byte handle of a device allocation */
int rsa_ipc_open(const void* handle64, int peer_device, void** dev_ptr);    /* map a peer's allocation here */
int rsa_ipc_close(void* dev_ptr);
int rsa_ipc_offset(const void* dev_ptr, int64_t* offset);                   /* dev_ptr - base of its allocation (a handle names the allocation) */

/* Tuning / diagnostics hook, not part of the data path.  Keys: "k5_w64" (head dim 128: 1 = the 64-rows-per-wave K5, the
 * product; 0 = the 32-row kernel, for A/B), "k5_gsync" (aligned starts of the sparse walks: bit 0 the 64-row kernel -- default --,
 * bit 1 the other K5 kernels, 0 off; a scheduling aid, outputs are byte-identical), "k5_text_last", "k5_tail_split" (0: the
 * walks of the last partial generation stay whole -- byte-identical results whatever the grid), "k5_tsplit" (0/1: split-KV of the text query blocks),
 * "k3_prefix" (0/1: sorted-head path of K3), "fp8_variant" (0: the product = hand-placed block, P through the e4m3 code map; 1: the same arithmetic as hipcc schedules it;
 * 2: P by v_exp_f32 + round-to-nearest e4m3 -- the two forms the tests compare the product with), "fp8_smooth_k"
 * (0: the fp8 producers take mu = 0 instead of the sampled K mean, for the same comparison).  The hook
 * is inert (RSA_ERR_UNSUPPORTED) unless the process was started with the environment variable RSA_TUNING=1. */
int rsa_set_tuning(const char* key, int value);

const char* rsa_status_string(int status);
/* hipGetErrorString of the HIP error behind the most recent RSA_ERR_LAUNCH (diagnostics only). */
const char* rsa_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RSA_H_ */
