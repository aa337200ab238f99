// FP8 operand producer for the fp8 K5 (rsa_attn_fp8_kernel.hip): the block-scaled e4m3 images of Q, K, V (format and
// per-block arithmetic: rsa_fp8_emit.h; CPU restatement: oracle.fp8_operands).  Everything is produced in ONE pass over the
// tensors:
//   * "smooth K": mu[bh][d] = the mean of up to 8 evenly spaced full 128-row blocks of K (kmean_sample_kernel; each block mean in
//     K1's arithmetic, block means added in order, divided by their number).  q.(k - mu) = q.k - q.mu shifts all scores of
//     a query row by the same amount, so every softmax -- kept-block, text-row, dense -- is unchanged in exact arithmetic,
//     while the common component real K tensors carry no longer eats the e4m3 mantissa; a sampled mean removes it as well
//     as the full one and costs 8 blocks instead of a pass.
//   * fused form (rsa_pool_stats_fp8): K1 writes the images of the blocks it pools (rsa_stats.hip, F8 instances); the
//     text-tail blocks of Q and K, which K1 does not pool, go through fp8_blocks_kernel below.
//   * stand-alone (rsa_quantize_fp8) and dense (rsa_dense_fwd_fp8) forms: fp8_blocks_kernel over every block.
#include "rsa_fp8_emit.h"

namespace {

struct BlocksArgs {
    const unsigned short* src[3];
    long sb[3], sh[3], ss[3];
    int blk0[3], blk1[3];   // block range [blk0, blk1) this launch covers, per tensor
    int H;
    Fp8Emit f8;
};

// the K1 load: thread (c, g) gets rows 16 i + g, elements 8c .. 8c+7 of block blk, rows >= valid as zero (2 D threads)
// (t: the thread's index inside its group of 2 D threads -- the whole workgroup except in kmean_sample_kernel)
template <int D, typename Tag>
__device__ __forceinline__ void load_block(const unsigned short* base, long ss, int blk, int valid, float (&x)[8][8], int t) {
    const int c = t % (D / 8), g = t / (D / 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = blk * RSA_BLOCK + 16 * i + g;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row < valid) raw = *reinterpret_cast<const uint4*>(base + (long)row * ss + c * 8);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[i][2 * e] = rsa_to_f32<Tag>((unsigned short)(w[e] & 0xFFFF));
            x[i][2 * e + 1] = rsa_to_f32<Tag>((unsigned short)(w[e] >> 16));
        }
    }
}

// grid (max block count, BH, tensors); 2 D threads
template <int D, typename Tag>
__global__ __launch_bounds__(2 * D) void fp8_blocks_kernel(BlocksArgs a) {
    const int which = blockIdx.z, bh = blockIdx.y;
    const int blk = a.blk0[which] + blockIdx.x;
    if (blk >= a.blk1[which]) return;
    const int b = bh / a.H, h = bh % a.H;
    float x[8][8];
    load_block<D, Tag>(a.src[which] + (long)b * a.sb[which] + (long)h * a.sh[which], a.ss[which], blk, a.f8.valid[which], x, threadIdx.x);
    __shared__ __attribute__((aligned(16))) unsigned char f8lds[rsa_f8_lds(D)];
    fp8_emit_block<D, Tag>(x, a.f8, which, blk, bh, f8lds);
}

// mu[bh][d]: grid (BH), 1 024 threads = G groups of 2 D.  Block mean = K1's (contract C2): per thread the 8 rows in order, xor-tree
// over the row groups of a wave, (w0 + w1) + (w2 + w3) over the waves of the group (w0 + w1 at head dim 64), times 1/128.  The up to 8
// sampled blocks are pooled side by side (round 6: one or two blocks per group of threads, every load issued before the first reduction,
// instead of one block after the other -- eight dependent trips to memory, 16.5 us at any size); their means are then added in block order by one thread per column,
// starting from zero, and divided by their number: the same bits as the serial form.
template <int D, typename Tag>
__global__ __launch_bounds__(1024) void kmean_sample_kernel(const unsigned short* k, long sb, long sh, long ss, int H, int valid,
                                                            float* kmean) {
    constexpr int CH = D / 8, TPG = 2 * D, NW = TPG / 64, G = 1024 / TPG;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int g = threadIdx.x / TPG, t = threadIdx.x % TPG, c = t % CH;
    const unsigned short* base = k + (long)b * sb + (long)h * sh;
    __shared__ float red[G][NW][D];
    __shared__ float bmean[8][D];
    const int nb = valid / RSA_BLOCK > 0 ? valid / RSA_BLOCK : 1;
    const int n = valid <= 0 ? 0 : (nb < 8 ? nb : 8);
    // group g pools blocks g, g + G, ...: PER = 8 / G of them, all loaded before the first is reduced (one trip to memory)
    // (kept as loaded -- 8 x 16 bytes per block -- and widened block by block: 1 024 threads leave 128 registers each)
    constexpr int PER = 8 / G;
    uint4 raw[PER][8];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int i = g + G * p;
        const int blk = i < n ? (int)(((long)i * nb) / n) : 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {          // load_block's rows and columns: thread (c, t / CH) reads rows 16 r + t / CH, elements 8 c .. 8 c + 7
            const int row = blk * RSA_BLOCK + 16 * r + t / CH;
            raw[p][r] = make_uint4(0, 0, 0, 0);
            if (i < n && row < valid) raw[p][r] = *reinterpret_cast<const uint4*>(base + (long)row * ss + c * 8);
        }
    }
#pragma unroll
    for (int p = 0; p < PER; ++p) {           // (uniform trip count: every group meets the barriers, a group past the last block idles)
        const int i = g + G * p;
        const bool on = i < n;
        if (on) {
            float s[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned w[4] = {raw[p][r].x, raw[p][r].y, raw[p][r].z, raw[p][r].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = rsa_to_f32<Tag>((unsigned short)(w[e] & 0xFFFF)), hi = rsa_to_f32<Tag>((unsigned short)(w[e] >> 16));
                    s[2 * e] = r == 0 ? lo : s[2 * e] + lo;
                    s[2 * e + 1] = r == 0 ? hi : s[2 * e + 1] + hi;
                }
            }
#pragma unroll
            for (int m = CH; m < 64; m <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = s[e] + __shfl_xor(s[e], m, 64);
            if ((t & 63) < CH)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[g][t >> 6][c * 8 + e] = s[e];
        }
        __syncthreads();
        if (on && t < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float tot = red[g][0][c * 8 + e] + red[g][1][c * 8 + e];
                if constexpr (NW == 4) tot = tot + (red[g][2][c * 8 + e] + red[g][3][c * 8 + e]);
                bmean[i][c * 8 + e] = tot * (1.0f / RSA_BLOCK);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < CH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = 0.0f;
            for (int i = 0; i < n; ++i) acc = acc + bmean[i][c * 8 + e];
            kmean[(long)bh * D + c * 8 + e] = n > 0 ? acc / (float)n : 0.0f;
        }
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int rsa_fp8_operand_bytes(const rsa_layout* l, size_t sizes[4], size_t* total) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!sizes || !total) return RSA_ERR_BAD_ARG;
    if (l->D != 128 && l->D != 64) return RSA_ERR_UNSUPPORTED;
    const size_t BH = (size_t)l->B * l->H, SP = (size_t)l->NB_total * RSA_BLOCK, D = l->D;
    const size_t s[4] = {BH * SP * D, BH * SP * D, BH * SP * D,
                         ((size_t)l->NB_total + D) * BH * 4};  // block exponents, K mean
    size_t tot = 0;
    for (int i = 0; i < 4; ++i) { sizes[i] = s[i]; tot += align256(s[i]); }
    *total = tot;
    return RSA_OK;
}

extern "C" int rsa_carve_fp8_operands(const rsa_layout* l, void* ws, size_t ws_bytes, rsa_fp8_operands* out) {
    size_t sizes[4], total;
    int st = rsa_fp8_operand_bytes(l, sizes, &total);
    if (st != RSA_OK) return st;
    if (!ws || !out || (reinterpret_cast<uintptr_t>(ws) & 255)) return RSA_ERR_BAD_ARG;
    if (ws_bytes < total) return RSA_ERR_WORKSPACE;
    uint8_t* p = static_cast<uint8_t*>(ws);
    out->q8 = p; p += align256(sizes[0]);
    out->k8 = p; p += align256(sizes[1]);
    out->v8t = p; p += align256(sizes[2]);
    out->scales = reinterpret_cast<uint32_t*>(p);
    return RSA_OK;
}

namespace {

float* kmean_of(const rsa_fp8_operands* ops, int BH, int NB_total) {
    return reinterpret_cast<float*>(ops->scales + (size_t)BH * NB_total);
}

int g_fp8_smooth_k = 1;   // tuning key "fp8_smooth_k" (0: mu = 0, for the accuracy comparison of the smoothing)

void launch_kmean(int D, int dtype, const rsa_tensor4& k, int BH, int H, int valid, float* kmean, hipStream_t s) {
    if (!g_fp8_smooth_k) { (void)hipMemsetAsync(kmean, 0, (size_t)BH * D * 4, s); return; }
    const unsigned short* kp = static_cast<const unsigned short*>(k.ptr);
#define RSA_KM(DD, TT) kmean_sample_kernel<DD, TT><<<BH, 1024, 0, s>>>(kp, k.stride_b, k.stride_h, k.stride_s, H, valid, kmean)
    if (D == 128) { if (dtype == RSA_BF16) RSA_KM(128, bf16_tag); else RSA_KM(128, fp16_tag); }
    else { if (dtype == RSA_BF16) RSA_KM(64, bf16_tag); else RSA_KM(64, fp16_tag); }
#undef RSA_KM
}

void launch_blocks(BlocksArgs& a, int D, int dtype, int BH, int ntensors, hipStream_t s) {
    int nmax = 0;
    for (int i = 0; i < ntensors; ++i) nmax = a.blk1[i] - a.blk0[i] > nmax ? a.blk1[i] - a.blk0[i] : nmax;
    if (nmax <= 0) return;
    const dim3 g(nmax, BH, ntensors);
    if (D == 128) {
        if (dtype == RSA_BF16) fp8_blocks_kernel<128, bf16_tag><<<g, 256, 0, s>>>(a);
        else fp8_blocks_kernel<128, fp16_tag><<<g, 256, 0, s>>>(a);
    } else {
        if (dtype == RSA_BF16) fp8_blocks_kernel<64, bf16_tag><<<g, 128, 0, s>>>(a);
        else fp8_blocks_kernel<64, fp16_tag><<<g, 128, 0, s>>>(a);
    }
}

int fill_args(const rsa_layout* l, const rsa_tensor4& q, const rsa_tensor4& k, const rsa_tensor4& v,
              const rsa_fp8_operands* ops, BlocksArgs& a, bool v_only = false) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (l->D != 128 && l->D != 64) return RSA_ERR_UNSUPPORTED;
    if (!ops || !ops->v8t || !ops->scales) return RSA_ERR_BAD_ARG;
    if (!v_only && (!ops->q8 || !ops->k8)) return RSA_ERR_BAD_ARG;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v))) return st;
    // every key K5 may read unmasked must be a row the images hold
    if (l->pool_valid < l->kv_valid || l->pool_valid < l->kv_text_valid) return RSA_ERR_BAD_ARG;
    const rsa_tensor4* ts[3] = {&q, &k, &v};
    for (int i = 0; i < 3; ++i) {
        a.src[i] = static_cast<const unsigned short*>(ts[i]->ptr);
        a.sb[i] = ts[i]->stride_b; a.sh[i] = ts[i]->stride_h; a.ss[i] = ts[i]->stride_s;
        a.blk0[i] = 0; a.blk1[i] = l->NB_total;
    }
    a.H = l->H;
    Fp8Emit& f = a.f8;
    f.q8 = ops->q8; f.k8 = ops->k8; f.v8t = ops->v8t; f.exps = ops->scales;
    f.kmean = kmean_of(ops, l->B * l->H, l->NB_total);
    f.qk_const = (float)((1.0 / sqrt((double)l->D)) * 1.44269504);
    f.S_pad = l->NB_total * RSA_BLOCK; f.NB_total = l->NB_total;
    f.valid[0] = l->S; f.valid[1] = l->pool_valid; f.valid[2] = l->pool_valid;
    return RSA_OK;
}

}  // namespace

void rsa_set_fp8_smooth_k(int v) { g_fp8_smooth_k = v; }

// stand-alone producer: mu, then one pass over every block of the three tensors
extern "C" int rsa_quantize_fp8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                const rsa_fp8_operands* ops, void* stream) {
    BlocksArgs a;
    int st = fill_args(l, q, k, v, ops, a);
    if (st != RSA_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_kmean(l->D, l->dtype, k, l->B * l->H, l->H, l->pool_valid, const_cast<float*>(a.f8.kmean), s);
    launch_blocks(a, l->D, l->dtype, l->B * l->H, 3, s);
    return rsa_launch_status();
}

// K1 (pool statistics) writing the images of the blocks it pools; the text-tail blocks of Q and K behind it.
extern "C" int rsa_pool_stats_fp8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                  const rsa_buffers* buf, const rsa_fp8_operands* ops, void* stream) {
    BlocksArgs a;
    // ops->q8 == ops->k8 == NULL: only the V image and the V exponents (what rsa_block_sparse_fwd_fp8pv reads): the Q and K blocks
    // are pooled without being quantised, no K mean, no text-tail pass
    const bool v_only = ops && !ops->q8 && !ops->k8;
    int st = fill_args(l, q, k, v, ops, a, v_only);
    if (st != RSA_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (v_only) return rsa_pool_stats_f8(l, q, k, v, buf, &a.f8, stream);
    launch_kmean(l->D, l->dtype, k, l->B * l->H, l->H, l->pool_valid, const_cast<float*>(a.f8.kmean), s);
    if ((st = rsa_pool_stats_f8(l, q, k, v, buf, &a.f8, stream))) return st;
    a.blk0[0] = a.blk0[1] = l->NBv;   // q and k blocks K1 does not pool
    launch_blocks(a, l->D, l->dtype, l->B * l->H, 2, s);
    return rsa_launch_status();
}

// ---- dense attention operands (rsa_dense_fwd_fp8): q [B,H,Sq,D], k/v [B,H,Sk,D]; workspace carved here ----
static size_t dense_fp8_carve(int BH, int Sq, int Sk, int D, void* ws, rsa_fp8_operands* ops, int* sqp, int* skp) {
    const size_t SQ = (size_t)((Sq + RSA_BLOCK - 1) / RSA_BLOCK) * RSA_BLOCK;
    const size_t SK = (size_t)((Sk + RSA_BLOCK - 1) / RSA_BLOCK) * RSA_BLOCK;
    const size_t SM = SQ > SK ? SQ : SK;   // one image height for all three (the block kernel's S_pad)
    const size_t s[4] = {align256(BH * SM * D), align256(BH * SM * D), align256(BH * SM * D),
                         align256((SM / RSA_BLOCK + D) * BH * 4)};  // block exponents, K mean
    if (ops) {
        uint8_t* p = static_cast<uint8_t*>(ws);
        ops->q8 = p; p += s[0];
        ops->k8 = p; p += s[1];
        ops->v8t = p; p += s[2];
        ops->scales = reinterpret_cast<uint32_t*>(p);
    }
    if (sqp) *sqp = (int)SM;
    if (skp) *skp = (int)SM;
    return s[0] + s[1] + s[2] + s[3];
}

extern "C" int rsa_dense_fp8_bytes(int B, int H, int Sq, int Sk, int D, size_t* total) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || !total) return RSA_ERR_BAD_ARG;
    if (D != 128 && D != 64) return RSA_ERR_UNSUPPORTED;
    *total = dense_fp8_carve(B * H, Sq, Sk, D, nullptr, nullptr, nullptr, nullptr);
    return RSA_OK;
}

// internal: producer for the dense kernel (declared in rsa_common.h).  Both images are S_pad = max(Sq, Sk) rounded to 128 rows.
int rsa_dense_quantize_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                           void* ws, size_t ws_bytes, rsa_fp8_operands* ops, hipStream_t s, int v_only) {
    if (D != 128 && D != 64) return RSA_ERR_UNSUPPORTED;
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return RSA_ERR_BAD_ARG;
    int sqp, skp;
    if (ws_bytes < dense_fp8_carve(B * H, Sq, Sk, D, ws, ops, &sqp, &skp)) return RSA_ERR_WORKSPACE;
    BlocksArgs a;
    const rsa_tensor4* ts[3] = {&q, &k, &v};
    const int rows[3] = {Sq, Sk, Sk};
    for (int i = 0; i < 3; ++i) {
        a.src[i] = static_cast<const unsigned short*>(ts[i]->ptr);
        a.sb[i] = ts[i]->stride_b; a.sh[i] = ts[i]->stride_h; a.ss[i] = ts[i]->stride_s;
        a.blk0[i] = 0; a.blk1[i] = (rows[i] + RSA_BLOCK - 1) / RSA_BLOCK;
        a.f8.valid[i] = rows[i];
    }
    a.H = H;
    Fp8Emit& f = a.f8;
    const int nb = sqp / RSA_BLOCK;
    f.q8 = ops->q8; f.k8 = ops->k8; f.v8t = ops->v8t; f.exps = ops->scales;
    float* kmean = reinterpret_cast<float*>(ops->scales + (size_t)B * H * nb);
    f.kmean = kmean;
    f.qk_const = (float)((1.0 / sqrt((double)D)) * 1.44269504);
    f.S_pad = sqp; f.NB_total = nb;
    if (v_only) {   // the pv form: the V image and its exponents only (no Q / K blocks, no K mean)
        f.q8 = nullptr; f.k8 = nullptr;
        a.blk1[0] = 0; a.blk1[1] = 0;
    } else {
        launch_kmean(D, dtype, k, B * H, H, Sk, kmean, s);
    }
    launch_blocks(a, D, dtype, B * H, 3, s);
    return rsa_launch_status();
}
