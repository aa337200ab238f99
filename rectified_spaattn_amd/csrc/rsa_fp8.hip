// FP8 operand producer for the fp8 K5 (rsa_attn_fp8_kernel.hip): per-(batch, head) absolute maxima of Q, K, V and
// the e4m3 (OCP "e4m3fn", gfx950's native fp8) images the kernel stages:
//      q8, k8 : [BH, NB_total*128, D]   row-major bytes, rows past the tensor's valid range (q: S, k and v: pool_valid,
//               the rows the pooling pass counts) are zero
//      v8t    : [BH, NB_total*2, D, 64] V transposed per 64-key tile; byte p = 32*h + j of a (tile, d) row holds the
//               key the f8f6f4 MFMA's k-slot (lane half h, byte j) meets in the P operand built from two 32-key
//               score accumulators:   key = 32*(j >> 4) + (j & 3) + 8*((j & 15) >> 2) + 4*h
//      scales : [4, BH] fp32: dequantisation scales of q, k, v, then c = scale_q * scale_k * sm_scale * log2(e)
//      kmean  : [BH, D] fp32 ("smooth K"): in the fused form (rsa_pool_stats_fp8) the mean over the visual blocks of
//               K1's block means is subtracted from every K row before quantisation.  q.(k - mu) = q.k - q.mu shifts all
//               scores of a query row by the same amount, so every softmax -- kept-block, text-row, dense -- is unchanged
//               in exact arithmetic, while the common component real K tensors carry no longer eats the e4m3 mantissa.
//               The K scale uses the bound amax|k - mu| <= amax|k| + max|mu| (no second pass; free for a float format).
//               The dense path (rsa_dense_fwd_fp8) gets mu from column sums its amax pass takes along; the stand-alone
//               rsa_quantize_fp8 uses mu = 0.
// Numeric contract (bit-exact against oracle.fp8_operands): scale = amax / 448 (1 when the tensor is all zero) for K and
// V; for Q the scale is stretched by less than 2x so that c is an exact power of two -- c = the smallest power of two
// >= (amax_q/448 * scale_k) * qk_const, scale_q = c / (scale_k * qk_const), all in fp32 -- which costs e4m3 (a floating
// point format) no precision and lets the kernel apply c through the MFMA's E8M0 scale operands.  Elements: widen
// exactly to fp32, (subtract the K mean,) multiply by inv = fl(1 / scale), clamp to +-448, v_cvt_pk_fp8_f32 (round to
// nearest even, subnormals kept).
#include "rsa_common.h"

namespace {

constexpr float E4M3_MAX = 448.0f;

struct QuantArgs {
    const unsigned short* src[3];
    long sb[3], sh[3], ss[3];
    int valid[3];        // rows >= valid[i] are zero in the image (and skipped by the amax)
    int lo[3];           // first row the amax kernel reads (rows below were covered by K1's side product)
    unsigned* amax_bits; // [3, BH] fp32 bit patterns (non-negative floats order like unsigned ints)
    float* scales;       // [4, BH]
    float qk_const;      // sm_scale * log2(e)
    const float* kmean;  // [BH, D] or nullptr (no smoothing)
    float* colsum_part;  // optional [BH, nchunk, D]: column sums of K per 1024-row chunk (dense path's K mean)
    int nchunk;
    uint8_t *q8, *k8, *v8t;
    int H, BH;
    int S_pad[3];        // padded rows of each image (multiple of 128)
};

template <typename Tag>
__global__ __launch_bounds__(256) void amax_kernel(QuantArgs a) {
    const int which = blockIdx.z, bh = blockIdx.y;
    const int b = bh / a.H, h = bh % a.H;
    const unsigned short* base = a.src[which] + (long)b * a.sb[which] + (long)h * a.sh[which];
    const int row0 = a.lo[which] + blockIdx.x * 1024;
    const int row1 = min(row0 + 1024, a.valid[which]);
    const int t = threadIdx.x, c = t & 15;
    float m = 0.0f;
    auto fold = [&](uint4 raw) {
        const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m = fmaxf(m, fabsf(rsa_to_f32<Tag>((unsigned short)(w4[e] & 0xFFFF))));
            m = fmaxf(m, fabsf(rsa_to_f32<Tag>((unsigned short)(w4[e] >> 16))));
        }
    };
    const long ss = a.ss[which];
    const bool sums = which == 1 && a.colsum_part != nullptr;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fold_sum = [&](uint4 raw) {  // column sums of K in row order (this thread: rows g, g + 16, ...)
        const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[2 * e] = cs[2 * e] + rsa_to_f32<Tag>((unsigned short)(w4[e] & 0xFFFF));
            cs[2 * e + 1] = cs[2 * e + 1] + rsa_to_f32<Tag>((unsigned short)(w4[e] >> 16));
        }
    };
    int row = row0 + (t >> 4);
    for (; row + 112 < row1; row += 128) {  // 8 independent 16-byte loads in flight per lane
        uint4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (long)(row + 16 * u) * ss + 8 * c);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            fold(raw[u]);
            if (sums) fold_sum(raw[u]);
        }
    }
    for (; row < row1; row += 16) {
        const uint4 raw = *reinterpret_cast<const uint4*>(base + (long)row * ss + 8 * c);
        fold(raw);
        if (sums) fold_sum(raw);
    }
    for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((t & 63) == 0 && m > 0.0f) atomicMax(a.amax_bits + which * a.BH + bh, __float_as_uint(m));
    if (a.colsum_part != nullptr) {  // uniform per launch: every workgroup reaches the barrier
        __shared__ float red[4][128];
        if (sums) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] = cs[e] + __shfl_xor(cs[e], 16, 64);
                cs[e] = cs[e] + __shfl_xor(cs[e], 32, 64);
            }
            if ((t & 63) < 16)
#pragma unroll
                for (int e = 0; e < 8; ++e) red[t >> 6][8 * c + e] = cs[e];
        }
        __syncthreads();
        if (sums && t < 128 && (int)blockIdx.x < a.nchunk)
            a.colsum_part[((long)bh * a.nchunk + blockIdx.x) * 128 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

// dense path's "smooth K": mu[bh][d] = (sum over the 1024-row chunks, in order, of the chunk column sums) / rows
__global__ __launch_bounds__(128) void colmean_kernel(const float* part, int nchunk, int rows, float* kmean) {
    const int bh = blockIdx.x, d = threadIdx.x;
    float sum = 0.0f;
    for (int cidx = 0; cidx < nchunk; ++cidx) sum = sum + part[((long)bh * nchunk + cidx) * 128 + d];
    kmean[(long)bh * 128 + d] = rows > 0 ? sum / (float)rows : 0.0f;
}

// "smooth K" vector: mu[bh][d] = tree16(P_0..P_15) / NBv, P_g = sum in block order of kbar[j][d] over j = g (mod 16);
// tree16 = the xor-tree of contract C6 (strides 1, 2, 4, 8).  grid (D / 16, BH), 256 threads = 16 g x 16 d.
__global__ __launch_bounds__(256) void kmean_kernel(const float* kbar, int NBv, int D, float* kmean) {
    const int bh = blockIdx.y, t = threadIdx.x;
    const int g = t & 15, d = blockIdx.x * 16 + (t >> 4);  // the 16 partial sums of one d sit in 16 adjacent lanes
    float p = 0.0f;
    for (int j = g; j < NBv; j += 16) p = p + kbar[((long)bh * NBv + j) * D + d];
    p = p + __shfl_xor(p, 1, 64);
    p = p + __shfl_xor(p, 2, 64);
    p = p + __shfl_xor(p, 4, 64);
    p = p + __shfl_xor(p, 8, 64);
    if (g == 0) kmean[(long)bh * D + d] = NBv > 0 ? p / (float)NBv : 0.0f;
}

// one wave per (b,h): amax = max(atomic word, per-block maxima written by K1) -> scales (K: + max |mu| when smoothing)
__global__ __launch_bounds__(64) void scales_kernel(const unsigned* amax_bits, const float* amax_part, int NB_total,
                                                    int nb_q, int nb_k, int nb_v, float* scales, int BH,
                                                    float qk_const, int D, const float* kmean) {
    const int bh = blockIdx.x, lane = threadIdx.x;
    float mu_max = 0.0f;
    if (kmean != nullptr) {
        for (int d = lane; d < D; d += 64) mu_max = fmaxf(mu_max, fabsf(kmean[(long)bh * D + d]));
        for (int s = 1; s < 64; s <<= 1) mu_max = fmaxf(mu_max, __shfl_xor(mu_max, s, 64));
    }
    float sc[3];
    const int nb[3] = {nb_q, nb_k, nb_v};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float m = __uint_as_float(amax_bits[i * BH + bh]);
        if (amax_part != nullptr)
            for (int j = lane; j < nb[i]; j += 64) m = fmaxf(m, amax_part[((long)i * BH + bh) * NB_total + j]);
        for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
        if (i == 1) m = m + mu_max;  // bound on amax |k - mu|
        sc[i] = m > 0.0f ? m / E4M3_MAX : 1.0f;
    }
    if (lane != 0) return;
    const float skc = sc[1] * qk_const;
    const float c0 = sc[0] * skc;
    int e;
    const float mant = frexpf(c0, &e);              // c0 = mant * 2^e, mant in [0.5, 1)
    if (mant == 0.5f) e -= 1;                        // already a power of two
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    const float c = ldexpf(1.0f, e);
    scales[bh] = c / skc;
    scales[BH + bh] = sc[1];
    scales[2 * BH + bh] = sc[2];
    scales[3 * BH + bh] = c;
}

// `inv` = 1 / scale (one IEEE division per thread); the per-element step is a single multiply
__device__ __forceinline__ float q_clamp(float x, float inv) {
    const float y = x * inv;
    return fminf(fmaxf(y, -E4M3_MAX), E4M3_MAX);
}

// 8 two-byte elements (one uint4) -> 8 e4m3 bytes (uint2); mu (8 floats) is subtracted first when given
template <typename Tag>
__device__ __forceinline__ uint2 quant8(uint4 raw, float inv, const float* mu = nullptr) {
    const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float lo = rsa_to_f32<Tag>((unsigned short)(w4[e] & 0xFFFF)), hi = rsa_to_f32<Tag>((unsigned short)(w4[e] >> 16));
        if (mu != nullptr) { lo = lo - mu[2 * e]; hi = hi - mu[2 * e + 1]; }
        f[2 * e] = q_clamp(lo, inv);
        f[2 * e + 1] = q_clamp(hi, inv);
    }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

// Q and K: 64 rows x D per workgroup, row-major bytes.  grid (S_pad / 64, BH, 2); a thread owns 32 elements of one row
template <int D, typename Tag, bool SMOOTH_K>
__global__ __launch_bounds__(256) void quant_rows_kernel(QuantArgs a) {
    constexpr int TPR = D / 32;  // threads per row
    const int which = blockIdx.z, bh = blockIdx.y;
    if (blockIdx.x * 64 >= a.S_pad[which]) return;
    const int b = bh / a.H, h = bh % a.H;
    const float scale = 1.0f / a.scales[which * a.BH + bh];  // reciprocal: elements are multiplied
    uint8_t* dst = (which == 0 ? a.q8 : a.k8) + (long)bh * a.S_pad[which] * D;
    const unsigned short* base = a.src[which] + (long)b * a.sb[which] + (long)h * a.sh[which];
    const int t = threadIdx.x, part = t % TPR;
    const int row = blockIdx.x * 64 + t / TPR;
    __shared__ __attribute__((aligned(16))) float smu[D];
    if (SMOOTH_K && which == 1) {  // the head's K mean: one global read per workgroup, LDS reads per thread
        if (t < D) smu[t] = a.kmean[(long)bh * D + t];
        __syncthreads();
    }
    uint2 o[4];
    if (row < a.valid[which]) {
        const unsigned short* p = base + (long)row * a.ss[which] + 32 * part;
        uint4 raw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const uint4*>(p + 8 * i);
        if (SMOOTH_K && which == 1) {
            const float4* mp = reinterpret_cast<const float4*>(smu + 32 * part);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 m0 = mp[2 * i], m1 = mp[2 * i + 1];
                const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                o[i] = quant8<Tag>(raw[i], scale, mu);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = quant8<Tag>(raw[i], scale);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = make_uint2(0, 0);
    }
    uint4* out = reinterpret_cast<uint4*>(dst + (long)row * D + 32 * part);
    out[0] = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
    out[1] = make_uint4(o[2].x, o[2].y, o[3].x, o[3].y);
}

// V: one 64-key tile per workgroup, transposed through LDS into the k-slot key order.  grid (S_pad / 64, BH)
template <int D, typename Tag>
__global__ __launch_bounds__(256) void quant_vt_kernel(QuantArgs a) {
    constexpr int TPR = D / 32;
    constexpr int LROW = D + 16;  // padded LDS row (bytes)
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * LROW];
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const float scale = 1.0f / a.scales[2 * a.BH + bh];  // reciprocal: elements are multiplied
    const unsigned short* base = a.src[2] + (long)b * a.sb[2] + (long)h * a.sh[2];
    const int t = threadIdx.x;
    for (int rr = t / TPR; rr < 64; rr += 256 / TPR) {
        const int row = blockIdx.x * 64 + rr, part = t % TPR;
        uint2 o[4];
        if (row < a.valid[2]) {
            const unsigned short* p = base + (long)row * a.ss[2] + 32 * part;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = quant8<Tag>(*reinterpret_cast<const uint4*>(p + 8 * i), scale);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = make_uint2(0, 0);
        }
        uint4* out = reinterpret_cast<uint4*>(tile + rr * LROW + 32 * part);
        out[0] = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
        out[1] = make_uint4(o[2].x, o[2].y, o[3].x, o[3].y);
    }
    __syncthreads();
    uint8_t* dst = a.v8t + ((long)bh * (a.S_pad[2] / 64) + blockIdx.x) * (long)(D * 64);
    for (int item = t; item < D * 2; item += 256) {
        const int d = item >> 1, hh = item & 1;
        unsigned w[8];
#pragma unroll
        for (int j4 = 0; j4 < 8; ++j4) {
            unsigned word = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * j4 + e;
                const int key = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * hh;
                word |= (unsigned)tile[key * LROW + d] << (8 * e);
            }
            w[j4] = word;
        }
        uint4* out = reinterpret_cast<uint4*>(dst + d * 64 + 32 * hh);
        out[0] = make_uint4(w[0], w[1], w[2], w[3]);
        out[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int rsa_fp8_operand_bytes(const rsa_layout* l, size_t sizes[4], size_t* total) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!sizes || !total) return RSA_ERR_BAD_ARG;
    if (l->D != 128) return RSA_ERR_UNSUPPORTED;
    const size_t BH = (size_t)l->B * l->H, SP = (size_t)l->NB_total * RSA_BLOCK, D = l->D;
    const size_t s[4] = {BH * SP * D, BH * SP * D, BH * SP * D,
                         (4 + 3 + 3 * (size_t)l->NB_total + D) * BH * 4};  // scales, amax words, per-block maxima, K mean
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
    out->scales = reinterpret_cast<float*>(p);
    return RSA_OK;
}

namespace {

int fill_args(const rsa_layout* l, const rsa_tensor4& q, const rsa_tensor4& k, const rsa_tensor4& v,
              const rsa_fp8_operands* ops, QuantArgs& a) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (l->D != 128) return RSA_ERR_UNSUPPORTED;
    if (!ops || !ops->q8 || !ops->k8 || !ops->v8t || !ops->scales) return RSA_ERR_BAD_ARG;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v))) return st;
    // every key K5 may read unmasked must be a row the images hold
    if (l->pool_valid < l->kv_valid || l->pool_valid < l->kv_text_valid) return RSA_ERR_BAD_ARG;
    const rsa_tensor4* ts[3] = {&q, &k, &v};
    for (int i = 0; i < 3; ++i) {
        a.src[i] = static_cast<const unsigned short*>(ts[i]->ptr);
        a.sb[i] = ts[i]->stride_b; a.sh[i] = ts[i]->stride_h; a.ss[i] = ts[i]->stride_s;
        a.lo[i] = 0;
    }
    a.valid[0] = l->S; a.valid[1] = l->pool_valid; a.valid[2] = l->pool_valid;
    a.H = l->H; a.BH = l->B * l->H;
    a.S_pad[0] = a.S_pad[1] = a.S_pad[2] = l->NB_total * RSA_BLOCK;
    a.scales = ops->scales;
    a.amax_bits = reinterpret_cast<unsigned*>(ops->scales + 4 * a.BH);
    a.qk_const = (float)((1.0 / sqrt((double)l->D)) * 1.44269504);
    a.q8 = ops->q8; a.k8 = ops->k8; a.v8t = ops->v8t;
    a.kmean = nullptr; a.colsum_part = nullptr; a.nchunk = 0;
    return RSA_OK;
}

// the K mean lives behind the scales, the amax words and K1's per-block maxima
float* kmean_ptr(const rsa_layout* l, const QuantArgs& a) {
    return reinterpret_cast<float*>(a.amax_bits) + (size_t)(3 + 3 * l->NB_total) * a.BH;
}

void launch_amax(const QuantArgs& a, int dtype, int ntensors, hipStream_t s) {
    int rows = 0;
    for (int i = 0; i < ntensors; ++i) rows = a.valid[i] - a.lo[i] > rows ? a.valid[i] - a.lo[i] : rows;
    if (rows <= 0) return;
    const dim3 g((rows + 1023) / 1024, a.BH, ntensors);
    if (dtype == RSA_BF16) amax_kernel<bf16_tag><<<g, 256, 0, s>>>(a);
    else amax_kernel<fp16_tag><<<g, 256, 0, s>>>(a);
}

int launch_images(const QuantArgs& a, int dtype, hipStream_t s) {
    const int sp = a.S_pad[0] > a.S_pad[1] ? a.S_pad[0] : a.S_pad[1];
    const dim3 g_rows(sp / 64, a.BH, 2), g_vt(a.S_pad[2] / 64, a.BH);
    if (dtype == RSA_BF16) {
        if (a.kmean) quant_rows_kernel<128, bf16_tag, true><<<g_rows, 256, 0, s>>>(a);
        else quant_rows_kernel<128, bf16_tag, false><<<g_rows, 256, 0, s>>>(a);
        quant_vt_kernel<128, bf16_tag><<<g_vt, 256, 0, s>>>(a);
    } else {
        if (a.kmean) quant_rows_kernel<128, fp16_tag, true><<<g_rows, 256, 0, s>>>(a);
        else quant_rows_kernel<128, fp16_tag, false><<<g_rows, 256, 0, s>>>(a);
        quant_vt_kernel<128, fp16_tag><<<g_vt, 256, 0, s>>>(a);
    }
    return rsa_launch_status();
}

}  // namespace

// stand-alone producer: amax pass + scales + images
extern "C" int rsa_quantize_fp8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                const rsa_fp8_operands* ops, void* stream) {
    QuantArgs a;
    int st = fill_args(l, q, k, v, ops, a);
    if (st != RSA_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(a.amax_bits, 0, (size_t)3 * a.BH * 4, s) != hipSuccess) return rsa_launch_status();
    launch_amax(a, l->dtype, 3, s);
    scales_kernel<<<a.BH, 64, 0, s>>>(a.amax_bits, nullptr, 0, 0, 0, 0, a.scales, a.BH, a.qk_const, 0, nullptr);
    return launch_images(a, l->dtype, s);
}

// K1 (pool statistics) with the |x| maxima folded in: K1 already reads every Q (visual), K (visual) and V row, so
// only the text-tail rows of Q and K need a (tiny) extra amax launch; then the scales.  Same scales, bit for bit, as
// the stand-alone pass (a max does not depend on the order).
extern "C" int rsa_pool_stats_fp8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                  const rsa_buffers* buf, const rsa_fp8_operands* ops, void* stream) {
    QuantArgs a;
    int st = fill_args(l, q, k, v, ops, a);
    if (st != RSA_OK) return st;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(a.amax_bits, 0, (size_t)3 * a.BH * 4, s) != hipSuccess) return rsa_launch_status();
    float* amax_part = reinterpret_cast<float*>(a.amax_bits + 3 * a.BH);
    if ((st = rsa_pool_stats_amax(l, q, k, v, buf, amax_part, stream))) return st;
    const int vis_tok = l->NBv * RSA_BLOCK;
    a.lo[0] = vis_tok < l->S ? vis_tok : l->S;                    // q rows K1 did not read
    a.lo[1] = vis_tok < l->pool_valid ? vis_tok : l->pool_valid;  // k rows K1 did not read
    launch_amax(a, l->dtype, 2, s);
    kmean_kernel<<<dim3(l->D / 16, a.BH), 256, 0, s>>>(buf->kbar, l->NBv, l->D, kmean_ptr(l, a));
    scales_kernel<<<a.BH, 64, 0, s>>>(a.amax_bits, amax_part, l->NB_total, l->NBv, l->NBv, l->NB_total, a.scales, a.BH,
                                      a.qk_const, l->D, kmean_ptr(l, a));
    return rsa_launch_status();
}

// the three e4m3 images from scales that are already in ops->scales (after rsa_pool_stats_fp8)
extern "C" int rsa_fp8_images(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                              const rsa_fp8_operands* ops, void* stream) {
    QuantArgs a;
    int st = fill_args(l, q, k, v, ops, a);
    if (st != RSA_OK) return st;
    a.kmean = kmean_ptr(l, a);  // written by rsa_pool_stats_fp8
    return launch_images(a, l->dtype, static_cast<hipStream_t>(stream));
}

// ---- dense attention operands (rsa_dense_fwd_fp8): q [B,H,Sq,D], k/v [B,H,Sk,D]; workspace carved here ----
static size_t dense_fp8_carve(int BH, int Sq, int Sk, int D, void* ws, rsa_fp8_operands* ops, int* sqp, int* skp) {
    const size_t SQ = (size_t)((Sq + RSA_BLOCK - 1) / RSA_BLOCK) * RSA_BLOCK;
    const size_t SK = (size_t)((Sk + RSA_BLOCK - 1) / RSA_BLOCK) * RSA_BLOCK;
    const size_t nchunk = (SK + 1023) / 1024;
    const size_t s[4] = {align256(BH * SQ * D), align256(BH * SK * D), align256(BH * SK * D),
                         align256((size_t)(7 + D + nchunk * D) * BH * 4)};  // scales, amax words, K mean, chunk sums
    if (ops) {
        uint8_t* p = static_cast<uint8_t*>(ws);
        ops->q8 = p; p += s[0];
        ops->k8 = p; p += s[1];
        ops->v8t = p; p += s[2];
        ops->scales = reinterpret_cast<float*>(p);
    }
    if (sqp) *sqp = (int)SQ;
    if (skp) *skp = (int)SK;
    return s[0] + s[1] + s[2] + s[3];
}

extern "C" int rsa_dense_fp8_bytes(int B, int H, int Sq, int Sk, int D, size_t* total) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || !total) return RSA_ERR_BAD_ARG;
    if (D != 128) return RSA_ERR_UNSUPPORTED;
    *total = dense_fp8_carve(B * H, Sq, Sk, D, nullptr, nullptr, nullptr, nullptr);
    return RSA_OK;
}

// internal: producer for the dense kernel (declared in rsa_common.h)
int rsa_dense_quantize_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                           void* ws, size_t ws_bytes, rsa_fp8_operands* ops, hipStream_t s) {
    if (D != 128) return RSA_ERR_UNSUPPORTED;
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return RSA_ERR_BAD_ARG;
    int sqp, skp;
    if (ws_bytes < dense_fp8_carve(B * H, Sq, Sk, D, ws, ops, &sqp, &skp)) return RSA_ERR_WORKSPACE;
    QuantArgs a;
    const rsa_tensor4* ts[3] = {&q, &k, &v};
    for (int i = 0; i < 3; ++i) {
        a.src[i] = static_cast<const unsigned short*>(ts[i]->ptr);
        a.sb[i] = ts[i]->stride_b; a.sh[i] = ts[i]->stride_h; a.ss[i] = ts[i]->stride_s;
        a.lo[i] = 0;
    }
    a.valid[0] = Sq; a.valid[1] = Sk; a.valid[2] = Sk;
    a.S_pad[0] = sqp; a.S_pad[1] = skp; a.S_pad[2] = skp;
    a.H = H; a.BH = B * H;
    a.scales = ops->scales;
    a.amax_bits = reinterpret_cast<unsigned*>(ops->scales + 4 * a.BH);
    a.qk_const = (float)((1.0 / sqrt((double)D)) * 1.44269504);
    a.q8 = ops->q8; a.k8 = ops->k8; a.v8t = ops->v8t;
    if (hipMemsetAsync(a.amax_bits, 0, (size_t)3 * a.BH * 4, s) != hipSuccess) return rsa_launch_status();
    // smooth K: the amax pass also sums K's columns per 1024-row chunk; mu = their ordered sum / Sk
    float* kmean = reinterpret_cast<float*>(a.amax_bits) + 3 * a.BH;
    a.colsum_part = kmean + (size_t)D * a.BH;
    a.nchunk = (Sk + 1023) / 1024;
    a.kmean = nullptr;
    launch_amax(a, dtype, 3, s);
    colmean_kernel<<<a.BH, 128, 0, s>>>(a.colsum_part, a.nchunk, Sk, kmean);
    a.kmean = kmean;
    scales_kernel<<<a.BH, 64, 0, s>>>(a.amax_bits, nullptr, 0, 0, 0, 0, a.scales, a.BH, a.qk_const, D, kmean);
    return launch_images(a, dtype, s);
}
