// Producers / consumers either side of the attention path; all HBM-bound, 16 B per lane.
//   rsa_permute_tokens : row gather  out[b, i, :] = x[b, order[i], :]   (Hilbert permute in / out)
//   rsa_qk_norm_rope   : per-head RMSNorm (optional) + rotary embedding (optional) in one pass, writing into a
//                        strided [B,S,H,D] destination (so the visual/text concat needs no extra copy)
// Compiled with -ffp-contract=off: the arithmetic follows diffusers' RMSNorm / apply_rotary_emb operation by
// operation (fp32 statistics, the same intermediate roundings), so results match the PyTorch ops to the last bit
// except where rsqrt or the mean's summation order differ by an fp32 ulp.
#include "rsa_common.h"

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void permute_tokens_kernel(const uint4* x, long xsb, long xss, const int32_t* order,
                                                            uint4* out, long osb, long oss, int S, int C16) {
    // one wave per output row chunk: grid.x covers rows, lanes stride over the row's 16-byte chunks
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= S) return;
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int src = order[row];
    const uint4* xp = x + ((long)b * xsb + (long)src * xss);
    uint4* op = out + ((long)b * osb + (long)row * oss);
    // four 16-byte chunks per lane in flight before the first store (a load -> store loop waits for each load on its own)
    for (int c0 = lane; c0 < C16; c0 += 256) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < C16) v[u] = xp[c0 + 64 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c0 + 64 * u < C16) op[c0 + 64 * u] = v[u];
    }
}

extern "C" int rsa_permute_tokens(int B, int S, int C, const void* x, int64_t x_stride_b, int64_t x_stride_s,
                                  const int32_t* order, void* out, int64_t o_stride_b, int64_t o_stride_s,
                                  void* stream) {
    if (B <= 0 || S <= 0 || C <= 0 || !x || !order || !out) return RSA_ERR_BAD_ARG;
    if ((C % 8) || (x_stride_b % 8) || (x_stride_s % 8) || (o_stride_b % 8) || (o_stride_s % 8)) return RSA_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return RSA_ERR_BAD_ARG;
    dim3 grid((S + 3) / 4, B);
    permute_tokens_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
        static_cast<const uint4*>(x), x_stride_b / 8, x_stride_s / 8, order, static_cast<uint4*>(out), o_stride_b / 8,
        o_stride_s / 8, S, C / 8);
    return rsa_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
template <typename Tag>
__device__ __forceinline__ unsigned short rsa_from_f32(float f);
template <>
__device__ __forceinline__ unsigned short rsa_from_f32<bf16_tag>(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <>
__device__ __forceinline__ unsigned short rsa_from_f32<fp16_tag>(float f) {
    return __builtin_bit_cast(unsigned short, (_Float16)f);
}

struct NormRopeArgs {
    const unsigned short* x;
    long xsb, xsh, xss;
    unsigned short* y;
    long ysb, ysh, yss;
    const float *weight, *bias, *cos, *sin;
    float eps;
    int H, S, S_rope, apply_norm;   // apply_norm: 0 none, 1 RMSNorm, 2 LayerNorm (weight and bias optional)
};

// D/8 lanes per token row, 8 elements (16 B) per lane; a 256-thread block covers 256/(D/8) tokens and one group of
// heads: every thread loads its 8 cos / 8 sin values ONCE per token and reuses them for all heads of the group (the
// fp32 tables are 4x the bytes of a 2-byte row; re-reading them per head made the kernel table-bound).
template <int D, typename Tag>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(NormRopeArgs a, int heads_per_group) {
    constexpr int LPR = D / 8;            // lanes per row (16 or 8)
    constexpr int TPB = 256 / LPR;        // tokens per block
    const int t = threadIdx.x, c = t % LPR;
    const int s = blockIdx.x * TPB + t / LPR;
    const int b = blockIdx.z;
    const int h0 = blockIdx.y * heads_per_group;
    if (s >= a.S) return;
    const bool rope = a.cos != nullptr && s < a.S_rope;
    float cs[8], sn[8];
    if (rope) {
        const float4* cp = reinterpret_cast<const float4*>(a.cos + (long)s * D + c * 8);
        const float4* sp = reinterpret_cast<const float4*>(a.sin + (long)s * D + c * 8);
        const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
        sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
    }
    float wt[8], bs[8];
    if (a.apply_norm && a.weight) {
#pragma unroll
        for (int e = 0; e < 8; ++e) wt[e] = a.weight[c * 8 + e];
    }
    if (a.apply_norm == 2 && a.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = a.bias[c * 8 + e];
    }
    const unsigned short* xp = a.x + (long)b * a.xsb + (long)s * a.xss + c * 8;
    unsigned short* yp = a.y + (long)b * a.ysb + (long)s * a.yss + c * 8;
    const int h1 = min(h0 + heads_per_group, a.H);
    // the next head's 16 bytes are requested before this head's arithmetic (a dynamic-trip loop otherwise waits for each
    // load on its own)
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (h0 < h1) nxt = *reinterpret_cast<const uint4*>(xp + (long)h0 * a.xsh);
    for (int h = h0; h < h1; ++h) {
        const uint4 raw = nxt;
        if (h + 1 < h1) nxt = *reinterpret_cast<const uint4*>(xp + (long)(h + 1) * a.xsh);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = rsa_to_f32<Tag>((unsigned short)(w[e] & 0xFFFF));
            v[2 * e + 1] = rsa_to_f32<Tag>((unsigned short)(w[e] >> 16));
        }
        if (a.apply_norm == 2) {
            // torch.nn.LayerNorm over the head dim (CogVideoX's qk_norm): fp32 mean and biased variance, then
            // (x - mean) * rstd * weight + bias in fp32, ONE rounding to the storage type
            float sm = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sm = sm + v[e];
#pragma unroll
            for (int m = 1; m < LPR; m <<= 1) sm = sm + __shfl_xor(sm, m, 64);
            const float mean = sm / (float)D;
            float sq = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sq = sq + (v[e] - mean) * (v[e] - mean);
#pragma unroll
            for (int m = 1; m < LPR; m <<= 1) sq = sq + __shfl_xor(sq, m, 64);
            const float rstd = rsqrtf(sq / (float)D + a.eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float n = (v[e] - mean) * rstd;
                if (a.weight) n = n * wt[e];
                if (a.bias) n = n + bs[e];
                v[e] = rsa_to_f32<Tag>(rsa_from_f32<Tag>(n));
            }
        } else if (a.apply_norm) {
            // variance = mean(x^2) in fp32; x * rsqrt(var + eps); round to the storage type; * weight; round
            float ss = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = ss + v[e] * v[e];
#pragma unroll
            for (int m = 1; m < LPR; m <<= 1) ss = ss + __shfl_xor(ss, m, 64);
            const float r = rsqrtf(ss * (1.0f / D) + a.eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float n = v[e] * r;
                if (a.weight) {
                    n = rsa_to_f32<Tag>(rsa_from_f32<Tag>(n));
                    n = n * wt[e];
                }
                v[e] = rsa_to_f32<Tag>(rsa_from_f32<Tag>(n));
            }
        }
        if (rope) {
            // out = x * cos + rotate_half_pairs(x) * sin, pairs (2i, 2i+1): rot[2i] = -x[2i+1], rot[2i+1] = x[2i]
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                o[e] = v[e] * cs[e] + (-v[e + 1]) * sn[e];
                o[e + 1] = v[e + 1] * cs[e + 1] + v[e] * sn[e + 1];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = o[e];
        }
        uint4 pk;
        pk.x = (unsigned)rsa_from_f32<Tag>(v[0]) | ((unsigned)rsa_from_f32<Tag>(v[1]) << 16);
        pk.y = (unsigned)rsa_from_f32<Tag>(v[2]) | ((unsigned)rsa_from_f32<Tag>(v[3]) << 16);
        pk.z = (unsigned)rsa_from_f32<Tag>(v[4]) | ((unsigned)rsa_from_f32<Tag>(v[5]) << 16);
        pk.w = (unsigned)rsa_from_f32<Tag>(v[6]) | ((unsigned)rsa_from_f32<Tag>(v[7]) << 16);
        *reinterpret_cast<uint4*>(yp + (long)h * a.ysh) = pk;
    }
}

static int launch_qk_norm_rope(int B, int H, int S, int D, int dtype, rsa_tensor4 x, const float* weight, const float* bias,
                               float eps, int apply_norm, const float* cos, const float* sin, int S_rope, rsa_out4 y,
                               void* stream) {
    if (B <= 0 || H <= 0 || S <= 0 || !y.ptr) return RSA_ERR_BAD_ARG;
    if (D != 64 && D != 128) return RSA_ERR_UNSUPPORTED;
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    int st = rsa_check_tensor(x);
    if (st != RSA_OK) return st;
    if ((reinterpret_cast<uintptr_t>(y.ptr) & 15) || (y.stride_b % 8) || (y.stride_h % 8) || (y.stride_s % 8))
        return RSA_ERR_BAD_ARG;
    if ((cos == nullptr) != (sin == nullptr) || S_rope < 0 || S_rope > S) return RSA_ERR_BAD_ARG;
    NormRopeArgs a;
    a.x = static_cast<const unsigned short*>(x.ptr); a.xsb = x.stride_b; a.xsh = x.stride_h; a.xss = x.stride_s;
    a.y = static_cast<unsigned short*>(y.ptr); a.ysb = y.stride_b; a.ysh = y.stride_h; a.yss = y.stride_s;
    a.weight = weight; a.bias = bias; a.cos = cos; a.sin = sin; a.eps = eps; a.H = H; a.S = S; a.S_rope = S_rope;
    a.apply_norm = apply_norm;
    const int tpb = 256 / (D / 8);
    const int hpg = H >= 8 ? 8 : H;  // heads sharing one cos/sin load
    dim3 grid((unsigned)((S + tpb - 1) / tpb), (unsigned)((H + hpg - 1) / hpg), B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D == 128) {
        if (dtype == RSA_BF16) qk_norm_rope_kernel<128, bf16_tag><<<grid, 256, 0, s>>>(a, hpg);
        else qk_norm_rope_kernel<128, fp16_tag><<<grid, 256, 0, s>>>(a, hpg);
    } else {
        if (dtype == RSA_BF16) qk_norm_rope_kernel<64, bf16_tag><<<grid, 256, 0, s>>>(a, hpg);
        else qk_norm_rope_kernel<64, fp16_tag><<<grid, 256, 0, s>>>(a, hpg);
    }
    return rsa_launch_status();
}

extern "C" int rsa_qk_norm_rope(int B, int H, int S, int D, int dtype, rsa_tensor4 x, const float* weight, float eps,
                                int apply_norm, const float* cos, const float* sin, int S_rope, rsa_out4 y,
                                void* stream) {
    return launch_qk_norm_rope(B, H, S, D, dtype, x, weight, nullptr, eps, apply_norm ? 1 : 0, cos, sin, S_rope, y, stream);
}

extern "C" int rsa_qk_layernorm_rope(int B, int H, int S, int D, int dtype, rsa_tensor4 x, const float* weight,
                                     const float* bias, float eps, const float* cos, const float* sin, int S_rope,
                                     rsa_out4 y, void* stream) {
    return launch_qk_norm_rope(B, H, S, D, dtype, x, weight, bias, eps, 2, cos, sin, S_rope, y, stream);
}

// =====================================================================================================
// rsa_rel_l1: sum |a - b| and sum |b| over two equal-length 2-byte tensors in ONE pass (TeaCache's step-skipping
// statistic, scripts/main_hunyuan.py:120: ((x - prev).abs().mean() / prev.abs().mean()) is their ratio).  Fixed
// reduction order: per thread in address order, per workgroup the C6 tree, then the workgroup partials in order.
// =====================================================================================================
namespace {
constexpr int RL1_WGS = 1024;

// ---------------------------------------------------------------------------------------------------------
// rsa_norm_rope_heads: the Wan producers.  RMSNorm ACROSS heads (one variance per token over all H*D channels, diffusers'
// qk_norm = "rms_norm_across_heads"), then the rotary embedding per head, written straight into a strided [B,H,S,D]
// destination -- one pass instead of attn.norm_q + unflatten + the fp64 complex multiply of Wan2.1
// (rectified_wan21_attn.py:430-438) or the cos/sin form of Wan2.2 (rectified_wan22_attn.py:54-66, :70-76).
// One wave per token: a lane owns the 16-byte chunks lane, lane + 64, ... of the token's row (up to NCH of them).
//   rope_kind 1: fa = complex128 [S, D/2] (cos, sin as doubles).  The rotation runs in fp64 like the reference, then
//                double -> float -> storage type, the conversion chain of torch's .type_as.
//   rope_kind 2: fa = cos fp32 [S, D], fb = sin fp32 [S, D] (per-pair values duplicated, Wan2.2's tables): cos of pair p
//                is cos[2p], sin is sin[2p + 1]; fp32 products and differences, each rounded, as torch evaluates them.
struct NormRopeHeadsArgs {
    const unsigned short* x;
    long xsb, xss;
    unsigned short* y;
    long ysb, ysh, yss;
    const float* weight;
    const void *fa, *fb;
    float eps;
    int S, C, D, apply_norm, rope_kind;
};

template <typename Tag, int NCH>
__global__ __launch_bounds__(256) void norm_rope_heads_kernel(NormRopeHeadsArgs a) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (tok >= a.S) return;
    const int C16 = a.C / 8;
    const uint4* xp = reinterpret_cast<const uint4*>(a.x + (long)b * a.xsb + (long)tok * a.xss);
    uint4 raw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        raw[i] = c < C16 ? xp[c] : make_uint4(0, 0, 0, 0);
    }
    float r = 1.0f;
    if (a.apply_norm) {
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const unsigned w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = rsa_to_f32<Tag>((unsigned short)(w[e] & 0xFFFF)), v1 = rsa_to_f32<Tag>((unsigned short)(w[e] >> 16));
                ss = ss + v0 * v0;
                ss = ss + v1 * v1;
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) ss = ss + __shfl_xor(ss, m, 64);
        r = rsqrtf(ss / (float)a.C + a.eps);
    }
    // a lane's chunks lane, lane + 64, ... are 512 channels apart: when D divides 512 they sit at the SAME offset d inside
    // their heads, so the lane's four rotation pairs are loaded once per token (checked by the host)
    const int dl = (lane * 8) % a.D;
    double2 fc[4];
    float cs[4], sn[4];
    if (a.rope_kind == 1) {
        const double2* fr = reinterpret_cast<const double2*>(a.fa) + (long)tok * (a.D / 2) + dl / 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) fc[e] = fr[e];
    } else if (a.rope_kind == 2) {
        const float* cp = reinterpret_cast<const float*>(a.fa) + (long)tok * a.D + dl;
        const float* sp = reinterpret_cast<const float*>(a.fb) + (long)tok * a.D + dl;
        const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
        cs[0] = c0.x; cs[1] = c0.z; cs[2] = c1.x; cs[3] = c1.z;     // cos[2p]
        sn[0] = s0.y; sn[1] = s0.w; sn[2] = s1.y; sn[3] = s1.w;     // sin[2p + 1]
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c >= C16) continue;
        const int ch = c * 8, head = ch / a.D, d = dl;
        const unsigned w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = rsa_to_f32<Tag>((unsigned short)(w[e] & 0xFFFF));
            v[2 * e + 1] = rsa_to_f32<Tag>((unsigned short)(w[e] >> 16));
        }
        if (a.apply_norm) {
            float wt[8];
            if (a.weight) {
                const float4 w0 = *reinterpret_cast<const float4*>(a.weight + ch), w1 = *reinterpret_cast<const float4*>(a.weight + ch + 4);
                wt[0] = w0.x; wt[1] = w0.y; wt[2] = w0.z; wt[3] = w0.w; wt[4] = w1.x; wt[5] = w1.y; wt[6] = w1.z; wt[7] = w1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float n = v[e] * r;
                if (a.weight) {
                    n = rsa_to_f32<Tag>(rsa_from_f32<Tag>(n));
                    n = n * wt[e];
                }
                v[e] = rsa_to_f32<Tag>(rsa_from_f32<Tag>(n));
            }
        }
        if (a.rope_kind == 1) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const double2 f = fc[e / 2];
                const double x0 = (double)v[e], x1 = (double)v[e + 1];
                const double o0 = x0 * f.x - x1 * f.y, o1 = x0 * f.y + x1 * f.x;
                v[e] = (float)o0;
                v[e + 1] = (float)o1;
            }
        } else if (a.rope_kind == 2) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float x0 = v[e], x1 = v[e + 1];
                v[e] = x0 * cs[e / 2] - x1 * sn[e / 2];
                v[e + 1] = x0 * sn[e / 2] + x1 * cs[e / 2];
            }
        }
        uint4 pk;
        pk.x = (unsigned)rsa_from_f32<Tag>(v[0]) | ((unsigned)rsa_from_f32<Tag>(v[1]) << 16);
        pk.y = (unsigned)rsa_from_f32<Tag>(v[2]) | ((unsigned)rsa_from_f32<Tag>(v[3]) << 16);
        pk.z = (unsigned)rsa_from_f32<Tag>(v[4]) | ((unsigned)rsa_from_f32<Tag>(v[5]) << 16);
        pk.w = (unsigned)rsa_from_f32<Tag>(v[6]) | ((unsigned)rsa_from_f32<Tag>(v[7]) << 16);
        *reinterpret_cast<uint4*>(a.y + (long)b * a.ysb + (long)head * a.ysh + (long)tok * a.yss + d) = pk;
    }
}

extern "C" int rsa_norm_rope_heads(int B, int H, int S, int D, int dtype, const void* x, int64_t x_stride_b,
                                   int64_t x_stride_s, const float* weight, float eps, int apply_norm, int rope_kind,
                                   const void* freqs_a, const void* freqs_b, rsa_out4 y, void* stream) {
    if (B <= 0 || H <= 0 || S <= 0 || D <= 0 || !x || !y.ptr) return RSA_ERR_BAD_ARG;
    if ((D % 8) || (512 % D) || (long)H * D > 64L * 8 * 16) return RSA_ERR_UNSUPPORTED;   // D | 512; a row = at most 16 chunks per lane
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    if (rope_kind < 0 || rope_kind > 2 || (rope_kind >= 1 && !freqs_a) || (rope_kind == 2 && !freqs_b)) return RSA_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (x_stride_b % 8) || (x_stride_s % 8)) return RSA_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(y.ptr) & 15) || (y.stride_b % 8) || (y.stride_h % 8) || (y.stride_s % 8))
        return RSA_ERR_BAD_ARG;
    if (rope_kind >= 1 && (reinterpret_cast<uintptr_t>(freqs_a) & 15)) return RSA_ERR_BAD_ARG;
    if (rope_kind == 2 && (reinterpret_cast<uintptr_t>(freqs_b) & 15)) return RSA_ERR_BAD_ARG;
    if (weight && (reinterpret_cast<uintptr_t>(weight) & 15)) return RSA_ERR_BAD_ARG;
    NormRopeHeadsArgs a;
    a.x = static_cast<const unsigned short*>(x); a.xsb = x_stride_b; a.xss = x_stride_s;
    a.y = static_cast<unsigned short*>(y.ptr); a.ysb = y.stride_b; a.ysh = y.stride_h; a.yss = y.stride_s;
    a.weight = weight; a.fa = freqs_a; a.fb = freqs_b; a.eps = eps; a.S = S; a.C = H * D; a.D = D;
    a.apply_norm = apply_norm; a.rope_kind = rope_kind;
    const int nch = (a.C / 8 + 63) / 64;
    dim3 grid((unsigned)((S + 3) / 4), (unsigned)B);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define RSA_NRH(N) \
    do { \
        if (dtype == RSA_BF16) norm_rope_heads_kernel<bf16_tag, N><<<grid, 256, 0, s>>>(a); \
        else norm_rope_heads_kernel<fp16_tag, N><<<grid, 256, 0, s>>>(a); \
    } while (0)
    if (nch <= 3) RSA_NRH(3);
    else if (nch <= 6) RSA_NRH(6);
    else if (nch <= 10) RSA_NRH(10);
    else RSA_NRH(16);
#undef RSA_NRH
    return rsa_launch_status();
}

template <typename Tag>
__global__ __launch_bounds__(256) void rel_l1_partial_kernel(const uint4* a, const uint4* b, long n16, float* part) {
    __shared__ float red[4];
    float sd = 0.0f, sb = 0.0f;
    auto acc16 = [&](const uint4& ra, const uint4& rb) {
        const unsigned wa[4] = {ra.x, ra.y, ra.z, ra.w}, wb[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = rsa_to_f32<Tag>((unsigned short)(wa[e] & 0xFFFF)), a1 = rsa_to_f32<Tag>((unsigned short)(wa[e] >> 16));
            const float b0 = rsa_to_f32<Tag>((unsigned short)(wb[e] & 0xFFFF)), b1 = rsa_to_f32<Tag>((unsigned short)(wb[e] >> 16));
            sd = sd + fabsf(a0 - b0); sd = sd + fabsf(a1 - b1);
            sb = sb + fabsf(b0); sb = sb + fabsf(b1);
        }
    };
    // four grid strides' loads in flight before the first add; the adds keep the order of the plain loop (same bits)
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        uint4 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) { ra[u] = a[i + u * stride]; rb[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n16) acc16(ra[u], rb[u]);
    }
    const float td = block_tree_sum(sd, red);
    const float tb = block_tree_sum(sb, red);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = td; part[2 * blockIdx.x + 1] = tb; }
}

template <typename Tag>
__global__ void rel_l1_final_kernel(const float* part, int nwg, const unsigned short* a, const unsigned short* b,
                                    long tail0, long n, float* out2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sd = 0.0f, sb = 0.0f;
    for (int i = 0; i < nwg; ++i) { sd = sd + part[2 * i]; sb = sb + part[2 * i + 1]; }
    for (long i = tail0; i < n; ++i) {  // < 8 leftover elements
        const float x = rsa_to_f32<Tag>(a[i]), y = rsa_to_f32<Tag>(b[i]);
        sd = sd + fabsf(x - y); sb = sb + fabsf(y);
    }
    out2[0] = sd; out2[1] = sb;
}
}  // namespace

extern "C" int rsa_rel_l1(const void* a, const void* b, int64_t n, int dtype, float* out2, float* scratch,
                          void* stream) {
    if (!a || !b || !out2 || !scratch || n < 0) return RSA_ERR_BAD_ARG;
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(b) & 15)) return RSA_ERR_BAD_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n16 = n / 8;
    long want = (n16 + 256 * 8 - 1) / (256 * 8);
    const int nwg = (int)(want < 1 ? 1 : (want > RL1_WGS ? RL1_WGS : want));
    const uint4* pa = static_cast<const uint4*>(a);
    const uint4* pb = static_cast<const uint4*>(b);
    const unsigned short* ea = static_cast<const unsigned short*>(a);
    const unsigned short* eb = static_cast<const unsigned short*>(b);
    if (dtype == RSA_BF16) {
        rel_l1_partial_kernel<bf16_tag><<<nwg, 256, 0, s>>>(pa, pb, n16, scratch);
        rel_l1_final_kernel<bf16_tag><<<1, 64, 0, s>>>(scratch, nwg, ea, eb, n16 * 8, n, out2);
    } else {
        rel_l1_partial_kernel<fp16_tag><<<nwg, 256, 0, s>>>(pa, pb, n16, scratch);
        rel_l1_final_kernel<fp16_tag><<<1, 64, 0, s>>>(scratch, nwg, ea, eb, n16 * 8, n, out2);
    }
    return rsa_launch_status();
}
