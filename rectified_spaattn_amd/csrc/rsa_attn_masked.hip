// Dense attention with an ARBITRARY mask: fullattn's "torch" / "vanilla" modes with an attn_mask that depends on the query
// row -- a boolean [b, 1|a, s, s1] mask (False = not attended) or an additive float one (attn.py:101-106 hands SDPA any
// broadcastable mask, :134-147 builds the bias the same way).  No pipeline of the reference uses such a mask (its masks are
// key masks, served by the K5 kernels through their key limit), so this is the plain kernel of the family, written for
// coverage, not for the roofline: one wave = 32 query rows, 32-key tiles through LDS, both GEMMs on
// v_mfma_f32_32x32x16 in the orientation of the K5 kernels (S^T = K.Q^T, keys on registers / query rows on lanes, so the
// softmax is lane-local and the packed P is the B operand of O^T += V^T.P^T as it stands), an online softmax per tile.
// A row without any attended key gives NaN (an explicit softmax over an all -inf row: the reference's "vanilla" mode) or zeros
// (what torch's fused SDPA returns for such a row since 2.5: the reference's "torch" mode), as the caller asks.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rsa_attn.h"

namespace {

struct MaskedArgs {
    const unsigned short *q, *k, *v;
    long qsb, qsh, qss, ksb, ksh, kss, vsb, vsh, vss;
    unsigned short* out;
    long osb, osh, oss;
    const void* mask;
    int mask_kind;                // RSA_MASK_BOOL / RSA_MASK_ADD_2BYTE / RSA_MASK_ADD_F32
    long msb, msh, msq, msk;      // element strides of the mask's [b, a, s, s1] view (0 = broadcast)
    int H, Sq, Sk;
    int empty_nan;                // a row without attended keys: 1 = NaN, 0 = zeros
    float scale_log2;             // log2(e) / sqrt(D)
    int causal;                   // key j attended by row i only if j <= i (the top-left triangle of attn.py:105, :129-133)
    unsigned drop_thresh;         // dropout: an attention weight is dropped when its 24-bit hash < drop_thresh (0 = no dropout)
    float keep_scale;             // 1 / (1 - drop_rate)
    unsigned long long seed;
};

// one uniform 24-bit number per (seed, batch*head, query row, key): SplitMix64's finaliser over the packed coordinates.  The
// stream is this library's own (torch's Philox offsets are not reproducible from outside); what the contract of
// torch.dropout / SDPA's dropout_p fixes is the distribution: independent Bernoulli(1 - p) keeps, kept weights x 1 / (1 - p).
__device__ __forceinline__ unsigned drop_hash(unsigned long long seed, int bh, int row, int key) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((((unsigned long long)(unsigned)bh) << 42) ^ (((unsigned long long)(unsigned)row) << 21) ^ (unsigned long long)(unsigned)key);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 40);
}

template <typename Tag, int D>
__global__ __launch_bounds__(256) void dense_masked_kernel(MaskedArgs a) {
    constexpr int KS = D / 16, DT = D / 32;
    constexpr int KLD = D + 8;        // K tile row stride (elements): 16-byte aligned rows, skewed banks
    constexpr int VLD = 32 + 8;       // V^T tile row stride
    using E = Elem<Tag>;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[32 * KLD];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[D * VLD];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int grow = blockIdx.x * 128 + 32 * wv + r;
    const int qrow = grow < a.Sq ? grow : a.Sq - 1;
    const unsigned short* qp = a.q + (long)b * a.qsb + (long)h * a.qsh + (long)qrow * a.qss;
    s16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const s16x8*>(qp + 16 * ks + 8 * hh);
    const unsigned short* kb = a.k + (long)b * a.ksb + (long)h * a.ksh;
    const unsigned short* vb = a.v + (long)b * a.vsb + (long)h * a.vsh;
    const char* mrow = reinterpret_cast<const char*>(a.mask);
    const long moff = (long)b * a.msb + (long)h * a.msh + (long)qrow * a.msq;

    f32x16 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[dt][j] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int ntiles = (a.Sk + 31) / 32;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        __syncthreads();
        // stage the K tile row-major and the V tile transposed; keys past the end as zeros
        for (int c = t; c < 32 * (D / 8); c += 256) {
            const int row = c / (D / 8), c8 = c % (D / 8);
            const int key = key0 + row;
            uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            if (key < a.Sk) {
                kk = *reinterpret_cast<const uint4*>(kb + (long)key * a.kss + 8 * c8);
                vv = *reinterpret_cast<const uint4*>(vb + (long)key * a.vss + 8 * c8);
            }
            *reinterpret_cast<uint4*>(&Ks[row * KLD + 8 * c8]) = kk;
            const unsigned w4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(8 * c8 + 2 * e) * VLD + row] = (unsigned short)(w4[e] & 0xFFFF);
                Vt[(8 * c8 + 2 * e + 1) * VLD + row] = (unsigned short)(w4[e] >> 16);
            }
        }
        __syncthreads();
        // S^T = K . Q^T : element i of lane (r, hh) = key (i & 3) + 8 (i >> 2) + 4 hh, query row r
        f32x16 S;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 kf = *reinterpret_cast<const s16x8*>(&Ks[r * KLD + 16 * ks + 8 * hh]);
            S = E::mfma(kf, qf[ks], S);
        }
        float s[16], mloc = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            float x = S[i] * a.scale_log2;
            if (key >= a.Sk || (a.causal && key > qrow)) {
                x = -INFINITY;
            } else if (a.mask_kind == 0) {
            } else if (a.mask_kind == RSA_MASK_BOOL) {
                if (!reinterpret_cast<const unsigned char*>(mrow)[moff + (long)key * a.msk]) x = -INFINITY;
            } else if (a.mask_kind == RSA_MASK_ADD_2BYTE) {
                x += 1.44269504f * rsa_to_f32<Tag>(reinterpret_cast<const unsigned short*>(mrow)[moff + (long)key * a.msk]);
            } else {
                x += 1.44269504f * reinterpret_cast<const float*>(mrow)[moff + (long)key * a.msk];
            }
            s[i] = x;
            mloc = fmaxf(mloc, x);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        // (m_new == -inf: nothing attended so far -- alpha = 1 keeps the zeros; +inf biases are not meaningful input)
        const float alpha = m_new == -INFINITY ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
        float p[16], psum = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            p[i] = m_new == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s[i] - m_new);
            psum += p[i];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (a.drop_thresh != 0u) {      // dropout acts on the softmax's OUTPUT (attn.py:148 / SDPA's dropout_p): the row sum keeps every weight
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                p[i] = drop_hash(a.seed, bh, qrow, key) < a.drop_thresh ? 0.0f : p[i] * a.keep_scale;
            }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 16; ++j) O[dt][j] *= alpha;
        // O^T += V^T . P^T, 16 keys per MFMA: this lane's P values of key half k2 (keys 4 hh + 0..3 and 8 + 4 hh + 0..3 of the
        // half) are the B fragment as they stand; the A fragment reads V^T in the same key order
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const s16x8 pf = E::cvt8(&p[8 * k2]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const unsigned short* vr = &Vt[(32 * dt + r) * VLD + 16 * k2 + 4 * hh];
                const s16x4 lo = *reinterpret_cast<const s16x4*>(vr), hi = *reinterpret_cast<const s16x4*>(vr + 8);
                s16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                O[dt] = E::mfma(vf, pf, O[dt]);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (grow >= a.Sq) return;
    // l_tot == 0 (no attended key): 0 * inf = NaN, like a softmax over an all -inf row -- or zeros, like torch's fused SDPA
    const float inv = (l_tot == 0.0f && !a.empty_nan) ? 0.0f : 1.0f / l_tot;
    unsigned short* op = a.out + (long)b * a.osb + (long)h * a.osh + (long)grow * a.oss;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            uint2 pk;
            pk.x = (unsigned)E::from_f32(O[dt][4 * g] * inv) | ((unsigned)E::from_f32(O[dt][4 * g + 1] * inv) << 16);
            pk.y = (unsigned)E::from_f32(O[dt][4 * g + 2] * inv) | ((unsigned)E::from_f32(O[dt][4 * g + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(op + d0) = pk;
        }
}

}  // namespace

static int dense_masked_launch(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                               const void* mask, int mask_kind, int64_t mask_stride_b, int64_t mask_stride_h,
                               int64_t mask_stride_q, int64_t mask_stride_k, int causal, int empty_rows_nan, float drop_rate,
                               unsigned long long seed, rsa_out4 out, void* stream) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return RSA_ERR_BAD_ARG;
    if (D != 64 && D != 128) return RSA_ERR_UNSUPPORTED;
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    if (mask_kind != 0 && mask_kind != RSA_MASK_BOOL && mask_kind != RSA_MASK_ADD_2BYTE && mask_kind != RSA_MASK_ADD_F32) return RSA_ERR_BAD_ARG;
    if ((mask_kind != 0) != (mask != nullptr)) return RSA_ERR_BAD_ARG;
    if (mask_stride_b < 0 || mask_stride_h < 0 || mask_stride_q < 0 || mask_stride_k < 0) return RSA_ERR_BAD_ARG;
    if (!(drop_rate >= 0.0f) || drop_rate > 1.0f) return RSA_ERR_BAD_ARG;
    int st;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v))) return st;
    if (!out.ptr || (reinterpret_cast<uintptr_t>(out.ptr) & 7) || (out.stride_b % 4) || (out.stride_h % 4) || (out.stride_s % 4))
        return RSA_ERR_BAD_ARG;
    if ((long)B * H > 65535) return RSA_ERR_UNSUPPORTED;
    MaskedArgs a;
    a.q = static_cast<const unsigned short*>(q.ptr); a.k = static_cast<const unsigned short*>(k.ptr);
    a.v = static_cast<const unsigned short*>(v.ptr);
    a.qsb = q.stride_b; a.qsh = q.stride_h; a.qss = q.stride_s;
    a.ksb = k.stride_b; a.ksh = k.stride_h; a.kss = k.stride_s;
    a.vsb = v.stride_b; a.vsh = v.stride_h; a.vss = v.stride_s;
    a.out = static_cast<unsigned short*>(out.ptr); a.osb = out.stride_b; a.osh = out.stride_h; a.oss = out.stride_s;
    a.mask = mask; a.mask_kind = mask_kind;
    a.msb = mask_stride_b; a.msh = mask_stride_h; a.msq = mask_stride_q; a.msk = mask_stride_k;
    a.H = H; a.Sq = Sq; a.Sk = Sk;
    a.empty_nan = empty_rows_nan != 0;
    a.scale_log2 = (float)((1.0 / sqrt((double)D)) * 1.44269504);
    a.causal = causal != 0;
    // drop_rate = 1 drops everything (torch.dropout(p = 1) returns zeros): threshold 2^24 is above every 24-bit hash
    a.drop_thresh = drop_rate > 0.0f ? (unsigned)((double)drop_rate * 16777216.0 + 0.5) : 0u;
    if (drop_rate > 0.0f && a.drop_thresh == 0u) a.drop_thresh = 1u;
    a.keep_scale = drop_rate < 1.0f ? 1.0f / (1.0f - drop_rate) : 0.0f;
    a.seed = seed;
    const dim3 grid((unsigned)((Sq + 127) / 128), (unsigned)(B * H));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (D == 128) {
        if (dtype == RSA_BF16) dense_masked_kernel<bf16_tag, 128><<<grid, 256, 0, s>>>(a);
        else dense_masked_kernel<fp16_tag, 128><<<grid, 256, 0, s>>>(a);
    } else {
        if (dtype == RSA_BF16) dense_masked_kernel<bf16_tag, 64><<<grid, 256, 0, s>>>(a);
        else dense_masked_kernel<fp16_tag, 64><<<grid, 256, 0, s>>>(a);
    }
    return rsa_launch_status();
}

extern "C" int rsa_dense_masked_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                    const void* mask, int mask_kind, int64_t mask_stride_b, int64_t mask_stride_h,
                                    int64_t mask_stride_q, int64_t mask_stride_k, int empty_rows_nan, rsa_out4 out, void* stream) {
    if (!mask) return RSA_ERR_BAD_ARG;
    return dense_masked_launch(B, H, Sq, Sk, D, dtype, q, k, v, mask, mask_kind, mask_stride_b, mask_stride_h, mask_stride_q,
                               mask_stride_k, 0, empty_rows_nan, 0.0f, 0ull, out, stream);
}

extern "C" int rsa_dense_dropout_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                     const void* mask, int mask_kind, int64_t mask_stride_b, int64_t mask_stride_h,
                                     int64_t mask_stride_q, int64_t mask_stride_k, int causal, int empty_rows_nan, float drop_rate,
                                     uint64_t seed, rsa_out4 out, void* stream) {
    return dense_masked_launch(B, H, Sq, Sk, D, dtype, q, k, v, mask, mask_kind, mask_stride_b, mask_stride_h, mask_stride_q,
                               mask_stride_k, causal, empty_rows_nan, drop_rate, (unsigned long long)seed, out, stream);
}
