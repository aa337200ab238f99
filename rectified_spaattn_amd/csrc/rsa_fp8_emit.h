// e4m3 images of one 128-row block, produced from the registers of a thread block that already holds it in the layout of
// K1 (pool_stats_kernel): 2 D threads, thread t owns the 8 head-dim elements 8c .. 8c+7 (c = t % (D/8)) of the rows
// 16 i + g (g = t / (D/8), i = 0..7).  Used by K1 itself (rsa_stats.hip: the pooling pass writes the images of the blocks
// it pools, so Q, K and V are read from HBM once) and by the stand-alone block kernel of rsa_fp8.hip (text-tail blocks,
// rsa_quantize_fp8, the dense path).  D = 128 or 64.
//
// Block-scaled format (bit-exact against oracle.fp8_block_images):
//      y = x * qk_const (Q: one fp32 multiply, qk_const = sm_scale * log2(e))  |  x - mu[d] (K, "smooth K")  |  x (V)
//      A = max |y| over the block's valid rows;  scale 2^e, e = the smallest integer with A * 2^-e <= 448 (clamped to
//      +-120; 0 for an all-zero block), stored as the E8M0 byte 127 + e in byte `which` of exps[bh][blk];
//      bytes = v_cvt_pk_fp8_f32(y * 2^-e) (round to nearest even, subnormals kept); rows >= valid are zero.
//      q8, k8 : [BH, S_pad, D] row-major;  v8t : [BH, S_pad / 64, D, 64] = V^T per 64-key tile, byte p = 32 h + j of a
//      (tile, d) row = key 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h (the f8f6f4 MFMA's k-slot order of the P operand).
// A power-of-two scale costs e4m3 (a floating-point format) nothing, needs no division and no second pass over the
// tensor, and the fp8 MFMA applies it for free through its E8M0 block-scale operands (rsa_attn_fp8_kernel.hip).
#pragma once
#include "rsa_common.h"

struct Fp8Emit {
    uint8_t *q8, *k8, *v8t;
    uint32_t* exps;        // [BH, NB_total]: byte 0 Q block, byte 1 K block, byte 2 V block (E8M0)
    const float* kmean;    // [BH, D] or nullptr
    float qk_const;
    int S_pad;             // rows of every image (multiple of 128)
    int NB_total;          // words per head in exps
    int valid[3];          // rows >= valid[which] are zero in the image
};

constexpr int rsa_f8_lrow(int D) { return D + 16; }                              // padded LDS row of the V transpose (bytes)
constexpr int rsa_f8_lds(int D) { return RSA_BLOCK * rsa_f8_lrow(D) + 16; }      // bytes of LDS fp8_emit_block needs (tile + maxima)

// E8M0 byte of the block scale from the block maximum (exponent field and one mantissa compare; no division)
__device__ __forceinline__ int rsa_e8m0_of_amax(float amax) {
    if (!(amax > 0.0f)) return 127;
    const unsigned bits = __float_as_uint(amax);
    const int field = (int)((bits >> 23) & 0xFF);
    if (field == 0) return 127 - 120;
    int e = (field - 126) - 9 + ((bits & 0x7FFFFFu) > 0x600000u ? 1 : 0);   // amax = m 2^x, m in [0.5, 1): m > 0.875 -> one more
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return 127 + e;
}

// x[i][e]: the block as K1 holds it (rows >= valid already zero).  All 2 D threads of the block must call it (barriers).
// lds: rsa_f8_lds(D) bytes, 16-byte aligned.
template <int D, typename Tag>
__device__ __forceinline__ void fp8_emit_block(const float (&x)[8][8], const Fp8Emit& f, int which, int blk, int bh,
                                               unsigned char* lds) {
    constexpr int CH = D / 8, NW = (2 * D) / 64, RSA_F8_LROW = rsa_f8_lrow(D);
    if (which < 2 && (which == 0 ? f.q8 : f.k8) == nullptr) return;      // V-only producer (the pv form of K5): uniform per block
    const int t = threadIdx.x, c = t % CH, g = t / CH;
    const int valid = f.valid[which];
    float mu[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (which == 1 && f.kmean != nullptr) {
        const float4* mp = reinterpret_cast<const float4*>(f.kmean + (long)bh * D + 8 * c);
        const float4 m0 = mp[0], m1 = mp[1];
        mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w;
        mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
    }
    // Rows >= valid arrive as zeros: for Q and V they then add nothing to the maximum and quantise to zero bytes by
    // themselves; for K (y = 0 - mu) they must be skipped, which only a block that straddles `valid` has to test.
    const bool ragged = which == 1 && (blk + 1) * RSA_BLOCK > valid;
    // ---- block maximum.  Q: max |x c| = fl(max |x| c) (rounding is monotone); K: max |x - mu|; V: max |x|
    float m = 0.0f;
    if (which == 1) {   // (uniform branches: `which` is the block's tensor)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ragged && blk * RSA_BLOCK + 16 * i + g >= valid) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[i][e] - mu[e]));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[i][e]));
        if (which == 0) m = m * f.qk_const;
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
    float* amx = reinterpret_cast<float*>(lds + RSA_BLOCK * RSA_F8_LROW);
    __syncthreads();   // (the caller may have used this LDS before)
    if ((t & 63) == 0) amx[t >> 6] = m;
    __syncthreads();
    m = fmaxf(amx[0], amx[1]);
    if constexpr (NW == 4) m = fmaxf(m, fmaxf(amx[2], amx[3]));
    const int eb = rsa_e8m0_of_amax(m);
    const float inv = __uint_as_float((unsigned)(254 - eb) << 23);   // 2^-(eb - 127), exact
    if (t == 0) reinterpret_cast<uint8_t*>(f.exps + (long)bh * f.NB_total + blk)[which] = (uint8_t)eb;
    // ---- bytes.  A power-of-two factor commutes with fp32 rounding, so fl(x c) 2^-e = fl(x (c 2^-e)) and
    // fl(x - mu) 2^-e = fma(x, 2^-e, -mu 2^-e): one operation per element; |y 2^-e| <= 448 by construction (no clamp)
    const float qs = which == 0 ? f.qk_const * inv : inv;
    float nmu[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) nmu[e] = -(mu[e] * inv);
    // each row's 8 bytes go out as soon as they exist: Q, K row-major to global (16 threads = one 128-byte row), V into the
    // LDS tile the transpose below reads
    uint8_t* grow = (which == 0 ? f.q8 : f.k8) + ((long)bh * f.S_pad + (long)blk * RSA_BLOCK) * D + 8 * c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float y[8];
        if (which == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = __builtin_fmaf(x[i][e], inv, nmu[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = x[i][e] * qs;
        }
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], hi, true);
        uint2 o = make_uint2((unsigned)lo, (unsigned)hi);
        if (ragged && blk * RSA_BLOCK + 16 * i + g >= valid) o = make_uint2(0u, 0u);
        if (which == 2) *reinterpret_cast<uint2*>(lds + (16 * i + g) * RSA_F8_LROW + 8 * c) = o;
        else *reinterpret_cast<uint2*>(grow + (long)(16 * i + g) * D) = o;
    }
    if (which < 2) return;
    // V: out of the LDS tile into the two transposed 64-key tiles
    __syncthreads();
    uint8_t* dst = f.v8t + ((long)bh * (f.S_pad / 64) + 2 * blk) * (long)(D * 64);
    // one item per thread: (tile, half jh of the 32 slot bytes of lane half hh, 4 head-dim columns dg): 16 four-byte LDS
    // reads [key][4 dg .. 4 dg + 3] (lanes differ in dg first: conflict-free), regrouped by byte into the 16-byte pieces
    // of the four d rows.  Slot j = 16 jh + 4 q + e of lane half hh is key 32 jh + e + 8 q + 4 hh of the tile.
    {
        const int dg = t % (D / 4), hh = (t / (D / 4)) & 1, jh = (t / (D / 2)) & 1, tile = t / D;
        unsigned W[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 64 * tile + 32 * jh + e + 8 * q + 4 * hh;
                W[q][e] = *reinterpret_cast<const unsigned*>(lds + key * RSA_F8_LROW + 4 * dg);
            }
#pragma unroll
        for (int bsel = 0; bsel < 4; ++bsel) {   // byte bsel of every word = head-dim column 4 dg + bsel
            const unsigned sel01 = (unsigned)bsel | ((unsigned)(4 + bsel) << 8);
            unsigned w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned p01 = __builtin_amdgcn_perm(W[q][1], W[q][0], sel01);
                const unsigned p23 = __builtin_amdgcn_perm(W[q][3], W[q][2], sel01);
                w[q] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
            }
            *reinterpret_cast<uint4*>(dst + (long)tile * (D * 64) + (4 * dg + bsel) * 64 + 32 * hh + 16 * jh) =
                make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}
