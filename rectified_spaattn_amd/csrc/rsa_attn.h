// Shared declarations of the K5 attention kernels (rsa_attn_kernel.hip: bf16 / fp16; rsa_attn_fp8_kernel.hip: e4m3)
// and their host sides (rsa_attn.hip).
#pragma once
#include <string.h>

#include <type_traits>

#include "rsa_common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename Tag>
struct Elem;
template <>
struct Elem<bf16_tag> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    }
    // D = A.B + C with D and C in DIFFERENT registers (hipcc ties them and copies C first when given the builtin)
    static __device__ __forceinline__ f32x16 mfma_from(s16x8 a, s16x8 b, const f32x16& c) {
        f32x16 d;
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ unsigned short from_f32(float f) {
        return __builtin_bit_cast(unsigned short, (__bf16)f);
    }
    static __device__ __forceinline__ s16x8 cvt8(const float* f) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (__bf16)f[i];
        return __builtin_bit_cast(s16x8, r);
    }
};
template <>
struct Elem<fp16_tag> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                      0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma_from(s16x8 a, s16x8 b, const f32x16& c) {
        f32x16 d;
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ unsigned short from_f32(float f) {
        return __builtin_bit_cast(unsigned short, (_Float16)f);
    }
    static __device__ __forceinline__ s16x8 cvt8(const float* f) {
        f16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (_Float16)f[i];
        return __builtin_bit_cast(s16x8, r);
    }
};

enum { MODE_SPARSE = 0, MODE_DENSE = 1 };

struct AttnArgs {
    const unsigned short *q, *k, *v;
    long qsb, qsh, qss, ksb, ksh, kss, vsb, vsh, vss;
    unsigned short* out;
    long osb, osh, oss;
    const int32_t* cols;    // [BH, NBv, NB_total]
    const int32_t* counts;  // [BH, NBv]
    const float* R;         // [BH, NBv] or null
    const float* comp;      // [BH, NBv, D] or null
    int mode, H, Sq, Sk;
    int NBv, NQB, NB_total;  // sparse: q blocks < NBv use lists; NQB = total q blocks
    int kv_valid, kv_text_valid, q_text_end;  // sparse mode (q_text_end = NBv*128 + q_text_valid)
    int q_split, kv_split;                    // dense mode
    int causal;                               // dense mode: key j of a segment visible to its row i iff j <= i + (keys - rows)
    int n_heavy_pad, NBp, BH;                 // work mapping
    float* tpart;                             // split-KV partials of the text query blocks, or null
    int tsplit, tper;                         // workgroups per text block, key blocks per workgroup
    float qk_scale;
#ifdef RSA_K5_DIAG
    unsigned long long* dbg;                  // diagnostics build only (make diag): per-wave s_memtime sums, see tools/diag_k5.py
#endif
};

// byte offset of 16-byte chunk `ch` of row `row` inside a [64][D] 2-byte tile.  The XOR keeps both the
// ds_read_b128 row reads (K as MFMA A operand) and the ds_read_b64_tr_b16 transposing reads (V^T as A
// operand) bank-conflict free for D = 128 (256-byte rows).
template <int D>
__device__ __forceinline__ int tile_off(int row, int ch) {
    if constexpr (D == 128) {
        return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    } else {
        return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
    }
}

