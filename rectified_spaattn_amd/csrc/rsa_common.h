// Shared device/host helpers for librsa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsa.h"

#define RSA_NT 256  // threads per workgroup in the row kernels == partial sums of contract C6

struct bf16_tag {};
struct fp16_tag {};

// raw 2-byte element -> fp32 (exact)
template <typename Tag>
__device__ __forceinline__ float rsa_to_f32(unsigned short bits);
template <>
__device__ __forceinline__ float rsa_to_f32<bf16_tag>(unsigned short bits) {
    return __uint_as_float(((unsigned)bits) << 16);
}
template <>
__device__ __forceinline__ float rsa_to_f32<fp16_tag>(unsigned short bits) {
    return (float)__builtin_bit_cast(_Float16, bits);
}

// Contract C5: exp from fp32 mul / fma / rint / ldexp only (bit-identical to oracle/rsa_oracle.c::orc_exp).
__device__ __forceinline__ float rsa_exp(float x) {
    const float y = x * 1.44269504088896340736f;
    if (!(y >= -126.0f)) return 0.0f;
    const float n = __builtin_rintf(y);
    const float f = y - n;
    float p = 1.52527338040598402800e-5f;
    p = __builtin_fmaf(p, f, 1.54035303933816099544e-4f);
    p = __builtin_fmaf(p, f, 1.33335581464284434234e-3f);
    p = __builtin_fmaf(p, f, 9.61812910762847716197e-3f);
    p = __builtin_fmaf(p, f, 5.55041086648215799532e-2f);
    p = __builtin_fmaf(p, f, 2.40226506959100712333e-1f);
    p = __builtin_fmaf(p, f, 6.93147180559945309417e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

__device__ __forceinline__ float wave_xor_add(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

// Contract C6 tree over the 256 per-thread partials: strides 1..32 inside the wave, 64 and 128 through LDS.
// `red` is a 4-float LDS scratch.  All threads return the same value.
__device__ __forceinline__ float block_tree_sum(float part, float* red) {
    part = wave_xor_add(part, 1);
    part = wave_xor_add(part, 2);
    part = wave_xor_add(part, 4);
    part = wave_xor_add(part, 8);
    part = wave_xor_add(part, 16);
    part = wave_xor_add(part, 32);
    __syncthreads();  // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float block_max(float v, float* red) {
    for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

static inline int rsa_check_layout(const rsa_layout* l) {
    if (!l) return RSA_ERR_BAD_ARG;
    if (l->B <= 0 || l->H <= 0 || l->S <= 0) return RSA_ERR_BAD_ARG;
    if (l->D != 64 && l->D != 128) return RSA_ERR_UNSUPPORTED;
    if (l->dtype != RSA_BF16 && l->dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    if (l->NB_total != (l->S + RSA_BLOCK - 1) / RSA_BLOCK) return RSA_ERR_BAD_ARG;
    if (l->NBv < 0 || l->NBv > l->NB_total) return RSA_ERR_BAD_ARG;
    if (l->n_txt < 0 || l->kv_valid < 0 || l->kv_valid > l->S) return RSA_ERR_BAD_ARG;
    if (l->n_txt > 0 && (long)l->NBv * RSA_BLOCK + l->n_txt > l->S) return RSA_ERR_BAD_ARG;
    if (l->pool_valid < 0 || l->pool_valid > l->S) return RSA_ERR_BAD_ARG;
    if (l->text_end_block < 0 || l->text_end_block > l->NB_total) return RSA_ERR_BAD_ARG;
    if (l->first_frame_blocks < 0) return RSA_ERR_BAD_ARG;
    if (l->q_text_valid < 0 || (long)l->NBv * RSA_BLOCK + l->q_text_valid > (long)l->NB_total * RSA_BLOCK)
        return RSA_ERR_BAD_ARG;
    if (l->kv_text_valid < 0 || l->kv_text_valid > l->S) return RSA_ERR_BAD_ARG;
    return RSA_OK;
}

static inline int rsa_check_tensor(const rsa_tensor4& t) {
    if (!t.ptr) return RSA_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(t.ptr) & 15) != 0) return RSA_ERR_BAD_ARG;
    if ((t.stride_b % 8) || (t.stride_h % 8) || (t.stride_s % 8)) return RSA_ERR_BAD_ARG;  // 16-B row chunks
    return RSA_OK;
}

// K1; with f8 != nullptr (D = 128) it also writes the e4m3 images and block exponents of every block it pools -- q and k
// blocks < NBv, v blocks < NB_total (rsa_fp8_emit.h).  rsa_stats.hip.
struct Fp8Emit;
int rsa_pool_stats_f8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v, const rsa_buffers* buf,
                      const Fp8Emit* f8, void* stream);

// e4m3 images for the dense fp8 kernel, carved out of `ws` (rsa_fp8.hip)
int rsa_dense_quantize_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                           void* ws, size_t ws_bytes, rsa_fp8_operands* ops, hipStream_t s, int v_only = 0);

// last HIP error seen by a launch of this library (for rsa_last_hip_error); defined in rsa_stats.hip
extern int g_rsa_last_hip_error;
static inline int rsa_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return RSA_OK;
    g_rsa_last_hip_error = (int)e;
    return RSA_ERR_LAUNCH;
}
