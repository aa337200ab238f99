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
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename Tag>
struct Elem;
template <>
struct Elem<bf16_tag> {
    static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    }
    // D = A.B + C with D and C in DIFFERENT registers (hipcc ties them and copies C first when given the builtin)
    static __device__ __forceinline__ f32x4 mfma_rowsum(s16x8 a, s16x8 b, f32x4 c) {   // 16 x 16 x 32 (row sums: rsa_attn_kernel.hip)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
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
    static __device__ __forceinline__ f32x4 mfma_rowsum(s16x8 a, s16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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
    int heavy_last;                           // 64-row kernel: the (split) text-row pieces are the LAST workgroups of the grid
    // 64-row kernel, tail split: sparse workgroups [0, tail_first) walk their whole list; the tail_n x tail_p workgroups behind them
    // are the pieces of the tail_n sparse blocks tail_first .. (piece i = block tail_first + i / tail_p, part i % tail_p of its
    // list), partials to tail_part; the text pieces follow.  tail_n = 0: no split
    int tail_first, tail_n, tail_p;
    float* tail_part;
    float qk_scale;
    unsigned* gsync;                          // 64-row kernel: start-alignment counters of this launch (rsa_attn_kernel64.hip), or null
    int gsync_gen;                            // ... workgroups an XCD holds at a time (a generation)
    int gsync_ratio;                          // ... walks that keep 1 / gsync_ratio of the keys or more are not held (default 2)
    int rows256;                              // 64-row kernel, dense calls: NQB / NBv count 256-row tiles (four waves per workgroup, one K/V ring)
    int k5_static;                            // 64-row kernel, bf16: the steady state keeps the softmax reference it is entered with (checked, redone if it overflowed)
#ifdef RSA_K5_DIAG
    unsigned long long* dbg;                  // diagnostics build only (make diag): per-wave s_memtime sums, see tools/diag_k5.py
#endif
};

// ---------------------------------------------------------------------------------------------------------------------
// Aligned starts of the sparse walks (all K5 kernels; host side: rsa_gsync_slot in rsa_attn.hip).
// Workgroup b runs on XCD b & 7 and is the (b >> 3)-th workgroup that XCD receives; the XCD holds `gen` of them at a time
// (its 32 CUs x the kernel's workgroups per CU, asked of the runtime by the launcher: 64 for the kernels at head dim 128),
// so "generation" g = (b >> 3) / gen can only be resident once generation g - 1 has left.  Every workgroup announces itself in the counter of (g, XCD) when it starts and, in front of its first
// staging instruction, waits until its whole generation has: the 64 walks of an XCD then start their ascending key lists
// TOGETHER and meet in the XCD's L2 (4 MiB = the K, V of ~60 key blocks) instead of each finding the other 63 at unrelated
// positions (HunyuanVideo R2, 10 % of the keys kept at random: L2 hit rate 15 % -> 41 %, 116 -> 79 GB over the fabric,
// which is what bounded that launch: profiles/r04_k5_gsync.md).  Advisory only -- the results do not depend on it: a
// bounded wait, switched off for the rest of the launch by the first workgroup that runs into the bound (word 0), so
// kernels of other processes sharing the device cannot stall this one.  Walks that keep half of the keys or more are not held
// back (round 4 drew that line at a fifth; measured at the reference scripts' operating points in round 5 --
// profiles/r05_k5_gsync_ratio.txt -- walks of 20 % of the keys gain 2.6 % from the wait, walks of 25 % on Wan2.1's 40 heads 8 %).
// ... nor are the walks of a head whose K, V mostly fit the L2 anyway (fewer than 256 key blocks = 16 MB: Wan2.2-TI2V's 214 blocks,
// 24.8 % kept, lost 2 % to the wait; measured with the guard above at 1/2, profiles/r05_k5_gsync_ratio.txt)
constexpr int RSA_GSYNC_MIN_BLOCKS = 256;
constexpr unsigned RSA_GSYNC_MAXG = 4096;    // generations with a counter (x 8 XCDs x 64 workgroups: 2 M workgroups)
constexpr int RSA_GSYNC_RING = 8;            // launches in flight with counters of their own
constexpr size_t RSA_GSYNC_SLOT_WORDS = 8 + 8 * (size_t)RSA_GSYNC_MAXG;
struct GsyncTicket { unsigned* cnt; unsigned expect; };

__device__ __forceinline__ GsyncTicket rsa_gsync_announce(unsigned* gsync, int gen) {
    GsyncTicket tk{nullptr, 0u};
    if (gsync) {
        const unsigned xcd = blockIdx.x & 7, n = blockIdx.x >> 3, g = n / (unsigned)gen;
        if (g < RSA_GSYNC_MAXG) {
            tk.cnt = gsync + 8 + g * 8 + xcd;
            const unsigned nx = (gridDim.x - xcd + 7) >> 3, left = nx - (unsigned)gen * g;
            tk.expect = left < (unsigned)gen ? left : (unsigned)gen;
            if (threadIdx.x == 0) __hip_atomic_fetch_add(tk.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return tk;
}

// (every thread of the workgroup calls it: ends in a workgroup barrier)
__device__ __forceinline__ void rsa_gsync_wait(unsigned* gsync, GsyncTicket tk, int n_items, int nb_total, int ratio = 2) {
    if (!tk.cnt) return;
    if (threadIdx.x == 0 && ratio * n_items < nb_total && nb_total >= RSA_GSYNC_MIN_BLOCKS &&
        __hip_atomic_load(gsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        const int bound = 32 + 3 * n_items;      // x (s_sleep 32 + one L2 round trip) ~ 2 us; a kept block takes ~3.4 us
        int it = 0;
        while (__hip_atomic_load(tk.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tk.expect) {
            __builtin_amdgcn_s_sleep(32);
            if (++it > bound) { __hip_atomic_store(gsync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}
// host: the counters of one launch (cleared in stream order in front of it) or null.  which: 1 = the 64-row kernel (on by
// default), 2 = the 32-row and the e4m3 kernels (off by default: with two waves per SIMD they gain nothing from it and lose the
// wait, profiles/r04_k5_gsync.md) -- bits of the tuning key "k5_gsync"; wg_per_cu = what the runtime says fits
// (hipOccupancyMaxActiveBlocksPerMultiprocessor); *gen = workgroups per XCD generation
unsigned* rsa_gsync_slot(int which, unsigned grid, int wg_per_cu, hipStream_t s, int* gen);
int rsa_k5_static();      // tuning key "k5_static" (default 1): rsa_attn_kernel64.hip, optimistic static reference
int rsa_gsync_ratio();    // tuning key "k5_gsync_ratio" (default 2: walks keeping half of the keys or more are not held)
int rsa_wg_per_cu(const void* kernel, int block, size_t lds_bytes);   // cached hipOccupancyMaxActiveBlocksPerMultiprocessor; 0 = unknown
// launch `kernel` with the launch's alignment counters filled into its argument struct (sparse lists only: dense walks share their keys anyway)
#define RSA_LAUNCH_GSYNC(which, kernel, args, MODE_IS_SPARSE, grid, block, lds_bytes, stream) \
    do { \
        auto kfn_ = kernel; \
        auto aa_ = args; \
        aa_.gsync = nullptr; aa_.gsync_gen = 64; aa_.gsync_ratio = rsa_gsync_ratio(); aa_.k5_static = rsa_k5_static(); \
        if (MODE_IS_SPARSE) \
            aa_.gsync = rsa_gsync_slot(which, (grid).x, rsa_wg_per_cu(reinterpret_cast<const void*>(kfn_), block, lds_bytes), stream, \
                                       &aa_.gsync_gen); \
        kfn_<<<grid, block, lds_bytes, stream>>>(aa_); \
    } while (0)

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

