// Mask-selection pass of the rectified block-sparse attention path: K1 pool_stats, K2 pooled_scores,
// K3 select_mask, K4 compensation.  gfx950 only.  Compiled with -ffp-contract=off: every fused multiply-add
// is an explicit __builtin_fmaf so the arithmetic equals the fp32 statistics contract (oracle/rsa_oracle.c)
// bit for bit.  All of this is HBM/LDS-bound integer-and-fp32 work; no MFMA on purpose.
#include "rsa_common.h"
#include "rsa_fp8_emit.h"

// =====================================================================================================
// K1: pool_stats -- block means (and mean |x - mean|) of Q, K, V in one launch.
//   grid (NB_total, BH, 3); D*2 threads: thread t owns 8 consecutive head-dim elements (16 B) of the rows
//   16*i + g (g = t / (D/8), i = 0..7): a wave reads 1 KiB contiguous per load instruction.
//   Reduction order = contract C2/C3.
// =====================================================================================================
struct PoolArgs {
    const unsigned short* src[3];
    long sb[3], sh[3], ss[3];
    float* mean[3];
    float* mad[3];   // mad[2] == nullptr
    const float* ext_mean[3];  // if set, deviations are taken from these means (estimate_pr_gain's q_pools)
    int nblk[3];     // blocks to produce per tensor
    int valid[3];    // rows >= valid are zero
    int H;
    int BH, NB_total;
    Fp8Emit f8;      // F8 instances only: the e4m3 images of every block this launch pools (rsa_fp8_emit.h)
};

// F8: the fp8 operand path's form -- the block is in registers anyway, so its e4m3 image is written in the same pass
template <int D, typename Tag, bool F8>
__global__ __launch_bounds__(D * 2) void pool_stats_kernel(PoolArgs a) {
    constexpr int CH = D / 8;        // 16-byte chunks per row
    constexpr int NTH = 16 * CH;     // threads
    constexpr int NW = NTH / 64;     // waves
    // F8: the three tensors' blocks interleaved in launch order (V blocks also transpose through LDS: mixed with the
    // streaming Q / K blocks on a CU they overlap instead of running as one heavy tail)
    const int which = F8 ? (int)(blockIdx.x % 3) : (int)blockIdx.z;
    const int blk = F8 ? (int)(blockIdx.x / 3) : (int)blockIdx.x;
    if (blk >= a.nblk[which]) return;
    const int bh = blockIdx.y;
    const int b = bh / a.H, h = bh % a.H;
    const int t = threadIdx.x;
    const int c = t % CH, g = t / CH;
    const unsigned short* base = a.src[which] + (long)b * a.sb[which] + (long)h * a.sh[which] + c * 8;
    const long ss = a.ss[which];
    const int valid = a.valid[which];

    float x[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = blk * RSA_BLOCK + 16 * i + g;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row < valid) raw = *reinterpret_cast<const uint4*>(base + (long)row * ss);
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[i][2 * e] = rsa_to_f32<Tag>((unsigned short)(w[e] & 0xFFFF));
            x[i][2 * e + 1] = rsa_to_f32<Tag>((unsigned short)(w[e] >> 16));
        }
    }
    // F8: the e4m3 image of the block goes out LAST (after the statistics): its V transpose then runs with x dead
    auto emit_f8 = [&]() {
        if constexpr (F8) {
            __shared__ __attribute__((aligned(16))) unsigned char f8lds[rsa_f8_lds(D)];
            fp8_emit_block<D, Tag>(x, a.f8, which, blk, bh, f8lds);
        }
    };
    __shared__ float red[NW][D];
    float mean[8];
    // ---- sum -> mean
    {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] = x[0][e];
#pragma unroll
            for (int i = 1; i < 8; ++i) s[e] = s[e] + x[i][e];
        }
        // tree over g: in-wave lanes differ in g by multiples of CH
#pragma unroll
        for (int m = CH; m < 64; m <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = s[e] + __shfl_xor(s[e], m, 64);
        if ((t & 63) < CH)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[t >> 6][c * 8 + e] = s[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float tot;
            if (NW == 4)
                tot = (red[0][c * 8 + e] + red[1][c * 8 + e]) + (red[2 % NW][c * 8 + e] + red[3 % NW][c * 8 + e]);
            else
                tot = red[0][c * 8 + e] + red[1][c * 8 + e];
            mean[e] = tot * (1.0f / RSA_BLOCK);
        }
    }
    const long orow = ((long)bh * a.nblk[which] + blk) * D + c * 8;
    if (g == 0) {
        float4* o = reinterpret_cast<float4*>(a.mean[which] + orow);
        o[0] = make_float4(mean[0], mean[1], mean[2], mean[3]);
        o[1] = make_float4(mean[4], mean[5], mean[6], mean[7]);
    }
    if (a.mad[which] == nullptr) { emit_f8(); return; }
    if (a.ext_mean[which] != nullptr) {
        const float4* em = reinterpret_cast<const float4*>(a.ext_mean[which] + orow);
        const float4 m0 = em[0], m1 = em[1];
        mean[0] = m0.x; mean[1] = m0.y; mean[2] = m0.z; mean[3] = m0.w;
        mean[4] = m1.x; mean[5] = m1.y; mean[6] = m1.z; mean[7] = m1.w;
    }
    // ---- mean absolute deviation
    {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] = fabsf(x[0][e] - mean[e]);
#pragma unroll
            for (int i = 1; i < 8; ++i) s[e] = s[e] + fabsf(x[i][e] - mean[e]);
        }
#pragma unroll
        for (int m = CH; m < 64; m <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = s[e] + __shfl_xor(s[e], m, 64);
        __syncthreads();
        if ((t & 63) < CH)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[t >> 6][c * 8 + e] = s[e];
        __syncthreads();
        if (g == 0) {
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float tot;
                if (NW == 4)
                    tot = (red[0][c * 8 + e] + red[1][c * 8 + e]) +
                          (red[2 % NW][c * 8 + e] + red[3 % NW][c * 8 + e]);
                else
                    tot = red[0][c * 8 + e] + red[1][c * 8 + e];
                r[e] = tot * (1.0f / RSA_BLOCK);
            }
            float4* o = reinterpret_cast<float4*>(a.mad[which] + orow);
            o[0] = make_float4(r[0], r[1], r[2], r[3]);
            o[1] = make_float4(r[4], r[5], r[6], r[7]);
        }
    }
    emit_f8();
}

// =====================================================================================================
// K2: pooled scores on the fp32 matrix pipe.  v_mfma_f32_32x32x2_f32 accumulates its two k-steps as sequential fused
// multiply-adds, k ascending (tools/probes/mfma_f32_order_probe.hip: d == fmaf(a1,b1,fmaf(a0,b0,c)) on every output),
// so a chain of D/2 MFMAs over k = 0,1 | 2,3 | ... IS the contract's k-ordered fmaf chain (C4), bit for bit, at the
// matrix pipe's rate and with one operand register per 4 096 FLOP instead of an LDS read per 32.
// Workgroup = 4 waves = 64(i) x 64(j) outputs, wave = one 32x32 MFMA tile with three accumulators (s, eq, ek).
// Operands go global -> registers (next chunk in flight during the MFMAs) -> LDS as [row][16 even k | 16 odd k | pad 4]:
// lane (r = lane & 31, h = lane >> 5) feeds k = 2j + h, so it reads 4 consecutive steps with one ds_read_b128;
// LD = 36 floats makes those reads conflict-free.  Waves whose 32 rows or 32 columns lie outside the matrix skip the MFMAs.
//   MODE 0: visual columns: s = qbar.kbar, eq = aq.kbar, ek = qbar.ak -> scores + GAPR byte
//   MODE 1: text columns:   s = qbar.K_u for every valid text token u
// =====================================================================================================
struct ScoreArgs {
    const float *qbar, *aq, *kbar, *ak;
    const unsigned short* ktxt;  // K base pointer (MODE 1)
    long ksb, ksh, kss;
    float* scores;
    uint8_t* unrel;
    int NBv, n_txt, NS, D, H, BH;
};

typedef float k2_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int ncols_of(const ScoreArgs& a, int mode) { return mode == 0 ? a.NBv : a.n_txt; }

template <int MODE, typename Tag>
__device__ __forceinline__ void pooled_scores_tile(const ScoreArgs& a, int bh, int i0, int j0, float (*tile)[64 * 36]) {
    constexpr int DK = 32, LD = 36, NOP = MODE == 0 ? 4 : 2;   // operands: q, k (, aq, ak)
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const int r = lane & 31, h = lane >> 5;
    const int ncols = MODE == 0 ? a.NBv : a.n_txt;
    const int D = a.D;
    const float* qb = a.qbar + (long)bh * a.NBv * D;
    const float* aqp = a.aq + (long)bh * a.NBv * D;
    const float* kb = a.kbar + (long)bh * a.NBv * D;
    const float* akp = a.ak + (long)bh * a.NBv * D;
    const unsigned short* kt = nullptr;
    if (MODE == 1) {
        const int b = bh / a.H, hd = bh % a.H;
        kt = a.ktxt + (long)b * a.ksb + (long)hd * a.ksh + (long)a.NBv * RSA_BLOCK * a.kss;
    }
    const bool live = i0 + 32 * wi < a.NBv && j0 + 32 * wj < ncols;   // wave-uniform

    // staging slots of this thread: rows srow, srow + 32; k = 4 * sc .. 4 * sc + 3 of the chunk
    const int srow = t >> 3, sc = t & 7;
    float4 st[NOP][2];
    // rows / columns past the matrix edge only feed outputs that are never stored: their loads are clamped to the last
    // valid row instead of zero-filled, so the fetch has no branches and stays in flight during the MFMAs
    const float *gq[2], *gaq[2], *gk[2], *gak[2];
    const unsigned short* gkt[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int ri = min(i0 + srow + 32 * rr, a.NBv - 1), rj = min(j0 + srow + 32 * rr, ncols - 1);
        gq[rr] = qb + (long)ri * D + 4 * sc;
        gaq[rr] = aqp + (long)ri * D + 4 * sc;
        gk[rr] = kb + (long)rj * D + 4 * sc;
        gak[rr] = akp + (long)rj * D + 4 * sc;
        gkt[rr] = MODE == 1 ? kt + (long)rj * a.kss + 4 * sc : nullptr;
    }
    auto fetch = [&](int d0) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            st[0][rr] = *reinterpret_cast<const float4*>(gq[rr] + d0);
            if (MODE == 0) {
                st[1][rr] = *reinterpret_cast<const float4*>(gk[rr] + d0);
                st[2][rr] = *reinterpret_cast<const float4*>(gaq[rr] + d0);
                st[3][rr] = *reinterpret_cast<const float4*>(gak[rr] + d0);
            } else {
                const uint2 raw = *reinterpret_cast<const uint2*>(gkt[rr] + d0);
                st[1][rr].x = rsa_to_f32<Tag>((unsigned short)(raw.x & 0xFFFF));
                st[1][rr].y = rsa_to_f32<Tag>((unsigned short)(raw.x >> 16));
                st[1][rr].z = rsa_to_f32<Tag>((unsigned short)(raw.y & 0xFFFF));
                st[1][rr].w = rsa_to_f32<Tag>((unsigned short)(raw.y >> 16));
            }
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int op = 0; op < NOP; ++op)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                float* base = &tile[op][(srow + 32 * rr) * LD];
                *reinterpret_cast<float2*>(base + 2 * sc) = make_float2(st[op][rr].x, st[op][rr].z);        // even k
                *reinterpret_cast<float2*>(base + 16 + 2 * sc) = make_float2(st[op][rr].y, st[op][rr].w);   // odd k
            }
    };

    k2_f32x16 s, eq, ek;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.0f; eq[i] = 0.0f; ek[i] = 0.0f; }

    fetch(0);
    if (!live) {   // this wave's 32 x 32 outputs lie outside the matrix: it only helps staging (same barriers)
        for (int d0 = 0; d0 < D; d0 += DK) {
            __syncthreads();
            put();
            __syncthreads();
            if (d0 + DK < D) fetch(d0 + DK);
        }
        return;
    }
    const float* pa = &tile[0][(32 * wi + r) * LD + 16 * h];
    const float* pb = &tile[1][(32 * wj + r) * LD + 16 * h];
    const float* paa = &tile[2 % NOP][(32 * wi + r) * LD + 16 * h];
    const float* pba = &tile[3 % NOP][(32 * wj + r) * LD + 16 * h];
    for (int d0 = 0; d0 < D; d0 += DK) {
        __syncthreads();          // previous chunk's reads are done
        put();
        __syncthreads();
        if (d0 + DK < D) fetch(d0 + DK);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 q4 = *reinterpret_cast<const float4*>(pa + 4 * m);
            const float4 k4 = *reinterpret_cast<const float4*>(pb + 4 * m);
            const float qa[4] = {q4.x, q4.y, q4.z, q4.w};
            const float ka[4] = {k4.x, k4.y, k4.z, k4.w};
            float aa[4] = {0, 0, 0, 0}, ba[4] = {0, 0, 0, 0};
            if (MODE == 0) {
                const float4 a4 = *reinterpret_cast<const float4*>(paa + 4 * m);
                const float4 b4 = *reinterpret_cast<const float4*>(pba + 4 * m);
                aa[0] = a4.x; aa[1] = a4.y; aa[2] = a4.z; aa[3] = a4.w;
                ba[0] = b4.x; ba[1] = b4.y; ba[2] = b4.z; ba[3] = b4.w;
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {   // step j = 4m + x of the chunk: k = d0 + 2j (h = 0), d0 + 2j + 1 (h = 1)
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[x], ka[x], s, 0, 0, 0);
                if (MODE == 0) {
                    eq = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[x], ka[x], eq, 0, 0, 0);
                    ek = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[x], ba[x], ek, 0, 0, 0);
                }
            }
        }
    }
    // accumulator element e of lane (r, h): row (e & 3) + 8 (e >> 2) + 4 h, column r of the wave tile
    const int colbase = MODE == 0 ? 0 : a.NBv;
    const int j = j0 + 32 * wj + r;
    if (j >= ncols) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + 32 * wi + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= a.NBv) continue;
        a.scores[((long)bh * a.NBv + i) * a.NS + colbase + j] = s[e];
        if (MODE == 0)
            a.unrel[((long)bh * a.NBv + i) * a.NBv + j] = !(fabsf(s[e]) > (fabsf(eq[e]) + fabsf(ek[e])));
    }
}

// K2, visual columns, round 6: the same three fp32 chains fed by LDS-DMA into a DOUBLE-BUFFERED tile, one barrier per 32-k chunk, no
// register staging (the lever profiles/r05_k2_experiments.md names).  The chunk of an operand is [64 rows][32 k] fp32 = 8 KiB as
// eight 1-KiB pieces (8 rows x 128 B); wave w stages operand w (q, k, aq, ak) with global_load_lds_dwordx4 -- the LDS image of a
// piece is lane-linear, so the bank swizzle is applied to the SOURCE: position c of row `row` holds the row's 16-byte chunk
// c ^ ((row >> 1) & 7).  Lane (r, h) of the MFMA phase reads the four k of TWO chain steps with one ds_read_b128 (conflict-free: the
// 16 lanes of a b128 group cover 16 distinct (row & 1, chunk position) pairs) and picks k = 2 step + h by its lane half -- the
// chain is the contract's, operand for operand, so scores and GAPR bytes are bit-identical to the register-staged form (tests:
// test_gpu_select_paths.py::test_k2_dma_form_equals_the_staged_form).  Per chunk: wait for the own pieces of chunk c, ONE barrier
// (chunk c visible to everyone, everyone past chunk c - 1), issue chunk c + 1 into the other buffer, 16 x 3 MFMAs.
template <typename Tag, int DK>
__device__ __forceinline__ void pooled_scores_tile_dma(const ScoreArgs& a, int bh, int i0, int j0, float* smem) {
    // DK = k per chunk: 32 (rows of 128 B, two 32-KiB buffers: two workgroups per CU) or 16 (rows of 64 B, two 16-KiB buffers: the
    // register-staged form's three workgroups per CU and more)
    constexpr int ROWB = DK * 4, OPB = 64 * ROWB, BUFB = 4 * OPB, NPC = OPB / 1024, RPP = 1024 / ROWB, LPR = ROWB / 16;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 1, wj = wv & 1;
    const int r = lane & 31, h = lane >> 5;
    const int D = a.D, NBv = a.NBv;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    auto swz = [](int row) -> int { return DK == 32 ? (row >> 1) & 7 : (row >> 2) & 3; };
    // ---- staging: wave w moves operand w; piece p = rows RPP p .. of the tile, lane (rr = lane / LPR, c = lane % LPR) ----
    const float* opbase = (wv == 0 ? a.qbar : wv == 1 ? a.kbar : wv == 2 ? a.aq : a.ak) + (long)bh * NBv * D;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)opbase);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)opbase >> 32));
    const unsigned char* base = reinterpret_cast<const unsigned char*>(((unsigned long)hi << 32) | lo);
    const int row0 = (wv & 1) ? j0 : i0;                          // q / aq rows follow i, k / ak rows follow j
    const int rr = lane / LPR, c = lane % LPR;
    unsigned voff[NPC];
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
        const int row = RPP * p + rr;
        const int grow = min(row0 + row, NBv - 1);                // rows past the matrix edge feed outputs that are never stored
        voff[p] = (unsigned)(((long)grow * D + 4 * (c ^ swz(row))) * 4);
    }
    auto stage = [&](int d0, int buf) {
        const unsigned dst = lds0 + buf * BUFB + wv * OPB;
#pragma unroll
        for (int p = 0; p < NPC; ++p)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff[p] + (unsigned)(d0 * 4)), "s"(base), "s"(dst + p * 1024) : "memory");
    };
    const bool live = i0 + 32 * wi < NBv && j0 + 32 * wj < NBv;   // wave-uniform: a dead wave only stages
    k2_f32x16 s, eq, ek;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.0f; eq[i] = 0.0f; ek[i] = 0.0f; }
    // per-lane read addresses (bytes inside a buffer): operand o, row, chunk position m ^ g -- g is the same for every operand
    const int g = swz(r);
    const unsigned char* sm = reinterpret_cast<const unsigned char*>(smem);
    const int offq = 0 * OPB + (32 * wi + r) * ROWB, offk = 1 * OPB + (32 * wj + r) * ROWB;
    const int offa = 2 * OPB + (32 * wi + r) * ROWB, offb = 3 * OPB + (32 * wj + r) * ROWB;
    stage(0, 0);
    int buf = 0;
    for (int d0 = 0; d0 < D; d0 += DK, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the chunk have landed
        __syncthreads();                                          // ... everyone's have, and everyone has left the other buffer
        if (d0 + DK < D) stage(d0 + DK, buf ^ 1);
        if (!live) continue;
        const unsigned char* bb = sm + buf * BUFB;
#pragma unroll
        for (int m = 0; m < DK / 4; ++m) {
            const int pos = (m ^ g) << 4;
            const float4 q4 = *reinterpret_cast<const float4*>(bb + offq + pos);
            const float4 k4 = *reinterpret_cast<const float4*>(bb + offk + pos);
            const float4 a4 = *reinterpret_cast<const float4*>(bb + offa + pos);
            const float4 b4 = *reinterpret_cast<const float4*>(bb + offb + pos);
            // step 2m: k = d0 + 4m + h; step 2m + 1: k = d0 + 4m + 2 + h
            const float q0 = h ? q4.y : q4.x, q1 = h ? q4.w : q4.z, k0 = h ? k4.y : k4.x, k1 = h ? k4.w : k4.z;
            const float a0 = h ? a4.y : a4.x, a1 = h ? a4.w : a4.z, b0 = h ? b4.y : b4.x, b1 = h ? b4.w : b4.z;
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, k0, s, 0, 0, 0);
            eq = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, k0, eq, 0, 0, 0);
            ek = __builtin_amdgcn_mfma_f32_32x32x2f32(q0, b0, ek, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, k1, s, 0, 0, 0);
            eq = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, k1, eq, 0, 0, 0);
            ek = __builtin_amdgcn_mfma_f32_32x32x2f32(q1, b1, ek, 0, 0, 0);
        }
    }
    if (!live) return;
    const int j = j0 + 32 * wj + r;
    if (j >= NBv) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + 32 * wi + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= NBv) continue;
        a.scores[((long)bh * NBv + i) * a.NS + j] = s[e];
        a.unrel[((long)bh * NBv + i) * NBv + j] = !(fabsf(s[e]) > (fabsf(eq[e]) + fabsf(ek[e])));
    }
}
// MEASURED SLOWER than the register-staged form on every shape (profiles/r06_k2_dma.txt, interleaved: HunyuanVideo 24 heads 197.7 us
// with 16-k chunks / 210.0 with 32-k chunks against 187.9; Wan2.1 40 heads 136.2 / 141.9 / 129.7; 3 heads and Wan2.2 equal within
// 1 us): the staging pipeline is NOT what bounds K2 -- it sits on the fp32 matrix pipe at the clock the chip holds (16.3 GFLOP in
// 188 us = 87 TFLOP/s of the 98 a 1.5 GHz clock gives).  Kept as an A/B form (bit-identical by test), off by default.
int g_rsa_k2_dma = 0;   // tuning key "k2_dma": k per LDS-DMA chunk (32 / 16); 0 = the register-staged form (the product)

// One launch for both column kinds.  1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs, so XCD c takes the
// c-th contiguous eighth of the (bh, i-tile, j-tile) space -- whole heads per XCD, whose pooled operands (1.8 MB per head)
// then stay in that L2.  Within a head's row of tiles the visual column tiles come first, then the text-token tiles.
template <typename Tag, int DMA>
__global__ __launch_bounds__(256) void pooled_scores_kernel(ScoreArgs a) {
    // DMA form: two buffers of four [64][32] operand chunks (64 KiB: two workgroups per CU); the register-staged form (text-token
    // tiles always, visual tiles when DMA is off) uses the first 36 KiB as its padded [4][64 x 36] tile
    __shared__ __attribute__((aligned(1024))) float smem[DMA == 32 ? 2 * 4 * 64 * 32 : 4 * 64 * 36];      // (DMA 16: 32 KiB of it)
    float (*tile)[64 * 36] = reinterpret_cast<float (*)[64 * 36]>(smem);
    const int nti = (a.NBv + 63) / 64, ntj0 = (a.NBv + 63) / 64, ntj1 = (a.n_txt + 63) / 64, ntj = ntj0 + ntj1;
    const int per = (int)(gridDim.x >> 3);          // the grid is a multiple of 8
    const int wid = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wid >= a.BH * nti * ntj) return;
    const int bh = wid / (nti * ntj);
    const int rem = wid % (nti * ntj);
    const int i0 = (rem / ntj) * 64, tj = rem % ntj;
    if (tj < ntj0) {
        if constexpr (DMA != 0) pooled_scores_tile_dma<Tag, DMA>(a, bh, i0, tj * 64, smem);
        else pooled_scores_tile<0, Tag>(a, bh, i0, tj * 64, tile);
    } else pooled_scores_tile<1, Tag>(a, bh, i0, (tj - ntj0) * 64, tile);
}

// =====================================================================================================
// K3: select_mask -- ONE WAVE per (bh, q-block) row (four rows per 256-thread workgroup, no workgroup barrier
// anywhere).  Contract C5..C8.  The row's keys (probability bits << 32 | ~index) are sorted in REGISTERS:
// lane l holds elements l*KPL .. l*KPL+KPL-1 of the bitonic network (N2 = 64*KPL), partner distances below KPL
// are in-lane compare-swaps, larger ones exchange through lane shuffles.
// The contract's 256 strided partial sums are kept as four accumulators per lane (partial t = lane + 64 w), reduced
// per w across the lanes with the xor tree (strides 1..32) and combined as (u0+u1)+(u2+u3): bit-identical to the
// 256-thread formulation of the oracle.
// dynamic LDS per wave: float et[max(n_txt, 512)] (text-column exponentials, later the sorted-head candidates) | float pr[L]
// | u8 kept[NB_total] -- 6.6 KB at the HunyuanVideo shape, six workgroups per CU; the visual columns stay in registers
// =====================================================================================================
struct SelectArgs {
    const float* scores;
    const uint8_t* unrel;
    const uint8_t* neighbor;
    float *probs, *w, *R;
    uint32_t* bitmask;
    int32_t *cols, *counts;
    int NBv, n_txt, NS, L, N2, NB_total, NW, text_end_block, ffb, top_k, rows_total, lds_per_wave;
    int use_prefix;   // 1 = try the sorted-head path first (same result; tuning key "k3_prefix" for the A/B tests)
    int et_len;       // floats in the et[] region: >= n_txt and >= 2 * RSA_SEL_CAP (the sorted-head candidates reuse it)
    float thr, scale;
};
int g_rsa_k3_prefix = 1;
int g_rsa_k3_long = 0;    // 1 = the workgroup-per-row kernel for every row length (tuning key "k3_long": the tests compare the two)

__device__ __forceinline__ float wave_tree4(const float (&part)[4]) {
    float u[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float v = part[w];
        v = wave_xor_add(v, 1); v = wave_xor_add(v, 2); v = wave_xor_add(v, 4);
        v = wave_xor_add(v, 8); v = wave_xor_add(v, 16); v = wave_xor_add(v, 32);
        u[w] = v;
    }
    return (u[0] + u[1]) + (u[2] + u[3]);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m, 64);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}


// bitonic network, descending, N = 64*K keys: lane l holds elements l*K .. l*K+K-1; partner distances below K are
// in-lane compare-swaps, larger ones go through lane shuffles.
template <int K>
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long (&key)[K], int lane) {
    for (int k = 2; k <= 64 * K; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= K) {  // partner in another lane
                const int lm = j / K;
                const bool upper = (lane & lm) != 0;
#pragma unroll
                for (int s_ = 0; s_ < K; ++s_) {
                    const int idx = lane * K + s_;
                    const bool desc = (idx & k) == 0;
                    const unsigned long long mine = key[s_], other = shfl_xor_u64(mine, lm);
                    const bool take_max = (desc != upper);  // lower element of a descending pair keeps the max
                    const bool other_gt = other > mine;
                    key[s_] = (other_gt == take_max) ? other : mine;
                }
            } else {  // partner in this lane: static slot pairs
#pragma unroll
                for (int jj = 1; jj < K; jj <<= 1) {
                    if (jj == j) {
#pragma unroll
                        for (int s_ = 0; s_ < K; ++s_) {
                            if ((s_ & jj) == 0) {
                                const int idx = lane * K + s_;
                                const bool desc = (idx & k) == 0;
                                const unsigned long long ka = key[s_], kb = key[s_ | jj];
                                const bool sw = (ka < kb) == desc;
                                key[s_] = sw ? kb : ka;
                                key[s_ | jj] = sw ? ka : kb;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Contract C8 on a sorted key array (K per lane): sequential fp32 sum c_i = c_{i-1} + p_i over positions < len, lane after
// lane; returns #{c_i <= thr} before the first c_i > thr; passed = such an element was met before position len.
template <int K>
__device__ __forceinline__ int cumsum_count(const unsigned long long (&key)[K], int lane, int len, float thr,
                                            bool& passed) {
    int count = 0;
    float c = 0.0f;
    bool done = false, over = false;
    for (int ln = 0; ln < 64 && !done; ++ln) {
        float cl = c;
        int cnt = 0;
        bool dl = false, ol = false;
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {
            const int pos = ln * K + s_;
            if (!dl && pos < len) {
                cl = cl + __uint_as_float((unsigned)(key[s_] >> 32));
                if (cl <= thr) ++cnt; else { dl = true; ol = true; }
            } else if (pos >= len) {
                dl = true;
            }
        }
        // take lane ln's result
        c = __shfl(cl, ln, 64);
        count += __shfl(cnt, ln, 64);
        done = __shfl((int)dl, ln, 64) != 0;
        over = __shfl((int)ol, ln, 64) != 0;
    }
    passed = over;
    return count;
}

#define RSA_SEL_CAP 256   // keys in the sorted head of the prefix path (4 per lane)

// PASS 0: sorted-head path only; a row it cannot decide gets counts[row] = -1 and is left to PASS 1 (a second launch: the
//         full sort needs twice the registers, and keeping it out of this kernel gives it 5 instead of 3 waves per SIMD).
// PASS 1: full sort, only for the rows PASS 0 marked.      PASS 2: full sort for every row (short rows, or A/B tests).
template <int KPL, int PASS>
__global__ __launch_bounds__(256) void select_mask_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wv;
    if (row >= a.rows_total) return;
    if constexpr (PASS == 1) {
        if (a.counts[row] != -1) return;
    }
    const int qblk = (int)(row % a.NBv);
    unsigned char* base = smem + (size_t)wv * a.lds_per_wave;
    float* et = reinterpret_cast<float*>(base);   // et[u] = column NBv + u
    float* pr = et + a.et_len;
    uint8_t* kept = reinterpret_cast<uint8_t*>(pr + ((a.L + 3) & ~3));
    const float* sc = a.scores + row * a.NS;
    const bool has_txt = a.n_txt > 0;

    // Every global read of the row is issued up front, all in flight together: the scores, the row's GAPR bytes and its
    // neighbour bytes.  (As dynamic-trip loops each load was waited for on its own -- 18 + 15 + 15 serialised L2 / Infinity
    // Cache latencies per wave, half of the wave's lifetime by the counters.)  L <= 64 KPL, so the visual columns fit
    // KPL registers; score columns past 64 (KPL + 8) -- more than 512 text tokens -- take the loop below.
    constexpr int NE = KPL + 8;
    float xv[NE];
    uint8_t urv[KPL], nbv[KPL];
#pragma unroll
    for (int m = 0; m < NE; ++m) {
        const int j = lane + 64 * m;
        xv[m] = j < a.NS ? sc[j] : 0.0f;
    }
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        const int j = lane + 64 * m;
        urv[m] = j < a.NBv ? a.unrel[row * a.NBv + j] : (uint8_t)0;
        nbv[m] = (j < a.NBv && a.neighbor) ? a.neighbor[(long)qblk * a.NBv + j] : (uint8_t)0;
    }
    // The row's first 64 NE columns live in registers (xv) through the softmax; only the text columns go to LDS (et[]),
    // where the text partial sums index them by token.  Columns past 64 NE keep the LDS form.  Partial sums take the same
    // elements in the same order as the contract's 256 strided sums: partial (m & 3) of lane `lane`, m ascending.
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < NE; ++m) {
        const int j = lane + 64 * m;
        if (j < a.NS) {
            xv[m] = xv[m] * a.scale;
            mx = fmaxf(mx, xv[m]);
        }
    }
    for (int j = lane + 64 * NE; j < a.NS; j += 64) {
        const float x = sc[j] * a.scale;
        et[j - a.NBv] = x;
        mx = fmaxf(mx, x);
    }
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    // exp and denominator: element j -> partial (j % 256) = lane + 64*((j >> 6) & 3), sequential in j
    float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < NE; ++m) {
        const int j = lane + 64 * m;
        if (j < a.NS) {
            xv[m] = rsa_exp(xv[m] - mx);
            part[m & 3] = part[m & 3] + xv[m];
        }
    }
    for (int j = lane + 64 * NE, m = NE; j < a.NS; j += 64, ++m) {
        const float v = rsa_exp(et[j - a.NBv] - mx);
        et[j - a.NBv] = v;
        if ((m & 3) == 0) part[0] = part[0] + v;
        else if ((m & 3) == 1) part[1] = part[1] + v;
        else if ((m & 3) == 2) part[2] = part[2] + v;
        else part[3] = part[3] + v;
    }
    const float Z = wave_tree4(part);
#pragma unroll
    for (int m = 0; m < NE; ++m) {
        const int j = lane + 64 * m;
        if (j < a.NS) {
            xv[m] = xv[m] / Z;
            if (j >= a.NBv) et[j - a.NBv] = xv[m];
        }
    }
    for (int j = lane + 64 * NE; j < a.NS; j += 64) et[j - a.NBv] = et[j - a.NBv] / Z;
    if (has_txt) {  // IPAR
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float pn[4] = {0.0f, 0.0f, 0.0f, 0.0f}, pt[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < KPL; ++m) {
            if (lane + 64 * m < a.NBv) pn[m & 3] = pn[m & 3] + xv[m];
        }
        for (int u = lane, m = 0; u < a.n_txt; u += 64, ++m) {
            const float v = et[u];
            if ((m & 3) == 0) pt[0] = pt[0] + v; else if ((m & 3) == 1) pt[1] = pt[1] + v;
            else if ((m & 3) == 2) pt[2] = pt[2] + v; else pt[3] = pt[3] + v;
        }
        const float normal_sum = wave_tree4(pn);
        const float text_sum = wave_tree4(pt);
        const float denom = normal_sum * 128.0f + text_sum;
#pragma unroll
        for (int m = 0; m < KPL; ++m) {
            const int j = lane + 64 * m;
            if (j < a.NBv) pr[j] = (xv[m] * 128.0f) / denom;
        }
        if (lane == 0) pr[a.NBv] = text_sum / denom;
    } else {
#pragma unroll
        for (int m = 0; m < KPL; ++m) {
            const int j = lane + 64 * m;
            if (j < a.NBv) pr[j] = xv[m];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int m = 0; m < KPL; ++m) {
        const int j = lane + 64 * m;
        if (j < a.L) a.probs[row * a.L + j] = pr[j];
    }

    // keys into registers: element idx = lane*KPL + s
    unsigned long long key[KPL];
#pragma unroll
    for (int s_ = 0; s_ < KPL; ++s_) {
        const int idx = lane * KPL + s_;
        unsigned long long kk = 0ull;
        if (idx < a.L)
            kk = ((unsigned long long)__float_as_uint(pr[idx]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
        key[s_] = kk;
    }
    int n = -1;
    bool marked = false;
    if constexpr (PASS == 0) {
        // ---- prefix path (C7/C8 unchanged): the decision needs only the sorted HEAD of the row -- the sequential sum
        // stops at the first element that exceeds thr, and the kept set is the first n = max(count+1, top_k) elements.
        // Find a probability threshold t with need <= #{p >= t} <= RSA_SEL_CAP by bisection on the fp32 bit patterns
        // (positive floats order like their bits), compact those keys (they are exactly the top-C of the total order),
        // sort them (256 keys = 4 per lane instead of 64*KPL), run the same sequential sum.  If no such t exists (a
        // plateau of equal probabilities across the window) or the sum does not pass thr inside the head, fall through
        // to the full sort below -- same result either way.
        const int need = a.top_k > RSA_SEL_CAP / 2 ? a.top_k : RSA_SEL_CAP / 2;
        {
            unsigned pmax = 0;
#pragma unroll
            for (int s_ = 0; s_ < KPL; ++s_) pmax = max(pmax, (unsigned)(key[s_] >> 32));
            for (int m = 1; m < 64; m <<= 1) pmax = max(pmax, (unsigned)__shfl_xor((int)pmax, m, 64));
            // (alternating the bisection with interpolation on the counts was measured: 236.4 vs 236 us -- the search is
            // not where this kernel's time goes)
            unsigned lo = 0u, hi = pmax + 1u, t = 0u;   // #{p >= lo} > CAP (all L > CAP keys), #{p >= hi} = 0 < need
            int C = 0;
            bool found = false;
            for (int it = 0; it < 34 && hi - lo > 1u; ++it) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
#pragma unroll
                for (int s_ = 0; s_ < KPL; ++s_) c += __popcll(__ballot((unsigned)(key[s_] >> 32) >= mid));
                if (c > RSA_SEL_CAP) lo = mid;
                else if (c < need) hi = mid;
                else { t = mid; C = c; found = true; break; }
            }
            if (found) {
                // candidate keys: over et[] (the text exponentials are dead once pr[] is written)
                unsigned long long* cand = reinterpret_cast<unsigned long long*>(et);
                for (int j = lane; j < RSA_SEL_CAP; j += 64) cand[j] = 0ull;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                int basep = 0;
#pragma unroll
                for (int s_ = 0; s_ < KPL; ++s_) {
                    const bool f = (unsigned)(key[s_] >> 32) >= t;
                    const unsigned long long mk = __ballot(f);
                    if (f) cand[basep + __popcll(mk & ((1ull << lane) - 1ull))] = key[s_];
                    basep += __popcll(mk);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                constexpr int CK = RSA_SEL_CAP / 64;
                unsigned long long ck[CK];
#pragma unroll
                for (int s_ = 0; s_ < CK; ++s_) ck[s_] = cand[lane * CK + s_];
                bitonic_sort_desc<CK>(ck, lane);
                bool passed = false;
                const int count = cumsum_count<CK>(ck, lane, C, a.thr, passed);
                if (passed || C >= a.L) {
                    n = count + 1;
                    if (n < a.top_k) n = a.top_k;
                    if (n > a.L) n = a.L;
                    if (n <= C) {
                        for (int j = lane; j < a.NB_total; j += 64) kept[j] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                        for (int s_ = 0; s_ < CK; ++s_) {
                            const int pos = lane * CK + s_;
                            if (pos < n) kept[0xFFFFFFFFu - (unsigned)(ck[s_] & 0xFFFFFFFFull)] = 1;
                        }
                        marked = true;
                    }
                }
            }
        }
    }
    if constexpr (PASS == 0) {
        if (!marked) {   // undecided: PASS 1 redoes this row with the full sort
            if (lane == 0) a.counts[row] = -1;
            return;
        }
    } else {
        bitonic_sort_desc<KPL>(key, lane);
        // sequential cumulative sum over the sorted order (C8), lane after lane, stop once it exceeds thr
        bool passed = false;
        const int count = cumsum_count<KPL>(key, lane, a.L, a.thr, passed);
        n = count + 1;
        if (n < a.top_k) n = a.top_k;
        if (n > a.L) n = a.L;
        for (int j = lane; j < a.NB_total; j += 64) kept[j] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int s_ = 0; s_ < KPL; ++s_) {
            const int pos = lane * KPL + s_;
            if (pos < n) kept[0xFFFFFFFFu - (unsigned)(key[s_] & 0xFFFFFFFFull)] = 1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int m = 0; m < KPL; ++m) {   // visual columns: j < NBv <= 64 KPL
        const int j = lane + 64 * m;
        if (j < a.NBv) {
            uint8_t kj = kept[j];
            kj |= (nbv[m] != 0);
            if (qblk < a.ffb && j < a.ffb) kj = 1;
            kept[j] = kj;
        }
    }
    for (int j = a.NBv + lane; j < a.NB_total; j += 64) {   // text blocks and the padding behind them
        uint8_t kj = kept[j];
        if (has_txt && j < a.text_end_block) kj = 1;
        if (qblk < a.ffb && j < a.ffb) kj = 1;
        kept[j] = kj;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // rectification factor and compensation weights
    float pR[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < KPL; ++m) {   // L <= 64 KPL; partial (m & 3) as in the contract's 256 strided sums
        const int j = lane + 64 * m;
        if (j < a.L) {
            bool mm = kept[j] != 0;
            if (j < a.NBv) mm = mm || (urv[m] != 0);
            const float pj = pr[j];
            const float v = mm ? pj : 0.0f;
            pR[m & 3] = pR[m & 3] + v;
            a.w[row * a.L + j] = mm ? 0.0f : pj;
        }
    }
    const float Rv = wave_tree4(pR);
    if (lane == 0) a.R[row] = Rv;
    // bitmask + ascending column list
    int off = 0;
#pragma unroll
    for (int mb = 0; mb <= KPL; ++mb) {   // NB_total <= 64 KPL + 63 (checked by the host): static trip count, LDS reads in flight together
        const int b0 = 64 * mb;
        if (b0 >= a.NB_total) break;
        const int j = b0 + lane;
        const bool f = j < a.NB_total && kept[j] != 0;
        const unsigned long long m = __ballot(f);
        if (f) a.cols[row * a.NB_total + off + __popcll(m & ((1ull << lane) - 1ull))] = j;
        off += __popcll(m);
        const int wi = (b0 >> 5) + (lane >> 5);
        if ((lane & 31) == 0 && wi < a.NW) a.bitmask[row * a.NW + wi] = (unsigned)(m >> (lane & 32));
    }
    if (lane == 0) a.counts[row] = off;
}

// =====================================================================================================
// K3 for LONG rows (round 5): one 256-thread WORKGROUP per (bh, q-block) row, the row in LDS instead of registers.
// The one-wave-per-row kernel above keeps a row's probabilities, text exponentials and keep bytes in 16 KB of LDS per wave and
// its keys in registers: rows beyond ~2 860 visual blocks (a 257-frame 720p video) did not fit and were refused (rounds 2-4).
// This form serves them up to L = 8 192 sorted entries (the walk of K5 holds its kept list as u16 in LDS: 8 192 key blocks is
// the path's limit anyway).  Same contract C5..C8, formulated as the oracle formulates it: thread t owns the elements
// j = t (mod 256) in ascending j -- which IS the contract's 256 strided partial sums --, block_tree_sum is its pairwise tree;
// the total order (probability desc, index asc) comes from a full bitonic sort of the 64-bit keys in LDS (no sorted-head
// shortcut: 91 passes of 16 compare-exchanges per thread at 8 192 keys, ~12 us per row -- 1 % of what K5 spends on such a row's
// kept blocks); the sequential cumulative sum runs on one thread and stops at the first sum above the threshold.
// dynamic LDS: float xs[NS] | u64 keys[N2] | u8 kept[NB_total]  (<= 36 + 64 + 8 KB)
// =====================================================================================================
__global__ __launch_bounds__(256) void select_mask_long_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float red[4];
    __shared__ int sh_n;
    __shared__ int sh_scan[256];
    const int t = threadIdx.x;
    const long row = blockIdx.x;
    const int qblk = (int)(row % a.NBv);
    float* xs = reinterpret_cast<float*>(smem);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem + (((size_t)a.NS * 4 + 15) & ~(size_t)15));
    uint8_t* kept = reinterpret_cast<uint8_t*>(keys + a.N2);
    const float* sc = a.scores + row * a.NS;
    const bool has_txt = a.n_txt > 0;

    // scaled scores, row maximum
    float mx = -INFINITY;
    for (int j = t; j < a.NS; j += 256) {
        const float x = sc[j] * a.scale;
        xs[j] = x;
        mx = fmaxf(mx, x);
    }
    mx = block_max(mx, red);
    // exp and denominator (C6: partial t takes the elements j = t mod 256, ascending)
    float part = 0.0f;
    for (int j = t; j < a.NS; j += 256) {
        const float e = rsa_exp(xs[j] - mx);
        xs[j] = e;
        part = part + e;
    }
    const float Z = block_tree_sum(part, red);
    for (int j = t; j < a.NS; j += 256) xs[j] = xs[j] / Z;
    if (has_txt) {   // IPAR: the text tokens' probabilities collapse into one entry (column NBv)
        __syncthreads();      // (text entry u is divided by thread (NBv + u) mod 256 and summed by thread u mod 256)
        float pn = 0.0f, pt = 0.0f;
        for (int j = t; j < a.NBv; j += 256) pn = pn + xs[j];
        for (int u = t; u < a.n_txt; u += 256) pt = pt + xs[a.NBv + u];
        const float normal_sum = block_tree_sum(pn, red);
        const float text_sum = block_tree_sum(pt, red);      // (its barriers sit behind every thread's reads of the text entries)
        const float denom = normal_sum * 128.0f + text_sum;
        __syncthreads();
        for (int j = t; j < a.NBv; j += 256) xs[j] = (xs[j] * 128.0f) / denom;
        if (t == 0) xs[a.NBv] = text_sum / denom;
    }
    __syncthreads();
    for (int j = t; j < a.L; j += 256) a.probs[row * a.L + j] = xs[j];
    // keys (probability bits << 32 | ~index), padded with zeros to the power of two; full bitonic sort, descending
    for (int idx = t; idx < a.N2; idx += 256)
        keys[idx] = idx < a.L ? (((unsigned long long)__float_as_uint(xs[idx]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx)) : 0ull;
    for (int k = 2; k <= a.N2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int i = t; i < (a.N2 >> 1); i += 256) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                const bool desc = (lo & k) == 0;
                const unsigned long long ka = keys[lo], kb = keys[hi];
                if ((ka < kb) == desc) { keys[lo] = kb; keys[hi] = ka; }
            }
        }
    }
    __syncthreads();
    // C8: sequential fp32 sum in sorted order; the sums never decrease, so the count ends at the first one above thr
    if (t == 0) {
        int count = 0;
        float c = 0.0f;
        for (int k = 0; k < a.L; ++k) {
            c = c + __uint_as_float((unsigned)(keys[k] >> 32));
            if (c <= a.thr) ++count; else break;
        }
        int n = count + 1;
        if (n < a.top_k) n = a.top_k;
        if (n > a.L) n = a.L;
        sh_n = n;
    }
    for (int j = t; j < a.NB_total; j += 256) kept[j] = 0;
    __syncthreads();
    const int n = sh_n;
    for (int k = t; k < n; k += 256) kept[0xFFFFFFFFu - (unsigned)(keys[k] & 0xFFFFFFFFull)] = 1;   // column NBv = text block NBv
    __syncthreads();
    for (int j = t; j < a.NB_total; j += 256) {
        uint8_t kj = kept[j];
        if (j < a.NBv) {
            if (a.neighbor && a.neighbor[(long)qblk * a.NBv + j] != 0) kj = 1;
        } else if (has_txt && j < a.text_end_block) {
            kj = 1;
        }
        if (qblk < a.ffb && j < a.ffb) kj = 1;
        kept[j] = kj;
    }
    __syncthreads();
    // rectification factor and compensation weights (C6 partials again)
    float pr = 0.0f;
    for (int j = t; j < a.L; j += 256) {
        bool mm = kept[j] != 0;
        if (j < a.NBv) mm = mm || (a.unrel[row * a.NBv + j] != 0);
        const float pj = xs[j];
        pr = pr + (mm ? pj : 0.0f);
        a.w[row * a.L + j] = mm ? 0.0f : pj;
    }
    const float Rv = block_tree_sum(pr, red);
    if (t == 0) a.R[row] = Rv;
    // bitmask words
    for (int wi = t; wi < a.NW; wi += 256) {
        unsigned wd = 0;
        for (int b = 0; b < 32; ++b) {
            const int j = 32 * wi + b;
            if (j < a.NB_total && kept[j]) wd |= 1u << b;
        }
        a.bitmask[row * a.NW + wi] = wd;
    }
    // ascending list: thread t owns blocks [t * ch, (t + 1) * ch); exclusive scan of the per-thread counts
    const int ch = (a.NB_total + 255) / 256;
    const int b0 = t * ch, b1 = min(b0 + ch, a.NB_total);
    int mine = 0;
    for (int j = b0; j < b1; ++j) mine += kept[j] != 0;
    sh_scan[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = t >= d ? sh_scan[t - d] : 0;
        __syncthreads();
        sh_scan[t] += v;
        __syncthreads();
    }
    int off = sh_scan[t] - mine;
    for (int j = b0; j < b1; ++j)
        if (kept[j]) a.cols[row * a.NB_total + off++] = j;
    if (t == 255) a.counts[row] = sh_scan[255];
}

// =====================================================================================================
// K4: comp[i, :] = sum_j w[i, j] * vbar[j, :]   (tolerance-only quantity; fp32, j ascending) on the fp32 matrix pipe,
// same chain form as K2: v_mfma_f32_32x32x2_f32 over j = 0,1 | 2,3 | ...
// Workgroup = D/32 waves = 32 query-block rows x D; wave = one 32 x 32 tile of comp (16 accumulator registers), all waves
// share the A operand (the w rows).  j staged in chunks of 32 through registers -> LDS:
//   w  as [row][16 even j | 16 odd j | pad 4]  (lane (r, h) feeds j = 2s + h: 4 steps per ds_read_b128, conflict-free)
//   vbar as [j][D]                              (lane (r, h) reads element d0 + r of row 2s + h: one ds_read_b32 per MFMA)
// 32-row tiles keep the grid at 696 workgroups for the Hunyuan shape (2.7 per CU).  Measured 67 us (97 on the VALU) with
// the matrix pipe 49 % busy.  Tried on top, all within +-5 us: one wave per tile without barriers, 64-wide chunks, two
// LDS buffers with one or two register stages in flight, all LDS operand reads of a chunk ahead of its MFMAs, an occupancy
// cap of 3 per CU; a timing-only build that never refills the tiles runs 57 us.  In-kernel stamps: workgroups spread
// 2-3 per CU, a wave lives 128k cycles of which its own 464 MFMAs are 30k.
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(D * 2) void compensation_kernel(const float* w, const float* vbar, float* comp, int NBv,
                                                             int L, int NB_total) {
    constexpr int TJ = 32, LDW = TJ + 4, NT = D * 2, NWK = 32 * TJ / NT, NVK = TJ * D / 4 / NT;
    __shared__ __attribute__((aligned(16))) float Ws[32 * LDW];
    __shared__ __attribute__((aligned(16))) float Vs[TJ * D];
    const int bh = blockIdx.y, i0 = blockIdx.x * 32, t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const float* wp = w + (long)bh * NBv * L;
    const float* vp = vbar + (long)bh * NB_total * D;
    k2_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // register-staged double buffering: chunk j0 + TJ is loaded from global memory while chunk j0 is multiplied
    float wreg[NWK];
    float4 vreg[NVK];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int k = 0; k < NWK; ++k) {
            const int idx = t + k * NT;
            const int ii = idx / TJ, jj = idx % TJ;   // rows past NBv are clamped: they feed outputs that are not stored
            const int row = min(i0 + ii, NBv - 1);
            wreg[k] = j0 + jj < L ? wp[(long)row * L + j0 + jj] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NVK; ++k) {
            const int idx = t + k * NT;
            const int jj = idx / (D / 4), dd = (idx % (D / 4)) * 4;
            vreg[k] = make_float4(0, 0, 0, 0);
            if (j0 + jj < L) vreg[k] = *reinterpret_cast<const float4*>(vp + (long)(j0 + jj) * D + dd);
        }
    };
    load_chunk(0);
    const float* pa = &Ws[r * LDW + (TJ / 2) * h];
    const float* pb = &Vs[h * D + 32 * wv + r];
    for (int j0 = 0; j0 < L; j0 += TJ) {
        __syncthreads();   // the previous chunk's LDS reads are done
#pragma unroll
        for (int k = 0; k < NWK; ++k) {
            const int idx = t + k * NT;
            const int ii = idx / TJ, jj = idx % TJ;
            Ws[ii * LDW + (TJ / 2) * (jj & 1) + (jj >> 1)] = wreg[k];
        }
#pragma unroll
        for (int k = 0; k < NVK; ++k) {
            const int idx = t + k * NT;
            *reinterpret_cast<float4*>(&Vs[(idx / (D / 4)) * D + (idx % (D / 4)) * 4]) = vreg[k];
        }
        __syncthreads();
        if (j0 + TJ < L) load_chunk(j0 + TJ);   // in flight during the MFMAs below
#pragma unroll
        for (int m = 0; m < TJ / 8; ++m) {
            const float4 a4 = *reinterpret_cast<const float4*>(pa + 4 * m);
            const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int x = 0; x < 4; ++x)   // step s = 4m + x of the chunk: j = j0 + 2s (h = 0), j0 + 2s + 1 (h = 1)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[x], pb[(2 * (4 * m + x)) * D], acc, 0, 0, 0);
        }
    }
    // accumulator element e of lane (r, h): row (e & 3) + 8 (e >> 2) + 4 h, column r of the wave's 32 x 32 tile
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i < NBv) comp[((long)bh * NBv + i) * D + 32 * wv + r] = acc[e];
    }
}

// =====================================================================================================
// K4, split form (round 6; the product): the same product on the 2-byte matrix pipe.  comp is a tolerance-only quantity
// (<= 1e-5 against the oracle's fp32 chain), so the fp32 operands are split x = hi + lo + rest, hi = bf16(x), lo = bf16(x - hi)
// (|rest| <= 2^-17 |x|), and w . vbar = hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: three MFMAs of
// 32 cycles per 16 j where the fp32 chain takes eight of 64 -- 5.3x fewer matrix cycles; the dropped terms are 2^-16 relative per
// product (measured on the headline shape: max |comp - oracle| 3e-7, tests/test_gpu_select_paths.py).
// Same tiling as the chain form (32 rows x D per workgroup, wave = one 32 x 32 tile, j in chunks, register-staged loads of the
// next chunk in flight during the MFMAs).  The STAGING threads do the split, once per element: the chunk sits in LDS as four
// bf16 planes [row][64 j] -- w hi / lo by query-block row, vbar hi / lo TRANSPOSED ([d][j], written as packed pairs of adjacent
// j, so both operands are k-contiguous: lane (r, h) of the MFMA phase reads its 8 bf16 of a k-step with two ds_read_b64).
// Row stride 136 B: writes of a wave fall on 16 banks twice (free), the b64 reads on 32 distinct even bank pairs.
// =====================================================================================================
typedef short k4_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 k4_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 k4_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void k4_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    k4_bf16x2 h2;
    h2[0] = (__bf16)x0; h2[1] = (__bf16)x1;
    hi = __builtin_bit_cast(unsigned, h2);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
    k4_bf16x2 l2;
    l2[0] = (__bf16)r0; l2[1] = (__bf16)r1;
    lo = __builtin_bit_cast(unsigned, l2);
}
__device__ __forceinline__ k4_bf16x8 k4_read8(const unsigned char* p) {   // 8 bf16 = two 8-byte aligned LDS reads
    const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 8);
    const uint4 v = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(k4_bf16x8, v);
}

template <int D>
__global__ __launch_bounds__(D * 2) void compensation_split_kernel(const float* w, const float* vbar, float* comp, int NBv,
                                                                   int L, int NB_total) {
    constexpr int TJ = 64, RS = 2 * TJ + 8, NT = D * 2, RPP = NT / 32, WP = 32 / RPP, VP = TJ / 4;
    __shared__ __attribute__((aligned(16))) unsigned char Wh[32 * RS], Wl[32 * RS], Vh[D * RS], Vl[D * RS];
    const int bh = blockIdx.y, i0 = blockIdx.x * 32, t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const float* wp = w + (long)bh * NBv * L;
    const float* vp = vbar + (long)bh * NB_total * D;
    k2_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    // staging slots of this thread: w pairs (row (t >> 5) + RPP k, j = 2 (t & 31)), vbar pairs (d = t % D, j = 2 (t / D + 2 k))
    const int wj = 2 * (t & 31), wi = t >> 5, vd = t % D, vj = 2 * (t / D);
    float wr[WP][2], vr[VP][2];
    auto load_chunk = [&](int j0) {
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            const float* row = wp + (long)min(i0 + wi + RPP * k, NBv - 1) * L;   // rows past NBv feed outputs that are not stored
            const int j = j0 + wj;
            wr[k][0] = j < L ? row[j] : 0.0f;
            wr[k][1] = j + 1 < L ? row[j + 1] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < VP; ++k) {
            const int j = j0 + vj + 4 * k;
            vr[k][0] = j < L ? vp[(long)j * D + vd] : 0.0f;
            vr[k][1] = j + 1 < L ? vp[(long)(j + 1) * D + vd] : 0.0f;
        }
    };
    load_chunk(0);
    const unsigned char* pah = Wh + r * RS + 16 * h;
    const unsigned char* pal = Wl + r * RS + 16 * h;
    const unsigned char* pbh = Vh + (32 * wv + r) * RS + 16 * h;
    const unsigned char* pbl = Vl + (32 * wv + r) * RS + 16 * h;
    for (int j0 = 0; j0 < L; j0 += TJ) {
        __syncthreads();   // the previous chunk's LDS reads are done
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            unsigned hi, lo;
            k4_split2(wr[k][0], wr[k][1], hi, lo);
            const int off = (wi + RPP * k) * RS + 2 * wj;
            *reinterpret_cast<unsigned*>(Wh + off) = hi;
            *reinterpret_cast<unsigned*>(Wl + off) = lo;
        }
#pragma unroll
        for (int k = 0; k < VP; ++k) {
            unsigned hi, lo;
            k4_split2(vr[k][0], vr[k][1], hi, lo);
            const int off = vd * RS + 2 * (vj + 4 * k);
            *reinterpret_cast<unsigned*>(Vh + off) = hi;
            *reinterpret_cast<unsigned*>(Vl + off) = lo;
        }
        __syncthreads();
        if (j0 + TJ < L) load_chunk(j0 + TJ);   // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < TJ / 16; ++ks) {
            const k4_bf16x8 ah = k4_read8(pah + 32 * ks), al = k4_read8(pal + 32 * ks);
            const k4_bf16x8 bh8 = k4_read8(pbh + 32 * ks), bl8 = k4_read8(pbl + 32 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh8, acc, 0, 0, 0);   // the small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh8, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i < NBv) comp[((long)bh * NBv + i) * D + 32 * wv + r] = acc[e];
    }
}
int g_rsa_k4_split = 1;   // tuning key "k4_split": 0 = the fp32 chain form (A/B, tests)

// =====================================================================================================
// stand-alone GAPR mask for estimate_pr_gain callers: mask = !(|s| > |aq.kbar| + |qbar.ak|)
// =====================================================================================================
__global__ void gapr_compare_kernel(const float* qbar, const float* aq, const float* kbar, const float* ak,
                                    const float* scores, uint8_t* mask, int NQ, int NK, int D) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, bh = blockIdx.z;
    if (j >= NK) return;
    const float* q = qbar + ((long)bh * NQ + i) * D;
    const float* a = aq + ((long)bh * NQ + i) * D;
    const float* k = kbar + ((long)bh * NK + j) * D;
    const float* b = ak + ((long)bh * NK + j) * D;
    float eq = 0.0f, ek = 0.0f;
    for (int d = 0; d < D; ++d) {
        eq = __builtin_fmaf(a[d], k[d], eq);
        ek = __builtin_fmaf(q[d], b[d], ek);
    }
    const float s = scores[((long)bh * NQ + i) * NK + j];
    mask[((long)bh * NQ + i) * NK + j] = !(fabsf(s) > (fabsf(eq) + fabsf(ek)));
}

// =====================================================================================================
// host entry points
// =====================================================================================================
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int rsa_buffer_bytes(const rsa_layout* l, size_t sizes[RSA_NUM_BUFFERS], size_t* total) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!sizes || !total) return RSA_ERR_BAD_ARG;
    const size_t BH = (size_t)l->B * l->H, NBv = l->NBv, NB = l->NB_total, D = l->D;
    const size_t NS = NBv + l->n_txt, L = NBv + (l->n_txt > 0 ? 1 : 0), NW = (NB + 31) / 32;
    const size_t s[RSA_NUM_BUFFERS] = {
        BH * NBv * D * 4, BH * NBv * D * 4, BH * NBv * D * 4, BH * NBv * D * 4, BH * NB * D * 4,
        BH * NBv * NS * 4, BH * NBv * NBv,  BH * NBv * L * 4, BH * NBv * L * 4, BH * NBv * 4,
        BH * NBv * D * 4,  BH * NBv * NW * 4, BH * NBv * NB * 4, BH * NBv * 4,
        (BH * (NB - NBv) * RSA_TEXT_SPLIT + RSA_TAIL_PIECES) * 128 * (D + 2) * 4};
    size_t tot = 0;
    for (int i = 0; i < RSA_NUM_BUFFERS; ++i) {
        sizes[i] = s[i];
        tot += align256(s[i] ? s[i] : 1);
    }
    *total = tot;
    return RSA_OK;
}

extern "C" int rsa_carve_workspace(const rsa_layout* l, void* ws, size_t ws_bytes, rsa_buffers* out) {
    size_t sizes[RSA_NUM_BUFFERS], total;
    int st = rsa_buffer_bytes(l, sizes, &total);
    if (st != RSA_OK) return st;
    if (!ws || !out || (reinterpret_cast<uintptr_t>(ws) & 255)) return RSA_ERR_BAD_ARG;
    if (ws_bytes < total) return RSA_ERR_WORKSPACE;
    unsigned char* p = static_cast<unsigned char*>(ws);
    void* ptrs[RSA_NUM_BUFFERS];
    for (int i = 0; i < RSA_NUM_BUFFERS; ++i) {
        ptrs[i] = p;
        p += align256(sizes[i] ? sizes[i] : 1);
    }
    out->qbar = (float*)ptrs[0]; out->aq = (float*)ptrs[1]; out->kbar = (float*)ptrs[2]; out->ak = (float*)ptrs[3];
    out->vbar = (float*)ptrs[4]; out->scores = (float*)ptrs[5]; out->unrel = (uint8_t*)ptrs[6];
    out->probs = (float*)ptrs[7]; out->w = (float*)ptrs[8]; out->R = (float*)ptrs[9]; out->comp = (float*)ptrs[10];
    out->bitmask = (uint32_t*)ptrs[11]; out->cols = (int32_t*)ptrs[12]; out->counts = (int32_t*)ptrs[13];
    out->tpart = (float*)ptrs[14];   // every layout has the tail region (layouts without text rows too: Wan)
    out->tpart_bytes = sizes[14];
    return RSA_OK;
}

// K1, optionally writing the e4m3 images of the blocks it pools (rsa_common.h); rsa_pool_stats is the public form without
int rsa_pool_stats_f8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v, const rsa_buffers* buf,
                      const Fp8Emit* f8, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!buf || !buf->qbar || !buf->aq || !buf->kbar || !buf->ak || !buf->vbar) return RSA_ERR_BAD_ARG;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v))) return st;
    PoolArgs a;
    const rsa_tensor4* ts[3] = {&q, &k, &v};
    for (int i = 0; i < 3; ++i) {
        a.src[i] = static_cast<const unsigned short*>(ts[i]->ptr);
        a.sb[i] = ts[i]->stride_b; a.sh[i] = ts[i]->stride_h; a.ss[i] = ts[i]->stride_s;
        a.ext_mean[i] = nullptr;
    }
    const int vis_tok = l->NBv * RSA_BLOCK;
    a.mean[0] = buf->qbar; a.mad[0] = buf->aq; a.nblk[0] = l->NBv; a.valid[0] = l->S < vis_tok ? l->S : vis_tok;
    a.mean[1] = buf->kbar; a.mad[1] = buf->ak; a.nblk[1] = l->NBv;
    a.valid[1] = l->pool_valid < vis_tok ? l->pool_valid : vis_tok;
    a.mean[2] = buf->vbar; a.mad[2] = nullptr; a.nblk[2] = l->NB_total; a.valid[2] = l->pool_valid;
    a.H = l->H;
    a.BH = l->B * l->H; a.NB_total = l->NB_total;
    dim3 grid(l->NB_total, l->B * l->H, 3);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (f8 != nullptr) {
        a.f8 = *f8;
        const dim3 grid8(3 * l->NB_total, l->B * l->H, 1);
        if (l->D == 128) {
            if (l->dtype == RSA_BF16) pool_stats_kernel<128, bf16_tag, true><<<grid8, 256, 0, s>>>(a);
            else pool_stats_kernel<128, fp16_tag, true><<<grid8, 256, 0, s>>>(a);
        } else {
            if (l->dtype == RSA_BF16) pool_stats_kernel<64, bf16_tag, true><<<grid8, 128, 0, s>>>(a);
            else pool_stats_kernel<64, fp16_tag, true><<<grid8, 128, 0, s>>>(a);
        }
        return rsa_launch_status();
    }
    if (l->D == 128) {
        if (l->dtype == RSA_BF16) pool_stats_kernel<128, bf16_tag, false><<<grid, 256, 0, s>>>(a);
        else pool_stats_kernel<128, fp16_tag, false><<<grid, 256, 0, s>>>(a);
    } else {
        if (l->dtype == RSA_BF16) pool_stats_kernel<64, bf16_tag, false><<<grid, 128, 0, s>>>(a);
        else pool_stats_kernel<64, fp16_tag, false><<<grid, 128, 0, s>>>(a);
    }
    return rsa_launch_status();
}

extern "C" int rsa_pool_stats(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                              const rsa_buffers* buf, void* stream) {
    return rsa_pool_stats_f8(l, q, k, v, buf, nullptr, stream);
}

extern "C" int rsa_pooled_scores(const rsa_layout* l, rsa_tensor4 k, const rsa_buffers* buf, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!buf || !buf->qbar || !buf->aq || !buf->kbar || !buf->ak || !buf->scores || !buf->unrel)
        return RSA_ERR_BAD_ARG;
    if ((st = rsa_check_tensor(k))) return st;
    if (l->NBv == 0) return RSA_OK;
    ScoreArgs a;
    a.qbar = buf->qbar; a.aq = buf->aq; a.kbar = buf->kbar; a.ak = buf->ak;
    a.ktxt = static_cast<const unsigned short*>(k.ptr);
    a.ksb = k.stride_b; a.ksh = k.stride_h; a.kss = k.stride_s;
    a.scores = buf->scores; a.unrel = buf->unrel;
    a.NBv = l->NBv; a.n_txt = l->n_txt; a.NS = l->NBv + l->n_txt; a.D = l->D; a.H = l->H;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int BH = l->B * l->H;
    const unsigned nti = (unsigned)((l->NBv + 63) / 64);
    a.BH = BH;
    dim3 g0((nti * (nti + (unsigned)((l->n_txt + 63) / 64)) * BH + 7) / 8 * 8);
    // (the DMA form addresses an operand row as a 32-bit byte offset from its head's base: NBv * D * 4 < 2^32 always -- NBv <= 8 192)
    if (g_rsa_k2_dma == 32 && (l->D % 32) == 0) {
        if (l->dtype == RSA_BF16) pooled_scores_kernel<bf16_tag, 32><<<g0, 256, 0, s>>>(a);
        else pooled_scores_kernel<fp16_tag, 32><<<g0, 256, 0, s>>>(a);
    } else if (g_rsa_k2_dma == 16 && (l->D % 32) == 0) {
        if (l->dtype == RSA_BF16) pooled_scores_kernel<bf16_tag, 16><<<g0, 256, 0, s>>>(a);
        else pooled_scores_kernel<fp16_tag, 16><<<g0, 256, 0, s>>>(a);
    } else {
        if (l->dtype == RSA_BF16) pooled_scores_kernel<bf16_tag, 0><<<g0, 256, 0, s>>>(a);
        else pooled_scores_kernel<fp16_tag, 0><<<g0, 256, 0, s>>>(a);
    }
    return rsa_launch_status();
}

extern "C" int rsa_select_mask(const rsa_layout* l, const uint8_t* neighbor, int top_k, float p_remain,
                               const rsa_buffers* buf, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!buf || !buf->scores || !buf->unrel || !buf->probs || !buf->w || !buf->R || !buf->bitmask || !buf->cols ||
        !buf->counts || top_k < 0)
        return RSA_ERR_BAD_ARG;
    if (l->NBv == 0) return RSA_OK;
    SelectArgs a;
    a.scores = buf->scores; a.unrel = buf->unrel; a.neighbor = neighbor;
    a.probs = buf->probs; a.w = buf->w; a.R = buf->R; a.bitmask = buf->bitmask; a.cols = buf->cols;
    a.counts = buf->counts;
    a.NBv = l->NBv; a.n_txt = l->n_txt; a.NS = l->NBv + l->n_txt; a.L = l->NBv + (l->n_txt > 0 ? 1 : 0);
    int n2 = 64;
    while (n2 < a.L) n2 <<= 1;
    a.N2 = n2; a.NB_total = l->NB_total; a.NW = (l->NB_total + 31) / 32;
    a.text_end_block = l->text_end_block; a.ffb = l->first_frame_blocks; a.top_k = top_k;
    a.thr = p_remain;
    a.scale = (float)(1.0 / sqrt((double)l->D));  // head_dim ** -0.5 rounded to fp32 (hunyuan :208)
    a.rows_total = l->B * l->H * l->NBv;
    a.use_prefix = g_rsa_k3_prefix;
    a.et_len = (a.n_txt + 3) & ~3;
    if (a.et_len < 2 * RSA_SEL_CAP) a.et_len = 2 * RSA_SEL_CAP;
    const size_t per_wave = (((size_t)a.et_len * 4 + (size_t)((a.L + 3) & ~3) * 4 + (size_t)((a.NB_total + 7) & ~7)) + 15) &
                            ~(size_t)15;
    a.lds_per_wave = (int)per_wave;
    const size_t lds = per_wave * 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_rsa_k3_long || lds > 64 * 1024 || n2 > 4096 || a.NB_total > n2 + 63) {
        // rows too long for the one-wave-per-row kernel (more than ~2 860 visual blocks): the workgroup-per-row form, up to
        // 8 192 sorted entries / key blocks (K5's own limit)
        if (n2 > 8192 || a.NB_total > 8192) return RSA_ERR_UNSUPPORTED;
        const size_t lds_long = (((size_t)a.NS * 4 + 15) & ~(size_t)15) + (size_t)n2 * 8 + (((size_t)a.NB_total + 15) & ~(size_t)15);
        if (lds_long > 150 * 1024) return RSA_ERR_UNSUPPORTED;     // (more than 8 192 + ~5 000 individually scored text tokens)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(select_mask_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_long);
        (void)hipGetLastError();
        select_mask_long_kernel<<<dim3((unsigned)a.rows_total), 256, lds_long, s>>>(a);
        return rsa_launch_status();
    }
    dim3 grid((unsigned)((a.rows_total + 3) / 4));
    // rows longer than the sorted head (256 keys) take the two-pass form when the head can hold top_k
    const int need = top_k > RSA_SEL_CAP / 2 ? top_k : RSA_SEL_CAP / 2;
    const bool two_pass = a.use_prefix && n2 / 64 > RSA_SEL_CAP / 64 && need <= RSA_SEL_CAP && a.L > RSA_SEL_CAP;
#define RSA_K3(K) \
    do { \
        if (two_pass) { \
            select_mask_kernel<K, 0><<<grid, 256, lds, s>>>(a); \
            select_mask_kernel<K, 1><<<grid, 256, lds, s>>>(a); \
        } else { \
            select_mask_kernel<K, 2><<<grid, 256, lds, s>>>(a); \
        } \
    } while (0)
    switch (n2 / 64) {
        case 1: select_mask_kernel<1, 2><<<grid, 256, lds, s>>>(a); break;
        case 2: select_mask_kernel<2, 2><<<grid, 256, lds, s>>>(a); break;
        case 4: select_mask_kernel<4, 2><<<grid, 256, lds, s>>>(a); break;
        case 8: RSA_K3(8); break;
        case 16: RSA_K3(16); break;
        case 32: RSA_K3(32); break;
        default: RSA_K3(64); break;
    }
#undef RSA_K3
    st = rsa_launch_status();
    if (st != RSA_OK) return st;
    return st;
}

extern "C" int rsa_compensation(const rsa_layout* l, const rsa_buffers* buf, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (!buf || !buf->w || !buf->vbar || !buf->comp) return RSA_ERR_BAD_ARG;
    if (l->NBv == 0) return RSA_OK;
    const int L = l->NBv + (l->n_txt > 0 ? 1 : 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((l->NBv + 31) / 32, l->B * l->H);
    // (measured, HunyuanVideo 720p: 24 heads 66.8 -> 58.8 us, 3 heads 30.7 -> 34.3 us: the split form only where the grid fills the chip;
    // k4_split = 2 forces it)
    if (g_rsa_k4_split == 2 || (g_rsa_k4_split == 1 && (long)grid.x * grid.y >= 512)) {
        if (l->D == 128) compensation_split_kernel<128><<<grid, 256, 0, s>>>(buf->w, buf->vbar, buf->comp, l->NBv, L, l->NB_total);
        else compensation_split_kernel<64><<<grid, 128, 0, s>>>(buf->w, buf->vbar, buf->comp, l->NBv, L, l->NB_total);
        return rsa_launch_status();
    }
    if (l->D == 128) compensation_kernel<128><<<grid, 256, 0, s>>>(buf->w, buf->vbar, buf->comp, l->NBv, L, l->NB_total);
    else compensation_kernel<64><<<grid, 128, 0, s>>>(buf->w, buf->vbar, buf->comp, l->NBv, L, l->NB_total);
    return rsa_launch_status();
}

extern "C" int rsa_estimate_pr_gain(int BH, int NQ, int NK, int D, int dtype, const void* q_blocks,
                                    const void* k_blocks, const float* q_pools, const float* k_pools,
                                    const float* scores, float* scratch_aq, float* scratch_ak, uint8_t* mask_out,
                                    void* stream) {
    if (BH <= 0 || NQ <= 0 || NK <= 0 || !q_blocks || !k_blocks || !q_pools || !k_pools || !scores || !scratch_aq ||
        !scratch_ak || !mask_out)
        return RSA_ERR_BAD_ARG;
    if (D != 64 && D != 128) return RSA_ERR_UNSUPPORTED;
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    // mean |X - pool| with the CALLER's pools (gapr_mask.py:19-23,:30); the pool kernel also writes its own block
    // means into the first half of each scratch (unused).  scratch_aq / scratch_ak hold 2 * BH * N * D floats.
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int which = 0; which < 2; ++which) {
        PoolArgs a;
        a.BH = BH; a.NB_total = 0;
        const int N = which == 0 ? NQ : NK;
        float* scratch = which == 0 ? scratch_aq : scratch_ak;
        for (int i = 0; i < 3; ++i) {
            a.src[i] = static_cast<const unsigned short*>(which == 0 ? q_blocks : k_blocks);
            a.sb[i] = 0; a.sh[i] = (long)N * RSA_BLOCK * D; a.ss[i] = D;
            a.mean[i] = scratch; a.mad[i] = scratch + (size_t)BH * N * D;
            a.nblk[i] = i == 0 ? N : 0; a.valid[i] = N * RSA_BLOCK;
            a.ext_mean[i] = which == 0 ? q_pools : k_pools;
        }
        a.H = BH;
        dim3 grid(N, BH, 1);
        if (D == 128) {
            if (dtype == RSA_BF16) pool_stats_kernel<128, bf16_tag, false><<<grid, 256, 0, s>>>(a);
            else pool_stats_kernel<128, fp16_tag, false><<<grid, 256, 0, s>>>(a);
        } else {
            if (dtype == RSA_BF16) pool_stats_kernel<64, bf16_tag, false><<<grid, 128, 0, s>>>(a);
            else pool_stats_kernel<64, fp16_tag, false><<<grid, 128, 0, s>>>(a);
        }
    }
    dim3 g((NK + 127) / 128, NQ, BH);
    gapr_compare_kernel<<<g, 128, 0, s>>>(q_pools, scratch_aq + (size_t)BH * NQ * D, k_pools,
                                          scratch_ak + (size_t)BH * NK * D, scores, mask_out, NQ, NK, D);
    return rsa_launch_status();
}

extern "C" const char* rsa_status_string(int status) {
    switch (status) {
        case RSA_OK: return "ok";
        case RSA_ERR_BAD_ARG: return "bad argument";
        case RSA_ERR_UNSUPPORTED: return "unsupported head_dim / dtype / row length";
        case RSA_ERR_WORKSPACE: return "workspace too small";
        case RSA_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

int g_rsa_last_hip_error = 0;
extern "C" const char* rsa_last_hip_error(void) { return hipGetErrorString((hipError_t)g_rsa_last_hip_error); }

extern "C" int rsa_abi_check(int header_version, size_t sizeof_rsa_buffers, size_t sizeof_rsa_layout) {
    return (header_version / 100 == RSA_HEADER_VERSION / 100 && sizeof_rsa_buffers == sizeof(rsa_buffers) &&
            sizeof_rsa_layout == sizeof(rsa_layout)) ? RSA_OK : RSA_ERR_UNSUPPORTED;
}
extern "C" int rsa_version(void) { return 600; }  // 0.6.0: rsa_abi_check; the pv entry points refuse K spans a 32-bit row offset cannot reach; 0.5.0: rsa_buffers.tpart_bytes (declared capacity of the partial buffer; carve_workspace hands tpart out for every layout), rsa_set_shard_invariant, rsa_comm_count; 0.4.0: rsa_p2p_state_alloc / _free / _timeout (fine-grained exchange state), rsa_dense_masked_fwd; 0.3.1: block-scaled fp8 operands (rsa_fp8_operands.scales = E8M0 words + K mean), K1 writes the images, rsa_fp8_images gone; 0.3.0: rsa_buffers has 15 members, rsa_allgather_heads_p2p is stream-ordered (+ state buffers), rsa_ipc_offset
