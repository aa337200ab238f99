// K2, second form (round 5): pooled scores + GAPR bytes without LDS staging and without a workgroup barrier.
//
// What the contract fixes bit for bit (oracle/rsa_oracle.c, C4): s = qbar . kbar as ONE k-ordered fmaf chain, and the BYTE
// unrel = !(|s| > |aq . kbar| + |qbar . ak|) with both error terms taken as k-ordered fmaf chains too.  The first form of K2
// (rsa_stats.hip::pooled_scores_kernel) runs all three products as chains of v_mfma_f32_32x32x2_f32 -- 16.3 GFLOP at the fp32
// matrix rate (1/16 of the 2-byte rate), two thirds of it for two numbers that are only ever COMPARED.  This form keeps the s
// chain on the fp32 matrix pipe and takes the two error terms from the 2-byte matrix pipe first:
//
//   x = hi + lo + rest, hi = bf16(x), lo = bf16(x - hi)  (|rest| <= 2^-16 |x|),
//   e~ = sum_k  a_hi b_hi + a_hi b_lo + a_lo b_hi         (three v_mfma_f32_32x32x16_bf16 per 16 k: 192 matrix cycles where the
//                                                          fp32 chain takes 512)
//   |e~ - e_chain| <= c_E ||a|| ||b||,  c_E = 2^-12:  2^-17 (the fp32 chain's own D roundings) + 3 * 2^-16 (dropped terms of the
//   split) + 3 D accumulation steps inside the MFMAs at <= 2^-22 each (twice what truncating every step would cost) = 2^-13.4,
//   all relative to sum |a_k b_k| <= ||a|| ||b||: together < 2^-12.7.
//
// With t~ = |eq~| + |ek~| and E = c_E (||aq_i|| ||kbar_j|| + ||qbar_i|| ||ak_j||) (+ rounding slack) the byte is DECIDED whenever
// ||s| - t~| > E: the exact chains cannot land on the other side.  The few elements inside the band (0.3 % on the bench's
// inputs) are queued in LDS by the wave that met them and recomputed with the exact scalar chains, 64 at a time, so the stored
// bytes are the contract's on every input -- non-finite or vanishing operands simply take the exact path.
//
// Work mapping: one WAVE = one item = a 32-row i tile x a range of 32-column j sub-tiles of one (batch, head).  The i side
// (the s chain's A operands, the bf16 images of qbar and aq: 192 registers) stays in registers for the whole item; the j side
// streams through registers straight from L2 in 32-k chunks, the next chunk in flight during the current chunk's MFMAs.  No
// LDS operands, no barrier: the wave owns its SIMD's register file (one wave per SIMD).  Items are dealt XCD-aware: a head's
// pooled operands (1.8 MB) stay in one L2.
#include "rsa_common.h"
#include "rsa_scores2.h"

typedef float s2_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 s2_bf16x8 __attribute__((ext_vector_type(8)));

#define RSA_S2_FIXCAP 1280  // queued undecided elements per wave: a sub-tile adds at most 1 024, the queue is emptied beyond 256

namespace {

constexpr int S2_LD = 36;                      // floats per staged row: [16 even k | 16 odd k | 4 pad] (conflict-free b128 reads)
constexpr int S2_ARR = 32 * S2_LD;             // floats per staged operand chunk (32 rows x 32 k)

// 8 fp32 values -> the two bf16 pieces of each (hi = RN_bf16(x), lo = RN_bf16(x - hi))
__device__ __forceinline__ void split8(const float (&x)[8], s2_bf16x8& hi, s2_bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        const float hf = (float)hb;
        hi[e] = hb;
        lo[e] = (__bf16)(x[e] - hf);
    }
}

__device__ __forceinline__ float sq4(const float4& v) { return (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w); }

// k = 16 mm + 8 h + 0..7 of a staged row in k order: four even and four odd elements, interleaved
__device__ __forceinline__ void image8(const float* rowp, int mm, int h, float (&x)[8]) {
    const float4 ev = *reinterpret_cast<const float4*>(rowp + 8 * mm + 4 * h);
    const float4 od = *reinterpret_cast<const float4*>(rowp + 16 + 8 * mm + 4 * h);
    x[0] = ev.x; x[1] = od.x; x[2] = ev.y; x[3] = od.y; x[4] = ev.z; x[5] = od.z; x[6] = ev.w; x[7] = od.w;
}

}  // namespace

// One staged chunk = 32 rows x 32 k of up to two fp32 operands (A: kbar / qbar, B: ak / aq), loaded COALESCED (8 lanes per
// 128-byte row segment, 8 rows per wave instruction), parked in registers while in flight (two chunks ahead), written to
// the wave's own LDS buffer one chunk ahead and read from there in MFMA operand order.  Everything is wave-local: no barrier.
struct S2Stage {
    float4 a[4], b[4];
};

template <int D, typename Tag>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void pooled_scores2_kernel(Score2Args a) {
    constexpr int NC = D / 32;   // 32-k chunks
    constexpr int NM = D / 16;   // k steps of the 2-byte MFMAs
    __shared__ __attribute__((aligned(16))) float stage[2][2][S2_ARR];   // [buffer][operand][row][S2_LD]
    __shared__ float2 rown[32];                         // (||qbar_i||, ||aq_i||) of the tile's rows
    __shared__ unsigned fix_idx[RSA_S2_FIXCAP];          // undecided elements: (local row << 16) | global column
    __shared__ float fix_s[RSA_S2_FIXCAP];               // ... and their s

    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int per = (int)(gridDim.x >> 3);              // the grid is a multiple of 8
    const int wid = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wid >= a.n_items) return;
    const int per_head = a.nti * a.JS;
    const int bh = wid / per_head, rem = wid % per_head;
    const int jr = rem / a.nti, ti = rem % a.nti;       // consecutive items: the i tiles of one (head, j range)
    const int i0 = ti * 32;
    const int sub0 = (int)((long)jr * a.ntj / a.JS), sub1 = (int)((long)(jr + 1) * a.ntj / a.JS);
    if (sub0 >= sub1) return;
    const int NBv = a.NBv;
    const int b_ = bh / a.H, hd = bh % a.H;
    const unsigned short* ktbase = a.ktxt + (long)b_ * a.ksb + (long)hd * a.ksh + (long)NBv * RSA_BLOCK * a.kss;

    // ---- the loader's lane roles: fp32 operands: row 8 i + (lane >> 3), k = 4 (lane & 7) .. + 3; text tokens (2-byte rows):
    // row 16 i + (lane >> 2), k = 8 (lane & 3) .. + 7
    const int lrow = lane >> 3, lkq = lane & 7;
    const int trow = lane >> 2, tkq = lane & 3;
    // sub = -1: the i tile itself (qbar, aq); 0 .. ntj0 - 1: visual columns (kbar, ak); beyond: text-token columns (K rows)
    auto issue = [&](S2Stage& st, int sub, int c) {
        if (sub < a.ntj0) {
            const float* pa = sub < 0 ? a.qbar : a.kbar;
            const float* pb = sub < 0 ? a.aq : a.ak;
            const int base = sub < 0 ? i0 : sub * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long off = ((long)bh * NBv + min(base + 8 * i + lrow, NBv - 1)) * D + 32 * c + 4 * lkq;
                st.a[i] = *reinterpret_cast<const float4*>(pa + off);
                st.b[i] = *reinterpret_cast<const float4*>(pb + off);
            }
        } else {
            const int base = (sub - a.ntj0) * 32;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = min(base + 16 * i + trow, a.n_txt - 1);
                const uint4 raw = *reinterpret_cast<const uint4*>(ktbase + (long)col * a.kss + 32 * c + 8 * tkq);
                st.a[i] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
            }
        }
    };
    // registers -> the wave's LDS buffer `buf` ([even k | odd k] per row)
    auto park = [&](const S2Stage& st, int sub, int buf) {
        if (sub < a.ntj0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* ra = &stage[buf][0][(8 * i + lrow) * S2_LD + 2 * lkq];
                float* rb = &stage[buf][1][(8 * i + lrow) * S2_LD + 2 * lkq];
                *reinterpret_cast<float2*>(ra) = make_float2(st.a[i].x, st.a[i].z);
                *reinterpret_cast<float2*>(ra + 16) = make_float2(st.a[i].y, st.a[i].w);
                *reinterpret_cast<float2*>(rb) = make_float2(st.b[i].x, st.b[i].z);
                *reinterpret_cast<float2*>(rb + 16) = make_float2(st.b[i].y, st.b[i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned w[4] = {__float_as_uint(st.a[i].x), __float_as_uint(st.a[i].y), __float_as_uint(st.a[i].z), __float_as_uint(st.a[i].w)};
                float* ra = &stage[buf][0][(16 * i + trow) * S2_LD + 4 * tkq];
                *reinterpret_cast<float4*>(ra) = make_float4(rsa_to_f32<Tag>((unsigned short)(w[0] & 0xFFFF)), rsa_to_f32<Tag>((unsigned short)(w[1] & 0xFFFF)),
                                                              rsa_to_f32<Tag>((unsigned short)(w[2] & 0xFFFF)), rsa_to_f32<Tag>((unsigned short)(w[3] & 0xFFFF)));
                *reinterpret_cast<float4*>(ra + 16) = make_float4(rsa_to_f32<Tag>((unsigned short)(w[0] >> 16)), rsa_to_f32<Tag>((unsigned short)(w[1] >> 16)),
                                                                   rsa_to_f32<Tag>((unsigned short)(w[2] >> 16)), rsa_to_f32<Tag>((unsigned short)(w[3] >> 16)));
            }
        }
    };
    // the flat sequence of chunks this wave streams: NC chunks of the i tile, then NC per j sub-tile
    const int n_seq = (1 + (sub1 - sub0)) * NC;
    auto sub_of = [&](int g) -> int { const int u = g / NC; return u == 0 ? -1 : sub0 + u - 1; };
    // (hipcc's scheduler moves independent loads freely, and vmcnt waits count in ISSUE order: a park that had to wait for loads
    // issued after its own chunk would drain the whole prefetch -- the scheduling barriers pin park | issue | reads in place)
    auto issue_seq = [&](S2Stage& st, int g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g < n_seq) issue(st, sub_of(g), g % NC);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto park_seq = [&](const S2Stage& st, int g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g < n_seq) park(st, sub_of(g), g & 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    S2Stage st0, st1;            // chunk g + 1 sits in st[(g + 1) & 1] while chunk g is multiplied; chunk g + 2 is in flight behind it
    issue_seq(st0, 0);
    issue_seq(st1, 1);
    park_seq(st0, 0);
    issue_seq(st0, 2);

    // ------------------------------------------------------------------ i side -> registers (chunks 0 .. NC - 1 of the sequence)
    float qA[D / 2];
    s2_bf16x8 qhi[NM], qlo[NM], ahi[NM], alo[NM];
    {
        float nq2 = 0.0f, na2 = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (c & 1) { park_seq(st0, c + 1); issue_seq(st0, c + 3); } else { park_seq(st1, c + 1); issue_seq(st1, c + 3); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const float* qp = &stage[c & 1][0][r * S2_LD];
            const float* ap = &stage[c & 1][1][r * S2_LD];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {       // s chain A operands: qbar[row][32c + 2t + h], t = 0..15
                const float4 f = *reinterpret_cast<const float4*>(qp + 16 * h + 4 * t4);
                qA[16 * c + 4 * t4 + 0] = f.x; qA[16 * c + 4 * t4 + 1] = f.y; qA[16 * c + 4 * t4 + 2] = f.z; qA[16 * c + 4 * t4 + 3] = f.w;
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                float x[8], y[8];
                image8(qp, mm, h, x);
                image8(ap, mm, h, y);
                split8(x, qhi[2 * c + mm], qlo[2 * c + mm]);
                split8(y, ahi[2 * c + mm], alo[2 * c + mm]);
#pragma unroll
                for (int e = 0; e < 8; ++e) { nq2 += x[e] * x[e]; na2 += y[e] * y[e]; }
            }
        }
        nq2 += __shfl_xor(nq2, 32, 64);
        na2 += __shfl_xor(na2, 32, 64);
        // ||qbar_i||, ||aq_i|| (a vanishing or non-finite norm becomes NaN: every comparison against the bound is then false
        // and the element takes the exact path)
        const bool bad = !(fminf(nq2, na2) >= 1e-30f) || !(fmaxf(nq2, na2) < 1e30f);
        if (!hb) rown[r] = bad ? make_float2(NAN, NAN) : make_float2(__builtin_sqrtf(nq2), __builtin_sqrtf(na2));
    }

    int nfix = 0;   // queued undecided elements (wave-uniform)
    // the queued elements, 64 at a time: both error terms as the contract's k-ordered fmaf chains, the byte from them
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int idx = lane; idx < ((a.form & 1) ? 0 : nfix); idx += 64) {
            const unsigned u = fix_idx[idx];
            const int i = i0 + (int)(u >> 16), j = (int)(u & 0xFFFF);
            const float4* pa = reinterpret_cast<const float4*>(a.aq + ((long)bh * NBv + i) * D);
            const float4* pq = reinterpret_cast<const float4*>(a.qbar + ((long)bh * NBv + i) * D);
            const float4* pk = reinterpret_cast<const float4*>(a.kbar + ((long)bh * NBv + j) * D);
            const float4* pb = reinterpret_cast<const float4*>(a.ak + ((long)bh * NBv + j) * D);
            float eq = 0.0f, ek = 0.0f;
            for (int d4 = 0; d4 < D / 4; ++d4) {
                const float4 a4 = pa[d4], q4 = pq[d4], k4 = pk[d4], b4 = pb[d4];
                eq = __builtin_fmaf(a4.x, k4.x, eq); eq = __builtin_fmaf(a4.y, k4.y, eq);
                eq = __builtin_fmaf(a4.z, k4.z, eq); eq = __builtin_fmaf(a4.w, k4.w, eq);
                ek = __builtin_fmaf(q4.x, b4.x, ek); ek = __builtin_fmaf(q4.y, b4.y, ek);
                ek = __builtin_fmaf(q4.z, b4.z, ek); ek = __builtin_fmaf(q4.w, b4.w, ek);
            }
            a.unrel[((long)bh * NBv + i) * NBv + j] = !(fabsf(fix_s[idx]) > (fabsf(eq) + fabsf(ek)));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        nfix = 0;
    };

    for (int sub = sub0; sub < sub1; ++sub) {
        const bool text = sub >= a.ntj0;     // wave-uniform
        const int g0 = (1 + sub - sub0) * NC;  // this sub-tile's first chunk in the sequence (a multiple of NC: even)
        s2_f32x16 s, eq, ek;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.0f; eq[i] = 0.0f; ek[i] = 0.0f; }
        float nk2 = 0.0f, nb2 = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (c & 1) { park_seq(st0, g0 + c + 1); issue_seq(st0, g0 + c + 3); } else { park_seq(st1, g0 + c + 1); issue_seq(st1, g0 + c + 3); }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const float* kp = &stage[c & 1][0][r * S2_LD];
            const float* bp = &stage[c & 1][1][r * S2_LD];
            // ---- s: the contract's chain, k ascending (B = kbar[col][32c + 2t + h])
            if (!(a.form & 8))
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const float4 f = *reinterpret_cast<const float4*>(kp + 16 * h + 4 * t4);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qA[16 * c + 4 * t4 + 0], f.x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qA[16 * c + 4 * t4 + 1], f.y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qA[16 * c + 4 * t4 + 2], f.z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qA[16 * c + 4 * t4 + 3], f.w, s, 0, 0, 0);
            }
            if (!text && !(a.form & 4)) {
                // ---- the two error terms from the bf16 pieces
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int m = 2 * c + mm;
                    float x[8], y[8];
                    image8(kp, mm, h, x);
                    image8(bp, mm, h, y);
                    s2_bf16x8 khi, klo, bhi, blo;
                    split8(x, khi, klo);
                    split8(y, bhi, blo);
                    eq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], khi, eq, 0, 0, 0);
                    ek = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qhi[m], bhi, ek, 0, 0, 0);
                    eq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], klo, eq, 0, 0, 0);
                    ek = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qhi[m], blo, ek, 0, 0, 0);
                    eq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[m], khi, eq, 0, 0, 0);
                    ek = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qlo[m], bhi, ek, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { nk2 += x[e] * x[e]; nb2 += y[e] * y[e]; }
                }
            }
        }
        // ---- outputs: accumulator element e of lane (r, h) = row (e & 3) + 8 (e >> 2) + 4 h, column r of the sub-tile
        if (text) {
            const int j = (sub - a.ntj0) * 32 + r;
            if (j < a.n_txt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (i < NBv) a.scores[((long)bh * NBv + i) * a.NS + NBv + j] = s[e];
                }
            }
            continue;
        }
        if (a.form & 2) { if (lane == 0 && s[0] + eq[0] + ek[0] + nk2 + nb2 == 12345.0f) a.scores[0] = 1.0f; continue; }
        nk2 += __shfl_xor(nk2, 32, 64);
        nb2 += __shfl_xor(nb2, 32, 64);
        const int j = sub * 32 + r;
        const bool jok = j < NBv;
        if (nfix > RSA_S2_FIXCAP - 1024) flush();        // (wave-uniform; the one call site inside the loop)
        // ||kbar_j||, ||ak_j|| x 1.01 * 2^-12 (the norms' own rounding inside the 1 %); NaN as above
        const bool cbad = !(fminf(nk2, nb2) >= 1e-30f) || !(fmaxf(nk2, nb2) < 1e30f);
        const float ck = cbad ? NAN : 2.4658203125e-4f * __builtin_sqrtf(nk2), cb = cbad ? NAN : 2.4658203125e-4f * __builtin_sqrtf(nb2);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int il = (e & 3) + 8 * (e >> 2) + 4 * h;
            const int i = i0 + il;
            const bool ok = jok && i < NBv;
            const float sv = s[e];
            const float t = fabsf(eq[e]) + fabsf(ek[e]);
            const float mg = fabsf(sv) - t;
            const float2 rn = rown[il];
            // E = c_E (||aq_i|| ||kbar_j|| + ||qbar_i|| ||ak_j||) + the rounding of t itself (2^-21 t)
            const float E = __builtin_fmaf(4.76837158203125e-7f, t, __builtin_fmaf(rn.x, cb, rn.y * ck));
            const bool decided = fabsf(mg) > E;          // false for a NaN bound, a non-finite t, ...
            if (ok) {
                a.scores[((long)bh * NBv + i) * a.NS + j] = sv;
                if (decided) a.unrel[((long)bh * NBv + i) * NBv + j] = !(fabsf(sv) > t);
            }
            const bool und = ok && !decided;
            const unsigned long long mk = __ballot(und);
            if (und) {
                const int pos = nfix + __popcll(mk & ((1ull << lane) - 1ull));
                fix_idx[pos] = ((unsigned)il << 16) | (unsigned)j;
                fix_s[pos] = sv;
            }
            nfix += __popcll(mk);
        }
    }
    if (nfix > 0) flush();
}

static int g_k2_slots = 0;
int g_k2_form = 0;    // timing-only forms of the kernel (tuning key "k2_form"; outputs are garbage unless 0)
int rsa_launch_pooled_scores2(Score2Args a, int D, int dtype, hipStream_t s) {
    if (a.NBv > 65535) return RSA_ERR_UNSUPPORTED;
    a.nti = (a.NBv + 31) / 32;
    a.ntj0 = a.nti;
    a.ntj = a.ntj0 + (a.n_txt + 31) / 32;
    if (g_k2_slots == 0) {   // one wave per SIMD: 4 per CU
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        g_k2_slots = 4 * cus;
    }
    // j ranges per i tile: whole rounds of `slots` equal items; at least two sub-tiles per item (the i side costs a sub-tile's
    // worth of loads), fewer ranges on ties
    const long base = (long)a.BH * a.nti;
    int best = 1;
    double best_eff = -1.0;
    for (int js = 1; js <= 16 && js * 2 <= a.ntj + 1; ++js) {
        const long W = base * js;
        const long rounds = (W + g_k2_slots - 1) / g_k2_slots;
        const double eff = (double)W / ((double)rounds * g_k2_slots) - 0.004 * js;
        if (eff > best_eff + 1e-9) { best_eff = eff; best = js; }
    }
    a.JS = best;
    a.form = g_k2_form;
    a.n_items = (int)(base * a.JS);
    const dim3 grid((unsigned)((a.n_items + 7) / 8 * 8));
    if (D == 128) {
        if (dtype == RSA_BF16) pooled_scores2_kernel<128, bf16_tag><<<grid, 64, 0, s>>>(a);
        else pooled_scores2_kernel<128, fp16_tag><<<grid, 64, 0, s>>>(a);
    } else {
        if (dtype == RSA_BF16) pooled_scores2_kernel<64, bf16_tag><<<grid, 64, 0, s>>>(a);
        else pooled_scores2_kernel<64, fp16_tag><<<grid, 64, 0, s>>>(a);
    }
    return rsa_launch_status();
}
