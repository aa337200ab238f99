// K5, 64-rows-per-wave form: block-sparse flash attention forward for gfx950 with the rectification
// epilogue fused -- same semantics, work mapping, per-row plan and epilogue as bsfwd_kernel (rsa_attn_kernel.hip; the
// reference kernel it follows: rectified_hunyuan_attn.py:15-105), different occupancy model:
//
//   one workgroup = 2 waves = one 128-row query block, wave w owns rows 64w .. 64w+63 as two 32-row halves and the WHOLE
//   512-entry register file of its SIMD (one wave per SIMD; two workgroups per CU).  O (128 registers) and the Q
//   fragments (64) live in the accumulator file, owned by the asm statements; S, P, the -m blocks and the LDS operand rings
//   in the arch VGPRs.  Every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) read from LDS feeds
//   TWO MFMAs, one per row half: 24 LDS operand reads per 32 MFMAs instead of per 16.
//
// With one wave per SIMD nothing hides an instruction that sits between two MFMA streams, so the steady state is ONE
// generated asm statement (gen_k5_block64.py -> rsa_attn_block64.h, which documents register map, LDS rings and schedule):
// per kept 128-key block four 32-key sub-steps, each = vmcnt(16) + barrier + a hand-placed block of 32 MFMAs with the
// softmax, the LDS operand reads, the wave's 8 LDS-DMA pieces (half-tiles u+3 of V and u+4 of K: two to three sub-steps of
// flight) and the scalar bookkeeping (list entry, DMA address walkers) in the MFMA shadows, and the deferred-rescale test,
// whose rare body sits out of line inside the statement.  C++ keeps the prologue, the epilogue and the sub-steps the loop
// does not take: kept blocks whose scores need the boundary mask or whose successor's rows must be clamped (the last one
// or two of a list, the diagonal blocks of a causal call) run block by block, staged from here.
//
// LDS = [K ring: 4 x 8 KiB | V ring: 4 x 8 KiB | list (u16)]: half-tile x (32 keys) of the walk sits in slot x & 3.
//
// (The text above describes head dim 128, two waves.)  Round 6 -- each described where it is implemented below:
//   * head dim 64 (template parameter D; CogVideoX): 8 + 8 MFMAs per sub-step, 4-KiB half-tiles of four pieces, O in a[0:63], Q in
//     a[64:95]; the streams come from the same generator (RSA_K5V_* beside RSA_K5W_*);
//   * NW = 4: a 256-row tile of a DENSE call on ONE K/V ring (each wave stages every second piece);
//   * the optimistic static softmax reference (bf16): a second body of the loop statement without row maxima and rescale test, the
//     walk checked afterwards and redone through the online body if anything overflowed.
#include "rsa_attn.h"
#include <atomic>
#include "rsa_attn_block64.h"
#ifdef RSA_K5_FORMS
#include "rsa_attn_block64_forms.h"
#endif

// One asm site serves both head dims: RSA_K5W_* (128) / RSA_K5V_* (64) of rsa_attn_block64.h, same arch-register map, different
// accumulator map (O a[0:32 DT - 1], Q behind it) -- hence the clobber lists by prefix.
#define RSA_K5W_CL_TO RSA_K5W_CLOBBER_TMP, RSA_K5W_CLOBBER_O
#define RSA_K5V_CL_TO RSA_K5V_CLOBBER_TMP, RSA_K5V_CLOBBER_O
#define RSA_K5W_CL_LOOP RSA_K5W_CLOBBER_TMP, RSA_K5W_CLOBBER_O, RSA_K5W_CLOBBER_LOOP
#define RSA_K5V_CL_LOOP RSA_K5V_CLOBBER_TMP, RSA_K5V_CLOBBER_O, RSA_K5V_CLOBBER_LOOP
#define RSA_K5_ASM(D_, BODY, OPS, CL, ...) \
    do { \
        if constexpr ((D_) == 128) asm volatile(RSA_K5W_##BODY RSA_K5W_##OPS : RSA_K5W_##CL __VA_ARGS__); \
        else asm volatile(RSA_K5V_##BODY RSA_K5V_##OPS : RSA_K5V_##CL __VA_ARGS__); \
    } while (0)

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// block U without LDS-DMA (the C++-driven sub-steps)
template <typename Tag, int U, int D>
__device__ __forceinline__ void k5w_block(f32x16 (&SA)[2], f32x16 (&SB)[2], const f32x16 (&nm)[2], float (&l)[2], float (&mx)[2],
                                          const i32x8& ka, const i32x8& va) {
    if constexpr (std::is_same<Tag, bf16_tag>::value) {
        if constexpr (U == 0) RSA_K5_ASM(D, BLOCK_BF16_U0, OPS, CL_TO, , "memory");
        else if constexpr (U == 1) RSA_K5_ASM(D, BLOCK_BF16_U1, OPS, CL_TO, , "memory");
        else if constexpr (U == 2) RSA_K5_ASM(D, BLOCK_BF16_U2, OPS, CL_TO, , "memory");
        else RSA_K5_ASM(D, BLOCK_BF16_U3, OPS, CL_TO, , "memory");
    } else {
        if constexpr (U == 0) RSA_K5_ASM(D, BLOCK_F16_U0, OPS, CL_TO, , "memory");
        else if constexpr (U == 1) RSA_K5_ASM(D, BLOCK_F16_U1, OPS, CL_TO, , "memory");
        else if constexpr (U == 2) RSA_K5_ASM(D, BLOCK_F16_U2, OPS, CL_TO, , "memory");
        else RSA_K5_ASM(D, BLOCK_F16_U3, OPS, CL_TO, , "memory");
    }
}

// one Q fragment (4 registers, pinned to v[96:99]) -> its place in the accumulator file
template <int Hh, int KSI, int D>
__device__ __forceinline__ void k5w_qwrite(const s16x8& f) {
#define RSA_QW(HH, KK) do { if constexpr (D == 128) asm volatile(RSA_K5W_QWRITE_H##HH##_K##KK :: "{v[96:99]}"(f) : RSA_K5W_CLOBBER_Q); \
                            else asm volatile(RSA_K5V_QWRITE_H##HH##_K##KK :: "{v[96:99]}"(f) : RSA_K5V_CLOBBER_Q); } while (0)
    if constexpr (Hh == 0) {
        if constexpr (KSI == 0) RSA_QW(0, 0); else if constexpr (KSI == 1) RSA_QW(0, 1); else if constexpr (KSI == 2) RSA_QW(0, 2);
        else if constexpr (KSI == 3) RSA_QW(0, 3); else if constexpr (KSI == 4) RSA_QW(0, 4); else if constexpr (KSI == 5) RSA_QW(0, 5);
        else if constexpr (KSI == 6) RSA_QW(0, 6); else RSA_QW(0, 7);
    } else {
        if constexpr (KSI == 0) RSA_QW(1, 0); else if constexpr (KSI == 1) RSA_QW(1, 1); else if constexpr (KSI == 2) RSA_QW(1, 2);
        else if constexpr (KSI == 3) RSA_QW(1, 3); else if constexpr (KSI == 4) RSA_QW(1, 4); else if constexpr (KSI == 5) RSA_QW(1, 5);
        else if constexpr (KSI == 6) RSA_QW(1, 6); else RSA_QW(1, 7);
    }
#undef RSA_QW
}
// one 32 x 32 tile of O (rows of half Hh, d = 32 DTI .. +31) out of the accumulator file
template <int Hh, int DTI, int D>
__device__ __forceinline__ f32x16 k5w_oread() {
    f32x16 t;
#define RSA_OR(HH, DD) do { if constexpr (D == 128) asm volatile(RSA_K5W_OREAD_H##HH##_D##DD : "={v[96:111]}"(t)); \
                            else asm volatile(RSA_K5V_OREAD_H##HH##_D##DD : "={v[96:111]}"(t)); } while (0)
    if constexpr (Hh == 0) {
        if constexpr (DTI == 0) RSA_OR(0, 0); else if constexpr (DTI == 1) RSA_OR(0, 1); else if constexpr (DTI == 2) RSA_OR(0, 2); else RSA_OR(0, 3);
    } else {
        if constexpr (DTI == 0) RSA_OR(1, 0); else if constexpr (DTI == 1) RSA_OR(1, 1); else if constexpr (DTI == 2) RSA_OR(1, 2); else RSA_OR(1, 3);
    }
#undef RSA_OR
    return t;
}

// blockIdx -> (batch*head, query block, key-range part of a split text block); false = padding workgroup
// Work mapping.  Sparse query blocks: workgroup v of a head's NBp (a multiple of 8) goes to XCD v & 7, which takes the
// (v & 7)-th contiguous eighth of the head's query blocks.  The dense text-row blocks come FIRST when each is one long walk
// over every key block (no split-KV buffer: the longest work first), and LAST when they are split into pieces shorter than a
// sparse walk (tsplit > 1): with aligned starts the launch advances in generations of 8 x 64 workgroups, and the short
// pieces then fill the slots the last, partial generation leaves idle instead of adding a generation of their own.
// Tail split (a.tail_n > 0; sparse blocks first): the sparse blocks from index tail_first on -- the last, partial generation --
// are walked by tail_p workgroups each (piece i of the region = block tail_first + i / tail_p, part i % tail_p of its kept
// list), which together fill the slots that generation would leave idle; tail >= 0 tells the caller (the piece's index).
__device__ __forceinline__ bool k5w_map(const AttnArgs& a, int work, int& bh, int& qblk, int& tsp, int& tail) {
    tsp = 0;
    tail = -1;
    const int n_sparse = a.BH * a.NBp;
    const bool heavy_last = a.heavy_last != 0;
    int wh = heavy_last ? work - n_sparse : work;                 // index among the text-row pieces
    int v = heavy_last ? work : work - a.n_heavy_pad;             // index among the sparse blocks
    bool text = heavy_last ? work >= n_sparse : work < a.n_heavy_pad;
    if (a.tail_n > 0) {
        const int tail_end = a.tail_first + a.tail_n * a.tail_p;
        text = work >= tail_end;
        wh = work - tail_end;
        if (work >= a.tail_first && !text) {
            tail = work - a.tail_first;
            v = a.tail_first + tail / a.tail_p;
            tsp = tail % a.tail_p;
        }
    }
    if (text) {
        const int ntq = a.NQB - a.NBv;
        const int per_bh = ntq * a.tsplit;      // text blocks x key-range splits (tsplit = 1: no split)
        if (ntq <= 0 || wh >= a.BH * per_bh) return false;
        bh = wh / per_bh;
        const int rem = wh % per_bh;
        qblk = a.NBv + rem / a.tsplit;
        tsp = rem % a.tsplit;
    } else {
        bh = v / a.NBp;
        const int j = v % a.NBp;
        const int chunk = a.NBp >> 3;
        qblk = (j & 7) * chunk + (j >> 3);
        if (qblk >= a.NBv) return false;
    }
    return true;
}

// WIDE: 16-byte output stores after a permlane32_swap regroup (needs 16-byte aligned output rows), else 8-byte stores.
// XF (A/B build only, -DRSA_K5_FORMS): one of the loop forms of rsa_attn_block64_forms.h instead of the product's loop.
// NW (round 6): waves per workgroup.  2 = one 128-row query block (every sparse call: the mask's granularity).  4 = a 256-ROW
// tile of a DENSE call (rsa_dense_fwd / _causal_: all rows walk the same keys): four waves, one per SIMD, ONE workgroup per CU on ONE
// K/V ring -- every half-tile is staged once per 256 rows instead of once per 128, each wave issues 4 LDS-DMA pieces per sub-step
// instead of 8 (what that buys at most: form x15 of profiles/r06_k5_forms.txt).  The loop statement is the same generator's with
// every second piece dropped (gen_k5_block64.py, RSA_K5W_LOOP_*_R256); `qblk` then counts 256-row tiles (the host sets NQB so).
// D (round 6): head dim 128 or 64 (CogVideoX).  At 64 a sub-step is 8 + 8 MFMAs against the same softmax, a half-tile 4 KiB = four
// LDS-DMA pieces of 8 rows (two per wave); same register map in the arch file, O in a[0:63], Q in a[64:95].
template <typename Tag, bool WIDE, int XF = 0, int NW = 2, int D = 128>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 1))) void bsfwd64_kernel(AttnArgs a) {
    constexpr int RW = 64 * NW;             // query rows per workgroup: NW waves x 64 rows
    constexpr int NPIECE = 32 * D * 2 / 1024;   // 1-KiB pieces of a 32-key half-tile: 8 (4 rows each) / 4 (8 rows each)
    constexpr int NPW = NPIECE / NW;        // ... of them each wave stages
    constexpr int RPP = 32 / NPIECE;        // key rows per piece
    constexpr int LPR = 64 / RPP;           // lanes (16-byte chunks) per key row
    static_assert(NW == 2 || NW == 4, "two or four waves");
    static_assert(D == 128 || (D == 64 && XF == 0), "head dim 64: the product forms only");
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    constexpr int HALF = 32 * D * 2;        // bytes of a 32-key half-tile
    constexpr int VRING = 4 * HALF;
    using E = Elem<Tag>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short* lds_list = reinterpret_cast<unsigned short*>(lds + 8 * HALF);
    __shared__ int redo_flag;               // optimistic static reference (below): a wave whose walk overflowed asks the workgroup for a second pass

    const GsyncTicket gs_tk = rsa_gsync_announce(a.gsync, a.gsync_gen);   // aligned starts (rsa_attn.h)

    // ---------------- work mapping: dense text-row blocks first, then the sparse blocks chunked per XCD ----------------
    int bh, qblk, tsp, tail;
    if (!k5w_map(a, blockIdx.x, bh, qblk, tsp, tail)) return;
    const int b = bh / a.H, h = bh % a.H;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int grow[2];
    grow[0] = qblk * RW + 64 * wv + r;
    grow[1] = grow[0] + 32;

    // ---------------- per-row plan (per row half) ----------------
    int lo_r[2] = {0, 0}, hi_r[2] = {0, 0};
    int n_items, first_blk = 0, lo_max, hi_min, hi_max;
    const int32_t* list = nullptr;
    if (a.mode == MODE_SPARSE) {
        if (qblk < a.NBv) {
            const long rowi = (long)bh * a.NBv + qblk;
            list = a.cols + rowi * a.NB_total;
            n_items = a.counts[rowi];
            if (tail >= 0) {   // this workgroup's part of the kept list (tail split)
                const int per = (n_items + a.tail_p - 1) / a.tail_p, first = tsp * per;
                const int left = n_items - first;
                list += first;
                n_items = left < 0 ? 0 : (left < per ? left : per);
            }
            lo_max = 0; hi_min = hi_max = a.kv_valid;
            hi_r[0] = hi_r[1] = a.kv_valid;
        } else {
            n_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
            if (a.tsplit > 1) {   // split-KV: this workgroup's slice of the key blocks
                first_blk = tsp * a.tper;
                n_items = n_items - first_blk < a.tper ? n_items - first_blk : a.tper;
                if (n_items < 0) n_items = 0;
            }
            lo_max = 0; hi_min = hi_max = a.kv_text_valid;
            hi_r[0] = hi_r[1] = a.kv_text_valid;
        }
    } else {
        // dense mode: one or two (query rows, key rows) segments; causal = bottom-right aligned inside a segment (see
        // rsa_attn_kernel.hip for the conventions)
        const int row0 = qblk * RW, row1 = row0 + RW;
        auto seg_hi = [&](int row) -> int {
            const bool s1 = row >= a.q_split;
            const int lo = s1 ? a.kv_split : 0, hi = s1 ? a.Sk : a.kv_split;
            if (!a.causal) return hi;
            const int rows = s1 ? a.Sq - a.q_split : a.q_split, rin = row - (s1 ? a.q_split : 0);
            const int lim = lo + rin + 1 + ((hi - lo) - rows);
            return lim < lo ? lo : (lim < hi ? lim : hi);
        };
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            lo_r[x] = grow[x] < a.q_split ? 0 : a.kv_split;
            hi_r[x] = seg_hi(grow[x] < a.Sq ? grow[x] : a.Sq - 1);
        }
        int lo_min;
        const int rlast = (row1 <= a.Sq ? row1 : a.Sq) - 1;
        if (row1 <= a.q_split) { lo_min = 0; lo_max = 0; }
        else if (row0 >= a.q_split) { lo_min = lo_max = a.kv_split; }
        else { lo_min = 0; lo_max = a.kv_split; }
        hi_min = seg_hi(row0);
        hi_max = seg_hi(rlast);
        if (row0 < a.q_split && rlast >= a.q_split) {
            const int h0 = seg_hi(a.q_split - 1), h1 = seg_hi(a.q_split);
            hi_min = hi_min < h1 ? hi_min : h1;
            hi_max = hi_max > h0 ? hi_max : h0;
        }
        first_blk = lo_min / RSA_BLOCK;
        n_items = (hi_max + RSA_BLOCK - 1) / RSA_BLOCK - first_blk;
        if (hi_max <= lo_min) n_items = 0;
    }
    n_items = __builtin_amdgcn_readfirstlane(n_items);
    // the kept list (sparse visual blocks) or the plain block range (text rows, dense mode) as u16 entries in LDS: the walk
    // reads its block indices from there in every mode (one entry of slack: the loop looks two blocks ahead)
    for (int i = t; i < n_items + 2; i += 64 * NW)
        lds_list[i] = (unsigned short)(i < n_items ? (list ? list[i] : first_blk + i) : 0);
    if (t == 0) redo_flag = 0;
    __syncthreads();
    auto blk_of = [&](int item) -> int { return __builtin_amdgcn_readfirstlane((int)lds_list[item]); };
    const int n_sub = 4 * n_items;            // 32-key sub-steps: four per kept block (scores past the row's range are masked)
    const int kv_limit = hi_max < a.Sk ? hi_max : a.Sk;
    auto key_of = [&](int x) -> int { return blk_of(x >> 2) * RSA_BLOCK + (x & 3) * 32; };   // first key of half-tile x

    // ---------------- Q fragments (B operand) -> accumulator file (O is zeroed at the head of a pass) ----------------
    {
        auto load_frag = [&](int x, int ks) -> s16x8 {
            const unsigned short* qp = a.q + (long)b * a.qsb + (long)h * a.qsh + (long)grow[x] * a.qss + 8 * hh;
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (grow[x] < a.Sq) raw = *reinterpret_cast<const uint4*>(qp + 16 * ks);
            const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = rsa_to_f32<Tag>((unsigned short)(w4[e] & 0xFFFF)) * a.qk_scale;
                f[2 * e + 1] = rsa_to_f32<Tag>((unsigned short)(w4[e] >> 16)) * a.qk_scale;
            }
            return E::cvt8(f);
        };
#define RSA_QF(HH, KK) do { if constexpr (KK < KS) k5w_qwrite<HH, KK, D>(load_frag(HH, KK)); } while (0)
        RSA_QF(0, 0); RSA_QF(0, 1); RSA_QF(0, 2); RSA_QF(0, 3); RSA_QF(0, 4); RSA_QF(0, 5); RSA_QF(0, 6); RSA_QF(0, 7);
        RSA_QF(1, 0); RSA_QF(1, 1); RSA_QF(1, 2); RSA_QF(1, 3); RSA_QF(1, 4); RSA_QF(1, 5); RSA_QF(1, 6); RSA_QF(1, 7);
#undef RSA_QF
    }

    // ---------------- LDS-DMA staging ----------------
    // (hipcc does 64-bit address arithmetic on the vector unit even when it is wave-uniform and then hands the VGPR pair to an
    // "s" operand as is: every base that reaches the staging asm is made scalar explicitly, half by half)
    auto uni64 = [](const unsigned char* p) -> const unsigned char* {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)p);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)p >> 32));
        return reinterpret_cast<const unsigned char*>(((unsigned long)hi << 32) | lo);
    };
    const unsigned char* kbase = uni64(reinterpret_cast<const unsigned char*>(a.k + (long)b * a.ksb + (long)h * a.ksh));
    const unsigned char* vbase = uni64(reinterpret_cast<const unsigned char*>(a.v + (long)b * a.vsb + (long)h * a.vsh));
    // A 32-key half-tile = NPIECE one-KiB pieces of RPP rows (8 x 4 rows at head dim 128, 4 x 8 rows at 64); wave w moves pieces
    // NW j + w, j = 0 .. NPW - 1.  The XOR swizzle of a row's source chunk (rsa_attn.h::tile_off) depends on the row: at head dim 128
    // on (row & 3) and ((row >> 2) & 3) = piece & 3 -- two per-lane offsets (even / odd j) with two waves, one with four --, at head
    // dim 64 on (row >> 1) & 7, the same for both of the wave's pieces; the piece walks a scalar base.
    const int rsub = lane / LPR, cl = lane % LPR;
    const int rowl = RPP * wv + rsub;
    const int gsw0 = D == 128 ? (cl ^ ((rsub << 2) | (wv & 3))) : (cl ^ ((4 * wv + (rsub >> 1)) & 7));
    const int gsw1 = D == 128 ? (cl ^ ((rsub << 2) | ((NW + wv) & 3))) : gsw0;
    const unsigned krow = (unsigned)(a.kss * 2), vrow = (unsigned)(a.vss * 2);   // bytes per key row (< 4 GiB)
    // lane offsets of the wave's pieces of a half-tile, for the loop's staging.  Head dim 128, two waves: piece j = rows 8j + rowl in
    // register j; odd pieces carry the instruction offset 2048 (their LDS destination), which also moves the source: taken out here.
    // Four waves: the loop issues only its even slots (registers 0 and 2): piece j' = rows 16 j' + rowl in register 2 j'.  Head dim
    // 64: two pieces, rows 16 j + rowl in registers 0 and 1 (the second with the instruction offset 2048).
    u32x4 vok, vov;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (D == 64) {      // (four waves: one piece each, register 0)
            vok[j] = j < 2 ? (16 * j + rowl) * krow + gsw0 * 16 - (j ? 2048u : 0u) : 0u;
            vov[j] = j < 2 ? (16 * j + rowl) * vrow + gsw0 * 16 - (j ? 2048u : 0u) : 0u;
        } else if constexpr (NW == 2) {
            vok[j] = (8 * j + rowl) * krow + ((j & 1) ? gsw1 : gsw0) * 16 - ((j & 1) ? 2048u : 0u);
            vov[j] = (8 * j + rowl) * vrow + ((j & 1) ? gsw1 : gsw0) * 16 - ((j & 1) ? 2048u : 0u);
        } else {
            vok[j] = (16 * (j >> 1) + rowl) * krow + gsw0 * 16;
            vov[j] = (16 * (j >> 1) + rowl) * vrow + gsw0 * 16;
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // staging from C++ (prologue, boundary blocks): the 32-key half-tile starting at key `key_first` -> LDS byte offset
    // `lds_off`; rows past the last valid key are clamped to it (their scores are masked)
    auto dma_half = [&](int is_v, int key_first, unsigned lds_off) {
        const unsigned ld0 = lds_base + lds_off + wv * 1024;
        const unsigned char* base = is_v ? vbase : kbase;
        const unsigned rowb = is_v ? vrow : krow;
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            int krow_ = key_first + RPP * NW * j + rowl;
            krow_ = krow_ < kv_limit ? krow_ : kv_limit - 1;
            const unsigned vo = (unsigned)krow_ * rowb + ((j & 1) ? gsw1 : gsw0) * 16;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(vo), "s"(base), "s"(ld0 + j * NW * 1024) : "memory");
        }
    };

    // ---------------- optimistic static reference (round 6; bf16 only) ----------------
    // The online softmax pays, per 32-key sub-step, a row maximum of 64 x 32 scores (20 vector instructions, two lane swaps), a
    // compare-and-branch on it and the rare rescale -- with one wave per SIMD all of it in the issue slots the MFMAs leave
    // (removal experiment, profiles/r06_k5_forms.txt: -3.9 % R2 / -5.3 % dense).  bf16 P has fp32's exponent range, so the reference
    // m_ref need not follow the running maximum: the steady-state loop keeps the reference it is ENTERED with (the finite row
    // maxima of the keys before it: at least the first 32) and computes no maxima at all (second body of the loop statement,
    // gen_k5_block64.py `static`).  exp2(S - m_ref) then overflows only if a later score exceeds that reference by more than
    // 127 -- and every overflow leaves a trace: an infinite P makes l infinite and O infinite or NaN.  So the walk is checked
    // afterwards (l below 2^100 and every O element finite, on the bit patterns: this file is compiled with -fno-honor-nans) and,
    // if any wave of the workgroup failed the check, the whole workgroup walks again through the online body and stores again.
    // Same softmax either way (the reference cancels in O / l); which body ran depends only on the rows' own data, never on the
    // launch.  Tuning key k5_static = 0: online body only.
    constexpr bool MAY_STATIC = std::is_same<Tag, bf16_tag>::value && XF == 0;
    static_assert(NW == 2 || XF == 0, "the A/B forms are forms of the 128-row kernel");
    const bool may_static = MAY_STATIC && a.k5_static != 0;
    // state the epilogue reads (set at the head of a pass)
    float l_run[2], m_ref[2], thr[2];
    int i0 = 0, i1 = 0;
#ifdef RSA_K5_DIAG
    // diagnostics build: s_memtime around the phases of the walk, summed per wave (scalar registers)
    unsigned long long tprev = 0, tsum[4] = {0, 0, 0, 0}, tkern0, tprologue = 0;
    auto stamp_now = [&]() -> unsigned long long {
        unsigned long long tt;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
        return tt;
    };
    tkern0 = stamp_now();
#define RSA_STAMP0() do { tprev = stamp_now(); } while (0)
#define RSA_STAMP(i) do { const unsigned long long tn_ = stamp_now(); tsum[i] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define RSA_STAMP0() do { } while (0)
#define RSA_STAMP(i) do { } while (0)
#endif
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
    if constexpr (D == 128) asm volatile(RSA_K5W_OZERO ::: RSA_K5W_CLOBBER_O); else asm volatile(RSA_K5V_OZERO ::: RSA_K5V_CLOBBER_O);
    bool used_static = false;
    // ---------------- state ----------------
    // m_ref = the finite reference the scores are taken against, nm = its negation in 16 registers (C operand of the first
    // QK^T MFMA), thr = how far a new row maximum may exceed it before the rescale (-inf until the row has seen a finite score)
    l_run[0] = l_run[1] = 0.0f;
    m_ref[0] = m_ref[1] = 0.0f;
    thr[0] = thr[1] = -INFINITY;
    f32x16 nm[2];
    asm volatile(RSA_K5W_NMZERO RSA_K5W_OPS_NMZERO);      // (arch registers only: the same for both head dims)

    // per-lane LDS read addressing (half-tile rows 0..31 of slot 0; slot and k-step are immediates of the block)
    const int g4 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    i32x8 ka, va;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ka[i] = 0; va[i] = 0; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = (int)lds_base + tile_off<D>(r, 2 * ks + hh);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int ch = 4 * dt + 2 * (g4 & 1) + (tp >> 1);
        va[2 * dt] = (int)lds_base + tile_off<D>(4 * hh + tq, ch) + 8 * (tp & 1);
        va[2 * dt + 1] = (int)lds_base + tile_off<D>(4 * hh + tq + 8, ch) + 8 * (tp & 1);
    }

    f32x16 SA[2], SB[2];
    float mxA[2] = {-INFINITY, -INFINITY}, mxB[2] = {-INFINITY, -INFINITY};
    const float ninf = -INFINITY, eight = 8.0f;

    // deferred rescale of the scores the next block consumes (S_cur = SA for even sub-steps), C++-driven form
    auto rescale_check = [&](auto UC, float (&mx_cur)[2], f32x16 (&SA)[2], f32x16 (&SB)[2], f32x16 (&nm)[2]) {
        constexpr int U = decltype(UC)::value;
        // S_cur, mx_cur are relative to m_ref as it was when they were computed, and that is still m_ref
        if (__builtin_amdgcn_ballot_w64(mx_cur[0] > thr[0] || mx_cur[1] > thr[1]) != 0ull) {
            float al[2], de[2], ng[2];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const bool move = __builtin_amdgcn_ballot_w64(mx_cur[x] > thr[x]) != 0ull;   // per half, wave-uniform
                const bool first = thr[x] == -INFINITY;
                float delta = first ? mx_cur[x] : fmaxf(mx_cur[x], 0.0f);
                if (delta == -INFINITY || !move) delta = 0.0f;      // nothing but masked keys so far / this half stays
                else thr[x] = 8.0f;
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // (first: O and l are still zero)
                m_ref[x] += delta;
                l_run[x] *= alpha;
                mx_cur[x] -= delta;     // the row maximum follows its scores to the new reference (the test may run again on them)
                al[x] = alpha; de[x] = delta; ng[x] = -m_ref[x];
            }
            const float al0 = al[0], al1 = al[1], de0 = de[0], de1 = de[1], ng0 = ng[0], ng1 = ng[1];
            if constexpr ((U & 1) == 0) RSA_K5_ASM(D, RESCALE_A, OPS_RESCALE_A, CL_TO);
            else RSA_K5_ASM(D, RESCALE_B, OPS_RESCALE_B, CL_TO);
        }
    };
    // One C++-driven sub-step u (U = u & 3): wait + barrier, staging of V(u+3) and K(u+4) (rows clamped), boundary mask and
    // deferred rescale on S_cur, block U.
    auto step = [&](auto UC, int u, f32x16 (&SA)[2], f32x16 (&SB)[2], f32x16 (&nm)[2]) {
        constexpr int U = decltype(UC)::value;
        float (&mx_cur)[2] = (U & 1) == 0 ? mxA : mxB;
        float (&mx_nxt)[2] = (U & 1) == 0 ? mxB : mxA;
        if (u + 4 <= n_sub) {      // the 2 NPW pieces of each of the last two sub-steps may still fly
            if constexpr (NPW == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (NPW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (u + 3 < n_sub) dma_half(1, key_of(u + 3), VRING + ((U + 3) & 3) * HALF);
        if (u + 4 < n_sub) dma_half(0, key_of(u + 4), U * HALF);
        const int kfirst = key_of(u);
        if (kfirst < lo_max || kfirst + 32 > hi_min) {   // scores outside the row's key range -> -inf, new row maxima
            const int kb0 = kfirst + 4 * hh - lo_r[0], kb1 = kfirst + 4 * hh - lo_r[1];
            const int sp0 = hi_r[0] > lo_r[0] ? hi_r[0] - lo_r[0] : 0, sp1 = hi_r[1] > lo_r[1] ? hi_r[1] - lo_r[1] : 0;
            float (&mx)[2] = mx_cur;
            if constexpr ((U & 1) == 0) {
                asm volatile(RSA_K5W_MASK_A RSA_K5W_OPS_MASK_A : "v146", "vcc");
                asm volatile(RSA_K5W_ROWMAX_A RSA_K5W_OPS_ROWMAX_A : "v146", "v147", "v148", "v149");
            } else {
                asm volatile(RSA_K5W_MASK_B RSA_K5W_OPS_MASK_B : "v146", "vcc");
                asm volatile(RSA_K5W_ROWMAX_B RSA_K5W_OPS_ROWMAX_B : "v146", "v147", "v148", "v149");
            }
        }
        rescale_check(UC, mx_cur, SA, SB, nm);
        k5w_block<Tag, U, D>(SA, SB, nm, l_run, mx_nxt, ka, va);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto run_items = [&](int i_begin, int i_end) {   // kept blocks [i_begin, i_end), C++-driven
        for (int i = i_begin; i < i_end; ++i) {
            step(I0{}, 4 * i, SA, SB, nm); step(I1{}, 4 * i + 1, SA, SB, nm); step(I2{}, 4 * i + 2, SA, SB, nm); step(I3{}, 4 * i + 3, SA, SB, nm);
        }
    };

    // aligned starts: in front of the first staging instruction (the pieces of a split tail and of the text rows do not wait:
    // short walks over different parts of the key range, the last workgroups of the launch)
    if (pass == 0 && tail < 0 && qblk < a.NBv) rsa_gsync_wait(a.gsync, gs_tk, n_items, a.NB_total, a.gsync_ratio);

    // ---------------- prologue: half-tiles K(0..3), V(0..2); scores of sub-step 0 ----------------
    if (n_sub > 0) {
#pragma unroll
        for (int x = 0; x < 4; ++x) dma_half(0, key_of(x), x * HALF);
#pragma unroll
        for (int x = 0; x < 3; ++x) dma_half(1, key_of(x), VRING + x * HALF);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float (&mx)[2] = mxA;
        if constexpr (std::is_same<Tag, bf16_tag>::value) RSA_K5_ASM(D, QK0_BF16, OPS_QK0, CLOBBER_TMP, , "memory");
        else RSA_K5_ASM(D, QK0_F16, OPS_QK0, CLOBBER_TMP, , "memory");
    }
    RSA_STAMP0();
#ifdef RSA_K5_DIAG
    tprologue = tprev - tkern0;
#endif
    // Which kept blocks the asm loop takes: [i0, i1) such that no score of block i needs the boundary mask and block i + 1
    // (whose half-tiles the loop stages while it works on i) exists and lies inside the valid keys: i0 = the leading blocks
    // below lo_max (second segment of a two-segment dense call), i1 from the end of the ascending list.
    i0 = 0;
    while (i0 < n_items && blk_of(i0) * RSA_BLOCK < lo_max) ++i0;
    int nfull = n_items;
    while (nfull > i0 && blk_of(nfull - 1) * RSA_BLOCK + RSA_BLOCK > hi_min) --nfull;
    i1 = nfull - 1 > i0 ? nfull - 1 : i0;
    run_items(0, i0);
    RSA_STAMP(0);
    {
        rescale_check(I0{}, mxA, SA, SB, nm);
        const unsigned cnt = (unsigned)(i1 - i0);
        const unsigned blk0 = cnt ? (unsigned)blk_of(i0) : 0u, blk1 = cnt ? (unsigned)blk_of(i0 + 1) : 0u;
        unsigned la = lds_base + 8 * HALF + 2 * (i0 + 2), lv;
        const unsigned long kb = (unsigned long)(uintptr_t)kbase, vb = (unsigned long)(uintptr_t)vbase;
        const unsigned ldsk = lds_base + wv * 1024, ldsv = lds_base + VRING + wv * 1024;
        float (&l)[2] = l_run;
        float (&mx)[2] = mxA;
        // static body iff every row of the wave has a finite reference by now (a row that has seen only masked keys has none)
        unsigned stat = 0;
        if (may_static && pass == 0 && cnt != 0 &&
            __builtin_amdgcn_ballot_w64(thr[0] == -INFINITY || thr[1] == -INFINITY) == 0ull) stat = 1;
        stat = __builtin_amdgcn_readfirstlane(stat);
        used_static = stat != 0;
#ifdef RSA_K5_DIAG
        unsigned d0 = 0, d1 = 0;   // in-loop stamps: cycles parked on the vmcnt wait / on the barrier (+ rescales taken << 24)
        if constexpr (std::is_same<Tag, bf16_tag>::value) RSA_K5_ASM(D, LOOP_BF16_DIAG, OPS_LOOP_DIAG, CL_LOOP, , RSA_K5W_CLOBBER_LOOP_DIAG, "memory");
        else RSA_K5_ASM(D, LOOP_F16_DIAG, OPS_LOOP_DIAG, CL_LOOP, , RSA_K5W_CLOBBER_LOOP_DIAG, "memory");
        tsum[3] = ((unsigned long long)d1 << 32) | d0;
#else
#ifdef RSA_K5_FORMS
#define RSA_K5W_XFORM(N) else if constexpr (XF == N) \
            asm volatile(RSA_K5W_LOOP_BF16_X##N RSA_K5W_OPS_LOOP : RSA_K5W_CLOBBER_TMP, RSA_K5W_CLOBBER_O, RSA_K5W_CLOBBER_LOOP, "memory");
        if constexpr (false) {}
        RSA_K5W_XFORM(1) RSA_K5W_XFORM(2) RSA_K5W_XFORM(3) RSA_K5W_XFORM(4) RSA_K5W_XFORM(5) RSA_K5W_XFORM(6) RSA_K5W_XFORM(7) RSA_K5W_XFORM(8) RSA_K5W_XFORM(9) RSA_K5W_XFORM(10) RSA_K5W_XFORM(11) RSA_K5W_XFORM(12) RSA_K5W_XFORM(13) RSA_K5W_XFORM(14) RSA_K5W_XFORM(15) RSA_K5W_XFORM(16) RSA_K5W_XFORM(17) RSA_K5W_XFORM(18) RSA_K5W_XFORM(19) RSA_K5W_XFORM(20) RSA_K5W_XFORM(21) RSA_K5W_XFORM(22) RSA_K5W_XFORM(23) RSA_K5W_XFORM(24) RSA_K5W_XFORM(25) RSA_K5W_XFORM(26) RSA_K5W_XFORM(27) RSA_K5W_XFORM(28) RSA_K5W_XFORM(29) RSA_K5W_XFORM(30) RSA_K5W_XFORM(31)
        else
#endif
        if constexpr (NW == 4 && std::is_same<Tag, bf16_tag>::value) RSA_K5_ASM(D, LOOP_BF16_R256, OPS_LOOP, CL_LOOP, , "memory");
        else if constexpr (NW == 4) RSA_K5_ASM(D, LOOP_F16_R256, OPS_LOOP, CL_LOOP, , "memory");
        else if constexpr (std::is_same<Tag, bf16_tag>::value) RSA_K5_ASM(D, LOOP_BF16, OPS_LOOP, CL_LOOP, , "memory");
        else RSA_K5_ASM(D, LOOP_F16, OPS_LOOP, CL_LOOP, , "memory");
#endif
        (void)lv;
    }
    RSA_STAMP(1);
    run_items(i1, n_items);
    RSA_STAMP(2);
    if (!may_static) break;                      // (uniform over the launch)
    {
        // the overflow trace of a static walk: l of either half at 2^100 or beyond (or infinite, or NaN), or an O element that is
        // not finite -- on the bit patterns (-fno-honor-nans).  Reading the whole tile costs ~400 vector instructions per wave
        // life (of ~500 k cycles); nothing has been stored yet.
        asm volatile("s_nop 11" ::: "memory");   // (the last block's last MFMA -> the reads of O below)
        unsigned emax = 0;
        auto track = [&](const f32x16& o) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned bits = __float_as_uint(o[i]) & 0x7FFFFFFFu;
                emax = emax > bits ? emax : bits;
            }
        };
        if (used_static) {
            track(k5w_oread<0, 0, D>()); track(k5w_oread<0, 1, D>()); track(k5w_oread<1, 0, D>()); track(k5w_oread<1, 1, D>());
            if constexpr (DT == 4) {
                track(k5w_oread<0, 2, D>()); track(k5w_oread<0, 3, D>()); track(k5w_oread<1, 2, D>()); track(k5w_oread<1, 3, D>());
            }
        }
        const bool lbad = (__float_as_uint(l_run[0]) & 0x7FFFFFFFu) >= 0x71800000u || (__float_as_uint(l_run[1]) & 0x7FFFFFFFu) >= 0x71800000u;
        const bool bad = used_static && (emax >= 0x7F800000u || lbad);
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) redo_flag = 1;
        __syncthreads();
        if (redo_flag == 0) break;               // (a second pass leaves the flag set: it is the last one)
    }
    }   // pass

    // ---------------- epilogue ----------------
    asm volatile("s_nop 11" ::: "memory");   // (the last block's last MFMA -> the reads of O below)
    // Everything the epilogue needs from the arguments is read AGAIN here, through a pointer the compiler cannot see through
    // (kept live in scalar registers across the walk it costs 25 of them).
    const AttnArgs* ep = (const AttnArgs*)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ep));
    const AttnArgs& e = *ep;
    int bh2, qblk2, tsp2, tail2;
    {
        int work2 = blockIdx.x;
        asm volatile("" : "+s"(work2));
        k5w_map(e, work2, bh2, qblk2, tsp2, tail2);
    }
    const int b2 = bh2 / e.H, h2 = bh2 % e.H;
    const bool partial = (e.mode == MODE_SPARSE && e.tsplit > 1 && qblk2 >= e.NBv) || tail2 >= 0;
    const bool rectify = e.mode == MODE_SPARSE && qblk2 < e.NBv && e.R != nullptr && tail2 < 0;   // (a tail piece is rectified by its combine pass)
    // the compensation row of this query block, the 64 values this lane adds (d = 32 dt + 8 g + 4 hh + 0..3), and R: ALL loads
    // issued here, back to back, one wait -- per (dt, g) inside the store loop each load's latency (L2 / HBM: 1-2 k cycles)
    // is exposed in turn: 30 k of a wave's 535 k cycles with nothing else on the SIMD to cover it (stamps: profiles/r04_k5_w64.md)
    float4 cv[DT][4];
    float Rv = 1.0f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) cv[dt][g] = make_float4(0, 0, 0, 0);
    if (rectify) {
        const long rowi = (long)bh2 * e.NBv + qblk2;
        const float* cp = e.comp + rowi * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[dt][g] = *reinterpret_cast<const float4*>(cp + 32 * dt + 8 * g + 4 * hh);
        Rv = e.R[rowi];
    }
    auto finish_half = [&](auto HX) {
        constexpr int x = decltype(HX)::value;
        const int grow2 = qblk2 * RW + 64 * wv + 32 * x + r;
        const float mrun = thr[x] == -INFINITY ? -INFINITY : m_ref[x];
        const auto swl = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[x]), __float_as_uint(l_run[x]), false, false);
        const float l_tot = __uint_as_float(swl[0]) + __uint_as_float(swl[1]);
        if (partial) {
            // split-KV partial of a text block or of a tail piece: unnormalised O (fp32), m (log2 domain) and l per row (merged by
            // text_combine_kernel / tail_combine_kernel, rsa_attn.hip)
            const int ntq = e.NQB - e.NBv;
            const int rowb = 64 * wv + 32 * x + r;
            float* pp = tail2 >= 0 ? e.tail_part + ((long)tail2 * RSA_BLOCK + rowb) * (D + 2)
                                   : e.tpart + ((((long)bh2 * ntq + (qblk2 - e.NBv)) * e.tsplit + tsp2) * RSA_BLOCK + rowb) * (D + 2);
            auto put = [&](int dt, const f32x16& o) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * dt + 8 * g + 4 * hh;
                    *reinterpret_cast<float2*>(pp + d0) = make_float2(o[4 * g + 0], o[4 * g + 1]);
                    *reinterpret_cast<float2*>(pp + d0 + 2) = make_float2(o[4 * g + 2], o[4 * g + 3]);
                }
            };
            put(0, k5w_oread<x, 0, D>()); put(1, k5w_oread<x, 1, D>());
            if constexpr (DT == 4) { put(2, k5w_oread<x, 2, D>()); put(3, k5w_oread<x, 3, D>()); }
            if (hh == 0) *reinterpret_cast<float2*>(pp + D) = make_float2(mrun, l_tot);
            return;
        }
        // rows that are stored / written as zeros (padded text rows), from the row index (not kept live across the main loop)
        const bool text_blk = e.mode == MODE_SPARSE && qblk2 >= e.NBv;
        const bool st_r = grow2 < (text_blk ? e.q_text_end : e.Sq);
        const bool zr = text_blk && !st_r && grow2 < e.Sq;
        if (!(st_r || zr)) return;
        float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
        if (zr) inv = 0.0f;
        const float sc = inv * Rv;
        unsigned short* op = e.out + (long)b2 * e.osb + (long)h2 * e.osh + (long)grow2 * e.oss;
        // O * sc + comp: one fma rounded to fp32, THEN the conversion to the storage type (what the oracle does); the empty asm
        // keeps hipcc from folding fma + conversion into v_fma_mixlo_f16 in one store form and not in the other
        auto fin = [&](float acc, float c) -> float {
            float rr = __builtin_fmaf(acc, sc, c);
            asm volatile("" : "+v"(rr));
            return rr;
        };
        auto put = [&](int dt, const f32x16& o) {
            if constexpr (WIDE) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint2 pk[2];
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int g = 2 * gp + gi;
                        float4 c4 = cv[dt][g];
                        if (zr) c4 = make_float4(0, 0, 0, 0);
                        const float v0 = fin(o[4 * g + 0], c4.x);
                        const float v1 = fin(o[4 * g + 1], c4.y);
                        const float v2 = fin(o[4 * g + 2], c4.z);
                        const float v3 = fin(o[4 * g + 3], c4.w);
                        pk[gi].x = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
                        pk[gi].y = (unsigned)E::from_f32(v2) | ((unsigned)E::from_f32(v3) << 16);
                    }
                    const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    uint4 w4;
                    w4.x = sx[0]; w4.y = sy[0]; w4.z = sx[1]; w4.w = sy[1];
                    *reinterpret_cast<uint4*>(op + 32 * dt + 8 * (2 * gp + hh)) = w4;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * dt + 8 * g + 4 * hh;
                    float4 c4 = cv[dt][g];
                    if (zr) c4 = make_float4(0, 0, 0, 0);
                    const float v0 = fin(o[4 * g + 0], c4.x);
                    const float v1 = fin(o[4 * g + 1], c4.y);
                    const float v2 = fin(o[4 * g + 2], c4.z);
                    const float v3 = fin(o[4 * g + 3], c4.w);
                    uint2 pk;
                    pk.x = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
                    pk.y = (unsigned)E::from_f32(v2) | ((unsigned)E::from_f32(v3) << 16);
                    *reinterpret_cast<uint2*>(op + d0) = pk;
                }
            }
        };
        put(0, k5w_oread<x, 0, D>()); put(1, k5w_oread<x, 1, D>());
        if constexpr (DT == 4) { put(2, k5w_oread<x, 2, D>()); put(3, k5w_oread<x, 3, D>()); }
    };
    finish_half(std::integral_constant<int, 0>{});
    finish_half(std::integral_constant<int, 1>{});
#ifdef RSA_K5_DIAG
    if (e.dbg && lane == 0) {   // [leading C++ blocks, asm loop, trailing C++ blocks, -, kept blocks, kernel cycles, blocks in the loop]
        const unsigned long long tend = stamp_now();
        unsigned long long* o8 = e.dbg + ((long)blockIdx.x * 4 + wv) * 8;
        o8[0] = tsum[0]; o8[1] = tsum[1]; o8[2] = tsum[2]; o8[3] = tsum[3]; o8[4] = (unsigned long long)n_items;
        o8[5] = tend - tkern0; o8[6] = (unsigned long long)(i1 - i0); o8[7] = (tprologue << 32) | ((tend - tprev) & 0xFFFFFFFFull);
    }
#endif
}

#ifdef RSA_K5_FORMS
int g_rsa_k5w_form = 0;   // A/B build: tuning key "k5w_form" (loop forms of rsa_attn_block64_forms.h)
#endif
// launch hook used by rsa_attn.hip::launch_attn (head dims 128 and 64)
int rsa_launch_bsfwd64(const AttnArgs& a, dim3 grid, size_t lds_bytes, int D, int dtype, hipStream_t s) {
    const bool wide = !(((uintptr_t)a.out & 15) || ((a.osb | a.osh | a.oss) & 7));
    lds_bytes += 16;   // the loop reads its list two entries ahead
    if (D == 64 && a.rows256) {
        if (a.mode != MODE_DENSE) return RSA_ERR_BAD_ARG;
        if (dtype == RSA_BF16) {
            if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, true, 0, 4, 64>), a, false, grid, 256, lds_bytes, s);
            else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, false, 0, 4, 64>), a, false, grid, 256, lds_bytes, s);
        } else {
            if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, true, 0, 4, 64>), a, false, grid, 256, lds_bytes, s);
            else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, false, 0, 4, 64>), a, false, grid, 256, lds_bytes, s);
        }
        return rsa_launch_status();
    }
    if (D == 64) {
        if (dtype == RSA_BF16) {
            if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, true, 0, 2, 64>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
            else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, false, 0, 2, 64>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
        } else {
            if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, true, 0, 2, 64>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
            else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, false, 0, 2, 64>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
        }
        return rsa_launch_status();
    }
    if (D != 128) return RSA_ERR_UNSUPPORTED;
#ifdef RSA_K5_FORMS
#define RSA_K5W_XLAUNCH(N) if (g_rsa_k5w_form == N && dtype == RSA_BF16 && wide) { RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, true, N>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s); return rsa_launch_status(); }
    RSA_K5W_XLAUNCH(1) RSA_K5W_XLAUNCH(2) RSA_K5W_XLAUNCH(3) RSA_K5W_XLAUNCH(4) RSA_K5W_XLAUNCH(5) RSA_K5W_XLAUNCH(6) RSA_K5W_XLAUNCH(7) RSA_K5W_XLAUNCH(8) RSA_K5W_XLAUNCH(9) RSA_K5W_XLAUNCH(10) RSA_K5W_XLAUNCH(11) RSA_K5W_XLAUNCH(12) RSA_K5W_XLAUNCH(13) RSA_K5W_XLAUNCH(14) RSA_K5W_XLAUNCH(15) RSA_K5W_XLAUNCH(16) RSA_K5W_XLAUNCH(17) RSA_K5W_XLAUNCH(18) RSA_K5W_XLAUNCH(19) RSA_K5W_XLAUNCH(20) RSA_K5W_XLAUNCH(21) RSA_K5W_XLAUNCH(22) RSA_K5W_XLAUNCH(23) RSA_K5W_XLAUNCH(24) RSA_K5W_XLAUNCH(25) RSA_K5W_XLAUNCH(26) RSA_K5W_XLAUNCH(27) RSA_K5W_XLAUNCH(28) RSA_K5W_XLAUNCH(29) RSA_K5W_XLAUNCH(30) RSA_K5W_XLAUNCH(31)
#endif
    if (a.rows256) {     // dense calls: 256-row tiles, four waves on one K/V ring (the host counted the grid in such tiles)
        if (a.mode != MODE_DENSE) return RSA_ERR_BAD_ARG;
        if (dtype == RSA_BF16) {
            if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, true, 0, 4>), a, false, grid, 256, lds_bytes, s);
            else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, false, 0, 4>), a, false, grid, 256, lds_bytes, s);
        } else {
            if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, true, 0, 4>), a, false, grid, 256, lds_bytes, s);
            else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, false, 0, 4>), a, false, grid, 256, lds_bytes, s);
        }
        return rsa_launch_status();
    }
    if (dtype == RSA_BF16) {
        if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, true>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
        else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<bf16_tag, false>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
    } else {
        if (wide) RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, true>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
        else RSA_LAUNCH_GSYNC(1, (bsfwd64_kernel<fp16_tag, false>), a, a.mode == MODE_SPARSE, grid, 128, lds_bytes, s);
    }
    return rsa_launch_status();
}
