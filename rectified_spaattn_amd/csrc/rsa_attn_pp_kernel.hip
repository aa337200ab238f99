// K5, "ping-pong" form: block-sparse flash attention forward for gfx950 with the rectification epilogue fused.
//
// One workgroup = 8 waves = TWO 128-row query blocks of one head (group A = waves 0-3, group B = waves 4-7; wave w of a
// group owns rows 32w..32w+31 of its block), one workgroup per CU, so every SIMD hosts one wave of A and one of B.  The
// two groups walk their OWN kept lists through their OWN K/V rings (LDS: 2 x [K0 K1 V0 V1] + the two lists), but they
// are phase-locked by the workgroup barrier: between two consecutive barriers one group runs a MATRIX segment
// (16 MFMAs: scores of sub-step u, PV of sub-step u-1 -- nothing but MFMA and LDS reads) while the other runs its VECTOR
// segment (softmax of the scores just produced, packing, LDS-DMA issue, scalar address work), then they swap.  The
// matrix pipe of a SIMD therefore never sees two waves' MFMA streams at once, and the vector / scalar / memory issue of
// one wave runs beside the other wave's MFMAs instead of beside its own (MI355X_MICROARCH.md, "Two waves per SIMD").
// Group B starts one segment late (one extra barrier up front); the group with the shorter list pads with idle barriers.
//
// Per group the data flow is the one of rsa_attn_kernel.hip ("key on the register, query row on the lane"):
//      S^T[key][q]  = K . Q^T      A = K rows (ds_read_b128 from the XOR-swizzled row-major tile), B = Q (registers)
//      O^T[d][q]   += V^T . P^T    A = V^T (ds_read_b64_tr_b16), B = P^T = the S^T accumulator converted in place
// with two differences that the segment split makes possible:
//   * S is single-buffered (produced in a matrix segment, consumed in the next vector segment): 16 registers fewer;
//   * those registers hold -m (the row's deferred reference maximum) as a 16-register block that seeds the QK^T
//     accumulator chain, so the accumulator IS s - m and the softmax needs no subtraction (16 VALU per sub-step less).
// Staging: LDS-DMA (global_load_lds_dwordx4) from inline asm, issued in vector segments: V(t+1) in the vector segment of
// sub-step (t,0), K(t+2) in that of (t,1); `s_waitcnt vmcnt(2 groups)` at the end of every vector segment leaves the two
// newest groups in flight, so every tile has two sub-steps to land.
//
// Semantics kept from the reference kernel (rectified_hunyuan_attn.py:15-105) are those listed in rsa_attn_kernel.hip.
#include "rsa_attn.h"

// OPT bits: 2 = issue priority 2 inside matrix segments; 256 = iglp_opt(0) on matrix segments;
//           4 = seed the QK^T accumulator chain with -m (16-register block) instead of subtracting m in the vector segment
//           16 / 32 = 2 / all 4 of a group's LDS-DMA pieces are issued inside the NEXT matrix segment (between its
//           MFMAs) instead of in the vector segment
template <int D, typename Tag, int OPT>
__global__ __launch_bounds__(512, 2) void bsfwd_pp_kernel(AttnArgs a) {
    constexpr bool SEED = (OPT & 4) != 0;
    constexpr int DMA_M = (OPT & 32) ? 4 : ((OPT & 16) ? 2 : 0);   // pieces (of NPC = 4) deferred to the matrix segment
    constexpr bool T_NODMA = (OPT & 64) != 0;    // TIMING-ONLY builds (wrong results): no staging in the loop
    constexpr bool T_NOSM = (OPT & 128) != 0;    //                                      no softmax arithmetic
    constexpr bool T_NOWAIT = (OPT & 512) != 0;  //                                      no vmcnt wait for the staged tiles
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    constexpr int CHR = D / 8;
    constexpr int RPI = 1024 / (D * 2);          // rows per 1-KiB piece
    constexpr int TILE_BYTES = 64 * D * 2;
    constexpr int NPC = TILE_BYTES / 1024 / 4;   // 1-KiB pieces per wave per tile operand (4 waves stage a tile)
    using E = Elem<Tag>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wv8 = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wv8 >> 2, wv = wv8 & 3;
    const int tg = t & 255;
    unsigned char* ring = lds + grp * 4 * TILE_BYTES;
    unsigned short* lds_list = reinterpret_cast<unsigned short*>(lds + 8 * TILE_BYTES) + grp * a.list_cap;
    int* lds_n = reinterpret_cast<int*>(lds + 8 * TILE_BYTES + 4 * a.list_cap);

    // ---------------- work mapping: pairs of dense text-row blocks first, then pairs of sparse blocks per XCD ----------
    int bh, qblk;
    bool live;
    {
        const int bid = blockIdx.x;
        if (bid < a.n_heavy_pad) {
            const int ntq = a.NQB - a.NBv;
            const int npt = (ntq + 1) >> 1;
            if (ntq <= 0 || bid >= a.BH * npt) return;  // whole workgroup
            bh = bid / npt;
            qblk = a.NBv + 2 * (bid % npt) + grp;
            live = qblk < a.NQB;
        } else {
            const int v = bid - a.n_heavy_pad;
            bh = v / a.NPp;
            const int j = v % a.NPp;
            const int chunk = a.NPp >> 3;
            const int pair = (j & 7) * chunk + (j >> 3);
            if (2 * pair >= a.NBv) return;              // whole workgroup
            qblk = 2 * pair + grp;
            live = qblk < a.NBv;
        }
    }
    const int b = bh / a.H, h = bh % a.H;
    const int r = lane & 31, hh = lane >> 5;
    const int grow = qblk * 128 + 32 * wv + r;

    // ---------------- per-row plan (as rsa_attn_kernel.hip) ----------------
    int lo_r = 0, hi_r = 0;
    bool store_r = false, zero_r = false;
    int n_items = 0, first_blk = 0, lo_max = 0, hi_min = 0, hi_max = 0;
    const int32_t* list = nullptr;
    bool rectify = false;
    if (live) {
        if (a.mode == MODE_SPARSE) {
            if (qblk < a.NBv) {
                const long rowi = (long)bh * a.NBv + qblk;
                list = a.cols + rowi * a.NB_total;
                n_items = a.counts[rowi];
                lo_max = 0; hi_min = hi_max = a.kv_valid;
                rectify = a.R != nullptr;
                hi_r = a.kv_valid; store_r = grow < a.Sq;
            } else {
                n_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
                lo_max = 0; hi_min = hi_max = a.kv_text_valid;
                hi_r = a.kv_text_valid;
                store_r = grow < a.q_text_end;
                zero_r = !store_r && grow < a.Sq;
            }
        } else {
            const int row0 = qblk * 128, row1 = row0 + 128;
            if (grow < a.q_split) { lo_r = 0; hi_r = a.kv_split; }
            else { lo_r = a.kv_split; hi_r = a.Sk; }
            store_r = grow < a.Sq;
            int lo_min;
            if (row1 <= a.q_split) { lo_min = 0; lo_max = 0; hi_min = hi_max = a.kv_split; }
            else if (row0 >= a.q_split) { lo_min = lo_max = a.kv_split; hi_min = hi_max = a.Sk; }
            else { lo_min = 0; lo_max = a.kv_split; hi_min = a.kv_split; hi_max = a.Sk; }
            first_blk = lo_min / RSA_BLOCK;
            n_items = (hi_max + RSA_BLOCK - 1) / RSA_BLOCK - first_blk;
            if (hi_max <= lo_min) n_items = 0;
        }
    }
    n_items = __builtin_amdgcn_readfirstlane(n_items);
    const bool use_list = list != nullptr;
    if (use_list)
        for (int i = tg; i < n_items; i += 256) lds_list[i] = (unsigned short)list[i];
    int n_tiles = 2 * n_items;
    if (n_items > 0) {
        const int last_blk = use_list ? (int)list[n_items - 1] : first_blk + n_items - 1;
        if (last_blk * RSA_BLOCK + 64 >= hi_max) n_tiles -= 1;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    if (tg == 0) lds_n[grp] = n_tiles;
    __syncthreads();
    const int n_tiles_max = __builtin_amdgcn_readfirstlane(lds_n[0] > lds_n[1] ? lds_n[0] : lds_n[1]);
    const int n_sub = 2 * n_tiles;                       // 32-key sub-steps of this group
    const int bar_total = n_tiles_max > 0 ? 4 * n_tiles_max + 2 : 0;
    int nbar = 0;
    auto blk_of = [&](int item) -> int { return use_list ? (int)lds_list[item] : first_blk + item; };
    const int kv_limit = hi_max < a.Sk ? hi_max : a.Sk;
    auto key0_of = [&](int tile) -> int {  // first key of tile `tile` (clamped index: callers guard tile < n_tiles)
        const int it = tile >> 1;
        const int blk = __builtin_amdgcn_readfirstlane(blk_of(it < n_items ? it : (n_items > 0 ? n_items - 1 : 0)));
        return blk * RSA_BLOCK + (tile & 1) * 64;
    };

    // ---------------- Q fragments (B operand) ----------------
    s16x8 qf[KS];
    {
        const unsigned short* qp = a.q + (long)b * a.qsb + (long)h * a.qsh + (long)grow * a.qss + 8 * hh;
        const bool qok = live && grow < a.Sq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (qok) raw = *reinterpret_cast<const uint4*>(qp + 16 * ks);
            const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = rsa_to_f32<Tag>((unsigned short)(w4[e] & 0xFFFF)) * a.qk_scale;
                f[2 * e + 1] = rsa_to_f32<Tag>((unsigned short)(w4[e] >> 16)) * a.qk_scale;
            }
            qf[ks] = E::cvt8(f);
        }
    }

    // ---------------- LDS-DMA staging (4 waves of the group move one tile) ----------------
    const unsigned char* kbase = reinterpret_cast<const unsigned char*>(a.k + (long)b * a.ksb + (long)h * a.ksh);
    const unsigned char* vbase = reinterpret_cast<const unsigned char*>(a.v + (long)b * a.vsb + (long)h * a.vsh);
    const int rsub = lane / CHR, cl = lane % CHR;
    const int rowl = wv * RPI + rsub;  // row inside the first group of 4 pieces
    int gsw;
    if constexpr (D == 128) gsw = cl ^ (((rowl & 3) << 2) | ((rowl >> 2) & 3));
    else gsw = cl ^ ((rowl >> 1) & 7);
    const unsigned voffk = (unsigned)(((long)rowl * a.kss + gsw * 8) * 2);
    const unsigned voffv = (unsigned)(((long)rowl * a.vss + gsw * 8) * 2);
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const long kstep = (long)(4 * RPI) * a.kss * 2, vstep = (long)(4 * RPI) * a.vss * 2;  // bytes per 4 pieces
    auto dma = [&](int is_v, int key0, int slot, int j0, int j1) {
        const unsigned ld0 = lds_ring + ((is_v ? 2 : 0) + slot) * TILE_BYTES + wv * 1024;
        const unsigned char* base = is_v ? vbase : kbase;
        const long ss = is_v ? a.vss : a.kss;
        if (key0 + 64 <= kv_limit) {
            const unsigned char* tb = base + (long)key0 * ss * 2;
            const long step = is_v ? vstep : kstep;
            const unsigned vo = is_v ? voffv : voffk;
#pragma unroll
            for (int j = 0; j < NPC; ++j)
                if (j >= j0 && j < j1)
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                                 :: "v"(vo), "s"(tb + j * step), "s"(ld0 + j * 4096) : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                if (j < j0 || j >= j1) continue;
                int krow = key0 + j * 4 * RPI + rowl;
                krow = krow < kv_limit ? krow : kv_limit - 1;
                const unsigned vo = (unsigned)(((long)krow * ss + gsw * 8) * 2);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(vo), "s"(base), "s"(ld0 + j * 4096) : "memory");
            }
        }
    };

    // ---------------- state ----------------
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    f32x16 negm, S;
#pragma unroll
    for (int i = 0; i < 16; ++i) { negm[i] = 0.0f; S[i] = 0.0f; }
    s16x8 pb[2];
    pb[0] = pb[1] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};

    int pend_isv = 0, pend_key = 0, pend_slot = 0;
    bool pend = false;  // a DMA group whose last DMA_M pieces the next matrix segment issues

    // per-lane read addressing
    const int kswz = ((r & 3) << 2) | ((r >> 2) & 3);
    const int g4 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    int vrd[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int ch = 4 * dt + 2 * (g4 & 1) + (tp >> 1);
        vrd[dt][0] = tile_off<D>(4 * hh + tq, ch) + 8 * (tp & 1);
        vrd[dt][1] = tile_off<D>(4 * hh + tq + 8, ch) + 8 * (tp & 1);
    }
    auto k_off = [&](int ks, int sub) {
        if constexpr (D == 128) return (32 * sub + r) * 256 + (((2 * ks + hh) ^ kswz) << 4);
        else return tile_off<D>(32 * sub + r, 2 * ks + hh);
    };

    // ---- matrix segment: S = -m + K[slot][sub] . Q^T   and/or   O^T += V[slot][sub]^T . P^T ----
    auto seg_m = [&](auto QK, auto KSLOT, auto KSUB, auto PV, auto VSLOT, auto VSUB) {
        constexpr bool qk = decltype(QK)::value != 0, pv = decltype(PV)::value != 0;
        constexpr int kslot = decltype(KSLOT)::value, ksub = decltype(KSUB)::value;
        constexpr int vslot = decltype(VSLOT)::value, vsub = decltype(VSUB)::value;
        if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(2);
        if constexpr (OPT & 256) __builtin_amdgcn_iglp_opt(0);
        if constexpr (qk) {
            const unsigned char* kt_ = ring + kslot * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (DMA_M > 0) {   // deferred LDS-DMA pieces ride between the first MFMAs
                    constexpr int per = DMA_M / 2;
                    if (ks == 1 && pend) dma(pend_isv, pend_key, pend_slot, NPC - DMA_M, NPC - DMA_M + per);
                    if (ks == 3 && pend) dma(pend_isv, pend_key, pend_slot, NPC - DMA_M + per, NPC);
                }
                const s16x8 a0 = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, ksub));
                if constexpr (SEED) {
                    if (ks == 0) S = E::mfma_from(a0, qf[0], negm);   // accumulator chain seeded with -m
                    else S = E::mfma(a0, qf[ks], S);
                } else {
                    if (ks == 0) {
                        f32x16 z;
#pragma unroll
                        for (int i = 0; i < 16; ++i) z[i] = 0.0f;
                        S = E::mfma(a0, qf[0], z);
                    } else {
                        S = E::mfma(a0, qf[ks], S);
                    }
                }
            }
        }
        if constexpr (pv) {
            const unsigned char* vt_ = ring + (2 + vslot) * TILE_BYTES;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int offa = vrd[dt][0] + (2 * vsub + k2) * 16 * D * 2;
                    const int offb = vrd[dt][1] + (2 * vsub + k2) * 16 * D * 2;
                    const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(vt_ + offa));
                    const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3)))*)(vt_ + offb));
                    const s16x8 av = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[dt] = E::mfma(av, pb[k2], o[dt]);
                }
            }
        }
        if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    int key0 = 0, kq1 = 0, kq2 = 0;  // first keys of the current tile and of tile+1 / tile+2

    // ---- vector segment of sub-step (tile, SUB): stage ahead, then softmax of S (= s - m of 32 keys) -> pb ----
    auto seg_v = [&](auto VS, auto SUB, int tile) {
        constexpr int vs = decltype(VS)::value, sub = decltype(SUB)::value;
        pend = false;
        if constexpr (T_NODMA) {
        } else if constexpr (sub == 0) {
            if (tile + 1 < n_tiles) {                      // V(tile+1) -> slot of V(tile-1)
                dma(1, kq1, vs ^ 1, 0, NPC - DMA_M);
                pend = DMA_M > 0; pend_isv = 1; pend_key = kq1; pend_slot = vs ^ 1;
            }
        } else {
            if (tile + 2 < n_tiles) {                      // K(tile+2) -> slot of K(tile)
                dma(0, kq2, vs, 0, NPC - DMA_M);
                pend = DMA_M > 0; pend_isv = 0; pend_key = kq2; pend_slot = vs;
            }
        }
        // SEED: S holds s - m_use; otherwise S holds the raw scores s and m_use is subtracted below
        float m_use = (m_run == -INFINITY) ? 0.0f : m_run;
        const int kfirst = key0 + 32 * sub;
        if (kfirst < lo_max || kfirst + 32 > hi_min) {             // rare: boundary tile
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = kfirst + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (kk < lo_r || kk >= hi_r) S[i] = -INFINITY;
            }
        }
        float mx;
        if constexpr (T_NOSM) {
            mx = S[0];
            asm volatile("" :: "v"(S));
        } else {
            float m0 = fmaxf(S[0], S[1]), m1 = fmaxf(S[2], S[3]);
#pragma unroll
            for (int i = 4; i < 16; i += 4) {
                m0 = fmaxf(m0, fmaxf(S[i], S[i + 1]));
                m1 = fmaxf(m1, fmaxf(S[i + 2], S[i + 3]));
            }
            const float m = fmaxf(m0, m1);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // deferred reference: move it only when the row maximum grew by more than 2^8 (P <= 2^8)
        const float mxs = SEED ? mx + m_use : mx;   // true row maximum of the 32 scores
        const bool grow_row = mxs > m_run + 8.0f;
        if (__builtin_amdgcn_ballot_w64(grow_row) != 0ull) {
            const float m_new = fmaxf(m_run, mxs);
            const float mu = (m_new == -INFINITY) ? 0.0f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - mu);
            m_run = m_new;
            l_run *= alpha;
            if constexpr (SEED) {
                const float delta = m_use - mu;
#pragma unroll
                for (int i = 0; i < 16; ++i) { S[i] += delta; negm[i] = -mu; }
            }
            m_use = mu;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        float ps0 = 0.0f, ps1 = 0.0f, ps2 = 0.0f, ps3 = 0.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float pv8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (T_NOSM) pv8[i] = m_use + (float)i;
                else pv8[i] = __builtin_amdgcn_exp2f(SEED ? S[8 * half + i] : S[8 * half + i] - m_use);
            }
            ps0 += pv8[0] + pv8[4];
            ps1 += pv8[1] + pv8[5];
            ps2 += pv8[2] + pv8[6];
            ps3 += pv8[3] + pv8[7];
            pb[half] = E::cvt8(pv8);
        }
        l_run += (ps0 + ps1) + (ps2 + ps3);
        asm volatile("" : "+v"(l_run));   // the sum is formed HERE (hipcc otherwise carries the 16 P values across the
                                          // matrix segment and adds them in the next vector segment)
        // the two newest DMA groups may stay in flight; everything older must have landed before the next matrix
        // segments read it (the barrier that follows publishes it to the other waves of the group)
        if constexpr (T_NOWAIT) {
        } else if (tile + 2 < n_tiles) {
            static_assert(NPC == 4 || DMA_M == 0, "piece deferral is written for D = 128 (4 pieces per wave)");
            if constexpr (NPC == 4 && DMA_M == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NPC == 4 && DMA_M == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (NPC == 4 && DMA_M == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() { __syncthreads(); ++nbar; __builtin_amdgcn_sched_barrier(0); };
    auto advance = [&](int tile) {  // after finishing `tile`: shift the key queue, fetch tile+3's first key
        key0 = kq1;
        kq1 = kq2;
        kq2 = key0_of(tile + 3);
    };

    // ---------------- prologue ----------------
    if (n_tiles > 0) {
        key0 = key0_of(0);
        kq1 = key0_of(1);
        kq2 = key0_of(2);
        dma(0, key0, 0, 0, NPC);
        dma(1, key0, 0, 0, NPC);
        if (n_tiles > 1) dma(0, kq1, 1, 0, NPC);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grp == 1 && bar_total > 0) bar();   // group B runs one segment behind group A

    // ---------------- main loop: sub-step u = 2*tile + sub; matrix segment u = QK(u) + PV(u-1) ----------------
    if (n_sub > 0) {
        bar(); seg_m(I1{}, I0{}, I0{}, I0{}, I0{}, I0{});            // QK(0)
        bar(); seg_v(I0{}, I0{}, 0);                                   // softmax(0), issue V(1)
        int tile = 0;
        // u = 4i+1 .. 4i+4  (tiles 2i, 2i+1, 2i+2): all LDS slots are compile-time constants
        for (; 2 * tile + 4 < n_sub; tile += 2) {
            bar(); seg_m(I1{}, I0{}, I1{}, I1{}, I0{}, I0{});        // QK(tile, sub 1)     + PV(tile, sub 0)
            bar(); seg_v(I0{}, I1{}, tile);                            // softmax(tile,1), issue K(tile+2)
            advance(tile);
            bar(); seg_m(I1{}, I1{}, I0{}, I1{}, I0{}, I1{});        // QK(tile+1, sub 0)   + PV(tile, sub 1)
            bar(); seg_v(I1{}, I0{}, tile + 1);                        // softmax(tile+1,0), issue V(tile+2)
            bar(); seg_m(I1{}, I1{}, I1{}, I1{}, I1{}, I0{});        // QK(tile+1, sub 1)   + PV(tile+1, sub 0)
            bar(); seg_v(I1{}, I1{}, tile + 1);                        // softmax(tile+1,1), issue K(tile+3)
            advance(tile + 1);
            bar(); seg_m(I1{}, I0{}, I0{}, I1{}, I1{}, I1{});        // QK(tile+2, sub 0)   + PV(tile+1, sub 1)
            bar(); seg_v(I0{}, I0{}, tile + 2);                        // softmax(tile+2,0), issue V(tile+3)
        }
        // here: tile is even, sub-step (tile, 0) is done up to its softmax; 1 or 3 sub-steps remain
        bar(); seg_m(I1{}, I0{}, I1{}, I1{}, I0{}, I0{});
        bar(); seg_v(I0{}, I1{}, tile);
        if (2 * tile + 2 < n_sub) {
            advance(tile);
            bar(); seg_m(I1{}, I1{}, I0{}, I1{}, I0{}, I1{});
            bar(); seg_v(I1{}, I0{}, tile + 1);
            bar(); seg_m(I1{}, I1{}, I1{}, I1{}, I1{}, I0{});
            bar(); seg_v(I1{}, I1{}, tile + 1);
            bar(); seg_m(I0{}, I0{}, I0{}, I1{}, I1{}, I1{});        // PV(tile+1, sub 1)
        } else {
            bar(); seg_m(I0{}, I0{}, I0{}, I1{}, I0{}, I1{});        // PV(tile, sub 1)
        }
    }
    while (nbar < bar_total) bar();   // the shorter (or idle) group keeps the barrier count of the longer one

    // ---------------- epilogue ----------------
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (!(store_r || zero_r)) return;
        float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
        float Rv = 1.0f;
        const float* cp = nullptr;
        if (rectify) {
            const long rowi = (long)bh * a.NBv + qblk;
            Rv = a.R[rowi];
            cp = a.comp + rowi * D;
        }
        if (zero_r) inv = 0.0f;
        const float sc = inv * Rv;
        unsigned short* op = a.out + (long)b * a.osb + (long)h * a.osh + (long)grow * a.oss;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * dt + 8 * g + 4 * hh;
                float4 c4 = make_float4(0, 0, 0, 0);
                if (cp && !zero_r) c4 = *reinterpret_cast<const float4*>(cp + d0);
                const float v0 = o[dt][4 * g + 0] * sc + c4.x;
                const float v1 = o[dt][4 * g + 1] * sc + c4.y;
                const float v2 = o[dt][4 * g + 2] * sc + c4.z;
                const float v3 = o[dt][4 * g + 3] * sc + c4.w;
                uint2 pk;
                pk.x = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
                pk.y = (unsigned)E::from_f32(v2) | ((unsigned)E::from_f32(v3) << 16);
                *reinterpret_cast<uint2*>(op + d0) = pk;
            }
        }
    }
}

// launch hook used by rsa_attn.hip::launch_attn_pp.  `opt` = tuning value of "k5_pp": 1 = plain, 2 = with issue priority
// (product form); the other values select experiment builds (see the OPT bits above; 64 / 128 / 512 give wrong results).
template <int DD, typename TT>
static void launch_pp(const AttnArgs& a, dim3 grid, size_t lds_bytes, int opt, hipStream_t s) {
    constexpr int dm = DD == 128 ? 1 : 0;
    switch (opt) {
#define RSA_PP_CASE(V, O) case V: bsfwd_pp_kernel<DD, TT, (O)><<<grid, 512, lds_bytes, s>>>(a); break;
        RSA_PP_CASE(1, 256)
        RSA_PP_CASE(6, 4 + 2 + 256)
        RSA_PP_CASE(18, 16 * dm + 2 + 256)
        RSA_PP_CASE(34, 32 * dm + 2 + 256)
        RSA_PP_CASE(66, 64 + 2 + 256)
        RSA_PP_CASE(130, 128 + 2 + 256)
        RSA_PP_CASE(194, 64 + 128 + 2 + 256)
        RSA_PP_CASE(514, 512 + 2 + 256)
        RSA_PP_CASE(642, 512 + 128 + 2 + 256)
#undef RSA_PP_CASE
        default: bsfwd_pp_kernel<DD, TT, 2 + 256><<<grid, 512, lds_bytes, s>>>(a); break;
    }
}

int rsa_launch_bsfwd_pp(const AttnArgs& a, dim3 grid, size_t lds_bytes, int D, int dtype, int opt, hipStream_t s) {
    if (D == 128) {
        if (dtype == RSA_BF16) launch_pp<128, bf16_tag>(a, grid, lds_bytes, opt, s);
        else launch_pp<128, fp16_tag>(a, grid, lds_bytes, opt, s);
    } else {
        if (dtype == RSA_BF16) launch_pp<64, bf16_tag>(a, grid, lds_bytes, opt, s);
        else launch_pp<64, fp16_tag>(a, grid, lds_bytes, opt, s);
    }
    return rsa_launch_status();
}
