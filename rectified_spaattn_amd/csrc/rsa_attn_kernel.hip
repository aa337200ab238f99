// K5: block-sparse flash attention forward for gfx950 (MI355X) with the rectification epilogue fused.
//
// One workgroup (4 waves, 256 threads, two workgroups per CU) owns one 128-row query block; wave w owns rows
// 32w..32w+31.  Both GEMMs run on v_mfma_f32_32x32x16_{bf16,f16} in the "key on the register, query row on
// the lane" orientation:
//      S^T[key][q]  = K . Q^T      A = K rows (ds_read_b128 from an XOR-swizzled row-major tile), B = Q (registers)
//      O^T[d][q]   += V^T . P^T    A = V^T (ds_read_b64_tr_b16 transposing reads), B = P^T = the S^T accumulator
//                                      converted in place (no LDS round trip, no cross-lane traffic)
// so the softmax state (m, l) of a query row lives on one lane pair and the only cross-lane operation per
// 32 keys is one v_permlane32_swap for the row max.
//
// Software pipeline (per wave, at 32-key granularity, S double-buffered in registers, loop unrolled by two
// 64-key tiles so every LDS address is a loop-invariant VGPR plus an immediate):
//      MFMA stream:  S(u+1)^T = K(u+1) . Q^T      then   O^T += V(u)^T . P(u)^T
//      VALU stream:  P(u) = exp2(S(u) - m) (+ row sums, 2-byte packing)   then   row max of S(u+1)
// The block is branch-free; hipcc emits the VALU part first and the MFMA cluster after it, and the two
// co-resident waves of a SIMD (one per workgroup) alternate between the two (measured: pinning a fine
// per-MFMA interleave with sched_group_barrier is 3.5 % slower).  The running max is deferred: the reference
// max only moves when some row's max grew by more than 2^8 since it was set (P <= 2^8: same relative precision
// in bf16/fp16 P, fp32 accumulators); that rare rescale and the boundary-tile mask sit in branches at the head
// of the sub-step, outside the pipelined block.
//
// Staging: K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) issued
// from inline asm so that hipcc neither counts nor drains them; LDS = [K0 K1 V0 V1 | kept list (u16)].  At the
// head of sub-step (t,0) V(t+1) is issued into V(t-1)'s slot, at the head of (t,1) K(t+2) into K(t)'s slot,
// each behind a counted vmcnt (the group issued half a tile ago stays in flight) + barrier; every tile has a
// full tile time to land.  The LDS image is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE
// chunk (same involution as tile_off on the read side); per-lane source offsets are tile-invariant 32-bit
// values and the tile only moves a scalar base.
//
// Semantics kept from the reference kernel (rectified_hunyuan_attn.py:15-105): Q is pre-multiplied by
// sm_scale*log2(e) and rounded to the input dtype (:61-62), P is rounded to the input dtype before PV (:97),
// fp32 softmax statistics and accumulators, kv columns outside the row's range are -inf (:86-87), rows
// beyond the sequence are not stored (:105).  Added: per-row kv ranges (the two-segment varlen semantics of
// the flash call, attn.py:107-120), a NaN-free fully-masked path, the fused O*R+comp epilogue (hunyuan :365)
// and a strided [B,S,H,D] store (hunyuan :383-387).
//
// (Forms measured and NOT kept in this library -- a ping-pong 8-wave kernel, paired 256-row workgroups over union lists,
// persistent workgroups, a 256-row dense tile, non-temporal K/V loads, in-kernel stamps: commit c37b5bb holds their code,
// profiles/r02_experiments.md their numbers.)
#include "rsa_attn.h"

// The pipelined block runs with issue priority 2 and LLVM's small-GEMM MFMA/DS interleave (iglp_opt(0): +1.7 % sparse,
// +2.7 % dense 16k; strategies 1-3 measured -0.5 ... -3 %; a per-MFMA interleave pinned with sched_group_barrier -3.5 %).
// WIDE: 16-byte output stores after a permlane32_swap regroup (needs 16-byte aligned output rows), else 8-byte stores.
template <int D, typename Tag, bool WIDE>
__global__ __launch_bounds__(256, 2) void bsfwd_kernel(AttnArgs a) {
    constexpr int NW = 4, QT = 1;                // 4 waves x 32 query rows
    constexpr int QROWS = 32 * NW * QT;          // query rows per workgroup
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    constexpr int CHR = D / 8;
    constexpr int RPI = 1024 / (D * 2);     // rows per 1-KiB piece
    constexpr int TILE_BYTES = 64 * D * 2;
    constexpr int NPC = TILE_BYTES / 1024 / NW;  // 1-KiB pieces per wave per tile operand
    constexpr int PG = NW >= 4 ? 1 : 4 / NW;    // pieces of each group of 4 that this wave moves (1 or 2)
    constexpr int TEAMS = NW >= 4 ? NW / 4 : 1; // wave teams that alternate over the groups of 4 pieces
    using E = Elem<Tag>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short* lds_list = reinterpret_cast<unsigned short*>(lds + 4 * TILE_BYTES);

    // (Persistent workgroups -- 2 per CU striding over the work items -- were measured: a fresh workgroup costs ~20k cycles of
    // dispatch, but with two workgroups per CU the partner runs faster meanwhile, and strided workgroups fall into lock step
    // (all prologues and epilogues at once): 18.1 vs 17.6 ms at 24 heads, 2.26 vs 2.30 ms at 3 heads; not adopted.)
    const int work = blockIdx.x;
    // ---------------- work mapping: dense text-row blocks first, then the sparse blocks chunked per XCD ----------------
    int bh, qblk, tsp = 0;   // tsp: which part of a text block's key range this workgroup walks
    {
        const int bid = work;
        if (bid < a.n_heavy_pad) {
            const int ntq = a.NQB - a.NBv;
            const int per_bh = ntq * a.tsplit;      // text blocks x key-range splits (tsplit = 1: no split)
            if (ntq <= 0 || bid >= a.BH * per_bh) return;
            bh = bid / per_bh;
            const int rem = bid % per_bh;
            qblk = a.NBv + rem / a.tsplit;
            tsp = rem % a.tsplit;
        } else {
            const int v = bid - a.n_heavy_pad;
            bh = v / a.NBp;
            const int j = v % a.NBp;
            const int chunk = a.NBp >> 3;
            qblk = (j & 7) * chunk + (j >> 3);
            if (qblk >= a.NBv) return;
        }
    }
    const int b = bh / a.H, h = bh % a.H;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int qb = qblk;
    int grow[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) grow[qt] = qblk * QROWS + 32 * QT * wv + 32 * qt + r;

    // ---------------- per-row plan ----------------
    int lo_r[QT], hi_r[QT];
    bool store_r[QT], zero_r[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { lo_r[qt] = 0; hi_r[qt] = 0; store_r[qt] = false; zero_r[qt] = false; }
    int n_items, first_blk = 0, lo_max, hi_min, hi_max;
    const int32_t* list = nullptr;
    bool rectify = false;
    if (a.mode == MODE_SPARSE) {
        if (qblk < a.NBv) {
            const long rowi = (long)bh * a.NBv + qblk;
            list = a.cols + rowi * a.NB_total;
            n_items = a.counts[rowi];
            lo_max = 0; hi_min = hi_max = a.kv_valid;
            rectify = a.R != nullptr;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) { hi_r[qt] = a.kv_valid; store_r[qt] = grow[qt] < a.Sq; }
        } else {
            n_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
            if (a.tsplit > 1) {   // split-KV: this workgroup's slice of the key blocks
                first_blk = tsp * a.tper;
                n_items = n_items - first_blk < a.tper ? n_items - first_blk : a.tper;
                if (n_items < 0) n_items = 0;
            }
            lo_max = 0; hi_min = hi_max = a.kv_text_valid;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                hi_r[qt] = a.kv_text_valid;
                store_r[qt] = grow[qt] < a.q_text_end;
                zero_r[qt] = !store_r[qt] && grow[qt] < a.Sq;
            }
        }
    } else {
        const int row0 = qblk * QROWS, row1 = row0 + QROWS;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (grow[qt] < a.q_split) { lo_r[qt] = 0; hi_r[qt] = a.kv_split; }
            else { lo_r[qt] = a.kv_split; hi_r[qt] = a.Sk; }
            store_r[qt] = grow[qt] < a.Sq;
        }
        int lo_min;
        if (row1 <= a.q_split) { lo_min = 0; lo_max = 0; hi_min = hi_max = a.kv_split; }
        else if (row0 >= a.q_split) { lo_min = lo_max = a.kv_split; hi_min = hi_max = a.Sk; }
        else { lo_min = 0; lo_max = a.kv_split; hi_min = a.kv_split; hi_max = a.Sk; }
        first_blk = lo_min / RSA_BLOCK;
        n_items = (hi_max + RSA_BLOCK - 1) / RSA_BLOCK - first_blk;
        if (hi_max <= lo_min) n_items = 0;
    }
    n_items = __builtin_amdgcn_readfirstlane(n_items);
    const bool use_list = list != nullptr;
    if (use_list) {
        for (int i = t; i < n_items; i += 64 * NW) lds_list[i] = (unsigned short)list[i];
        __syncthreads();
    }
    auto blk_of = [&](int item) -> int {
        return use_list ? (int)lds_list[item] : first_blk + item;
    };
    int n_tiles = 2 * n_items;
    if (n_items > 0) {
        const int last_blk = blk_of(n_items - 1);
        if (last_blk * RSA_BLOCK + 64 >= hi_max) n_tiles -= 1;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    const int kv_limit = hi_max < a.Sk ? hi_max : a.Sk;
    auto key0_of = [&](int tile) -> int {  // first key of tile `tile` (clamped index: callers guard tile < n_tiles)
        const int it = tile >> 1;
        const int blk = __builtin_amdgcn_readfirstlane(blk_of(it < n_items ? it : (n_items > 0 ? n_items - 1 : 0)));
        return blk * RSA_BLOCK + (tile & 1) * 64;
    };
    // ---------------- Q fragments (B operand), two 32-row sub-tiles ----------------
    s16x8 qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const unsigned short* qp = a.q + (long)b * a.qsb + (long)h * a.qsh + (long)grow[qt] * a.qss + 8 * hh;
        const bool qok = grow[qt] < a.Sq;
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
            qf[qt][ks] = E::cvt8(f);
        }
    }

    // ---------------- LDS-DMA staging ----------------
    const unsigned char* kbase = reinterpret_cast<const unsigned char*>(a.k + (long)b * a.ksb + (long)h * a.ksh);
    const unsigned char* vbase = reinterpret_cast<const unsigned char*>(a.v + (long)b * a.vsb + (long)h * a.vsh);
    // piece pc = 4*grp + PG*wslot + (j%PG), grp = TEAMS*(j/PG) + team, j = 0..NPC-1: tile rows pc*RPI .. +RPI-1.  The
    // source-chunk swizzle depends on pc & 3 = PG*wslot + (j%PG): PG per-lane offsets per operand.
    const int wslot = NW >= 4 ? (wv & 3) : wv, team = NW >= 4 ? (wv >> 2) : 0;
    const int rsub = lane / CHR, cl = lane % CHR;
    unsigned voffk[PG], voffv[PG];
    int gsw[PG];
#pragma unroll
    for (int par = 0; par < PG; ++par) {
        const int rowl = (PG * wslot + par) * RPI + rsub;  // row inside the first group of 4 pieces
        if constexpr (D == 128) gsw[par] = cl ^ (((rowl & 3) << 2) | ((rowl >> 2) & 3));
        else gsw[par] = cl ^ ((rowl >> 1) & 7);
        voffk[par] = (unsigned)(((long)rowl * a.kss + gsw[par] * 8) * 2);
        voffv[par] = (unsigned)(((long)rowl * a.vss + gsw[par] * 8) * 2);
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const long kstep = (long)(4 * RPI) * a.kss * 2, vstep = (long)(4 * RPI) * a.vss * 2;  // bytes per 4 pieces
    // is_v: 0 = K tile into K slot `slot`, 1 = V tile into V slot `slot`
    auto dma = [&](int is_v, int key0, int slot) {
        const unsigned ld0 = lds_base + (is_v ? 2 : 0) * TILE_BYTES + slot * TILE_BYTES + (PG * wslot) * 1024;
        const unsigned char* base = is_v ? vbase : kbase;
        const long ss = is_v ? a.vss : a.kss;
        if (key0 + 64 <= kv_limit) {
            const unsigned char* tb = base + (long)key0 * ss * 2;
            const long step = is_v ? vstep : kstep;
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                const unsigned vo = is_v ? voffv[j % PG] : voffk[j % PG];
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(vo), "s"(tb + (TEAMS * (j / PG) + team) * step),
                                "s"(ld0 + (TEAMS * (j / PG) + team) * 4096 + (j % PG) * 1024)
                             : "memory");
            }
        } else {
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                const int rowl = (PG * wslot + (j % PG)) * RPI + rsub;
                int krow = key0 + (TEAMS * (j / PG) + team) * 4 * RPI + rowl;
                krow = krow < kv_limit ? krow : kv_limit - 1;
                const unsigned vo = (unsigned)(((long)krow * ss + gsw[j % PG] * 8) * 2);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(vo), "s"(base), "s"(ld0 + (TEAMS * (j / PG) + team) * 4096 + (j % PG) * 1024)
                             : "memory");
            }
        }
    };

    // ---------------- state ----------------
    f32x16 o[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][qt][i] = 0.0f;
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.0f; }

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

    // S^T[qt] (32 keys x 32 rows per qt) = K[sub-tile SUB of K slot] . Q^T ; one K fragment feeds both query sub-tiles
    auto qk_sub = [&](auto KSLOT, auto SUB, f32x16 (&S)[QT]) {
        constexpr int slot = decltype(KSLOT)::value, sub = decltype(SUB)::value;
        const unsigned char* kt_ = lds + slot * TILE_BYTES;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int i = 0; i < 16; ++i) S[qt][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 a0 = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, sub));
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) S[qt] = E::mfma(a0, qf[qt][ks], S[qt]);
        }
    };
    auto rowmax_sub = [&](const f32x16 (&S)[QT], float (&mx)[QT]) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float m = S[qt][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = fmaxf(m, S[qt][i]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            mx[qt] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
    };
    auto apply_mask_sub = [&](f32x16 (&S)[QT], int key_first) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = key_first + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (kk < lo_r[qt] || kk >= hi_r[qt]) S[qt][i] = -INFINITY;
            }
    };

    int kq1 = 0, kq2 = 0;  // first keys of tile+1 / tile+2 (fetched from LDS ahead of use)

    // One pipelined sub-step u = 2*tile + SUB: consumes S_cur (scores of 32 keys, row max in mx_cur), produces
    // S_nxt / mx_nxt for sub-step u+1.  VS = slot parity of `tile` (its K and V slots).
    //   SUB = 0: head issues V(tile+1);  next scores = K(tile) sub-tile 1
    //   SUB = 1: head issues K(tile+2);  next scores = K(tile+1) sub-tile 0
    auto step = [&](auto VS, auto SUB, int tile, int key0, f32x16 (&S_cur)[QT], float (&mx_cur)[QT],
                    f32x16 (&S_nxt)[QT], float (&mx_nxt)[QT]) {
        constexpr int vs = decltype(VS)::value, sub = decltype(SUB)::value;
        // the newest DMA group (issued half a tile ago) may stay in flight; the one issued a tile ago must land
        if (tile + 1 < n_tiles) {
            if constexpr (NPC == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NPC == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NPC == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if constexpr (sub == 0) {
            if (tile + 1 < n_tiles) dma(1, kq1, vs ^ 1);   // V(tile+1) -> slot of V(tile-1)
        } else {
            if (tile + 2 < n_tiles) dma(0, kq2, vs);       // K(tile+2) -> slot of K(tile)
        }
        // ---- head (rare branches): boundary mask, deferred rescale ----
        const int kfirst = key0 + 32 * sub;
        if ((kfirst < lo_max || kfirst + 32 > hi_min)) {
            apply_mask_sub(S_cur, kfirst);
            rowmax_sub(S_cur, mx_cur);
        }
        bool grow_any = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) grow_any |= mx_cur[qt] > m_run[qt] + 8.0f;
        if (__builtin_amdgcn_ballot_w64(grow_any) != 0ull) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float m_new = fmaxf(m_run[qt], mx_cur[qt]);
                const float mu = (m_new == -INFINITY) ? 0.0f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - mu);
                m_run[qt] = m_new;
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][qt][i] *= alpha;
            }
        }
        float m_use[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) m_use[qt] = (m_run[qt] == -INFINITY) ? 0.0f : m_run[qt];

        // ---- pipelined block (branch-free on purpose: the scheduler interleaves the MFMA and VALU streams) ----
        __builtin_amdgcn_s_setprio(2);
        __builtin_amdgcn_iglp_opt(0);  // LLVM's small-GEMM MFMA/DS interleave for this region
        if constexpr (sub == 0) qk_sub(std::integral_constant<int, vs>{}, std::integral_constant<int, 1>{}, S_nxt);
        else qk_sub(std::integral_constant<int, vs ^ 1>{}, std::integral_constant<int, 0>{}, S_nxt);
        s16x8 pb[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float ps = 0.0f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float pv8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pv8[i] = __builtin_amdgcn_exp2f(S_cur[qt][8 * hf + i] - m_use[qt]);
                    ps += pv8[i];
                }
                pb[qt][hf] = E::cvt8(pv8);
            }
            l_run[qt] += ps;
        }
        const unsigned char* vt_ = lds + (2 + vs) * TILE_BYTES;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int offa = vrd[dt][0] + (2 * sub + k2) * 16 * D * 2;
                const int offb = vrd[dt][1] + (2 * sub + k2) * 16 * D * 2;
                const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vt_ + offa));
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vt_ + offb));
                const s16x8 av = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[dt][qt] = E::mfma(av, pb[qt][k2], o[dt][qt]);
            }
        }
        rowmax_sub(S_nxt, mx_nxt);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---------------- prologue + main loop ----------------
    f32x16 SA[QT], SB[QT];
    float mxA[QT], mxB[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { mxA[qt] = -INFINITY; mxB[qt] = -INFINITY; }
    int key0 = 0;
    if (n_tiles > 0) {
        key0 = key0_of(0);
        kq1 = key0_of(1);
        kq2 = key0_of(2);
        dma(0, key0, 0);
        dma(1, key0, 0);
        if (n_tiles > 1) dma(0, kq1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        qk_sub(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, SA);
        rowmax_sub(SA, mxA);
    }
    auto advance = [&](int tile) {  // after finishing `tile`: shift the key queue, fetch tile+3's first key
        key0 = kq1;
        kq1 = kq2;
        kq2 = key0_of(tile + 3);
    };
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        int tile = 0;
        for (; tile + 1 < n_tiles; tile += 2) {
            step(I0{}, I0{}, tile, key0, SA, mxA, SB, mxB);
            step(I0{}, I1{}, tile, key0, SB, mxB, SA, mxA);
            advance(tile);
            step(I1{}, I0{}, tile + 1, key0, SA, mxA, SB, mxB);
            step(I1{}, I1{}, tile + 1, key0, SB, mxB, SA, mxA);
            advance(tile + 1);
        }
        if (tile < n_tiles) {
            step(I0{}, I0{}, tile, key0, SA, mxA, SB, mxB);
            step(I0{}, I1{}, tile, key0, SB, mxB, SA, mxA);
        }
    }

    // ---------------- epilogue ----------------
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[qt]), __float_as_uint(l_run[qt]),
                                                         false, false);
        const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        {
            if (a.mode == MODE_SPARSE && a.tsplit > 1 && qblk >= a.NBv) {
                // split-KV partial of a text block: unnormalised O (fp32), m (log2 domain) and l per row; the combine
                // kernel (rsa_attn.hip) merges the tsplit parts
                const int ntq = a.NQB - a.NBv;
                const int rowb = 32 * QT * wv + 32 * qt + r;
                float* pp = a.tpart + ((((long)bh * ntq + (qblk - a.NBv)) * a.tsplit + tsp) * RSA_BLOCK + rowb) * (D + 2);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d0 = 32 * dt + 8 * g + 4 * hh;
                        *reinterpret_cast<float2*>(pp + d0) = make_float2(o[dt][qt][4 * g + 0], o[dt][qt][4 * g + 1]);
                        *reinterpret_cast<float2*>(pp + d0 + 2) = make_float2(o[dt][qt][4 * g + 2], o[dt][qt][4 * g + 3]);
                    }
                if (hh == 0) *reinterpret_cast<float2*>(pp + D) = make_float2(m_run[qt], l_tot);
                continue;
            }
        }
        if (!(store_r[qt] || zero_r[qt])) continue;
        float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
        float Rv = 1.0f;
        const float* cp = nullptr;
        if (rectify) {
            const long rowi = (long)bh * a.NBv + qb;
            Rv = a.R[rowi];
            cp = a.comp + rowi * D;
        }
        if (zero_r[qt]) inv = 0.0f;
        const float sc = inv * Rv;
        unsigned short* op = a.out + (long)b * a.osb + (long)h * a.osh + (long)grow[qt] * a.oss;
        // O * sc + comp: one fma rounded to fp32, THEN the conversion to the storage type (what the oracle does).  Left alone,
        // hipcc folds fma + conversion into v_fma_mixlo_f16 (one rounding, straight to fp16) in one store form and not in the
        // other; the empty asm keeps the fp32 value, so every instantiation (4-wave, 8-wave, paired) writes the same bytes.
        auto fin = [&](float acc, float c) -> float {
            float rr = __builtin_fmaf(acc, sc, c);
            asm volatile("" : "+v"(rr));
            return rr;
        };
        if constexpr (WIDE) {
            // wide stores: lane (r, 0) holds d = 8g .. 8g+3 and lane (r, 1) d = 8g+4 .. 8g+7 of a 32-wide d tile; one
            // v_permlane32_swap per packed register pair regroups two g's so that each lane owns 8 consecutive d:
            // 8 stores of 16 B per lane instead of 16 of 8 B (the store tail of a row-per-lane epilogue is issue-bound)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint2 pk[2];
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int g = 2 * gp + gi;
                        const int d0 = 32 * dt + 8 * g + 4 * hh;
                        float4 c4 = make_float4(0, 0, 0, 0);
                        if (cp && !zero_r[qt]) c4 = *reinterpret_cast<const float4*>(cp + d0);
                        const float v0 = fin(o[dt][qt][4 * g + 0], c4.x);
                        const float v1 = fin(o[dt][qt][4 * g + 1], c4.y);
                        const float v2 = fin(o[dt][qt][4 * g + 2], c4.z);
                        const float v3 = fin(o[dt][qt][4 * g + 3], c4.w);
                        pk[gi].x = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
                        pk[gi].y = (unsigned)E::from_f32(v2) | ((unsigned)E::from_f32(v3) << 16);
                    }
                    // X' = [X.lower, Y.lower], Y' = [X.upper, Y.upper]  (X = pk[0], Y = pk[1])
                    const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    uint4 w4;
                    w4.x = sx[0]; w4.y = sy[0]; w4.z = sx[1]; w4.w = sy[1];
                    *reinterpret_cast<uint4*>(op + 32 * dt + 8 * (2 * gp + hh)) = w4;
                }
            }
            continue;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * dt + 8 * g + 4 * hh;
                float4 c4 = make_float4(0, 0, 0, 0);
                if (cp && !zero_r[qt]) c4 = *reinterpret_cast<const float4*>(cp + d0);
                const float v0 = fin(o[dt][qt][4 * g + 0], c4.x);
                const float v1 = fin(o[dt][qt][4 * g + 1], c4.y);
                const float v2 = fin(o[dt][qt][4 * g + 2], c4.z);
                const float v3 = fin(o[dt][qt][4 * g + 3], c4.w);
                uint2 pk;
                pk.x = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
                pk.y = (unsigned)E::from_f32(v2) | ((unsigned)E::from_f32(v3) << 16);
                *reinterpret_cast<uint2*>(op + d0) = pk;
            }
        }
    }
}

// launch hook used by rsa_attn.hip::launch_attn
int rsa_launch_bsfwd(const AttnArgs& a, dim3 grid, size_t lds_bytes, int D, int dtype, hipStream_t s) {
    // the 16-byte output stores need 16-byte aligned rows; anything else takes the 8-byte form
    const bool wide = !(((uintptr_t)a.out & 15) || ((a.osb | a.osh | a.oss) & 7));
#define RSA_K5(DD, TT) \
    do { \
        if (wide) bsfwd_kernel<DD, TT, true><<<grid, 256, lds_bytes, s>>>(a); \
        else bsfwd_kernel<DD, TT, false><<<grid, 256, lds_bytes, s>>>(a); \
    } while (0)
    if (D == 128) {
        if (dtype == RSA_BF16) RSA_K5(128, bf16_tag); else RSA_K5(128, fp16_tag);
    } else {
        if (dtype == RSA_BF16) RSA_K5(64, bf16_tag); else RSA_K5(64, fp16_tag);
    }
#undef RSA_K5
    return rsa_launch_status();
}
