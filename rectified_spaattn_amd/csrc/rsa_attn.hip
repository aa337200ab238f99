// K5: block-sparse flash attention forward for gfx950 (MI355X), with the rectification epilogue fused.
//
// One workgroup (4 waves, 256 threads) owns one 128-row query block; wave w owns rows 32w..32w+31.
// Keys/values stream through LDS in 64-key tiles (two per kept 128-key block), double-buffered, staged
// through registers (global loads for tile t+1 are issued before tile t's MFMAs and written to LDS after
// the next barrier).  Both GEMMs run on v_mfma_f32_32x32x16_{bf16,f16} in the "key on the register, query
// row on the lane" orientation:
//      S^T[key][q]  = K . Q^T      A = K rows (ds_read_b128 from an XOR-swizzled row-major tile), B = Q (registers)
//      O^T[d][q]   += V^T . P^T    A = V^T (ds_read_b64_tr_b16 transposing reads), B = P^T = the S^T accumulator
//                                      converted in place (no LDS round trip, no cross-lane traffic)
// so the online-softmax state (m, l, rescale factor) of a query row lives on one lane (pair), and the only
// cross-lane operation per tile is one v_permlane32_swap for the row max.
//
// Semantics kept from the reference kernel (rectified_hunyuan_attn.py:15-105): Q is pre-multiplied by
// sm_scale*log2(e) and rounded to the input dtype (:61-62), P is rounded to the input dtype before PV (:97),
// fp32 softmax statistics and accumulators, kv columns outside the row's range are -inf (:86-87), rows
// beyond the sequence are not stored (:105).  Added: per-row kv ranges (the two-segment varlen semantics of
// the flash call, attn.py:107-120), a NaN-free fully-masked-tile path, the fused O*R+comp epilogue
// (hunyuan :365) and a strided [B,S,H,D] store (hunyuan :383-387).
#include "rsa_attn.h"

// OPT bits (tuning experiments, selected at launch by rsa_set_tuning("k5_opt", bits)):
//   2 deferred max (skip the O rescale while no row max of the wave grows by more than 2^8)
//   1: s_setprio 2 during the softmax phase (with 32: until the end of the PV phase)
//   8 / 4 / 12: K fragment reads software-pipelined 4 / 2 / 3 k-steps ahead of the QK^T MFMAs
//
// Staging: K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) issued from
// inline asm so that hipcc neither counts them nor drains them (vmcnt(0)) in front of the current tile's LDS
// reads; the only wait is the hand-placed vmcnt(0) + barrier at the top of the next iteration.  The LDS image
// is lane-linear, so the XOR swizzle is applied to the per-lane SOURCE chunk (same involution as tile_off on the
// read side).  Per-lane source offsets are tile-invariant 32-bit values; the tile only moves a scalar base.
template <int D, typename Tag, int OPT>
__global__ __launch_bounds__(256, 2) void bsfwd_kernel(AttnArgs a) {
    constexpr int KS = D / 16;             // k-steps of QK^T
    constexpr int DT = D / 32;             // 32-wide d tiles of O^T
    constexpr int CHR = D / 8;             // 16-byte chunks per row
    constexpr int NST = 64 * CHR / 256;    // 1-KiB pieces per wave per tile operand (4 or 2)
    constexpr int RPI = 1024 / (D * 2);    // rows per piece (4 or 8)
    constexpr int TILE_BYTES = 64 * D * 2;
    using E = Elem<Tag>;
    // one LDS object: [K0 V0 K1 V1 | kept-block list (u16)]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short* lds_list = reinterpret_cast<unsigned short*>(lds + 4 * TILE_BYTES);

    // ---------------- work mapping (heavy dense rows first; visual rows XCD-contiguous) ----------------
    int bh, qblk;
    {
        const int bid = blockIdx.x;
        if (bid < a.n_heavy_pad) {
            const int ntq = a.NQB - a.NBv;
            if (ntq <= 0 || bid >= a.BH * ntq) return;
            bh = bid / ntq;
            qblk = a.NBv + bid % ntq;
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
    const int grow = qblk * RSA_BLOCK + 32 * wv + r;  // this lane's global query row

    // ---------------- per-row plan ----------------
    int lo_r = 0, hi_r = 0;
    bool store_r = false, zero_r = false;
    int n_items, first_blk = 0, lo_max, hi_min, hi_max;
    const int32_t* list = nullptr;
    bool rectify = false;
    if (a.mode == MODE_SPARSE) {
        if (qblk < a.NBv) {
            hi_r = a.kv_valid;
            store_r = grow < a.Sq;
            const long rowi = (long)bh * a.NBv + qblk;
            list = a.cols + rowi * a.NB_total;
            n_items = a.counts[rowi];
            lo_max = 0; hi_min = hi_max = a.kv_valid;
            rectify = a.R != nullptr;
        } else {
            hi_r = a.kv_text_valid;
            store_r = grow < a.q_text_end;
            zero_r = !store_r && grow < a.Sq;
            n_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
            lo_max = 0; hi_min = hi_max = a.kv_text_valid;
        }
    } else {
        const int row0 = qblk * RSA_BLOCK, row1 = row0 + RSA_BLOCK;
        if (grow < a.q_split) { lo_r = 0; hi_r = a.kv_split; } else { lo_r = a.kv_split; hi_r = a.Sk; }
        store_r = grow < a.Sq;
        int lo_min;
        if (row1 <= a.q_split) { lo_min = 0; lo_max = 0; hi_min = hi_max = a.kv_split; }
        else if (row0 >= a.q_split) { lo_min = lo_max = a.kv_split; hi_min = hi_max = a.Sk; }
        else { lo_min = 0; lo_max = a.kv_split; hi_min = a.kv_split; hi_max = a.Sk; }
        first_blk = lo_min / RSA_BLOCK;
        n_items = (hi_max + RSA_BLOCK - 1) / RSA_BLOCK - first_blk;
        if (hi_max <= lo_min) n_items = 0;
    }
    n_items = __builtin_amdgcn_readfirstlane(n_items);
    // kept-block list -> LDS (u16), so the per-tile block index is an LDS broadcast read, not a dependent
    // global load at the head of every iteration
    const bool use_list = list != nullptr;
    if (use_list) {
        for (int i = t; i < n_items; i += 256) lds_list[i] = (unsigned short)list[i];
        __syncthreads();
    }
    auto blk_of = [&](int item) -> int { return use_list ? (int)lds_list[item] : first_blk + item; };
    int n_tiles = 2 * n_items;
    if (n_items > 0) {
        const int last_blk = blk_of(n_items - 1);
        if (last_blk * RSA_BLOCK + 64 >= hi_max) n_tiles -= 1;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    const int kv_limit = hi_max < a.Sk ? hi_max : a.Sk;  // rows >= this are never read (clamped to the last one)

    // ---------------- Q fragments: B operand, lane (r,hh) holds Q'[row][16ks + 8hh + 0..7] ----------------
    s16x8 qf[KS];
    {
        const unsigned short* qp = a.q + (long)b * a.qsb + (long)h * a.qsh + (long)grow * a.qss + 8 * hh;
        const bool qok = grow < a.Sq;
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

    // ---------------- LDS-DMA staging ----------------
    const unsigned char* kbase = reinterpret_cast<const unsigned char*>(a.k + (long)b * a.ksb + (long)h * a.ksh);
    const unsigned char* vbase = reinterpret_cast<const unsigned char*>(a.v + (long)b * a.vsb + (long)h * a.vsh);
    // piece pc = 4j + wv covers tile rows pc*RPI .. pc*RPI+RPI-1; this lane: row (wv*RPI + lane/CHR) of the
    // j-th group of 4*RPI rows, LDS chunk c = lane%CHR, source chunk g = c ^ f(row) (f does not depend on j)
    const int rowl = wv * RPI + lane / CHR;
    int gch;
    if constexpr (D == 128) gch = (lane % CHR) ^ (((rowl & 3) << 2) | ((rowl >> 2) & 3));
    else gch = (lane % CHR) ^ ((rowl >> 1) & 7);
    const unsigned voffk = (unsigned)(((long)rowl * a.kss + gch * 8) * 2);
    const unsigned voffv = (unsigned)(((long)rowl * a.vss + gch * 8) * 2);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const long kstep = (long)(4 * RPI) * a.kss * 2, vstep = (long)(4 * RPI) * a.vss * 2;  // bytes per j
    auto dma_tile = [&](int tile, int buf, int blk_v) -> int {
        const int blk = __builtin_amdgcn_readfirstlane(blk_v);
        const int key0 = blk * RSA_BLOCK + (tile & 1) * 64;
        const unsigned ldk = lds_base + buf * 2 * TILE_BYTES + wv * 1024;
        if (key0 + 64 <= kv_limit) {
            const unsigned char* kb = kbase + (long)key0 * a.kss * 2;
            const unsigned char* vb = vbase + (long)key0 * a.vss * 2;
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(voffk), "s"(kb + j * kstep), "s"(ldk + j * 4096) : "memory");
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(voffv), "s"(vb + j * vstep), "s"(ldk + TILE_BYTES + j * 4096) : "memory");
            }
        } else {  // last, partial tile: clamp the row (masked scores make P = 0 for the duplicates)
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                int krow = key0 + j * 4 * RPI + rowl;
                krow = krow < kv_limit ? krow : kv_limit - 1;
                const unsigned ok = (unsigned)(((long)krow * a.kss + gch * 8) * 2);
                const unsigned ov = (unsigned)(((long)krow * a.vss + gch * 8) * 2);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(ok), "s"(kbase), "s"(ldk + j * 4096) : "memory");
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(ov), "s"(vbase), "s"(ldk + TILE_BYTES + j * 4096) : "memory");
            }
        }
        return key0;
    };

    // ---------------- state ----------------
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // per-lane read addressing
    const int kswz = ((r & 3) << 2) | ((r >> 2) & 3);   // K row reads (D = 128)
    const int g4 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    // V^T transposing reads: per-lane byte offsets for (d tile, first/second 4-key group); the key group index
    // kk and the buffer only add immediates (tile_off is linear in multiples of 16 rows)
    int vrd[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const int ch = 4 * dt + 2 * (g4 & 1) + (tp >> 1);
        vrd[dt][0] = tile_off<D>(4 * hh + tq, ch) + 8 * (tp & 1);
        vrd[dt][1] = tile_off<D>(4 * hh + tq + 8, ch) + 8 * (tp & 1);
    }

    int key0_next = 0;
    int blk_pf = 0;  // block index of tile+1, read from LDS one iteration early
    if (n_tiles > 0) {
        key0_next = dma_tile(0, 0, blk_of(0));
        blk_pf = blk_of(0);  // tile 1 is the second half of item 0
    }
    // One tile of work; BUF is a compile-time constant (the tile loop is unrolled by two) so that every LDS
    // address is a loop-invariant VGPR plus an immediate offset -- no per-tile address arithmetic.
    auto tile_body = [&](auto BUFC, int tile) {
        constexpr int buf = decltype(BUFC)::value;
        const int key0 = key0_next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile `tile` have landed
        __syncthreads();                                  // ... and everybody else's; buffer buf^1 is free again
        if (tile + 1 < n_tiles) key0_next = dma_tile(tile + 1, buf ^ 1, blk_pf);
        {
            const int it2 = (tile + 2) >> 1;
            blk_pf = blk_of(it2 < n_items ? it2 : n_items - 1);
        }
        const unsigned char* kt_ = lds + buf * 2 * TILE_BYTES;
        const unsigned char* vt_ = kt_ + TILE_BYTES;

        // ---- S^T = K . Q^T  (two 32-key sub-tiles) ----
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.0f; s1[i] = 0.0f; }
        auto k_off = [&](int ks, int sub) {
            if constexpr (D == 128) return (32 * sub + r) * 256 + (((2 * ks + hh) ^ kswz) << 4);
            else return tile_off<D>(32 * sub + r, 2 * ks + hh);
        };
        if constexpr (OPT & 12) {
            constexpr int PD = (OPT & 8) ? (KS < 4 ? KS : 4) : ((OPT & 4) ? 2 : 3);  // prefetch depth in k-steps
            s16x8 fa[PD], fb[PD];
#pragma unroll
            for (int ks = 0; ks < PD; ++ks) {
                fa[ks] = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, 0));
                fb[ks] = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, 1));
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s0 = E::mfma(fa[ks % PD], qf[ks], s0);
                s1 = E::mfma(fb[ks % PD], qf[ks], s1);
                if (ks + PD < KS) {
                    fa[ks % PD] = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks + PD, 0));
                    fb[ks % PD] = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks + PD, 1));
                }
            }
            // pin the interleave: 2*PD reads up front, then {2 MFMA, 2 reads} per k-step (hipcc otherwise sinks
            // every read pair right in front of its MFMAs and exposes the LDS latency eight times per tile)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (ks + PD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const s16x8 a0 = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, 0));
                const s16x8 a1 = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, 1));
                s0 = E::mfma(a0, qf[ks], s0);
                s1 = E::mfma(a1, qf[ks], s1);
            }
        }
        // from here to the end of the tile (softmax VALU + PV) this wave gets issue priority over the co-resident
        // wave of the other workgroup, whose QK^T MFMA burst needs one issue slot per 32 cycles only (+1.5 % sparse,
        // +3 % dense measured; raising it for the softmax alone, or around the MFMA clusters, is negative)
        if constexpr (OPT & 1) __builtin_amdgcn_s_setprio(2);
        // ---- range mask (wave-uniform decision) ----
        if (key0 < lo_max || key0 + 64 > hi_min) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = key0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (kk < lo_r || kk >= hi_r) s0[i] = -INFINITY;
                if (kk + 32 < lo_r || kk + 32 >= hi_r) s1[i] = -INFINITY;
            }
        }
        // ---- online softmax (row = lane pair r, r+32) ----
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(s0[i], s1[i]));
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float m_use;
        bool rescale = true;
        if constexpr (OPT & 2) {
            // deferred max: keep the old reference max while no row of this wave grew by more than 2^8; P is then
            // bounded by 2^8 instead of 1 (fp32 l / O accumulators; bf16/fp16 P keeps its relative precision)
            const bool grow_r = mx > m_run + 8.0f;  // also true for the first tile (m_run = -inf, mx finite)
            rescale = __builtin_amdgcn_ballot_w64(grow_r) != 0ull;
        }
        if (rescale) {
            const float m_new = fmaxf(m_run, mx);
            m_use = (m_new == -INFINITY) ? 0.0f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        } else {
            m_use = (m_run == -INFINITY) ? 0.0f : m_run;
        }
        float psum = 0.0f;
        float p0[16], p1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            p0[i] = __builtin_amdgcn_exp2f(s0[i] - m_use);
            p1[i] = __builtin_amdgcn_exp2f(s1[i] - m_use);
            psum += p0[i] + p1[i];
        }
        l_run += psum;
        // P^T fragments: registers 8s..8s+7 of a 32-key sub-tile are k-step s of the B operand
        s16x8 pb[4];
        pb[0] = E::cvt8(p0);
        pb[1] = E::cvt8(p0 + 8);
        pb[2] = E::cvt8(p1);
        pb[3] = E::cvt8(p1 + 8);

        if constexpr ((OPT & 1) && !(OPT & 32)) __builtin_amdgcn_s_setprio(0);
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {        // kk = 2*kt + s : 16 keys each
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                // element j of lane half hh is key 16kk + 8(j>>2) + 4hh + (j&3); lane column d = 32dt + r
                const int offa = vrd[dt][0] + kk * 16 * D * 2;
                const int offb = vrd[dt][1] + kk * 16 * D * 2;
                const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vt_ + offa));
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3)))*)(vt_ + offb));
                const s16x8 av = __builtin_shufflevector(va, vb, 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = E::mfma(av, pb[kk], o[dt]);
            }
        }
        if constexpr ((OPT & 1) && (OPT & 32)) __builtin_amdgcn_s_setprio(0);
    };
    {
        int tile = 0;
        for (; tile + 1 < n_tiles; tile += 2) {
            tile_body(std::integral_constant<int, 0>{}, tile);
            tile_body(std::integral_constant<int, 1>{}, tile + 1);
        }
        if (tile < n_tiles) tile_body(std::integral_constant<int, 0>{}, tile);
    }

    // ---------------- epilogue ----------------
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (!(store_r || zero_r)) return;
    float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    float Rv = 1.0f;
    const float* cp = nullptr;
    if (rectify) {
        const long rowi = (long)bh * a.NBv + qblk;
        Rv = a.R[rowi];
        cp = a.comp + rowi * D;
    }
    if (zero_r) { inv = 0.0f; }
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

// =====================================================================================================
// host side
// =====================================================================================================
static int g_k5_opt = 37;

// Tuning / diagnostics hook (not part of the data path): "k5_opt" selects the K5 variant bits.
extern "C" int rsa_set_tuning(const char* key, int value) {
    if (!key) return RSA_ERR_BAD_ARG;
    if (strcmp(key, "k5_opt") == 0) { g_k5_opt = value; return RSA_OK; }
    return RSA_ERR_BAD_ARG;
}

static int launch_attn(AttnArgs& a, int BH, int D, int dtype, hipStream_t s) {
    const int ntq = a.NQB - a.NBv;
    const int n_heavy = ntq > 0 ? BH * ntq : 0;
    a.BH = BH;
    a.n_heavy_pad = (n_heavy + 7) & ~7;
    a.NBp = (a.NBv + 7) & ~7;
    const long nblocks = (long)a.n_heavy_pad + (long)BH * a.NBp;
    if (nblocks <= 0) return RSA_OK;
    if (nblocks > 0x7FFFFFFF) return RSA_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nblocks);
    if (a.NB_total > 8192) return RSA_ERR_UNSUPPORTED;  // kept list lives in LDS as u16, 16 KiB max
    const int opt = g_k5_opt;
    const size_t lds_bytes = (size_t)4 * 64 * D * 2 + (((size_t)a.NB_total * 2 + 15) & ~(size_t)15);
#define RSA_LAUNCH(DD, TT, OO) bsfwd_kernel<DD, TT, OO><<<grid, 256, lds_bytes, s>>>(a)
    if (D == 128 && dtype == RSA_BF16) {
        switch (opt) {
            case 0: RSA_LAUNCH(128, bf16_tag, 0); break;
            case 2: RSA_LAUNCH(128, bf16_tag, 2); break;
            case 4: RSA_LAUNCH(128, bf16_tag, 4); break;
            case 8: RSA_LAUNCH(128, bf16_tag, 8); break;
            default: RSA_LAUNCH(128, bf16_tag, 37); break;
        }
    } else if (D == 128) {
        RSA_LAUNCH(128, fp16_tag, 37);
    } else if (dtype == RSA_BF16) {
        RSA_LAUNCH(64, bf16_tag, 37);
    } else {
        RSA_LAUNCH(64, fp16_tag, 37);
    }
#undef RSA_LAUNCH
    return rsa_launch_status();
}

static void fill_qkv(AttnArgs& a, const rsa_tensor4& q, const rsa_tensor4& k, const rsa_tensor4& v,
                     const rsa_out4& out) {
    a.q = static_cast<const unsigned short*>(q.ptr); a.qsb = q.stride_b; a.qsh = q.stride_h; a.qss = q.stride_s;
    a.k = static_cast<const unsigned short*>(k.ptr); a.ksb = k.stride_b; a.ksh = k.stride_h; a.kss = k.stride_s;
    a.v = static_cast<const unsigned short*>(v.ptr); a.vsb = v.stride_b; a.vsh = v.stride_h; a.vss = v.stride_s;
    a.out = static_cast<unsigned short*>(out.ptr); a.osb = out.stride_b; a.osh = out.stride_h; a.oss = out.stride_s;
}

static int check_out(const rsa_out4& o) {
    if (!o.ptr || (reinterpret_cast<uintptr_t>(o.ptr) & 7)) return RSA_ERR_BAD_ARG;
    if ((o.stride_b % 4) || (o.stride_h % 4) || (o.stride_s % 4)) return RSA_ERR_BAD_ARG;  // 8-byte stores
    return RSA_OK;
}

extern "C" int rsa_block_sparse_fwd(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                    const rsa_buffers* buf, rsa_out4 out, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v)) ||
        (st = check_out(out)))
        return st;
    if (!buf || (l->NBv > 0 && (!buf->cols || !buf->counts))) return RSA_ERR_BAD_ARG;
    if ((buf->R == nullptr) != (buf->comp == nullptr)) return RSA_ERR_BAD_ARG;
    AttnArgs a;
    fill_qkv(a, q, k, v, out);
    a.cols = buf->cols; a.counts = buf->counts; a.R = buf->R; a.comp = buf->comp;
    a.mode = MODE_SPARSE; a.H = l->H; a.Sq = l->S; a.Sk = l->S;
    a.NBv = l->NBv; a.NQB = l->NB_total; a.NB_total = l->NB_total;
    a.kv_valid = l->kv_valid; a.kv_text_valid = l->kv_text_valid;
    a.q_text_end = l->NBv * RSA_BLOCK + l->q_text_valid;
    a.q_split = 0; a.kv_split = 0;
    a.qk_scale = (float)((1.0 / sqrt((double)l->D)) * 1.44269504);  // sm_scale * 1.44269504 (hunyuan :145)
    return launch_attn(a, l->B * l->H, l->D, l->dtype, static_cast<hipStream_t>(stream));
}

extern "C" int rsa_dense_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k,
                             rsa_tensor4 v, int q_split, int kv_split, rsa_out4 out, void* stream) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return RSA_ERR_BAD_ARG;
    if (D != 64 && D != 128) return RSA_ERR_UNSUPPORTED;
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    if (q_split < 0 || q_split > Sq || kv_split < 0 || kv_split > Sk) return RSA_ERR_BAD_ARG;
    int st;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v)) ||
        (st = check_out(out)))
        return st;
    AttnArgs a;
    fill_qkv(a, q, k, v, out);
    a.cols = nullptr; a.counts = nullptr; a.R = nullptr; a.comp = nullptr;
    a.mode = MODE_DENSE; a.H = H; a.Sq = Sq; a.Sk = Sk;
    a.NQB = (Sq + RSA_BLOCK - 1) / RSA_BLOCK; a.NBv = a.NQB; a.NB_total = (Sk + RSA_BLOCK - 1) / RSA_BLOCK;
    a.kv_valid = Sk; a.kv_text_valid = Sk; a.q_text_end = 0;
    a.q_split = q_split; a.kv_split = kv_split;
    a.qk_scale = (float)((1.0 / sqrt((double)D)) * 1.44269504);
    return launch_attn(a, B * H, D, dtype, static_cast<hipStream_t>(stream));
}

extern "C" int rsa_rectified_attention(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                       const uint8_t* neighbor, int top_k, float p_remain, void* workspace,
                                       size_t workspace_bytes, rsa_out4 out, void* stream) {
    rsa_buffers buf;
    int st = rsa_carve_workspace(l, workspace, workspace_bytes, &buf);
    if (st != RSA_OK) return st;
    if ((st = rsa_pool_stats(l, q, k, v, &buf, stream))) return st;
    if ((st = rsa_pooled_scores(l, k, &buf, stream))) return st;
    if ((st = rsa_select_mask(l, neighbor, top_k, p_remain, &buf, stream))) return st;
    if ((st = rsa_compensation(l, &buf, stream))) return st;
    return rsa_block_sparse_fwd(l, q, k, v, &buf, out, stream);
}
