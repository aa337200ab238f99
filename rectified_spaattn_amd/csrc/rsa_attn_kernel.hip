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
// That block is ONE hand-placed instruction stream (gen_k5_block.py -> rsa_attn_block.h: inline asm, every register
// pinned): round 2 showed that leaving it to hipcc makes the kernel's speed a matter of which schedule the compiler
// happens to emit (profiles/r03_k5_r1_vs_head.md: the same source went from read-ahead operand reads to one
// lgkmcnt(0) per MFMA between two rounds, -8 %).  The running max is deferred: the reference max only moves when some
// row's max grew by more than 2^8 since it was set (P <= 2^8: same relative precision in bf16/fp16 P, fp32
// accumulators); that rare rescale and the boundary-tile mask sit in branches in front of the block.
//
// Staging: K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) issued
// from inline asm so that hipcc neither counts nor drains them; LDS = [K0 K1 V0 V1 | kept list (u16)], 64-key tiles.  At the
// head of sub-step (t,0) V(t+1) is issued into V(t-1)'s slot, at the head of (t,1) K(t+2) into K(t)'s slot, each behind a
// counted vmcnt (the group issued half a tile ago stays in flight) + barrier; every tile has a full tile time to land.
// (One wait + barrier + issue point per 64-key TILE -- half the barriers, 8 pieces per issue point -- measured the same:
// 16.33 vs 16.23 ms, profiles/r03_k5_block.md; not kept.)  The LDS image is lane-linear, so the XOR swizzle is applied to
// the per-lane SOURCE chunk (same involution as tile_off on the read side); per-lane source offsets are tile-invariant
// 32-bit values and the tile only moves a scalar base.
//
// The chip runs this loop at its board power cap (1.39 kW at 1.72 GHz, tools/clock_probe.py), so cycles saved from stalls
// come back as a lower clock; what pays is fewer instructions per MFMA: the score chain starts from -m (a 16-register
// block, C operand of the first QK^T MFMA), so the accumulator is S - m and the softmax needs no subtraction.
//
// Semantics kept from the reference kernel (rectified_hunyuan_attn.py:15-105): Q is pre-multiplied by
// sm_scale*log2(e) and rounded to the input dtype (:61-62), P is rounded to the input dtype before PV (:97),
// fp32 softmax statistics and accumulators, kv columns outside the row's range are -inf (:86-87), rows
// beyond the sequence are not stored (:105).  Added: per-row kv ranges (the two-segment varlen semantics of
// the flash call, attn.py:107-120), a NaN-free fully-masked path, the fused O*R+comp epilogue (hunyuan :365)
// and a strided [B,S,H,D] store (hunyuan :383-387).
//
// (Forms measured and NOT kept in this library -- a ping-pong 8-wave kernel, paired 256-row workgroups over union lists,
// persistent workgroups, a 256-row dense tile, non-temporal K/V loads, the LDS-DMA pieces spread over the block's MFMA
// shadows, 16x16x32 MFMAs, one staging point per tile: commit c37b5bb / profiles/r02_experiments.md, profiles/r03_k5_block.md.)
#include "rsa_attn.h"
#include "rsa_attn_block.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// The hand-placed pipelined block (gen_k5_block.py): one asm statement per sub-step with every register pinned (register
// map and operand lists: rsa_attn_block.h, RSA_K5_OPS*; e.g. D = 128: O v[0:63], Q v[64:95], SA v[96:111], SB v[112:127],
// -m v[128:143]).  S_cur = SA on even sub-steps, SB on odd ones.
#define RSA_K5_PICK(NAME, DD, TT, OPS, CLOB) \
    do { \
        if constexpr (VS == 0 && SUB == 0) asm volatile(NAME##_##DD##_##TT##_V0_S0 OPS : CLOB, "memory"); \
        else if constexpr (VS == 0 && SUB == 1) asm volatile(NAME##_##DD##_##TT##_V0_S1 OPS : CLOB, "memory"); \
        else if constexpr (VS == 1 && SUB == 0) asm volatile(NAME##_##DD##_##TT##_V1_S0 OPS : CLOB, "memory"); \
        else asm volatile(NAME##_##DD##_##TT##_V1_S1 OPS : CLOB, "memory"); \
    } while (0)

// the product's block: S_cur / S_nxt hold S - m (nm = -m in 16 registers), no subtraction in the softmax
template <int D, typename Tag, int VS, int SUB, typename KA, typename VA>
__device__ __forceinline__ void k5_block_nm(f32x16 (&o)[D / 32], const s16x8 (&q)[D / 16], f32x16& S_cur, f32x16& S_nxt,
                                            const f32x16& nm, float& l, f32x4& lacc, const s16x8& onesv, float& mx, const KA& ka,
                                            const VA& va) {
    f32x16& SA = SUB == 0 ? S_cur : S_nxt;
    f32x16& SB = SUB == 0 ? S_nxt : S_cur;
    constexpr bool BF = std::is_same<Tag, bf16_tag>::value;
    if constexpr (D == 128) {
        if constexpr (BF) RSA_K5_PICK(RSA_K5_BLOCKN, 128, BF16, RSA_K5_OPSN_128, RSA_K5_CLOBBERN_128);
        else RSA_K5_PICK(RSA_K5_BLOCKN, 128, F16, RSA_K5_OPSN_128, RSA_K5_CLOBBERN_128);
    } else {
        if constexpr (BF) RSA_K5_PICK(RSA_K5_BLOCKN, 64, BF16, RSA_K5_OPSN_64, RSA_K5_CLOBBERN_64);
        else RSA_K5_PICK(RSA_K5_BLOCKN, 64, F16, RSA_K5_OPSN_64, RSA_K5_CLOBBERN_64);
    }
}
// the classic form (S, then S - m by v_sub: the compiled block's arithmetic, bit-identical to the block as hipcc emits it):
// the product at head dim 64, where it measures 4 % faster than the -m form (at 128 the -m form wins by 3.7 %)
// (head dim 64, round 5: the row sums ride the matrix pipe -- l lives in lacc, four registers with the lane's complete row sum of
// the rounded P, onesv is the ones operand of those products; l itself is not touched.  At 128 it is the other way round.)
template <int D, typename Tag, int VS, int SUB, typename KA, typename VA>
__device__ __forceinline__ void k5_block(f32x16 (&o)[D / 32], const s16x8 (&q)[D / 16], f32x16& S_cur, f32x16& S_nxt, float m,
                                         float& l, f32x4& lacc, const s16x8& onesv, float& mx, const KA& ka, const VA& va) {
    f32x16& SA = SUB == 0 ? S_cur : S_nxt;
    f32x16& SB = SUB == 0 ? S_nxt : S_cur;
    constexpr bool BF = std::is_same<Tag, bf16_tag>::value;
    if constexpr (D == 128) {
        if constexpr (BF) RSA_K5_PICK(RSA_K5_BLOCK, 128, BF16, RSA_K5_OPS_128, RSA_K5_CLOBBER_128);
        else RSA_K5_PICK(RSA_K5_BLOCK, 128, F16, RSA_K5_OPS_128, RSA_K5_CLOBBER_128);
    } else {
        if constexpr (BF) RSA_K5_PICK(RSA_K5_BLOCK, 64, BF16, RSA_K5_OPS_64, RSA_K5_CLOBBER_64);
        else RSA_K5_PICK(RSA_K5_BLOCK, 64, F16, RSA_K5_OPS_64, RSA_K5_CLOBBER_64);
    }
}

// WIDE: 16-byte output stores after a permlane32_swap regroup (needs 16-byte aligned output rows), else 8-byte stores.
// FORM: 2 = hand-placed block, score chain started from -m (the product at head dim 128); 1 = hand-placed block with the
// compiled block's arithmetic (the product at head dim 64); the A/B and diagnostics builds (make ab / make diag,
// -DRSA_K5_FORMS) carry both at either head dim plus 0 = the block as hipcc schedules it (round 2's kernel).  0 and 1 are
// bit-identical to each other, 2 differs from them by the rounding order of S - m.
constexpr int k5_product_form(int D) { return D == 128 ? 2 : 1; }
template <int D, typename Tag, bool WIDE, int FORM>
__global__ __launch_bounds__(256, 2) void bsfwd_kernel(AttnArgs a) {
    constexpr int NW = 4;                   // 4 waves x 32 query rows
    constexpr int KS = D / 16;
    constexpr int DT = D / 32;
    constexpr int CHR = D / 8;
    constexpr int RPI = 1024 / (D * 2);     // rows per 1-KiB piece
    constexpr int TILE_BYTES = 64 * D * 2;
    constexpr int NPC = TILE_BYTES / 1024 / NW;  // 1-KiB pieces per wave per tile operand (4 at head dim 128, 2 at 64)
    using E = Elem<Tag>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned short* lds_list = reinterpret_cast<unsigned short*>(lds + 4 * TILE_BYTES);

    const int work = blockIdx.x;
    const GsyncTicket gs_tk = rsa_gsync_announce(a.gsync, a.gsync_gen);   // aligned starts (rsa_attn.h)
    // ---------------- work mapping: dense text-row blocks first, then the sparse blocks chunked per XCD ----------------
    int bh, qblk, tsp = 0;   // tsp: which part of a text block's key range this workgroup walks
    // (the split text-row pieces are the LAST workgroups of the grid -- a.heavy_last, as in rsa_attn_kernel64.hip)
    const int n_sparse = a.BH * a.NBp;
    const bool text = a.heavy_last ? work >= n_sparse : work < a.n_heavy_pad;
    if (text) {
        const int wh = a.heavy_last ? work - n_sparse : work;
        const int ntq = a.NQB - a.NBv;
        const int per_bh = ntq * a.tsplit;      // text blocks x key-range splits (tsplit = 1: no split)
        if (ntq <= 0 || wh >= a.BH * per_bh) return;
        bh = wh / per_bh;
        const int rem = wh % per_bh;
        qblk = a.NBv + rem / a.tsplit;
        tsp = rem % a.tsplit;
    } else {
        const int v = a.heavy_last ? work : work - a.n_heavy_pad;
        bh = v / a.NBp;
        const int j = v % a.NBp;
        const int chunk = a.NBp >> 3;
        qblk = (j & 7) * chunk + (j >> 3);
        if (qblk >= a.NBv) return;
    }
    const int b = bh / a.H, h = bh % a.H;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int grow = qblk * 128 + 32 * wv + r;

    // ---------------- per-row plan ----------------
    int lo_r = 0, hi_r = 0;
    bool store_r = false, zero_r = false;
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
            hi_r = a.kv_valid; store_r = grow < a.Sq;
        } else {
            n_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
            if (a.tsplit > 1) {   // split-KV: this workgroup's slice of the key blocks
                first_blk = tsp * a.tper;
                n_items = n_items - first_blk < a.tper ? n_items - first_blk : a.tper;
                if (n_items < 0) n_items = 0;
            }
            lo_max = 0; hi_min = hi_max = a.kv_text_valid;
            hi_r = a.kv_text_valid;
            store_r = grow < a.q_text_end;
            zero_r = !store_r && grow < a.Sq;
        }
    } else {
        // dense mode: one or two (query rows, key rows) segments (attn.py:107-120); causal = bottom-right aligned inside a
        // segment (flash-attn's convention; equal to the top-left form of the reference's "torch" / "vanilla" modes,
        // attn.py:101-106 / :129-133, whenever a segment has as many keys as rows -- the only case the Python side lets
        // through.  The reference's "flash" mode never passes `causal` on, attn.py:107-116, and neither does attn.py here)
        const int row0 = qblk * 128, row1 = row0 + 128;
        auto seg_hi = [&](int row) -> int {   // one past the last key row `row` may see
            const bool s1 = row >= a.q_split;
            const int lo = s1 ? a.kv_split : 0, hi = s1 ? a.Sk : a.kv_split;
            if (!a.causal) return hi;
            const int rows = s1 ? a.Sq - a.q_split : a.q_split, rin = row - (s1 ? a.q_split : 0);
            const int lim = lo + rin + 1 + ((hi - lo) - rows);
            return lim < lo ? lo : (lim < hi ? lim : hi);
        };
        lo_r = grow < a.q_split ? 0 : a.kv_split;
        hi_r = seg_hi(grow < a.Sq ? grow : a.Sq - 1);
        store_r = grow < a.Sq;
        int lo_min;
        const int rlast = (row1 <= a.Sq ? row1 : a.Sq) - 1;   // last real row of the block
        if (row1 <= a.q_split) { lo_min = 0; lo_max = 0; }
        else if (row0 >= a.q_split) { lo_min = lo_max = a.kv_split; }
        else { lo_min = 0; lo_max = a.kv_split; }
        // seg_hi grows with the row inside a segment: extremes of the block sit at its first / last row of each segment
        hi_min = seg_hi(row0);
        hi_max = seg_hi(rlast);
        if (row0 < a.q_split && rlast >= a.q_split) {   // the block straddles the two segments
            const int h0 = seg_hi(a.q_split - 1), h1 = seg_hi(a.q_split);
            hi_min = hi_min < h1 ? hi_min : h1;
            hi_max = hi_max > h0 ? hi_max : h0;
        }
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
    auto blk_of = [&](int item) -> int { return use_list ? (int)lds_list[item] : first_blk + item; };
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

    // ---------------- Q fragments (B operand) ----------------
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
    // A region of keys is staged in groups of 4 pieces (4 KiB = 4*RPI rows); in every group wave w moves piece w: rows
    // w*RPI .. +RPI-1 of the group.  The source-chunk swizzle depends on the row inside the group only.
    const int rsub = lane / CHR, cl = lane % CHR;
    const int rowl = wv * RPI + rsub;  // row inside a group
    const int gsw = D == 128 ? (cl ^ (((rowl & 3) << 2) | ((rowl >> 2) & 3))) : (cl ^ ((rowl >> 1) & 7));
    const unsigned voffk = (unsigned)(((long)rowl * a.kss + gsw * 8) * 2);
    const unsigned voffv = (unsigned)(((long)rowl * a.vss + gsw * 8) * 2);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const long kstep = (long)(4 * RPI) * a.kss * 2, vstep = (long)(4 * RPI) * a.vss * 2;  // bytes per group
    // the 64-key tile starting at key `key_first` -> LDS byte offset `lds_off` (tile slots: K0 K1 V0 V1)
    auto dma = [&](int is_v, int key_first, unsigned lds_off) {
        const unsigned ld0 = lds_base + lds_off + wv * 1024;
        const unsigned char* base = is_v ? vbase : kbase;
        const long ss = is_v ? a.vss : a.kss;
        if (key_first + 64 <= kv_limit) {
            const unsigned char* tb = base + (long)key_first * ss * 2;
            const long step = is_v ? vstep : kstep;
            const unsigned vo = is_v ? voffv : voffk;
#pragma unroll
            for (int j = 0; j < NPC; ++j)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(vo), "s"(tb + j * step), "s"(ld0 + j * 4096) : "memory");
        } else {   // the tile runs past the last valid key: rows clamped (their scores are masked)
#pragma unroll
            for (int j = 0; j < NPC; ++j) {
                int krow = key_first + j * 4 * RPI + rowl;
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
    // Row sums on the matrix pipe (RSM: head dim 64, gen_k5_block.py::rsm_form): lacc = the lane's complete
    // row sum of the rounded P in all four registers.  A operand of v_mfma_f32_16x16x32 (lane: row a = l & 15, k-block l >> 4): ones
    // iff (k-block & 1) == ((a >> 2) & 1) -- the B operand is the P . V product's P fragment, whose k-blocks 0 / 2 are the two lane
    // halves of query row n and 1 / 3 those of row n + 16, and the lane that owns C rows 4 (l >> 4) .. + 3 of column l & 15 is the
    // lane of exactly that query row (the construction of the e4m3 kernel's row-sum product).
#ifdef RSA_K5X_NORSM   // (A/B twin of tools/history/r5_d64x_build.sh: row sums by vector additions, as rounds 3-4)
    constexpr bool RSM = false;
#else
    constexpr bool RSM = D == 64;
#endif
    f32x4 lacc = {0.0f, 0.0f, 0.0f, 0.0f};
    s16x8 onesv;
    {
        const short one = std::is_same<Tag, bf16_tag>::value ? (short)0x3F80 : (short)0x3C00;
        const short w = (((lane >> 4) & 1) == ((lane >> 2) & 1)) ? one : (short)0;
#pragma unroll
        for (int i = 0; i < 8; ++i) onesv[i] = w;
    }
    // m_ref = the finite reference the scores are taken against (S_cur, S_nxt hold S - m_ref), nm = its negation in 16
    // registers (C operand of the first QK^T MFMA), thr = how far a new row maximum may exceed it before the rescale (-inf
    // until the row has seen a finite score: the first finite maximum always becomes the reference).  m_run is the running
    // maximum itself (-inf = nothing seen): what the split-KV partials carry, and the A/B forms' reference.
    float m_ref = 0.0f, thr = -INFINITY;
    f32x16 nm;
#pragma unroll
    for (int i = 0; i < 16; ++i) nm[i] = 0.0f;

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
    // read addresses of the hand-placed block: LDS byte addresses of sub-tile 0 / slot 0; slot, sub-tile and k-step are immediates
    using KAv = typename std::conditional<D == 128, i32x8, i32x4>::type;
    using VAv = typename std::conditional<D == 128, i32x8, i32x4>::type;
    KAv ka;
    VAv va;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = (int)lds_base + k_off(ks, 0);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { va[2 * dt] = (int)lds_base + vrd[dt][0]; va[2 * dt + 1] = (int)lds_base + vrd[dt][1]; }

    // S^T (32 keys x 32 rows) = K[sub-tile SUB of K slot] . Q^T
    auto qk_sub = [&](auto KSLOT, auto SUB, f32x16& S) {
        constexpr int slot = decltype(KSLOT)::value, sub = decltype(SUB)::value;
        const unsigned char* kt_ = lds + slot * TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 16; ++i) S[i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const s16x8 a0 = *reinterpret_cast<const s16x8*>(kt_ + k_off(ks, sub));
            S = E::mfma(a0, qf[ks], S);
        }
    };
    auto rowmax_sub = [&](const f32x16& S, float& mx) {
        float m = S[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, S[i]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    auto apply_mask_sub = [&](f32x16& S, int key_first) {
        int kbase = key_first + 4 * hh;
        asm volatile("" : "+v"(kbase));   // rare branch: keep its 16 key indices out of the loop's live registers
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kbase + (i & 3) + 8 * (i >> 2);
            if (kk < lo_r || kk >= hi_r) S[i] = -INFINITY;
        }
    };

    int kq1 = 0, kq2 = 0;  // first keys of tile+1 / tile+2 (fetched from LDS ahead of use)
#ifdef RSA_K5_DIAG
    // diagnostics build: s_memtime at four points of every sub-step, differences summed per wave (scalar registers)
    unsigned long long tstamp[4] = {0, 0, 0, 0}, tsum[4] = {0, 0, 0, 0}, tkern0;
    auto stamp_now = [&]() -> unsigned long long {
        unsigned long long tt;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return tt;
    };
    tkern0 = stamp_now();
#define RSA_STAMP(i) do { tstamp[i] = stamp_now(); if ((i) > 0) tsum[i] += tstamp[i] - tstamp[(i) > 0 ? (i) - 1 : 0]; } while (0)
#else
#define RSA_STAMP(i) do { } while (0)
#endif

    // The part of a sub-step u = 2*tile + SUB behind its staging: rare branches (boundary mask, deferred rescale), then the
    // pipelined block: consumes S_cur (scores of 32 keys, row max in mx_cur), produces S_nxt / mx_nxt for sub-step u+1.
    // VS = slot parity of `tile` (its K and V slots); next scores = K(tile) sub-tile 1 (SUB 0) / K(tile+1) sub-tile 0 (SUB 1).
    auto half = [&](auto VS, auto SUB, int key0, f32x16& S_cur, float& mx_cur, f32x16& S_nxt, float& mx_nxt) {
        constexpr int vs = decltype(VS)::value, sub = decltype(SUB)::value;
        const int kfirst = key0 + 32 * sub;
        if (kfirst < lo_max || kfirst + 32 > hi_min) {
            apply_mask_sub(S_cur, kfirst);
            rowmax_sub(S_cur, mx_cur);
        }
        if constexpr (FORM == 2) {
            // S_cur, mx_cur are relative to m_ref as it was when they were computed, and that is still m_ref
            if (__builtin_amdgcn_ballot_w64(mx_cur > thr) != 0ull) {
                asm volatile("s_nop 11" ::: "memory");   // the block's last MFMA wrote O: 12 wait states before a VALU touches it
                const bool first = thr == -INFINITY;
                float delta = first ? mx_cur : fmaxf(mx_cur, 0.0f);
                if (delta == -INFINITY) delta = 0.0f;      // nothing but masked keys so far
                else thr = 8.0f;
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // (first: O and l are still zero)
                m_ref += delta;
                l_run *= alpha;
                if constexpr (RSM) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) lacc[i] *= alpha;
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { S_cur[i] -= delta; nm[i] = -m_ref; }
            }
            RSA_STAMP(2);
            k5_block_nm<D, Tag, vs, sub>(o, qf, S_cur, S_nxt, nm, l_run, lacc, onesv, mx_nxt, ka, va);
            RSA_STAMP(3);
        }
        else {   // the classic arithmetic (scores S, reference m_run subtracted in the softmax)
            if (__builtin_amdgcn_ballot_w64(mx_cur > m_run + 8.0f) != 0ull) {
                asm volatile("s_nop 11" ::: "memory");
                const float m_new = fmaxf(m_run, mx_cur);
                const float mu = (m_new == -INFINITY) ? 0.0f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - mu);
                m_run = m_new;
                l_run *= alpha;
                if constexpr (RSM) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) lacc[i] *= alpha;
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            }
            const float m_use = (m_run == -INFINITY) ? 0.0f : m_run;
            RSA_STAMP(2);
            if constexpr (FORM == 1) {
                k5_block<D, Tag, vs, sub>(o, qf, S_cur, S_nxt, m_use, l_run, lacc, onesv, mx_nxt, ka, va);
            }
#ifdef RSA_K5_FORMS
            else {   // A/B build: the block left to hipcc; iglp_opt(0) = LLVM's small-GEMM MFMA/DS interleave
                __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_iglp_opt(0);
                if constexpr (sub == 0) qk_sub(std::integral_constant<int, vs>{}, std::integral_constant<int, 1>{}, S_nxt);
                else qk_sub(std::integral_constant<int, vs ^ 1>{}, std::integral_constant<int, 0>{}, S_nxt);
                s16x8 pb[2];
                float ps = 0.0f;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float pv8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        pv8[i] = __builtin_amdgcn_exp2f(S_cur[8 * hf + i] - m_use);
                        ps += pv8[i];
                    }
                    pb[hf] = E::cvt8(pv8);
                }
                if constexpr (RSM) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) lacc = E::mfma_rowsum(onesv, pb[hf], lacc);
                } else {
                    l_run += ps;
                }
                const unsigned char* vt_ = lds + (2 + vs) * TILE_BYTES;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int offa = vrd[dt][0] + (2 * sub + k2) * 16 * D * 2;
                        const int offb = vrd[dt][1] + (2 * sub + k2) * 16 * D * 2;
                        const s16x4 va_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(vt_ + offa));
                        const s16x4 vb_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(vt_ + offb));
                        const s16x8 av = __builtin_shufflevector(va_, vb_, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[dt] = E::mfma(av, pb[k2], o[dt]);
                    }
                }
                rowmax_sub(S_nxt, mx_nxt);
                __builtin_amdgcn_s_setprio(0);
            }
#endif
            RSA_STAMP(3);
        }
    };

    // staging: a wait + barrier + issue point in front of every sub-step.  SUB 0 issues V(tile+1) into V(tile-1)'s slot,
    // SUB 1 K(tile+2) into K(tile)'s slot; the newest group (issued half a tile ago) may stay in flight, the one issued a
    // tile ago must land.
    auto stage = [&](auto VS, auto SUB, int tile) {
        constexpr int vs = decltype(VS)::value, sub = decltype(SUB)::value;
        RSA_STAMP(0);
        if (tile + 1 < n_tiles) {
            if constexpr (NPC == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifndef RSA_K5X_NOBAR     // (RSA_K5X_*: timing experiments of tools/history/r5_d64x_build.sh, never defined in the product)
        __syncthreads();
#endif
        RSA_STAMP(1);
#ifndef RSA_K5X_NODMA
        if constexpr (sub == 0) {
            if (tile + 1 < n_tiles) dma(1, kq1, (2 + (vs ^ 1)) * TILE_BYTES);
        } else {
            if (tile + 2 < n_tiles) dma(0, kq2, vs * TILE_BYTES);
        }
#endif
    };

    // ---------------- prologue + main loop ----------------
    if (qblk < a.NBv || a.mode != MODE_SPARSE) rsa_gsync_wait(a.gsync, gs_tk, n_items, a.NB_total, a.gsync_ratio);   // aligned starts: in front of the first staging instruction (text-row pieces do not wait, as in the 64-row and e4m3 kernels)
    f32x16 SA, SB;
    float mxA = -INFINITY, mxB = -INFINITY;
    int key0 = 0;
    if (n_tiles > 0) {
        key0 = key0_of(0);
        kq1 = key0_of(1);
        kq2 = key0_of(2);
        dma(0, key0, 0);
        dma(1, key0, 2 * TILE_BYTES);
        if (n_tiles > 1) dma(0, kq1, TILE_BYTES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        qk_sub(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, SA);
        rowmax_sub(SA, mxA);
    }
    // The kept-list entry of tile+3 is read from LDS one advance() EARLY into a register (pref_raw) and only made scalar
    // here: the LDS round trip (~100 cycles, once per tile and wave) is off the wave's critical path.
    auto raw_item = [&](int tile) -> int {   // block index of `tile`'s list entry, still per lane (index clamped)
        const int it = tile >> 1;
        return blk_of(it < n_items ? it : (n_items > 0 ? n_items - 1 : 0));
    };
    int pref_raw = n_tiles > 0 ? raw_item(3) : 0;
    auto advance = [&](int tile) {  // after finishing `tile`: shift the key queue, tile+3's first key from the prefetched entry
        key0 = kq1;
        kq1 = kq2;
        kq2 = __builtin_amdgcn_readfirstlane(pref_raw) * RSA_BLOCK + ((tile + 3) & 1) * 64;
        pref_raw = raw_item(tile + 4);
    };
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        auto tile_step = [&](auto VS, int tile) {
            stage(VS, I0{}, tile);
            half(VS, I0{}, key0, SA, mxA, SB, mxB);
            stage(VS, I1{}, tile);
            half(VS, I1{}, key0, SB, mxB, SA, mxA);
        };
        int tile = 0;
        for (; tile + 1 < n_tiles; tile += 2) {
            tile_step(I0{}, tile);
            advance(tile);
            tile_step(I1{}, tile + 1);
            advance(tile + 1);
        }
        if (tile < n_tiles) tile_step(I0{}, tile);
    }

    // ---------------- epilogue ----------------
    asm volatile("s_nop 11" ::: "memory");   // (the last block's last MFMA -> the reads of O below)
    if constexpr (FORM == 2) m_run = thr == -INFINITY ? -INFINITY : m_ref;
    const auto swl = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l_tot = RSM ? lacc[0] : __uint_as_float(swl[0]) + __uint_as_float(swl[1]);   // (RSM: already complete over both lane halves)
    bool done = false;
    if (a.mode == MODE_SPARSE && a.tsplit > 1 && qblk >= a.NBv) {
        // split-KV partial of a text block: unnormalised O (fp32), m (log2 domain) and l per row; the combine
        // kernel (rsa_attn.hip) merges the tsplit parts
        const int ntq = a.NQB - a.NBv;
        const int rowb = 32 * wv + r;
        float* pp = a.tpart + ((((long)bh * ntq + (qblk - a.NBv)) * a.tsplit + tsp) * RSA_BLOCK + rowb) * (D + 2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * dt + 8 * g + 4 * hh;
                *reinterpret_cast<float2*>(pp + d0) = make_float2(o[dt][4 * g + 0], o[dt][4 * g + 1]);
                *reinterpret_cast<float2*>(pp + d0 + 2) = make_float2(o[dt][4 * g + 2], o[dt][4 * g + 3]);
            }
        if (hh == 0) *reinterpret_cast<float2*>(pp + D) = make_float2(m_run, l_tot);
        done = true;
    }
    if (!done && (store_r || zero_r)) {
        float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
        float Rv = 1.0f;
        // the compensation values this lane adds (d = 32 dt + 8 g + 4 hh + 0..3) and R: all loads issued here, back to back,
        // one wait (loaded per (dt, g) inside the store loop each load's latency is exposed in turn: round 4, profiles/r04_k5_w64.md)
        float4 cv[DT][4];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) cv[dt][g] = make_float4(0, 0, 0, 0);
        if (rectify && !zero_r) {
            const long rowi = (long)bh * a.NBv + qblk;
            const float* cp = a.comp + rowi * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) cv[dt][g] = *reinterpret_cast<const float4*>(cp + 32 * dt + 8 * g + 4 * hh);
        }
        if (rectify) Rv = a.R[(long)bh * a.NBv + qblk];
        if (zero_r) inv = 0.0f;
        const float sc = inv * Rv;
        unsigned short* op = a.out + (long)b * a.osb + (long)h * a.osh + (long)grow * a.oss;
        // O * sc + comp: one fma rounded to fp32, THEN the conversion to the storage type (what the oracle does).  Left alone,
        // hipcc folds fma + conversion into v_fma_mixlo_f16 (one rounding, straight to fp16) in one store form and not in the
        // other; the empty asm keeps the fp32 value, so both store forms write the same bytes.
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
                        const float4 c4 = cv[dt][g];
                        const float v0 = fin(o[dt][4 * g + 0], c4.x);
                        const float v1 = fin(o[dt][4 * g + 1], c4.y);
                        const float v2 = fin(o[dt][4 * g + 2], c4.z);
                        const float v3 = fin(o[dt][4 * g + 3], c4.w);
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
        } else {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * dt + 8 * g + 4 * hh;
                    const float4 c4 = cv[dt][g];
                    const float v0 = fin(o[dt][4 * g + 0], c4.x);
                    const float v1 = fin(o[dt][4 * g + 1], c4.y);
                    const float v2 = fin(o[dt][4 * g + 2], c4.z);
                    const float v3 = fin(o[dt][4 * g + 3], c4.w);
                    uint2 pk;
                    pk.x = (unsigned)E::from_f32(v0) | ((unsigned)E::from_f32(v1) << 16);
                    pk.y = (unsigned)E::from_f32(v2) | ((unsigned)E::from_f32(v3) << 16);
                    *reinterpret_cast<uint2*>(op + d0) = pk;
                }
            }
        }
    }
#ifdef RSA_K5_DIAG
    if (a.dbg && lane == 0) {
        const unsigned long long tend = stamp_now();
        unsigned long long* o8 = a.dbg + ((long)work * 4 + wv) * 8;
        o8[0] = tsum[1]; o8[1] = tsum[2]; o8[2] = tsum[3]; o8[3] = tend - tkern0; o8[4] = (unsigned long long)n_tiles;
        o8[5] = (unsigned long long)qblk; o8[6] = tkern0; o8[7] = tend;
    }
#endif
}

// launch hook used by rsa_attn.hip::launch_attn
#ifdef RSA_K5_FORMS
int g_rsa_k5_form = -1;   // A/B and diagnostics builds: tuning key "k5_form" (see the FORM template parameter); -1 = the product's choice
#endif
int rsa_launch_bsfwd(const AttnArgs& a, dim3 grid, size_t lds_bytes, int D, int dtype, hipStream_t s) {
    // the 16-byte output stores need 16-byte aligned rows; anything else takes the 8-byte form
    const bool wide = !(((uintptr_t)a.out & 15) || ((a.osb | a.osh | a.oss) & 7));
#ifdef RSA_K5_FORMS
#define RSA_K5(DD, TT) \
    do { \
        if (!wide) RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, false, k5_product_form(DD)>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
        else if (g_rsa_k5_form == 0) RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, true, 0>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
        else if (g_rsa_k5_form == 1) RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, true, 1>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
        else if (g_rsa_k5_form == 2) RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, true, 2>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
        else RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, true, k5_product_form(DD)>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
    } while (0)
#else
#define RSA_K5(DD, TT) \
    do { \
        if (wide) RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, true, k5_product_form(DD)>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
        else RSA_LAUNCH_GSYNC(2, (bsfwd_kernel<DD, TT, false, k5_product_form(DD)>), a, a.mode == MODE_SPARSE, grid, 256, lds_bytes, s); \
    } while (0)
#endif
    if (D == 128) {
        if (dtype == RSA_BF16) RSA_K5(128, bf16_tag); else RSA_K5(128, fp16_tag);
    } else {
        if (dtype == RSA_BF16) RSA_K5(64, bf16_tag); else RSA_K5(64, fp16_tag);
    }
#undef RSA_K5
    return rsa_launch_status();
}
