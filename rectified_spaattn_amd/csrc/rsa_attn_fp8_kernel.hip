// K5, fp8 operands: block-sparse flash attention forward on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3 -> f32, twice the
// bf16 MFMA rate) with the rectification epilogue fused.  Same work mapping, per-row plan, kept lists and epilogue as
// the 2-byte kernel (rsa_attn_kernel.hip); the operands are the block-scaled images of rsa_fp8_emit.h.
//
// One workgroup (4 waves, two workgroups per CU) owns one 128-row query block, wave w rows 32w..32w+31, "key on the
// register, query row on the lane":
//      S^T[key][q]  = K8 . Q8^T     A = K8 rows (2 x ds_read_b128 of a 128-byte row), B = Q8 (16 registers, loaded once)
//      O^T[d][q]   += V8^T . P8^T   A = rows of the pre-transposed V tile (2 x ds_read_b128), B = P = the two 32-key score
//                                   accumulators of a 64-key tile converted in place
// The 64 k-slots of the PV product are (lane half h, byte j); the producer stores V^T with the keys of a tile in exactly
// that order, so neither operand needs a transposing read or any cross-lane traffic.
// Scales: every 128-row block of Q, K, V carries a power-of-two scale (one E8M0 byte), which the MFMAs apply for free through
// their block-scale operands: Q block and K tile on the QK^T products (Q already holds q * sm_scale * log2(e)), V tile on
// the PV products; the kept-list entry a wave reads from LDS per tile carries the key block's two exponents beside its index.
// Scores: the QK^T chain starts from a 16-register block holding the row's (deferred) reference, so the accumulator is
// relative to it on arrival, in the unit the P form wants (PMap below): the product turns it into the e4m3 CODE of P with one
// v_cvt_pk_u8_f32 per score; the verification form (tuning key fp8_variant = 2) takes v_exp_f32 + v_cvt_pk_fp8_f32.  The
// reference moves only when a tile maximum exceeds it by the form's threshold; that rare move also shifts the scores already
// computed for the next tile.  The row sum l comes from the matrix pipe too: one v_mfma_f32_16x16x128_f8f6f4 per tile
// multiplies the SAME packed P operand by a ones/zeros pattern (read from a 64-byte LDS table) chosen so that every lane's
// accumulator receives the full 64-key sum of its own query row (tools/probes/fp8_rowsum_probe.hip).  So l sums the P that
// the PV product uses -- numerator and denominator round together (the reference kernel sums its P before the 2-byte
// rounding, rectified_hunyuan_attn.py:93-97; with 3 mantissa bits that mismatch would be a 2^-4 relative error on rows
// carried by a few keys).  1/l and R meet in the epilogue.
//
// Pipeline (per wave, 64-key tiles, S double-buffered): step t computes S(t+1) while P(t) and O += V(t) P(t) run; the body
// of a step is one hand-placed instruction block (gen_k5_block.py::gen_block8), four copies per loop trip so that the
// LDS ring slot is a compile-time constant of each.
// Staging: K and V tiles are 8 KiB each; 4-slot rings; step t stages K(t+3) and V(t+2) behind `s_waitcnt vmcnt(4)` + barrier, so
// every tile has two full steps to land.  Since round 5 (head dim 128, PIPE_OPT bit 3) the wave's 2 + 2 LDS-DMA pieces are issued
// INSIDE the hand-placed block, one per MFMA shadow, and for every tile (past the end of the walk the last tile again, into a slot
// nobody reads), not as a burst of the eight waves behind the barrier: -2..3 % (profiles/r05_pv_hand_placed.txt).
//
// The "pv" form (round 5; HYB instances, head dim 128): Q . K^T on the 2-byte q and k themselves -- K tiles of 64 keys x 256 bytes in
// the 2-byte kernels' image, sixteen v_mfma_f32_32x32x16 per tile whose chain starts from the same reference block (Q carries
// sm_scale log2e 8, so the accumulator still arrives as the e4m3 code of P) -- and everything behind the scores as above: code-map
// P, row sum and P . V on the e4m3 V image.  Three-slot rings (K 3 x 16 KiB, V 3 x 8 KiB), six tiles per loop trip, its own
// hand-placed block (gen_block8h), the kept list as a 1 024-entry LDS window.  Scores of the 2-byte path, P . V at the fp8 rate.
#include "rsa_attn.h"
#include "rsa_attn_block.h"

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef RSA_PV_LIST_WINDOW
#define RSA_PV_LIST_WINDOW 1024      // entries of the pv form's kept-list window in LDS (a power of two; tests build nothing else)
#endif

struct Attn8Args {
    const uint8_t *q8, *k8, *v8t;  // [BH, S_pad, 128], [BH, S_pad, 128], [BH, S_pad/64, 128, 64]
    const uint32_t* exps;          // [BH, exps_stride] E8M0 block exponents: byte 0 Q, byte 1 K, byte 2 V (rsa_fp8_emit.h)
    int exps_stride;
    unsigned short* out;
    long osb, osh, oss;
    const int32_t* cols;
    const int32_t* counts;
    const float* R;
    const float* comp;
    int mode, H, Sq, Sk, Sq_pad, Sk_pad;  // padded rows of q8 and of k8 / v8t
    int NBv, NQB, NB_total;
    int kv_valid, kv_text_valid, q_text_end;
    int q_split, kv_split, causal;
    int n_heavy_pad, NBp, BH;
    int out_fp16;
    float* tpart;        // split-KV partials of the text query blocks (layout of rsa_attn.hip's combine kernel) or null
    int tsplit, tper;
    unsigned* gsync;     // aligned starts (rsa_attn.h): this launch's counters or null
    int gsync_gen, gsync_ratio, k5_static;   // (k5_static: set by RSA_LAUNCH_GSYNC for every K5 args struct; only the 64-row 2-byte kernel reads it)
    int heavy_last;      // the split text-row pieces behind the sparse blocks in the grid
    int tail_first, tail_n, tail_p;   // tail split (rsa_attn.hip::launch_attn, rsa_attn_kernel64.hip::k5w_map): head dim 128 only
    float* tail_part;
    // "pv" form (round 5, HYB instances): Q . K^T on the 2-byte operands as they are, only P . V on e4m3
    const unsigned short *q16, *k16;
    long qsb, qsh, qss, ksb, ksh, kss;
    float qk_scale;                   // sm_scale * log2(e) * PMap::U, folded into Q
};

// rsa_attn.hip: merge of the split-KV partials of the text blocks, and the switch for the split
int rsa_launch_text_combine(const float* tpart, unsigned short* out, long osb, long osh, long oss, int D, int H, int NBv,
                            int ntq, int tsplit, int q_text_end, int Sq, int BH, int dtype, hipStream_t s);
int rsa_text_split_enabled();
int rsa_text_last_enabled();
int rsa_text_split_capacity(size_t tpart_bytes, int BH, int ntq, int D);
bool rsa_tail_fits(size_t tpart_bytes, int BH, int ntq, int D, int tail_n, int tail_p);
int rsa_plan_tail_split(long n_sparse, long n_heavy_pad, int* tail_first, int* tail_n, int* tail_p);
int rsa_launch_tail_combine(const float* part, unsigned short* out, long osb, long osh, long oss, int H, int NBv, int NBp,
                            int tail_first, int tail_n, int tail_p, const float* R, const float* comp, int Sq, int dtype,
                            hipStream_t s);

namespace {

constexpr int NSLOT = 4;   // K / V tiles are 64 keys x D8 bytes resp. D8 rows x 64 bytes: 64 D8 bytes each (8 KiB at head dim 128)
// How P reaches e4m3.  Both forms keep the QK^T accumulator relative to the row's reference m (the chain starts from a block
// holding the constant), so a score s arrives as U (s - m + OFFSET) + BIAS:
//   exact form (PIPE_OPT bit 2 clear): U = 1, BIAS = 0: log2(P); P = v_exp_f32, then v_cvt_pk_fp8_f32 (round to nearest even);
//   code-map form (bit 2 set, product):  U = 8, BIAS = 56: the e4m3 CODE of P on the piecewise-linear log2 of the format
//     (code c = 8 (e + 7) + m3 means 2^e (1 + m3 / 8), so log2(value) ~ c / 8 - 7 with an error below 0.086 inside a binade);
//     one v_cvt_pk_u8_f32 per score rounds it (nearest even, negative and NaN -> 0: tools/probes/cvt_pk_u8_probe.hip) and
//     places the byte.  P is then a function of the score that differs from 2^x by a smooth factor in [1, 1.0615] times the
//     format's own rounding; numerator (PV) and denominator (row sum, same codes, on the matrix pipe) see the same P.
//     Measured against exact P: ~1.2x the RMS error of round-to-nearest e4m3 P (tests/diag_fp8_pmap.py), i.e. +0.3 % of the
//     fp8 path's total error against the bf16 oracle, for 32 v_exp_f32 + 16 conversions -> 32 conversions per tile.
// OFFSET / THRESH: the reference m moves only when a tile's maximum exceeds it by THRESH (deferred rescale), so
// log2(P) < OFFSET + THRESH: 8 < log2(448) for the exact form; 8.5 for the code map = code 124 <= 126 (127 is NaN).
template <bool CODEMAP> struct PMap;
template <> struct PMap<false> { static constexpr float U = 1.0f, BIAS = 0.0f, OFFSET = 4.0f, THRESH = 4.0f; static constexpr int EXP = 0; };
template <> struct PMap<true> { static constexpr float U = 8.0f, BIAS = 56.0f, OFFSET = 6.5f, THRESH = 2.0f; static constexpr int EXP = 3; };

__device__ __forceinline__ f32x16 mfma8(i32x8 a, i32x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);  // e4m3 x e4m3, unscaled
}
// same with E8M0 block scales 2^(sa-127) on A and 2^(sb-127) on B: byte 0 of each lane's scale register ...
__device__ __forceinline__ f32x16 mfma8s(i32x8 a, i32x8 b, f32x16 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}
// ... and byte 1 of the same registers (the PV product: V block scale on A, 1.0 on the P operand)
__device__ __forceinline__ f32x16 mfma8s1(i32x8 a, i32x8 b, f32x16 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa, 1, sb);
}

// the hand-placed tile block (gen_k5_block.py, RSA_K5F8_*): one asm statement per 64-key tile, registers pinned; TS = tile & 3
template <int TS, bool CODEMAP, int D8, typename KA>
__device__ __forceinline__ void k5f8_block(f32x16 (&o)[D8 / 32], const i32x8 (&q)[D8 / 64], f32x16 (&SA)[2], f32x16 (&SB)[2],
                                           const f32x16& mblk, f32x4& lacc, float& mx, int sca, int scb, const KA& ka,
                                           const i32x2& va, int ona) {
    if constexpr (D8 == 64) {
        static_assert(CODEMAP, "head dim 64 has the code-map block only");
        if constexpr (TS == 0) asm volatile(RSA_K5F8_BLOCKC64_T0 RSA_K5F8_OPS64 : RSA_K5F8_CLOBBER64, "memory");
        else if constexpr (TS == 1) asm volatile(RSA_K5F8_BLOCKC64_T1 RSA_K5F8_OPS64 : RSA_K5F8_CLOBBER64, "memory");
        else if constexpr (TS == 2) asm volatile(RSA_K5F8_BLOCKC64_T2 RSA_K5F8_OPS64 : RSA_K5F8_CLOBBER64, "memory");
        else asm volatile(RSA_K5F8_BLOCKC64_T3 RSA_K5F8_OPS64 : RSA_K5F8_CLOBBER64, "memory");
        return;
    }
    if constexpr (CODEMAP) {
        if constexpr (TS == 0) asm volatile(RSA_K5F8_BLOCKC_T0 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
        else if constexpr (TS == 1) asm volatile(RSA_K5F8_BLOCKC_T1 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
        else if constexpr (TS == 2) asm volatile(RSA_K5F8_BLOCKC_T2 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
        else asm volatile(RSA_K5F8_BLOCKC_T3 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
    } else {
        if constexpr (TS == 0) asm volatile(RSA_K5F8_BLOCK_T0 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
        else if constexpr (TS == 1) asm volatile(RSA_K5F8_BLOCK_T1 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
        else if constexpr (TS == 2) asm volatile(RSA_K5F8_BLOCK_T2 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
        else asm volatile(RSA_K5F8_BLOCK_T3 RSA_K5F8_OPS : RSA_K5F8_CLOBBER, "memory");
    }
}

// the product's block at head dim 128 (round 5): code map + the wave's four LDS-DMA pieces of K(tile + 3) / V(tile + 2) inside
template <int TS, typename KA>
__device__ __forceinline__ void k5f8_block_dma(f32x16 (&o)[4], const i32x8 (&q)[2], f32x16 (&SA)[2], f32x16 (&SB)[2], const f32x16& mblk,
                                               f32x4& lacc, float& mx, int sca, int scb, const KA& ka, const i32x2& va, const i32x8& onesv,
                                               const i32x2& dk, const i32x2& dv, const unsigned char* ksrc, const unsigned char* vsrc,
                                               unsigned ldsw) {
    if constexpr (TS == 0) asm volatile(RSA_K5F8_BLOCKCD_T0 RSA_K5F8_OPSD : RSA_K5F8_CLOBBER, "scc", "memory");
    else if constexpr (TS == 1) asm volatile(RSA_K5F8_BLOCKCD_T1 RSA_K5F8_OPSD : RSA_K5F8_CLOBBER, "scc", "memory");
    else if constexpr (TS == 2) asm volatile(RSA_K5F8_BLOCKCD_T2 RSA_K5F8_OPSD : RSA_K5F8_CLOBBER, "scc", "memory");
    else asm volatile(RSA_K5F8_BLOCKCD_T3 RSA_K5F8_OPSD : RSA_K5F8_CLOBBER, "scc", "memory");
}

// ... and at head dim 64: one K and one V piece per wave and tile, the row-sum product's ones operand in registers
template <int TS>
__device__ __forceinline__ void k5f8_block_dma64(f32x16 (&o)[2], const i32x8 (&q)[1], f32x16 (&SA)[2], f32x16 (&SB)[2], const f32x16& mblk,
                                                 f32x4& lacc, float& mx, int sca, int scb, const i32x2& ka, const i32x2& va, const i32x8& onesv,
                                                 int dk, int dv, const unsigned char* ksrc, const unsigned char* vsrc, unsigned ldsw) {
    if constexpr (TS == 0) asm volatile(RSA_K5F8_BLOCKCD64_T0 RSA_K5F8_OPS64D : RSA_K5F8_CLOBBER64, "scc", "memory");
    else if constexpr (TS == 1) asm volatile(RSA_K5F8_BLOCKCD64_T1 RSA_K5F8_OPS64D : RSA_K5F8_CLOBBER64, "scc", "memory");
    else if constexpr (TS == 2) asm volatile(RSA_K5F8_BLOCKCD64_T2 RSA_K5F8_OPS64D : RSA_K5F8_CLOBBER64, "scc", "memory");
    else asm volatile(RSA_K5F8_BLOCKCD64_T3 RSA_K5F8_OPS64D : RSA_K5F8_CLOBBER64, "scc", "memory");
}

// the pv form's hand-placed block (gen_k5_block.py::gen_block8h, RSA_K5F8H_*): T6 = tile % 6 (three-slot rings, S_cur = SA on even tiles)
// (qv: the 2-byte Q fragments of four k-steps per 16-register value: two values at head dim 128, one at 64)
template <int T6, int HYB, int D8>
__device__ __forceinline__ void k5f8h_block(f32x16 (&o)[D8 / 32], const f32x16 (&qv)[D8 / 64], f32x16 (&SA)[2], f32x16 (&SB)[2], const f32x16& mblk,
                                            f32x4& lacc, float& mx, int sca, int scb, int kah, const i32x2& vah, const i32x8& onesv) {
#define RSA_K5F8H_CASE(T_) \
    if constexpr (T6 == T_ && D8 == 128) { \
        if constexpr (HYB == 2) asm volatile(RSA_K5F8H_BLOCK_F16_T##T_ RSA_K5F8H_OPS : RSA_K5F8H_CLOBBER, "memory"); \
        else asm volatile(RSA_K5F8H_BLOCK_BF16_T##T_ RSA_K5F8H_OPS : RSA_K5F8H_CLOBBER, "memory"); \
    } else if constexpr (T6 == T_) { \
        if constexpr (HYB == 2) asm volatile(RSA_K5F8H64_BLOCK_F16_T##T_ RSA_K5F8H64_OPS : RSA_K5F8H64_CLOBBER, "memory"); \
        else asm volatile(RSA_K5F8H64_BLOCK_BF16_T##T_ RSA_K5F8H64_OPS : RSA_K5F8H64_CLOBBER, "memory"); \
    }
    RSA_K5F8H_CASE(0) RSA_K5F8H_CASE(1) RSA_K5F8H_CASE(2) RSA_K5F8H_CASE(3) RSA_K5F8H_CASE(4) RSA_K5F8H_CASE(5)
#undef RSA_K5F8H_CASE
}
// ... with the wave's six LDS-DMA pieces of K(tile + 3) / V(tile + 2) inside (the product; s_add_u32 m0: SCC is clobbered)
// (dk: lane offsets of the wave's K pieces -- four at head dim 128, two at 64; dv: of its V pieces -- two / one)
template <int T6, int HYB, int D8, typename DK, typename DV>
__device__ __forceinline__ void k5f8h_block_dma(f32x16 (&o)[D8 / 32], const f32x16 (&qv)[D8 / 64], f32x16 (&SA)[2], f32x16 (&SB)[2],
                                                const f32x16& mblk, f32x4& lacc, float& mx, int sca, int scb, int kah, const i32x2& vah,
                                                const i32x8& onesv, const DK& dk, const DV& dv, const unsigned char* kb16,
                                                const unsigned char* vsrc, unsigned ldsw) {
#define RSA_K5F8H_CASE(T_) \
    if constexpr (T6 == T_ && D8 == 128) { \
        if constexpr (HYB == 2) asm volatile(RSA_K5F8H_BLOCKD_F16_T##T_ RSA_K5F8H_OPSD : RSA_K5F8H_CLOBBER, "scc", "memory"); \
        else asm volatile(RSA_K5F8H_BLOCKD_BF16_T##T_ RSA_K5F8H_OPSD : RSA_K5F8H_CLOBBER, "scc", "memory"); \
    } else if constexpr (T6 == T_) { \
        if constexpr (HYB == 2) asm volatile(RSA_K5F8H64_BLOCKD_F16_T##T_ RSA_K5F8H64_OPSD : RSA_K5F8H64_CLOBBER, "scc", "memory"); \
        else asm volatile(RSA_K5F8H64_BLOCKD_BF16_T##T_ RSA_K5F8H64_OPSD : RSA_K5F8H64_CLOBBER, "scc", "memory"); \
    }
    RSA_K5F8H_CASE(0) RSA_K5F8H_CASE(1) RSA_K5F8H_CASE(2) RSA_K5F8H_CASE(3) RSA_K5F8H_CASE(4) RSA_K5F8H_CASE(5)
#undef RSA_K5F8H_CASE
}

// PIPE_OPT bit 0: the hand-placed block (clear: the block as hipcc schedules it, same arithmetic, for A/B);
// bit 1: s_setprio around the compiled block; bit 2: the code-map form of P (PMap above); bit 3 (round 5, head dim 128, needs
// bits 0 and 2): the block issues the wave's LDS-DMA pieces itself, one per MFMA shadow, and issues them for every tile (past
// the end of the walk the last tile again, into a slot nobody reads).  Product = 15.
// D8: head dim = bytes per Q / K row (128; 64 = the CogVideoX shape, hand-placed code-map form only).
// HYB (round 5, "pv" form): 0 = e4m3 everywhere; 1 / 2 = Q . K^T on the bf16 / fp16 inputs themselves (K tiles of 64 keys x 256
// bytes staged like the 2-byte kernels' -- rsa_attn_kernel.hip -- and multiplied by v_mfma_f32_32x32x16), e4m3 only for P and V.
// The scores are then the 2-byte path's (the e4m3 rounding of Q and K is 90 % of the fp8 path's error: DESIGN 4b), P . V runs at
// the fp8 rate: 800 instead of 1 024 (2-byte) or 544 (e4m3) matrix cycles per 64 keys and 32 rows.  Three-slot rings (K 3 x 16
// KiB, V 3 x 8 KiB: two workgroups per CU); PIPE_OPT 7 = the hand-placed block (six tiles per loop trip), 6 = its compiled twin.
template <int PIPE_OPT, int D8 = 128, int HYB = 0>
__global__ __launch_bounds__(256, 2) void bsfwd_fp8_kernel(Attn8Args a) {
    static_assert(HYB == 0 || (PIPE_OPT & 4) != 0, "the pv form: code-map P");
    constexpr int TILE8 = 64 * D8;        // bytes of one K tile (64 keys x D8) and of one V tile (D8 rows x 64 keys)
    constexpr int TILEK = HYB ? 64 * 2 * D8 : TILE8;     // K tile: e4m3 rows, or the 2-byte rows themselves
    constexpr int NSK = HYB ? 3 : NSLOT, NSV = HYB ? 3 : NSLOT;
    constexpr int VBASE = NSK * TILEK, ONES = VBASE + NSV * TILE8;
    constexpr int KS8 = D8 / 64;          // QK^T MFMAs per 32-key half (k = 64 each)
    constexpr int DT8 = D8 / 32;          // 32-row d tiles of O^T
    constexpr int NPC8 = TILE8 / 4096;    // 1-KiB LDS-DMA pieces per wave and tile operand
    constexpr bool CODEMAP = (PIPE_OPT & 4) != 0;
    constexpr bool DMAB = (PIPE_OPT & 8) != 0;
    static_assert(!DMAB || (CODEMAP && (PIPE_OPT & 1) != 0), "LDS-DMA inside the block: hand-placed code-map forms");
    using PM = PMap<CODEMAP>;
    constexpr float P_BASE = PM::U * PM::OFFSET + PM::BIAS;   // accumulator value of a score equal to the reference m
    constexpr float P_GROW = PM::U * PM::THRESH + P_BASE;     // above it the reference moves
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* lds_ones = lds + ONES;  // 32 bytes of e4m3 1.0, then 32 bytes of 0
    unsigned* lds_list = reinterpret_cast<unsigned*>(lds + ONES + 64);

    const GsyncTicket gs_tk = rsa_gsync_announce(a.gsync, a.gsync_gen);   // aligned starts (rsa_attn.h)
    // ---------------- work mapping (as rsa_attn_kernel.hip) ----------------
    int bh, qblk, tsp = 0, tail = -1;
    {
        // (the split text-row pieces are the LAST workgroups of the grid -- heavy_last, as in rsa_attn_kernel64.hip: they fill the
        // slots the last generation of sparse blocks leaves idle; an unsplit text row, one long walk, still comes first)
        const int n_sparse = a.BH * a.NBp;
        bool text = a.heavy_last ? (int)blockIdx.x >= n_sparse : (int)blockIdx.x < a.n_heavy_pad;
        int bid = a.heavy_last ? (int)blockIdx.x - n_sparse : (int)blockIdx.x;        // index among the text pieces
        int vtail = -1;
        if (a.tail_n > 0) {    // (sparse blocks first: the tail's pieces sit between the whole walks and the text pieces)
            const int tail_end = a.tail_first + a.tail_n * a.tail_p;
            text = (int)blockIdx.x >= tail_end;
            bid = (int)blockIdx.x - tail_end;
            if ((int)blockIdx.x >= a.tail_first && !text) {
                tail = (int)blockIdx.x - a.tail_first;
                vtail = a.tail_first + tail / a.tail_p;
                tsp = tail % a.tail_p;
            }
        }
        if (text) {
            const int ntq = a.NQB - a.NBv;
            const int per_bh = ntq * a.tsplit;
            if (ntq <= 0 || bid >= a.BH * per_bh) return;
            bh = bid / per_bh;
            const int rem = bid % per_bh;
            qblk = a.NBv + rem / a.tsplit;
            tsp = rem % a.tsplit;
        } else {
            const int v = vtail >= 0 ? vtail : (a.heavy_last ? (int)blockIdx.x : (int)blockIdx.x - a.n_heavy_pad);
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
    const int grow = qblk * RSA_BLOCK + 32 * wv + r;

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
            if (tail >= 0) {   // this workgroup's part of the kept list (tail split)
                const int per = (n_items + a.tail_p - 1) / a.tail_p, first = tsp * per;
                const int left = n_items - first;
                list += first;
                n_items = left < 0 ? 0 : (left < per ? left : per);
            }
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
        // dense mode: one or two (query rows, key rows) segments; causal = bottom-right aligned inside a segment (as
        // rsa_attn_kernel.hip: the flash-attn convention of the reference's "flash" mode, attn.py:108-116)
        const int row0 = qblk * RSA_BLOCK, row1 = row0 + RSA_BLOCK;
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
    // the walk as LDS entries: key block | its K exponent << 16 | its V exponent << 24 (every mode: a kept list, a split
    // of the text range or a dense range), so one LDS read per tile brings the block index AND its two scales
    const uint32_t* ex = a.exps + (long)bh * a.exps_stride;
    if (t < 16) reinterpret_cast<unsigned*>(lds_ones)[t] = t < 8 ? 0x38383838u : 0u;
    // pv form: the LDS holds a WINDOW of LWIN entries (a ring: entry i at i & (LWIN - 1)), refilled half a window at a time by the
    // walk itself (step(), below) -- with the whole list of a long head the rings' 72 KiB would leave one workgroup per CU beyond
    // ~2 000 key blocks.  The other forms keep the whole list (their rings are 64 KiB).
    constexpr int LWIN = HYB ? RSA_PV_LIST_WINDOW : 0;
    auto make_entry = [&](int i) -> unsigned {
        const int blk = list != nullptr ? list[i] : first_blk + i;
        return (unsigned)blk | ((ex[blk] >> 8) << 16);
    };
    for (int i = t; i < (LWIN && n_items > LWIN ? LWIN : n_items); i += 256) lds_list[i] = make_entry(i);
    __syncthreads();
    auto entry_of = [&](int item) -> unsigned { return lds_list[LWIN ? item & (LWIN - 1) : item]; };
    int n_tiles = 2 * n_items;
    if (n_items > 0) {
        const int last_blk = list != nullptr ? list[n_items - 1] : first_blk + n_items - 1;   // (from memory: it may lie past the window)
        if (last_blk * RSA_BLOCK + 64 >= hi_max) n_tiles -= 1;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    auto raw_item = [&](int tile) -> unsigned {   // entry of `tile`'s key block, still per lane (index clamped)
        const int it = tile >> 1;
        return entry_of(it < n_items ? it : (n_items > 0 ? n_items - 1 : 0));
    };
    auto entry_sc = [&](int tile) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)raw_item(tile)); };
    auto key0_from = [&](unsigned entry, int tile) -> int { return (int)(entry & 0xFFFFu) * RSA_BLOCK + (tile & 1) * 64; };

    // ---------------- scales, Q fragments ----------------
    // Block scales are powers of two (rsa_fp8_emit.h) and ride on the MFMAs' E8M0 operands: scale register B holds the Q
    // block's exponent (+ the code map's unit of 1/8) in byte 0 and 1.0 in byte 1; scale register A is rebuilt per tile:
    // byte 0 = K exponent of the tile whose scores the block computes, byte 1 = V exponent of the tile it multiplies P with.
    const int sc_b = (int)(((ex[qblk < a.exps_stride ? qblk : 0] & 0xFFu) + (unsigned)PM::EXP) | (127u << 8));
    i32x8 qf[KS8];
    if constexpr (HYB == 0) {   // (the pv form has no e4m3 image of Q: a.q8 is null there)
        const uint8_t* qp = a.q8 + ((long)bh * a.Sq_pad + grow) * D8 + 32 * hh;  // rows < Sq_pad always exist (zero-padded)
#pragma unroll
        for (int ks = 0; ks < KS8; ++ks) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(qp + 64 * ks);
            const i32x4 hi = *reinterpret_cast<const i32x4*>(qp + 64 * ks + 16);
            qf[ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }

    // "pv" form: the 2-byte Q fragments (B operand of v_mfma_f32_32x32x16: lane (r, hh) holds k = 16 ks + 8 hh .. + 7), scaled
    using HT = typename std::conditional<HYB == 2, fp16_tag, bf16_tag>::type;
    constexpr int KSH = D8 / 16;          // k-steps of the 2-byte Q . K^T per 32-key half
    s16x8 qh[HYB ? KSH : 1];
    if constexpr (HYB != 0) {
        const unsigned short* qp16 = a.q16 + (long)b * a.qsb + (long)h * a.qsh + (long)grow * a.qss + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            uint4 raw = make_uint4(0, 0, 0, 0);
            if (grow < a.Sq) raw = *reinterpret_cast<const uint4*>(qp16 + 16 * ks);
            const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = rsa_to_f32<HT>((unsigned short)(w4[e] & 0xFFFF)) * a.qk_scale;
                f[2 * e + 1] = rsa_to_f32<HT>((unsigned short)(w4[e] >> 16)) * a.qk_scale;
            }
            qh[ks] = Elem<HT>::cvt8(f);
        }
    }

    // ---------------- LDS-DMA staging ----------------
    // K tile [64 keys][128 B]: 1-KiB piece pc = rows 8pc..8pc+7; wave w moves pieces w and w+4 (same swizzle phase).
    // V tile [128 d][64 B]:    1-KiB piece pc = rows 16pc..16pc+15; wave w moves pieces w and w+4.
    // Head dim 64: both tiles are [64 rows][64 B] = 4 pieces, wave w moves piece w, and K takes V's swizzle.
    // The LDS image is lane-linear, so the bank swizzle is applied to the SOURCE chunk.
    const unsigned char* kbase = a.k8 + (long)bh * a.Sk_pad * D8;
    const unsigned char* vbase = a.v8t + (long)bh * a.Sk_pad * D8;  // Sk_pad/64 tiles x TILE8 bytes
    const unsigned voffv = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
    const unsigned voffk = D8 == 128 ? (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ (4 * (wv & 1) + (lane >> 4))) << 4)) : voffv;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto dma2 = [&](const unsigned char* tile_src, unsigned lds_dst, unsigned voff) {
#pragma unroll
        for (int j = 0; j < NPC8; ++j) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(voff), "s"(tile_src + (wv + 4 * j) * 1024), "s"(lds_dst + (wv + 4 * j) * 1024)
                         : "memory");
        }
    };
    // "pv" form: the 64-key K tile from the 2-byte tensor, image and swizzle of rsa_attn_kernel.hip (groups of 4 pieces = 16 rows at
    // head dim 128, 32 rows at 64; wave w moves piece w of every group; source chunk XOR-swizzled, rows past the last valid key clamped)
    constexpr int CHR_H = D8 / 8, RPI_H = 1024 / (2 * D8), NPK_H = 64 / (4 * RPI_H);   // chunks per row, rows per piece, K pieces per wave
    const int kv_limit_h = hi_max < a.Sk ? hi_max : a.Sk;
    const int rsub_h = lane / CHR_H, cl_h = lane % CHR_H, rowl_h = wv * RPI_H + rsub_h;
    const int gsw_h = D8 == 128 ? (cl_h ^ (((rowl_h & 3) << 2) | ((rowl_h >> 2) & 3))) : (cl_h ^ ((rowl_h >> 1) & 7));
    auto dma_k = [&](int key0, int slot) {
        if constexpr (HYB == 0) {
            dma2(kbase + (long)key0 * D8, lds_base + slot * TILE8, voffk);
        } else {
            const unsigned char* kb16 = reinterpret_cast<const unsigned char*>(a.k16 + (long)b * a.ksb + (long)h * a.ksh);
            const unsigned ld0 = lds_base + slot * TILEK + wv * 1024;
#pragma unroll
            for (int j = 0; j < NPK_H; ++j) {
                int krow = key0 + 4 * RPI_H * j + rowl_h;
                krow = krow < kv_limit_h ? krow : kv_limit_h - 1;
                const unsigned vo = (unsigned)(((long)krow * a.kss + gsw_h * 8) * 2);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(vo), "s"(kb16), "s"(ld0 + j * 4096) : "memory");
            }
        }
    };
    auto dma_v = [&](int key0, int slot) {
        dma2(vbase + (long)(key0 >> 6) * TILE8, lds_base + VBASE + slot * TILE8, voffv);
    };

    // ---------------- state ----------------
    f32x16 o[DT8];
#pragma unroll
    for (int dt = 0; dt < DT8; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.0f;
    float m_run = -INFINITY;
    f32x4 lacc = {0.0f, 0.0f, 0.0f, 0.0f};  // row sum of the rounded P (all four registers hold the lane's own row)
    // A operand of the row-sum product: lane (row a = l & 15, k-block b = l >> 4) is all ones iff (b & 1) == ((a >> 2) & 1);
    // B k-block b of column c is P of query row c + 16 (b & 1), lane half b >> 1, so C[a][c] = l(row c + 16 ((a >> 2) & 1))
    // and the lane that owns C rows 4 (l >> 4) .. +3 of column l & 15 is exactly the lane of that query row.
    const int ones_off = ((((lane >> 4) & 1) == ((lane >> 2) & 1)) ? 0 : 32);
    f32x16 mblk;  // P_BASE - U m_eff in all 16 registers: the start value of every QK^T chain (m_eff = 0 while m_run = -inf)
#pragma unroll
    for (int i = 0; i < 16; ++i) mblk[i] = P_BASE;

    // per-lane read offsets (slot base added per step)
    int koff[2][KS8][2];  // [sub][ks][chunk]
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int row = 32 * sub + r;
#pragma unroll
        for (int ks = 0; ks < KS8; ++ks)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
                koff[sub][ks][c2] = D8 == 128 ? row * 128 + (((4 * ks + 2 * hh + c2) ^ ((row >> 1) & 7)) << 4)
                                              : row * 64 + (((2 * hh + c2) ^ ((row >> 2) & 3)) << 4);
    }
    int voff_rd[2];  // row r of a 32-row d-tile; +2048 per d-tile
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) voff_rd[c2] = r * 64 + (((2 * hh + c2) ^ ((r >> 2) & 3)) << 4);

    // read addresses of the hand-placed block (LDS byte addresses; ring slot, sub-tile and d-tile are immediates)
    typename std::conditional<D8 == 128, i32x4, i32x2>::type ka;
    i32x2 va;
#pragma unroll
    for (int ks = 0; ks < KS8; ++ks)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) ka[2 * ks + c2] = (int)lds_base + koff[0][ks][c2];
    va[0] = (int)lds_base + voff_rd[0];
    va[1] = (int)lds_base + voff_rd[1];
    const int ona = (int)lds_base + ONES + ones_off;
    // pv form: K fragment of k-step ks = 16-byte chunk 2 ks + hh of row r (+ 8 192 per 32-key half, + 16 384 per ring slot):
    // tile_off's XOR puts ks into address bits 5..7, the block derives the eight addresses from k-step 0's (lds_base is 1 KiB
    // aligned: the kernel's first dynamic LDS byte).  The V addresses carry the V ring's base.
    const int kah = (int)lds_base + (HYB ? tile_off<D8>(r, hh) : 0);
    i32x2 vah;
    // (the hand-placed pv block keeps the row-sum product's A operand -- e4m3 1.0 or 0 in every byte -- in registers)
    i32x8 onesv;
#pragma unroll
    for (int i = 0; i < 8; ++i) onesv[i] = ones_off == 0 ? 0x38383838 : 0;
    vah[0] = va[0] + VBASE;
    vah[1] = va[1] + VBASE;
    f32x16 qv[D8 / 64];
    if constexpr (HYB != 0) {
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            const i32x4 w = __builtin_bit_cast(i32x4, qh[ks]);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[ks >> 2][4 * (ks & 3) + e] = __int_as_float(w[e]);
        }
    }
    // LDS-DMA inside the block (DMAB): lane offsets of a tile operand's two pieces (wave w moves pieces w and w + 4), LDS address
    // of the wave's first piece in slot 0 of the K ring
    const i32x2 dv2 = {(int)voffv, (int)voffv + 4096};
    const i32x2 dk2 = {(int)voffk, (int)voffk + 4096};
    const unsigned ldsw = lds_base + (unsigned)wv * 1024u;

    auto ld32 = [&](const unsigned char* p0, const unsigned char* p1) -> i32x8 {
        const i32x4 lo = *reinterpret_cast<const i32x4*>(p0);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(p1);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // S^T (two 32-key halves) of the tile in K slot `slot`
    auto qk_tile = [&](int slot, f32x16 (&S)[2], int sc_k) {
        if constexpr (HYB != 0) {
            const unsigned char* kt16 = lds + slot * TILEK;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 acc = mblk;
#pragma unroll
                for (int ks = 0; ks < KSH; ++ks) {
                    const s16x8 kf = *reinterpret_cast<const s16x8*>(kt16 + tile_off<D8>(32 * sub + r, 2 * ks + hh));
                    acc = Elem<HT>::mfma(kf, qh[ks], acc);
                }
                S[sub] = acc;
            }
            return;
        }
        const unsigned char* kt_ = lds + slot * TILE8;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            S[sub] = mfma8s(ld32(kt_ + koff[sub][0][0], kt_ + koff[sub][0][1]), qf[0], mblk, sc_k, sc_b);
            if constexpr (KS8 == 2)
                S[sub] = mfma8s(ld32(kt_ + koff[sub][1][0], kt_ + koff[sub][1][1]), qf[1], S[sub], sc_k, sc_b);
        }
    };
    auto rowmax_tile = [&](const f32x16 (&S)[2]) -> float {
        float m = S[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, S[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, S[1][i]);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // = U (score_max - m_eff) + P_BASE
    };
    auto apply_mask = [&](f32x16 (&S)[2], int key0) {
        int kbase = key0 + 4 * hh;
        asm volatile("" : "+v"(kbase));   // rare branch: keep its 32 key indices out of the loop's live registers
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = kbase + 32 * sub + (i & 3) + 8 * (i >> 2);
                if (kk < lo_r || kk >= hi_r) S[sub][i] = -INFINITY;
            }
    };

    int kq1 = 0, kq2 = 0, kq3 = 0;  // first keys of tiles t+1, t+2, t+3
    unsigned sw0 = 0, sw1 = 0, sw2 = 0, sw3 = 0;  // entries >> 16 (K exponent | V exponent << 8) of tiles t .. t+3, wave-uniform

    // TS = tile & 3 (compile-time: every LDS address is a loop-invariant VGPR plus an immediate)
    auto step = [&](auto TS, int tile, int key0, f32x16 (&S_cur)[2], float& mx_cur, f32x16 (&S_nxt)[2], float& mx_nxt) {
        const int ts = TS;  // integral_constant (static LDS addresses) or the runtime tile & 3
        const int sc_a = (int)((sw1 & 0xFFu) | (sw0 & 0xFF00u));   // K(tile + 1) in byte 0, V(tile) in byte 1
        if (DMAB || tile + 2 < n_tiles) {   // (the newest tile's pieces may stay in flight: 2 NPC8 per wave; pv form: 4 of K + 2 of V)
            if constexpr (HYB != 0 && D8 == 128) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (HYB != 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");     // head dim 64: 2 of K + 1 of V
            else if constexpr (NPC8 == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifndef RSA_PVX_NOBAR
        __syncthreads();
#endif
        if constexpr (LWIN != 0) {
            // every LWIN / 2 items: entries [item + LWIN / 2, item + LWIN) replace those of [item - LWIN / 2, item), which are behind
            // every reader (the entry of tile + 5 is the furthest read ahead); they are first read LWIN / 2 items -- barriers -- later
            if ((tile & (LWIN - 1)) == 0 && tile > 0) {
                const int base = (tile >> 1) + LWIN / 2;
                for (int i = base + t; i < base + LWIN / 2 && i < n_items; i += 256) lds_list[i & (LWIN - 1)] = make_entry(i);
            }
        }
        // ring slots: tile & 3 of four (ts), or tile mod 3 of three in the pv form
        // (pv form, hand-placed: TS = tile % 6 at compile time)
        const int t3 = (PIPE_OPT & 1) != 0 ? ts % 3 : tile % 3;
        const int ks_dma = HYB ? t3 : (ts + 3) & (NSLOT - 1), vs_dma = HYB ? (t3 + 2) % 3 : (ts + 2) & (NSLOT - 1);
        const int ks_nxt = HYB ? (t3 + 1) % 3 : (ts + 1) & (NSLOT - 1), vs_cur = HYB ? t3 : ts;
#if defined(RSA_PVX_HOTDMA)   // (RSA_PVX_*: timing experiments of tools/history/r5_pvx.sh, never defined in the product)
        if (tile + 3 < n_tiles) dma_k((tile & 1) * 64, ks_dma);   // every piece issued, every line L2-resident
        if (tile + 2 < n_tiles) dma_v((tile & 1) * 64, vs_dma);
#elif !defined(RSA_PVX_NODMA)
        if constexpr (!DMAB) {
            if (tile + 3 < n_tiles) dma_k(kq3, ks_dma);
            if (tile + 2 < n_tiles) dma_v(kq2, vs_dma);
        }
#endif
        // ---- head (rare branches): boundary mask, deferred rescale ----
        if (key0 < lo_max || key0 + 64 > hi_min) {
            apply_mask(S_cur, key0);
            mx_cur = rowmax_tile(S_cur);
        }
        // mx_cur and S_cur are relative to the block's reference: U (score - m_eff) + P_BASE
        const bool grow = (m_run == -INFINITY) ? (mx_cur > -INFINITY) : (mx_cur > P_GROW);
        if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
            asm volatile("s_nop 15\n\ts_nop 5" ::: "memory");   // the block's last (16-pass) MFMA wrote O: wait states before VALU
            const float m_eff_old = (m_run == -INFINITY) ? 0.0f : m_run;
            const float m_new = fmaxf(m_run, (mx_cur - P_BASE) * (1.0f / PM::U) + m_eff_old);
            const float m_eff = (m_new == -INFINITY) ? 0.0f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_eff);
            const float shift = PM::U * (m_eff - m_eff_old);
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 4; ++i) lacc[i] *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT8; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int i = 0; i < 16; ++i) S_cur[sub][i] -= shift;
#pragma unroll
            for (int i = 0; i < 16; ++i) mblk[i] = P_BASE - PM::U * m_eff;
        }

        // ---- pipelined block ----
        if constexpr ((PIPE_OPT & 1) != 0) {
            constexpr int tsc = decltype(TS)::value;
            // S_cur is SA on even tiles, SB on odd ones (tile & 1 == TS & 1)
            if constexpr (HYB != 0 && DMAB) {
                // K(tile + 3): the four pieces' rows, clamped like dma_k's; V(tile + 2): first byte of the wave's first piece
                typename std::conditional<D8 == 128, i32x4, i32x2>::type dkp;
#pragma unroll
                for (int j = 0; j < NPK_H; ++j) {
                    int krow = kq3 + 4 * RPI_H * j + rowl_h;
                    krow = krow < kv_limit_h ? krow : kv_limit_h - 1;
                    dkp[j] = (int)(unsigned)(((long)krow * a.kss + gsw_h * 8) * 2);
                }
                const unsigned char* kb16 = reinterpret_cast<const unsigned char*>(a.k16 + (long)b * a.ksb + (long)h * a.ksh);
                const unsigned char* vsrc = vbase + (long)(kq2 >> 6) * TILE8 + wv * 1024;
                if constexpr (D8 == 128) {
                    if constexpr ((tsc & 1) == 0) k5f8h_block_dma<tsc, HYB, D8>(o, qv, S_cur, S_nxt, mblk, lacc, mx_nxt, sc_a, sc_b, kah, vah, onesv, dkp, dv2, kb16, vsrc, ldsw);
                    else k5f8h_block_dma<tsc, HYB, D8>(o, qv, S_nxt, S_cur, mblk, lacc, mx_nxt, sc_a, sc_b, kah, vah, onesv, dkp, dv2, kb16, vsrc, ldsw);
                } else {
                    const int dv1 = (int)voffv;
                    if constexpr ((tsc & 1) == 0) k5f8h_block_dma<tsc, HYB, D8>(o, qv, S_cur, S_nxt, mblk, lacc, mx_nxt, sc_a, sc_b, kah, vah, onesv, dkp, dv1, kb16, vsrc, ldsw);
                    else k5f8h_block_dma<tsc, HYB, D8>(o, qv, S_nxt, S_cur, mblk, lacc, mx_nxt, sc_a, sc_b, kah, vah, onesv, dkp, dv1, kb16, vsrc, ldsw);
                }
            } else if constexpr (HYB != 0) {
                if constexpr ((tsc & 1) == 0) k5f8h_block<tsc, HYB, D8>(o, qv, S_cur, S_nxt, mblk, lacc, mx_nxt, sc_a, sc_b, kah, vah, onesv);
                else k5f8h_block<tsc, HYB, D8>(o, qv, S_nxt, S_cur, mblk, lacc, mx_nxt, sc_a, sc_b, kah, vah, onesv);
            } else if constexpr (DMAB && D8 == 64) {
                const unsigned char* ksrc = kbase + (long)kq3 * D8 + wv * 1024;
                const unsigned char* vsrc = vbase + (long)(kq2 >> 6) * TILE8 + wv * 1024;
                if constexpr ((tsc & 1) == 0) k5f8_block_dma64<tsc>(o, qf, S_cur, S_nxt, mblk, lacc, mx_nxt, sc_a, sc_b, ka, va, onesv, (int)voffk, (int)voffv, ksrc, vsrc, ldsw);
                else k5f8_block_dma64<tsc>(o, qf, S_nxt, S_cur, mblk, lacc, mx_nxt, sc_a, sc_b, ka, va, onesv, (int)voffk, (int)voffv, ksrc, vsrc, ldsw);
            } else if constexpr (DMAB) {
                const unsigned char* ksrc = kbase + (long)kq3 * D8 + wv * 1024;
                const unsigned char* vsrc = vbase + (long)(kq2 >> 6) * TILE8 + wv * 1024;
                if constexpr ((tsc & 1) == 0) k5f8_block_dma<tsc>(o, qf, S_cur, S_nxt, mblk, lacc, mx_nxt, sc_a, sc_b, ka, va, onesv, dk2, dv2, ksrc, vsrc, ldsw);
                else k5f8_block_dma<tsc>(o, qf, S_nxt, S_cur, mblk, lacc, mx_nxt, sc_a, sc_b, ka, va, onesv, dk2, dv2, ksrc, vsrc, ldsw);
            } else
            if constexpr ((tsc & 1) == 0) k5f8_block<tsc, CODEMAP, D8>(o, qf, S_cur, S_nxt, mblk, lacc, mx_nxt, sc_a, sc_b, ka, va, ona);
            else k5f8_block<tsc, CODEMAP, D8>(o, qf, S_nxt, S_cur, mblk, lacc, mx_nxt, sc_a, sc_b, ka, va, ona);
            return;
        }
        if constexpr (PIPE_OPT & 2) __builtin_amdgcn_s_setprio(2);
        qk_tile(ks_nxt, S_nxt, sc_a);
        i32x8 pb;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                // the packed word is built in a score register that is dead by now (no zero-initialised temporary)
                int word = __float_as_int(S_cur[0][4 * sub + w4]);
                if constexpr (CODEMAP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        word = (int)__builtin_amdgcn_cvt_pk_u8_f32(S_cur[sub][4 * w4 + e], (unsigned)e, (unsigned)word);
                } else {
                    float p4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) p4[e] = __builtin_amdgcn_exp2f(S_cur[sub][4 * w4 + e]);
                    word = __builtin_amdgcn_cvt_pk_fp8_f32(p4[0], p4[1], word, false);
                    word = __builtin_amdgcn_cvt_pk_fp8_f32(p4[2], p4[3], word, true);
                }
                pb[4 * sub + w4] = word;
            }
        lacc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ld32(lds_ones + ones_off, lds_ones + ones_off + 16), pb,
                                                                lacc, 0, 0, 0, 0, 0, 0);
        const unsigned char* vt_ = lds + VBASE + vs_cur * TILE8;
#pragma unroll
        for (int dt = 0; dt < DT8; ++dt)
            o[dt] = mfma8s1(ld32(vt_ + dt * 2048 + voff_rd[0], vt_ + dt * 2048 + voff_rd[1]), pb, o[dt], sc_a, sc_b);
        mx_nxt = rowmax_tile(S_nxt);
        if constexpr (PIPE_OPT & 2) __builtin_amdgcn_s_setprio(0);
    };

    // ---------------- prologue + main loop ----------------
    if (tail < 0 && qblk < a.NBv) rsa_gsync_wait(a.gsync, gs_tk, n_items, a.NB_total, a.gsync_ratio);   // aligned starts (off by default here)
    f32x16 SA[2], SB[2];
    float mxA = -INFINITY, mxB = -INFINITY;
    int key0 = 0;
    if (n_tiles > 0) {
        const unsigned e0 = entry_sc(0), e1 = entry_sc(1), e2 = entry_sc(2), e3 = entry_sc(3);
        key0 = key0_from(e0, 0); kq1 = key0_from(e1, 1); kq2 = key0_from(e2, 2); kq3 = key0_from(e3, 3);
        sw0 = e0 >> 16; sw1 = e1 >> 16; sw2 = e2 >> 16; sw3 = e3 >> 16;
        dma_k(key0, 0);
        dma_v(key0, 0);
        if (n_tiles > 1) { dma_k(kq1, 1); dma_v(kq1, 1); }
        if (n_tiles > 2) dma_k(kq2, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        qk_tile(0, SA, (int)(sw0 & 0xFFu));
        mxA = rowmax_tile(SA);
    }
    // The kept-list entry of tile+4 is read from LDS one advance() EARLY into a register (pref_raw) and only made scalar
    // here: the LDS round trip (~100 cycles, once per tile and wave) is off the wave's critical path.
    unsigned pref_raw = n_tiles > 0 ? raw_item(4) : 0u;
    auto advance = [&](int tile) {  // after finishing `tile`: shift the queues, tile+4's entry from the prefetched register
        key0 = kq1;
        kq1 = kq2;
        kq2 = kq3;
        sw0 = sw1; sw1 = sw2; sw2 = sw3;
        const unsigned e4 = (unsigned)__builtin_amdgcn_readfirstlane((int)pref_raw);
        kq3 = key0_from(e4, tile + 4);
        sw3 = e4 >> 16;
        pref_raw = raw_item(tile + 5);
    };
    if constexpr ((PIPE_OPT & 1) != 0 && HYB != 0) {   // pv form: six tiles per trip (three ring slots x two score registers sets)
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>;
        using I5 = std::integral_constant<int, 5>;
        int tile = 0;
        for (; tile + 5 < n_tiles; tile += 6) {
            step(I0{}, tile, key0, SA, mxA, SB, mxB);
            advance(tile);
            step(I1{}, tile + 1, key0, SB, mxB, SA, mxA);
            advance(tile + 1);
            step(I2{}, tile + 2, key0, SA, mxA, SB, mxB);
            advance(tile + 2);
            step(I3{}, tile + 3, key0, SB, mxB, SA, mxA);
            advance(tile + 3);
            step(I4{}, tile + 4, key0, SA, mxA, SB, mxB);
            advance(tile + 4);
            step(I5{}, tile + 5, key0, SB, mxB, SA, mxA);
            advance(tile + 5);
        }
        if (tile < n_tiles) { step(I0{}, tile, key0, SA, mxA, SB, mxB); advance(tile); }
        if (tile + 1 < n_tiles) { step(I1{}, tile + 1, key0, SB, mxB, SA, mxA); advance(tile + 1); }
        if (tile + 2 < n_tiles) { step(I2{}, tile + 2, key0, SA, mxA, SB, mxB); advance(tile + 2); }
        if (tile + 3 < n_tiles) { step(I3{}, tile + 3, key0, SB, mxB, SA, mxA); advance(tile + 3); }
        if (tile + 4 < n_tiles) step(I4{}, tile + 4, key0, SA, mxA, SB, mxB);
    } else
    if constexpr ((PIPE_OPT & 1) != 0) {   // four tiles per trip: the ring slot is a compile-time constant of every block
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        int tile = 0;
        for (; tile + 3 < n_tiles; tile += 4) {
            step(I0{}, tile, key0, SA, mxA, SB, mxB);
            advance(tile);
            step(I1{}, tile + 1, key0, SB, mxB, SA, mxA);
            advance(tile + 1);
            step(I2{}, tile + 2, key0, SA, mxA, SB, mxB);
            advance(tile + 2);
            step(I3{}, tile + 3, key0, SB, mxB, SA, mxA);
            advance(tile + 3);
        }
        if (tile < n_tiles) { step(I0{}, tile, key0, SA, mxA, SB, mxB); advance(tile); }
        if (tile + 1 < n_tiles) { step(I1{}, tile + 1, key0, SB, mxB, SA, mxA); advance(tile + 1); }
        if (tile + 2 < n_tiles) step(I2{}, tile + 2, key0, SA, mxA, SB, mxB);
    } else {
        int tile = 0;
        for (; tile + 1 < n_tiles; tile += 2) {
            step(tile & 3, tile, key0, SA, mxA, SB, mxB);
            advance(tile);
            step((tile + 1) & 3, tile + 1, key0, SB, mxB, SA, mxA);
            advance(tile + 1);
        }
        if (tile < n_tiles) step(tile & 3, tile, key0, SA, mxA, SB, mxB);
    }

    // ---------------- epilogue ----------------
    if constexpr (DMAB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the pieces issued past the end of the walk: the LDS is given back at exit)
    asm volatile("s_nop 15\n\ts_nop 5" ::: "memory");   // (the last block's last MFMA -> the reads of O / l below)
    const float l_tot = lacc[0];
    if ((a.mode == MODE_SPARSE && a.tsplit > 1 && qblk >= a.NBv) || tail >= 0) {
        // split-KV partial of a text block or of a tail piece (merged by rsa_attn.hip's combine kernels): O in V's units, m, l
        const int ntq = a.NQB - a.NBv;
        const int rowb = 32 * wv + r;
        float* pp = tail >= 0 ? a.tail_part + ((long)tail * RSA_BLOCK + rowb) * (D8 + 2)
                              : a.tpart + ((((long)bh * ntq + (qblk - a.NBv)) * a.tsplit + tsp) * RSA_BLOCK + rowb) * (D8 + 2);
#pragma unroll
        for (int dt = 0; dt < DT8; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = 32 * dt + 8 * g + 4 * hh;
                *reinterpret_cast<float2*>(pp + d0) = make_float2(o[dt][4 * g + 0], o[dt][4 * g + 1]);
                *reinterpret_cast<float2*>(pp + d0 + 2) = make_float2(o[dt][4 * g + 2], o[dt][4 * g + 3]);
            }
        if (hh == 0) *reinterpret_cast<float2*>(pp + D8) = make_float2(m_run, l_tot);
        return;
    }
    if (!(store_r || zero_r)) return;
    float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    float Rv = 1.0f;
    // the compensation values this lane adds and R: all loads issued here, back to back, one wait (loaded per (dt, g) inside the
    // store loop each load's latency is exposed in turn: round 4, profiles/r04_k5_w64.md)
    float4 cv[DT8][4];
#pragma unroll
    for (int dt = 0; dt < DT8; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) cv[dt][g] = make_float4(0, 0, 0, 0);
    if (rectify) {
        const long rowi = (long)bh * a.NBv + qblk;
        if (!zero_r) {
            const float* cp = a.comp + rowi * D8;
#pragma unroll
            for (int dt = 0; dt < DT8; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) cv[dt][g] = *reinterpret_cast<const float4*>(cp + 32 * dt + 8 * g + 4 * hh);
        }
        Rv = a.R[rowi];
    }
    if (zero_r) inv = 0.0f;
    const float sc = inv * Rv;
    unsigned short* op = a.out + (long)b * a.osb + (long)h * a.osh + (long)grow * a.oss;
    // (16-byte stores after a v_permlane32_swap regroup, the 2-byte kernel's default, measured neutral here: 9.60 vs 9.60 ms)
#pragma unroll
    for (int dt = 0; dt < DT8; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            const float4 c4 = cv[dt][g];
            const float v0 = o[dt][4 * g + 0] * sc + c4.x;
            const float v1 = o[dt][4 * g + 1] * sc + c4.y;
            const float v2 = o[dt][4 * g + 2] * sc + c4.z;
            const float v3 = o[dt][4 * g + 3] * sc + c4.w;
            uint2 pk;
            if (a.out_fp16) {
                pk.x = (unsigned)Elem<fp16_tag>::from_f32(v0) | ((unsigned)Elem<fp16_tag>::from_f32(v1) << 16);
                pk.y = (unsigned)Elem<fp16_tag>::from_f32(v2) | ((unsigned)Elem<fp16_tag>::from_f32(v3) << 16);
            } else {
                pk.x = (unsigned)Elem<bf16_tag>::from_f32(v0) | ((unsigned)Elem<bf16_tag>::from_f32(v1) << 16);
                pk.y = (unsigned)Elem<bf16_tag>::from_f32(v2) | ((unsigned)Elem<bf16_tag>::from_f32(v3) << 16);
            }
            *reinterpret_cast<uint2*>(op + d0) = pk;
        }
    }
}

int g_fp8_variant = 0;
int launch_attn8(Attn8Args& a, int BH, int D8, size_t tpart_bytes, hipStream_t s, int hyb = 0) {
    const int ntq = a.NQB - a.NBv;
    if (a.tpart && tpart_bytes == 0) return RSA_ERR_WORKSPACE;   // capacity not declared (rsa_buffers.tpart_bytes, 0.5.0)
    const int n_txt_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
    a.tsplit = 1; a.tper = n_txt_items;
    if (a.mode == MODE_SPARSE && ntq > 0 && a.tpart && rsa_text_split_enabled() && n_txt_items >= 32) {
        const int sp = n_txt_items / 16;
        int cap = rsa_text_split_capacity(tpart_bytes, BH, ntq, D8);   // (sized for RSA_TEXT_SPLIT = 32 by rsa_buffer_bytes; the 2-byte kernel uses them on short grids)
        if (cap > 16) cap = 16;
        a.tsplit = sp > cap ? cap : sp;
        if (a.tsplit < 2) a.tsplit = 1;
        a.tper = (n_txt_items + a.tsplit - 1) / a.tsplit;
    }
    const int n_heavy = ntq > 0 ? BH * ntq * a.tsplit : 0;
    a.heavy_last = a.tsplit > 1 && rsa_text_last_enabled();
    a.BH = BH;
    a.n_heavy_pad = (n_heavy + 7) & ~7;
    a.NBp = (a.NBv + 7) & ~7;
    long nblocks = (long)a.n_heavy_pad + (long)BH * a.NBp;
    a.tail_first = a.tail_n = a.tail_p = 0; a.tail_part = nullptr;
    // (as rsa_attn.hip::launch_attn, but only for layouts without text rows: with this kernel's shorter lives and two waves per
    // SIMD the split measured +1.6 % on Wan2.2-TI2V and -2.3 % at 3 heads of the HunyuanVideo shape, where the text pieces end the
    // launch either way: profiles/r04_k5_tail_split.txt)
    if (D8 == 128 && a.mode == MODE_SPARSE && a.tpart && n_heavy == 0 &&
        rsa_plan_tail_split((long)BH * a.NBp, a.n_heavy_pad, &a.tail_first, &a.tail_n, &a.tail_p) &&
        !rsa_tail_fits(tpart_bytes, BH, ntq, D8, a.tail_n, a.tail_p))
        a.tail_first = a.tail_n = a.tail_p = 0;
    if (a.tail_n > 0) {
        a.tail_part = a.tpart + (long)BH * ntq * RSA_TEXT_SPLIT * RSA_BLOCK * (D8 + 2);
        nblocks = (long)a.tail_first + (long)a.tail_n * a.tail_p + a.n_heavy_pad;
    }
    if (nblocks <= 0) return RSA_OK;
    if (nblocks > 0x7FFFFFFF) return RSA_ERR_UNSUPPORTED;
    if (a.NB_total > 8192) return RSA_ERR_UNSUPPORTED;
    // pv form: a K row is addressed as a 32-bit byte offset from its head's base (the LDS-DMA's vector offset): refuse spans the
    // offset cannot reach instead of reading wrapped addresses (a [B,S,H,D]-strided K at 6 144 B per row wraps near 700 k keys;
    // a head-contiguous K at 256 B per row reaches 16 M).  The 2-byte and e4m3 kernels walk a 64-bit scalar base instead.
    if (hyb != 0 && (unsigned long long)a.Sk * (unsigned long long)(a.kss < 0 ? -a.kss : a.kss) * 2ull >= (1ull << 32))
        return RSA_ERR_UNSUPPORTED;
    const size_t lds_bytes = (size_t)2 * NSLOT * 64 * D8 + 64 + (((size_t)a.NB_total * 4 + 15) & ~(size_t)15);
    if (hyb != 0 && D8 == 64) {   // the pv form at head dim 64 (K 3 x 8 KiB, V 3 x 4 KiB): product and compiled twin
        const size_t n_list = a.NB_total < RSA_PV_LIST_WINDOW ? a.NB_total : RSA_PV_LIST_WINDOW;
        const size_t lds_h = (size_t)3 * 8192 + (size_t)3 * 4096 + 64 + ((n_list * 4 + 15) & ~(size_t)15);
        if (g_fp8_variant == 1) {
            if (hyb == 2) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<6, 64, 2>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
            else RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<6, 64, 1>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
        } else {
            if (hyb == 2) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<15, 64, 2>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
            else RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<15, 64, 1>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
        }
    } else
    if (hyb != 0) {   // the pv form: 2-byte Q . K^T, e4m3 P . V (three-slot rings: K 3 x 16 KiB, V 3 x 8 KiB)
        if (D8 != 128) return RSA_ERR_UNSUPPORTED;
        const size_t n_list = a.NB_total < RSA_PV_LIST_WINDOW ? a.NB_total : RSA_PV_LIST_WINDOW;     // (the kernel's LWIN)
        const size_t lds_h = (size_t)3 * 16384 + (size_t)3 * 8192 + 64 + ((n_list * 4 + 15) & ~(size_t)15);
        // (tuning key fp8_variant 1 = the compiled twin of the hand-placed block: same arithmetic, hipcc's schedule)
        // and 3 = the hand-placed block with the staging behind the barrier, as the other forms have it)
        if (g_fp8_variant == 1) {
            if (hyb == 2) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<6, 128, 2>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
            else RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<6, 128, 1>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
        } else if (g_fp8_variant == 3) {
            if (hyb == 2) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<7, 128, 2>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
            else RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<7, 128, 1>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
        } else {
            if (hyb == 2) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<15, 128, 2>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
            else RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<15, 128, 1>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_h, s);
        }
    } else
    if (D8 == 64) {   // head dim 64: the product form, its compiled twin (1), the hand-placed block with the staging behind the barrier (3)
        if (g_fp8_variant == 1) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<6, 64>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s);
        else if (g_fp8_variant == 3) RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<7, 64>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s);
        else RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<15, 64>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s);
    } else {
        switch (g_fp8_variant) {   // tuning key fp8_variant: 0 = product; 1, 2 = the two verification forms the tests compare it with
            case 1: RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<6>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s); break;   // product arithmetic, hipcc's schedule
            case 2: RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<3>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s); break;   // exact-exponential P, hand-placed
            case 3: RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<7>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s); break;   // the product's block, staging behind the barrier (rounds 3-4)
            default: RSA_LAUNCH_GSYNC(2, (bsfwd_fp8_kernel<15>), a, a.mode == MODE_SPARSE, dim3((unsigned)nblocks), 256, lds_bytes, s);
        }
    }
    const int st = rsa_launch_status();
    if (st != RSA_OK) return st;
    if (a.tail_n > 0) {
        const int st2 = rsa_launch_tail_combine(a.tail_part, a.out, a.osb, a.osh, a.oss, a.H, a.NBv, a.NBp, a.tail_first, a.tail_n,
                                                a.tail_p, a.R, a.comp, a.Sq, a.out_fp16 ? RSA_FP16 : RSA_BF16, s);
        if (st2 != RSA_OK) return st2;
    }
    if (a.tsplit <= 1) return st;
    return rsa_launch_text_combine(a.tpart, a.out, a.osb, a.osh, a.oss, D8, a.H, a.NBv, ntq, a.tsplit, a.q_text_end, a.Sq,
                                   BH, a.out_fp16 ? RSA_FP16 : RSA_BF16, s);
}

int check_out8(const rsa_out4& o) {
    if (!o.ptr || (reinterpret_cast<uintptr_t>(o.ptr) & 7)) return RSA_ERR_BAD_ARG;
    if ((o.stride_b % 4) || (o.stride_h % 4) || (o.stride_s % 4)) return RSA_ERR_BAD_ARG;
    return RSA_OK;
}

}  // namespace

void rsa_set_fp8_variant(int v) { g_fp8_variant = v; }

extern "C" int rsa_block_sparse_fwd_fp8(const rsa_layout* l, const rsa_fp8_operands* ops, const rsa_buffers* buf,
                                        rsa_out4 out, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (l->D != 128 && l->D != 64) return RSA_ERR_UNSUPPORTED;
    if (!ops || !ops->q8 || !ops->k8 || !ops->v8t || !ops->scales) return RSA_ERR_BAD_ARG;
    if ((st = check_out8(out))) return st;
    if (!buf || (l->NBv > 0 && (!buf->cols || !buf->counts))) return RSA_ERR_BAD_ARG;
    if ((buf->R == nullptr) != (buf->comp == nullptr)) return RSA_ERR_BAD_ARG;
    Attn8Args a;
    a.q8 = ops->q8; a.k8 = ops->k8; a.v8t = ops->v8t; a.exps = ops->scales; a.exps_stride = l->NB_total;
    a.out = static_cast<unsigned short*>(out.ptr); a.osb = out.stride_b; a.osh = out.stride_h; a.oss = out.stride_s;
    a.cols = buf->cols; a.counts = buf->counts; a.R = buf->R; a.comp = buf->comp; a.tpart = buf->tpart;
    a.mode = MODE_SPARSE; a.H = l->H; a.Sq = l->S; a.Sk = l->S;
    a.Sq_pad = a.Sk_pad = l->NB_total * RSA_BLOCK;
    a.NBv = l->NBv; a.NQB = l->NB_total; a.NB_total = l->NB_total;
    a.kv_valid = l->kv_valid; a.kv_text_valid = l->kv_text_valid;
    a.q_text_end = l->NBv * RSA_BLOCK + l->q_text_valid;
    a.q_split = 0; a.kv_split = 0; a.causal = 0;
    a.out_fp16 = l->dtype == RSA_FP16;
    a.q16 = a.k16 = nullptr; a.qsb = a.qsh = a.qss = a.ksb = a.ksh = a.kss = 0; a.qk_scale = 0.0f;
    return launch_attn8(a, l->B * l->H, l->D, buf->tpart_bytes, static_cast<hipStream_t>(stream));
}

// The "pv" form (round 5): Q . K^T on the 2-byte q and k themselves, e4m3 only for P and V (ops->v8t and the V bytes of
// ops->scales are read; q8 / k8 may be NULL).
extern "C" int rsa_block_sparse_fwd_fp8pv(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, const rsa_fp8_operands* ops,
                                          const rsa_buffers* buf, rsa_out4 out, void* stream) {
    int st = rsa_check_layout(l);
    if (st != RSA_OK) return st;
    if (l->D != 128 && l->D != 64) return RSA_ERR_UNSUPPORTED;
    if (!ops || !ops->v8t || !ops->scales) return RSA_ERR_BAD_ARG;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = check_out8(out))) return st;
    if (!buf || (l->NBv > 0 && (!buf->cols || !buf->counts))) return RSA_ERR_BAD_ARG;
    if ((buf->R == nullptr) != (buf->comp == nullptr)) return RSA_ERR_BAD_ARG;
    Attn8Args a;
    a.q8 = ops->q8; a.k8 = ops->k8; a.v8t = ops->v8t; a.exps = ops->scales; a.exps_stride = l->NB_total;
    a.out = static_cast<unsigned short*>(out.ptr); a.osb = out.stride_b; a.osh = out.stride_h; a.oss = out.stride_s;
    a.cols = buf->cols; a.counts = buf->counts; a.R = buf->R; a.comp = buf->comp; a.tpart = buf->tpart;
    a.mode = MODE_SPARSE; a.H = l->H; a.Sq = l->S; a.Sk = l->S;
    a.Sq_pad = a.Sk_pad = l->NB_total * RSA_BLOCK;
    a.NBv = l->NBv; a.NQB = l->NB_total; a.NB_total = l->NB_total;
    a.kv_valid = l->kv_valid; a.kv_text_valid = l->kv_text_valid;
    a.q_text_end = l->NBv * RSA_BLOCK + l->q_text_valid;
    a.q_split = 0; a.kv_split = 0; a.causal = 0;
    a.out_fp16 = l->dtype == RSA_FP16;
    a.q16 = static_cast<const unsigned short*>(q.ptr); a.qsb = q.stride_b; a.qsh = q.stride_h; a.qss = q.stride_s;
    a.k16 = static_cast<const unsigned short*>(k.ptr); a.ksb = k.stride_b; a.ksh = k.stride_h; a.kss = k.stride_s;
    a.qk_scale = (float)((1.0 / sqrt((double)l->D)) * 1.44269504 * 8.0);      // sm_scale * log2(e) * PMap<true>::U
    return launch_attn8(a, l->B * l->H, l->D, buf->tpart_bytes, static_cast<hipStream_t>(stream), l->dtype == RSA_FP16 ? 2 : 1);
}

extern "C" int rsa_rectified_attention_fp8(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                           const uint8_t* neighbor, int top_k, float p_remain, void* workspace,
                                           size_t workspace_bytes, void* fp8_workspace, size_t fp8_workspace_bytes,
                                           rsa_out4 out, void* stream) {
    rsa_buffers buf;
    rsa_fp8_operands ops;
    int st = rsa_carve_workspace(l, workspace, workspace_bytes, &buf);
    if (st != RSA_OK) return st;
    if ((st = rsa_carve_fp8_operands(l, fp8_workspace, fp8_workspace_bytes, &ops))) return st;
    if ((st = rsa_pool_stats_fp8(l, q, k, v, &buf, &ops, stream))) return st;
    if ((st = rsa_pooled_scores(l, k, &buf, stream))) return st;
    if ((st = rsa_select_mask(l, neighbor, top_k, p_remain, &buf, stream))) return st;
    if ((st = rsa_compensation(l, &buf, stream))) return st;
    return rsa_block_sparse_fwd_fp8(l, &ops, &buf, out, stream);
}

extern "C" int rsa_rectified_attention_fp8pv(const rsa_layout* l, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                             const uint8_t* neighbor, int top_k, float p_remain, void* workspace,
                                             size_t workspace_bytes, void* fp8_workspace, size_t fp8_workspace_bytes,
                                             rsa_out4 out, void* stream) {
    rsa_buffers buf;
    rsa_fp8_operands ops;
    int st = rsa_carve_workspace(l, workspace, workspace_bytes, &buf);
    if (st != RSA_OK) return st;
    if (l->D != 128 && l->D != 64) return RSA_ERR_UNSUPPORTED;
    if ((st = rsa_carve_fp8_operands(l, fp8_workspace, fp8_workspace_bytes, &ops))) return st;
    ops.q8 = nullptr; ops.k8 = nullptr;       // K1 then writes the V image and the V exponents only
    if ((st = rsa_pool_stats_fp8(l, q, k, v, &buf, &ops, stream))) return st;
    if ((st = rsa_pooled_scores(l, k, &buf, stream))) return st;
    if ((st = rsa_select_mask(l, neighbor, top_k, p_remain, &buf, stream))) return st;
    if ((st = rsa_compensation(l, &buf, stream))) return st;
    return rsa_block_sparse_fwd_fp8pv(l, q, k, &ops, &buf, out, stream);
}

// Dense attention (two-segment varlen semantics of rsa_dense_fwd) with fp8 operands: quantise, then the same kernel in
// its dense mode.  Workspace: rsa_dense_fp8_bytes.
static int dense_fwd_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                         int q_split, int kv_split, int causal, void* workspace, size_t workspace_bytes, rsa_out4 out,
                         void* stream, int pv = 0) {
    if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return RSA_ERR_BAD_ARG;
    if (D != 128 && D != 64) return RSA_ERR_UNSUPPORTED;
    if (dtype != RSA_BF16 && dtype != RSA_FP16) return RSA_ERR_UNSUPPORTED;
    if (q_split < 0 || q_split > Sq || kv_split < 0 || kv_split > Sk) return RSA_ERR_BAD_ARG;
    int st;
    if ((st = rsa_check_tensor(q)) || (st = rsa_check_tensor(k)) || (st = rsa_check_tensor(v)) ||
        (st = check_out8(out)))
        return st;
    rsa_fp8_operands ops;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((st = rsa_dense_quantize_fp8(B, H, Sq, Sk, D, dtype, q, k, v, workspace, workspace_bytes, &ops, s, pv))) return st;
    Attn8Args a;
    a.q8 = ops.q8; a.k8 = ops.k8; a.v8t = ops.v8t; a.exps = ops.scales;
    a.out = static_cast<unsigned short*>(out.ptr); a.osb = out.stride_b; a.osh = out.stride_h; a.oss = out.stride_s;
    a.cols = nullptr; a.counts = nullptr; a.R = nullptr; a.comp = nullptr; a.tpart = nullptr;
    a.mode = MODE_DENSE; a.H = H; a.Sq = Sq; a.Sk = Sk;
    a.NQB = (Sq + RSA_BLOCK - 1) / RSA_BLOCK; a.NBv = a.NQB; a.NB_total = (Sk + RSA_BLOCK - 1) / RSA_BLOCK;
    {   // every image of the dense producer is max(Sq, Sk) rounded up to 128 rows high (rsa_fp8.hip::dense_fp8_carve)
        const int nbm = a.NQB > a.NB_total ? a.NQB : a.NB_total;
        a.Sq_pad = a.Sk_pad = nbm * RSA_BLOCK;
        a.exps_stride = nbm;
    }
    a.kv_valid = Sk; a.kv_text_valid = Sk; a.q_text_end = 0;
    a.q_split = q_split; a.kv_split = kv_split; a.causal = causal;
    a.out_fp16 = dtype == RSA_FP16;
    a.q16 = a.k16 = nullptr; a.qsb = a.qsh = a.qss = a.ksb = a.ksh = a.kss = 0; a.qk_scale = 0.0f;
    if (pv) {   // scores from the 2-byte q and k themselves (rsa_block_sparse_fwd_fp8pv's operands)
        a.q16 = static_cast<const unsigned short*>(q.ptr); a.qsb = q.stride_b; a.qsh = q.stride_h; a.qss = q.stride_s;
        a.k16 = static_cast<const unsigned short*>(k.ptr); a.ksb = k.stride_b; a.ksh = k.stride_h; a.kss = k.stride_s;
        a.qk_scale = (float)((1.0 / sqrt((double)D)) * 1.44269504 * 8.0);
    }
    return launch_attn8(a, B * H, D, 0, s, pv ? (dtype == RSA_FP16 ? 2 : 1) : 0);
}

extern "C" int rsa_dense_fwd_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k,
                                 rsa_tensor4 v, int q_split, int kv_split, void* workspace, size_t workspace_bytes,
                                 rsa_out4 out, void* stream) {
    return dense_fwd_fp8(B, H, Sq, Sk, D, dtype, q, k, v, q_split, kv_split, 0, workspace, workspace_bytes, out, stream);
}

extern "C" int rsa_dense_fwd_fp8pv(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                                   int q_split, int kv_split, int causal, void* workspace, size_t workspace_bytes, rsa_out4 out,
                                   void* stream) {
    return dense_fwd_fp8(B, H, Sq, Sk, D, dtype, q, k, v, q_split, kv_split, causal ? 1 : 0, workspace, workspace_bytes, out, stream, 1);
}

extern "C" int rsa_dense_causal_fwd_fp8(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k,
                                        rsa_tensor4 v, int q_split, int kv_split, void* workspace, size_t workspace_bytes,
                                        rsa_out4 out, void* stream) {
    return dense_fwd_fp8(B, H, Sq, Sk, D, dtype, q, k, v, q_split, kv_split, 1, workspace, workspace_bytes, out, stream);
}
