// K5 host side: argument checks and launches of the block-sparse attention kernel (rsa_attn_kernel.hip).
// rsa_block_sparse_fwd, rsa_dense_fwd and rsa_rectified_attention of include/rsa.h live here.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include "rsa_attn.h"

// =====================================================================================================
// host side
// =====================================================================================================
static int g_k5_tsplit = 1;     // 1 = split-KV for the text query blocks when the partial buffer is given
extern int g_rsa_k3_prefix;
extern int g_rsa_k3_long;
extern int g_rsa_k4_split;
extern int g_rsa_k2_dma;
static int g_k5_tail_split = 1; // 64-row kernel: the last, partial generation's walks split over its idle slots (k5w_map)
static int g_k5_text_last = 1;  // 64-row kernel: split text-row pieces at the end of the grid (rsa_attn_kernel64.hip::k5w_map)
static int g_shard_invariant = 0; // rsa_set_shard_invariant: nothing about a row's arithmetic may depend on the size of the launch
static int g_k5_gsync_ratio = 2; // aligned starts: walks that keep 1 / ratio of the keys or more are not held back
int rsa_gsync_ratio() { return g_k5_gsync_ratio; }
static int g_k5_rows256 = 1;    // dense calls at head dim 128: 256-row tiles (0 = 128-row tiles, the sparse calls' form)
static int g_k5_static = 1;     // 64-row kernel, bf16: optimistic static softmax reference in the steady-state loop (0 = online body only)
int rsa_k5_static() { return g_k5_static; }
static int g_k5_gsync = 1;      // aligned starts of the sparse walks (rsa_attn.h): bit 0 = in the 64-row kernel, bit 1 = in the 32-row and e4m3 kernels
#ifdef RSA_K5_FORMS
extern int g_rsa_k5_form;
extern int g_rsa_k5w_form;
#endif
#ifdef RSA_K5_DIAG
static unsigned long long g_dbg_ptr = 0;   // diagnostics build: device buffer for K5's in-kernel time sums
#endif

void rsa_set_fp8_variant(int v);
void rsa_set_fp8_smooth_k(int v);
int rsa_launch_bsfwd(const AttnArgs& a, dim3 grid, size_t lds_bytes, int D, int dtype, hipStream_t s);
int rsa_launch_bsfwd64(const AttnArgs& a, dim3 grid, size_t lds_bytes, int D, int dtype, hipStream_t s);
static int g_k5_w64 = 3;        // the 64-rows-per-wave kernel (rsa_attn_kernel64.hip): bit 0 = at head dim 128, bit 1 = at head dim 64 (round 6); a clear bit = the 32-row kernel (A/B)

// Tuning / diagnostics hook (not part of the data path).  The switches are process-global, so the hook only works in a
// process that opted in with the environment variable RSA_TUNING=1 (the A/B tools and the variant tests); a production
// host cannot have its kernels changed under it by another library user.
extern "C" int rsa_set_tuning(const char* key, int value) {
    static const bool enabled = [] { const char* e = getenv("RSA_TUNING"); return e && e[0] == '1'; }();
    if (!key) return RSA_ERR_BAD_ARG;
    if (!enabled) return RSA_ERR_UNSUPPORTED;
    if (strcmp(key, "k3_prefix") == 0) { g_rsa_k3_prefix = value; return RSA_OK; }
    if (strcmp(key, "k3_long") == 0) { g_rsa_k3_long = value; return RSA_OK; }
    if (strcmp(key, "k4_split") == 0) { g_rsa_k4_split = value; return RSA_OK; }
    if (strcmp(key, "k2_dma") == 0) { g_rsa_k2_dma = value; return RSA_OK; }
#ifdef RSA_K5_FORMS
    if (strcmp(key, "k5_form") == 0) { g_rsa_k5_form = value; return RSA_OK; }
    if (strcmp(key, "k5w_form") == 0) { g_rsa_k5w_form = value; return RSA_OK; }
#endif
#ifdef RSA_K5_DIAG
    if (strcmp(key, "dbg_lo") == 0) { g_dbg_ptr = (g_dbg_ptr & 0xFFFFFFFF00000000ull) | (unsigned)value; return RSA_OK; }
    if (strcmp(key, "dbg_hi") == 0) { g_dbg_ptr = (g_dbg_ptr & 0xFFFFFFFFull) | ((unsigned long long)(unsigned)value << 32); return RSA_OK; }
#endif
    if (strcmp(key, "k5_static") == 0) { g_k5_static = value; return RSA_OK; }
    if (strcmp(key, "k5_rows256") == 0) { g_k5_rows256 = value; return RSA_OK; }
    if (strcmp(key, "k5_tsplit") == 0) { g_k5_tsplit = value; return RSA_OK; }
    if (strcmp(key, "k5_w64") == 0) { g_k5_w64 = value; return RSA_OK; }
    if (strcmp(key, "k5_gsync") == 0) { g_k5_gsync = value; return RSA_OK; }
    if (strcmp(key, "k5_gsync_ratio") == 0) { g_k5_gsync_ratio = value; return RSA_OK; }
    if (strcmp(key, "k5_text_last") == 0) { g_k5_text_last = value; return RSA_OK; }
    if (strcmp(key, "k5_tail_split") == 0) { g_k5_tail_split = value; return RSA_OK; }
    if (strcmp(key, "fp8_variant") == 0) { rsa_set_fp8_variant(value); return RSA_OK; }
    if (strcmp(key, "fp8_smooth_k") == 0) { rsa_set_fp8_smooth_k(value); return RSA_OK; }
    return RSA_ERR_BAD_ARG;
}

// Merge of the split-KV partials of the text query blocks (K5 wrote, per part, unnormalised O, the running maximum m in
// the log2 domain and the row sum l): one wave per query row, lanes over the head dim.  Rows beyond the valid text rows
// are written as zeros (as the 1-workgroup form does).
template <typename Tag>
__global__ __launch_bounds__(256) void text_combine_kernel(const float* __restrict__ tpart, unsigned short* out,
                                                           long osb, long osh, long oss, int D, int H, int NBv, int ntq,
                                                           int tsplit, int q_text_end, int Sq, long rows_total) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_total) return;
    const long bhq = row / RSA_BLOCK;
    const int r = (int)(row % RSA_BLOCK);
    const long bh = bhq / ntq;
    const int tq = (int)(bhq % ntq);
    const int grow = (NBv + tq) * RSA_BLOCK + r;
    if (grow >= Sq) return;
    const float* base = tpart + (bhq * tsplit * RSA_BLOCK + r) * (long)(D + 2);
    const long pstride = (long)RSA_BLOCK * (D + 2);
    // (round 6: every load of the row issued up front -- (m, l) of all pieces, then the pieces' values eight at a time -- instead of
    // one dependent round trip per piece; the sums run in the same order, piece 0 first: same bytes)
    constexpr int MAXP = RSA_TEXT_SPLIT;
    float mv[MAXP], lv[MAXP];
#pragma unroll
    for (int s = 0; s < MAXP; ++s) {
        const int sc = s < tsplit ? s : 0;
        const float2 ml = *reinterpret_cast<const float2*>(base + sc * pstride + D);
        mv[s] = s < tsplit ? ml.x : -INFINITY;
        lv[s] = ml.y;
    }
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < MAXP; ++s) M = fmaxf(M, mv[s]);
    float L = 0.0f;
    float acc[2] = {0.0f, 0.0f};
#pragma unroll
    for (int s0 = 0; s0 < MAXP; s0 += 8) {
        if (s0 >= tsplit) break;
        float x[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sc = s0 + u < tsplit ? s0 + u : 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + 64 * e;
                x[u][e] = d < D ? base[sc * pstride + d] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s0 + u < tsplit) {
                const float wgt = (mv[s0 + u] == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(mv[s0 + u] - M);
                L += lv[s0 + u] * wgt;
                acc[0] += x[u][0] * wgt;
                acc[1] += x[u][1] * wgt;
            }
        }
    }
    const float inv = (grow < q_text_end && L > 0.0f) ? 1.0f / L : 0.0f;
    unsigned short* op = out + (bh / H) * osb + (bh % H) * osh + (long)grow * oss;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = lane + 64 * e;
        if (d < D) op[d] = Elem<Tag>::from_f32(acc[e] * inv);
    }
}

// (also used by the fp8 kernel's host side, rsa_attn_fp8_kernel.hip)
int rsa_launch_text_combine(const float* tpart, unsigned short* out, long osb, long osh, long oss, int D, int H, int NBv,
                            int ntq, int tsplit, int q_text_end, int Sq, int BH, int dtype, hipStream_t s) {
    const long rows = (long)BH * ntq * RSA_BLOCK;
    if (rows <= 0) return RSA_OK;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == RSA_BF16)
        text_combine_kernel<bf16_tag><<<grid, 256, 0, s>>>(tpart, out, osb, osh, oss, D, H, NBv, ntq, tsplit, q_text_end,
                                                         Sq, rows);
    else
        text_combine_kernel<fp16_tag><<<grid, 256, 0, s>>>(tpart, out, osb, osh, oss, D, H, NBv, ntq, tsplit, q_text_end,
                                                         Sq, rows);
    return rsa_launch_status();
}
int rsa_text_split_enabled() { return g_k5_tsplit; }
int rsa_shard_invariant() { return g_shard_invariant; }
extern "C" int rsa_set_shard_invariant(int on) {
    const int prev = g_shard_invariant;
    g_shard_invariant = on != 0;
    return prev;
}
// How many pieces the partial buffer has room for per text block (capacity in bytes declared by the caller, rsa_buffers.tpart_bytes)
int rsa_text_split_capacity(size_t tpart_bytes, int BH, int ntq, int D) {
    const size_t per = (size_t)BH * (size_t)(ntq > 0 ? ntq : 1) * RSA_BLOCK * (size_t)(D + 2) * sizeof(float);
    const size_t n = tpart_bytes / per;
    return n > (size_t)RSA_TEXT_SPLIT ? RSA_TEXT_SPLIT : (int)n;
}
// Room for the tail pieces behind the text region (which always starts RSA_TEXT_SPLIT pieces per text block in)
bool rsa_tail_fits(size_t tpart_bytes, int BH, int ntq, int D, int tail_n, int tail_p) {
    const size_t blk = (size_t)RSA_BLOCK * (size_t)(D + 2) * sizeof(float);
    return ((size_t)BH * (size_t)ntq * RSA_TEXT_SPLIT + (size_t)tail_n * (size_t)tail_p) * blk <= tpart_bytes;
}
int rsa_text_last_enabled() { return g_k5_text_last; }

// Tail split of the 64-row kernel (rsa_attn_kernel64.hip::k5w_map): merge the tail_p partials of every tail block, then what the
// kernel's own epilogue does -- normalise, rectify (O . R / l + comp as one fma rounded to fp32), convert, store.  One wave per
// query row, lane = d and d + 64.
template <typename Tag>
__global__ __launch_bounds__(256) void tail_combine_kernel(const float* __restrict__ part, unsigned short* out, long osb, long osh,
                                                           long oss, int H, int NBv, int NBp, int tail_first, int tail_n, int tail_p,
                                                           const float* __restrict__ R, const float* __restrict__ comp, int Sq) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)tail_n * RSA_BLOCK) return;
    const int t = (int)(row / RSA_BLOCK), r = (int)(row % RSA_BLOCK);
    const int v = tail_first + t, bh = v / NBp, j = v % NBp;
    const int qblk = (j & 7) * (NBp >> 3) + (j >> 3);
    const int grow = qblk * RSA_BLOCK + r;
    if (qblk >= NBv || grow >= Sq) return;
    const float* base = part + ((long)t * tail_p * RSA_BLOCK + r) * (long)(D + 2);
    const long pstride = (long)RSA_BLOCK * (D + 2);
    float M = -INFINITY;
    for (int p = 0; p < tail_p; ++p) M = fmaxf(M, base[p * pstride + D]);
    float L = 0.0f, acc[2] = {0.0f, 0.0f};
    for (int p = 0; p < tail_p; ++p) {
        const float m = base[p * pstride + D], l = base[p * pstride + D + 1];
        const float wgt = (m == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(m - M);
        L += l * wgt;
        acc[0] += base[p * pstride + lane] * wgt;
        acc[1] += base[p * pstride + lane + 64] * wgt;
    }
    const long rowi = (long)bh * NBv + qblk;
    const float sc = (L > 0.0f ? 1.0f / L : 0.0f) * (R ? R[rowi] : 1.0f);
    unsigned short* op = out + (long)(bh / H) * osb + (long)(bh % H) * osh + (long)grow * oss;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int d = lane + 64 * e;
        float x = __builtin_fmaf(acc[e], sc, comp ? comp[rowi * D + d] : 0.0f);
        asm volatile("" : "+v"(x));     // (fma, THEN the conversion: as the kernel's epilogue)
        op[d] = Elem<Tag>::from_f32(x);
    }
}

// (also used by the e4m3 kernel's host side, rsa_attn_fp8_kernel.hip)
int rsa_launch_tail_combine(const float* part, unsigned short* out, long osb, long osh, long oss, int H, int NBv, int NBp,
                            int tail_first, int tail_n, int tail_p, const float* R, const float* comp, int Sq, int dtype,
                            hipStream_t s) {
    const long rows = (long)tail_n * RSA_BLOCK;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == RSA_BF16)
        tail_combine_kernel<bf16_tag><<<grid, 256, 0, s>>>(part, out, osb, osh, oss, H, NBv, NBp, tail_first, tail_n, tail_p, R, comp, Sq);
    else
        tail_combine_kernel<fp16_tag><<<grid, 256, 0, s>>>(part, out, osb, osh, oss, H, NBv, NBp, tail_first, tail_n, tail_p, R, comp, Sq);
    return rsa_launch_status();
}
// the plan of a tail split (see launch_attn): n_sparse = BH x NBp sparse workgroups, n_heavy_pad text pieces behind them
int rsa_plan_tail_split(long n_sparse, long n_heavy_pad, int* tail_first, int* tail_n, int* tail_p) {
    *tail_first = *tail_n = *tail_p = 0;
    if (!g_k5_tail_split || g_shard_invariant) return 0;
    const long full = n_sparse / 512, T = n_sparse % 512;
    const long room = 512 - n_heavy_pad;
    const long P = T > 0 ? (room / T < 4 ? room / T : 4) : 0;
    if (full < 1 || T <= 0 || P < 2) return 0;
    *tail_first = (int)(full * 512); *tail_n = (int)T; *tail_p = (int)P;       // T x P <= 512 = RSA_TAIL_PIECES
    return 1;
}
static int launch_tail_combine(const AttnArgs& a, int dtype, hipStream_t s) {
    return rsa_launch_tail_combine(a.tail_part, a.out, a.osb, a.osh, a.oss, a.H, a.NBv, a.NBp, a.tail_first, a.tail_n, a.tail_p, a.R,
                                   a.comp, a.Sq, dtype, s);
}

static int launch_text_combine(const AttnArgs& a, int BH, int D, int dtype, hipStream_t s) {
    return rsa_launch_text_combine(a.tpart, a.out, a.osb, a.osh, a.oss, D, a.H, a.NBv, a.NQB - a.NBv, a.tsplit,
                                   a.q_text_end, a.Sq, BH, dtype, s);
}

// Aligned starts (rsa_attn.h): the counters are a ring of slots in a __device__ array of the code object (no allocation, nothing
// to free), one slot per launch, cleared in stream order in front of it; a slot shared by two launches in flight (more than
// RING of them, on different streams) costs alignment, never correctness.  Grids of more than two generations only.
__device__ unsigned g_rsa_gsync[RSA_GSYNC_RING][RSA_GSYNC_SLOT_WORDS];
unsigned* rsa_gsync_slot(int which, unsigned grid, int wg_per_cu, hipStream_t s, int* gen) {
    static std::atomic<unsigned*> base[64];
    static std::atomic<unsigned> ticket{0};
    *gen = 32 * (wg_per_cu > 0 ? wg_per_cu : 2);     // 32 CUs per XCD
    if (!(g_k5_gsync & which) || wg_per_cu <= 0 || grid <= 16u * (unsigned)*gen) return nullptr;   // two generations or fewer: nothing to align
    const unsigned gens = ((grid + 7) / 8 + *gen - 1) / *gen;
    int dev = -1;
    if (gens > RSA_GSYNC_MAXG || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned* b = base[dev].load();
    if (!b) {
        // the generation arithmetic is the full chip's: 8 XCDs x 32 CUs, workgroup b on XCD b & 7 (a partitioned device -- CPX,
        // fewer CUs -- holds fewer workgroups than a generation expects: every first wait would run into its bound)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus != 256) {
            base[dev].store(reinterpret_cast<unsigned*>(1));
            return nullptr;
        }
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&b), HIP_SYMBOL(g_rsa_gsync)) != hipSuccess) return nullptr;
        base[dev].store(b);
    }
    if (b == reinterpret_cast<unsigned*>(1)) return nullptr;     // (not a full MI355X: see above)
    unsigned* slot = b + (size_t)(ticket.fetch_add(1) % RSA_GSYNC_RING) * RSA_GSYNC_SLOT_WORDS;
    return hipMemsetAsync(slot, 0, (8 + 8 * (size_t)gens) * sizeof(unsigned), s) == hipSuccess ? slot : nullptr;
}
int rsa_wg_per_cu(const void* kernel, int block, size_t lds_bytes) {
    struct Entry { const void* k; size_t lds; int dev, n; };
    static Entry cache[64];
    static std::atomic<int> used{0};
    static std::mutex mu;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int nu = used.load(std::memory_order_acquire);
    for (int i = 0; i < nu; ++i)
        if (cache[i].k == kernel && cache[i].lds == lds_bytes && cache[i].dev == dev) return cache[i].n;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, lds_bytes) != hipSuccess) n = 0;
    std::lock_guard<std::mutex> lk(mu);
    const int at = used.load();
    if (at < 64) { cache[at] = Entry{kernel, lds_bytes, dev, n}; used.store(at + 1, std::memory_order_release); }
    return n;
}

static int launch_attn(AttnArgs& a, int BH, int D, int dtype, size_t tpart_bytes, hipStream_t s) {
    const int ntq = a.NQB - a.NBv;
    if (a.tpart && tpart_bytes == 0) return RSA_ERR_WORKSPACE;   // capacity not declared (rsa_buffers.tpart_bytes, 0.5.0)
    a.gsync = nullptr; a.gsync_gen = 64; a.gsync_ratio = 2; a.k5_static = g_k5_static;
    // split-KV for the dense text rows: without it one workgroup walks every key block of a text query block (902 at the
    // HunyuanVideo shape = 10 kept lists) -- hidden among 21 600 sparse blocks on one GPU, the critical path when the
    // heads are sharded over 8
    const int n_txt_items = (a.kv_text_valid + RSA_BLOCK - 1) / RSA_BLOCK;
    a.tsplit = 1; a.tper = n_txt_items;
    if (a.mode == MODE_SPARSE && ntq > 0 && a.tpart && g_k5_tsplit && n_txt_items >= 32) {
        int sp = n_txt_items / 16;
        // 16 pieces per text block; 32 (RSA_TEXT_SPLIT, what tpart is sized for) on grids of fewer than 8 generations, where the
        // pieces of 0.6 of a sparse walk's life would be the last to finish behind a split tail (and the combine pass that
        // doubles with them is still small)
        int cap = ((long)BH * ((a.NBv + 7) & ~7) < 8 * 512 && !g_shard_invariant) ? RSA_TEXT_SPLIT : 16;
        const int room = rsa_text_split_capacity(tpart_bytes, BH, ntq, D);   // what the caller's buffer holds
        if (cap > room) cap = room;
        a.tsplit = sp > cap ? cap : sp;
        if (a.tsplit < 2) a.tsplit = 1;
        a.tper = (n_txt_items + a.tsplit - 1) / a.tsplit;
    }
    const int n_heavy = ntq > 0 ? BH * ntq * a.tsplit : 0;
    a.heavy_last = a.tsplit > 1 && g_k5_text_last;
    a.BH = BH;
    a.n_heavy_pad = (n_heavy + 7) & ~7;
    a.NBp = (a.NBv + 7) & ~7;
    long nblocks = (long)a.n_heavy_pad + (long)BH * a.NBp;
    // Tail split (64-row kernel, sparse lists): 512 workgroups run at a time (2 per CU), each for about as long as the others, so
    // a launch costs ceil(workgroups / 512) lives; when the last generation of sparse blocks is less than half full, its blocks'
    // walks are split over the idle slots (tail_p pieces each, partials behind the text region of tpart) and merged by a combine
    // pass.  Which blocks are split depends on the grid: a sharded and an unsharded run then agree on those blocks within
    // rounding, not byte for byte (tuning key k5_tail_split = 0 keeps every walk whole).
    a.tail_first = a.tail_n = a.tail_p = 0; a.tail_part = nullptr;
    const bool w64 = D == 128 && (g_k5_w64 & 1);
    if (w64 && a.mode == MODE_SPARSE && a.tpart && (a.heavy_last || n_heavy == 0)) {
        // the pieces AND the text-row pieces behind them must fit the 512 slots together: otherwise whatever starts late (0.6 of a
        // life for a text piece) ends the launch as late as the unsplit tail did (measured: 3 heads of the headline shape, 456
        // pieces + 96 text pieces: 1.92 ms against 1.88 unsplit)
        if (rsa_plan_tail_split((long)BH * a.NBp, a.n_heavy_pad, &a.tail_first, &a.tail_n, &a.tail_p) &&
            !rsa_tail_fits(tpart_bytes, BH, ntq, D, a.tail_n, a.tail_p))
            a.tail_first = a.tail_n = a.tail_p = 0;
        if (a.tail_n > 0) {
            a.tail_part = a.tpart + (long)BH * ntq * RSA_TEXT_SPLIT * RSA_BLOCK * (D + 2);
            nblocks = (long)a.tail_first + (long)a.tail_n * a.tail_p + a.n_heavy_pad;
        }
    }
    if (nblocks <= 0) return RSA_OK;
    if (nblocks > 0x7FFFFFFF) return RSA_ERR_UNSUPPORTED;
    if (a.NB_total > 8192) return RSA_ERR_UNSUPPORTED;  // kept list lives in LDS as u16, 16 KiB max
    const size_t lds_bytes = (size_t)4 * 64 * D * 2 + (((size_t)a.NB_total * 2 + 15) & ~(size_t)15);
    const bool use64 = (D == 128 && (g_k5_w64 & 1)) || (D == 64 && (g_k5_w64 & 2));
    const int st = use64 ? rsa_launch_bsfwd64(a, dim3((unsigned)nblocks), lds_bytes, D, dtype, s)
                         : rsa_launch_bsfwd(a, dim3((unsigned)nblocks), lds_bytes, D, dtype, s);
    if (st != RSA_OK) return st;
    if (a.tail_n > 0) {
        const int st2 = launch_tail_combine(a, dtype, s);
        if (st2 != RSA_OK) return st2;
    }
    if (a.tsplit <= 1) return st;
    return launch_text_combine(a, BH, D, dtype, s);
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
    a.tpart = buf->tpart;
    a.mode = MODE_SPARSE; a.H = l->H; a.Sq = l->S; a.Sk = l->S;
    a.NBv = l->NBv; a.NQB = l->NB_total; a.NB_total = l->NB_total;
    a.kv_valid = l->kv_valid; a.kv_text_valid = l->kv_text_valid;
    a.q_text_end = l->NBv * RSA_BLOCK + l->q_text_valid;
    a.q_split = 0; a.kv_split = 0; a.causal = 0; a.rows256 = 0;
    a.qk_scale = (float)((1.0 / sqrt((double)l->D)) * 1.44269504);  // sm_scale * 1.44269504 (hunyuan :145)
#ifdef RSA_K5_DIAG
    a.dbg = reinterpret_cast<unsigned long long*>(g_dbg_ptr);
#endif
    return launch_attn(a, l->B * l->H, l->D, l->dtype, buf->tpart_bytes, static_cast<hipStream_t>(stream));
}

static int dense_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k, rsa_tensor4 v,
                     int q_split, int kv_split, int causal, rsa_out4 out, void* stream) {
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
    a.tpart = nullptr;
    a.tsplit = 1; a.tper = 0;
    a.mode = MODE_DENSE; a.H = H; a.Sq = Sq; a.Sk = Sk;
    // head dim 128 through the 64-row kernel: 256-row tiles once there are at least two of them (a shorter call keeps 128-row tiles)
    a.rows256 = (((D == 128 && (g_k5_w64 & 1)) || (D == 64 && (g_k5_w64 & 2))) && g_k5_rows256 && Sq > 256) ? 1 : 0;
#ifdef RSA_K5_FORMS
    if (g_rsa_k5w_form != 0) a.rows256 = 0;     // (the A/B forms are forms of the 128-row kernel)
#endif
    const int rw = a.rows256 ? 2 * RSA_BLOCK : RSA_BLOCK;
    a.NQB = (Sq + rw - 1) / rw; a.NBv = a.NQB; a.NB_total = (Sk + RSA_BLOCK - 1) / RSA_BLOCK;
    a.kv_valid = Sk; a.kv_text_valid = Sk; a.q_text_end = 0;
    a.q_split = q_split; a.kv_split = kv_split; a.causal = causal;
    a.qk_scale = (float)((1.0 / sqrt((double)D)) * 1.44269504);
#ifdef RSA_K5_DIAG
    a.dbg = nullptr;
#endif
    return launch_attn(a, B * H, D, dtype, 0, static_cast<hipStream_t>(stream));
}

extern "C" int rsa_dense_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k,
                             rsa_tensor4 v, int q_split, int kv_split, rsa_out4 out, void* stream) {
    return dense_fwd(B, H, Sq, Sk, D, dtype, q, k, v, q_split, kv_split, 0, out, stream);
}

extern "C" int rsa_dense_causal_fwd(int B, int H, int Sq, int Sk, int D, int dtype, rsa_tensor4 q, rsa_tensor4 k,
                                    rsa_tensor4 v, int q_split, int kv_split, rsa_out4 out, void* stream) {
    return dense_fwd(B, H, Sq, Sk, D, dtype, q, k, v, q_split, kv_split, 1, out, stream);
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
