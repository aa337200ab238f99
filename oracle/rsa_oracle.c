/*
 * rsa_oracle.c -- CPU restatement of the Rectified-SpaAttn mask-selection statistics.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rectified_spaattn_amd/, bench.py's timed
 * region) may link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker.
 *
 * What it restates (reference = /root/reference, citations are file:line):
 *   - block pooling of Q/K/V               rectified_hunyuan_attn.py:189-194, :356 ; rectified_wan21_attn.py:189-192, :337
 *   - GAPR pooling-error statistics         gapr_mask.py:19-42
 *   - pooled scores + softmax               rectified_hunyuan_attn.py:198-211 ; rectified_wan21_attn.py:196-213
 *   - IPAR re-allocation                    rectified_hunyuan_attn.py:218-223
 *   - sort / cumsum / count / top-k select  rectified_hunyuan_attn.py:226-262
 *   - neighbour / text / first-frame union  rectified_hunyuan_attn.py:265-277 ; rectified_wan21_attn.py:259-271
 *   - rectification factor R and weights    rectified_hunyuan_attn.py:348-357
 *
 * The reference runs these in the input dtype (bf16) with unspecified reduction orders and an unstable
 * sort, so "bit-exact block mask" needs a numeric contract.  This file IS the contract ("fp32 statistics
 * contract v1"); the HIP kernels implement exactly the same operation order, so mask bits are equal for
 * every input, not only for well-separated ones:
 *
 *   C1  inputs are bf16/fp16 values widened exactly to fp32; every statistic is fp32.
 *   C2  block sums: rows r = 16*i + g (g = 0..15 row group, i = 0..T/16-1): per group a sequential sum
 *       over i, then a pairwise tree over g with strides 1,2,4,8.  mean = sum * (1/T).
 *   C3  mean-abs-deviation: |x - mean| summed in the same order.
 *   C4  dot products: acc = 0; for d = 0..D-1: acc = fmaf(a[d], b[d], acc)   (one rounding per step; this
 *       is also bit-for-bit what v_mfma_f32_32x32x2_f32 chains compute).
 *   C5  exp: orc_exp() below, built only from fp32 mul/fma/rint/ldexp -> identical on CPU and GPU.
 *   C6  row sums (softmax denominator, IPAR sums, R): 256 strided partial sums (element j goes to
 *       partial j%256, sequential in j), then a pairwise tree with strides 1,2,...,128.
 *   C7  sort: descending probability, ties -> lower column index first (a total order).
 *   C8  cumulative sum: sequential fp32 in sorted order; count = #{c_k <= p}, n = max(count+1, top_k).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma; fmaf must be a real fused multiply-add).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NT 256 /* strided partial sums (C6) */

/* ---- C5: exp ----------------------------------------------------------------------------------------- */
float orc_exp(float x) {
    const float LOG2E = 1.44269504088896340736f;
    float y = x * LOG2E;
    if (!(y >= -126.0f)) return 0.0f; /* also NaN -> 0 */
    float n = rintf(y);
    float f = y - n; /* exact, |f| <= 0.5 */
    /* 2^f, Taylor in f*ln2, degree 7, Horner with fused multiply-adds */
    float p = 1.52527338040598402800e-5f;
    p = fmaf(p, f, 1.54035303933816099544e-4f);
    p = fmaf(p, f, 1.33335581464284434234e-3f);
    p = fmaf(p, f, 9.61812910762847716197e-3f);
    p = fmaf(p, f, 5.55041086648215799532e-2f);
    p = fmaf(p, f, 2.40226506959100712333e-1f);
    p = fmaf(p, f, 6.93147180559945309417e-1f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

/* ---- C6: strided partial sums + pairwise tree ------------------------------------------------------------ */
static float strided_tree_sum(const float* v, int n, const uint8_t* keep, int keep_value) {
    float part[ORC_NT];
    for (int t = 0; t < ORC_NT; ++t) part[t] = 0.0f;
    for (int j = 0; j < n; ++j) {
        float x = v[j];
        if (keep && ((keep[j] != 0) != (keep_value != 0))) x = 0.0f;
        part[j % ORC_NT] = part[j % ORC_NT] + x;
    }
    for (int s = 1; s < ORC_NT; s <<= 1)
        for (int a = 0; a < ORC_NT; a += 2 * s) part[a] = part[a] + part[a + s];
    return part[0];
}

float orc_row_sum(const float* v, int n) { return strided_tree_sum(v, n, NULL, 0); }

/* ---- C2/C3: block pooling ------------------------------------------------------------------------------- */
/* x: [S_valid, D] fp32 (rows >= S_valid are zero -- the reference zero-pads, rectified_wan21_attn.py:299-302)
 * mean, mad: [NB, D]; mad may be NULL.  T must be a multiple of 16. */
void orc_pool(const float* x, long S_valid, int NB, int T, int D, float* mean, float* mad) {
    const int G = 16, I = T / G;
    const float invT = 1.0f / (float)T;
    float* part = (float*)malloc(sizeof(float) * G);
    for (int b = 0; b < NB; ++b) {
        for (int d = 0; d < D; ++d) {
            for (int g = 0; g < G; ++g) {
                float acc = 0.0f;
                for (int i = 0; i < I; ++i) {
                    long r = (long)b * T + 16 * i + g;
                    float xv = (r < S_valid) ? x[r * D + d] : 0.0f;
                    acc = (i == 0) ? xv : acc + xv;
                }
                part[g] = acc;
            }
            for (int s = 1; s < G; s <<= 1)
                for (int a = 0; a < G; a += 2 * s) part[a] = part[a] + part[a + s];
            float m = part[0] * invT;
            mean[(long)b * D + d] = m;
            if (mad) {
                for (int g = 0; g < G; ++g) {
                    float acc = 0.0f;
                    for (int i = 0; i < I; ++i) {
                        long r = (long)b * T + 16 * i + g;
                        float xv = (r < S_valid) ? x[r * D + d] : 0.0f;
                        float dv = fabsf(xv - m);
                        acc = (i == 0) ? dv : acc + dv;
                    }
                    part[g] = acc;
                }
                for (int s = 1; s < G; s <<= 1)
                    for (int a = 0; a < G; a += 2 * s) part[a] = part[a] + part[a + s];
                mad[(long)b * D + d] = part[0] * invT;
            }
        }
    }
    free(part);
}

/* ---- C4: dot products ----------------------------------------------------------------------------------- */
/* out[i][j] = sum_d a[i][d]*b[j][d], k-ordered fmaf chain */
void orc_dots(const float* a, const float* b, int NA, int NBc, int D, float* out) {
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NBc; ++j) {
            float acc = 0.0f;
            const float* pa = a + (long)i * D;
            const float* pb = b + (long)j * D;
            for (int d = 0; d < D; ++d) acc = fmaf(pa[d], pb[d], acc);
            out[(long)i * NBc + j] = acc;
        }
}

/* ---- row selection (C5..C8) ----------------------------------------------------------------------------- */
typedef struct {
    float p;
    int idx;
} orc_kv;

static int cmp_desc(const void* a, const void* b) {
    const orc_kv* x = (const orc_kv*)a;
    const orc_kv* y = (const orc_kv*)b;
    if (x->p > y->p) return -1;
    if (x->p < y->p) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/*
 * One (b, h, q-block i) row.
 *  s_vis[NBv]   unscaled pooled scores qbar_i . kbar_j
 *  s_txt[n_txt] unscaled scores qbar_i . K_u for every valid text token (n_txt may be 0 -> Wan layout, no IPAR)
 *  eq[NBv], ek[NBv] raw GAPR dots  a^q_i . kbar_j  and  qbar_i . a^k_j
 *  nbr[NBv]     neighbour row (0/1) or NULL
 *  outputs: kept[NB_total] (0/1), unrel[NBv] (0/1), probs[L] (L = NBv + (n_txt>0)), w[L] compensation
 *           weights, *n_needed, *R
 */
void orc_select_row(const float* s_vis, const float* s_txt, const float* eq, const float* ek,
                    const uint8_t* nbr, int NBv, int n_txt, int NB_total, int text_end_block, int qblk,
                    int first_frame_blocks, int top_k, float thr, float scale, uint8_t* kept,
                    uint8_t* unrel, float* probs, float* w, int* n_needed, float* R) {
    const int has_txt = n_txt > 0;
    const int L = NBv + (has_txt ? 1 : 0);
    const int NS = NBv + n_txt;
    float* x = (float*)malloc(sizeof(float) * (NS > 0 ? NS : 1));
    /* scaled scores, max */
    float m = -INFINITY;
    for (int j = 0; j < NBv; ++j) {
        x[j] = s_vis[j] * scale;
        m = fmaxf(m, x[j]);
    }
    for (int u = 0; u < n_txt; ++u) {
        x[NBv + u] = s_txt[u] * scale;
        m = fmaxf(m, x[NBv + u]);
    }
    for (int j = 0; j < NS; ++j) x[j] = orc_exp(x[j] - m);
    float Z = strided_tree_sum(x, NS, NULL, 0);
    for (int j = 0; j < NS; ++j) x[j] = x[j] / Z;
    /* IPAR (rectified_hunyuan_attn.py:218-223) */
    if (has_txt) {
        float normal_sum = strided_tree_sum(x, NBv, NULL, 0);
        float text_sum = strided_tree_sum(x + NBv, n_txt, NULL, 0);
        float denom = normal_sum * 128.0f + text_sum;
        for (int j = 0; j < NBv; ++j) probs[j] = (x[j] * 128.0f) / denom;
        probs[NBv] = text_sum / denom;
    } else {
        for (int j = 0; j < NBv; ++j) probs[j] = x[j];
    }
    /* GAPR (gapr_mask.py:26-42); IQ*JK = 2^14 cancels exactly */
    for (int j = 0; j < NBv; ++j) unrel[j] = !(fabsf(s_vis[j]) > (fabsf(eq[j]) + fabsf(ek[j])));
    /* sort, cumsum, count */
    orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * L);
    for (int j = 0; j < L; ++j) {
        kv[j].p = probs[j];
        kv[j].idx = j;
    }
    qsort(kv, L, sizeof(orc_kv), cmp_desc);
    int count = 0;
    float c = 0.0f;
    for (int k = 0; k < L; ++k) {
        c = (k == 0) ? kv[0].p : c + kv[k].p;
        if (c <= thr) count++;
    }
    int n = count + 1;
    if (n < top_k) n = top_k;
    *n_needed = n;
    if (n > L) n = L;
    memset(kept, 0, NB_total);
    for (int k = 0; k < n; ++k) kept[kv[k].idx] = 1; /* column NBv (text) maps to block NBv */
    if (nbr)
        for (int j = 0; j < NBv; ++j) kept[j] |= (nbr[j] != 0);
    if (has_txt)
        for (int j = NBv; j < text_end_block && j < NB_total; ++j) kept[j] = 1;
    if (first_frame_blocks > 0 && qblk < first_frame_blocks)
        for (int j = 0; j < first_frame_blocks && j < NB_total; ++j) kept[j] = 1;
    /* rectification (rectified_hunyuan_attn.py:348-357) */
    uint8_t* M = (uint8_t*)malloc(L);
    for (int j = 0; j < NBv; ++j) M[j] = kept[j] | unrel[j];
    if (has_txt) M[NBv] = kept[NBv];
    *R = strided_tree_sum(probs, L, M, 1);
    for (int j = 0; j < L; ++j) w[j] = M[j] ? 0.0f : probs[j];
    free(M);
    free(kv);
    free(x);
}
