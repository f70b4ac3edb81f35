/*
 * vtm_oracle.c -- CPU restatement of VidToMe's cross-frame token-merging hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vidtome_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and there only as the checker / the timed CPU baseline.
 *
 * Each function cites the reference lines (under /root/reference) it restates.
 * Parity pin: oracle/oracle.py + this file are checked against golden vectors that
 * tests/golden/make_golden.py produced by importing the reference's own
 * vidtome/merge.py, vidtome/patch.py and utils/pnp_utils.py in the build container
 * (the reference has no tests or fixtures of its own -- SURVEY.md section 4).
 *
 * Canonical arithmetic (the contract the HIP kernels reproduce BITWISE):
 *   n_i   = sqrtf( acc_C ),  acc_0 = +0, acc_{k+1} = fmaf(x_ik, x_ik, acc_k)
 *   xh_ik = x_ik / n_i                         (IEEE-754 binary32 divide, no eps)
 *   s_ij  = acc_C,  acc_0 = +0, acc_{k+1} = fmaf(a_ik, b_jk, acc_k)   (k ascending)
 *   row max: first index among equal values; NaN beats everything and the FIRST
 *            NaN is kept (torch CPU max semantics); -0 == +0.
 *   sort  : descending by value, NaN first, ties (incl. -0/+0) by ascending index.
 * The reference computes the same quantities with torch CPU kernels whose summation
 * order is unspecified (merge.py:84,87); index parity with the reference is therefore
 * pinned on inputs whose top-1 / adjacent-rank gaps exceed that rounding noise.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off, AVX2+FMA baseline).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VTMO_API __attribute__((visibility("default")))

VTMO_API int vtmo_version(void) { return 1; }

VTMO_API int vtmo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

VTMO_API void vtmo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------
 * vtmo_normalize_gather  --  merge.py:84 (`metric / metric.norm(dim=-1, keepdim=True)`)
 * fused with `split` (merge.py:76-81 / 383-388): out[b,i,:] = xhat[b, rows[b,i], :].
 * x is (B, P, C) fp32 row-major, rows is (B, n) int32, out is (B, n, C).
 * Zero rows give 0/0 = NaN exactly like the reference (no eps).
 * ---------------------------------------------------------------------------------- */
VTMO_API int vtmo_normalize_gather(const float *x, int64_t B, int64_t P, int64_t C,
                                   const int32_t *rows, int64_t n, float *out) {
    if (!x || !rows || !out || B < 0 || P < 0 || C <= 0 || n < 0) return -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t i = 0; i < n; ++i) {
            int64_t r = rows[b * n + i];
            const float *src = x + (b * P + r) * C;
            float *dst = out + (b * n + i) * C;
            float acc = 0.0f;
            for (int64_t k = 0; k < C; ++k) acc = fmaf(src[k], src[k], acc);
            float nrm = sqrtf(acc);
            for (int64_t k = 0; k < C; ++k) dst[k] = src[k] / nrm;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Score micro-kernel.  Computes a TI x TJ tile of s_ij with a k-ascending fmaf chain per
 * (i,j).  Vectorisation runs over j (independent chains), so every lane performs exactly
 * the scalar chain: the result is bitwise independent of ISA width / thread count.
 * bt is the dst matrix transposed to (C, Ndp) so that j is the contiguous axis.
 * ---------------------------------------------------------------------------------- */
#define TI 4
#define TJ 64

static inline void score_tile(const float *a, int64_t lda, const float *bt, int64_t ldb,
                              int64_t C, float *restrict s /* TI*TJ */) {
    float acc[TI][TJ];
    for (int i = 0; i < TI; ++i)
        for (int j = 0; j < TJ; ++j) acc[i][j] = 0.0f;
    for (int64_t k = 0; k < C; ++k) {
        const float *brow = bt + k * ldb;
        for (int i = 0; i < TI; ++i) {
            const float av = a[i * lda + k];
#pragma omp simd
            for (int j = 0; j < TJ; ++j) acc[i][j] = fmaf(av, brow[j], acc[i][j]);
        }
    }
    for (int i = 0; i < TI; ++i)
        for (int j = 0; j < TJ; ++j) s[i * TJ + j] = acc[i][j];
}

/* torch CPU `max(dim)` update rule (merge.py:97,112 / 401,416): strictly-greater keeps the
 * first index; a NaN replaces any non-NaN and then sticks. */
static inline void max_update(float s, int64_t j, float *best, int64_t *besti) {
    if (s > *best || (s != s && *best == *best)) {
        *best = s;
        *besti = j;
    }
}

/* ------------------------------------------------------------------------------------
 * vtmo_match  --  merge.py:87 (`scores = a @ b.transpose(-1,-2)`) fused with
 * merge.py:109-113 (non-aligned: per-sample row max/argmax) or merge.py:93-97 (aligned:
 * scores of all samples concatenated on the dst axis, one max/argmax per src row; the
 * returned index lives in [0, B*Nd)).  The score matrix is never materialised.
 *   a: (B, Ns, C) normalised src rows;  b: (B, Nd, C) normalised dst rows.
 *   align == 0: node_max (B, Ns), node_idx (B, Ns).
 *   align != 0: node_max (Ns),    node_idx (Ns).
 * Row range [i_begin, i_end) restricts the src rows processed (bounded CPU-baseline
 * samples); pass 0, Ns for everything.
 * ---------------------------------------------------------------------------------- */
VTMO_API int vtmo_match_rows(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd,
                             int64_t C, int align, int64_t i_begin, int64_t i_end,
                             float *node_max, int32_t *node_idx) {
    if (!a || !b || !node_max || !node_idx || B <= 0 || Ns < 0 || Nd <= 0 || C <= 0) return -1;
    if (i_begin < 0 || i_end > Ns || i_begin > i_end) return -1;
    const int64_t Ndp = (Nd + TJ - 1) / TJ * TJ;
    float *bt = (float *)calloc((size_t)(B * C * Ndp), sizeof(float));
    if (!bt) return -2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t bi = 0; bi < B; ++bi)
        for (int64_t j = 0; j < Nd; ++j)
            for (int64_t k = 0; k < C; ++k) bt[(bi * C + k) * Ndp + j] = b[(bi * Nd + j) * C + k];

    const int64_t nblk = (i_end - i_begin + TI - 1) / TI;
    const int64_t outer = align ? 1 : B;
#pragma omp parallel
    {
        float s[TI * TJ];
        float apad[TI * 4096];
#pragma omp for collapse(2) schedule(dynamic, 8)
        for (int64_t bo = 0; bo < outer; ++bo) {
            for (int64_t blk = 0; blk < nblk; ++blk) {
                const int64_t i0 = i_begin + blk * TI;
                const int ni = (int)((i_end - i0) < TI ? (i_end - i0) : TI);
                float best[TI];
                int64_t besti[TI];
                for (int i = 0; i < TI; ++i) {
                    best[i] = -INFINITY;
                    besti[i] = 0;
                }
                /* In aligned mode the concatenated dst axis is [sample0 | sample1 | ...]
                 * (torch.cat([*scores], dim=-1), merge.py:96). */
                const int64_t bs = align ? 0 : bo, be = align ? B : bo + 1;
                for (int64_t bi = bs; bi < be; ++bi) {
                    const float *arow = a + (bi * Ns + i0) * C;
                    const float *ap = arow;
                    int64_t lda = C;
                    if (ni < TI) { /* ragged tail: pad with copies of the last valid row */
                        if (C > 4096) { /* fall back to row-at-a-time for huge C */
                            lda = 0;
                        } else {
                            for (int i = 0; i < TI; ++i)
                                memcpy(apad + i * C, arow + (i < ni ? i : ni - 1) * C,
                                       (size_t)C * sizeof(float));
                            ap = apad;
                        }
                    }
                    for (int64_t j0 = 0; j0 < Ndp; j0 += TJ) {
                        if (lda == 0) {
                            for (int i = 0; i < ni; ++i)
                                for (int j = 0; j < TJ; ++j) {
                                    float acc = 0.0f;
                                    for (int64_t k = 0; k < C; ++k)
                                        acc = fmaf(arow[i * C + k], bt[(bi * C + k) * Ndp + j0 + j], acc);
                                    s[i * TJ + j] = acc;
                                }
                        } else {
                            score_tile(ap, lda, bt + bi * C * Ndp + j0, Ndp, C, s);
                        }
                        const int nj = (int)((Nd - j0) < TJ ? (Nd - j0) : TJ);
                        for (int i = 0; i < ni; ++i)
                            for (int j = 0; j < nj; ++j) {
                                const int64_t col = (align ? bi * Nd : 0) + j0 + j;
                                max_update(s[i * TJ + j], col, &best[i], &besti[i]);
                            }
                    }
                }
                for (int i = 0; i < ni; ++i) {
                    node_max[bo * Ns + i0 + i] = best[i];
                    node_idx[bo * Ns + i0 + i] = (int32_t)besti[i];
                }
            }
        }
    }
    free(bt);
    return 0;
}

VTMO_API int vtmo_match(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd,
                        int64_t C, int align, float *node_max, int32_t *node_idx) {
    return vtmo_match_rows(a, b, B, Ns, Nd, C, align, 0, Ns, node_max, node_idx);
}

/* Straight scalar restatement (no tiling) used by the tests to pin the tiled kernel. */
VTMO_API int vtmo_match_scalar(const float *a, const float *b, int64_t B, int64_t Ns, int64_t Nd,
                               int64_t C, int align, float *node_max, int32_t *node_idx) {
    const int64_t outer = align ? 1 : B;
    for (int64_t bo = 0; bo < outer; ++bo)
        for (int64_t i = 0; i < Ns; ++i) {
            float best = 0.0f;
            int64_t besti = 0;
            int first = 1;
            const int64_t bs = align ? 0 : bo, be = align ? B : bo + 1;
            for (int64_t bi = bs; bi < be; ++bi)
                for (int64_t j = 0; j < Nd; ++j) {
                    float acc = 0.0f;
                    for (int64_t k = 0; k < C; ++k)
                        acc = fmaf(a[(bi * Ns + i) * C + k], b[(bi * Nd + j) * C + k], acc);
                    const int64_t col = (align ? bi * Nd : 0) + j;
                    if (first) {
                        best = acc;
                        besti = col;
                        first = 0;
                    } else {
                        max_update(acc, col, &best, &besti);
                    }
                }
            node_max[bo * Ns + i] = best;
            node_idx[bo * Ns + i] = (int32_t)besti;
        }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * vtmo_sort_desc  --  merge.py:98,113 / 402,417 (`node_max.argsort(dim=-1, descending=True)`).
 * torch's CPU argsort is not stable for n >= ~1000; the canonical order is
 * the stable one: descending value, NaN first, ties (and -0/+0) by ascending index.
 * keys (B, n) -> perm (B, n) int32.  LSD radix sort on an order-preserving key.
 * ---------------------------------------------------------------------------------- */
static inline uint32_t desc_key(float f) {
    uint32_t u;
    if (f != f) return 0u; /* NaN sorts first */
    f = f + 0.0f;          /* -0 -> +0 */
    memcpy(&u, &f, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u); /* ascending-orderable */
    return ~u; /* descending; ~orderable(+inf) = 0x007FFFFF > 0, so NaN (0) is strictly first */
}

VTMO_API int vtmo_sort_desc(const float *keys, int64_t B, int64_t n, int32_t *perm) {
    if (!keys || !perm || B < 0 || n < 0) return -1;
    int err = 0;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        uint32_t *k0 = (uint32_t *)malloc((size_t)n * 4), *k1 = (uint32_t *)malloc((size_t)n * 4);
        int32_t *p0 = (int32_t *)malloc((size_t)n * 4), *p1 = (int32_t *)malloc((size_t)n * 4);
        if (!k0 || !k1 || !p0 || !p1) {
            err = 1;
        } else {
            for (int64_t i = 0; i < n; ++i) {
                k0[i] = desc_key(keys[b * n + i]);
                p0[i] = (int32_t)i;
            }
            for (int pass = 0; pass < 4; ++pass) {
                int64_t cnt[257];
                memset(cnt, 0, sizeof cnt);
                const int sh = pass * 8;
                for (int64_t i = 0; i < n; ++i) cnt[((k0[i] >> sh) & 255u) + 1]++;
                for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
                for (int64_t i = 0; i < n; ++i) {
                    int64_t pos = cnt[(k0[i] >> sh) & 255u]++;
                    k1[pos] = k0[i];
                    p1[pos] = p0[i];
                }
                uint32_t *tk = k0; k0 = k1; k1 = tk;
                int32_t *tp = p0; p0 = p1; p1 = tp;
            }
            memcpy(perm + b * n, p0, (size_t)n * 4);
        }
        free(k0); free(k1); free(p0); free(p1);
    }
    return err ? -2 : 0;
}

/* ------------------------------------------------------------------------------------
 * vtmo_gather_rows  --  the reference's `merge` closure in replace mode
 * (merge.py:119-133 / 423-437) once the chain is composed into one row map:
 * out[b,p,:] = x[b, map[b,p], :].  Also the `unmerge` closure (merge.py:135-155 /
 * 439-460): every output row is written exactly once, so it is a gather with the inverse
 * map.  resid (optional, same shape as out) adds patch.py:169's residual.
 * ---------------------------------------------------------------------------------- */
VTMO_API int vtmo_gather_rows(const float *x, int64_t B, int64_t P, int64_t C, const int32_t *map,
                              int64_t M, const float *resid, float *out) {
    if (!x || !map || !out) return -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < M; ++p) {
            const float *src = x + (b * P + map[b * M + p]) * C;
            float *dst = out + (b * M + p) * C;
            if (resid) {
                const float *r = resid + (b * M + p) * C;
                for (int64_t k = 0; k < C; ++k) dst[k] = src[k] + r[k];
            } else {
                memcpy(dst, src, (size_t)C * sizeof(float));
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * vtmo_attention  --  utils/pnp_utils.py:47-95 (`sa_forward`), the only statement of the
 * self-attention arithmetic inside the reference (patch.py:157-162 calls Diffusers'
 * Attention, which is not vendored):  out = softmax(q k^T * scale) v  per head, no mask.
 * q,k,v,out are (B, M, h*d) fp32 with heads interleaved on the channel axis
 * (head_to_batch_dim / batch_to_head_dim are pure reshapes of that layout).
 * share_groups > 1 restates the injection branch (pnp_utils.py:57-67,86-90): the
 * probabilities come from the FIRST B/share_groups samples and are reused (`repeat`) for
 * every group; v stays per-sample.
 * Row range [m_begin, m_end) restricts the query rows (bounded CPU-baseline samples).
 * ---------------------------------------------------------------------------------- */
VTMO_API int vtmo_attention_rows(const float *q, const float *k, const float *v, int64_t B,
                                 int64_t h, int64_t M, int64_t d, float scale, int share_groups,
                                 int64_t m_begin, int64_t m_end, float *out) {
    if (!q || !k || !v || !out || B <= 0 || h <= 0 || M <= 0 || d <= 0) return -1;
    if (share_groups < 1) share_groups = 1;
    if (B % share_groups) return -1;
    if (m_begin < 0 || m_end > M || m_begin > m_end) return -1;
    const int64_t C = h * d;
    const int64_t sb = B / share_groups; /* source batch size */
    const int64_t QB = 16;
    const int64_t nqb = (m_end - m_begin + QB - 1) / QB;
    int err = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t bq = share_groups > 1 ? (b % sb) : b;
        for (int64_t hh = 0; hh < h; ++hh) {
            float *kt = (float *)malloc((size_t)(d * M) * sizeof(float));
            if (!kt) return -2;
#pragma omp parallel for schedule(static)
            for (int64_t j = 0; j < M; ++j)
                for (int64_t dd = 0; dd < d; ++dd) kt[dd * M + j] = k[(bq * M + j) * C + hh * d + dd];
#pragma omp parallel
            {
                float *s = (float *)malloc((size_t)M * sizeof(float));
                float *o = (float *)malloc((size_t)d * sizeof(float));
                if (!s || !o) err = 1;
#pragma omp for schedule(dynamic, 1)
                for (int64_t qb = 0; qb < nqb; ++qb) {
                    if (!s || !o) continue;
                    for (int64_t m = m_begin + qb * QB; m < m_end && m < m_begin + (qb + 1) * QB; ++m) {
                        const float *qr = q + (bq * M + m) * C + hh * d;
                        for (int64_t j = 0; j < M; ++j) s[j] = 0.0f;
                        for (int64_t dd = 0; dd < d; ++dd) {
                            const float qv = qr[dd];
                            const float *kr = kt + dd * M;
#pragma omp simd
                            for (int64_t j = 0; j < M; ++j) s[j] += qv * kr[j];
                        }
                        float mx = -INFINITY;
                        for (int64_t j = 0; j < M; ++j) {
                            s[j] *= scale;
                            mx = s[j] > mx ? s[j] : mx;
                        }
                        float sum = 0.0f;
                        for (int64_t j = 0; j < M; ++j) {
                            s[j] = expf(s[j] - mx);
                            sum += s[j];
                        }
                        for (int64_t dd = 0; dd < d; ++dd) o[dd] = 0.0f;
                        for (int64_t j = 0; j < M; ++j) {
                            const float p = s[j];
                            const float *vr = v + (b * M + j) * C + hh * d;
                            for (int64_t dd = 0; dd < d; ++dd) o[dd] += p * vr[dd];
                        }
                        float *orow = out + (b * M + m) * C + hh * d;
                        for (int64_t dd = 0; dd < d; ++dd) orow[dd] = o[dd] / sum;
                    }
                }
                free(s);
                free(o);
            }
            free(kt);
        }
    }
    return err ? -2 : 0;
}

VTMO_API int vtmo_attention(const float *q, const float *k, const float *v, int64_t B, int64_t h,
                            int64_t M, int64_t d, float scale, int share_groups, float *out) {
    return vtmo_attention_rows(q, k, v, B, h, M, d, scale, share_groups, 0, M, out);
}

/* attention with a separate query set: q (B, Mq, C), k and v (B, Mk, C) -> out (B, Mq, C).  Same arithmetic as
 * vtmo_attention_rows (pnp_utils.py:47-95 with q / k of different lengths, which is what the patched block's
 * cross-attention, patch.py:178-183, and its live-query self-attention evaluate); the softmax denominator and the
 * PV sums are accumulated in double so that a 1e-3 comparison over ~50 000 keys measures the kernel under test,
 * not the summation order of the checker.  Used with a SAMPLE of query rows at the full BASELINE sizes. */
VTMO_API int vtmo_attention_qkv(const float *q, const float *k, const float *v, int64_t B, int64_t h, int64_t Mq,
                                int64_t Mk, int64_t d, float scale, float *out) {
    if (!q || !k || !v || !out || B <= 0 || h <= 0 || Mq <= 0 || Mk <= 0 || d <= 0) return -1;
    const int64_t C = h * d;
    int err = 0;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t hh = 0; hh < h; ++hh) {
            float *kt = (float *)malloc((size_t)(d * Mk) * sizeof(float));
            if (!kt) return -2;
#pragma omp parallel for schedule(static)
            for (int64_t j = 0; j < Mk; ++j)
                for (int64_t dd = 0; dd < d; ++dd) kt[dd * Mk + j] = k[(b * Mk + j) * C + hh * d + dd];
#pragma omp parallel
            {
                float *s = (float *)malloc((size_t)Mk * sizeof(float));
                double *o = (double *)malloc((size_t)d * sizeof(double));
                if (!s || !o) err = 1;
#pragma omp for schedule(dynamic, 4)
                for (int64_t m = 0; m < Mq; ++m) {
                    if (!s || !o) continue;
                    const float *qr = q + (b * Mq + m) * C + hh * d;
                    for (int64_t j = 0; j < Mk; ++j) s[j] = 0.0f;
                    for (int64_t dd = 0; dd < d; ++dd) {
                        const float qv = qr[dd];
                        const float *kr = kt + dd * Mk;
#pragma omp simd
                        for (int64_t j = 0; j < Mk; ++j) s[j] += qv * kr[j];
                    }
                    float mx = -INFINITY;
                    for (int64_t j = 0; j < Mk; ++j) {
                        s[j] *= scale;
                        mx = s[j] > mx ? s[j] : mx;
                    }
                    double sum = 0.0;
                    for (int64_t dd = 0; dd < d; ++dd) o[dd] = 0.0;
                    for (int64_t j = 0; j < Mk; ++j) {
                        const double p = (double)expf(s[j] - mx);
                        sum += p;
                        const float *vr = v + (b * Mk + j) * C + hh * d;
                        for (int64_t dd = 0; dd < d; ++dd) o[dd] += p * (double)vr[dd];
                    }
                    float *orow = out + (b * Mq + m) * C + hh * d;
                    for (int64_t dd = 0; dd < d; ++dd) orow[dd] = (float)(o[dd] / sum);
                }
                free(s);
                free(o);
            }
            free(kt);
        }
    return err ? -2 : 0;
}
