/*
 * cpu_scan.c -- CPU BASELINE, TEST / MEASUREMENT INFRASTRUCTURE ONLY.
 *
 * Only bench.py's `cpu_baseline` leg and tests/ may load this library; it is never linked into
 * libknn355.so and the product package never imports it.
 *
 * What it is: the CPU path the reference runs -- faiss-cpu 1.7.2 `IndexFlat.search`
 * (/root/reference/seqvec_search/main.py:45, cath/search.py:24, pfam/proteins_search.py:49) -- written the
 * way a maintainer would write it for a many-core x86 host, so that the number printed next to the GPU's
 * is one they would recognise: an OpenMP scan over contiguous row ranges, a register-blocked AVX-512 (or
 * AVX2+FMA) sgemm-style micro-kernel (8 rows x 32 queries, 16 vector accumulators, no horizontal sums), FAISS's L2 formula
 * |x|^2 + |y|^2 - 2<x,y> clamped at 0 on top of it, a per-thread threshold + candidate buffer per query
 * (FAISS: a heap per query below k = 100, a reservoir from k = 100 on), and one final merge, best first,
 * ties to the lower row id.  Rows live in memory first-touched by the thread that scans them (NUMA).
 *
 * What it is not: the bit-level oracle.  Its fp32 sums run in SIMD order, not in the knn355 chain order
 * of knn_oracle.c, so ids may differ from the oracle's inside fp32 noise clusters
 * (tests/test_cpu_scan.py applies the tie-tolerant comparator).
 */
#define _GNU_SOURCE
#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float v;      /* smaller = better (IP: -score, L2: distance) */
    int64_t id;
} cand_t;

static int cand_less(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* ---- memory first-touched by the scanning threads ------------------------------------------- */
static void row_range(int64_t n, int nth, int t, int64_t *lo, int64_t *hi) {
    /* contiguous ranges in multiples of 4 rows (the micro-kernel's row block) */
    int64_t blocks = (n + 3) / 4;
    int64_t b0 = blocks * t / nth, b1 = blocks * (t + 1) / nth;
    *lo = b0 * 4 < n ? b0 * 4 : n;
    *hi = b1 * 4 < n ? b1 * 4 : n;
}

/* n x d floats, 64-byte aligned, every page first written by the thread that cpu_scan_search will
 * scan it with (same thread count!).  Free with cpu_scan_free. */
float *cpu_scan_alloc(int64_t n, int32_t d, int32_t threads) {
    float *p = NULL;
    size_t bytes = (size_t)n * d * sizeof(float);
    if (posix_memalign((void **)&p, 2u << 20, bytes ? bytes : 64)) return NULL;
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        int64_t lo, hi;
        row_range(n, omp_get_num_threads(), omp_get_thread_num(), &lo, &hi);
        if (hi > lo) memset(p + lo * d, 0, (size_t)(hi - lo) * d * sizeof(float));
    }
    return p;
}
void cpu_scan_free(void *p) { free(p); }

/* parallel copy into such a buffer (keeps the pages where they are) */
void cpu_scan_copy(float *dst, const float *src, int64_t n, int32_t d, int32_t threads) {
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
    {
        int64_t lo, hi;
        row_range(n, omp_get_num_threads(), omp_get_thread_num(), &lo, &hi);
        if (hi > lo) memcpy(dst + lo * d, src + lo * d, (size_t)(hi - lo) * d * sizeof(float));
    }
}

int cpu_scan_has_avx512(void) { return __builtin_cpu_supports("avx512f") ? 1 : 0; }
int cpu_scan_max_threads(void) { return omp_get_max_threads(); }

/* ---- the host's DRAM read rate over the same buffer, same partition: the floor of any scan ---- */
__attribute__((target("avx512f"))) static double sum_avx512(const float *p, int64_t n) {
    __m512 a0 = _mm512_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int64_t i = 0;
    for (; i + 64 <= n; i += 64) {
        a0 = _mm512_add_ps(a0, _mm512_load_ps(p + i));
        a1 = _mm512_add_ps(a1, _mm512_load_ps(p + i + 16));
        a2 = _mm512_add_ps(a2, _mm512_load_ps(p + i + 32));
        a3 = _mm512_add_ps(a3, _mm512_load_ps(p + i + 48));
    }
    return (double)_mm512_reduce_add_ps(_mm512_add_ps(_mm512_add_ps(a0, a1), _mm512_add_ps(a2, a3)));
}
__attribute__((target("avx2"))) static double sum_avx2(const float *p, int64_t n) {
    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
    int64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        a0 = _mm256_add_ps(a0, _mm256_load_ps(p + i));
        a1 = _mm256_add_ps(a1, _mm256_load_ps(p + i + 8));
        a2 = _mm256_add_ps(a2, _mm256_load_ps(p + i + 16));
        a3 = _mm256_add_ps(a3, _mm256_load_ps(p + i + 24));
    }
    float t[8];
    _mm256_storeu_ps(t, _mm256_add_ps(_mm256_add_ps(a0, a1), _mm256_add_ps(a2, a3)));
    return (double)(t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7]);
}
/* reads every row once (vector adds only); returns seconds, *sink gets the sum so nothing is elided */
double cpu_scan_read_seconds(const float *xb, int64_t n, int32_t d, int32_t threads, double *sink) {
    if (threads <= 0) threads = omp_get_max_threads();
    int wide = cpu_scan_has_avx512();
    double total = 0.0;
    double t0 = omp_get_wtime();
#pragma omp parallel num_threads(threads) reduction(+ : total)
    {
        int64_t lo, hi;
        row_range(n, omp_get_num_threads(), omp_get_thread_num(), &lo, &hi);
        if (hi > lo) total += wide ? sum_avx512(xb + lo * d, (hi - lo) * d) : sum_avx2(xb + lo * d, (hi - lo) * d);
    }
    double t1 = omp_get_wtime();
    if (sink) *sink = total;
    return t1 - t0;
}

/* ---- the cores' own FMA rate (register-only, 16 independent accumulators per thread): the compute floor of a scan ---- */
__attribute__((target("avx512f"))) static double fma_avx512(int64_t iters) {
    __m512 c[16];
    for (int i = 0; i < 16; i++) c[i] = _mm512_set1_ps(1.0f + 1e-3f * i);
    const __m512 a = _mm512_set1_ps(1.0000001f), b = _mm512_set1_ps(1e-9f);
    for (int64_t it = 0; it < iters; it++) {
#pragma GCC unroll 16
        for (int i = 0; i < 16; i++) c[i] = _mm512_fmadd_ps(c[i], a, b);
    }
    __m512 t = c[0];
    for (int i = 1; i < 16; i++) t = _mm512_add_ps(t, c[i]);
    return (double)_mm512_reduce_add_ps(t);
}
__attribute__((target("avx2,fma"))) static double fma_avx2(int64_t iters) {
    __m256 c[12];
    for (int i = 0; i < 12; i++) c[i] = _mm256_set1_ps(1.0f + 1e-3f * i);
    const __m256 a = _mm256_set1_ps(1.0000001f), b = _mm256_set1_ps(1e-9f);
    for (int64_t it = 0; it < iters; it++) {
#pragma GCC unroll 12
        for (int i = 0; i < 12; i++) c[i] = _mm256_fmadd_ps(c[i], a, b);
    }
    float t[8];
    __m256 u = c[0];
    for (int i = 1; i < 12; i++) u = _mm256_add_ps(u, c[i]);
    _mm256_storeu_ps(t, u);
    return (double)(t[0] + t[7]);
}
/* GFLOP/s of `threads` threads running nothing but vector FMAs for about `seconds` */
double cpu_scan_fma_gflops(int32_t threads, double seconds, double *sink) {
    if (threads <= 0) threads = omp_get_max_threads();
    const int wide = cpu_scan_has_avx512();
    const int64_t iters = 20 * 1000 * 1000;
    double total = 0.0, flops = 0.0;
    const double t0 = omp_get_wtime();
    int rounds = 0;
    do {
#pragma omp parallel num_threads(threads) reduction(+ : total)
        total += wide ? fma_avx512(iters) : fma_avx2(iters);
        flops += (double)threads * (double)iters * (wide ? 16 * 16 * 2 : 12 * 8 * 2);
        rounds++;
    } while (omp_get_wtime() - t0 < seconds && rounds < 1000);
    const double t1 = omp_get_wtime();
    if (sink) *sink = total;
    return flops / (t1 - t0) / 1e9;
}

/* ---- micro-kernels: 8 rows x 32 queries (AVX-512) / 4 rows x 16 queries (AVX2) ------------------
 * The queries are transposed once per search into groups qt[g][j][G] (G = 32 or 16 queries side by side), so the
 * vector dimension runs over QUERIES: for every coordinate j one broadcast of the row's y[j] feeds two FMAs, the
 * accumulators are the finished scores (no horizontal sums) and the threshold test is one vector compare per 16
 * scores.  Per coordinate: 8 broadcasts + 2 query vectors loaded, 16 FMAs -- the shape of a BLAS sgemm micro-kernel
 * with the top-k filter fused behind it.  A group's transposed queries (d x 32 x 4 B = 128 KB at d = 1024) stay in L2,
 * the 8 rows (32 KB) in L1 across the groups. */
#define ROWS512 8
#define G512 32
__attribute__((target("avx512f"))) static void scores_avx512(const float *const y[ROWS512], const float *qt, int32_t d,
                                                             float *out /* [ROWS512][G512] */) {
    __m512 c[ROWS512][2];
    for (int r = 0; r < ROWS512; r++) c[r][0] = c[r][1] = _mm512_setzero_ps();
    for (int32_t j = 0; j < d; j++) {
        const __m512 b0 = _mm512_load_ps(qt + (size_t)j * G512), b1 = _mm512_load_ps(qt + (size_t)j * G512 + 16);
#pragma GCC unroll 8
        for (int r = 0; r < ROWS512; r++) {
            const __m512 a = _mm512_set1_ps(y[r][j]);
            c[r][0] = _mm512_fmadd_ps(a, b0, c[r][0]);
            c[r][1] = _mm512_fmadd_ps(a, b1, c[r][1]);
        }
    }
    for (int r = 0; r < ROWS512; r++) {
        _mm512_storeu_ps(out + r * G512, c[r][0]);
        _mm512_storeu_ps(out + r * G512 + 16, c[r][1]);
    }
}
#define ROWS256 4
#define G256 16
__attribute__((target("avx2,fma"))) static void scores_avx2(const float *const y[ROWS256], const float *qt, int32_t d,
                                                            float *out /* [ROWS256][G256] */) {
    __m256 c[ROWS256][2];
    for (int r = 0; r < ROWS256; r++) c[r][0] = c[r][1] = _mm256_setzero_ps();
    for (int32_t j = 0; j < d; j++) {
        const __m256 b0 = _mm256_load_ps(qt + (size_t)j * G256), b1 = _mm256_load_ps(qt + (size_t)j * G256 + 8);
#pragma GCC unroll 4
        for (int r = 0; r < ROWS256; r++) {
            const __m256 a = _mm256_broadcast_ss(y[r] + j);
            c[r][0] = _mm256_fmadd_ps(a, b0, c[r][0]);
            c[r][1] = _mm256_fmadd_ps(a, b1, c[r][1]);
        }
    }
    for (int r = 0; r < ROWS256; r++) {
        _mm256_storeu_ps(out + r * G256, c[r][0]);
        _mm256_storeu_ps(out + r * G256 + 8, c[r][1]);
    }
}

/* which of a row's G scores pass their query's threshold (v <= thr): one or two vector compares instead of G scalar ones --
 * after the first few thousand rows almost nothing passes */
__attribute__((target("avx512f"))) static uint32_t pass_mask_avx512(const float *sc, const float *thr, const float *xn, float ynr, int metric) {
    uint32_t m = 0;
    for (int h = 0; h < 2; h++) {
        const __m512 s = _mm512_loadu_ps(sc + 16 * h), t = _mm512_loadu_ps(thr + 16 * h);
        __m512 v;
        if (metric == 0) {
            v = _mm512_sub_ps(_mm512_setzero_ps(), s);
        } else {
            v = _mm512_sub_ps(_mm512_add_ps(_mm512_loadu_ps(xn + 16 * h), _mm512_set1_ps(ynr)), _mm512_add_ps(s, s));
            v = _mm512_max_ps(v, _mm512_setzero_ps());
        }
        m |= (uint32_t)_mm512_cmp_ps_mask(v, t, _CMP_LE_OQ) << (16 * h);
    }
    return m;
}

static float dot_scalar(const float *a, const float *b, int32_t d) {
    float t = 0.f;
    for (int32_t j = 0; j < d; j++) t += a[j] * b[j];
    return t;
}

/* |y|^2 of every row (FAISS computes these per block inside knn_L2sqr_blas; kept outside the timed
 * scan the way the GPU index keeps them from add()) */
void cpu_scan_norms(const float *xb, int64_t n, int32_t d, float *yn, int32_t threads) {
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < n; i++) yn[i] = dot_scalar(xb + i * d, xb + i * d, d);
}

/* per-thread, per-query candidate buffer: keeps everything that beats thr; at capacity the best k stay
 * and thr becomes the k-th's value */
typedef struct {
    cand_t *c;
    int32_t cnt;
    float thr;
} qbuf_t;

static void qbuf_shrink(qbuf_t *b, int64_t k) {
    qsort(b->c, (size_t)b->cnt, sizeof(cand_t), cand_less);
    if (b->cnt > k) b->cnt = (int32_t)k;
    if (b->cnt == k) b->thr = b->c[k - 1].v;
}

/*
 * metric 0: inner product (D = score, descending), 1: squared L2 by FAISS's BLAS formula (ascending).
 * yn: |y|^2 per row (L2 only; may be NULL for IP).  D/I: [nq][k], unfilled slots id -1, -/+FLT_MAX.
 * Returns 0, or -1 on bad arguments / out of memory.
 */
int cpu_scan_search(const float *xb, const float *yn, int64_t nb, const float *xq, int64_t nq, int32_t d, int32_t metric,
                    int64_t k, float *D, int64_t *I, int32_t threads) {
    if (nb < 0 || nq < 0 || d <= 0 || k <= 0 || (metric != 0 && metric != 1)) return -1;
    if (metric == 1 && !yn && nb > 0) return -1;
    if (threads <= 0) threads = omp_get_max_threads();
    const int wide = cpu_scan_has_avx512();
    const int G = wide ? G512 : G256, RB = wide ? ROWS512 : ROWS256;
    const int64_t cap = 2 * k + 16;
    const int64_t ngroups = (nq + G - 1) / G, nqp = ngroups * G;

    /* queries transposed by group: qt[g][j][G], zero padded; their norms; the running thresholds */
    float *qt = NULL, *xn = (float *)calloc((size_t)(nqp ? nqp : 1), sizeof(float));
    if (posix_memalign((void **)&qt, 64, (size_t)(nqp ? nqp : G) * d * sizeof(float)) || !xn) return -1;
    memset(qt, 0, (size_t)(nqp ? nqp : G) * d * sizeof(float));
    for (int64_t i = 0; i < nq; i++) {
        const float *src = xq + i * d;
        float *dst = qt + (i / G) * (size_t)G * d + (i % G);
        for (int32_t j = 0; j < d; j++) dst[(size_t)j * G] = src[j];
        if (metric == 1) xn[i] = dot_scalar(src, src, d);
    }

    qbuf_t *bufs = (qbuf_t *)calloc((size_t)threads * (nq ? nq : 1), sizeof(qbuf_t));
    int failed = bufs == NULL;
    int used_threads = threads;
#pragma omp parallel num_threads(threads)
    {
        const int nth = omp_get_num_threads(), t = omp_get_thread_num();
#pragma omp single
        used_threads = nth;
        qbuf_t *mine = bufs ? bufs + (size_t)t * nq : NULL;
        float *thr = (float *)malloc((size_t)(nqp ? nqp : 1) * sizeof(float)); /* thr[q]: vector-loadable copy of mine[q].thr */
        if (mine && thr) {
            for (int64_t i = 0; i < nqp; i++) thr[i] = i < nq ? FLT_MAX : -FLT_MAX; /* padding queries admit nothing */
            for (int64_t i = 0; i < nq; i++) {
                mine[i].c = (cand_t *)malloc((size_t)cap * sizeof(cand_t));
                mine[i].cnt = 0;
                mine[i].thr = FLT_MAX;
                if (!mine[i].c) {
#pragma omp atomic write
                    failed = 1;
                }
            }
        } else {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp barrier
        if (!failed) {
            int64_t lo, hi;
            row_range(nb, nth, t, &lo, &hi);
            float sc[ROWS512 * G512] __attribute__((aligned(64)));
            const float *y[ROWS512];
            for (int64_t r = lo; r < hi; r += RB) {
                /* a ragged last block re-reads the last row; its extra results are dropped */
                const int rows = (int)(hi - r < RB ? hi - r : RB);
                for (int i = 0; i < RB; i++) y[i] = xb + (r + (i < rows ? i : rows - 1)) * d;
                for (int64_t g = 0; g < ngroups; g++) {
                    if (wide) scores_avx512(y, qt + (size_t)g * G * d, d, sc);
                    else scores_avx2(y, qt + (size_t)g * G * d, d, sc);
                    const float *tg = thr + g * G;
                    for (int rr = 0; rr < rows; rr++) {
                        const float ynr = metric == 1 ? yn[r + rr] : 0.f;
                        uint32_t pm = wide ? pass_mask_avx512(sc + rr * G, tg, xn + g * G, ynr, metric) : 0xFFFFu;
                        for (int qq = 0; qq < G && pm; qq++, pm >>= 1) {
                            if (!(pm & 1u)) continue;
                            float v;
                            if (metric == 0) {
                                v = -sc[rr * G + qq];
                            } else {
                                v = xn[g * G + qq] + ynr - 2.f * sc[rr * G + qq];
                                if (v < 0.f) v = 0.f;
                            }
                            if (__builtin_expect(v <= tg[qq], 0)) {
                                qbuf_t *b = &mine[g * G + qq];
                                if (v < b->thr || b->cnt < k) {
                                    b->c[b->cnt].v = v;
                                    b->c[b->cnt].id = r + rr;
                                    if (++b->cnt == cap) {
                                        qbuf_shrink(b, k);
                                        thr[g * G + qq] = b->thr;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        free(thr);
    }
    /* merge: per query, all threads' candidates, best k */
    if (!failed) {
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (int64_t i = 0; i < nq; i++) {
            int64_t tot = 0;
            for (int t = 0; t < used_threads; t++) tot += bufs[(size_t)t * nq + i].cnt;
            cand_t *all = (cand_t *)malloc((size_t)(tot ? tot : 1) * sizeof(cand_t));
            int64_t m = 0;
            for (int t = 0; t < used_threads; t++) {
                qbuf_t *b = &bufs[(size_t)t * nq + i];
                memcpy(all + m, b->c, (size_t)b->cnt * sizeof(cand_t));
                m += b->cnt;
            }
            qsort(all, (size_t)m, sizeof(cand_t), cand_less);
            for (int64_t j = 0; j < k; j++) {
                if (j < m && !isnan(all[j].v)) {
                    D[i * k + j] = metric == 0 ? -all[j].v : all[j].v;
                    I[i * k + j] = all[j].id;
                } else {
                    D[i * k + j] = metric == 0 ? -FLT_MAX : FLT_MAX;
                    I[i * k + j] = -1;
                }
            }
            free(all);
        }
    }
    if (bufs) {
        for (size_t i = 0; i < (size_t)used_threads * nq; i++) free(bufs[i].c);
        free(bufs);
    }
    free(qt);
    free(xn);
    return failed ? -1 : 0;
}
