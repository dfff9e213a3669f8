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
 * AVX2+FMA) inner-product micro-kernel (4 rows x 4 queries, 16 vector accumulators), FAISS's L2 formula
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

/* ---- micro-kernels: dots of 4 rows x 4 queries ------------------------------------------------ */
__attribute__((target("avx512f"))) static void dots4x4_avx512(const float *y0, const float *y1, const float *y2,
                                                              const float *y3, const float *q, int64_t qstride,
                                                              int32_t d, float out[16]) {
    __m512 c[16];
    for (int i = 0; i < 16; i++) c[i] = _mm512_setzero_ps();
    const float *q0 = q, *q1 = q + qstride, *q2 = q + 2 * qstride, *q3 = q + 3 * qstride;
    int32_t j = 0;
    for (; j + 16 <= d; j += 16) {
        __m512 a0 = _mm512_loadu_ps(y0 + j), a1 = _mm512_loadu_ps(y1 + j);
        __m512 a2 = _mm512_loadu_ps(y2 + j), a3 = _mm512_loadu_ps(y3 + j);
        __m512 b = _mm512_loadu_ps(q0 + j);
        c[0] = _mm512_fmadd_ps(a0, b, c[0]); c[4] = _mm512_fmadd_ps(a1, b, c[4]);
        c[8] = _mm512_fmadd_ps(a2, b, c[8]); c[12] = _mm512_fmadd_ps(a3, b, c[12]);
        b = _mm512_loadu_ps(q1 + j);
        c[1] = _mm512_fmadd_ps(a0, b, c[1]); c[5] = _mm512_fmadd_ps(a1, b, c[5]);
        c[9] = _mm512_fmadd_ps(a2, b, c[9]); c[13] = _mm512_fmadd_ps(a3, b, c[13]);
        b = _mm512_loadu_ps(q2 + j);
        c[2] = _mm512_fmadd_ps(a0, b, c[2]); c[6] = _mm512_fmadd_ps(a1, b, c[6]);
        c[10] = _mm512_fmadd_ps(a2, b, c[10]); c[14] = _mm512_fmadd_ps(a3, b, c[14]);
        b = _mm512_loadu_ps(q3 + j);
        c[3] = _mm512_fmadd_ps(a0, b, c[3]); c[7] = _mm512_fmadd_ps(a1, b, c[7]);
        c[11] = _mm512_fmadd_ps(a2, b, c[11]); c[15] = _mm512_fmadd_ps(a3, b, c[15]);
    }
    for (int i = 0; i < 16; i++) out[i] = _mm512_reduce_add_ps(c[i]);
    if (j < d) {
        const float *ys[4] = {y0, y1, y2, y3};
        const float *qs[4] = {q0, q1, q2, q3};
        for (int r = 0; r < 4; r++)
            for (int s = 0; s < 4; s++) {
                float t = 0.f;
                for (int32_t jj = j; jj < d; jj++) t += ys[r][jj] * qs[s][jj];
                out[r * 4 + s] += t;
            }
    }
}

__attribute__((target("avx2,fma"))) static float hsum256(__m256 v) {
    __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
    lo = _mm_add_ps(lo, hi);
    lo = _mm_hadd_ps(lo, lo);
    lo = _mm_hadd_ps(lo, lo);
    return _mm_cvtss_f32(lo);
}
/* AVX2 has 16 vector registers: 2 rows x 4 queries per pass, twice */
__attribute__((target("avx2,fma"))) static void dots4x4_avx2(const float *y0, const float *y1, const float *y2,
                                                             const float *y3, const float *q, int64_t qstride, int32_t d,
                                                             float out[16]) {
    const float *ys[4] = {y0, y1, y2, y3};
    const float *q0 = q, *q1 = q + qstride, *q2 = q + 2 * qstride, *q3 = q + 3 * qstride;
    for (int half = 0; half < 2; half++) {
        const float *ya = ys[2 * half], *yb = ys[2 * half + 1];
        __m256 c[8];
        for (int i = 0; i < 8; i++) c[i] = _mm256_setzero_ps();
        int32_t j = 0;
        for (; j + 8 <= d; j += 8) {
            __m256 a0 = _mm256_loadu_ps(ya + j), a1 = _mm256_loadu_ps(yb + j);
            __m256 b = _mm256_loadu_ps(q0 + j);
            c[0] = _mm256_fmadd_ps(a0, b, c[0]); c[4] = _mm256_fmadd_ps(a1, b, c[4]);
            b = _mm256_loadu_ps(q1 + j);
            c[1] = _mm256_fmadd_ps(a0, b, c[1]); c[5] = _mm256_fmadd_ps(a1, b, c[5]);
            b = _mm256_loadu_ps(q2 + j);
            c[2] = _mm256_fmadd_ps(a0, b, c[2]); c[6] = _mm256_fmadd_ps(a1, b, c[6]);
            b = _mm256_loadu_ps(q3 + j);
            c[3] = _mm256_fmadd_ps(a0, b, c[3]); c[7] = _mm256_fmadd_ps(a1, b, c[7]);
        }
        for (int i = 0; i < 8; i++) out[half * 8 + i] = hsum256(c[i]);
        if (j < d) {
            const float *qs[4] = {q0, q1, q2, q3};
            for (int r = 0; r < 2; r++)
                for (int s = 0; s < 4; s++) {
                    float t = 0.f;
                    for (int32_t jj = j; jj < d; jj++) t += ys[2 * half + r][jj] * qs[s][jj];
                    out[half * 8 + r * 4 + s] += t;
                }
        }
    }
}

typedef void (*dots_fn)(const float *, const float *, const float *, const float *, const float *, int64_t, int32_t,
                        float[16]);

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
    dots_fn dots = cpu_scan_has_avx512() ? dots4x4_avx512 : dots4x4_avx2;
    const int64_t cap = 2 * k + 16;
    const int64_t nq4 = (nq + 3) & ~(int64_t)3;

    /* queries padded to a multiple of 4 rows + their norms */
    float *q = NULL, *xn = (float *)calloc((size_t)(nq4 ? nq4 : 1), sizeof(float));
    if (posix_memalign((void **)&q, 64, (size_t)(nq4 ? nq4 : 4) * d * sizeof(float)) || !xn) return -1;
    memset(q, 0, (size_t)(nq4 ? nq4 : 4) * d * sizeof(float));
    memcpy(q, xq, (size_t)nq * d * sizeof(float));
    if (metric == 1)
        for (int64_t i = 0; i < nq; i++) xn[i] = dot_scalar(q + i * d, q + i * d, d);

    qbuf_t *bufs = (qbuf_t *)calloc((size_t)threads * (nq ? nq : 1), sizeof(qbuf_t));
    int failed = bufs == NULL;
    int used_threads = threads;
#pragma omp parallel num_threads(threads)
    {
        const int nth = omp_get_num_threads(), t = omp_get_thread_num();
#pragma omp single
        used_threads = nth;
        qbuf_t *mine = bufs ? bufs + (size_t)t * nq : NULL;
        if (mine) {
            for (int64_t i = 0; i < nq; i++) {
                mine[i].c = (cand_t *)malloc((size_t)cap * sizeof(cand_t));
                mine[i].cnt = 0;
                mine[i].thr = FLT_MAX;
                if (!mine[i].c) {
#pragma omp atomic write
                    failed = 1;
                }
            }
        }
#pragma omp barrier
        if (!failed) {
            int64_t lo, hi;
            row_range(nb, nth, t, &lo, &hi);
            float s[16];
            for (int64_t r = lo; r < hi; r += 4) {
                /* a ragged last block re-reads the last row; its extra results are dropped */
                const int64_t last = hi - 1;
                const float *y0 = xb + r * d;
                const float *y1 = xb + (r + 1 <= last ? r + 1 : last) * d;
                const float *y2 = xb + (r + 2 <= last ? r + 2 : last) * d;
                const float *y3 = xb + (r + 3 <= last ? r + 3 : last) * d;
                const int rows = (int)(hi - r < 4 ? hi - r : 4);
                for (int64_t qb = 0; qb < nq4; qb += 4) {
                    dots(y0, y1, y2, y3, q + qb * d, d, d, s);
                    for (int rr = 0; rr < rows; rr++)
                        for (int qq = 0; qq < 4 && qb + qq < nq; qq++) {
                            float ip = s[rr * 4 + qq], v;
                            if (metric == 0) {
                                v = -ip;
                            } else {
                                v = xn[qb + qq] + yn[r + rr] - 2.f * ip;
                                if (v < 0.f) v = 0.f;
                            }
                            qbuf_t *b = &mine[qb + qq];
                            if (v < b->thr || (v == b->thr && b->cnt < k)) {
                                b->c[b->cnt].v = v;
                                b->c[b->cnt].id = r + rr;
                                if (++b->cnt == cap) qbuf_shrink(b, k);
                            }
                        }
                }
            }
        }
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
    free(q);
    free(xn);
    return failed ? -1 : 0;
}
