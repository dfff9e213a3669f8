/*
 * oracle/knn_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product path (libknn355.so) never
 * links, loads or calls it.
 *
 * What it restates
 * ----------------
 * The reference (konstin/knn-for-homology) has no arithmetic of its own on
 * this path: every distance and every top-k is a call into the third-party
 * wheel faiss-cpu 1.7.2 (pyproject.toml:18, poetry.lock:100-101), which is
 * NOT present under /root/reference.  Call sites being restated:
 *   faiss.normalize_L2      cath/search.py:19, pfam/proteins_search.py:22,
 *                           seqvec_search/main.py:31,34, pfam/search.py:18,20
 *   IndexFlat.add/.search   cath/search.py:20-24, seqvec_search/main.py:35-45,
 *                           pfam/proteins_search.py:24,37,49, pfam/search.py:44-51
 * FAISS 1.7.2's published flat algorithm (utils/distances.cpp: knn_inner_product,
 * knn_L2sqr; fvec_renorm_L2):
 *   - IP score   = <x, y>                          (larger is better)
 *   - L2 score   = ||x||^2 + ||y||^2 - 2<x,y>, negative values clamped to 0
 *                  (the BLAS path used for nq >= 20; squared, no sqrt)
 *   - normalize  = x[j] *= (float)(1.0 / sqrtf(sum x[j]^2)); rows whose squared
 *                  norm is 0 are left untouched
 *   - results sorted best-first; slots that cannot be filled get id -1 and
 *                  -FLT_MAX (IP) / +FLT_MAX (L2)
 * FAISS's own summation order depends on the BLAS kernel and thread count, so
 * raw fp32 bits are "parity unpinned" against FAISS itself; the pin is
 *   (1) the reference's known-answer tests (tests/test_main.py:10-27) which
 *       this oracle reproduces through the reference's own faiss_search +
 *       evaluate_faiss (tests/golden/make_golden.py), and
 *   (2) an fp64 ground truth (oracle/knn_oracle.py) with a tie-tolerant
 *       comparator.
 *
 * The fp32 evaluation order ("knn355 arithmetic contract")
 * --------------------------------------------------------
 * IEEE-754 binary32, round-to-nearest-even, one rounding per fused
 * multiply-add, no wider accumulator:
 *   dot(x,y):  acc = 0; for each block of 8 consecutive k (zero padded),
 *              in the order 0,4,1,5,2,6,3,7:  acc = fmaf(x[k], y[k], acc)
 *   nrm(x):    dot(x, x) -- the same chain, so L2(x, x) is exactly 0
 *   L2(x,y):   t = nrm(x) + nrm(y);  s = fmaf(-2, dot(x,y), t);  max(s, 0)
 *   ties:      equal scores are ordered by ascending database row id
 * The HIP kernels are written to produce exactly these bits, so GPU results
 * are compared bit-for-bit (ids and distances) against this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_METRIC_INNER_PRODUCT 0
#define ORC_METRIC_L2 1

static const int P8[8] = {0, 4, 1, 5, 2, 6, 3, 7};

float orc_dot(const float *x, const float *y, int d)
{
    float acc = 0.0f;
    int k0;
    for (k0 = 0; k0 + 8 <= d; k0 += 8)
        for (int j = 0; j < 8; j++)
            acc = fmaf(x[k0 + P8[j]], y[k0 + P8[j]], acc);
    if (k0 < d) {
        for (int j = 0; j < 8; j++) {
            int k = k0 + P8[j];
            float a = k < d ? x[k] : 0.0f, b = k < d ? y[k] : 0.0f;
            acc = fmaf(a, b, acc);
        }
    }
    return acc;
}

/* nrm(x) is by definition dot(x, x): the same chain, so an L2 self distance
 * nrm(x)+nrm(x)-2*dot(x,x) is exactly 0 */
float orc_norm_one(const float *x, int d) { return orc_dot(x, x, d); }

void orc_norm_l2sqr(const float *x, int64_t n, int d, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = orc_norm_one(x + i * (int64_t)d, d);
}

/* faiss.normalize_L2 (fvec_renorm_L2): in place, zero rows untouched */
void orc_normalize_l2(float *x, int64_t n, int d)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float *xi = x + i * (int64_t)d;
        float nr = orc_norm_one(xi, d);
        if (nr > 0) {
            const float inv = (float)(1.0 / (double)sqrtf(nr));
            for (int j = 0; j < d; j++) xi[j] *= inv;
        }
    }
}

/* orderable key: smaller key == better hit.  hi = monotone map of the
 * "smaller is better" float value, lo = row id (ties -> lower id first). */
static inline uint32_t f2ord(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float ord2f(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline float score_to_v(int metric, float ip, float xn, float yn)
{
    float v;
    if (metric == ORC_METRIC_INNER_PRODUCT) {
        v = -ip;
    } else {
        float t = xn + yn;
        v = fmaf(-2.0f, ip, t);
        if (v < 0.0f) v = 0.0f;
    }
    return v + 0.0f; /* canonical +0 */
}

/* bounded max-heap on u64 keys (keeps the k smallest) */
static inline void heap_sift_down(uint64_t *h, int64_t n, int64_t i)
{
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] > h[m]) m = l;
        if (r < n && h[r] > h[m]) m = r;
        if (m == i) return;
        uint64_t t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

#define RB 16 /* database rows per transposed block (one SIMD lane per chain) */

/*
 * IndexFlat.search restated: xb [nb,d], xq [nq,d] row-major fp32.
 * D [nq,k] fp32, I [nq,k] int64, best first.  Returns 0.
 */
/* FAISS computes squared L2 two ways [ext: IndexFlat -> knn_L2sqr, utils/distances.cpp]: batches of fewer than
 * distance_compute_blas_threshold = 20 queries take the SIMD path, fvec_L2sqr = sum (x - y)^2; larger batches the BLAS
 * path, |x|^2 + |y|^2 - 2<x,y> clamped at 0.  Mode 0 (default) follows that rule; 1 = the norm formula whatever the batch
 * (the arithmetic of the large-batch kernels, used by tests that compare the two); 2 = differences whatever the batch.
 * Contract of the difference form: one fp32 chain per pair, acc = fmaf(t, t, acc), t = x[k] - y[k], k in the dot
 * product's order. */
static int g_l2_mode = 0;
void orc_set_l2_mode(int mode) { g_l2_mode = mode; }
int orc_l2_is_direct(int64_t nq) { return g_l2_mode == 2 || (g_l2_mode == 0 && nq < 20); }

int orc_flat_search(const float *xb, int64_t nb, const float *xq, int64_t nq, int d,
                    int metric, int64_t k, float *D, int64_t *I)
{
    if (d <= 0 || k <= 0 || nb < 0 || nq < 0) return -1;
    const int direct = metric == ORC_METRIC_L2 && orc_l2_is_direct(nq);
    const int dp = (d + 7) & ~7;
    float *yn = NULL, *xn = NULL;
    if (metric == ORC_METRIC_L2) {
        yn = (float *)malloc(sizeof(float) * (size_t)(nb > 0 ? nb : 1));
        xn = (float *)malloc(sizeof(float) * (size_t)(nq > 0 ? nq : 1));
        orc_norm_l2sqr(xb, nb, d, yn);
        orc_norm_l2sqr(xq, nq, d, xn);
    }
    const int64_t nblk = (nb + RB - 1) / RB;
    /* k-major transposed copy of the database, chain order already applied */
    float *yt = (float *)malloc(sizeof(float) * (size_t)(nblk > 0 ? nblk : 1) * RB * dp);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nblk; b++) {
        float *dst = yt + (size_t)b * RB * dp;
        for (int s = 0; s < dp; s++) {
            int k0 = s & ~7, kk = k0 + P8[s & 7];
            for (int r = 0; r < RB; r++) {
                int64_t row = b * RB + r;
                dst[(size_t)s * RB + r] = (row < nb && kk < d) ? xb[row * (int64_t)d + kk] : 0.0f;
            }
        }
    }
    const float pad_d = metric == ORC_METRIC_INNER_PRODUCT ? -FLT_MAX : FLT_MAX;
#pragma omp parallel
    {
        float *qp = (float *)malloc(sizeof(float) * dp);
        uint64_t *heap = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)k);
#pragma omp for schedule(dynamic, 4)
        for (int64_t qi = 0; qi < nq; qi++) {
            const float *q = xq + qi * (int64_t)d;
            for (int s = 0; s < dp; s++) {
                int kk = (s & ~7) + P8[s & 7];
                qp[s] = kk < d ? q[kk] : 0.0f;
            }
            int64_t hn = 0;
            for (int64_t b = 0; b < nblk; b++) {
                const float *blk = yt + (size_t)b * RB * dp;
                float acc[RB];
                for (int r = 0; r < RB; r++) acc[r] = 0.0f;
                if (direct) {
                    for (int s = 0; s < dp; s++) {
                        const float qs = qp[s];
                        const float *row = blk + (size_t)s * RB;
                        for (int r = 0; r < RB; r++) {
                            const float t = qs - row[r];
                            acc[r] = fmaf(t, t, acc[r]);
                        }
                    }
                } else {
                    for (int s = 0; s < dp; s++) {
                        const float qs = qp[s];
                        const float *row = blk + (size_t)s * RB;
                        for (int r = 0; r < RB; r++) acc[r] = fmaf(qs, row[r], acc[r]);
                    }
                }
                for (int r = 0; r < RB; r++) {
                    int64_t row = b * RB + r;
                    if (row >= nb) break;
                    float v = direct ? acc[r] + 0.0f : score_to_v(metric, acc[r], xn ? xn[qi] : 0.0f, yn ? yn[row] : 0.0f);
                    if (!(v < INFINITY)) continue; /* NaN / +inf ("worse than everything") never becomes a hit */
                    uint64_t key = ((uint64_t)f2ord(v) << 32) | (uint32_t)row;
                    if (hn < k) {
                        heap[hn++] = key;
                        if (hn == k)
                            for (int64_t i = k / 2 - 1; i >= 0; i--) heap_sift_down(heap, k, i);
                    } else if (key < heap[0]) {
                        heap[0] = key;
                        heap_sift_down(heap, k, 0);
                    }
                }
            }
            qsort(heap, (size_t)hn, sizeof(uint64_t), cmp_u64);
            for (int64_t j = 0; j < k; j++) {
                if (j < hn) {
                    float v = ord2f((uint32_t)(heap[j] >> 32));
                    D[qi * k + j] = metric == ORC_METRIC_INNER_PRODUCT ? -v : v;
                    I[qi * k + j] = (int64_t)(uint32_t)heap[j];
                } else {
                    D[qi * k + j] = pad_d;
                    I[qi * k + j] = -1;
                }
            }
        }
        free(qp);
        free(heap);
    }
    free(yt);
    free(yn);
    free(xn);
    return 0;
}

/*
 * Distances for explicit (query, row) pairs -- the HNSW candidate path.
 * out[p] = IP: <q,y>;  L2: max(0, nrm(q)+nrm(y)-2<q,y>)
 */
void orc_pair_distances(const float *xb, const float *xq, int d, int metric, int64_t npairs,
                        const int64_t *qidx, const int64_t *ridx, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < npairs; p++) {
        const float *q = xq + qidx[p] * (int64_t)d, *y = xb + ridx[p] * (int64_t)d;
        float ip = orc_dot(q, y, d);
        if (metric == ORC_METRIC_INNER_PRODUCT) out[p] = ip;
        else {
            float v = fmaf(-2.0f, ip, orc_norm_one(q, d) + orc_norm_one(y, d));
            out[p] = v < 0.0f ? 0.0f : v;
        }
    }
}
