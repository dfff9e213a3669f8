/*
 * oracle/hnsw_oracle.c -- CPU ORACLE for faiss.IndexHNSWFlat (TEST INFRASTRUCTURE ONLY).
 *
 * Restates the published HNSW algorithm with FAISS 1.7.2's parameters, as the reference uses
 * it (pfam/proteins_search.py:27-31: IndexHNSWFlat(d, 42, METRIC_INNER_PRODUCT),
 * hnsw.efSearch = 256; FAISS defaults efConstruction = 40, level multiplier 1/ln(M), 2M links
 * at level 0): strictly SEQUENTIAL insertion (one point at a time, every earlier point
 * visible), greedy descent above the point's level, beam search with efConstruction at and
 * below it, neighbour selection by the "closer to the centre than to any kept neighbour"
 * heuristic for both the new point's links and overflowing reverse links, search with
 * ef = max(efSearch, k).  Distances come from knn_oracle.c (same fp32 contract).
 *
 * FAISS's own graph is not reproducible (OpenMP insertion order), so this oracle does not
 * pin ids; it pins QUALITY: tests require the GPU-offloaded, batch-synchronous build of
 * libknn355 to reach the recall this sequential build reaches (within a small margin).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float orc_dot(const float *x, const float *y, int d);
float orc_norm_one(const float *x, int d);

typedef struct { float v; int32_t id; } DI;

typedef struct {
    int d, M, metric, efc, efs;
    int64_t n, cap;
    float *x, *nrm;
    int *level;
    int64_t *off;    /* slot offset of node i */
    int32_t *nb;     /* neighbour slots, -1 empty */
    int64_t nslots, slots_cap;
    int maxlevel;
    int64_t entry;
    uint64_t rng;
    int *cum; int ncum;
    double *probas;
    uint32_t *vis; uint32_t epoch; int64_t vis_cap;
} H;

static double rnd(H *h)
{
    h->rng = h->rng * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(h->rng >> 11) / 9007199254740992.0;
}
static float dist(const H *h, const float *q, float qn, int64_t j)
{
    float ip = orc_dot(q, h->x + j * h->d, h->d);
    if (h->metric == 0) return -ip + 0.0f;
    float v = fmaf(-2.0f, ip, qn + h->nrm[j]);
    return v < 0 ? 0 : v;
}
static int lt(DI a, DI b) { return a.v < b.v || (a.v == b.v && a.id < b.id); }
static int nbn(const H *h, int l) { return h->cum[l + 1] - h->cum[l]; }
static int32_t *lst(const H *h, int64_t i, int l) { return h->nb + h->off[i] + h->cum[l]; }

H *hnsw_orc_new(int d, int M, int metric, int efc)
{
    H *h = calloc(1, sizeof(H));
    h->d = d; h->M = M; h->metric = metric; h->efc = efc; h->efs = 16;
    h->maxlevel = -1; h->entry = -1; h->rng = 12345;
    double mult = 1.0 / log((double)M);
    h->cum = malloc(sizeof(int) * 64); h->probas = malloc(sizeof(double) * 64);
    h->cum[0] = 0; int nn = 0;
    for (int l = 0;; l++) {
        double p = exp(-l / mult) * (1 - exp(-1 / mult));
        if (p < 1e-9) break;
        h->probas[l] = p; nn += l == 0 ? 2 * M : M; h->cum[l + 1] = nn; h->ncum = l + 2;
    }
    return h;
}
void hnsw_orc_free(H *h)
{
    free(h->x); free(h->nrm); free(h->level); free(h->off); free(h->nb); free(h->cum); free(h->probas); free(h->vis); free(h);
}
void hnsw_orc_set_ef(H *h, int efs) { h->efs = efs; }

/* max-heap / min-heap helpers on DI arrays */
static void push_max(DI *a, int *n, DI e) { int i = (*n)++; a[i] = e; while (i) { int p = (i - 1) / 2; if (lt(a[p], a[i])) { DI t = a[p]; a[p] = a[i]; a[i] = t; i = p; } else break; } }
static void pop_max(DI *a, int *n) { a[0] = a[--(*n)]; int i = 0; for (;;) { int l = 2 * i + 1, r = l + 1, m = i; if (l < *n && lt(a[m], a[l])) m = l; if (r < *n && lt(a[m], a[r])) m = r; if (m == i) break; DI t = a[m]; a[m] = a[i]; a[i] = t; i = m; } }
static void push_min(DI *a, int *n, DI e) { int i = (*n)++; a[i] = e; while (i) { int p = (i - 1) / 2; if (lt(a[i], a[p])) { DI t = a[p]; a[p] = a[i]; a[i] = t; i = p; } else break; } }
static void pop_min(DI *a, int *n) { a[0] = a[--(*n)]; int i = 0; for (;;) { int l = 2 * i + 1, r = l + 1, m = i; if (l < *n && lt(a[l], a[m])) m = l; if (r < *n && lt(a[r], a[m])) m = r; if (m == i) break; DI t = a[m]; a[m] = a[i]; a[i] = t; i = m; } }
static int cmp_di(const void *a, const void *b) { DI x = *(const DI *)a, y = *(const DI *)b; return lt(x, y) ? -1 : lt(y, x) ? 1 : 0; }

/* beam search at one level; W (sorted ascending) returns up to ef results */
static int beam(H *h, const float *q, float qn, DI start, int level, int ef, DI *W, DI *C, int64_t nvis)
{
    if (h->vis_cap < nvis) { h->vis = realloc(h->vis, sizeof(uint32_t) * nvis); memset(h->vis + h->vis_cap, 0, sizeof(uint32_t) * (nvis - h->vis_cap)); h->vis_cap = nvis; }
    if (++h->epoch == 0) { memset(h->vis, 0, sizeof(uint32_t) * h->vis_cap); h->epoch = 1; }
    int nw = 0, nc = 0;
    push_max(W, &nw, start); push_min(C, &nc, start); h->vis[start.id] = h->epoch;
    while (nc) {
        DI c = C[0];
        if (nw >= ef && lt(W[0], c)) break;
        pop_min(C, &nc);
        int32_t *l = lst(h, c.id, level);
        for (int j = 0; j < nbn(h, level) && l[j] >= 0; j++) {
            int32_t e = l[j];
            if (h->vis[e] == h->epoch) continue;
            h->vis[e] = h->epoch;
            DI de = { dist(h, q, qn, e), e };
            if (nw < ef || lt(de, W[0])) {
                push_min(C, &nc, de); push_max(W, &nw, de);
                if (nw > ef) pop_max(W, &nw);
            }
        }
    }
    qsort(W, nw, sizeof(DI), cmp_di);
    return nw;
}

/* cand sorted ascending by distance to the centre; keeps at most maxo */
static int shrink(H *h, const DI *cand, int nc, int maxo, DI *out)
{
    int no = 0;
    for (int i = 0; i < nc && no < maxo; i++) {
        int good = 1;
        const float *xi = h->x + (int64_t)cand[i].id * h->d;
        for (int o = 0; o < no; o++) {
            if (out[o].id == cand[i].id) { good = 0; break; }
            float dv = dist(h, xi, h->nrm[cand[i].id], out[o].id);
            if (dv < cand[i].v) { good = 0; break; }
        }
        if (good) out[no++] = cand[i];
    }
    return no;
}

void hnsw_orc_add(H *h, const float *x, int64_t n)
{
    int64_t n0 = h->n, n1 = n0 + n;
    h->x = realloc(h->x, sizeof(float) * n1 * h->d);
    h->nrm = realloc(h->nrm, sizeof(float) * n1);
    h->level = realloc(h->level, sizeof(int) * n1);
    h->off = realloc(h->off, sizeof(int64_t) * (n1 + 1));
    memcpy(h->x + n0 * h->d, x, sizeof(float) * n * h->d);
    int efmax = h->efc > 2 * h->M + 2 ? h->efc : 2 * h->M + 2;
    DI *W = malloc(sizeof(DI) * (efmax + 2)), *C = malloc(sizeof(DI) * (n1 + 8)), *sel = malloc(sizeof(DI) * (2 * h->M + 2)), *tmp = malloc(sizeof(DI) * (2 * h->M + 4));
    if (n0 == 0) h->off[0] = 0;
    for (int64_t i = n0; i < n1; i++) {
        const float *q = h->x + i * h->d;
        h->nrm[i] = orc_norm_one(q, h->d);
        double f = rnd(h); int lv = h->ncum - 2;
        for (int l = 0; l < h->ncum - 1; l++) { if (f < h->probas[l]) { lv = l; break; } f -= h->probas[l]; }
        h->level[i] = lv;
        h->off[i + 1] = h->off[i] + h->cum[lv + 1];
        h->nb = realloc(h->nb, sizeof(int32_t) * h->off[i + 1]);
        for (int64_t s = h->off[i]; s < h->off[i + 1]; s++) h->nb[s] = -1;
        h->n = i + 1;
        if (h->entry < 0) { h->entry = i; h->maxlevel = lv; continue; }
        float qn = h->nrm[i];
        DI cur = { dist(h, q, qn, h->entry), (int32_t)h->entry };
        for (int l = h->maxlevel; l > lv; l--) {
            for (int changed = 1; changed;) {
                changed = 0;
                int32_t *ls = lst(h, cur.id, l);
                for (int j = 0; j < nbn(h, l) && ls[j] >= 0; j++) {
                    DI e = { dist(h, q, qn, ls[j]), ls[j] };
                    if (lt(e, cur)) { cur = e; changed = 1; }
                }
            }
        }
        for (int l = lv < h->maxlevel ? lv : h->maxlevel; l >= 0; l--) {
            int nw = beam(h, q, qn, cur, l, h->efc, W, C, n1);
            int ns = shrink(h, W, nw, nbn(h, l), sel);
            int32_t *mine = lst(h, i, l);
            for (int s = 0; s < ns; s++) mine[s] = sel[s].id;
            for (int s = 0; s < ns; s++) {
                int32_t e = sel[s].id;
                int32_t *le = lst(h, e, l);
                int cap = nbn(h, l), have = 0;
                while (have < cap && le[have] >= 0) have++;
                if (have < cap) { le[have] = (int32_t)i; continue; }
                const float *xe = h->x + (int64_t)e * h->d;
                int nt = 0;
                for (int j = 0; j < cap; j++) { tmp[nt].id = le[j]; tmp[nt].v = dist(h, xe, h->nrm[e], le[j]); nt++; }
                tmp[nt].id = (int32_t)i; tmp[nt].v = sel[s].v; nt++;
                qsort(tmp, nt, sizeof(DI), cmp_di);
                DI *keep = malloc(sizeof(DI) * (cap + 1));
                int nk = shrink(h, tmp, nt, cap, keep);
                for (int j = 0; j < cap; j++) le[j] = j < nk ? keep[j].id : -1;
                free(keep);
            }
            cur = W[0];
        }
        if (lv > h->maxlevel) { h->maxlevel = lv; h->entry = i; }
    }
    free(W); free(C); free(sel); free(tmp);
}

void hnsw_orc_search(H *h, const float *xq, int64_t nq, int k, float *D, int64_t *I)
{
    int ef = h->efs > k ? h->efs : k;
    DI *W = malloc(sizeof(DI) * (ef + 2)), *C = malloc(sizeof(DI) * (h->n + 8));
    for (int64_t qi = 0; qi < nq; qi++) {
        const float *q = xq + qi * h->d;
        float qn = orc_norm_one(q, h->d);
        int nw = 0;
        if (h->entry >= 0) {
            DI cur = { dist(h, q, qn, h->entry), (int32_t)h->entry };
            for (int l = h->maxlevel; l > 0; l--) {
                for (int changed = 1; changed;) {
                    changed = 0;
                    int32_t *ls = lst(h, cur.id, l);
                    for (int j = 0; j < nbn(h, l) && ls[j] >= 0; j++) {
                        DI e = { dist(h, q, qn, ls[j]), ls[j] };
                        if (lt(e, cur)) { cur = e; changed = 1; }
                    }
                }
            }
            nw = beam(h, q, qn, cur, 0, ef, W, C, h->n);
        }
        for (int j = 0; j < k; j++) {
            if (j < nw) { D[qi * k + j] = h->metric == 0 ? -W[j].v : W[j].v; I[qi * k + j] = W[j].id; }
            else { D[qi * k + j] = h->metric == 0 ? -3.4028235e38f : 3.4028235e38f; I[qi * k + j] = -1; }
        }
    }
    free(W); free(C);
}
