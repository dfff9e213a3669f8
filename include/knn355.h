/*
 * knn355.h -- C ABI of libknn355.so, the MI355X (gfx950) kNN hot path.
 *
 * The reference (konstin/knn-for-homology) has no FFI layer of its own on this
 * path: its boundary is the slice of the `faiss` Python module it calls.  Every
 * entry point below names the faiss call it stands in for and the reference
 * call sites (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, no C++/torch types; all matrices row-major, C-contiguous float32
 *   - "host" functions take host pointers and copy (the caller keeps ownership,
 *     nothing is retained past return); "_dev" functions take device pointers
 *     on the index's device and an optional hipStream_t passed as void*
 *     (NULL = the library's own stream, synchronised before return)
 *   - return 0 on success, negative on error; knn_last_error() gives the
 *     thread-local message.  No exceptions or abort() cross the ABI.
 *   - there is NO CPU fallback: without a usable HIP device every compute
 *     entry point fails with KNN_ERR_NO_DEVICE.
 *   - calls on different handles may run concurrently; calls on one handle
 *     are serialised internally.
 */
#ifndef KNN355_H
#define KNN355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* faiss.METRIC_INNER_PRODUCT / faiss.METRIC_L2 (cath/search.py:14,31-32;
 * seqvec_search/main.py:26) */
#define KNN_METRIC_INNER_PRODUCT 0
#define KNN_METRIC_L2 1

#define KNN_OK 0
#define KNN_ERR_INVALID (-1)
#define KNN_ERR_NO_DEVICE (-2)
#define KNN_ERR_HIP (-3)
#define KNN_ERR_UNSUPPORTED (-4)
#define KNN_ERR_IO (-5)
#define KNN_ERR_TIMEOUT (-6) /* a bounded wait of the multi-GPU path ran out (knn_comm_create, knn_sharded_search_dev) */

/* largest k the fused top-k supports (the reference uses k <= 1000:
 * pfam/proteins_search.py:49) */
#define KNN_MAX_K 2048

typedef struct knn_index_s *knn_handle;

/* ---- library ---------------------------------------------------------- */
const char *knn_last_error(void);
const char *knn_version(void);
/* number of visible HIP devices (0 if none); never initialises a context */
int knn_device_count(void);
/* freed index storage and scratch buffers are kept in a per-device pool (<= 8 GiB) for the
 * next index; knn_trim() returns them to the driver and reports the bytes released */
int64_t knn_trim(void);
/* page-locked host memory (hipHostMalloc) for result arrays: a search result is downloaded into it
 * by one DMA at PCIe line rate; into pageable memory the runtime stages the copy.  The faiss-shaped
 * facade allocates large D / I arrays here (numpy arrays over a small pool).  NULL on failure. */
void *knn_host_alloc(int64_t bytes);
void knn_host_free(void *p);
/* selects the device used by indexes created afterwards by this thread and
 * lazily creates its context (fork safe: nothing happens at load time;
 * cath/compare_seqvec_layer.py:58-64 calls search from forked workers) */
int knn_init(int device);

/* ---- faiss.normalize_L2(x) --------------------------------------------
 * in place; rows with zero norm untouched.  cath/search.py:19,
 * pfam/proteins_search.py:22, seqvec_search/main.py:31,34, pfam/search.py:18,20,
 * pfam/slices/slices_search.py:18 */
int knn_normalize_l2(float *x_host, int64_t n, int32_t d);
int knn_normalize_l2_dev(float *x_dev, int64_t n, int32_t d, void *stream);

/* ---- faiss.IndexFlat(d, metric) ----------------------------------------
 * cath/search.py:20, pfam/proteins_search.py:24, seqvec_search/main.py:35,
 * pfam/search.py:44, pfam/slices/slices_search.py:19 */
int knn_flat_create(int32_t d, int32_t metric, knn_handle *out);
/* index.add(x): appends n rows (ids = insertion order).  cath/search.py:22,
 * pfam/proteins_search.py:37, seqvec_search/main.py:39 */
int knn_flat_add(knn_handle h, const float *x_host, int64_t n);
int knn_flat_add_dev(knn_handle h, const float *x_dev, int64_t n, void *stream);
/* index.search(x, k) -> D float32 [nq,k], I int64 [nq,k], best first;
 * unfilled slots: I=-1, D=-FLT_MAX (IP) / +FLT_MAX (L2).
 * Squared L2 follows FAISS's two formulas [ext: knn_L2sqr, distance_compute_blas_threshold = 20]: a call with
 * fewer than 20 queries returns sum (x - y)^2, a larger one max(0, |x|^2 + |y|^2 - 2<x,y>).
 * cath/search.py:24, pfam/proteins_search.py:49, seqvec_search/main.py:45,
 * pfam/search.py:51, pfam/slices/slices_search.py:28 */
int knn_flat_search(knn_handle h, const float *q_host, int64_t nq, int64_t k, float *D_host,
                    int64_t *I_host);
int knn_flat_search_dev(knn_handle h, const float *q_dev, int64_t nq, int64_t k, float *D_dev,
                        int64_t *I_dev, void *stream);
/* All-vs-all without the round trips: the reference searches the very array it
 * just added (cath/search.py:22-24 index.add(embeddings); index.search(embeddings,
 * hits + 1); pfam/proteins_search.py:37,49).  knn_flat_search_self uses rows
 * [row0, row0 + nrows) of the index, already resident and padded, as the queries --
 * same results as knn_flat_search on those rows, no query upload.
 * knn_flat_normalize_rows L2-normalises the stored rows in place on the device
 * (cath/search.py:17-19 normalises a private copy before add: add the raw rows,
 * normalise them here, and the caller's array stays untouched as in the reference). */
int knn_flat_search_self(knn_handle h, int64_t row0, int64_t nrows, int64_t k, float *D_host,
                         int64_t *I_host);
/* A self-search of the WHOLE index (row0 = 0, nrows = ntotal) multiplies only the score tiles on and
 * above the diagonal: dot(x, y) = dot(y, x) bit for bit, so the tile of (query tile I, database tile J)
 * also serves (query tile J, database tile I) -- half the matrix work, the same bits.  Results of 32 MB
 * or more reach D_host / I_host in groups of query tiles (four; eight from 256 MB on) while the later
 * groups are still being multiplied: the arrays are complete when the call returns, not before.
 * knn_flat_search_self_dev: the same search with the results left on the device (synchronous). */
int knn_flat_search_self_dev(knn_handle h, int64_t k, float *D_dev, int64_t *I_dev);
int knn_flat_normalize_rows(knn_handle h);
/* A second handle on the same device-resident rows with its own stream and scratch
 * memory (read-only: add/reset/reserve/normalize_rows fail on it; it sees the rows present
 * when it was made and must be freed before its parent).  Two searches on a handle and
 * its view overlap on the GPU -- the small launches at either end of one search (seed
 * sample, merges, the multi-GPU all-gather) hide behind the other one's scan.  FAISS's
 * GPU flat index does the same with two streams; the reference's CPU path has no
 * equivalent (seqvec_search/main.py:45 index.search is one blocking call). */
int knn_flat_view(knn_handle parent, knn_handle *out);
/* per-shard result as packed sortable keys (uint64: order-preserving score
 * bits << 32 | id_base + local row), k per query, ascending = best first,
 * padded with UINT64_MAX.  This is what ranks exchange (RCCL all-gather). */
int knn_flat_search_keys_dev(knn_handle h, const float *q_dev, int64_t nq, int64_t k,
                             uint32_t id_base, uint64_t *keys_dev, void *stream);
/* merges nlists key lists per query ([nlists][nq][k], e.g. an all-gather
 * buffer) into final D/I on h's device, in h's metric: ONE selection launch that reads the
 * buffer in place whatever nlists * k is -- no scratch memory, nothing of h is written (h
 * supplies the device and the metric only), so merges issued through different handles or
 * streams cannot interfere */
int knn_merge_keys_dev(knn_handle h, const uint64_t *keys_dev, int32_t nlists, int64_t nq, int64_t k,
                       float *D_dev, int64_t *I_dev, void *stream);
/* pre-sizes the device storage for nrows rows (faiss has no equivalent; avoids
 * regrowth copies when a shard is filled by several add calls) */
int knn_flat_reserve(knn_handle h, int64_t nrows);
/* index.ntotal / index.d / index.metric_type */
int64_t knn_ntotal(knn_handle h);
int32_t knn_dim(knn_handle h);
int32_t knn_metric(knn_handle h);
int32_t knn_device_of(knn_handle h);
/* index.reset() */
int knn_reset(knn_handle h);
/* copies rows [i0, i0+n) back to the host (index.reconstruct_n; used by
 * write_index and by the HNSW builder) */
int knn_flat_reconstruct(knn_handle h, int64_t i0, int64_t n, float *out_host);
void knn_free(knn_handle h);

/* ---- several GPUs without torch: RCCL inside the library -------------------------
 * The reference has no multi-device code; the north star asks for a database
 * row-sharded over the GPUs of a node with one all-gather of the per-shard top-k.  One process
 * per GPU: rank 0 makes a unique id and hands it to the other ranks (a file, MPI, a socket --
 * the caller's business), every rank creates its communicator, and knn_sharded_search_dev runs
 * local scan -> ncclAllGather of the [nq][k] packed keys -> selection on one stream; the result
 * (global ids: id_base + local row) is the same on every rank and does not depend on the number
 * of shards.  librccl is dlopen'ed on first use.  (Python callers can use torch.distributed
 * instead: knn_for_homology_amd.sharded.)
 * One search at a time per communicator: its key / gather buffers are reused, a call on another
 * stream waits (on the GPU) for the previous call's selection.  Two searches in flight need two
 * communicators.  A rank whose local scan fails still enters the all-gather with "no rows" so
 * that its peers do not deadlock, and returns its error: a failure on ANY rank is a failure of
 * the search -- exchange the return codes before trusting a result.
 * Bounded waits: knn_comm_create gives a peer KNN355_COMM_TIMEOUT_S seconds (environment; default 120) to
 * arrive, the synchronous form of knn_sharded_search_dev (stream == NULL) gives the search the same time to
 * complete; both return KNN_ERR_TIMEOUT with the rank named in knn_last_error() instead of hanging, and the
 * communicator is unusable afterwards (free it). */
typedef struct knn_comm_s *knn_comm;
int knn_comm_unique_id(uint8_t *id128);
int knn_comm_create(const uint8_t *id128, int32_t world, int32_t rank, int32_t device, knn_comm *out);
void knn_comm_free(knn_comm c);
int knn_sharded_search_dev(knn_handle h, knn_comm c, const float *q_dev, int64_t nq, int64_t k,
                           uint32_t id_base, float *D_dev, int64_t *I_dev, void *stream);

/* ---- faiss.IndexHNSWFlat(d, M, metric) ------------------------------------
 * pfam/proteins_search.py:27-31 (M = 42, inner product, hnsw.efSearch = 256),
 * .train (no-op) / .add :35-37, .search(x, 1000) :49.  The levels above 0 live on the host;
 * the level-0 lists live on the device (the host copy is brought up to date when the graph is
 * exported or walked on the host).  A search runs on the device: an exact scan of the
 * rows of all nodes above level 0 picks the entry points (knn_hnsw_set_entry; 0 = FAISS's
 * greedy descent through the upper levels, on the host), one wave per query walks level 0
 * with a beam of ef = max(efSearch, k) entries, and the beam's rows are re-scored with the
 * flat search's arithmetic: every returned distance carries the flat index's bits (squared L2:
 * the sum of squared differences at every batch size, as FAISS's HNSW distance computer --
 * the bits a flat search of fewer than 20 queries returns).
 * Construction is batch-synchronous and deterministic; the rows of an add call are linked in a
 * shuffled order (FAISS shuffles too: rows grouped by family must not be inserted blind to their
 * own neighbours).  Level-0 candidates from the same
 * device pipeline and -- since round 4 -- level-0 selection, links, reverse links and pruning
 * on the device as well (the host path's graph, bit for bit; KNN355_HNSW_HOST_LINKS=1 is that
 * path), the candidates of the levels above from exact scans of the coarse index
 * (points whose highest levels that scan cannot fill fall back to host walkers with GPU
 * distances; knn_gather_distances is that offload as a public entry).
 * Results use the flat index's layout; slots the walk could not fill hold id -1. */
typedef struct knn_hnsw_s *knn_hnsw_handle;
int knn_hnsw_create(int32_t d, int32_t M, int32_t metric, knn_hnsw_handle *out);
/* index.hnsw.efSearch / index.hnsw.efConstruction (values <= 0 leave the setting alone) */
int knn_hnsw_set_ef(knn_hnsw_handle h, int32_t efSearch, int32_t efConstruction);
/* walk tuning (not in faiss): candidates expanded per walker per lock-step round (default 8;
 * 1 = strict best-first) and walkers / inserted rows per batch (default 16384); values <= 0 keep the setting */
int knn_hnsw_set_walk(knn_hnsw_handle h, int32_t expand, int32_t max_batch);
/* where the level-0 walk of a search starts (not in faiss): coarse_entries > 0 -- at the coarse_entries
 * nearest nodes above level 0, found by an exact scan of those rows on the flat kernel (default 4);
 * 0 -- where FAISS's greedy descent through the upper levels ends */
int knn_hnsw_set_entry(knn_hnsw_handle h, int32_t coarse_entries);
int knn_hnsw_get_params(knn_hnsw_handle h, int32_t *M, int32_t *efSearch, int32_t *efConstruction,
                        int32_t *max_level, int64_t *entry_point);
int knn_hnsw_add(knn_hnsw_handle h, const float *x_host, int64_t n);
/* the same for rows already on the index's device ([n][d] contiguous): a graph over a database that
 * never existed in host memory */
int knn_hnsw_add_dev(knn_hnsw_handle h, const float *x_dev, int64_t n, void *stream);
int knn_hnsw_search(knn_hnsw_handle h, const float *q_host, int64_t nq, int64_t k, float *D_host,
                    int64_t *I_host);
int64_t knn_hnsw_ntotal(knn_hnsw_handle h);
/* the flat storage underneath (index.storage); owned by the HNSW handle */
knn_handle knn_hnsw_storage(knn_hnsw_handle h);
void knn_hnsw_free(knn_hnsw_handle h);
/* graph tables for write_index / read_index: levels[n], offsets[n+1], nbrs[nslots]
 * (-1 = empty), cum_nb[nlevels_tab] (slots below each level), assign_probas */
int knn_hnsw_graph_sizes(knn_hnsw_handle h, int64_t *ntotal, int64_t *nslots, int32_t *nlevels_tab);
int knn_hnsw_graph_export(knn_hnsw_handle h, int32_t *levels, int64_t *offsets, int32_t *nbrs,
                          int32_t *cum_nb, double *assign_probas);
int knn_hnsw_graph_import(knn_hnsw_handle h, int64_t ntotal, const int32_t *levels, const int32_t *nbrs,
                          int64_t nslots, int32_t max_level, int64_t entry_point);
/* work counters since the last reset: (query,row) pairs evaluated, lock-step rounds,
 * pruned neighbour lists, seconds spent on GPU round trips and on the host walk */
int knn_hnsw_stats(knn_hnsw_handle h, int64_t *pairs, int64_t *rounds, int64_t *shrinks, double *gpu_s,
                   double *host_s, int32_t reset);

/* ---- faiss.IndexLSH(d, nbits) -----------------------------------------------
 * seqvec_search/create_index.py:41-45, pfam/search.py:27-37, pfam/proteins_search.py:25-26.
 * rotation_host: [nbits][d] row-major projection rows (FAISS uses a random orthonormal
 * matrix from its own RNG; the caller supplies one).  Codes: bit j = (x . rot_j >= 0);
 * search returns the k smallest Hamming distances as float32, ties by lower id. */
typedef struct knn_lsh_s *knn_lsh_handle;
int knn_lsh_create(int32_t d, int32_t nbits, const float *rotation_host, knn_lsh_handle *out);
int knn_lsh_add(knn_lsh_handle h, const float *x_host, int64_t n);
int knn_lsh_search(knn_lsh_handle h, const float *q_host, int64_t nq, int64_t k, float *D_host,
                   int64_t *I_host);
int64_t knn_lsh_ntotal(knn_lsh_handle h);
int32_t knn_lsh_code_words(knn_lsh_handle h);
/* codes in FAISS byte order ([ntotal][bytes_per_vec], bit i -> byte i>>3, bit i&7) */
int knn_lsh_get_codes(knn_lsh_handle h, uint8_t *out_host, int32_t bytes_per_vec);
int knn_lsh_add_codes(knn_lsh_handle h, const uint8_t *codes_host, int64_t n, int32_t bytes_per_vec);
void knn_lsh_free(knn_lsh_handle h);

/* ---- consumers of (hits, scores): SURVEY section 8(f) N4 --------------------------
 * Host buffers in and out.  hits are int64 [nq][k] as returned by search. */
/* pfam/proteins.py:85-122 remove_self_hit: drops the self id from each row (or the last
 * hit when the row does not contain it: missing_out[r] = 1); outputs are [nq][k-1] */
int knn_eval_remove_self_hit(const int64_t *hits, const float *scores, int64_t nq, int64_t k,
                             const int64_t *self_ids, int64_t *hits_out, float *scores_out,
                             int32_t *missing_out);
/* seqvec_search/main.py:64-82 evaluate: lead_out[r] = hits of the query's label before the
 * first foreign hit, tp_out[r] = hits of the query's label anywhere; is_correct_out
 * (uint8 [nq][k], may be NULL) is the per-hit match matrix of seqvec_search/tp_cumulative.py */
int knn_eval_labels(const int64_t *hits, int64_t nq, int64_t k, const int32_t *labels_q,
                    const int32_t *labels_db, int64_t nb, uint8_t *is_correct_out, int32_t *lead_out,
                    int32_t *tp_out);
/* pfam/proteins_shared.py:139-157 compute_auc1: query r is homologous to the SORTED target
 * rows set_members[set_offsets[r] .. set_offsets[r+1]) */
int knn_eval_sets(const int64_t *hits, int64_t nq, int64_t k, const int64_t *set_offsets,
                  const int64_t *set_members, int32_t *lead_out, int32_t *tp_out);
/* cath/cath.py:76-84 compute_is_correct: out uint8 [nq][nlevels][k],
 * out[q][l][j] = mapping[query_rows[q]][l] == mapping[hits[q][j]][l]; mapping int32 [n][nlevels] */
int knn_eval_levels(const int64_t *hits, int64_t nq, int64_t k, const int64_t *query_rows,
                    const int32_t *mapping, int64_t n, int32_t nlevels, uint8_t *out);

/* ---- MMseqs2 prefilter database: SURVEY section 8(f) N3 ----------------------------
 * seqvec_search/mmseqs/_write_prefilter_db.py:52-97 write_prefilter_db: data file
 * ("<prefilter>.0") and index file ("<prefilter>.index"); queries[i] is the faiss row of
 * query i, *_map translate faiss rows to MMseqs2 ids.  clip: 0 = the reference's clip=False
 * (scores * 100 in float32); 1 = clip=True as the reference's pinned numpy 1.22 evaluates
 * numpy.clip(scores, -(10**30), 10**30) * 100 (in double); 2 = clip=True under numpy >= 2
 * (NEP 50: the expression stays float32). */
int knn_write_prefilter_db(const char *data_path, const char *index_path, const int64_t *hits,
                           const float *scores, int64_t nq, int64_t k, const int64_t *queries,
                           const int64_t *test_map, int64_t n_test, const int64_t *train_map,
                           int64_t n_train, int32_t clip);

/* ---- distances for explicit candidate lists (HNSW walk offload) --------
 * For query i (row of q_dev [nq,d]) and candidates cand[off[i] .. off[i+1]),
 * out[p] = <q,y> (IP) or max(0, |q|^2+|y|^2-2<q,y>) (L2), same arithmetic as
 * the flat scan.  Replaces the per-candidate distance computer inside
 * faiss.IndexHNSWFlat (pfam/proteins_search.py:30-31,49). */
int knn_gather_distances(knn_handle h, const float *q_host, int64_t nq, const int64_t *cand_ids,
                         const int64_t *cand_offsets, float *out_host);

/* ---- tuning / introspection (not part of the faiss surface) ------------ */
/* name of the scan kernel the last search on this handle dispatched, and its
 * geometry; used by bench.py for the roofline line */
int knn_last_scan_info(knn_handle h, char *name, int32_t name_len, int32_t *query_tile,
                       int32_t *db_tile, int32_t *nchunks, int32_t *grid);
/* device time (ms) of the scan kernel launches of the last search, measured
 * with hipEvents on the launch stream */
float knn_last_scan_ms(knn_handle h);
/* durations (ms) of the most recent scan launches (oldest first, at most 64 and at
 * most max_n), each from a hipEvent pair recorded around the launch on the stream it
 * was launched on; -1 for a launch that has not finished.  Returns the count. */
int32_t knn_scan_times(knn_handle h, float *out_ms, int32_t max_n);
/* how the last search on this handle was seeded: seed_stride = 0 (no seed sample), the
 * stride of the sample searched first, or -r (tile-minimum seed: no sample pass, every chunk of the
 * scan publishes r keys per query -- the best key of each of its first r tiles (r = 1, 2) or, for a k beyond
 * what that supports, the best key of each wave's rows of the first tile (r = 4 with the 32-query tile, 2 with the
 * 64-query tile; 4 with the 48-query tile) -- and the k-th smallest published key is the bound); stat_rank = 0 (the sample's k-th score, a proven bound)
 * or j (statistical seed: the sample's j-th score, result verified); stat_redo = searches
 * repeated so far because a statistical threshold failed its verification; sample_rows =
 * rows scanned by the sample pass (the main scan kernel skips them) */
int knn_last_seed_info(knn_handle h, int32_t *seed_stride, int32_t *stat_rank, int64_t *stat_redo,
                       int64_t *sample_rows);
/* force a scan configuration: query_tile in {0(auto),32,48,64,96,128,256} (48 and 96: the builds on 16-query MFMA blocks, one
 * query tile per launch -- honoured when the batch fits the tile, ignored otherwise); nchunks 0=auto (a forced count also turns the paired
 * walk off); flags, all off by default:
 *      2  no shared pool of tiles in a paired (one-query-tile) launch
 *      4  never pair the workgroups of a one-query-tile launch: static chunks instead
 *      8  no seeding at all (every chunk warms its thresholds up on its own)
 *     16  force the exact seed (a sample pass of its own in front of the scan)
 *     32  squared L2 by the norm formula |x|^2 + |y|^2 - 2<x,y> whatever the batch size (default: FAISS's rule --
 *         batches of fewer than 20 queries use the sum of squared differences, larger ones the norm formula)
 *     64  always launch the state-reset kernel in front of a streaming scan (default: the previous search's final
 *         selection leaves the state reset when the next search has the same shape)
 *    128  force the statistical seed (synchronous entry points only)
 *    256  the two workgroups of a CU do not take turns in their K loops (batch launches)
 *    512  never use the statistical seed
 *   1024  never use the symmetric launch of a whole-index self-search
 *   2048  never use the tile-minimum seed (a streaming search then runs its seed sample as a launch of its own)
 *   bits 12-13  publication rounds of the tile-minimum seed (0 = the library's choice)
 *  16384  never search the remainder behind the full 128-query tiles as a piece of its own (large databases: 129 queries
 *         are one 128-query launch and one streaming launch instead of two 128-query passes)
 * 131072  plans without the 48- and 96-query tiles (33..48 queries then pay for a 64-query tile, 65..96 for 128 or two pieces)
 * 262144  never the 256 x 256 tile (one workgroup per CU; the library's choice for synchronous searches of >= 256 queries over
 *         >= 65 536 rows with long chunks, always under the statistical seed)
 * 524288  the 256 x 256 tile wherever a batch holds more than 128 queries and the index >= 1024 rows, and 256-row tiles in the
 *         symmetric whole-index self-search (tests, A/B)
 * Only 32 changes what a search returns (the other formula's rounding); every other combination returns the same bits. */
int knn_set_tuning(knn_handle h, int32_t query_tile, int32_t nchunks, int32_t flags);

/* A caller that hands one batch of queries over in pieces (its own blocks, one slice per GPU) says how large the
 * whole batch is: FAISS chooses the squared-L2 formula by the batch ITS caller passed to index.search
 * (/root/reference/seqvec_search/main.py:45; fewer than 20 queries: the sum of squared differences), so pieces of
 * fewer than 20 queries of a larger batch keep the norm formula and the pieces return the bits of the one call.
 * nq_whole = 0 (default): every call is its own batch.  The setting stays until changed; it is per handle (views made
 * with knn_flat_view have their own). */
int knn_flat_set_batch(knn_handle h, int64_t nq_whole);

/* Measurement aid (bench.py's roofline): what THIS box's HBM delivers to a plain read kernel over the index's own rows
 * -- grid-stride 16-byte loads, nothing else, the best of `reps` launches at three grid sizes.  best_ms = its duration,
 * bytes_read = ntotal x padded row bytes.  A flat scan cannot be faster than this; boxes of one pool differ by a few
 * percent, the 8 TB/s of the data sheet is not reached by any kernel. */
int knn_flat_read_rate(knn_handle h, int32_t reps, float *best_ms, int64_t *bytes_read);

/* Measurement aid (bench.py's batch roofline): what THIS box's fp32 matrix pipes sustain -- every wave issues
 * back-to-back v_mfma_f32_32x32x2_f32 on independent accumulators, operands in registers (random values), two
 * workgroups per CU, no memory traffic.  Launches run back to back for warm_ms milliseconds first (the chip settles on
 * the clock it holds under this load), then the best of eight is reported: tflops, and clock_mhz (may be NULL) = the
 * in-kernel shader clock of that launch.  The data sheet's dense fp32 MFMA peak is 157.3 TFLOP/s at 2.4 GHz. */
int knn_mfma_rate(int32_t warm_ms, float *tflops, float *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* KNN355_H */
