// knn355.hip -- MI355X (gfx950 / CDNA4) flat kNN: distance + fused top-k.
//
// Stands in for faiss-cpu 1.7.2's IndexFlat / normalize_L2 as called by the
// reference (cath/search.py:13-26, seqvec_search/main.py:22-50,
// pfam/proteins_search.py:21-50, pfam/search.py:42-53,
// pfam/slices/slices_search.py:15-31).  C ABI: include/knn355.h.
//
// Arithmetic contract (bit-exact with oracle/knn_oracle.c):
//   dot(x,y)  one fp32 fma chain per (query,row) pair, k visited per 8-block in
//             the order 0,4,1,5,2,6,3,7 -- exactly what v_mfma_f32_32x32x2_f32
//             produces when lane (i,h) feeds floats 8t+4h+m for m = 0..3
//   nrm(x)    dot(x, x): the same chain, so L2(x, x) == 0 exactly
//   L2        max(0, fma(-2, dot, nrm(x)+nrm(y)))
//   ties      lower row id first (packed 64-bit keys: score bits << 32 | row)
//
// Kernel plan (DESIGN.md has the long form):
//   flat_scan_kernel     one workgroup = QT queries x one chunk of database rows.  Per
//                        32-float K step both operands are staged into LDS with
//                        global_load_lds_dwordx4 (full 128-B lines, XOR-swizzled on the
//                        source address), fragments come back with ds_read_b128, fp32 MFMA
//                        accumulates the QT x DT score tile; the epilogue compares every
//                        score with a running per-query threshold and appends the survivors
//                        as packed keys; lists are cut back by wave_select (one wave per
//                        query, registers only).
//   merge_select_kernel  one wave per (query, group of lists): exact top-k of <= 4096 keys,
//                        last round sorts and writes D / I.
//   hnsw.inc / lsh.inc / eval.inc   IndexHNSWFlat, IndexLSH, consumers of (hits, scores).
#include <hip/hip_runtime.h>
#include <thread>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <float.h>
#include <math.h>
#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <memory>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/knn355.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) f32x4 *gptr4; // explicit global (not flat) loads
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define KEY_PAD 0xFFFFFFFFFFFFFFFFull
#ifndef KNN355_DIFF_SCALAR_Q
#define KNN355_DIFF_SCALAR_Q 1 // difference builds read their query fragments with scalar loads (0: broadcast LDS reads, round-4 first form)
#endif
#ifndef KNN355_DIFF_SCALAR_PAIRS
#define KNN355_DIFF_SCALAR_PAIRS 3 // ... the first three query pairs of a thread's half (all of them up to 12 queries), the others through the LDS
#endif
#ifndef KNN355_EPI_PRIO
#define KNN355_EPI_PRIO 0 // issue priority of a wave behind its K loop (two workgroups per CU).  Measured with 2 against 0 on one box: CATH-sized
                          // symmetric kernel 2.21-2.22 vs 2.25 ms, 10 M-row step 6.874 vs 6.886 ms, shard step 0.916 vs 0.914 ms -- nothing: off
#endif
#ifndef KNN355_STREAM_DMA_FIRST
#define KNN355_STREAM_DMA_FIRST 1 // streaming launches issue a K step's staging instructions in front of its first MFMA (flat_scan_kernel)
#endif

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// nrm(x) = dot(x, x) in the contract's chain order: one lane walks one row.
template <bool ALIGNED>
__device__ __forceinline__ float lane_norm_row(const float *__restrict__ r, int d)
{
    float acc = 0.0f;
    int k0 = 0;
    for (; k0 + 8 <= d; k0 += 8) {
        float a[8];
        if constexpr (ALIGNED) {
            f32x4 lo = *(const f32x4 *)(r + k0), hi = *(const f32x4 *)(r + k0 + 4);
#pragma unroll
            for (int m = 0; m < 4; m++) { a[m] = lo[m]; a[4 + m] = hi[m]; }
        } else {
#pragma unroll
            for (int m = 0; m < 8; m++) a[m] = r[k0 + m];
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc = __builtin_fmaf(a[m], a[m], acc);
            acc = __builtin_fmaf(a[4 + m], a[4 + m], acc);
        }
    }
    if (k0 < d) {
        float a[8];
#pragma unroll
        for (int m = 0; m < 8; m++) a[m] = (k0 + m < d) ? r[k0 + m] : 0.0f;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc = __builtin_fmaf(a[m], a[m], acc);
            acc = __builtin_fmaf(a[4 + m], a[4 + m], acc);
        }
    }
    return acc;
}

// one lane per row; 64-thread blocks
template <bool ALIGNED>
__global__ __launch_bounds__(64) void norm_rows_kernel(const float *__restrict__ x, int64_t n, int d, int64_t stride,
                                                        float *__restrict__ out)
{
    int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= n) return;
    out[row] = lane_norm_row<ALIGNED>(x + row * stride, d);
}

// faiss.normalize_L2: each lane computes the norm of its own row, then the wave
// rescales the block's 64 rows one after the other with coalesced accesses.
template <bool ALIGNED>
__global__ __launch_bounds__(64) void normalize_rows_kernel(float *__restrict__ x, int64_t n, int d, int64_t stride)
{
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int64_t row = row0 + lane;
    float inv = 0.0f; // 0 == leave the row alone
    if (row < n) {
        float nr = lane_norm_row<ALIGNED>(x + row * stride, d);
        if (nr > 0.0f) inv = (float)(1.0 / (double)sqrtf(nr));
    }
    const int nrows = (int)min((int64_t)64, n - row0);
    for (int r = 0; r < nrows; r++) {
        const float s = __shfl(inv, r, 64);
        if (s == 0.0f) continue;
        float *p = x + (row0 + r) * stride;
        if constexpr (ALIGNED) {
            for (int e = 4 * lane; e < d; e += 256) {
                if (e + 3 < d) {
                    f32x4 v = *(f32x4 *)(p + e);
                    v[0] *= s; v[1] *= s; v[2] *= s; v[3] *= s;
                    *(f32x4 *)(p + e) = v;
                } else {
                    for (int c = 0; c < 3; c++)
                        if (e + c < d) p[e + c] *= s;
                }
            }
        } else {
            for (int e = lane; e < d; e += 64) p[e] *= s;
        }
    }
}

// dst[n][dp] <- src[n][d], zero padded
__global__ void pad_rows_kernel(const float *__restrict__ src, int64_t n, int d,
                                float *__restrict__ dst, int dp)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * dp;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / dp;
        int c = (int)(i - r * dp);
        dst[i] = c < d ? src[r * d + c] : 0.0f;
    }
}

// dst[n][dp] (bf16) <- src[n][dp] (fp32), round to nearest even: the operand copies of an approximate (bf16) index
__global__ void to_bf16_kernel(const float *__restrict__ src, int64_t total, __bf16 *__restrict__ dst)
{
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *(const f32x4 *)(src + i); // (total is a multiple of 64)
        dst[i] = (__bf16)v[0];
        dst[i + 1] = (__bf16)v[1];
        dst[i + 2] = (__bf16)v[2];
        dst[i + 3] = (__bf16)v[3];
    }
}

// The queries of a difference-build scan (fewer than 20 of them), interleaved by pairs so that one 16-byte LDS read hands a
// lane two consecutive k of BOTH queries of a pair as two aligned register pairs (flat_scan_kernel<..., DNQ>):
//   out[2p    ][32 s + i] = query (2p + (i & 1)) [32 s + i / 2]        (k = 0..15 of K step s)
//   out[2p + 1][32 s + i] = query (2p + (i & 1)) [32 s + 16 + i / 2]   (k = 16..31)
// for p < dnq / 2; queries past the last one repeat it (their scores are never looked at).
__global__ void diff_interleave_kernel(const float *__restrict__ xq, int64_t nq, int dp, int dnq, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)dnq * dp) return;
    const int row = (int)(i / dp), col = (int)(i - (int64_t)row * dp);
    const int pr = row >> 1, half = row & 1, step = col >> 5, w = col & 31;
    const int64_t q = min((int64_t)(2 * pr + (w & 1)), nq - 1);
    out[i] = xq[q * dp + 32 * step + 16 * half + (w >> 1)];
}

// ---------------------------------------------------------------------------
// workgroup-wide bitonic sort of P (power of two) u64 keys in LDS, ascending
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wg_bitonic_sort(uint64_t *sb, int P, int tid, int nthreads)
{
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += nthreads) {
                int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                int b = a + j;
                bool up = (a & k2) == 0;
                uint64_t x = sb[a], y = sb[b];
                if ((x > y) == up) { sb[a] = y; sb[b] = x; }
            }
            __syncthreads();
        }
    }
}

// One wave, P = 64 E keys (E = 1, 2, 4, 8, 16): the same sort in registers.  Lane l holds elements l E .. l E + E - 1, so a
// compare-exchange at distance j < E stays inside the lane (register indices and, for k2 <= E, directions known at compile
// time) and one at distance j >= E meets lane l ^ (j / E) in the same register: two cross-lane moves per key instead of two
// LDS reads, two LDS writes and a barrier per pair.  (The LDS sort of 512 keys costs a wave 720 eight-byte LDS operations;
// with 56 one-wave selections per CU -- a CATH-sized all-vs-all -- that was 67 us of a 219 us kernel on the LDS pipe alone.)
// value of lane (l ^ M) without the LDS crossbar where the vector pipe can do it: DPP inside a row of 16 lanes (quad
// permutations for 1 and 2; 4 = "mirror the half row" (l ^ 7) then "reverse the quad" (l ^ 3); 8 = rotate the row by 8),
// v_permlane16_swap / v_permlane32_swap (gfx950) across rows and wave halves
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane)
{
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
    else if constexpr (M == 4) return (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
    else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);
    else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); // r[0]: odd rows <- the even rows below them; r[1]: even rows <- the odd rows above
        return (lane & 16) ? r[0] : r[1];
    } else {
        static_assert(M == 32, "lane distance");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

// one stage of the network: compare-exchange at distance J inside runs of K2 (element index = lane E + register)
template <int E, int K2, int J>
__device__ __forceinline__ void bitonic_stage_regs(uint64_t (&v)[E], int lane)
{
    if constexpr (J >= E) {
        // (K2 >= 2 J >= 2 E here: the direction depends on the lane only; the last merge, K2 = 64 E, ascends everywhere)
        const bool up = K2 >= 64 * E || (lane & (K2 / E)) == 0;
        const bool keep_min = ((lane & (J / E)) == 0) == up;
#pragma unroll
        for (int r = 0; r < E; r++) {
            const uint32_t ohi = lane_xor<J / E>((uint32_t)(v[r] >> 32), lane);
            const uint32_t olo = lane_xor<J / E>((uint32_t)v[r], lane);
            const uint64_t o = ((uint64_t)ohi << 32) | olo;
            v[r] = ((o < v[r]) == keep_min) ? o : v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < E; r++) {
            if ((r & J) == 0) {
                const uint64_t a = v[r], b = v[r | J];
                bool up; // bit K2 of the element index is a bit of the register index when K2 < E, of the lane otherwise
                if (K2 < E) up = (r & K2) == 0;
                else up = K2 >= 64 * E || (lane & (K2 / E)) == 0;
                const bool sw = (a > b) == up;
                v[r] = sw ? b : a;
                v[r | J] = sw ? a : b;
            }
        }
    }
    if constexpr (J > 1) bitonic_stage_regs<E, K2, J / 2>(v, lane);
    else if constexpr (K2 < 64 * E) bitonic_stage_regs<E, 2 * K2, K2>(v, lane);
}

template <int E>
__device__ __forceinline__ void wave_bitonic_sort_regs(uint64_t *sb, int lane)
{
    uint64_t v[E];
#pragma unroll
    for (int r = 0; r < E; r++) v[r] = sb[lane * E + r];
    bitonic_stage_regs<E, 2, 1>(v, lane);
#pragma unroll
    for (int r = 0; r < E; r++) sb[lane * E + r] = v[r];
}

__device__ __forceinline__ int next_pow2_dev(int n)
{
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------
// wave-wide selection over one candidate list (n <= 64*R packed keys in global
// memory, pulled into registers).  Keeps the best keys -- exactly k when
// kmax == k, otherwise any count in [k, kmax] -- and writes them packed to dst
// (dst may alias lst).  No LDS, no barriers: four waves of a workgroup select
// for four different queries at once.
//   *thr_ord receives T with count(hi <= T) >= k, a valid admission threshold.
//   dst may be NULL: only T is wanted (the tile-minimum seed of the scan kernel).
// Method: MSB-first binary search on the 32-bit score word ("largest P with
// count(hi < P) <= k-1"), each probe one v_cmp + s_bcnt1 per register, started
// below the bits all keys share and stopped as soon as a probe leaves between k
// and kmax survivors; ties that straddle the cut are broken on the id word.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int x)
{
    // DPP butterfly inside each row of 16 lanes, then row broadcasts; lane 63 ends with the total
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false); // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false); // row_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(x, 63);
}

typedef const __attribute__((address_space(1))) uint64_t *gptr_u64; // explicit global (not flat) loads
struct LoadContig {
    const uint64_t *p;
    __device__ __forceinline__ uint64_t operator()(int idx) const { return ((gptr_u64)p)[idx]; }
};
// keys other workgroups of the SAME launch are still publishing (agent scope: served from memory, never from a
// stale line of this XCD's L2)
struct LoadAgent {
    const uint64_t *p;
    __device__ __forceinline__ uint64_t operator()(int idx) const { return __hip_atomic_load(p + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
// element idx of the concatenation of lists l0.. of query q inside a [list][query][k] buffer
struct LoadListMajor {
    const uint64_t *base;
    int64_t nq, q;
    int k, l0;
    __device__ __forceinline__ uint64_t operator()(int idx) const
    {
        const int l = idx / k, j = idx - l * k;
        return ((gptr_u64)base)[((size_t)(l0 + l) * nq + q) * k + j];
    }
};

// min / max over the wave and inclusive prefix sum, all with DPP row operations (no LDS
// crossbar round trips): butterfly inside rows of 16 lanes, then row broadcasts
#define DPP_U32(x, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp((int)(x), (int)(x), ctrl, rmask, 0xF, false))
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    x = min(x, DPP_U32(x, 0xB1, 0xF));
    x = min(x, DPP_U32(x, 0x4E, 0xF));
    x = min(x, DPP_U32(x, 0x141, 0xF));
    x = min(x, DPP_U32(x, 0x140, 0xF));
    x = min(x, DPP_U32(x, 0x142, 0xA));
    x = min(x, DPP_U32(x, 0x143, 0xC));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x)
{
    x = max(x, DPP_U32(x, 0xB1, 0xF));
    x = max(x, DPP_U32(x, 0x4E, 0xF));
    x = max(x, DPP_U32(x, 0x141, 0xF));
    x = max(x, DPP_U32(x, 0x140, 0xF));
    x = max(x, DPP_U32(x, 0x142, 0xA));
    x = max(x, DPP_U32(x, 0x143, 0xC));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2,3
    return x;
}

template <int R, typename LD>
__device__ __attribute__((noinline)) int wave_select(LD load, int n, int k, int kmax, int lane, uint32_t *thr_ord, uint64_t *dst)
{
    static_assert(R <= 64, "at most 4096 keys per wave");
    // wave-uniform by contract; arguments of a non-inlined function arrive in VGPRs and the
    // compiler would otherwise run the search loop on lane masks
    n = __builtin_amdgcn_readfirstlane(n);
    k = __builtin_amdgcn_readfirstlane(k);
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    uint32_t hi[R], lo[R];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int idx = r * 64 + lane;
        const bool valid = idx < n;
        uint64_t key = load(valid ? idx : n - 1); // unconditional: loads go out back to back
        if (!valid) key = KEY_PAD;
        hi[r] = (uint32_t)(key >> 32);
        lo[r] = (uint32_t)key;
        mn = min(mn, hi[r]);              // padding reads as the largest word
        mx = max(mx, valid ? hi[r] : 0u);
        // groups of 16 loads: keeps the address registers of at most 16 loads alive at a time
        if ((r & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    // per-lane partial counts, one DPP reduction per probe.  The compare/add-carry pair is
    // spelled out: left to the scheduler, R compares are hoisted together and their R lane
    // masks spill out of the SGPR file.
    auto count_lt = [&](uint32_t X) {
        int c = 0;
#pragma unroll
        for (int r = 0; r < R; r++)
            asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(c) : "v"(hi[r]), "v"(X) : "vcc");
        return wave_sum(c);
    };
    uint32_t T = mx;
    int cnt = n;
    if (mx == 0xFFFFFFFFu && mn != mx) {
        // padded input (merge of per-chunk lists, mostly padding after a seeded scan): if no
        // more than kmax real keys exist they all survive and there is nothing to search;
        // otherwise search only the bit range of the real keys
        const int real = count_lt(0xFFFFFFFFu);
        if (real <= kmax) {
            mn = mx; // skip the search: T = 0xFFFFFFFF keeps every real key
        } else {
            uint32_t mr = 0u;
#pragma unroll
            for (int r = 0; r < R; r++) mr = max(mr, hi[r] == 0xFFFFFFFFu ? 0u : hi[r]);
            mx = wave_max_u32(mr);
            T = mx;
            cnt = real;
        }
    }
    if (mn != mx && cnt > kmax) {
        // Bracket search on the score word for an X with k <= #(keys < X) <= kmax.  f(X) = #(keys < X)
        // is monotone, f(mn) = 0 < k and f(mx + 1) = cnt > kmax.  Probes alternate between linear
        // interpolation of the target rank inside the bracket (scores between mn and mx are close
        // to evenly spread in the order-preserving integer image, so 3-5 probes usually do where
        // an MSB-first bit search spends 15-25, most of them on the empty leading bits) and
        // bisection (so the bracket at least halves every second probe).  A bracket of width one
        // means the k-th key ties with its neighbours on the score word: T is that word exactly and
        // the id tie-break below finishes the job.
        uint32_t lo_x = mn, hi_x = mx + 1u; // mx < 0xFFFFFFFF here (padding was split off above)
        int f_lo = 0, f_hi = cnt;
        const float target = 0.5f * (float)(k + kmax);
        bool interpolate = true, done = false;
        while (hi_x - lo_x > 1u) {
            uint32_t X;
            if (interpolate)
                X = lo_x + (uint32_t)((float)(hi_x - lo_x) * ((target - (float)f_lo) / (float)(f_hi - f_lo)));
            else
                X = lo_x + ((hi_x - lo_x) >> 1);
            X = max(lo_x + 1u, min(X, hi_x - 1u));
            interpolate = !interpolate;
            const int c = count_lt(X);
            if (c < k) {
                lo_x = X;
                f_lo = c;
            } else if (c > kmax) {
                hi_x = X;
                f_hi = c;
            } else {
                T = X - 1u;
                cnt = c;
                done = true;
                break;
            }
        }
        if (!done) { // lo_x is the exact k-th smallest score word
            T = lo_x;
            cnt = f_hi;
        }
    }
    uint32_t Q = 0xFFFFFFFFu;
    const uint32_t Tlt = T; // keys with hi < Tlt are kept unconditionally, hi == T only with lo <= Q
    if (cnt > kmax && T != 0xFFFFFFFFu) { // (padding keys are interchangeable: the bounded store below trims them)
        // more keys tie at T than may be kept: lowest ids win (rare: exact duplicates)
        const int need = k - count_lt(T);
        Q = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t Qt = Q | (1u << bit);
            int c = 0;
#pragma unroll
            for (int r = 0; r < R; r++)
                c += (hi[r] == T && lo[r] < Qt) ? 1 : 0;
            if (wave_sum(c) <= need - 1) Q = Qt;
        }
    }
    // survivors: lane-major packing (order inside a list is irrelevant).  When the cut falls
    // into the padding (fewer than k real keys) only real keys survive; the caller pads.
    int total;
    if (Q == 0xFFFFFFFFu) {
        // common case, no tie-break: keep hi < Tk
        const uint32_t Tk = T == 0xFFFFFFFFu ? T : T + 1u;
        int mine = 0;
#pragma unroll
        for (int r = 0; r < R; r++)
            asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(mine) : "v"(hi[r]), "v"(Tk) : "vcc");
        const int incl = wave_inclusive_scan(mine);
        int pos = incl - mine;
        total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (hi[r] < Tk) {
                if (pos < kmax && dst) dst[pos] = ((uint64_t)hi[r] << 32) | lo[r];
                pos++;
            }
        }
    } else {
        int mine = 0;
#pragma unroll
        for (int r = 0; r < R; r++) mine += (hi[r] < Tlt || (hi[r] == T && lo[r] <= Q)) ? 1 : 0;
        const int incl = wave_inclusive_scan(mine);
        int pos = incl - mine;
        total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (hi[r] < Tlt || (hi[r] == T && lo[r] <= Q)) {
                if (pos < kmax && dst) dst[pos] = ((uint64_t)hi[r] << 32) | lo[r];
                pos++;
            }
        }
    }
    *thr_ord = T;
    return min(total, kmax);
}

template <typename LD>
__device__ __forceinline__ int wave_select_dispatch(int R, LD load, int n, int k, int kmax, int lane,
                                                    uint32_t *thr_ord, uint64_t *dst)
{
    if (R <= 8) return wave_select<8, LD>(load, n, k, kmax, lane, thr_ord, dst);
    if (R <= 16) return wave_select<16, LD>(load, n, k, kmax, lane, thr_ord, dst);
    return wave_select<32, LD>(load, n, k, kmax, lane, thr_ord, dst); // callers guarantee n <= 2048
}

// The merge kernel may also use 64 keys per lane (4096 keys): it is its own kernel, so the
// 250 registers of that variant do not cost the scan kernels their second wave per SIMD
// (a non-inlined callee's registers count for every kernel that can reach it).
template <int RMAX, typename LD>
__device__ __forceinline__ int wave_select_dispatch_max(int R, LD load, int n, int k, int kmax, int lane,
                                                        uint32_t *thr_ord, uint64_t *dst)
{
    if constexpr (RMAX <= 16) {
        if (R <= 8) return wave_select<8, LD>(load, n, k, kmax, lane, thr_ord, dst);
        return wave_select<16, LD>(load, n, k, kmax, lane, thr_ord, dst);
    } else if constexpr (RMAX <= 32) {
        return wave_select_dispatch(R, load, n, k, kmax, lane, thr_ord, dst);
    } else {
        if (R <= 32) return wave_select_dispatch(R, load, n, k, kmax, lane, thr_ord, dst);
        return wave_select<64, LD>(load, n, k, kmax, lane, thr_ord, dst);
    }
}

// Keeps the keys of a list whose score word is <= T (packed to the front, in place), one wave, registers only.
// Used at the end of a chunk of a tile-minimum-seeded scan: the first tiles were appended before any bound existed,
// by now the shared threshold is tight and a compare does what a selection would.
template <int R>
__device__ __attribute__((noinline)) int wave_filter(uint64_t *lst, int n, uint32_t T, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n);
    T = __builtin_amdgcn_readfirstlane(T);
    uint64_t key[R];
    int mine = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int idx = r * 64 + lane;
        key[r] = idx < n ? ((gptr_u64)lst)[idx] : KEY_PAD;
        if ((r & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < R; r++) mine += (key[r] != KEY_PAD && (uint32_t)(key[r] >> 32) <= T) ? 1 : 0;
    const int incl = wave_inclusive_scan(mine);
    int pos = incl - mine;
#pragma unroll
    for (int r = 0; r < R; r++)
        if (key[r] != KEY_PAD && (uint32_t)(key[r] >> 32) <= T) lst[pos++] = key[r];
    return __builtin_amdgcn_readlane(incl, 63);
}
__device__ __forceinline__ int wave_filter_dispatch(int R, uint64_t *lst, int n, uint32_t T, int lane)
{
    if (R <= 8) return wave_filter<8>(lst, n, T, lane);
    if (R <= 16) return wave_filter<16>(lst, n, T, lane);
    return wave_filter<32>(lst, n, T, lane);
}

// The same selection for arrays too long for one wave's registers (up to 4096 keys: lists of a search with k in
// (1536, 2048], 4096 published keys): nothing is cached, every probe re-reads the array (64-bit loads, coalesced, L2
// hits; eight in flight per lane).  A handful of probes of <= 64 loads per lane against a workgroup-wide LDS sort of
// 4096 keys (78 barrier-separated stages) that kept all four waves on one query.  Same contract as wave_select:
// keeps k .. kmax keys (the k best among them), packed to the front of dst, unsorted; dst may be the array itself --
// the packing of iteration r writes below index 64 (r + 1), all of which has been read by then.
template <typename LD>
__device__ __attribute__((noinline)) int wave_select_mem(LD load, int n, int k, int kmax, int lane, uint32_t *thr_ord, uint64_t *dst)
{
    n = __builtin_amdgcn_readfirstlane(n);
    k = __builtin_amdgcn_readfirstlane(k);
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    const int iters = (n + 63) >> 6;
    // fn(key, valid) for every key of this lane, eight loads in flight
    auto for_each = [&](auto &&fn) {
        int r = 0;
        for (; r + 8 <= iters; r += 8) {
            uint64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int idx = (r + u) * 64 + lane;
                v[u] = load(idx < n ? idx : n - 1);
                if (idx >= n) v[u] = KEY_PAD;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) fn(v[u]);
        }
        for (; r < iters; r++) {
            const int idx = r * 64 + lane;
            uint64_t v = load(idx < n ? idx : n - 1);
            if (idx >= n) v = KEY_PAD;
            fn(v);
        }
    };
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    int real = 0;
    for_each([&](uint64_t key) {
        const uint32_t hi = (uint32_t)(key >> 32);
        mn = min(mn, hi);
        if (hi != 0xFFFFFFFFu) {
            mx = max(mx, hi);
            real++;
        }
    });
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    real = wave_sum(real); // keys whose score word is not the padding word
    auto count_lt = [&](uint32_t X) {
        int c = 0;
        for_each([&](uint64_t key) { c += (uint32_t)(key >> 32) < X ? 1 : 0; });
        return wave_sum(c);
    };
    uint32_t T = 0xFFFFFFFFu; // keep every key with score word <= T (0xFFFFFFFF: every real key)
    int cnt = real;
    if (real > kmax && mn != mx) {
        uint32_t lo_x = mn, hi_x = mx + 1u;
        int f_lo = 0, f_hi = real;
        const float target = 0.5f * (float)(k + kmax);
        bool interpolate = true, done = false;
        while (hi_x - lo_x > 1u) {
            uint32_t X;
            if (interpolate)
                X = lo_x + (uint32_t)((float)(hi_x - lo_x) * ((target - (float)f_lo) / (float)(f_hi - f_lo)));
            else
                X = lo_x + ((hi_x - lo_x) >> 1);
            X = max(lo_x + 1u, min(X, hi_x - 1u));
            interpolate = !interpolate;
            const int c = count_lt(X);
            if (c < k) {
                lo_x = X;
                f_lo = c;
            } else if (c > kmax) {
                hi_x = X;
                f_hi = c;
            } else {
                T = X - 1u;
                cnt = c;
                done = true;
                break;
            }
        }
        if (!done) { // lo_x is the exact k-th smallest score word
            T = lo_x;
            cnt = f_hi;
        }
    } else if (real > kmax) {
        T = mn; // every real key carries the same score word: the id tie-break below decides
    }
    uint32_t Q = 0xFFFFFFFFu;
    if (cnt > kmax && T != 0xFFFFFFFFu) {
        // more keys tie at T than may be kept: lowest ids win
        const int need = k - count_lt(T);
        Q = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t Qt = Q | (1u << bit);
            int c = 0;
            for_each([&](uint64_t key) { c += ((uint32_t)(key >> 32) == T && (uint32_t)key < Qt) ? 1 : 0; });
            if (wave_sum(c) <= need - 1) Q = Qt;
        }
    }
    // survivors, 64 keys per step, packed in place with a ballot (see above for why dst may alias the array)
    int base = 0;
    for (int r = 0; r < iters; r++) {
        const int idx = r * 64 + lane;
        uint64_t key = load(idx < n ? idx : n - 1);
        if (idx >= n) key = KEY_PAD;
        const uint32_t hi = (uint32_t)(key >> 32);
        const bool keep = hi != 0xFFFFFFFFu && (hi < T || (hi == T && (uint32_t)key <= Q));
        const unsigned long long m = __ballot(keep);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < kmax && dst) dst[pos] = key;
        base += __popcll(m);
    }
    *thr_ord = T;
    return min(base, kmax);
}

// ... and the threshold filter for such an array: keeps the keys with score word <= T, packed in place
__device__ __attribute__((noinline)) int wave_filter_mem(uint64_t *lst, int n, uint32_t T, int lane)
{
    n = __builtin_amdgcn_readfirstlane(n);
    T = __builtin_amdgcn_readfirstlane(T);
    int base = 0;
    for (int r = 0; r < (n + 63) >> 6; r++) {
        const int idx = r * 64 + lane;
        const uint64_t key = idx < n ? ((gptr_u64)lst)[idx] : KEY_PAD;
        const bool keep = key != KEY_PAD && (uint32_t)(key >> 32) <= T;
        const unsigned long long m = __ballot(keep);
        if (keep) lst[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
        base += __popcll(m);
    }
    return base;
}

// The bound of a tile-minimum seed: the k-th smallest of the keys published so far (one call site in the scan kernel:
// a second one -- the memory-probing variant next to the register variant -- cost a 1.25 M-row launch 70 us)
__device__ __attribute__((noinline)) void published_bound(const uint64_t *pub, int n, int k, int kmax, int lane, uint32_t *thr_ord)
{
    n = __builtin_amdgcn_readfirstlane(n);
    if (n <= 2048) wave_select_dispatch((n + 63) >> 6, LoadAgent{pub}, n, k, kmax, lane, thr_ord, (uint64_t *)nullptr);
    else wave_select_mem(LoadAgent{pub}, n, k, kmax, lane, thr_ord, (uint64_t *)nullptr);
}

// ---------------------------------------------------------------------------
// scan kernel
// ---------------------------------------------------------------------------
// Symmetric all-vs-all (the queries ARE the database rows): workgroup w multiplies query tile qtile with the
// database tiles [jt0, jt0 + jcount), all of them on or above the diagonal; see flat_scan_kernel<..., SYM>.
struct SymItem {
    int qtile, jt0, jcount;
};

struct ScanParams {
    const float *xb;   // [nb][dp] database rows, zero padded to dp
    const float *yn;   // [nb] squared norms (L2 only)
    const float *xq;   // [nq][dp] queries
    const float *xn;   // [nq] squared norms (L2 only)
    const float *xq_diff; // difference builds: [DNQ][dp] the queries interleaved by pairs (diff_interleave_kernel)
    int64_t nb, nq;
    int dp;            // multiple of 32
    int k;
    int cap;           // candidate list capacity per (workgroup, query); power of two
    int nqtiles, nchunks;
    int tiles_base, tiles_rem; // chunk c walks tiles_base (+1 if c < tiles_rem) database tiles: a balanced split
    uint64_t *lists;   // [grid][QT][cap]
    uint32_t *gthr;    // [nqtiles*QT] shared running thresholds (order-mapped floats)
    uint64_t *qlist;   // [nq][qcap] compact candidate arrays: the chunks' survivors, appended as they finish
    uint32_t *qcnt;    // [nq] fill of each array
    int qcap;
    uint32_t id_base;
    int row_mul;       // this launch scans a block-strided view of the database, see view_row()
    int vshift;        // log2 of the view's block size (3: blocks of 8 rows, 0: single rows)
    int skip_mask;     // >= 0: rows whose block b has (b & skip_mask) == 0 belong to the seed sample, skip them
    int kslot;         // most keys a chunk hands on per query: k + k/4
    const struct SymItem *sym_items; // symmetric all-vs-all launch: the work of each workgroup (see flat_scan_kernel)
    int *fail;         // symmetric launch: set when a candidate array overflows
    // tile-minimum seed (streaming regime, see flat_scan_kernel): after each of its first pub_rounds tiles a workgroup
    // publishes every query's best key of that tile; the k-th smallest published key bounds the global k-th
    uint64_t *pub;     // [nqtiles*QT][pub_n], pub_n = pub_rounds * nchunks, KEY_PAD = not published yet; NULL: off
    uint32_t *arrive;  // [nqtiles] publications so far
    int pub_n, pub_rounds;
    int pub_m;         // keys a workgroup publishes per query and round: 1 (the tile's best) or WM (each wave's best of its own rows)
    float *defer;      // [grid][QT * DT] the parked scores of every workgroup's first tile
    // paired workgroups (one-query-tile launches): workgroups w and w + npairs share ONE range of tiles, w walks it from
    // the front, w + npairs from the back; every tile is claimed with a ticket (see flat_scan_kernel)
    uint32_t *pair_ctr; // [npairs] tickets handed out so far (starts at 2: the first tile of either side); NULL: off
                        // pair_ctr[npairs]: tickets of the pool
    int npairs;
    int pool_tiles;     // the last pool_tiles tiles of every pair's range belong to a pool shared by ALL workgroups
    uint32_t *cu_turn;  // [2048] batch launches: one word per CU -- the two resident workgroups take turns in their K loops; NULL: off
    int sparse_epi;     // 128 x 128 batch builds: filter the tiles through the sparse epilogue (thresholds that let a percent of the
                        // scores pass: statistical seed, no seed) instead of filter_tile (exactly seeded scans: almost nothing passes,
                        // its wave-wide early-outs cost next to nothing; the sparse epilogue's fixed 2 us per tile cost a 10 M-row
                        // scan of 1024 queries 4 %)
#ifdef KNN355_TRACE
    int ablate;                // developer build: 1 = skip the filter, 2 = masks only (no reservations / stores), 4 = wait for the accumulators before the stamp
    unsigned long long *trace; // developer build: [grid][128] wall-clock stamps (100 MHz) of each workgroup's progress (64.. : inside the epilogue)
#endif
};
#ifdef KNN355_TRACE
#define KNN_TRACE(slot) do { if (threadIdx.x == 0 && p.trace && (slot) < 128) p.trace[(size_t)blockIdx.x * 128 + (slot)] = wall_clock64(); } while (0)
#define KNN_TRACE_AT(cond, slot) do { if (cond) KNN_TRACE(slot); } while (0)
#define KNN_TRACE_COUNT(cond, slot) do { if (threadIdx.x == 0 && p.trace && (cond)) p.trace[(size_t)blockIdx.x * 128 + (slot)] += 1; } while (0)
#else
#define KNN_TRACE(slot) do { } while (0)
#define KNN_TRACE_AT(cond, slot) do { } while (0)
#define KNN_TRACE_COUNT(cond, slot) do { } while (0)
#endif

// Views: view row r of a launch with stride row_mul and block size B = 1 << vshift is database row
// (r / B) * B * row_mul + r % B -- blocks of B consecutive rows, every row_mul-th block (row_mul = 1:
// the database itself).  Seed samples of the streaming regime use blocks of 8 rows, because one
// staging instruction fetches 8 rows: 8 rows 32 KB apart in one page instead of 8 pages (a
// row-strided sample of a 41 GB database ran at 1.9 TB/s, bound by address translation); the
// statistical samples of the batch regime use single rows (a stratified sample of a database
// whose families sit next to each other).
__host__ __device__ __forceinline__ int64_t view_row(int64_t r, int row_mul, int vshift)
{
    return ((r >> vshift) * row_mul << vshift) + (r & (((int64_t)1 << vshift) - 1));
}
// rows of the view with stride row_mul over a database of n rows
static inline int64_t view_rows(int64_t n, int row_mul, int vshift)
{
    const int64_t B = (int64_t)1 << vshift, span = B * row_mul;
    if (n <= 0) return 0;
    const int64_t last = (n - 1) / span; // last block that starts inside the database
    return last * B + std::min<int64_t>(B, n - last * span);
}

// ---- per-workgroup candidate lists shared by the scan kernels -------------------------
// s_thr / s_cnt / s_need live in LDS, the lists themselves in global memory (L2).
struct ListCtx {
    float *s_thr;     // [QT] local admission threshold (value of the loosest kept key)
    int *s_cnt;       // [QT] list fill
    int *s_need;      // some list could overflow on the next tile
    uint64_t *lists;  // [QT][cap] this workgroup's lists
    uint32_t *gthr;   // [QT] shared running thresholds of these queries
    int cap, k;
    int kslot;        // size of this workgroup's output slot per query (k + k/4)
    bool prefilter;   // end of chunk: drop what the shared threshold has overtaken before any selection

    __device__ __forceinline__ void init(int tid, int QT, int nthreads = 256)
    {
        for (int i = tid; i < QT; i += nthreads) {
            s_thr[i] = INFINITY;
            s_cnt[i] = 0;
        }
        if (tid == 0) *s_need = 0;
        prefilter = false;
    }
    __device__ __forceinline__ float threshold(int ql) const
    {
        const uint32_t g = __hip_atomic_load(&gthr[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return fminf(s_thr[ql], ord2f(g));
    }
    // v: "smaller is better" score of row `id` for local query ql (caller checked v <= threshold)
    __device__ __forceinline__ void append(int ql, float v, uint32_t id, int tile_rows)
    {
        v = v + 0.0f;
        const int slot = atomicAdd(&s_cnt[ql], 1);
        if (slot < cap) lists[(size_t)ql * cap + slot] = ((uint64_t)f2ord(v) << 32) | id;
        if (slot + 1 > cap - tile_rows) *s_need = 1;
    }
};

// Called by all threads after a tile's appends (and a barrier).  Lists that could overflow on the
// next tile are cut back to the best [k, min(1.25k, cap - tile)] keys; with `final` set (end of the
// chunk) every list longer than kslot is cut to at most kslot keys instead.  Cuts happen in place
// (survivors packed to the front of the list, unsorted); exactness is the final selection's job.
template <int QT, int NT = 256>
__device__ __forceinline__ void lists_compact(ListCtx &L, char *smem, int tile_rows, bool final, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int R = L.cap >> 6;
    if (R <= 64) {
        // one wave per query: one query in flight per wave of the workgroup.  Lists of up to 2048 keys are selected in
        // registers, the 4096-key lists of k in (1536, 2048] by probing the list in memory (wave_select_mem)
        auto select = [&](uint64_t *lst, int n, int kmax, uint32_t *T) -> int {
            if (R <= 32) return wave_select_dispatch(R, LoadContig{lst}, n, L.k, kmax, lane, T, lst);
            return wave_select_mem(LoadContig{lst}, n, L.k, kmax, lane, T, lst);
        };
        for (int ql = wave; ql < QT; ql += NT / 64) {
            const int n = __builtin_amdgcn_readfirstlane(min(L.s_cnt[ql], L.cap));
            uint64_t *lst = L.lists + (size_t)ql * L.cap;
            uint32_t T = 0;
            int kmax;
            if (!final) {
                if (n <= L.cap - tile_rows) continue;
                // at most cap - tile_rows keys may stay: the next tile can append tile_rows more
                // (k in (1433, 1536] with 256-row tiles: 1.25 k alone would leave too little room)
                kmax = min(L.k + (L.k >> 2), L.cap - tile_rows);
            } else {
                if (n <= L.kslot) {
                    if (lane == 0) L.s_cnt[ql] = n;
                    continue;
                }
                kmax = L.kslot;
                if (L.prefilter) {
                    const int m = R <= 32 ? wave_filter_dispatch(R, lst, n, f2ord(L.threshold(ql)), lane)
                                          : wave_filter_mem(lst, n, f2ord(L.threshold(ql)), lane);
                    if (m <= L.kslot) {
                        if (lane == 0) L.s_cnt[ql] = m;
                        continue;
                    }
                    const int cnt2 = select(lst, m, kmax, &T);
                    if (lane == 0) {
                        L.s_cnt[ql] = cnt2;
                        if (T != 0xFFFFFFFFu) { // (0xFFFFFFFF: the list held no more than kmax real keys beside its KEY_PAD entries -- no bound, and ord2f of it is a NaN)
                            L.s_thr[ql] = ord2f(T);
                            atomicMin(&L.gthr[ql], T);
                        }
                    }
                    continue;
                }
            }
            const int cnt = select(lst, n, kmax, &T);
            if (lane == 0) {
                L.s_cnt[ql] = cnt;
                if (T != 0xFFFFFFFFu) { // (see above: a cut that fell into the padding gives no bound)
                    L.s_thr[ql] = ord2f(T);
                    atomicMin(&L.gthr[ql], T);
                }
            }
        }
        __syncthreads();
    } else {
        // lists longer than 4096 keys (no plan makes them: k <= 2048): whole-workgroup LDS sort
        uint64_t *sb = (uint64_t *)smem;
        for (int ql = 0; ql < QT; ql++) {
            const int n = min(L.s_cnt[ql], L.cap);
            if (!final && n <= L.cap - tile_rows) continue;
            if (final && n <= L.kslot) {
                if (tid == 0) L.s_cnt[ql] = n; // (readers clamp with cap themselves: old or new value, same n)
                continue;
            }
            uint64_t *lst = L.lists + (size_t)ql * L.cap;
            const int P = next_pow2_dev(n > 0 ? n : 1);
            for (int i = tid; i < P; i += NT) sb[i] = i < n ? lst[i] : KEY_PAD;
            __syncthreads();
            wg_bitonic_sort(sb, P, tid, NT);
            const int keep = min(n, L.k);
            for (int i = tid; i < keep; i += NT) lst[i] = sb[i];
            if (tid == 0) {
                L.s_cnt[ql] = keep;
                if (n >= L.k) {
                    const uint32_t o = (uint32_t)(sb[L.k - 1] >> 32);
                    if (o != 0xFFFFFFFFu) { // (a list may hold KEY_PAD entries -- rows that are not there: fewer than k real keys, no bound)
                        L.s_thr[ql] = ord2f(o);
                        atomicMin(&L.gthr[ql], o);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) *L.s_need = 0;
    __syncthreads();
}

// End of a chunk (after lists_compact(final)): every query's survivors are appended to its compact
// candidate array qlist[q][0 .. qcap) behind whatever other chunks (and the seed sample) have put
// there: one atomic add per (workgroup, query) reserves the slots, then the keys are copied, one
// wave per query.  qcap = (lists per query) x kslot: the reservations can never run past it.
template <int QT, int NT = 256>
__device__ __forceinline__ void lists_flush(ListCtx &L, int *s_base, int64_t q0, int64_t nq, uint64_t *qlist, uint32_t *qcnt,
                                            int qcap, int tid, int *fail = nullptr)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int nqv = (int)min((int64_t)QT, nq - q0);
    for (int ql = tid; ql < nqv; ql += NT) {
        const int n = L.s_cnt[ql];
        s_base[ql] = n > 0 ? (int)atomicAdd(&qcnt[q0 + ql], (uint32_t)n) : 0;
    }
    __syncthreads();
    // four queries per step and wave: their loads go out together (one query at a time, a 128-query workgroup spent 57 us
    // here waiting for 64 dependent round trips to L2 -- 3 % of a CATH-sized symmetric launch with its short runs)
    constexpr int NW = NT / 64, U = 4;
    for (int ql0 = wave; ql0 < nqv; ql0 += NW * U) {
        uint64_t v[U][2];
        int n[U], base[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int ql = ql0 + u * NW;
            n[u] = ql < nqv ? L.s_cnt[ql] : 0;
            base[u] = ql < nqv ? s_base[ql] : 0;
            const uint64_t *lst = L.lists + (size_t)(ql < nqv ? ql : 0) * L.cap;
#pragma unroll
            for (int j = 0; j < 2; j++) v[u][j] = lane + 64 * j < n[u] ? ((gptr_u64)lst)[lane + 64 * j] : KEY_PAD;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int ql = ql0 + u * NW;
            if (ql >= nqv) continue;
            const uint64_t *lst = L.lists + (size_t)ql * L.cap;
            uint64_t *dst = qlist + (size_t)(q0 + ql) * qcap + base[u];
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (lane + 64 * j < n[u] && base[u] + lane + 64 * j < qcap) dst[lane + 64 * j] = v[u][j];
            for (int i = lane + 128; i < n[u]; i += 64)
                if (base[u] + i < qcap) dst[i] = lst[i];
            // (only a symmetric launch sizes the arrays by expectation: an overflow there repeats the search the plain way)
            if (fail && lane == 0 && base[u] + n[u] > qcap) *fail = 1;
        }
    }
}

// one staging instruction: 64 lanes x 16 B straight from global memory into LDS (LDS-DMA, no VGPR
// round trip).  NT (aux bit 1 = non-temporal): for data that is read once.
template <bool NT = false>
__device__ __forceinline__ void stage_issue(const float *src, char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, NT ? 2 : 0);
}

// scheduling pattern "one MFMA, one LDS read", N times: a sub-step's fragment reads go out one per MFMA -- four waves that
// issue eight ds_read_b128 each in one burst keep the LDS array busy for two MFMA times (one wave per SIMD: 3.5 % of the
// 256 x 256 tile's K loop, measured by ablation)
template <int N>
__device__ __forceinline__ void sched_reads()
{
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        sched_reads<N - 1>();
    }
}

// scheduling pattern "PER MFMAs, one vector-memory instruction", N times
template <int N, int PER>
__device__ __forceinline__ void sched_spread()
{
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        sched_spread<N - 1, PER>();
    }
}

// WM x WN waves; each wave owns TM x TN MFMA tiles of 32(db rows) x 32(queries)
// NTDB: the launch has ONE query tile, so every database row is read by exactly one workgroup:
// its staging loads are non-temporal and do not displace the queries (re-read by every workgroup
// each K step) from L2 / Infinity Cache -- 10 M x 32 queries +4 %.  With several query tiles the
// workgroups of a chunk share the rows through L2 and non-temporal loads cost 3 %.
// SYM: all-vs-all with the index's own rows as queries.  dot(x, y) = dot(y, x) bit for bit (the chain multiplies the same
// pairs in the same order) and nrm(x) + nrm(y) commutes, so the score tile of (query tile I, database tile J) also holds
// the scores of (query tile J, database tile I): only the tiles on and above the diagonal are multiplied -- half the
// MFMA work.  Every off-diagonal tile is filtered twice: lane-wise for the queries of tile I (the resident tile, as
// always), and row-wise for the rows of tile J taken as queries, whose survivors go straight to those queries' compact
// arrays (they have no workgroup-local list here: the statistical seed keeps them few, an overflow raises the
// verification flag and the search is redone the plain way).
// BF16: the operands are bf16 copies of the rows (an index made approximate on purpose: the coarse entry scan of the
// HNSW index, where only the neighbourhood matters and every returned distance is re-scored exactly afterwards).
// A staged 128-byte row segment then holds 64 values instead of 32, the same 16-byte fragment feeds ONE
// v_mfma_f32_32x32x16_bf16 instead of four fp32 MFMAs (1/16 of the matrix time), and everything else -- staging,
// swizzle, lists, thresholds, selection -- is shared.  p.dp counts 4-byte units of a row here (d / 2).
// DNQ > 0 ("DIFF"): FAISS's squared L2 for batches of fewer than 20 queries (distance_compute_blas_threshold [ext]; reachable
// through seqvec_search/main.py:22-45 with metric=METRIC_L2 and a handful of queries): the sum of (x - y)^2 itself, not
// |x|^2 + |y|^2 - 2<x,y>.  Contract: ONE fp32 chain per (query, row), acc = fma(t, t, acc) with t = x[k] - y[k], k in the
// same order as the dot product (0,4,1,5,2,6,3,7 per block of 8).  No matrix instruction computes that: the 32-query
// tile's staging, lists, thresholds, seeding and selection are kept and the K step's MFMAs are replaced by vector
// subtract / fma chains -- thread = two rows of the 256-row tile (their 32 floats of the K step in registers), one chain per
// (query, row), the queries' floats read from LDS as broadcasts.  Behind the K loop the scores are transposed through the
// staging buffers into the accumulator layout the MFMA would have left (lane (i, h) = query i, registers = 32 rows), so the
// filter, the tile-minimum seed and everything else are shared.  The chains of the two queries of a PAIR against one row
// share packed fp32 instructions (v_pk_add_f32 / v_pk_fma_f32; round 4 -- round 3 packed two rows of one query: 12.2 ms per
// 10 M x 1024 rows for 13-19 queries, bound by LDS bank conflicts; scalar chains: 19.4; the MFMA scan with the norm formula:
// 7.0-7.4 ms); builds for up to 12 and up to 4 queries (a single query: its HBM time).
// Q16 > 0: the wave's query columns are Q16 blocks of SIXTEEN (v_mfma_f32_16x16x4_f32 -- the same k-ordered fma chain, bit for
// bit: tools/micro/mfma16.hip; lane (i, g = lane / 16) feeds k = g of each instruction: floats 8t + {0, 4, 1, 5}[g] and
// 8t + {2, 6, 3, 7}[g] per block of 8) instead of TN blocks of 32: a 48-query tile (4 x 1 waves, 33..48 queries) and a
// 96-query tile (2 x 2 waves, 65..96 queries) pay for 48 / 96 columns of matrix work, not 64 / 128.  One-query-tile launches
// only (plain fp32).
// TM x TN = 4 x 4 ("BIG", round 5): a 256 x 256 tile on four waves, ONE workgroup per CU, one wave per SIMD with up to 512
// registers -- 256 of them the wave's 128 x 128 accumulators (AGPRs), 128 KB of LDS for the two staging buffers.  Half the
// staged bytes per flop of the 128 x 128 tile, a quarter of the K-step barriers and first-fragment waits per flop; nothing
// else on the CU fills the epilogue's gaps, so this build is for long chunks only (make_plan).
template <int WM, int WN, int TM, int TN, bool L2, bool NTDB = false, bool SYM = false, bool BF16 = false, int DNQ = 0, int Q16 = 0>
__global__ __launch_bounds__(256, (TM * TN >= 16 ? 1 : 2)) void flat_scan_kernel(ScanParams p)
{
    constexpr bool DIFF = DNQ > 0; // the difference build, for batches of up to DNQ queries (8, 12, 16 or 20)
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(!DIFF || (L2 && !SYM && !BF16 && WN == 1 && TN == 1), "the difference build: one 32-query tile, squared L2");
    constexpr bool M16 = Q16 > 0;
    static_assert(!M16 || (NTDB && !SYM && !BF16 && DNQ == 0 && TN == 1), "16-query blocks: plain fp32, one query tile per launch");
    constexpr int DT = WM * TM * 32;        // database rows per tile
    constexpr int QT = WN * (M16 ? Q16 * 16 : TN * 32); // queries per workgroup
    constexpr int NB = M16 ? Q16 : TN;      // query blocks of a wave (of 16 / 32 queries)
    constexpr int NS = TM * (M16 ? 8 : 16); // scores of one query block a lane holds per tile
    constexpr bool BIGT = TM * TN >= 16;    // the 256 x 256 tile: one workgroup per CU, accumulators in AGPRs, late-barrier K loop
    // the batch builds (several query tiles per launch, plain fp32): the 128 x 128 and the 256 x 256 tile filter their scores
    // through the sparse epilogue
    constexpr bool SPARSE = !NTDB && !BF16 && DNQ == 0 && Q16 == 0 && WM == 2 && WN == 2 && TM == TN && (TM == 2 || TM == 4);
    constexpr int ROWS = DT + QT;           // staged rows per K step
    constexpr int NGRP = ROWS / 8;          // staging instructions per K step (8 rows each) ...
    constexpr int NI = (NGRP + 3) / 4;      // ... per wave; a wave short of one (304 rows: 38 over 4 waves) repeats its previous one
    static_assert(ROWS % 8 == 0, "staging split");
    constexpr int STAGE_BYTES = ROWS * 128; // 32 floats per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *stage0 = smem;
    char *stage1 = smem + STAGE_BYTES;
    // (the symmetric 128 x 128 build's sparse epilogue keeps 36 slots of 2 KB: 8 KB more than the two staging buffers)
    const int lds_main = max(max(2 * STAGE_BYTES, p.cap * 8), SYM && TM * TN == 4 ? 73728 : 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // (scalar: the LDS targets of the staging instructions are then SGPR arithmetic, no v_readfirstlane per instruction)
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int lq = M16 ? (lane & 15) : li, lg = M16 ? (lane >> 4) : lh; // this lane's query of a block / its row group (accumulator layout)

    int qtile, chunk;
    int64_t c_lo, c_hi;
    if constexpr (SYM) {
        static_assert(DT == QT, "square tiles");
        const SymItem it = p.sym_items[blockIdx.x];
        qtile = it.qtile;
        chunk = 0;
        c_lo = (int64_t)it.jt0 * DT;
        c_hi = min(p.nb, c_lo + (int64_t)it.jcount * DT);
    } else {
        qtile = blockIdx.x % p.nqtiles;
        chunk = blockIdx.x / p.nqtiles;
        if (NTDB && p.pair_ctr) chunk = blockIdx.x % p.npairs; // (one query tile: blockIdx.x is the workgroup of the pair)
        c_lo = ((int64_t)chunk * p.tiles_base + min(chunk, p.tiles_rem)) * DT;
        c_hi = min(p.nb, c_lo + (int64_t)(p.tiles_base + (chunk < p.tiles_rem ? 1 : 0)) * DT);
    }
    // Paired walk (NTDB launches = one query tile = the HBM-bound streaming regime).  Two workgroups per CU is what the
    // LDS holds, and the two residents of a CU do not progress alike: the older one wins the arbitration and finished its
    // 10 tiles of a 1.25 M-row shard one tile time before the younger one, which then ran its last tile alone (per-
    // workgroup stamps: every CU's later workgroup ended at 1000 us, its earlier one at 907; 23 CUs had one workgroup and
    // idled from 600 us on).  So a launch has exactly 2 x CUs workgroups, workgroups w and w + npairs (dispatched one
    // round apart: an older and a younger resident) share ONE range of tiles, w takes tiles from the front, w + npairs
    // from the back, and each tile is claimed with a ticket from the pair's counter: both stay busy until the range is
    // used up, whatever their speeds.  Each side still walks contiguous rows (translations and DRAM pages stay warm --
    // pulling single tiles from one global counter cost 2.5-10 %).  The ticket of the NEXT tile is drawn at the start of
    // the current one: its latency hides behind the K loop.
    const bool paired = NTDB && !SYM && p.pair_ctr != nullptr;
    const int side = paired ? (int)(blockIdx.x / p.npairs) : 0;
    const int ntl = (int)((c_hi - c_lo + DT - 1) / DT); // tiles of this chunk / of the pair's range
    // The pool: pairs do not progress alike either (per-XCD means of the workgroups' lives differed by 4 %, the slowest CU
    // ended 6 % after the median one): the last pool_tiles tiles of every pair's range are nobody's -- a workgroup whose
    // pair has used up its own tiles draws them, any pair's, from one global counter until they are gone.
    const int n_own = paired ? ntl - p.pool_tiles : ntl;
    const int tile_first = (int)(c_lo / DT);
    int *s_next = nullptr;
    const int64_t q0 = (int64_t)qtile * QT;
    const int KT = p.dp / 32;
    ListCtx L;
    L.s_thr = (float *)(smem + lds_main);
    L.s_cnt = (int *)(L.s_thr + QT);
    L.s_need = L.s_cnt + QT;
    s_next = L.s_need + 1;                 // the ticket drawn for this workgroup's next tile (paired walk)
    int *s_base = L.s_need + 4;            // [QT] first slot of each query's survivors in its compact array
    float *s_yn = (float *)(s_base + QT);  // [DT] squared norms of the current tile's rows (L2 only)
    float *s_thr2 = s_yn + DT;             // SYM: [DT] thresholds of the tile's rows taken as queries
    int *s_cnt2 = (int *)(s_thr2 + DT);    // SYM: [WN][DT] survivors per row found by the waves of each query half
    int *s_base2 = s_cnt2 + WN * DT;       // SYM: [DT] first slot reserved in each row's compact array
    L.lists = p.lists + (size_t)blockIdx.x * QT * p.cap;
    L.gthr = p.gthr + (size_t)qtile * QT;
    L.cap = p.cap;
    L.k = p.k;
    L.kslot = p.kslot;
    L.init(tid, QT);
    // Tile-minimum seed (plain fp32 launches of the streaming regime; the host turns it on when the launch has a few
    // times k chunks).  No sample pass in front of the launch: every workgroup filters its first tile against nothing,
    // publishes each query's BEST key of that tile, and the k-th smallest of all keys published so far -- keys of k
    // different rows -- is an upper bound of the global k-th: one wave of the publishing workgroup works it out for one
    // query (wave_select over <= 2048 published keys) and lowers the shared threshold.  With c chunks of DT-row tiles
    // the bound sits near the k / (c DT) quantile (1.25 M rows, k = 100: ~1100 candidates per query for the whole
    // launch; the sample pass it replaces cost 90 us in front of a 0.85 ms scan and admitted 12 k).  A publication that
    // a reader does not see yet reads as KEY_PAD: the bound is then looser, never wrong.
    constexpr bool CAN_PUB = !SYM && !BF16 && QT <= 64;
    uint64_t *s_pub = (uint64_t *)s_thr2; // [QT] (a plain launch has no s_thr2)
    const bool pub_on = CAN_PUB && p.pub != nullptr;
    L.prefilter = pub_on;
    if (pub_on && tid < QT) s_pub[tid] = KEY_PAD;
    __syncthreads();
    // per-lane staging bookkeeping: instruction ii covers combined rows 8*ii..8*ii+7
    const float *srcp[NI];
    int lds_off[NI];
    bool is_db[NI];
    int rloc[NI];
#pragma unroll
    for (int n = 0; n < NI; n++) {
        int ii = wave + 4 * n;
        if (ii >= NGRP) ii -= 4; // (the same rows once more, to the same place: harmless)
        int row_local = 8 * ii + (lane >> 3);
        int sp = lane & 7;
        lds_off[n] = ii * 1024;
        if (row_local < DT) {
            is_db[n] = true;
            rloc[n] = row_local;
            int s = sp ^ ((row_local >> 1) & 7);
            srcp[n] = p.xb + 4 * s; // row added per tile
        } else {
            is_db[n] = false;
            int rq = row_local - DT;
            rloc[n] = rq;
            int s = sp ^ ((rq >> 1) & 7);
            if constexpr (DNQ > 0) { // the difference build stages the queries interleaved by pairs: DNQ rows of diff_interleave_kernel's matrix
                srcp[n] = p.xq_diff + (int64_t)min(rq, DNQ - 1) * p.dp + 4 * s;
            } else {
                int64_t q = min(q0 + rq, p.nq - 1);
                srcp[n] = p.xq + q * p.dp + 4 * s;
            }
        }
    }
    const int swz = (li >> 1) & 7;

    uint32_t *my_turn = nullptr;
    if constexpr (!NTDB && !BF16) {
        if (p.cu_turn) { // this CU's word: XCC_ID (hwreg 20) and HW_ID (hwreg 4: cu_id [11:8], sh_id [12], se_id [15:13])
            const uint32_t hw = (uint32_t)__builtin_amdgcn_s_getreg(0xF804), xcc = (uint32_t)__builtin_amdgcn_s_getreg(0xF814) & 7u;
            my_turn = p.cu_turn + ((xcc << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u));
        }
    }
    // (256 x 256 builds) what the sparse epilogue needs per query block and never changes: the queries' norms, and which lanes'
    // queries may be candidates of the symmetric launch's second direction (real rows the sample pass has not handed on)
    float big_xnq[NB];
    uint64_t big_cokm[NB];
    if constexpr (SPARSE) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int64_t q = q0 + (M16 ? 0 : (wn * TN + b) * 32 + li);
            const bool qok = q < p.nq;
            big_xnq[b] = 0.0f;
            if constexpr (L2) big_xnq[b] = p.xn[qok ? q : 0];
            big_cokm[b] = __ballot(qok && !(p.skip_mask >= 0 && ((int)(q >> p.vshift) & p.skip_mask) == 0));
        }
    }
    int tile_idx = 0; // tiles this workgroup has walked
    // the tile being walked / the one after it, as tile numbers of the view (-1: none)
    int cur_tile = paired ? (side == 0 ? tile_first : tile_first + n_own - 1) : tile_first;
    if (paired ? side >= n_own : ntl <= 0) cur_tile = -1;
    int own_walked = 0;     // (thread 0) tiles drawn from the pair's own range, the first one included
    bool in_pool = false;   // (thread 0) the pair's own tiles are gone
    KNN_TRACE(0);
#ifdef KNN355_TRACE
    if (threadIdx.x == 0 && p.trace) // where this workgroup runs: HW_ID (hwreg 4) and XCC_ID (hwreg 20)
        p.trace[(size_t)blockIdx.x * 128 + 62] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804);
#endif
    int64_t first_row0 = c_lo;
    const float *tsrc[NI]; // this lane's source of each of its staging instructions for the tile being walked (first K step)
    bool staged0 = false;  // (256 x 256 builds) K step 0 of the tile about to start is already on its way into buffer 0
    while (cur_tile >= 0) {
        const int64_t row0 = (int64_t)cur_tile * DT;
        if (tile_idx == 0) first_row0 = row0;
        int next_tile = cur_tile + 1 < tile_first + ntl ? cur_tile + 1 : -1; // (unpaired: the chunk in order)
        if (paired && tid == 0) {
            next_tile = -1;
            if (!in_pool) {
                own_walked++;
                const int t = (int)__hip_atomic_fetch_add(&p.pair_ctr[chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t < n_own) next_tile = side == 0 ? tile_first + own_walked : tile_first + n_own - 1 - own_walked;
                else in_pool = true;
            }
            if (in_pool && p.pool_tiles > 0) {
                const int g = (int)__hip_atomic_fetch_add(&p.pair_ctr[p.npairs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g < p.npairs * p.pool_tiles) {
                    // pool tile g: the (g / npairs)-th reserved tile of pair g % npairs
                    const int pj = g % p.npairs, pi = g / p.npairs;
                    next_tile = pj * p.tiles_base + min(pj, p.tiles_rem) + (p.tiles_base + (pj < p.tiles_rem ? 1 : 0)) - p.pool_tiles + pi;
                }
            }
        }
        f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
        f32x4 acc16[M16 ? 2 * TM : 1][M16 ? Q16 : 1]; // 16-query blocks: [row tile of 16][query block], lane (q = lane & 15, g = lane / 16): rows 4 g + r
#pragma unroll
        for (int a = 0; a < (M16 ? 1 : TM); a++)
#pragma unroll
            for (int b = 0; b < (M16 ? 1 : TN); b++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
#pragma unroll
        for (int a = 0; a < (M16 ? 2 * TM : 1); a++)
#pragma unroll
            for (int b = 0; b < (M16 ? Q16 : 1); b++) acc16[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // score s (0 .. NS) of query block b in this lane: its accumulator register, and its row inside the tile
        auto accv = [&](int b, int s) -> float {
            if constexpr (M16) return acc16[s >> 2][b][s & 3];
            else return acc[s >> 4][b][s & 15];
        };
        auto rowl = [&](int s) -> int {
            if constexpr (M16) return (wm * TM * 2 + (s >> 2)) * 16 + 4 * lg + (s & 3);
            else return (wm * TM + (s >> 4)) * 32 + 4 * lh + (s & 3) + 8 * ((s & 15) >> 2);
        };
        auto qloc = [&](int b) -> int { return M16 ? (wn * Q16 + b) * 16 + lq : (wn * TN + b) * 32 + li; };
        constexpr int DIFF_NQ = DIFF ? DNQ : 1; // (the difference build serves batches of fewer than 20 queries: one chain per query)
        f32x2 dacc2[(DIFF_NQ + 1) / 2];        // ... as one chain per query and row; a thread holds two rows x half the query pairs: [row][pair] (see compute)
#pragma unroll
        for (int qi = 0; qi < (DIFF_NQ + 1) / 2; qi++) dacc2[qi] = f32x2{0.0f, 0.0f};

        if (!staged0) { // (256 x 256 builds: the previous tile's epilogue has worked these out already -- and staged the first K step)
#pragma unroll
            for (int n = 0; n < NI; n++) {
                if (is_db[n]) {
                    int64_t r = view_row(min(row0 + rloc[n], p.nb - 1), p.row_mul, p.vshift);
                    tsrc[n] = srcp[n] + r * p.dp;
                } else {
                    tsrc[n] = srcp[n];
                }
            }
        }
        if constexpr (L2) {
            // the tile's squared norms: one coalesced load per thread now, LDS reads in the epilogue
            // (visible after the K loop's barriers) instead of 16 * TM * TN scattered global loads
            // per lane at the end of the tile
            if (tid < DT) s_yn[tid] = p.yn[view_row(min(row0 + tid, p.nb - 1), p.row_mul, p.vshift)];
        }
        const bool off_diag = SYM && row0 != q0; // (square tiles: the diagonal tile starts at the query tile's first row)
        // (256 x 256 builds) the queries' thresholds as they stand NOW: an agent-scope load is a round trip to memory, four of
        // them one after the other were 8 of the epilogue's 10 us; read in front of the K loop they cost nothing, and a bound
        // one tile old only admits a few candidates more (thresholds only ever tighten)
        float big_thr[NB];
        if constexpr (SPARSE) {
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int ql = M16 ? 0 : (wn * TN + b) * 32 + li;
                big_thr[b] = q0 + ql < p.nq ? L.threshold(ql) : -INFINITY; // (a query past the end admits nothing)
            }
        }
        if constexpr (SYM) {
            if (tid < DT) // rows past the end are nobody's query: nothing beats -inf
                s_thr2[tid] = row0 + tid < p.nb ? ord2f(__hip_atomic_load(&p.gthr[row0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : -INFINITY;
            if constexpr (SPARSE) { // (the sparse epilogue's per-row counters and its "this tile is dense" flag; visible behind the K loop's barriers)
                if (tid < DT) s_cnt2[tid] = 0;
                if (tid == 0) L.s_need[2] = 0;
            }
        }
        // One K step of MFMA work from buffer `cur`.  `dma(n)` (n < NI) issues this wave's n-th
        // staging instruction of the NEXT K step; the NI of them are spread between the MFMAs so
        // that their issue cost (60-180 cycles each) is paid while the matrix pipe is busy, not
        // in front of it.  ND = 0: nothing to stage.
        int diff_koff = 0; // (difference builds) first float of the K step being multiplied
        auto compute = [&](const char *cur, auto &&dma, auto nd_tag) {
            constexpr int ND = decltype(nd_tag)::value;
            const char *A = cur;
            const char *B = cur + DT * 128;
            if constexpr (DIFF) {
#pragma unroll
                for (int n = 0; n < ND; n++) dma(n); // (the next K step's staging goes out first: the chains below take microseconds)
                // thread = TWO rows (r0, r0 + 128) and half of the query PAIRS; the two halves of a packed fp32 instruction are
                // the two queries of a pair: (qa[k] - y[k], qb[k] - y[k]), squared and added into the pair's two chains
                // (v_pk_add_f32 / v_pk_fma_f32).  The queries are staged INTERLEAVED by pairs (diff_interleave_kernel: staged
                // row 2p holds qa[k], qb[k] for k = 0..15 of the K step, row 2p + 1 for k = 16..31), so one ds_read_b128 (a
                // broadcast: every lane reads the same address) delivers two k of both queries as two aligned register pairs,
                // and the row's own floats come as plain ds_read_b128 fragments (consecutive lanes = consecutive rows: no bank
                // conflict), broadcast into both halves by op_sel.  Round 3 packed two ROWS per instruction instead: their
                // floats needed 32 ds_read2_b32 per K step that hit 8 of the 32 banks (4-way conflicts), every wave re-read
                // every query float, and the K step was bound by the LDS (12.2 ms per 10 M rows x 19 queries).
                //   Chains: 2 rows x NPH pairs per thread, all independent, interleaved instruction by instruction; the query
                // fragments of the half-block after next are requested while this one is computed.  Every chain visits
                // k = 8t + m, 8t + 4 + m (m = 0..3, t = 0..3): the dot product's order.
                constexpr int NPH = DIFF_NQ / 4;
                const int r0 = (wave & 1) * 64 + lane, qh = wave >> 1;
                const int fy = (lane >> 1) & 7; // (r0 >> 1) & 7 -- the same for r0 + 128
                f32x4 y0[8], y1[8];
#pragma unroll
                for (int sl = 0; sl < 8; sl++) {
                    y0[sl] = *(const f32x4 *)(A + r0 * 128 + ((sl ^ fy) * 16));
                    y1[sl] = *(const f32x4 *)(A + (r0 + 128) * 128 + ((sl ^ fy) * 16));
                }
                f32x4 qv[2][NPH][2];
                // half-block hb (8 per K step): block t = hb / 2 (8 k), its half u = hb % 2: k = 8t + 2u, 8t + 4 + 2u, 8t + 2u + 1,
                // 8t + 4 + 2u + 1 -- 16-byte slots 4 (t & 1) + u and 4 (t & 1) + 2 + u of staged row 2p + t / 2
                auto loadq = [&](int hb) {
                    const int t = hb >> 1, u = hb & 1;
                    // the first SQ query pairs of this thread's half come by SCALAR loads from the interleaved query matrix (the
                    // address is the same for every lane; constant address space + an address made of scalars only: s_load_dwordx4
                    // -- through a flat pointer the compiler takes vector loads and spills): they never touch the LDS pipe, which the
                    // broadcast reads of the wider builds had saturated.  Three pairs at most: with every pair in scalar registers
                    // the 16- / 20-query builds spill THOSE (9.7 / 12.5 ms per 10 M rows against 8.9 / 10.4 through the LDS)
                    // (not in the 8-query build: it runs at the HBM rate either way, and with the lighter LDS load the chip settled on a
                    // lower clock -- 6.60 against 6.31 ms per 10 M rows in steady state, three alternating runs on one box)
                    constexpr int SQ = KNN355_DIFF_SCALAR_Q && NPH >= 3 ? (NPH < KNN355_DIFF_SCALAR_PAIRS ? NPH : KNN355_DIFF_SCALAR_PAIRS) : 0;
                    typedef const __attribute__((address_space(4))) f32x4 *cq4;
#pragma unroll
                    for (int j = 0; j < SQ; j++) {
                        const int pr = qh * NPH + j;
                        const float *qg = p.xq_diff + (int64_t)(2 * pr + (t >> 1)) * p.dp + diff_koff + 4 * (4 * (t & 1) + u);
                        qv[hb & 1][j][0] = *(cq4)qg;
                        qv[hb & 1][j][1] = *(cq4)(qg + 8);
                    }
#pragma unroll
                    for (int j = SQ; j < NPH; j++) {
                        const int pr = qh * NPH + j;
                        const char *qrow = B + (2 * pr + (t >> 1)) * 128;
                        const int fq = pr & 7; // ((2 pr) >> 1) & 7 -- the same for 2 pr + 1
                        qv[hb & 1][j][0] = *(const f32x4 *)(qrow + (((4 * (t & 1) + u) ^ fq) * 16));
                        qv[hb & 1][j][1] = *(const f32x4 *)(qrow + (((4 * (t & 1) + 2 + u) ^ fq) * 16));
                    }
                };
                auto chains = [&](int hb) {
                    const int t = hb >> 1, u = hb & 1;
#pragma unroll
                    for (int e = 0; e < 2; e++) {       // first / second k of the slots
#pragma unroll
                        for (int ab = 0; ab < 2; ab++) { // slot 4 (t & 1) + u (k = 8t + ..), then slot .. + 2 (k = 8t + 4 + ..)
                            const float ya = y0[2 * t + ab][2 * u + e], yb = y1[2 * t + ab][2 * u + e];
#pragma unroll
                            for (int j = 0; j < NPH; j++) {
                                const f32x4 v = qv[hb & 1][j][ab];
                                const f32x2 qp = e == 0 ? f32x2{v[0], v[1]} : f32x2{v[2], v[3]};
                                const f32x2 ta = qp - f32x2{ya, ya};
                                dacc2[j] = __builtin_elementwise_fma(ta, ta, dacc2[j]);
                                const f32x2 tb = qp - f32x2{yb, yb};
                                dacc2[NPH + j] = __builtin_elementwise_fma(tb, tb, dacc2[NPH + j]);
                            }
                        }
                    }
                };
                loadq(0);
                loadq(1);
#pragma unroll
                for (int hb = 0; hb < 8; hb++) {
                    chains(hb);
                    if (hb + 2 < 8) loadq(hb + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            if constexpr (M16) {
                // 16 x 16 x 4 tiles: a wave's 64 rows are RT = 2 TM row tiles, its queries Q16 blocks.  Lane (i = lane & 15,
                // g = lane / 16) feeds k = g of every instruction, and the chain must visit k = 0,4,1,5 | 2,6,3,7 of each
                // block of 8: the lane reads the 16-byte slot 2t + (g & 1) of its row (floats 8t + 4 (g & 1) ..) and hands
                // element g / 2 to the block's first instruction, element g / 2 + 2 to its second.
                constexpr int RT = 2 * TM;
                const int g = lane >> 4, swz16 = ((lane & 15) >> 1) & 7;
                const bool upper = (g >> 1) != 0;
                f32x4 af16[2][RT], bf16[2][Q16];
                auto frag16 = [&](int t) {
                    const int slot = ((2 * t + (g & 1)) ^ swz16) * 16;
#pragma unroll
                    for (int a = 0; a < RT; a++)
                        af16[t & 1][a] = *(const f32x4 *)(A + ((wm * RT + a) * 16 + (lane & 15)) * 128 + slot);
#pragma unroll
                    for (int b = 0; b < Q16; b++)
                        bf16[t & 1][b] = *(const f32x4 *)(B + ((wn * Q16 + b) * 16 + (lane & 15)) * 128 + slot);
                };
                constexpr int M = 2 * RT * Q16; // MFMAs per sub-step
                frag16(0);
                __builtin_amdgcn_sched_group_barrier(0x100, RT + Q16, 0);
                auto substep16 = [&](auto t_tag) {
                    constexpr int t = decltype(t_tag)::value;
                    if constexpr (t < 3) frag16(t + 1);
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        float av[RT], bv[Q16];
                        // (one v_cndmask per operand.  The empty asm keeps the two candidates opaque: left alone, the compiler
                        // folds a fragment's two selects into ONE extraction with a four-way dynamic index and spends three
                        // v_cndmask on each)
                        auto pick = [&](const f32x4 &v) -> float {
                            float lo = v[2 * half], hi = v[2 * half + 1];
                            asm("" : "+v"(lo), "+v"(hi));
                            return upper ? hi : lo;
                        };
#pragma unroll
                        for (int a = 0; a < RT; a++) av[a] = pick(af16[t & 1][a]);
#pragma unroll
                        for (int b = 0; b < Q16; b++) bv[b] = pick(bf16[t & 1][b]);
#pragma unroll
                        for (int a = 0; a < RT; a++)
#pragma unroll
                            for (int b = 0; b < Q16; b++)
                                acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc16[a][b], 0, 0, 0);
                    }
                    constexpr int TD = 2;
                    constexpr int n0 = t < TD ? ND * t / TD : ND, n1 = t < TD ? ND * (t + 1) / TD : ND; // (spread over the first half: see the 32 x 32 x 2 path)
#pragma unroll
                    for (int n = n0; n < n1; n++) dma(n);
                    if constexpr (t < 3) __builtin_amdgcn_sched_group_barrier(0x100, RT + Q16, 0);
                    constexpr int nd = n1 - n0, per = M / (nd + 1);
                    static_assert(per >= 1, "more staging instructions than MFMAs in a sub-step");
                    sched_spread<nd, per>();
                    __builtin_amdgcn_sched_group_barrier(0x008, M - nd * per, 0);
                };
                substep16(std::integral_constant<int, 0>{});
                substep16(std::integral_constant<int, 1>{});
                substep16(std::integral_constant<int, 2>{});
                substep16(std::integral_constant<int, 3>{});
                return;
            }
            // fragments of sub-step t+1 are requested before the MFMAs of sub-step t are issued,
            // so the LDS latency hides behind the matrix pipe (two register sets, fully unrolled)
            f32x4 af[2][TM], bf[2][TN];
            auto frag = [&](int t) {
                const int slot = ((2 * t + lh) ^ swz) * 16;
#pragma unroll
                for (int a = 0; a < TM; a++)
                    af[t & 1][a] = *(const f32x4 *)(A + ((wm * TM + a) * 32 + li) * 128 + slot);
#pragma unroll
                for (int b = 0; b < TN; b++)
                    bf[t & 1][b] = *(const f32x4 *)(B + ((wn * TN + b) * 32 + li) * 128 + slot);
            };
            constexpr int M = BF16 ? TM * TN : 4 * TM * TN; // MFMAs per sub-step
            frag(0);
            if constexpr (!BF16) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
            auto substep = [&](auto t_tag) {
                constexpr int t = decltype(t_tag)::value;
                if constexpr (t < 3) frag(t + 1);
                if constexpr (BF16) {
#pragma unroll
                    for (int a = 0; a < TM; a++)
#pragma unroll
                        for (int b = 0; b < TN; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[t & 1][a]),
                                                                                __builtin_bit_cast(bf16x8, bf[t & 1][b]), acc[a][b], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; m++)
#pragma unroll
                        for (int a = 0; a < TM; a++)
#pragma unroll
                            for (int b = 0; b < TN; b++)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t & 1][a][m], bf[t & 1][b][m], acc[a][b], 0, 0, 0);
                }
                // staging instructions [n0, n1) of the next K step belong to this sub-step; all of
                // them go out in the first TD sub-steps of the K step so that the rest covers
                // their latency before the next barrier's vmcnt(0)
                constexpr int TD = NTDB ? 2 : 3; // (batch launches hit L2: spread wide, +1 %)
                // Streaming launches (one query tile, rows straight from HBM): EVERY staging instruction of the next K step goes
                // out in front of this K step's first MFMA -- the longest lead the double buffer allows.  (Rounds 2-3 spread them
                // over the first half of the K step so that their issue cost hid behind MFMAs: 1.2-3 % slower once the epilogue and
                // the walk were what they are now -- 10 M x 32 queries 7.39 -> 7.30 ms, 1.25 M rows 0.972 -> 0.945 ms, same box.)
                // (the 32- and 64-query builds only: the 128-query one-tile launch is bound by the matrix pipe and lost 2 % with it,
                // the builds on 16-query blocks 0.5 %)
                constexpr bool FIRST = NTDB && QT <= 64 && KNN355_STREAM_DMA_FIRST;
                if constexpr (FIRST && t == 0) {
#pragma unroll
                    for (int n = 0; n < ND; n++) dma(n);
                    __builtin_amdgcn_sched_group_barrier(0x010, ND, 0);
                }
                constexpr int n0 = FIRST ? ND : (t < TD ? ND * t / TD : ND), n1 = FIRST ? ND : (t < TD ? ND * (t + 1) / TD : ND);
#pragma unroll
                for (int n = n0; n < n1; n++) dma(n);
                // pin the order: LDS reads of t+1, then the MFMAs of t with the staging
                // instructions spread between them (left alone, the scheduler sinks the reads
                // behind the MFMAs to save registers and the wait is exposed)
                if constexpr (!BF16) { // (the bf16 build is bound by staging, not by the matrix pipe: the compiler's order will do)
                    if constexpr (t < 3) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                    constexpr int nd = n1 - n0, per = M / (nd + 1);
                    static_assert(per >= 1, "more staging instructions than MFMAs in a sub-step");
                    sched_spread<nd, per>();
                    __builtin_amdgcn_sched_group_barrier(0x008, M - nd * per, 0);
                }
            };
            substep(std::integral_constant<int, 0>{});
            substep(std::integral_constant<int, 1>{});
            substep(std::integral_constant<int, 2>{});
            substep(std::integral_constant<int, 3>{});
        };
        auto no_dma = [](int) {};
        using nd_none = std::integral_constant<int, 0>;
        using nd_all = std::integral_constant<int, NI>;
        if constexpr (BIGT) {
            // ---- the 256 x 256 tile's K loop: ONE wave per SIMD, so nothing but this wave's own instruction order hides a
            // latency.  One barrier per K step, in front of the step's LAST sub-step ("late barrier"):
            //   sub-step 0, 1   MFMAs; the staging instructions of K step kt + 1 (groups A, B) between them
            //   sub-step 2      MFMAs only -- the staged bytes have a whole sub-step (4096 matrix cycles) to land --, then
            //                   vmcnt(0) + barrier: stage kt + 1 is in LDS for everyone, and everyone has read stage kt for the
            //                   last time (the fragments of sub-step 3 were requested at the start of sub-step 2)
            //   sub-step 3      MFMAs from registers; the FIRST fragments of K step kt + 1 are requested behind the first MFMA
            //                   (64 MFMAs cover their latency: the next K step starts without a wait), and group C of K step
            //                   kt + 2 is staged into the buffer stage kt has just left
            // Every sub-step issues one MFMA, then the fragment reads of the next sub-step: the compiler waits lgkmcnt(0) in front
            // of a sub-step's first MFMA, and with the reads behind that MFMA the wait only ever covers reads a sub-step old.
            constexpr int MS = 4 * TM * TN;                        // MFMAs per sub-step
            // staging instructions per group: half of K step kt + 1's in sub-step 3 of step kt - 1 (C), half in sub-step 0 of step kt
            // (A), none in sub-step 1 (B) -- two whole sub-steps (3.7 us) for the bytes to land before the wait in front of the
            // barrier.  (Thirds, the last of them with one sub-step of lead: fine while the rows come from L2 / the Infinity Cache
            // -- a Pfam-sized index -- and 5 % slower than the 128 x 128 tile on a 10 M-row index, whose rows come from HBM.)
            constexpr int NA = NI / 2, NBg = 0, NC = NI - NA - NBg;
#ifdef KNN355_TRACE
            // (developer build: K loop without its staging instructions / its barrier / its fragment reads -- wrong scores, right time)
            const bool stage_on = !(p.ablate & 16), barrier_on = !(p.ablate & 32), reads_on = !(p.ablate & 128);
#else
            constexpr bool stage_on = true, barrier_on = true, reads_on = true;
#endif
            f32x4 fa[2][TM], fb[2][TN];
            auto rd = [&](const char *buf, int t, int set) {
                if (!reads_on) return;
                const int slot = ((2 * t + lh) ^ swz) * 16;
#pragma unroll
                for (int a = 0; a < TM; a++) fa[set][a] = *(const f32x4 *)(buf + ((wm * TM + a) * 32 + li) * 128 + slot);
#pragma unroll
                for (int b = 0; b < TN; b++) fb[set][b] = *(const f32x4 *)(buf + DT * 128 + ((wn * TN + b) * 32 + li) * 128 + slot);
            };
            auto mfmas = [&](int set) {
#pragma unroll
                for (int m = 0; m < 4; m++)
#pragma unroll
                    for (int a = 0; a < TM; a++)
#pragma unroll
                        for (int b = 0; b < TN; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a][m], fb[set][b][m], acc[a][b], 0, 0, 0);
            };
            auto kstep = [&](int kt, auto h1_tag, auto h2_tag) {
                constexpr bool H1 = decltype(h1_tag)::value, H2 = decltype(h2_tag)::value; // K steps kt + 1 / kt + 2 exist
                char *cur = (kt & 1) ? stage1 : stage0;
                char *nxt = (kt & 1) ? stage0 : stage1;
                const int k1 = (kt + 1) * 32, k2 = (kt + 2) * 32;
                // sub-step 0 (set 0), fragments of sub-step 1 -> set 1, group A
                mfmas(0);
                rd(cur, 1, 1);
                if constexpr (H1) {
#pragma unroll
                    for (int n = 0; n < NA; n++) if (stage_on) stage_issue<false>(tsrc[n] + k1, nxt + lds_off[n]);
                }
                sched_reads<TM + TN>();
                sched_spread<(H1 ? NA : 0), (MS - (TM + TN)) / (NA + 1)>();
                __builtin_amdgcn_sched_group_barrier(0x008, MS - (TM + TN) - (H1 ? NA : 0) * ((MS - (TM + TN)) / (NA + 1)), 0);
                // sub-step 1 (set 1), fragments of sub-step 2 -> set 0, group B
                mfmas(1);
                rd(cur, 2, 0);
                if constexpr (H1) {
#pragma unroll
                    for (int n = NA; n < NA + NBg; n++) if (stage_on) stage_issue<false>(tsrc[n] + k1, nxt + lds_off[n]);
                }
                sched_reads<TM + TN>();
                sched_spread<(H1 ? NBg : 0), (MS - (TM + TN)) / (NBg + 1)>();
                __builtin_amdgcn_sched_group_barrier(0x008, MS - (TM + TN) - (H1 ? NBg : 0) * ((MS - (TM + TN)) / (NBg + 1)), 0);
                // sub-step 2 (set 0), fragments of sub-step 3 -> set 1, nothing staged; then the K step's one barrier
                mfmas(0);
                rd(cur, 3, 1);
                sched_reads<TM + TN>();
                __builtin_amdgcn_sched_group_barrier(0x008, MS - (TM + TN), 0);
                __builtin_amdgcn_sched_barrier(0); // (the wait and the barrier stay BEHIND this sub-step's MFMAs: they are the bytes' time to land)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's part of stage kt + 1 has landed (see the generic loop)
                if (barrier_on) __syncthreads();
                // sub-step 3 (set 1), first fragments of K step kt + 1 -> set 0, group C of K step kt + 2 into the buffer just left
                // (pinned by hand, one fragment read behind each of the first MFMAs and group C in NC pieces between the later ones:
                // a sched_group_barrier pipeline in the region behind the barrier came out as one burst of NC staging
                // instructions -- ISA checked)
                constexpr int PER = MS / (NC + 1);
                const int slot0 = (lh ^ swz) * 16; // (the first fragments of a K step: 16-byte slot 2 * 0 + lh)
#pragma unroll
                for (int i = 0; i < MS; i++) {
                    const int m = i / (TM * TN), a = (i / TN) % TM, b = i % TN;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][a][m], fb[1][b][m], acc[a][b], 0, 0, 0);
                    bool pinned = false;
                    if (H1 && i < TM + TN && reads_on) { // one fragment read behind each of the first MFMAs
                        if (i < TM) fa[0][i] = *(const f32x4 *)(nxt + ((wm * TM + i) * 32 + li) * 128 + slot0);
                        else fb[0][i - TM] = *(const f32x4 *)(nxt + DT * 128 + ((wn * TN + (i - TM)) * 32 + li) * 128 + slot0);
                        pinned = true;
                    }
                    if (H2 && (i + 1) % PER == 0 && (i + 1) / PER <= NC) {
                        const int n = NA + NBg + (i + 1) / PER - 1;
                        if (stage_on) stage_issue<false>(tsrc[n] + k2, cur + lds_off[n]);
                        pinned = true;
                    }
                    if (pinned) __builtin_amdgcn_sched_barrier(0);
                }
            };
            // prologue: stage K step 0, wait, barrier; group C of K step 1; the first fragments (the one exposed read of the tile)
            if (!staged0) {
#pragma unroll
                for (int n = 0; n < NI; n++) stage_issue<false>(tsrc[n], stage0 + lds_off[n]);
            }
            staged0 = false;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (KT > 1) {
#pragma unroll
                for (int n = NA + NBg; n < NI; n++) stage_issue<false>(tsrc[n] + 32, stage1 + lds_off[n]);
            }
            rd(stage0, 0, 0);
            using yes = std::true_type;
            using no = std::false_type;
            int kt = 0;
            for (; kt + 2 < KT; kt++) kstep(kt, yes{}, yes{});
            if (kt + 1 < KT) {
                kstep(kt, yes{}, no{});
                kt++;
            }
            kstep(kt, no{}, no{});
        } else {
            // prologue: stage K step 0 into buffer 0
#pragma unroll
            for (int n = 0; n < NI; n++) {
                if (NTDB && n < DT / 32) stage_issue<true>(tsrc[n], stage0 + lds_off[n]); // rows, not queries
                else stage_issue<false>(tsrc[n], stage0 + lds_off[n]);
            }
            // Batch launches: the two workgroups of a CU take TURNS in their K loops.  Per-workgroup stamps showed a K loop
            // running alone (its neighbour in the epilogue) at 94-97 % of the matrix pipe's rate, and two K loops side by side
            // at 73-84 % together -- every wave's K-step barrier then also waits for the neighbour's waves on the other three
            // SIMDs.  One word per CU (found through HW_ID / XCC_ID), taken by thread 0 before the K loop and given back behind
            // it; the neighbour filters its previous tile meanwhile and waits out the rest.  (The word is cleared in front of
            // every launch; a workgroup alone on its CU never waits.)  Measured: +0.7 % on Pfam-sized launches, +1 % on CATH-sized
            // ones -- a K loop on its own reaches ~85 % of the pipe as well (one wave per SIMD: nothing fills its barrier bubbles).
            if constexpr (!NTDB && !BF16) {
                if (p.cu_turn) {
                    if (tid == 0) {
                        // (bounded: a turn that never comes -- it cannot, every holder gives the word back behind its K loop -- costs
                        // a few milliseconds, not the launch)
                        for (int it = 0; it < 8192; it++) {
                            if (__hip_atomic_exchange(my_turn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
                            __builtin_amdgcn_s_sleep(32);
                        }
                    }
                    // (no barrier of its own: the K loop's first barrier holds the other waves until thread 0 is through)
                }
            }
            for (int kt = 0; kt + 1 < KT; kt++) {
                char *cur = (kt & 1) ? stage1 : stage0;
                char *nxt = (kt & 1) ? stage0 : stage1;
                diff_koff = kt * 32;
                // Stage kt has landed in LDS for EVERY wave's reads: this wave's staging instructions are waited for HERE,
                // explicitly, then the barrier.  (Nothing else orders a ds_read behind a pending LDS-DMA: until round 4 the wait
                // was the compiler's -- it puts a vmcnt(0) in front of LDS reads it cannot tell apart from a pending DMA's target
                // -- and one instantiation lost it when the address arithmetic of the staging instructions changed: wrong scores
                // from the third K step on.  Inline asm: the waitcnt pass cannot drop it.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads(); // buffer nxt is free
                const int koff = (kt + 1) * 32;
                compute(cur, [&](int n) {
                    if (NTDB && n < DT / 32) stage_issue<true>(tsrc[n] + koff, nxt + lds_off[n]);
                    else stage_issue<false>(tsrc[n] + koff, nxt + lds_off[n]);
                }, nd_all{});
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (see above)
            __syncthreads();
            diff_koff = (KT - 1) * 32;
            compute(((KT - 1) & 1) ? stage1 : stage0, no_dma, nd_none{});
            if constexpr (!NTDB && !BF16) {
                if (p.cu_turn && tid == 0) __hip_atomic_store(my_turn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if constexpr (DIFF) {
            // the scores change hands: from "thread = row, register = query" to the accumulator layout the MFMA leaves behind
            // (lane (i, h) = query i, registers = 32 rows), through the staging buffers -- everything behind the K loop is shared
            static_assert(DT == 256 && QT == 32, "thread = row");
            float *sT = (float *)smem; // [DIFF_NQ][DT]
            __syncthreads();           // (every wave has read its last fragments)
            {
                constexpr int NPH = DIFF_NQ / 4;
                const int r0 = (wave & 1) * 64 + lane, qh = wave >> 1;
#pragma unroll
                for (int j = 0; j < NPH; j++) {
                    const int pr = qh * NPH + j; // queries 2 pr, 2 pr + 1
                    sT[(2 * pr) * DT + r0] = dacc2[j][0];
                    sT[(2 * pr + 1) * DT + r0] = dacc2[j][1];
                    sT[(2 * pr) * DT + r0 + 128] = dacc2[NPH + j][0];
                    sT[(2 * pr + 1) * DT + r0 + 128] = dacc2[NPH + j][1];
                }
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < TM; a++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[a][0][r] = li < DIFF_NQ ? sT[li * DT + (wm * TM + a) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)] : INFINITY;
            __syncthreads();           // (the next tile's prologue stages into these buffers)
        }
        KNN_TRACE(1 + 2 * tile_idx);
        // Two workgroups per CU: behind its K loop a wave's instructions run beside the MFMAs of the OTHER workgroup's wave on
        // the same SIMD.  Straight-line vector code gets through at its own rate (the sparse epilogue's pass 1: 1.6 us for 64
        // scores, alone on the CU or not); anything that WAITS -- an LDS round trip, an atomic's return, a barrier -- takes three
        // to five times what it takes with the CU to itself (per-workgroup stamps of a CATH-sized symmetric tile: epilogue 57 us
        // with two workgroups per CU, 17.5 us with one).  Raising the wave's issue priority here (s_setprio, compile-time knob)
        // changes nothing measurable.
        if constexpr (!BIGT && KNN355_EPI_PRIO > 0) __builtin_amdgcn_s_setprio(KNN355_EPI_PRIO);
        if (paired && tid == 0) *s_next = next_tile; // (read by everyone behind the barrier that ends the epilogue)

        // score s of query block b in this lane (see accv / rowl) -- "smaller is better"
        auto score_of = [&](int b, int s, float xnq) -> float {
            if constexpr (DIFF) {
                return accv(b, s);
            } else if constexpr (L2) {
                const float ynr = s_yn[rowl(s)];
                const float v = __builtin_fmaf(-2.0f, accv(b, s), xnq + ynr);
                return v < 0.0f ? 0.0f : v;
            } else {
                return -accv(b, s);
            }
        };
        // threshold filter + append of one tile's scores (getv(a, b, r, xnq): from the accumulators, or read back from
        // the deferred first tile)
        // Threshold filter + append of one tile's scores.  passes(a, b, r, thr, xnq): does the score beat thr; value(a, b, r,
        // xnq): the score -- from the accumulators, or read back from the parked first tile.
        //   This code runs beside the OTHER resident workgroup's K loop, whose MFMAs own the SIMD's issue port: per-workgroup
        // stamps of a CATH-sized search showed the filter of round 2 (~45 vector instructions per score: row numbers, view
        // and sample-mask arithmetic in 64 bits, the key, then a divergent append with its own LDS round trip) taking 15-60 us
        // per tile behind a 70-110 us K loop -- about one vector instruction per MFMA of the neighbour.  So: per score ONE
        // compare and a mask bit; everything else only for blocks of 8 scores in which some lane of the wave has a survivor,
        // with the 8 slot reservations (one LDS atomic per score from every lane, adding 0 where the score failed: no
        // divergence) in flight together.  Rows that are not really there (past the end in the last tile, rows of the seed
        // sample) are found only among the survivors and stored as KEY_PAD: every reader of a list drops those.
        auto filter_tile = [&](int64_t trow0, auto &&passes, auto &&value) {
            const bool ragged = trow0 + DT > p.nb, sampled = p.skip_mask >= 0, plain_rows = p.row_mul == 1;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int ql = qloc(b);
                const int64_t q = q0 + ql;
                const bool qok = q < p.nq;
                const float thr = qok ? L.threshold(ql) : -INFINITY; // (a query past the end admits nothing)
                float xnq = 0.0f;
                if constexpr (L2) xnq = p.xn[qok ? q : 0];
                uint64_t *lst = L.lists + (size_t)ql * L.cap;
                {
#pragma unroll
                    for (int s8 = 0; s8 < NS / 8; s8++) {
                        uint32_t passm = 0;
#pragma unroll
                        for (int r8 = 0; r8 < 8; r8++) passm |= (passes(b, s8 * 8 + r8, thr, xnq) ? 1u : 0u) << r8;
                        if (__ballot(passm != 0u) == 0ull) continue;
#ifdef KNN355_TRACE
                        if (p.ablate & 2) continue;
#endif
                        int slot[8];
#pragma unroll
                        for (int r8 = 0; r8 < 8; r8++) slot[r8] = atomicAdd(&L.s_cnt[ql], (int)((passm >> r8) & 1u));
                        bool near_full = false;
#pragma unroll
                        for (int r8 = 0; r8 < 8; r8++) {
                            if ((passm >> r8) & 1u) {
                                const int sc = s8 * 8 + r8;
                                const int64_t row = trow0 + rowl(sc);
                                const float vv = value(b, sc, xnq) + 0.0f;
                                const uint32_t id = p.id_base + (plain_rows ? (uint32_t)row : (uint32_t)view_row(row, p.row_mul, p.vshift));
                                const bool gone = (ragged && row >= p.nb) || (sampled && ((int)(row >> p.vshift) & p.skip_mask) == 0);
                                if (slot[r8] < L.cap) lst[slot[r8]] = gone ? KEY_PAD : (((uint64_t)f2ord(vv) << 32) | id);
                                near_full |= slot[r8] + 1 > L.cap - DT;
                            }
                        }
                        if (near_full) *L.s_need = 1;
                    }
                }
            }
        };
        // (squared L2: max(0, x) <= thr is x <= thr for every thr >= 0, and a threshold is a score or +inf)
        auto passes_acc = [&](int b, int sc, float thr, float xnq) -> bool {
            if constexpr (DIFF) {
                return accv(b, sc) <= thr;
            } else if constexpr (L2) {
                const float ynr = s_yn[rowl(sc)];
                return __builtin_fmaf(-2.0f, accv(b, sc), xnq + ynr) <= thr;
            } else {
                return -accv(b, sc) <= thr;
            }
        };
        constexpr int NV4 = NB * NS / 4; // 16-byte pieces of a lane's scores of one tile
        bool deferred = false;
        if constexpr (CAN_PUB) {
            if (pub_on && tile_idx < p.pub_rounds) {
                // ---- tile-minimum seed: publish this tile's best key per query BEFORE the tile is filtered ----
                // lane minimum over its 16 TM TN scores, the other lane half, then the waves through LDS.  The FIRST tile
                // has nothing to filter with yet (unfiltered it would append all of its 256 x QT scores: 32 MB of list
                // writes per launch on a 1.25 M-row shard, +110 us; waiting for the bound costs every workgroup the spread
                // of the arrivals): its scores are parked in global memory (32 KB per workgroup, one coalesced store per
                // 4 accumulator registers) and filtered behind the second tile, when the bound has long been there.
                float4 *park = (float4 *)p.defer + (size_t)blockIdx.x * NV4 * 256 + tid;
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const int ql = qloc(b);
                    const int64_t q = q0 + ql;
                    float xnq = 0.0f;
                    if constexpr (L2) xnq = p.xn[q < p.nq ? q : 0];
                    uint64_t best = KEY_PAD;
#pragma unroll
                    for (int s4 = 0; s4 < NS / 4; s4++) {
                        float vs[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int64_t row = row0 + rowl(4 * s4 + e);
                            const float v = score_of(b, 4 * s4 + e, xnq) + 0.0f;
                            vs[e] = v;
                            // (a key that is published must be one the filter would append: a real row of this chunk, finite)
                            if (v < INFINITY && row < p.nb && !(p.skip_mask >= 0 && ((int)(row >> p.vshift) & p.skip_mask) == 0)) {
                                const uint64_t key = ((uint64_t)f2ord(v) << 32) | (p.id_base + (uint32_t)view_row(row, p.row_mul, p.vshift));
                                best = key < best ? key : best;
                            }
                        }
                        if (tile_idx == 0) park[(size_t)(b * (NS / 4) + s4) * 256] = make_float4(vs[0], vs[1], vs[2], vs[3]);
                    }
                    // the other lanes that hold scores of this query: the other lane half (32-query blocks), the three other
                    // lane quarters (16-query blocks)
#pragma unroll
                    for (int off = M16 ? 16 : 32; off < 64; off <<= 1) {
                        const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), off);
                        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)best, off);
                        const uint64_t other = ((uint64_t)ohi << 32) | olo;
                        best = other < best ? other : best;
                    }
                    if (p.pub_m > 1) {
                        // every wave publishes the best key of ITS rows (WM keys of WM different rows per workgroup and query):
                        // enough publications for a k beyond the number of workgroups
                        if (lg == 0 && q < p.nq)
                            __hip_atomic_store(&p.pub[(size_t)q * p.pub_n + ((size_t)tile_idx * p.nchunks + (paired ? (int)blockIdx.x : chunk)) * WM + wm], best,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (lg == 0 && q < p.nq) {
                        atomicMin((unsigned long long *)&s_pub[ql], (unsigned long long)best);
                    }
                }
                deferred = tile_idx == 0;
                __syncthreads();
                if (wave == 0) {
                    if (lane < QT && p.pub_m == 1) {
                        if (q0 + lane < p.nq)
                            __hip_atomic_store(&p.pub[(size_t)(q0 + lane) * p.pub_n + (size_t)tile_idx * p.nchunks + (paired ? (int)blockIdx.x : chunk)], s_pub[lane],
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_pub[lane] = KEY_PAD; // (the next round's minima land after the next K loop's barriers)
                    }
                    // No fence, relaxed counter: a release at agent scope writes back this XCD's whole L2 and an acquire
                    // invalidates it (measured: +120 us per launch).  Nothing here needs the order: a publication a reader
                    // misses reads as KEY_PAD.
                    int a = 0;
                    if (lane == 0) a = (int)__hip_atomic_fetch_add(&p.arrive[qtile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a = __builtin_amdgcn_readfirstlane(a);
                    // arrivals take the queries in turn: from the (k + QT)-th on every query has a bound, the last QT
                    // arrivals of a round see (almost) every publication
                    const int ql = a % QT;
                    if (q0 + ql < p.nq && (a + 1) * p.pub_m >= p.k) {
                        uint32_t T = 0xFFFFFFFFu;
                        const int kmax = p.k + max(p.k >> 2, 32);
                        published_bound(p.pub + (size_t)(q0 + ql) * p.pub_n, p.pub_n, p.k, kmax, lane, &T);
                        if (lane == 0 && T != 0xFFFFFFFFu) atomicMin(&L.gthr[ql], T);
                    }
                }
            }
        }

        // ---- the 256 x 256 tile's epilogue ("sparse"): ONE workgroup per CU, so nothing runs beside it -- what filter_tile
        // spends per tile (60-120 us of a 240 us K loop, measured: nearly every block of 8 scores has a survivor in SOME lane,
        // and the wave then walks eight divergent appends) is all exposed.  A lane's 256 scores hold one to three survivors at
        // the densities a warmed-up or seeded threshold leaves (0.3-1.5 %), at random places, and registers cannot be indexed
        // by a lane-varying number.  So:
        //   pass 1  straight-line, eight vector instructions per score (inline asm: left to the compiler the accumulators were
        //           copied out of their AGPRs wholesale and spilled): value, ONE compare, a store of (value, score number) into
        //           the lane's next slot of the (idle) staging buffers -- [slot][thread] x 8 bytes -- by EVERY lane, and a
        //           pointer bump where the compare passed (a failed score's entry is overwritten by the lane's next survivor;
        //           first form, with the store under the compare's exec mask: 11.6 us per tile, the wave waiting on every
        //           vector-to-scalar hand-over); a lane that runs out of slots writes into a dump slot and keeps counting;
        //   pass 2  a wave-wide loop over the slots in use (as many rounds as the busiest lane has survivors): each lane reads
        //           its entry back, works out row and key, reserves a place in the query's list (LDS atomic) and stores.
        // A wave with a lane that ran out of slots (the first tiles of an unseeded chunk, tiles full of near-duplicates) filters
        // the tile the dense way instead.  The symmetric launch's second direction (row R of this tile as a query, the resident
        // tile's rows its candidates) goes the same way with a second compare per score; its survivors are ranked per row in
        // LDS (the atomic's return), one global atomic per row reserves the places, then the keys are stored; a lane out of
        // slots hands the whole tile's second direction to the dense code below.
        bool dir2_dense = SYM && off_diag; // the second direction is (still) the dense code's
        auto sparse_epilogue = [&](int64_t trow0, bool second, auto *dense2) { // (generic: only the 256 x 256 builds instantiate it)
            // slots per lane and tile, first / second direction (+ a dump slot each), all inside staging buffer 1: buffer 0 is
            // taking the next tile's first K step meanwhile
            // Slots per lane and tile (+ a dump slot).  256 x 256 tile: inside staging buffer 1 (buffer 0 is taking the next tile's
            // first K step meanwhile); 128 x 128 tile: both of its 32-KB staging buffers are idle during the epilogue, the
            // symmetric build adds 8 KB (lds_main).  The symmetric launch's two directions use the SAME slots, one after the
            // other: their survivor counts per lane have different tails (squared L2 on raw embeddings has hubs -- a candidate
            // of small norm beats the bound of every row: CATH-sized k = 301, a lane's largest count in a launch: first direction
            // 15, second 27 of its 64 scores, the mean 4.7), and half the slots each sent the second direction of nearly every
            // tile the dense way.
            constexpr int C = BIGT ? 31 : (SYM ? 35 : 30);
            constexpr int OFF1 = BIGT ? STAGE_BYTES : 0;
            static_assert(OFF1 + (C + 1) * 2048 <= (SYM && !BIGT ? 73728 : 2 * STAGE_BYTES), "the lanes' slots fit the staging buffer(s)");
            static_assert(NS % 16 == 0 && NS <= 64 && NB <= 4, "score numbers are sc | block << 8");
            const bool ragged = trow0 + DT > p.nb, sampled = p.skip_mask >= 0, plain_rows = p.row_mul == 1;
            const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
            int *s_dense = L.s_need + 2;
            const float *thr = big_thr, *xnq = big_xnq;
            const uint64_t *cokm = big_cokm;
            KNN_TRACE_AT(tile_idx == 5, 124);
            const uint32_t w_0 = lds0 + OFF1 + tid * 8;              // LDS address of this lane's first slot
            const uint32_t dump = w_0 + C * 2048, step = 2048;
            const float ninf = -INFINITY;
            uint32_t w = w_0;                                        // ... of its next slot (counts on past the last)
            uint32_t nblk[NB] = {};                                  // first direction: 2048 x this lane's survivors of each query block
            KNN_TRACE_AT(tile_idx == 5, 125);
            if constexpr (BIGT) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // (the K loop's last MFMAs -> v_accvgpr_read inside the statements below)
            // Pass 1 of one direction.  Four scores per statement: their value / compare chains are independent and issue back to
            // back (a chain on its own waits out every instruction's latency), only the pointer bumps are serial.  No exec mask, no
            // scalar instruction: EVERY lane stores (value, score number) into its next slot -- a score that fails is overwritten
            // by the lane's next survivor, which lands in the same slot -- and only a survivor moves the pointer on.
            auto pass1 = [&](auto dir_tag) {
                constexpr int DIR = decltype(dir_tag)::value;
                (void)&ninf; (void)&cokm; (void)&thr; (void)&nblk; // (named here: captures that only a discarded branch uses are not made)
#pragma unroll
                for (int c = 0; c < NS; c += 16) {
                    // the norms / second-direction thresholds of this chunk's 16 rows: four consecutive rows per 16-byte LDS read
                    // (all 64 rows of the 256-row tile at once cost 128 registers in the symmetric L2 build: spills between the statements)
                    float ynl[16], t2l[16];
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        f32x4 y4 = {0.0f, 0.0f, 0.0f, 0.0f}, h4 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                        if constexpr (L2) y4 = *(const f32x4 *)&s_yn[rowl(c + i)];
                        if constexpr (DIR == 2) h4 = *(const f32x4 *)&s_thr2[rowl(c + i)];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            ynl[i + e] = y4[e];
                            t2l[i + e] = h4[e];
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NB; b++) {
                        const uint32_t w_b = w; // (this lane's survivors of query block b: what the 16 scores below add to the pointer)
#pragma unroll
                        for (int sc = c; sc < c + 16; sc += 4) {
                            uint32_t t[4], ad[4], ix[4], e[4], u;
                            uint64_t mk[4];
                            (void)e; (void)t2l;
                            const float a0 = accv(b, sc), a1 = accv(b, sc + 1), a2 = accv(b, sc + 2), a3 = accv(b, sc + 3);
                            const float s0 = xnq[b] + ynl[sc - c], s1 = xnq[b] + ynl[sc - c + 1], s2 = xnq[b] + ynl[sc - c + 2], s3 = xnq[b] + ynl[sc - c + 3];
                            (void)s0; (void)s1; (void)s2; (void)s3;
                            if constexpr (DIR == 1) {
                                if constexpr (BIGT && L2) {
                                    asm volatile("v_accvgpr_read_b32 %[t0], %[a0]\n\t"
                                             "v_accvgpr_read_b32 %[t1], %[a1]\n\t"
                                             "v_accvgpr_read_b32 %[t2], %[a2]\n\t"
                                             "v_accvgpr_read_b32 %[t3], %[a3]\n\t"
                                             "v_fma_f32 %[t0], %[t0], -2.0, %[s0]\n\t"
                                             "v_fma_f32 %[t1], %[t1], -2.0, %[s1]\n\t"
                                             "v_fma_f32 %[t2], %[t2], -2.0, %[s2]\n\t"
                                             "v_fma_f32 %[t3], %[t3], -2.0, %[s3]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[thr]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w)
                                             : [a0] "a"(a0), [a1] "a"(a1), [a2] "a"(a2), [a3] "a"(a3), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [thr] "v"(thr[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                } else if constexpr (BIGT) {
                                    asm volatile("v_accvgpr_read_b32 %[t0], %[a0]\n\t"
                                             "v_accvgpr_read_b32 %[t1], %[a1]\n\t"
                                             "v_accvgpr_read_b32 %[t2], %[a2]\n\t"
                                             "v_accvgpr_read_b32 %[t3], %[a3]\n\t"
                                             "v_xor_b32_e32 %[t0], 0x80000000, %[t0]\n\t"
                                             "v_xor_b32_e32 %[t1], 0x80000000, %[t1]\n\t"
                                             "v_xor_b32_e32 %[t2], 0x80000000, %[t2]\n\t"
                                             "v_xor_b32_e32 %[t3], 0x80000000, %[t3]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[thr]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w)
                                             : [a0] "a"(a0), [a1] "a"(a1), [a2] "a"(a2), [a3] "a"(a3), [thr] "v"(thr[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                } else if constexpr (L2) { // (the 128 x 128 tile's accumulators are VGPRs: read in place)
                                    asm volatile("v_fma_f32 %[t0], %[a0], -2.0, %[s0]\n\t"
                                             "v_fma_f32 %[t1], %[a1], -2.0, %[s1]\n\t"
                                             "v_fma_f32 %[t2], %[a2], -2.0, %[s2]\n\t"
                                             "v_fma_f32 %[t3], %[a3], -2.0, %[s3]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[thr]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w)
                                             : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [thr] "v"(thr[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                } else {
                                    asm volatile("v_xor_b32_e32 %[t0], 0x80000000, %[a0]\n\t"
                                             "v_xor_b32_e32 %[t1], 0x80000000, %[a1]\n\t"
                                             "v_xor_b32_e32 %[t2], 0x80000000, %[a2]\n\t"
                                             "v_xor_b32_e32 %[t3], 0x80000000, %[a3]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[thr]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[thr]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w)
                                             : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [thr] "v"(thr[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                }
                            } else { // (the second direction: the same values decide for rows rowl(sc ..) as queries; a lane whose own query is no
                                     // candidate -- past the end, or a row of the sample -- compares against -inf)
                                if constexpr (BIGT && L2) {
                                    asm volatile("v_accvgpr_read_b32 %[t0], %[a0]\n\t"
                                             "v_accvgpr_read_b32 %[t1], %[a1]\n\t"
                                             "v_accvgpr_read_b32 %[t2], %[a2]\n\t"
                                             "v_accvgpr_read_b32 %[t3], %[a3]\n\t"
                                             "v_fma_f32 %[t0], %[t0], -2.0, %[s0]\n\t"
                                             "v_fma_f32 %[t1], %[t1], -2.0, %[s1]\n\t"
                                             "v_fma_f32 %[t2], %[t2], -2.0, %[s2]\n\t"
                                             "v_fma_f32 %[t3], %[t3], -2.0, %[s3]\n\t"
                                             "v_cndmask_b32_e64 %[e0], %[ninf], %[h0], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e1], %[ninf], %[h1], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e2], %[ninf], %[h2], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e3], %[ninf], %[h3], %[cok]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[e0]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[e1]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[e2]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[e3]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
                                             : [a0] "a"(a0), [a1] "a"(a1), [a2] "a"(a2), [a3] "a"(a3), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [h0] "v"(t2l[sc - c + 0]), [h1] "v"(t2l[sc - c + 1]), [h2] "v"(t2l[sc - c + 2]), [h3] "v"(t2l[sc - c + 3]), [ninf] "v"(ninf), [cok] "s"(cokm[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                } else if constexpr (BIGT) {
                                    asm volatile("v_accvgpr_read_b32 %[t0], %[a0]\n\t"
                                             "v_accvgpr_read_b32 %[t1], %[a1]\n\t"
                                             "v_accvgpr_read_b32 %[t2], %[a2]\n\t"
                                             "v_accvgpr_read_b32 %[t3], %[a3]\n\t"
                                             "v_xor_b32_e32 %[t0], 0x80000000, %[t0]\n\t"
                                             "v_xor_b32_e32 %[t1], 0x80000000, %[t1]\n\t"
                                             "v_xor_b32_e32 %[t2], 0x80000000, %[t2]\n\t"
                                             "v_xor_b32_e32 %[t3], 0x80000000, %[t3]\n\t"
                                             "v_cndmask_b32_e64 %[e0], %[ninf], %[h0], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e1], %[ninf], %[h1], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e2], %[ninf], %[h2], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e3], %[ninf], %[h3], %[cok]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[e0]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[e1]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[e2]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[e3]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
                                             : [a0] "a"(a0), [a1] "a"(a1), [a2] "a"(a2), [a3] "a"(a3), [h0] "v"(t2l[sc - c + 0]), [h1] "v"(t2l[sc - c + 1]), [h2] "v"(t2l[sc - c + 2]), [h3] "v"(t2l[sc - c + 3]), [ninf] "v"(ninf), [cok] "s"(cokm[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                } else if constexpr (L2) { // (the 128 x 128 tile's accumulators are VGPRs: read in place)
                                    asm volatile("v_fma_f32 %[t0], %[a0], -2.0, %[s0]\n\t"
                                             "v_fma_f32 %[t1], %[a1], -2.0, %[s1]\n\t"
                                             "v_fma_f32 %[t2], %[a2], -2.0, %[s2]\n\t"
                                             "v_fma_f32 %[t3], %[a3], -2.0, %[s3]\n\t"
                                             "v_cndmask_b32_e64 %[e0], %[ninf], %[h0], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e1], %[ninf], %[h1], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e2], %[ninf], %[h2], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e3], %[ninf], %[h3], %[cok]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[e0]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[e1]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[e2]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[e3]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
                                             : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [h0] "v"(t2l[sc - c + 0]), [h1] "v"(t2l[sc - c + 1]), [h2] "v"(t2l[sc - c + 2]), [h3] "v"(t2l[sc - c + 3]), [ninf] "v"(ninf), [cok] "s"(cokm[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                } else {
                                    asm volatile("v_xor_b32_e32 %[t0], 0x80000000, %[a0]\n\t"
                                             "v_xor_b32_e32 %[t1], 0x80000000, %[a1]\n\t"
                                             "v_xor_b32_e32 %[t2], 0x80000000, %[a2]\n\t"
                                             "v_xor_b32_e32 %[t3], 0x80000000, %[a3]\n\t"
                                             "v_cndmask_b32_e64 %[e0], %[ninf], %[h0], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e1], %[ninf], %[h1], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e2], %[ninf], %[h2], %[cok]\n\t"
                                             "v_cndmask_b32_e64 %[e3], %[ninf], %[h3], %[cok]\n\t"
                                             "v_cmp_le_f32_e64 %[m0], %[t0], %[e0]\n\t"
                                             "v_cmp_le_f32_e64 %[m1], %[t1], %[e1]\n\t"
                                             "v_cmp_le_f32_e64 %[m2], %[t2], %[e2]\n\t"
                                             "v_cmp_le_f32_e64 %[m3], %[t3], %[e3]\n\t"
                                             "v_min_u32_e32 %[ad0], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m0]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad1], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m1]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad2], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m2]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_min_u32_e32 %[ad3], %[w], %[dump]\n\t"
                                             "v_cndmask_b32_e64 %[u], 0, %[step], %[m3]\n\t"
                                             "v_add_u32_e32 %[w], %[w], %[u]\n\t"
                                             "v_mov_b32_e32 %[i0], %[idx0]\n\t"
                                             "v_mov_b32_e32 %[i1], %[idx1]\n\t"
                                             "v_mov_b32_e32 %[i2], %[idx2]\n\t"
                                             "v_mov_b32_e32 %[i3], %[idx3]\n\t"
                                             "ds_write2_b32 %[ad0], %[t0], %[i0] offset1:1\n\t"
                                             "ds_write2_b32 %[ad1], %[t1], %[i1] offset1:1\n\t"
                                             "ds_write2_b32 %[ad2], %[t2], %[i2] offset1:1\n\t"
                                             "ds_write2_b32 %[ad3], %[t3], %[i3] offset1:1"
                                             : [t0] "=&v"(t[0]), [t1] "=&v"(t[1]), [t2] "=&v"(t[2]), [t3] "=&v"(t[3]), [ad0] "=&v"(ad[0]), [ad1] "=&v"(ad[1]), [ad2] "=&v"(ad[2]), [ad3] "=&v"(ad[3]), [i0] "=&v"(ix[0]), [i1] "=&v"(ix[1]), [i2] "=&v"(ix[2]), [i3] "=&v"(ix[3]), [m0] "=&s"(mk[0]), [m1] "=&s"(mk[1]), [m2] "=&s"(mk[2]), [m3] "=&s"(mk[3]), [u] "=&v"(u), [w] "+v"(w), [e0] "=&v"(e[0]), [e1] "=&v"(e[1]), [e2] "=&v"(e[2]), [e3] "=&v"(e[3])
                                             : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [h0] "v"(t2l[sc - c + 0]), [h1] "v"(t2l[sc - c + 1]), [h2] "v"(t2l[sc - c + 2]), [h3] "v"(t2l[sc - c + 3]), [ninf] "v"(ninf), [cok] "s"(cokm[b]), [dump] "v"(dump), [step] "v"(step), [idx0] "n"((sc + 0) | (b << 8)), [idx1] "n"((sc + 1) | (b << 8)), [idx2] "n"((sc + 2) | (b << 8)), [idx3] "n"((sc + 3) | (b << 8))
                                             : "memory");
                                }
                            }
                        }
                        if constexpr (DIR == 1) nblk[b] += w - w_b;
                    }
                }
            };
            pass1(std::integral_constant<int, 1>{});
            const uint32_t w1 = w, w1_0 = w_0;
            KNN_TRACE_AT(tile_idx == 5, 126);
            const int n1 = (int)((w1 - w1_0) >> 11);
            // ---- first direction: this wave's survivors -> the queries' lists
            KNN_TRACE_COUNT(__ballot(n1 > C) != 0ull, 60); // (developer build: tiles wave 0 filtered the dense way; slot 59: tiles whose second direction went dense; 58: the wave's largest n1)
#ifdef KNN355_TRACE
            if (p.trace) { // (developer build: the workgroup's largest per-lane survivor counts, first / second direction)
                atomicMax((unsigned long long *)&p.trace[(size_t)blockIdx.x * 128 + 58], (unsigned long long)n1);
            }
#endif
            if (__ballot(n1 > C) != 0ull) {
                filter_tile(trow0, passes_acc, score_of); // (a lane ran out of slots: the dense way, from the accumulators)
            } else {
                // One reservation per lane and query block (LDS atomics) and the lane's first U1 entries, all requested together and
                // waited for ONCE, then stores only; a lane with more entries goes round again.  (Beside the other workgroup's K loop
                // a wave that waits gets back in slowly -- per-workgroup stamps of a CATH-sized symmetric tile: pass 1, straight-line,
                // 1.6 us; this pass with four entries per round and the reservations in front, five waits: 10 us -- so what counts here
                // is the number of waits, not of instructions.)
                constexpr int U1 = 8;
                bool near_full = false;
                int base[NB], run[NB];
                uint2 e[U1];
#pragma unroll
                for (int u = 0; u < U1; u++) e[u] = *(const uint2 *)(smem + OFF1 + (min(u, C) * 256 + tid) * 8);
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const int cnt = (int)(nblk[b] >> 11);
                    base[b] = atomicAdd(&L.s_cnt[qloc(b)], cnt);
                    run[b] = 0;
                    near_full |= cnt > 0 && base[b] + cnt > L.cap - DT;
                }
                for (int j0 = 0;;) {
#pragma unroll
                    for (int u = 0; u < U1; u++) {
                        if (j0 + u < n1) {
                            const int b = (int)((e[u].y >> 8) & 3u);
                            // (b is a runtime number: the block's base and running count picked out of NB registers each)
                            int slot = base[0] + run[0];
#pragma unroll
                            for (int bb = 1; bb < NB; bb++) slot = b == bb ? base[bb] + run[bb] : slot;
#pragma unroll
                            for (int bb = 0; bb < NB; bb++) run[bb] += b == bb;
                            const int ql = (wn * TN + b) * 32 + li;
                            float vv = __uint_as_float(e[u].x);
                            if constexpr (L2) vv = vv < 0.0f ? 0.0f : vv;
                            vv += 0.0f;
                            const int64_t row = trow0 + rowl((int)(e[u].y & 255u));
                            const uint32_t id = p.id_base + (plain_rows ? (uint32_t)row : (uint32_t)view_row(row, p.row_mul, p.vshift));
                            const bool gone = (ragged && row >= p.nb) || (sampled && ((int)(row >> p.vshift) & p.skip_mask) == 0);
                            if (slot < L.cap) L.lists[(size_t)ql * L.cap + slot] = gone ? KEY_PAD : (((uint64_t)f2ord(vv) << 32) | id);
                        }
                    }
                    j0 += U1;
                    if (__ballot(j0 < n1) == 0ull) break;
#pragma unroll
                    for (int u = 0; u < U1; u++) e[u] = *(const uint2 *)(smem + OFF1 + (min(j0 + u, C) * 256 + tid) * 8);
                }
                if (near_full) *L.s_need = 1;
            }
            KNN_TRACE_AT(tile_idx == 5, 127);
            // ---- second direction (symmetric launch, off the diagonal): the rows of this tile as queries
            if constexpr (SYM) {
                if (!second) return; // (workgroup-uniform: the diagonal tile)
                // pass 1 again, with the rows' bounds, into the same slots (this lane's own: the first direction is through with them)
                w = w_0;
                pass1(std::integral_constant<int, 2>{});
                const int n2 = (int)((w - w_0) >> 11);
                if (n2 > C) *s_dense = 1;
                __syncthreads();
                const bool dense = *s_dense != 0;
                KNN_TRACE_AT(tile_idx == 5, 120);
                KNN_TRACE_COUNT(dense, 59);
#ifdef KNN355_TRACE
                if (p.trace) atomicMax((unsigned long long *)&p.trace[(size_t)blockIdx.x * 128 + 57], (unsigned long long)n2);
#endif
                *dense2 = dense;
                if (dense) return; // (the flag is cleared at the start of the next tile)
                // Ranks inside each row's survivors of this tile (LDS atomics), one global reservation per row, then the stores.
                // A lane's first U2 entries stay in REGISTERS from the rank phase to the stores, a lane with more takes the rest
                // through its LDS slots, four at a time.  (First form: every entry read from LDS in both phases, four entries per
                // round in the rank phase and one in the store phase -- 24 + 16 us of a CATH-sized symmetric tile's 59-us epilogue:
                // a dozen waits and a few dozen, beside the other workgroup's K loop.)
                constexpr int U2 = 16;
                uint2 g[U2];
                int rk[U2];
#pragma unroll
                for (int u = 0; u < U2; u++) g[u] = *(const uint2 *)(smem + OFF1 + (min(u, C) * 256 + tid) * 8);
#pragma unroll
                for (int u = 0; u < U2; u++) rk[u] = atomicAdd(&s_cnt2[rowl((int)(g[u].y & 63u))], u < n2 ? 1 : 0);
                for (int j0 = U2; j0 < n2; j0 += 4) { // (the entries past the registers: rank written back into the slot)
                    uint32_t *yp[4], y[4];
                    int r4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        yp[u] = (uint32_t *)(smem + OFF1 + (min(j0 + u, C) * 256 + tid) * 8 + 4);
                        y[u] = *yp[u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) r4[u] = atomicAdd(&s_cnt2[rowl((int)(y[u] & 63u))], j0 + u < n2 ? 1 : 0);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (j0 + u < n2) *yp[u] = y[u] | ((unsigned)r4[u] << 16);
                }
                __syncthreads();
                KNN_TRACE_AT(tile_idx == 5, 121);
                if (tid < DT) {
                    const int c = s_cnt2[tid];
#ifdef KNN355_TRACE
                    if (p.ablate & 64) s_base2[tid] = 0; // (developer build: without the global reservation -- wrong places, right time)
                    else
#endif
                    s_base2[tid] = c > 0 ? (int)atomicAdd(&p.qcnt[trow0 + tid], (uint32_t)c) : 0;
                    s_cnt2[tid] = 0; // (for the next tile)
                }
                __syncthreads();
                KNN_TRACE_AT(tile_idx == 5, 122);
                auto put = [&](float value, int sc, int b, int at_in_row) { // (b is a runtime number: the lane's query of block b, spelled out)
                    const int rl = rowl(sc);
                    float vv = value;
                    if constexpr (L2) vv = vv < 0.0f ? 0.0f : vv;
                    const uint32_t id = p.id_base + (uint32_t)(q0 + (wn * TN + b) * 32 + li);
                    if (at_in_row < p.qcap) p.qlist[(size_t)(trow0 + rl) * p.qcap + at_in_row] = ((uint64_t)f2ord(vv + 0.0f) << 32) | id;
                    else *p.fail = 1;
                };
                int bs[U2];
#pragma unroll
                for (int u = 0; u < U2; u++) bs[u] = s_base2[rowl((int)(g[u].y & 63u))];
#ifdef KNN355_TRACE
                if (p.ablate & 256) return; // (developer build: without the stores)
#endif
#pragma unroll
                for (int u = 0; u < U2; u++)
                    if (u < n2) put(__uint_as_float(g[u].x), (int)(g[u].y & 255u), (int)((g[u].y >> 8) & 3u), bs[u] + rk[u]);
                for (int j = U2; j < n2; j++) {
                    const uint2 e = *(const uint2 *)(smem + OFF1 + (j * 256 + tid) * 8);
                    const int sc = (int)(e.y & 255u);
                    put(__uint_as_float(e.x), sc, (int)((e.y >> 8) & 3u), s_base2[rowl(sc)] + (int)(e.y >> 16));
                }
            }
        };

        // ---- epilogue: threshold filter + append ----
#ifdef KNN355_TRACE
        if (p.ablate & 4) { // all MFMA results of this wave have landed before the "K loop ended" stamp is taken
            float sink = 0.0f;
#pragma unroll
            for (int b = 0; b < NB; b++) sink += accv(b, NS - 1) + accv(b, NS / 2 - 1);
            if (sink == 12345.678f) KNN_TRACE(61);
            KNN_TRACE(1 + 2 * tile_idx);
        }
        if (!(p.ablate & 1))
#endif
        {
            if (SPARSE && (BIGT || SYM || p.sparse_epi)) {
                // The next tile's first K step goes out NOW, into buffer 0: behind the K loop's last barrier no wave reads a
                // staging buffer any more (the last sub-step runs from registers), the epilogue below keeps its slots in
                // buffer 1, and its ~8 us cover the bytes' way from L2 / HBM -- the tile prologue found them exposed (2-3 us
                // per tile with nothing else on the CU).
                if constexpr (!BIGT) __syncthreads(); // (the generic K loop's last step reads its fragments from LDS: every wave is through before the slots below are written)
                if (BIGT && next_tile >= 0) {
                    const int64_t nrow0 = (int64_t)next_tile * DT;
#pragma unroll
                    for (int n = 0; n < NI; n++) {
                        tsrc[n] = is_db[n] ? srcp[n] + view_row(min(nrow0 + rloc[n], p.nb - 1), p.row_mul, p.vshift) * p.dp : srcp[n];
                        stage_issue<false>(tsrc[n], stage0 + lds_off[n]);
                    }
                    staged0 = true;
                }
                if constexpr (SPARSE) sparse_epilogue(row0, off_diag, &dir2_dense);
            } else if (!deferred) filter_tile(row0, passes_acc, score_of);
        }
        KNN_TRACE(64 + 2 * tile_idx);
        if constexpr (CAN_PUB) {
            if (pub_on && (tile_idx == 1 || (deferred && (paired ? *s_next : next_tile) < 0))) {
                // the parked first tile (every lane reads back what it stored itself); a chunk of a single tile filters
                // it on the spot
                const float4 *park = (const float4 *)p.defer + (size_t)blockIdx.x * NV4 * 256 + tid;
                float4 pv[NV4];
#pragma unroll
                for (int j = 0; j < NV4; j++) pv[j] = park[(size_t)j * 256];
                auto parked = [&](int b, int sc, float) -> float {
                    const float4 w = pv[b * (NS / 4) + (sc >> 2)];
                    return (sc & 3) == 0 ? w.x : ((sc & 3) == 1 ? w.y : ((sc & 3) == 2 ? w.z : w.w));
                };
                filter_tile(first_row0, [&](int b, int sc, float thr, float) -> bool { return parked(b, sc, 0.0f) <= thr; }, parked);
            }
        }
        if constexpr (SYM) {
#ifdef KNN355_TRACE
            if (dir2_dense && !(p.ablate & 8)) {
#else
            if (dir2_dense) {
#endif
                // the same scores, read the other way round: row R of this tile is a query, the resident tile's rows
                // are its candidates.  A (register, lane half) pair of one wave is one row R and 32 candidates.
                auto score = [&](int a, int b, int r, float xnq) -> float {
                    if constexpr (L2) {
                        const float ynr = s_yn[(wm * TM + a) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)];
                        const float v = __builtin_fmaf(-2.0f, acc[a][b][r], xnq + ynr);
                        return v < 0.0f ? 0.0f : v;
                    } else {
                        return -acc[a][b][r];
                    }
                };
                float xnq[TN];
                bool cand_ok[TN];
#pragma unroll
                for (int b = 0; b < TN; b++) {
                    const int64_t q = q0 + (wn * TN + b) * 32 + li;
                    // a candidate must be a real row that the sample pass has not already handed on
                    cand_ok[b] = q < p.nq && !(p.skip_mask >= 0 && ((int)(q >> p.vshift) & p.skip_mask) == 0);
                    xnq[b] = 0.0f;
                    if constexpr (L2) xnq[b] = p.xn[q < p.nq ? q : 0];
                }
                // pass A: survivors per row (this wave's 64 rows x its 64 candidates)
#pragma unroll
                for (int a = 0; a < TM; a++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int rl = (wm * TM + a) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                        const float thr2 = s_thr2[rl];
                        int c_lo2 = 0, c_hi2 = 0;
#pragma unroll
                        for (int b = 0; b < TN; b++) {
                            const float v = score(a, b, r, xnq[b]);
                            const uint64_t m = __ballot(v <= thr2 && cand_ok[b]);
                            c_lo2 += __popc((uint32_t)m);
                            c_hi2 += __popc((uint32_t)(m >> 32));
                        }
                        if (li == 0) s_cnt2[wn * DT + rl] = lh ? c_hi2 : c_lo2;
                    }
                }
                __syncthreads();
                if (tid < DT) {
                    int c = 0;
#pragma unroll
                    for (int w2 = 0; w2 < WN; w2++) c += s_cnt2[w2 * DT + tid];
                    s_base2[tid] = c > 0 ? (int)atomicAdd(&p.qcnt[row0 + tid], (uint32_t)c) : 0;
                }
                __syncthreads();
                // pass B: the keys, behind the survivors of the waves with a lower query half
#pragma unroll
                for (int a = 0; a < TM; a++) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int rl = (wm * TM + a) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                        const float thr2 = s_thr2[rl];
                        int pos = s_base2[rl];
                        for (int w2 = 0; w2 < wn; w2++) pos += s_cnt2[w2 * DT + rl];
                        uint64_t *dst = p.qlist + (size_t)(row0 + rl) * p.qcap;
#pragma unroll
                        for (int b = 0; b < TN; b++) {
                            const float v = score(a, b, r, xnq[b]);
                            const bool pass = v <= thr2 && cand_ok[b];
                            const uint64_t m = __ballot(pass);
                            const uint32_t mh = lh ? (uint32_t)(m >> 32) : (uint32_t)m;
                            if (pass) {
                                const int at = pos + __popc(mh & ((1u << li) - 1u));
                                const uint32_t id = p.id_base + (uint32_t)(q0 + (wn * TN + b) * 32 + li);
                                if (at < p.qcap) dst[at] = ((uint64_t)f2ord(v + 0.0f) << 32) | id;
                                else *p.fail = 1;
                            }
                            pos += __popc(mh);
                        }
                    }
                }
            }
        }
        __syncthreads();
        KNN_TRACE(65 + 2 * tile_idx);
        // ---- compaction of lists that could overflow on the next tile ----
        if (paired) next_tile = *s_next;
        const bool last_tile = next_tile < 0;
        if (*L.s_need || last_tile) lists_compact<QT>(L, smem, DT, last_tile, tid);
        KNN_TRACE(2 + 2 * tile_idx);
        if constexpr (!BIGT && KNN355_EPI_PRIO > 0) __builtin_amdgcn_s_setprio(0);
        cur_tile = next_tile;
        tile_idx++;
    }
    if constexpr (!BIGT && KNN355_EPI_PRIO > 0) __builtin_amdgcn_s_setprio(KNN355_EPI_PRIO); // (the flush is epilogue work too)
    lists_flush<QT>(L, s_base, q0, p.nq, p.qlist, p.qcnt, p.qcap, tid, SYM ? p.fail : nullptr);
    KNN_TRACE(63);
}

// ---------------------------------------------------------------------------
// final selection: one workgroup per query picks the best k of that query's candidate keys,
// sorts them and emits D / I, sorted keys, or the seed of the enclosing search.
//   input   contiguous: keys in + q * in_stride, count cnt[q] (clamped to cap) or n_fixed
//           list-major: an all-gather buffer [lm_lists][nq][lm_k] (KEY_PAD = no key)
//   method  keys are distinct 64-bit numbers (score word << 32 | row id), so f(X) = #(keys < X)
//           takes every value: a bracket search on X -- probes alternate between linear
//           interpolation of the target rank and bisection -- stops at the first X with
//           k <= f(X) <= kmax (3-6 probes; heavy ties on the score word are spread by the id
//           word).  Up to 256 * R keys live in registers (one compare + add per key and probe);
//           longer arrays are re-read from L2 on every probe.  The <= kmax survivors are packed
//           into LDS and sorted by the whole workgroup.
// ---------------------------------------------------------------------------
struct SelectParams {
    const uint64_t *in;
    int64_t in_stride;
    const uint32_t *cnt;
    int n_fixed, cap;
    int n_expect;         // host's estimate of a typical count (0: cap) -- picks the build, any count is handled
    // end-of-search reset (streaming searches with the tile-minimum seed): the final selection leaves the level's state the
    // way init_level_kernel would -- the next search of the same shape needs no launch in front of its scan
    uint32_t *rs_gthr, *rs_qcnt, *rs_qthr, *rs_arrive, *rs_pair;
    uint64_t *rs_pub;
    int rs_nslots, rs_pub_n, rs_narrive, rs_npairs;
    int seg_len, nseg;    // segment pass (nseg > 0): workgroup (q, s) selects among keys [s * seg_len, (s + 1) * seg_len) of query q
                          // and writes output row q * nseg + s
    int lm_lists, lm_k;
    int64_t nq;
    int k, metric;
    uint64_t *out_keys;   // sorted keys (may be null): k per query at out_keys + q * out_stride, padded with KEY_PAD up to out_fill
    int64_t out_stride;
    int out_fill;
    float *D;             // final distances / ids (may be null)
    int64_t *I;
    // seeding the enclosing search from this (sample) result: its candidate count, its running
    // threshold = the score word of the seed_j-th key (no bound if there are fewer), and -- for a
    // statistical seed -- the same word as the bound its own result is verified against
    uint32_t *seed_cnt, *seed_gthr, *seed_qthr;
    int seed_j, seed_stat;
    int64_t seed_nslots;  // threshold slots of the enclosing search (>= nq: the last query tile is padded)
    const uint32_t *qthr; // verification: a k-th score word above qthr[q] means the statistical threshold was too tight
    int *fail;
};

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t y = __shfl_xor(x, o, 64);
        x = y < x ? y : x;
    }
    return x;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t y = __shfl_xor(x, o, 64);
        x = y > x ? y : x;
    }
    return x;
}

// R keys per thread in registers, NT threads.  SEED: this is a seed sample's selection -- no sorted
// output; it hands the enclosing search the sample's rows that beat T = the sample's seed_j-th
// score (unsorted candidates, at most k of them), T itself as the running threshold and, for a
// statistical seed, as the bound the enclosing result is verified against.
template <int R, int NT, bool SEED>
__global__ __launch_bounds__(NT) void select_topk_kernel(SelectParams p)
{
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t *sb = (uint64_t *)smem; // [P] survivors, sorted in place (final mode)
    __shared__ int s_red[2][NW];
    __shared__ int s_scan[NW];
    __shared__ uint64_t s_mm[2][NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const int k = p.k;
    int n = p.lm_lists > 0 ? p.lm_lists * p.lm_k : (p.cnt ? (int)min(p.cnt[q], (uint32_t)p.cap) : p.n_fixed);
    int seg_base = 0;
    int64_t orow = q; // output row
    if (p.nseg > 0) {
        seg_base = (int)blockIdx.y * p.seg_len;
        n = max(0, min(n - seg_base, p.seg_len));
        orow = q * p.nseg + blockIdx.y;
    }
    auto load = [&](int idx) -> uint64_t {
        if (p.lm_lists > 0) {
            const int l = idx / p.lm_k, j = idx - l * p.lm_k;
            return ((gptr_u64)p.in)[((size_t)l * p.nq + q) * p.lm_k + j];
        }
        return ((gptr_u64)p.in)[(size_t)q * p.in_stride + seg_base + idx];
    };
    const bool in_regs = n <= NT * R;
    // memory path (more keys than the registers hold): several independent loads in flight per thread
    auto for_each_mem = [&](auto &&fn) {
        constexpr int U = NT >= 1024 ? 4 : 8; // (the 1024-thread build has 128 registers per lane)
        int idx = tid;
        for (; idx + (U - 1) * NT < n; idx += U * NT) {
            uint64_t v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = load(idx + u * NT);
#pragma unroll
            for (int u = 0; u < U; u++) fn(v[u]);
        }
        for (; idx < n; idx += NT) fn(load(idx));
    };
    uint64_t key[R];
    uint64_t mn = KEY_PAD, mx = 0;
    int real = 0;
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int idx = r * NT + tid;
            uint64_t v = KEY_PAD;
            if (r * NT < n) { // (uniform) registers past the end of a short array cost nothing
                v = load(idx < n ? idx : n - 1);
                if (idx >= n) v = KEY_PAD;
            }
            key[r] = v;
            if (v != KEY_PAD) {
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
                real++;
            }
        }
    } else {
        for_each_mem([&](uint64_t v) {
            if (v != KEY_PAD) {
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
                real++;
            }
        });
    }
    mn = wave_min_u64(mn);
    mx = wave_max_u64(mx);
    real = wave_sum(real);
    if (lane == 0) {
        s_mm[0][wave] = mn;
        s_mm[1][wave] = mx;
        s_scan[wave] = real;
    }
    __syncthreads();
    real = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; w2++) {
        mn = min(mn, s_mm[0][w2]);
        mx = max(mx, s_mm[1][w2]);
        real += s_scan[w2];
    }
    __syncthreads(); // s_scan is reused below

    int par = 0;
    auto count_lt = [&](uint64_t X) -> int {
        int c = 0;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r * NT < n) c += key[r] < X ? 1 : 0; // (uniform: registers past the end hold padding)
        } else {
            for_each_mem([&](uint64_t v) { c += v < X ? 1 : 0; });
        }
        c = wave_sum(c);
        if (lane == 0) s_red[par][wave] = c;
        __syncthreads();
        c = 0;
#pragma unroll
        for (int w2 = 0; w2 < NW; w2++) c += s_red[par][w2];
        par ^= 1; // the next probe writes the other set: one barrier per probe is enough
        return c;
    };
    // X with klo <= #(keys < X) <= khi, or KEY_PAD when no more than khi real keys exist; *cnt_out = that count
    auto bracket = [&](int klo, int khi, int *cnt_out) -> uint64_t {
        if (real <= khi) {
            *cnt_out = real;
            return KEY_PAD;
        }
        uint64_t lo = mn, hi = mx + 1; // f(lo) = 0 < klo, f(hi) = real > khi
        int f_lo = 0, f_hi = real;
        const double target = 0.5 * (double)(klo + khi);
        bool interpolate = true;
        for (int it = 0; it < 200; it++) {
            const uint64_t w = hi - lo;
            if (w <= 1) break; // cannot happen while the keys are distinct
            uint64_t X;
            if (interpolate) X = lo + (uint64_t)((double)w * ((target - (double)f_lo) / (double)(f_hi - f_lo)));
            else X = lo + (w >> 1);
            X = max(lo + 1, min(X, hi - 1));
            interpolate = !interpolate;
            const int c = count_lt(X);
            if (c < klo) {
                lo = X;
                f_lo = c;
            } else if (c > khi) {
                hi = X;
                f_hi = c;
            } else {
                *cnt_out = c;
                return X;
            }
        }
        *cnt_out = f_hi; // duplicate keys (never in a well-formed input): keep the tie, the caller clamps
        return hi;
    };
    // position of this thread's first survivor in the packed output (workgroup-wide exclusive scan)
    auto survivors_base = [&](uint64_t T) -> int {
        int mine = 0;
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r * NT < n) mine += key[r] < T ? 1 : 0;
        } else {
            for_each_mem([&](uint64_t v) { mine += v < T ? 1 : 0; });
        }
        const int incl = wave_inclusive_scan(mine);
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        int pos = incl - mine;
        for (int w2 = 0; w2 < wave; w2++) pos += s_scan[w2];
        return pos;
    };
    auto pack = [&](uint64_t T, int pos, uint64_t *dst, int limit) {
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r * NT < n && key[r] < T) {
                    if (pos < limit) dst[pos] = key[r];
                    pos++;
                }
        } else {
            for_each_mem([&](uint64_t v) {
                if (v < T) {
                    if (pos < limit) dst[pos] = v;
                    pos++;
                }
            });
        }
    };

    if constexpr (SEED) {
        // T = the score word of a key of rank r in [seed_j, 1.25 seed_j] (any rank >= seed_j is a bound at least as
        // safe as the seed_j-th, and a loose target converges in 3-5 probes where the exact rank takes dozens);
        // the sample hands on every row that beats T (score word <= T: ties included, the main pass admits
        // them too) -- at most kmax of them, the k best if more rows than that tie into the bound
        const int kmax = k + max(k >> 2, 32);
        int cj = 0;
        uint32_t T = 0xFFFFFFFFu;
        uint64_t Tkey = KEY_PAD;
        int cnt = real;
        if (real >= p.seed_j) {
            const uint64_t Xj = bracket(p.seed_j, p.seed_j + max(p.seed_j >> 2, 8), &cj);
            // the largest key below Xj carries the bound
            uint64_t m = 0;
            if (in_regs) {
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (r * NT < n) m = (key[r] < Xj && key[r] > m) ? key[r] : m;
            } else {
                for_each_mem([&](uint64_t v) { m = (v < Xj && v > m) ? v : m; });
            }
            m = wave_max_u64(m);
            if (lane == 0) s_mm[0][wave] = m;
            __syncthreads();
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) m = max(m, s_mm[0][w2]);
            T = (uint32_t)(m >> 32);
            Tkey = T == 0xFFFFFFFFu ? KEY_PAD : ((uint64_t)(T + 1u) << 32);
            cnt = Tkey == KEY_PAD ? real : count_lt(Tkey);
            if (cnt > kmax) {
                // (heavy ties into the bound) the best k..kmax of them.  For an EXACT seed k is the caller's k and no row
                // beyond the k-th can matter.  A STATISTICAL seed searched its sample with k_sample ~ 1.25 j << the caller's
                // k: sample rows cut here are masked out of the main scan, and one of them could belong to the caller's
                // top k while the verification (k-th <= T) still passes.  Too many ties to hand on: the estimate is
                // withdrawn, the caller repeats the search without it.
                if (p.seed_stat && tid == 0 && p.fail) *p.fail = 1;
                Tkey = bracket(k, kmax, &cnt);
            }
        }
        cnt = min(cnt, kmax);
        const int pos = survivors_base(Tkey);
        pack(Tkey, pos, p.out_keys + (size_t)q * p.out_stride, kmax);
        if (tid == 0) {
            p.seed_cnt[q] = (uint32_t)cnt;
            p.seed_gthr[q] = T;
            p.seed_qthr[q] = p.seed_stat ? T : 0xFFFFFFFFu;
        }
        // threshold slots of the padding queries of the last query tile (they mirror the last query and append nothing)
        if (q == 0)
            for (int64_t i = p.nq + tid; i < p.seed_nslots; i += NT) p.seed_gthr[i] = 0xFFFFFFFFu;
        return;
    } else {
        const int kmax = k + max(k >> 2, 32);
        int cnt = real;
        const uint64_t T = bracket(k, kmax, &cnt); // survivors: keys < T (KEY_PAD: every real key)
        int P = 64;
        while (P < cnt) P <<= 1;
        const int Pcap = 1 << (32 - __builtin_clz(kmax - 1)); // what the host sized the LDS for: next_pow2(kmax)
        if (P > Pcap) P = Pcap;
        const int pos = survivors_base(T);
        pack(T, pos, sb, P);
        for (int i = cnt + tid; i < P; i += NT) sb[i] = KEY_PAD;
        __syncthreads();
        if (NT == 64 && P <= 1024) { // (one wave per query: the sort stays in registers)
            switch (P) {
            case 64: wave_bitonic_sort_regs<1>(sb, lane); break;
            case 128: wave_bitonic_sort_regs<2>(sb, lane); break;
            case 256: wave_bitonic_sort_regs<4>(sb, lane); break;
            case 512: wave_bitonic_sort_regs<8>(sb, lane); break;
            default: wave_bitonic_sort_regs<16>(sb, lane); break;
            }
            __syncthreads();
        } else {
            wg_bitonic_sort(sb, P, tid, NT);
        }
        cnt = min(cnt, P);
        const int have = min(cnt, k); // real keys among the first k
        for (int i = tid; i < max(k, p.out_fill); i += NT) {
            const uint64_t v = i < have ? sb[i] : KEY_PAD;
            if (p.out_keys) p.out_keys[(size_t)orow * p.out_stride + i] = v;
            if (p.D && i < k) {
                const size_t o = (size_t)q * k + i;
                if (v == KEY_PAD) {
                    p.D[o] = p.metric == KNN_METRIC_INNER_PRODUCT ? -FLT_MAX : FLT_MAX;
                    p.I[o] = -1;
                } else {
                    const float f = ord2f((uint32_t)(v >> 32));
                    p.D[o] = p.metric == KNN_METRIC_INNER_PRODUCT ? -f : f;
                    p.I[o] = (int64_t)(uint32_t)v;
                }
            }
        }
        if (tid == 0 && p.qthr && p.fail) {
            const uint32_t kth = cnt >= k ? (uint32_t)(sb[k - 1] >> 32) : 0xFFFFFFFFu;
            if (kth > p.qthr[q]) *p.fail = 1;
        }
        if (p.rs_gthr) { // (this query's candidates have been read: barriers above)
            if (tid == 0) {
                p.rs_qcnt[q] = 0u;
                p.rs_qthr[q] = 0xFFFFFFFFu;
                p.rs_gthr[q] = 0xFFFFFFFFu;
            }
            for (int j = tid; j < p.rs_pub_n; j += NT) p.rs_pub[(size_t)q * p.rs_pub_n + j] = KEY_PAD;
            if (q == 0) {
                for (int64_t i = p.nq + tid; i < p.rs_nslots; i += NT) p.rs_gthr[i] = 0xFFFFFFFFu;
                for (int i = tid; i < p.rs_narrive; i += NT) p.rs_arrive[i] = 0u;
                for (int i = tid; i <= p.rs_npairs; i += NT) p.rs_pair[i] = i < p.rs_npairs ? 2u : 0u;
            }
        }
    }
}

// (re)initialises a level's per-query state in one launch: running thresholds "no bound", empty candidate
// arrays, no verification bound
__global__ void init_level_kernel(uint32_t *__restrict__ gthr, int64_t nslots, uint32_t *__restrict__ qcnt,
                                  uint32_t *__restrict__ qthr, int64_t nq, int *__restrict__ flag,
                                  uint64_t *__restrict__ pub = nullptr, int64_t npub = 0, uint32_t *__restrict__ arrive = nullptr,
                                  int64_t narrive = 0, uint32_t *__restrict__ pair_ctr = nullptr, int64_t npairs = 0)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pair_ctr) // paired walk: tickets 0 and 1 are the two workgroups' first tiles
        for (int64_t j = i; j <= npairs; j += (int64_t)gridDim.x * blockDim.x) pair_ctr[j] = j < npairs ? 2u : 0u; // ([npairs]: the pool's)
    if (pub) { // tile-minimum seed: nothing published yet
        for (int64_t j = i; j < npub; j += (int64_t)gridDim.x * blockDim.x) pub[j] = KEY_PAD;
        if (i < narrive) arrive[i] = 0u;
    }
    if (gthr && i < nslots) gthr[i] = 0xFFFFFFFFu;
    if (qcnt && i < nq) {
        qcnt[i] = 0u;
        qthr[i] = 0xFFFFFFFFu;
    }
    if (flag && i == 0) *flag = 0;
}

// ---------------------------------------------------------------------------
// gather distances: one wave per (query, candidate) pair, same fma chain order
// ---------------------------------------------------------------------------
__global__ void pair_distance_kernel(const float *__restrict__ xb, const float *__restrict__ yn,
                                     const float *__restrict__ xq, const float *__restrict__ xn,
                                     int dp, int metric, int64_t npairs,
                                     const int32_t *__restrict__ pair_q, const int64_t *__restrict__ pair_r,
                                     float *__restrict__ out)
{
    // Each lane runs whole chains for separate pairs (the chain is sequential in k).
    int64_t pidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= npairs) return;
    const float *q = xq + (int64_t)pair_q[pidx] * dp;
    const float *y = xb + pair_r[pidx] * dp;
    float acc = 0.0f;
    for (int k0 = 0; k0 < dp; k0 += 8) {
        f32x4 q0 = *(const f32x4 *)(q + k0), q1 = *(const f32x4 *)(q + k0 + 4);
        f32x4 y0 = *(const f32x4 *)(y + k0), y1 = *(const f32x4 *)(y + k0 + 4);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            acc = __builtin_fmaf(q0[m], y0[m], acc);
            acc = __builtin_fmaf(q1[m], y1[m], acc);
        }
    }
    if (metric == KNN_METRIC_L2) {
        float t = xn[pair_q[pidx]] + yn[pair_r[pidx]];
        float v = __builtin_fmaf(-2.0f, acc, t);
        acc = v < 0.0f ? 0.0f : v;
    }
    out[pidx] = acc;
}

// ===========================================================================
// host side
// ===========================================================================
static thread_local std::string g_err;
static thread_local int g_device = 0;

static int set_err(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return set_err(KNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

// Device scratch buffers are recycled through a small per-device pool: the reference's scripts
// build a fresh index per embedding file (cath/search.py:20-24), and hipMalloc/hipFree of the
// gigabyte-sized candidate-list workspace would otherwise dominate their end-to-end time.
struct DevPool {
    struct Item { void *p; size_t bytes; int device; };
    std::mutex mu;
    std::vector<Item> items;
    size_t total = 0;
    static constexpr size_t kMaxBytes = 8ull << 30;
    void *take(size_t need, int device, size_t *got)
    {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (int i = 0; i < (int)items.size(); i++)
            if (items[i].device == device && items[i].bytes >= need && items[i].bytes <= need * 4 + (1u << 20) &&
                (best < 0 || items[i].bytes < items[best].bytes))
                best = i;
        if (best < 0) return nullptr;
        void *p = items[best].p;
        *got = items[best].bytes;
        total -= items[best].bytes;
        items.erase(items.begin() + best);
        return p;
    }
    void give(void *p, size_t bytes, int device)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (total + bytes <= kMaxBytes && items.size() < 64) {
                items.push_back({p, bytes, device});
                total += bytes;
                return;
            }
        }
        (void)hipFree(p);
    }
};
static DevPool g_pool;

extern "C" int64_t knn_trim(void)
{
    std::vector<DevPool::Item> items;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        items.swap(g_pool.items);
        g_pool.total = 0;
    }
    int64_t freed = 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &it : items) {
        if (hipSetDevice(it.device) == hipSuccess && hipFree(it.p) == hipSuccess) freed += (int64_t)it.bytes;
    }
    (void)hipSetDevice(cur);
    return freed;
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int device = 0;
    // done / s: the owner's "everything my earlier calls enqueued" event and the stream of the call in progress.  A buffer
    // that grows goes back to the pool, which may hand it to another handle right away: whatever still uses it must have
    // finished.  With both given that is a host wait for the OWNER's work only (its previous calls through `done`, this
    // call's launches through `s`); rounds 1-2 called hipDeviceSynchronize() here, which also stalled the other lane and
    // every caller stream at the first searches of a handle.  Without them (users unknown): the whole device.
    int ensure(size_t need, hipEvent_t done = nullptr, hipStream_t s = nullptr)
    {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (need <= bytes && cur == device) return 0;
        if (p) {
            (void)hipSetDevice(device);
            if (done && cur == device) {
                (void)hipEventSynchronize(done);
                (void)hipStreamSynchronize(s);
            } else {
                (void)hipDeviceSynchronize();
            }
            (void)hipSetDevice(cur);
        }
        release();
        device = cur;
        size_t got = 0;
        if (void *q = g_pool.take(need, device, &got)) {
            p = q;
            bytes = got;
            return 0;
        }
        size_t want = need + need / 4;
        if (hipMalloc(&p, want) != hipSuccess) {
            if (hipMalloc(&p, need) != hipSuccess) { p = nullptr; return -1; }
            want = need;
        }
        bytes = want;
        return 0;
    }
    void release()
    {
        if (p) g_pool.give(p, bytes, device);
        p = nullptr;
        bytes = 0;
    }
};

#ifdef KNN355_TRACE
static DevBuf g_trace_buf;
static int g_trace_grid = 0;
// developer build only: the stamps of the last traced scan launch, [grid][128]; returns the grid
extern "C" int knn_dev_trace_read(unsigned long long *out, int max_wgs)
{
    const int n = std::min(max_wgs, g_trace_grid);
    if (n > 0 && hipMemcpy(out, g_trace_buf.p, (size_t)n * 128 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return g_trace_grid;
}
#endif

// per seed-recursion level: compact candidate arrays [nq][qcap] + their fill, the shared running
// thresholds and the verification bounds of a statistically seeded pass
struct LevelBufs {
    DevBuf qlist, qcnt, gthr, qthr;
    DevBuf pub, arrive; // tile-minimum seed: published keys [nslots][rounds * nchunks], arrival counters [nqtiles]
    DevBuf pair_ctr;    // paired walk: ticket counter of every pair of workgroups
    uint64_t clean_sig = 0; // != 0: the last search's selection reset this state for a search of exactly this shape
};

// HIP streams are recycled: creating one costs a third of a millisecond, and the reference's scripts build a fresh
// index per embedding file (cath/search.py:20-24).  A stream goes back to the pool fully drained.
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> free_list;
    hipStream_t take(int device)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].first == device) {
                    hipStream_t s = free_list[i].second;
                    free_list.erase(free_list.begin() + i);
                    return s;
                }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        return s;
    }
    void give(int device, hipStream_t s)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (free_list.size() < 16) free_list.push_back({device, s});
        else (void)hipStreamDestroy(s);
    }
};
static StreamPool g_streams;

struct knn_index_s {
    int d = 0, dp = 0, metric = 0, device = 0;
    int num_cus = 256;
    int64_t ntotal = 0, cap_rows = 0;
    bool is_view = false; // shares another handle's database (knn_flat_view): read-only, frees nothing of it
    // storage generation: bumped whenever xb/yn are reallocated, reset or freed.  A view remembers the
    // value it was made at and refuses to search once its parent's storage has moved on.
    std::shared_ptr<std::atomic<int64_t>> storage_gen = std::make_shared<std::atomic<int64_t>>(0);
    int64_t view_gen = 0;
    float *xb = nullptr; // [cap_rows][dp]
    bool approx16 = false;  // the scan multiplies bf16 copies of rows and queries (HNSW's coarse entry index: approximate on purpose)
    bool keep16 = false;    // bf16 copies of the rows are kept beside the fp32 ones (HNSW storage: the beam walks on them); scans stay fp32
    DevBuf xb16, ws_q16;    // approx16: [cap_rows][dp] bf16 rows, [nq][dp] bf16 queries
    float *yn = nullptr; // [cap_rows + pad]
    size_t xb_bytes = 0, yn_bytes = 0; // allocation sizes (may exceed the row capacity: pooled)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;               // recorded behind everything a search of this handle enqueued (DevBuf::ensure)
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // the pair of the most recent scan launch
    static const int RING = 64;
    hipEvent_t ring0[RING] = {nullptr}, ring1[RING] = {nullptr};
    int64_t nlaunches = 0;
    std::mutex mu;
    DevBuf ws_q, ws_qn, ws_lists, ws_D, ws_I, ws_tmp, ws_tmp2;
    DevBuf ws_D1, ws_I1, ws_tmp3; // second set for the pipelined host search
    DevBuf ws_flag;               // [0]: a statistically seeded search failed its verification
    DevBuf ws_sym;                // work table of a symmetric all-vs-all launch
    int sym_tiles = -1, sym_run = 0; // ... which is the table for this many tiles (run length sym_run, sym_items entries)
    int sym_ts = 0;               // ... of this many rows each
    int64_t sym_items = 0;
    int sym_groups = 1;           // ... in this many groups of query tiles, one launch each (results streamed to the host group by group)
    std::vector<int64_t> sym_gstart; // [sym_groups + 1] first entry of each group
    DevBuf ws_qdiff;              // difference builds: the queries interleaved by pairs
    DevBuf ws_defer;              // tile-minimum seed: the parked first-tile scores of every workgroup
    DevBuf ws_turn;               // batch launches: one word per CU (the resident workgroups take turns in their K loops)
    void *turn_zeroed = nullptr;  // the ws_turn allocation that has been cleared (every holder gives its word back: a launch leaves them all zero)
    int64_t sym_searches = 0;     // self-searches served by the symmetric path
    static const int MAX_LEVELS = 8;
    LevelBufs ws_level[MAX_LEVELS]; // per seed-recursion level
    int last_seed_stride = 0, last_seed_stat = 0;
    int64_t last_sample_rows = 0; // rows the seed sample took out of the main pass of the last search
    int64_t stat_redo = 0;        // searches repeated because the statistical threshold was too tight
    int *flag_host = nullptr;     // pinned: [slot] copy of ws_flag behind each batch of a host search
    // tuning + introspection
    int force_qt = 0, force_chunks = 0, flags = 0;
    int64_t batch_nq = 0; // the caller's whole batch while a search works through it in pieces (BatchScope); 0: outside a search
    int64_t batch_hint = 0; // knn_flat_set_batch: the calls that follow are pieces of a batch of this many queries (0: each its own)
    int pub_rounds_force = 0; // tile-minimum seed: rounds of publications (0: the host's choice)
    std::string last_kernel;
    int last_qt = 0, last_dt = 0, last_chunks = 0, last_grid = 0;
    float last_ms = 0.f;
};

static int ensure_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_err(KNN_ERR_NO_DEVICE, "no HIP device available (libknn355 has no CPU fallback)");
    if (device < 0 || device >= n) return set_err(KNN_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return 0;
}

extern "C" const char *knn_last_error(void) { return g_err.c_str(); }
extern "C" const char *knn_version(void) { return "knn355 0.1 (gfx950)"; }
extern "C" int knn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int knn_init(int device)
{
    int rc = ensure_device(device);
    if (rc) return rc;
    g_device = device;
    return 0;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- normalize ------------------------------------------------------------
static int normalize_dev_impl(float *x, int64_t n, int d, int64_t stride, hipStream_t s)
{
    if (n == 0) return 0;
    unsigned grid = (unsigned)((n + 63) / 64);
    if ((stride % 4) == 0 && (((uintptr_t)x) % 16) == 0)
        hipLaunchKernelGGL(normalize_rows_kernel<true>, dim3(grid), dim3(64), 0, s, x, n, d, stride);
    else
        hipLaunchKernelGGL(normalize_rows_kernel<false>, dim3(grid), dim3(64), 0, s, x, n, d, stride);
    HIP_TRY(hipGetLastError());
    return 0;
}
static int norms_dev_impl(const float *x, int64_t n, int d, int64_t stride, float *out, hipStream_t s)
{
    if (n == 0) return 0;
    unsigned grid = (unsigned)((n + 63) / 64);
    if ((stride % 4) == 0 && (((uintptr_t)x) % 16) == 0)
        hipLaunchKernelGGL(norm_rows_kernel<true>, dim3(grid), dim3(64), 0, s, x, n, d, stride, out);
    else
        hipLaunchKernelGGL(norm_rows_kernel<false>, dim3(grid), dim3(64), 0, s, x, n, d, stride, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int knn_normalize_l2_dev(float *x_dev, int64_t n, int32_t d, void *stream)
{
    if (n < 0 || d <= 0) return set_err(KNN_ERR_INVALID, "normalize_l2: bad shape");
    int rc = normalize_dev_impl(x_dev, n, d, d, (hipStream_t)stream);
    if (rc) return rc;
    if (!stream) HIP_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

static void *pool_alloc(size_t bytes, int device, size_t *got);
extern "C" int knn_normalize_l2(float *x_host, int64_t n, int32_t d)
{
    if (n < 0 || d <= 0) return set_err(KNN_ERR_INVALID, "normalize_l2: bad shape");
    if (n == 0) return 0;
    if (!x_host) return set_err(KNN_ERR_INVALID, "normalize_l2: null pointer");
    int rc = ensure_device(g_device);
    if (rc) return rc;
    // Two pooled device buffers (no hipMalloc / hipFree per call: the reference's scripts normalise per embedding file)
    // and two streams taking turns in chunks of <= 64 MB: the download of chunk i overlaps the upload of chunk i + 1 (PCIe
    // is full duplex).  (A caller that adds the rows to an index anyway should add them raw and use knn_flat_normalize_rows
    // + knn_flat_reconstruct: one crossing each way -- pfam/proteins_search.py's flat mode does.)
    const int64_t rows_per = std::max<int64_t>(1, (int64_t)(64ull << 20) / ((int64_t)d * 4));
    const int64_t nbuf = std::min(n, rows_per);
    float *buf[2] = {nullptr, nullptr};
    size_t got[2] = {0, 0};
    hipStream_t st[2] = {nullptr, nullptr};
    auto cleanup = [&]() {
        for (int i = 0; i < 2; i++) {
            if (st[i]) {
                (void)hipStreamSynchronize(st[i]);
                g_streams.give(g_device, st[i]);
            }
            if (buf[i]) g_pool.give(buf[i], got[i], g_device);
        }
    };
    const int lanes = n > nbuf ? 2 : 1;
    for (int i = 0; i < lanes; i++) {
        buf[i] = (float *)pool_alloc((size_t)nbuf * d * 4, g_device, &got[i]);
        st[i] = g_streams.take(g_device);
        if (!buf[i] || !st[i]) {
            cleanup();
            return set_err(KNN_ERR_HIP, "normalize_l2: out of device memory");
        }
    }
    int turn = 0;
    for (int64_t i0 = 0; i0 < n; i0 += nbuf, turn ^= (lanes - 1)) {
        const int64_t m = std::min(nbuf, n - i0);
        hipError_t e = hipStreamSynchronize(st[turn]); // (this buffer's previous chunk is home)
        if (e == hipSuccess) e = hipMemcpyAsync(buf[turn], x_host + i0 * d, (size_t)m * d * 4, hipMemcpyHostToDevice, st[turn]);
        if (e == hipSuccess) {
            rc = normalize_dev_impl(buf[turn], m, d, d, st[turn]);
            if (rc) { cleanup(); return rc; }
            e = hipMemcpyAsync(x_host + i0 * d, buf[turn], (size_t)m * d * 4, hipMemcpyDeviceToHost, st[turn]);
        }
        if (e != hipSuccess) {
            cleanup();
            return set_err(KNN_ERR_HIP, std::string("normalize_l2 copy: ") + hipGetErrorString(e));
        }
    }
    for (int i = 0; i < lanes; i++) {
        hipError_t e = hipStreamSynchronize(st[i]);
        if (e != hipSuccess) {
            cleanup();
            return set_err(KNN_ERR_HIP, std::string("normalize_l2: ") + hipGetErrorString(e));
        }
    }
    cleanup();
    return 0;
}

// ---- index lifecycle ------------------------------------------------------
extern "C" int knn_flat_create(int32_t d, int32_t metric, knn_handle *out)
{
    if (!out) return set_err(KNN_ERR_INVALID, "flat_create: null out");
    if (d <= 0) return set_err(KNN_ERR_INVALID, "flat_create: d must be positive");
    if (metric != KNN_METRIC_INNER_PRODUCT && metric != KNN_METRIC_L2)
        return set_err(KNN_ERR_UNSUPPORTED, "flat_create: metric must be METRIC_INNER_PRODUCT (0) or METRIC_L2 (1)");
    int rc = ensure_device(g_device);
    if (rc) return rc;
    knn_index_s *h = new knn_index_s();
    h->d = d;
    h->dp = round_up(d, 32);
    h->metric = metric;
    h->device = g_device;
#ifdef KNN355_DEV
    if (const char *e = getenv("KNN355_FLAGS")) h->flags = atoi(e); // (developer build only, A/B runs: the tuning flags every index starts with)
#endif
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g_device) == hipSuccess && cus > 0) h->num_cus = cus;
    }
    if (!(h->stream = g_streams.take(h->device))) {
        delete h;
        return set_err(KNN_ERR_HIP, "flat_create: stream creation failed");
    }
    (void)hipEventCreateWithFlags(&h->done, hipEventDisableTiming); // (without it a growing buffer waits for the whole device)
    *out = h;
    return 0;
}

extern "C" int knn_flat_view(knn_handle parent, knn_handle *out)
{
    if (!parent || !out) return set_err(KNN_ERR_INVALID, "flat_view: null argument");
    std::lock_guard<std::mutex> lk(parent->mu);
    if (parent->approx16) return set_err(KNN_ERR_UNSUPPORTED, "flat_view: not for an approximate (bf16) index");
    HIP_TRY(hipSetDevice(parent->device));
    knn_index_s *h = new knn_index_s();
    h->flags = parent->flags;
    h->pub_rounds_force = parent->pub_rounds_force;
    h->force_qt = parent->force_qt;
    h->force_chunks = parent->force_chunks;
    h->d = parent->d;
    h->dp = parent->dp;
    h->metric = parent->metric;
    h->device = parent->device;
    h->num_cus = parent->num_cus;
    h->is_view = true;
    h->storage_gen = parent->storage_gen;
    h->view_gen = parent->storage_gen->load();
    h->xb = parent->xb;
    h->yn = parent->yn;
    h->ntotal = parent->ntotal;
    h->cap_rows = parent->ntotal;
    if (!(h->stream = g_streams.take(h->device))) {
        delete h;
        return set_err(KNN_ERR_HIP, "flat_view: stream creation failed");
    }
    (void)hipEventCreateWithFlags(&h->done, hipEventDisableTiming);
    *out = h;
    return 0;
}

// database storage goes through the same pool as the scratch buffers
static void *pool_alloc(size_t bytes, int device, size_t *got)
{
    if (void *q = g_pool.take(bytes, device, got)) return q;
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    *got = bytes;
    return p;
}

static void free_index_buffers(knn_index_s *h)
{
    if (!h->is_view) h->storage_gen->fetch_add(1);
    if (h->xb && !h->is_view) g_pool.give(h->xb, h->xb_bytes, h->device);
    if (h->yn && !h->is_view) g_pool.give(h->yn, h->yn_bytes, h->device);
    h->xb = nullptr;
    h->yn = nullptr;
    h->ntotal = 0;
    h->cap_rows = 0;
}

extern "C" void knn_free(knn_handle h)
{
    if (!h) return;
    if (hipSetDevice(h->device) == hipSuccess) {
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (!h->is_view && h->xb) (void)hipDeviceSynchronize(); // a view's stream may still be scanning these rows
        free_index_buffers(h);
        DevBuf *bufs[] = {&h->xb16, &h->ws_q16, &h->ws_sym, &h->ws_qdiff, &h->ws_defer, &h->ws_turn, &h->ws_flag, &h->ws_q, &h->ws_qn, &h->ws_lists, &h->ws_D, &h->ws_I, &h->ws_tmp, &h->ws_tmp2, &h->ws_D1, &h->ws_I1, &h->ws_tmp3};
        for (DevBuf *b : bufs) b->release();
        for (LevelBufs &b : h->ws_level) {
            b.qlist.release();
            b.qcnt.release();
            b.gthr.release();
            b.qthr.release();
            b.pub.release();
            b.arrive.release();
            b.pair_ctr.release();
        }
        for (int i = 0; i < knn_index_s::RING; i++) {
            if (h->ring0[i]) (void)hipEventDestroy(h->ring0[i]);
            if (h->ring1[i]) (void)hipEventDestroy(h->ring1[i]);
        }
        if (h->stream) g_streams.give(h->device, h->stream); // (synchronised above)
        if (h->flag_host) (void)hipHostFree(h->flag_host);
        if (h->done) (void)hipEventDestroy(h->done);
    }
    delete h;
}

extern "C" int64_t knn_ntotal(knn_handle h) { return h ? h->ntotal : -1; }
extern "C" int32_t knn_dim(knn_handle h) { return h ? h->d : -1; }
extern "C" int32_t knn_metric(knn_handle h) { return h ? h->metric : -1; }
extern "C" int32_t knn_device_of(knn_handle h) { return h ? h->device : -1; }

extern "C" int knn_reset(knn_handle h)
{
    if (!h) return set_err(KNN_ERR_INVALID, "reset: null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->is_view) return set_err(KNN_ERR_INVALID, "reset: a view is read-only");
    HIP_TRY(hipSetDevice(h->device));
    free_index_buffers(h);
    return 0;
}

static int grow_index(knn_index_s *h, int64_t need_rows)
{
    if (need_rows <= h->cap_rows) return 0;
    int64_t new_cap = std::max<int64_t>(need_rows, h->cap_rows + h->cap_rows / 2);
    if (h->cap_rows == 0) new_cap = need_rows; // first add: exact fit (the reference adds once)
    size_t row_bytes = (size_t)h->dp * 4;
    size_t xb_got = 0, yn_got = 0;
    float *nxb = (float *)pool_alloc((size_t)new_cap * row_bytes, h->device, &xb_got);
    if (!nxb) {
        new_cap = need_rows;
        nxb = (float *)pool_alloc((size_t)new_cap * row_bytes, h->device, &xb_got);
        if (!nxb) return set_err(KNN_ERR_HIP, "add: out of device memory");
    }
    float *nyn = (float *)pool_alloc(((size_t)new_cap + 64) * 4, h->device, &yn_got);
    if (!nyn) {
        g_pool.give(nxb, xb_got, h->device);
        return set_err(KNN_ERR_HIP, "add: out of device memory");
    }
    if (h->ntotal > 0) {
        HIP_TRY(hipMemcpyAsync(nxb, h->xb, (size_t)h->ntotal * row_bytes, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(nyn, h->yn, (size_t)h->ntotal * 4, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->xb) {
        h->storage_gen->fetch_add(1); // views of the old storage are dead
        (void)hipDeviceSynchronize(); // ... and nothing may still be scanning it
        g_pool.give(h->xb, h->xb_bytes, h->device);
    }
    if (h->yn) g_pool.give(h->yn, h->yn_bytes, h->device);
    h->xb = nxb;
    h->yn = nyn;
    h->xb_bytes = xb_got;
    h->yn_bytes = yn_got;
    h->cap_rows = new_cap;
    return 0;
}

// bf16 copies of rows [r0, r0 + n) of an approximate index (kept beside the fp32 rows: reconstruct, norms and the
// exact re-scoring use those)
static int approx16_sync_rows(knn_index_s *h, int64_t r0, int64_t n, hipStream_t s)
{
    if (!(h->approx16 || h->keep16) || n <= 0) return 0;
    const size_t need = (size_t)h->cap_rows * h->dp * 2;
    if (h->xb16.bytes < need) {
        // (regrown with the fp32 storage: convert everything that is there)
        if (h->xb16.ensure(need)) return set_err(KNN_ERR_HIP, "add: out of device memory (bf16 rows)");
        n += r0;
        r0 = 0;
    }
    const int64_t total = n * h->dp;
    const unsigned grid = (unsigned)std::min<int64_t>((total / 4 + 255) / 256, 65535);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(grid), dim3(256), 0, s, h->xb + (size_t)r0 * h->dp, total, (__bf16 *)h->xb16.p + (size_t)r0 * h->dp);
    HIP_TRY(hipGetLastError());
    return 0;
}

// before the first add: the index multiplies bf16 copies (rows padded to a multiple of 64 values)
static int flat_set_approx16(knn_index_s *h)
{
    if (h->ntotal != 0) return set_err(KNN_ERR_INVALID, "approx16: the index already holds rows");
    h->approx16 = true;
    h->dp = round_up(h->d, 64);
    return 0;
}

// before the first add: bf16 copies of the rows are kept (the scans do not use them)
static int flat_keep16(knn_index_s *h)
{
    if (h->ntotal != 0) return set_err(KNN_ERR_INVALID, "keep16: the index already holds rows");
    h->keep16 = true;
    return 0;
}

// appends rows that already live on the device ([n][d], contiguous)
static int add_dev_impl(knn_index_s *h, const float *x_dev, int64_t n, hipStream_t s)
{
    float *dst = h->xb + (size_t)h->ntotal * h->dp;
    if (h->dp == h->d) {
        HIP_TRY(hipMemcpyAsync(dst, x_dev, (size_t)n * h->d * 4, hipMemcpyDeviceToDevice, s));
    } else {
        int64_t total = n * h->dp;
        unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 65535);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, s, x_dev, n, h->d, dst, h->dp);
        HIP_TRY(hipGetLastError());
    }
    // norms are kept for both metrics: reconstruct/HNSW-L2 need them and they cost one pass
    int rc = norms_dev_impl(dst, n, h->d, h->dp, h->yn + h->ntotal, s);
    if (rc) return rc;
    rc = approx16_sync_rows(h, h->ntotal, n, s);
    if (rc) return rc;
    h->ntotal += n;
    return 0;
}

extern "C" int knn_flat_add_dev(knn_handle h, const float *x_dev, int64_t n, void *stream)
{
    if (!h) return set_err(KNN_ERR_INVALID, "add: null handle");
    if (n < 0) return set_err(KNN_ERR_INVALID, "add: negative n");
    if (n == 0) return 0;
    if (!x_dev) return set_err(KNN_ERR_INVALID, "add: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->is_view) return set_err(KNN_ERR_INVALID, "add: a view is read-only");
    HIP_TRY(hipSetDevice(h->device));
    if (h->ntotal + n > 0xFFFFFFF0ll) return set_err(KNN_ERR_UNSUPPORTED, "add: more than 2^32 rows per index");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (stream) HIP_TRY(hipStreamSynchronize(h->stream));
    int rc = grow_index(h, h->ntotal + n);
    if (rc) return rc;
    rc = add_dev_impl(h, x_dev, n, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int knn_flat_add(knn_handle h, const float *x_host, int64_t n)
{
    if (!h) return set_err(KNN_ERR_INVALID, "add: null handle");
    if (n < 0) return set_err(KNN_ERR_INVALID, "add: negative n");
    if (n == 0) return 0;
    if (!x_host) return set_err(KNN_ERR_INVALID, "add: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->is_view) return set_err(KNN_ERR_INVALID, "add: a view is read-only");
    HIP_TRY(hipSetDevice(h->device));
    if (h->ntotal + n > 0xFFFFFFF0ll) return set_err(KNN_ERR_UNSUPPORTED, "add: more than 2^32 rows per index");
    int rc = grow_index(h, h->ntotal + n);
    if (rc) return rc;
    if (h->dp == h->d) {
        float *dst = h->xb + (size_t)h->ntotal * h->dp;
        HIP_TRY(hipMemcpy(dst, x_host, (size_t)n * h->d * 4, hipMemcpyHostToDevice));
        rc = norms_dev_impl(dst, n, h->d, h->dp, h->yn + h->ntotal, h->stream);
        if (rc) return rc;
        rc = approx16_sync_rows(h, h->ntotal, n, h->stream);
        if (rc) return rc;
        h->ntotal += n;
    } else {
        const int64_t rows_per = std::max<int64_t>(1, (int64_t)(256ull << 20) / ((int64_t)h->d * 4));
        if (h->ws_tmp.ensure((size_t)std::min(n, rows_per) * h->d * 4)) return set_err(KNN_ERR_HIP, "add: out of device memory");
        for (int64_t i0 = 0; i0 < n; i0 += rows_per) {
            int64_t m = std::min(rows_per, n - i0);
            HIP_TRY(hipMemcpy(h->ws_tmp.p, x_host + i0 * h->d, (size_t)m * h->d * 4, hipMemcpyHostToDevice));
            rc = add_dev_impl(h, (const float *)h->ws_tmp.p, m, h->stream);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int knn_flat_reconstruct(knn_handle h, int64_t i0, int64_t n, float *out_host)
{
    if (!h) return set_err(KNN_ERR_INVALID, "reconstruct: null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (i0 < 0 || n < 0 || i0 + n > h->ntotal) return set_err(KNN_ERR_INVALID, "reconstruct: range out of bounds");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy2D(out_host, (size_t)h->d * 4, h->xb + (size_t)i0 * h->dp, (size_t)h->dp * 4, (size_t)h->d * 4,
                        (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

// ---- search ---------------------------------------------------------------
// A chunk hands on at most kslot = 1.25 k keys per query (k for k > 1536); the final selection
// is exact.  The register select inside the scan kernels serves k <= 1536 (lists of <= 2048
// keys, 32 per lane, so that 1.25 k + one tile fits); beyond that the workgroup sort takes
// exactly k.
static const int KNN_WAVE_SELECT_MAX_K = 2048;    // one wave selects a list: in registers up to ...
// ... this k (2048-key lists), by probing a 4096-key list in memory beyond.  1.25 k keys + a tile of appends fit a 2048-key
// list up to k = 1536, but the closer k gets the less room a cut leaves and the more often a list is cut: measured crossover
// at k ~ 1400 (2 M rows x 1024 queries: k = 1400 43.4 ms either way, k = 1536 51.3 ms with 2048-key lists, 44.8 with 4096)
static const int KNN_REGISTER_SELECT_MAX_K = 1400;
// (developer build: integer knobs from the environment; the shipped library has the defaults compiled in)
static int dev_knob(const char *name, int dflt)
{
#ifdef KNN355_DEV
    if (getenv(name)) return atoi(getenv(name));
#endif
    (void)name;
    return dflt;
}
static int knn_kslot(int k) { return k > KNN_WAVE_SELECT_MAX_K ? k : k + k / 4; }

static int next_pow2_host(int n)
{
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}

// one workgroup per query: best k of its candidate keys (see select_topk_kernel).  tmp: scratch for the segment
// pass of arrays longer than one workgroup's registers hold.
static const int KNN_SELECT_SEG = 32768;
static int launch_select(SelectParams sp, hipStream_t s, DevBuf *tmp = nullptr)
{
    if (sp.nq <= 0) return 0;
    // (an array whose CAPACITY is larger but which is expected to hold far fewer keys -- the compact arrays of an
    // exactly seeded scan are sized for the worst case of every chunk -- goes straight to the one-launch selection:
    // 32768 keys fit its registers, a longer array is re-read from L2 on every probe)
    const bool expect_short = sp.n_expect > 0 && sp.n_expect <= 24576;
    if (tmp && sp.lm_lists == 0 && sp.cnt && sp.cap > KNN_SELECT_SEG && !expect_short) {
        // Arrays of up to `cap` > 32768 keys (a seed sample of a 10 M-row database hands on 39 k scores per
        // query): workgroup (q, s) reduces segment s of query q to its kk = kmax best keys in registers, then
        // one more launch selects among the nseg * kk survivors -- two launches instead of re-reading the
        // array from L2 on every probe.
        const int kk = sp.k + std::max(sp.k >> 2, 32);
        const int nseg = (sp.cap + KNN_SELECT_SEG - 1) / KNN_SELECT_SEG;
        if (tmp->ensure((size_t)sp.nq * nseg * kk * 8)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        SelectParams a = {};
        a.in = sp.in; a.in_stride = sp.in_stride; a.cnt = sp.cnt; a.cap = sp.cap;
        a.seg_len = KNN_SELECT_SEG; a.nseg = nseg;
        a.nq = sp.nq; a.k = kk; a.metric = sp.metric;
        a.out_keys = (uint64_t *)tmp->p; a.out_stride = kk; a.out_fill = 0;
        const size_t lds = (size_t)next_pow2_host(kk + std::max(kk >> 2, 32)) * 8;
        hipLaunchKernelGGL((select_topk_kernel<32, 1024, false>), dim3((unsigned)sp.nq, (unsigned)nseg), dim3(1024), lds, s, a);
        HIP_TRY(hipGetLastError());
        sp.in = (const uint64_t *)tmp->p;
        sp.in_stride = (int64_t)nseg * kk;
        sp.cnt = nullptr;
        sp.n_fixed = nseg * kk;
        sp.cap = nseg * kk;
        sp.n_expect = 0;
    }
    const bool seed = sp.seed_cnt != nullptr;
    const int kmax = sp.k + std::max(sp.k >> 2, 32);
    const size_t lds = seed ? 0 : (size_t)next_pow2_host(kmax) * 8;
    const int nmax = sp.lm_lists > 0 ? sp.lm_lists * sp.lm_k : (sp.cnt ? sp.cap : sp.n_fixed);
    void (*kern)(SelectParams);
    int nt = 256;
    // Many queries with short candidate arrays (the batch regime): ONE WAVE per query -- four times the queries
    // in flight per CU, and the kernel is bound by the latency of its dependent loads, not by arithmetic
    // (14433 queries x ~1400 keys: 265 us with a workgroup per query).  Few queries with long arrays (the
    // streaming regime): a workgroup per query, up to 32768 keys in registers.  Arrays longer than the
    // registers of the chosen build hold are re-read from L2 on every probe.
    const int nexp = sp.n_expect > 0 ? std::min(sp.n_expect, nmax) : nmax;
    if (sp.nq >= 512 && nexp <= 64 * 32) {
        nt = 64;
        if (seed) kern = nexp <= 64 * 8 ? select_topk_kernel<8, 64, true> : select_topk_kernel<32, 64, true>;
        else kern = nexp <= 64 * 8 ? select_topk_kernel<8, 64, false> : select_topk_kernel<32, 64, false>;
    } else if (seed) {
        if (nmax <= 256 * 4) kern = select_topk_kernel<4, 256, true>;
        else if (nmax <= 256 * 16) kern = select_topk_kernel<16, 256, true>;
        else if (nmax <= 256 * 32) kern = select_topk_kernel<32, 256, true>;
        else { kern = select_topk_kernel<32, 1024, true>; nt = 1024; }
    } else {
        // (by the EXPECTED count when the arrays carry their own: the compact arrays of a seeded scan are sized for the worst
        // case of every chunk -- 64 k slots per query on a 1.25 M-row shard -- and hold 1-2 k keys; the 1024-thread build
        // took 32-36 us for them, its probes are workgroup-wide reductions over 16 waves)
        const int nsel = sp.cnt ? nexp : nmax;
        if (nsel <= 256 * 4) kern = select_topk_kernel<4, 256, false>;
        else if (nsel <= 256 * 16) kern = select_topk_kernel<16, 256, false>;
        else if (nsel <= 256 * 32) kern = select_topk_kernel<32, 256, false>;
        else { kern = select_topk_kernel<32, 1024, false>; nt = 1024; }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)sp.nq), dim3(nt), lds, s, sp);
    HIP_TRY(hipGetLastError());
    return 0;
}

struct ScanPlan {
    int qt, dt, nqtiles, nchunks, cap, grid;
    int npairs; // > 0: paired walk -- grid = nchunks = 2 npairs workgroups, tiles_base / tiles_rem split the tiles over the PAIRS
    bool diff;  // squared L2 as the sum of squared differences (FAISS's small-batch formula), see flat_scan_kernel<..., DIFF>
    int64_t chunk_rows;
    int tiles_base, tiles_rem;
    size_t lds;
    const char *name;
};


// the difference build that serves a batch of nq < 20 queries: its width (a multiple of 4: two query pairs per thread half)
// (no 4-query build: with two chains per thread it is bound by their dependent latency -- 6.9 ms per 10 M rows where the 8-query
// build, four chains per thread, reads at the HBM rate: 6.25 ms)
static int diff_build_width(int64_t nq) { return nq <= 8 ? 8 : (nq <= 12 ? 12 : (nq <= 16 ? 16 : 20)); }

template <int WM, int WN, int TM, int TN>
static int launch_scan_cfg(const knn_index_s *h, const ScanParams &p, const ScanPlan &plan, hipStream_t s)
{
    const bool l2 = h->metric == KNN_METRIC_L2;
    void (*kern)(ScanParams) = nullptr;
    // one query tile: rows are read once, non-temporal staging loads
    if constexpr (WM == 4 && TM == 2) {
        if (plan.diff) {
            // (builds for up to 8, 12, 16 and 19 queries: the vector work of a K step grows with the build's width -- up to 8
            // queries scan at the speed of their HBM traffic)
            switch (diff_build_width(p.nq)) {
            case 8: kern = flat_scan_kernel<4, 1, 2, 1, true, true, false, false, 8>; break;
            case 12: kern = flat_scan_kernel<4, 1, 2, 1, true, true, false, false, 12>; break;
            case 16: kern = flat_scan_kernel<4, 1, 2, 1, true, true, false, false, 16>; break;
            default: kern = flat_scan_kernel<4, 1, 2, 1, true, true, false, false, 20>; break;
            }
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));
            hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(256), plan.lds, s, p);
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    if (h->approx16) {
        if (p.nqtiles == 1) kern = l2 ? flat_scan_kernel<WM, WN, TM, TN, true, true, false, true> : flat_scan_kernel<WM, WN, TM, TN, false, true, false, true>;
        else kern = l2 ? flat_scan_kernel<WM, WN, TM, TN, true, false, false, true> : flat_scan_kernel<WM, WN, TM, TN, false, false, false, true>;
    } else if (p.nqtiles == 1) kern = l2 ? flat_scan_kernel<WM, WN, TM, TN, true, true> : flat_scan_kernel<WM, WN, TM, TN, false, true>;
    else kern = l2 ? flat_scan_kernel<WM, WN, TM, TN, true, false> : flat_scan_kernel<WM, WN, TM, TN, false, false>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));
    hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(256), plan.lds, s, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the 256 x 256 tile (2 x 2 waves of 4 x 4 MFMA tiles, one workgroup per CU): plain fp32 rows, several query tiles per launch
static int launch_scan_big(const knn_index_s *h, const ScanParams &p, const ScanPlan &plan, hipStream_t s)
{
    if (h->approx16) return set_err(KNN_ERR_INVALID, "scan: the 256-query tile serves plain fp32 rows");
    void (*kern)(ScanParams) = h->metric == KNN_METRIC_L2 ? flat_scan_kernel<2, 2, 4, 4, true, false> : flat_scan_kernel<2, 2, 4, 4, false, false>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));
    hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(256), plan.lds, s, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

// the builds on 16-query blocks (48 queries: 4 x 1 waves, 96: 2 x 2 waves; Q16 blocks per wave): one query tile per launch
template <int WM, int WN, int TM, int Q16>
static int launch_scan16(const knn_index_s *h, const ScanParams &p, const ScanPlan &plan, hipStream_t s)
{
    if (p.nqtiles != 1 || h->approx16) return set_err(KNN_ERR_INVALID, "scan: the 16-query-block builds serve one query tile of plain fp32 rows");
    void (*kern)(ScanParams) = h->metric == KNN_METRIC_L2 ? flat_scan_kernel<WM, WN, TM, 1, true, true, false, false, 0, Q16>
                                                          : flat_scan_kernel<WM, WN, TM, 1, false, true, false, false, 0, Q16>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));
    hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(256), plan.lds, s, p);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Does the 256 x 256 tile (flat_scan_kernel<2, 2, 4, 4>: one workgroup per CU) serve this search?  Its K loop keeps the matrix
// pipe busier than two co-resident 128 x 128 workgroups do, but nothing hides its epilogues, a launch has half the
// workgroups and four times the tile: it wants long chunks on every CU -- Pfam-sized batches, not CATH-sized ones.
// flags & 262144: never; flags & 524288: wherever a batch has more than 128 queries (tests, A/B).
static bool big_tile_pays(const knn_index_s *h, int64_t nb, int64_t nq)
{
    if (h->approx16 || (h->flags & 262144) || nq <= 128) return false;
    if (h->flags & 524288) return nb >= 1024;
    if (nb < 65536) return false; // (never a seed sample's own scan)
    const int64_t cus = std::max(1, h->num_cus);
    const int64_t work = ((nq + 255) / 256) * ((nb + 255) / 256); // 256 x 256 tiles of the search
    // (24 tiles per CU: 1 M rows x 512 queries 8.27 against 8.78 ms, 500 k x 1024 8.16 against 8.47, 400 k x 1024 7.11 against 7.3,
    // 200 k x 2048 6.73 against 6.80 -- the bound was 32 until late in round 5.)
    // (from ONE wide query tile on: 10 M rows x 256 / 512 / 768 / 1024 / 1536 queries 38.1 / 79.1 / 117.6 / 152.4 / 234.4 ms against
    // 41.4 / 89.4 / 130.8 / 172.0 / 248.8 on the 128 x 128 tile, one box; until late in round 5 the bound was 2048)
    return nq >= dev_knob("KNN355_BIG_MIN_NQ", 256) && work >= (int64_t)dev_knob("KNN355_BIG_MIN_TILES_PER_CU", 24) * cus;
}

static void make_plan(const knn_index_s *h, int64_t nb, int64_t nq, int k, bool seeded, ScanPlan &pl, bool allow_pairs = false, bool allow_big = true)
{
    int qt = h->force_qt;
    // (48 and 96: the 16-query-block builds, one query tile per launch only -- a forced one is honoured if it holds the batch)
    if ((qt == 48 || qt == 96) && (nq > qt || h->approx16)) qt = 0;
    if (qt == 256 && (h->approx16 || nq <= 128)) qt = 0; // (the 256 x 256 tile: plain fp32 rows, more than one 128-query tile of queries)
    if (qt != 32 && qt != 48 && qt != 64 && qt != 96 && qt != 128 && qt != 256) {
        qt = nq <= 32 ? 32 : (nq <= 64 ? 64 : 128);
        if (!h->approx16 && !(h->flags & 131072)) { // (flags & 131072: without the 16-query-block builds)
            if (nq > 32 && nq <= 48) qt = 48;
            else if (nq > 64 && nq <= 96) qt = 96;
        }
        if (allow_big && big_tile_pays(h, nb, nq)) qt = 256;
    }
    // FAISS's squared L2 for fewer than 20 queries: the sum of squared differences (flags & 32: the norm formula throughout).
    // FAISS decides on the batch its caller handed over, so a piece of a larger batch (the last block of 16384 queries, the
    // remainder behind the full query tiles) keeps the formula of the whole.
    const bool small_batch = (h->batch_nq ? h->batch_nq : nq) < 20 && nq < 20;
    if (h->metric == KNN_METRIC_L2 && small_batch && !h->approx16 && !(h->flags & 32)) qt = 32; // (the difference build exists for the 32-query tile only)
    pl.qt = qt;
    pl.dt = (qt == 32 || qt == 48 || qt == 256) ? 256 : 128;
    pl.diff = h->metric == KNN_METRIC_L2 && small_batch && qt == 32 && !h->approx16 && !(h->flags & 32);
    pl.name = pl.diff ? "flat_scan_q32_d256_l2diff"
                      : (qt == 256 ? "flat_scan_q256_d256" : qt == 128 ? "flat_scan_q128_d128" : (qt == 96 ? "flat_scan_q96_d128" : (qt == 64 ? "flat_scan_q64_d128" : (qt == 48 ? "flat_scan_q48_d256" : "flat_scan_q32_d256"))));
    pl.nqtiles = (int)((nq + qt - 1) / qt);
    pl.cap = next_pow2_host(2 * k + pl.dt);
    if (pl.cap < 512) pl.cap = 512;
    // (the 256 x 256 tile: nothing hides a cut of its 256 lists on four waves -- 220 us, a K loop's worth: room for three tiles
    // of appends between cuts)
    if (qt == 256 && pl.cap < 1024) pl.cap = 1024;
    int regsel_max = KNN_REGISTER_SELECT_MAX_K;
#ifdef KNN355_DEV
    if (getenv("KNN355_REGSEL_MAX_K")) regsel_max = atoi(getenv("KNN355_REGSEL_MAX_K")); // (developer build: where the 4096-key lists take over)
#endif
    if (k <= regsel_max) pl.cap = std::min(pl.cap, 2048); // register select: <= 32 keys per lane
    else if (k <= KNN_WAVE_SELECT_MAX_K) pl.cap = std::min(pl.cap, 4096);  // wave_select_mem: 1.25 k + a tile of appends fit
    const int64_t ntiles = (nb + pl.dt - 1) / pl.dt;
    pl.npairs = 0;
    if (allow_pairs && pl.nqtiles == 1 && qt != 256 && h->force_chunks <= 0 && !(h->flags & 4) && ntiles >= (int64_t)dev_knob("KNN355_PAIR_MIN_TILES", 64)) {
        // one query tile, plenty of tiles: two workgroups per CU, paired (see flat_scan_kernel): each pair shares a
        // contiguous range of ~ ntiles / CUs tiles (at least two: fewer pairs than CUs on a small database)
        pl.npairs = (int)std::min<int64_t>(std::max(1, h->num_cus), ntiles / 2);
        pl.nchunks = 2 * pl.npairs;
        pl.grid = pl.nchunks;
        pl.tiles_base = (int)(ntiles / pl.npairs);
        pl.tiles_rem = (int)(ntiles % pl.npairs);
        pl.chunk_rows = (int64_t)(pl.tiles_base + (pl.tiles_rem ? 1 : 0)) * pl.dt;
        pl.lds = std::max((size_t)2 * (pl.dt + pl.qt) * 128, (size_t)pl.cap * 8) + (size_t)qt * 12 + 16 + (size_t)pl.dt * 4 + (size_t)qt * 8;
        return;
    }
    // workgroups resident per CU: two, or the one 256 x 256 workgroup
    const int per_cu = qt == 256 ? 1 : 2;
    int64_t want = h->force_chunks > 0 ? h->force_chunks : (512 * per_cu + pl.nqtiles - 1) / pl.nqtiles;
    // an unseeded chunk should see enough rows to amortise its threshold warm-up; a seeded pass
    // starts with good thresholds and a tiny view (a seed sample) just wants parallelism
    int64_t min_tiles = std::max<int64_t>(2, (4 * (int64_t)k + pl.dt - 1) / pl.dt);
    if (h->force_chunks <= 0 && !seeded && nb >= 8192) want = std::min(want, std::max<int64_t>(1, ntiles / min_tiles));
    want = std::max<int64_t>(1, std::min(want, ntiles));
    // bound the candidate-list workspace (<= 2 GiB)
    const size_t per_wg = (size_t)qt * pl.cap * 8;
    int64_t max_wgs = std::max<int64_t>(pl.nqtiles, (int64_t)((2ull << 30) / per_wg));
    const int64_t want_max = std::max<int64_t>(1, std::min(ntiles, max_wgs / pl.nqtiles));
    want = std::min(want, want_max);
    if (h->force_chunks <= 0) {
        // wave quantisation: workgroups run in rounds of (2 per CU); pick the chunk count near
        // `want` that minimises rounds x tiles-per-chunk (14433 x 14433: 9 chunks = 2 full
        // rounds of 13 tiles beat 10 chunks = 2.2 rounds of 12)
        const int64_t slots = per_cu * (int64_t)std::max(1, h->num_cus);
        int64_t best = want, best_cost = INT64_MAX;
        for (int64_t c = std::max<int64_t>(1, want / 2); c <= std::min(want_max, want + want / 2 + 1); c++) {
            const int64_t rounds = (pl.nqtiles * c + slots - 1) / slots;
            // per chunk: its tiles + about half a tile of fixed work (prologue, cutting the lists
            // and writing the survivors) -- 1.25 M rows x 32 queries: 489 chunks of 10 tiles in
            // one round beat 977 chunks of 5 tiles in two
            const int64_t cost = rounds * (2 * ((ntiles + c - 1) / c) + 1);
            if (cost < best_cost || (cost == best_cost && c > best)) { best = c; best_cost = cost; }
        }
        want = best;
    }
    int64_t tiles_per = (ntiles + want - 1) / want;
    // balanced split: the first (ntiles mod nchunks) chunks walk one tile more.  (A uniform chunk length left the
    // last chunk short -- 14433 rows in 9 chunks: 8 x 13 tiles + 9 -- and the two rounds of workgroups took 13 + 13
    // tile times; 5 x 13 + 4 x 12 takes 13 + 12.)
    pl.nchunks = (int)((ntiles + tiles_per - 1) / tiles_per);
    pl.tiles_base = (int)(ntiles / pl.nchunks);
    pl.tiles_rem = (int)(ntiles % pl.nchunks);
    pl.chunk_rows = (int64_t)(pl.tiles_base + (pl.tiles_rem ? 1 : 0)) * pl.dt; // (the longest chunk)
    pl.grid = pl.nqtiles * pl.nchunks;
    pl.lds = std::max((size_t)2 * (pl.dt + pl.qt) * 128, (size_t)pl.cap * 8) + (size_t)qt * 12 + 16 + (size_t)pl.dt * 4 + (size_t)qt * 8; // (+ s_pub)
}

// Seed stride of a view with nb rows: a power of two s such that the sample (every s-th 8-row block)
// has about max(2 * chunk_rows, 64 k) rows (and at most nb/8).  The sample is searched first,
// exactly; its k-th score bounds the global k-th from above, so every chunk of the main pass
// starts with a tight threshold and appends about chunk_rows * k / sample_rows <= k/2
// candidates per query instead of warming up (and compacting) on its own.
static int seed_stride(int64_t nb, int k, int64_t chunk_rows)
{
    const int64_t target = std::min<int64_t>(std::max<int64_t>(2 * chunk_rows, 64 * (int64_t)k), nb / 8);
    int s = 8;
    while ((int64_t)s * 2 * target <= nb && s < (1 << 20)) s *= 2;
    return s;
}

// Statistical seed (batch regime).  The exact seed above needs a sample of >= k rows whose k-th
// score is a PROVEN bound of the global k-th: with k = 301 of 14433 rows no affordable sample gives a
// useful one, every chunk warms up on its own and half of all scores go through the candidate lists.
// Instead: T = the j-th best score of a sample of S rows, j << k chosen so that, were the sample
// drawn at random, fewer than k of the N rows beat T with probability <= 1e-9 per query
// (P[Binomial(S, k/N) >= j] <= 1e-9).  Every chunk filters with T from its first tile on (about
// j N / S candidates per query instead of ~ chunks x k (1 + ln(rows per chunk / k))).  T is only
// an estimate, so the result is VERIFIED: the final selection checks that the k-th score it found is
// <= T -- then at least k rows beat T, all of them were candidates, and the result is exact.  A
// query that fails the check raises a flag and the search is redone without the estimate (the
// caller must be able to wait for the flag: synchronous entry points only).
static int stat_seed_rank(int64_t S, int64_t N, int k)
{
    if (S < 64 || N <= 0 || k >= N) return -1;
    const double pr = (double)k / (double)N, eps = 1e-9;
    // smallest j with P[Bin(S, pr) >= j] <= eps: walk the pmf upwards, accumulating the lower tail
    const double lp = log(pr), lq = log1p(-pr);
    double cdf = 0.0;
    const int64_t jmax = std::min<int64_t>(S, k);
    for (int64_t i = 0; i <= jmax; i++) {
        if (1.0 - cdf <= eps) return i >= 1 ? (int)i : 1; // P[X >= i] = 1 - P[X <= i-1]
        cdf += exp(lgamma((double)S + 1.0) - lgamma((double)i + 1.0) - lgamma((double)(S - i) + 1.0) + (double)i * lp + (double)(S - i) * lq);
    }
    return -1; // would need more than min(S, k) sample hits: no statistical seed
}

// where the result of a (view) search goes
struct SearchOut {
    uint64_t *keys = nullptr; // sorted keys, k per query
    int64_t keys_stride = 0;
    int keys_fill = 0;
    float *D = nullptr;
    int64_t *I = nullptr;
    // seeding the enclosing search (see SelectParams)
    uint32_t *seed_cnt = nullptr, *seed_gthr = nullptr, *seed_qthr = nullptr;
    int seed_j = 0, seed_stat = 0;
    int64_t seed_nslots = 0;
};

// Exact top-k of the block-strided view (view_row) with stride row_mul / block shift vshift for
// queries [nq][dp] on the device.  allow_stat: the caller checks h->ws_flag afterwards.
static int search_view(knn_index_s *h, const float *q_dev, const float *xn, int64_t nq, int k, uint32_t id_base, int row_mul,
                       int vshift, int level, const SearchOut &out, bool allow_stat, hipStream_t s, int *reset_flag = nullptr)
{
    const int64_t nb = view_rows(h->ntotal, row_mul, vshift);
    ScanPlan pl;
    const bool allow_pairs = !h->approx16; // (the bf16 build has no one-query-tile streaming case worth pairing)
    // (the 256 x 256 tile needs the statistical seed -- see below -- so only callers that can check its verification flag get it)
    const bool allow_big = (allow_stat && level == 0 && row_mul == 1 && !(h->flags & (8 | 16 | 512))) || (h->flags & 524288);
    make_plan(h, nb, nq, k, true, pl, allow_pairs, allow_big);
    if (level >= knn_index_s::MAX_LEVELS) return set_err(KNN_ERR_INVALID, "search: seed recursion too deep");
    // Exact seeding pays when the sample that gives every chunk a tight threshold (about two chunks'
    // worth of rows, at least 64 k) is a small fraction of the view: the streaming regime (few
    // queries, huge database, hundreds of chunks).  flags & 8 turns all seeding off, flags & 16
    // forces the exact seed, flags & 128 forces the statistical one, flags & 512 forbids it (tests).
    bool seed = nb >= 512 * (int64_t)k && std::max<int64_t>(2 * pl.chunk_rows, 64 * (int64_t)k) <= nb / 32;
    if (h->flags & 16) seed = nb >= 8192 && nb >= 32 * (int64_t)k;
    if (h->flags & (8 | 128)) seed = false;
    // (the 256 x 256 tile under a caller that can check the verification flag: the statistical estimate from every 256th row
    // instead of an exact search of every 32nd -- 10 M rows x 512 / 768 / 1536 queries 0.854 / 0.869 / 0.858 of the MFMA peak with
    // the exact seed where 1024 / 2048 queries, which the rule above leaves to the estimate, reach 0.887; the exact seed stays
    // the fallback when no rank qualifies)
    const bool exact_ok = seed;
    if (seed && pl.qt == 256 && allow_stat && level == 0 && row_mul == 1 && !(h->flags & (8 | 16 | 512))) seed = false;
    int sstride = seed ? seed_stride(nb, k, pl.chunk_rows) : 0;
    int seed_j = k, seed_stat = 0, svshift = vshift;
    double expect_n = 0; // typical candidates per query at the final selection (0: unknown, assume the capacity)
    if (!seed && allow_stat && level == 0 && row_mul == 1 && !(h->flags & (8 | 16 | 512))) {
        // statistical seed: single rows, every 32nd (every 16th of a small database, every 64th of a
        // large one): a few percent of the work, one round of workgroups at CATH size
        int st = nb >= (1 << 20) ? 64 : (nb >= 8192 ? 32 : 16);
        // The 256 x 256 tile (one workgroup per CU) ALWAYS wants the estimate, and from a sparser sample: nothing on the CU hides
        // an unseeded chunk's warm-up -- every score appended until the lists first fill, a stale bound for the ~25 tiles up to
        // the next cut (10 % of the scores pass: the sparse epilogue's lanes run out of slots and the tiles are filtered the
        // dense way), cuts of 256 lists on four waves -- Pfam-sized k = 100: 52.2 ms per 16384-query launch unseeded, 48.1
        // seeded; and the sample pass is what the seed costs: 1.93 ms per launch with every 32nd row, a quarter of that with
        // every 128th, for twice the candidates (0.8 % of the scores instead of 0.4 %: the epilogue does not notice).
        ScanPlan un;
        make_plan(h, nb, nq, k, false, un, false, allow_big);
        const bool big = un.qt == 256;
        if (big) st *= k <= 256 ? 4 : 2;
#ifdef KNN355_DEV
        if (getenv("KNN355_STAT_STRIDE")) st = atoi(getenv("KNN355_STAT_STRIDE")); // (developer build: the statistical sample's stride)
#endif
        const bool force = (h->flags & 128) != 0;
        const int64_t S = view_rows(nb, st, 0);
        const int j = stat_seed_rank(S, nb, k);
        if (j > 0 && (force || (nq >= dev_knob("KNN355_STAT_MIN_NQ", 65) && nb >= 8192))) {
            // worth it only if it removes most of the candidates an unseeded pass would collect
            // (an unseeded chunk appends every score until its list first fills, cap - dt keys, whatever k is; then about
            // k more per e-fold of rows)
            const double warm = std::min<double>((double)un.cap - un.dt, (double)un.chunk_rows);
            const double unseeded = (double)un.nchunks * std::max(warm, k * (1.0 + log(std::max(1.0, (double)un.chunk_rows / k))));
            const double seeded = (double)j * (double)nb / (double)S;
            // ... and if those candidates are a sizeable share of all scores (Pfam-sized k = 100: 5 % of the scores go
            // through the lists, an unseeded scan loses ~5 % to them and the sample pass would cost 3 %: not worth it;
            // k = 1000: 17 %, CATH-sized k = 301: 50 %)
            if (force || big || (seeded <= 0.5 * unseeded && unseeded >= 0.08 * (double)nb)) {
                sstride = st;
                seed_j = j;
                seed_stat = 1;
                svshift = 0;
                expect_n = 1.3 * seeded + 1.25 * k; // (the bound's rank is in [j, 1.25 j]) + the sample's own rows
            }
        }
        if (!seed_stat && exact_ok) { // (no estimate after all: the exact seed the rule had chosen)
            seed = true;
            sstride = seed_stride(nb, k, pl.chunk_rows);
        }
    }
    // Tile-minimum seed (see flat_scan_kernel): where the exact seed would run a sample pass first, a launch with enough
    // chunks seeds itself -- each chunk publishes its first tiles' best key per query, the k-th smallest published key
    // is the bound.  Needs: the 32- or 64-query tile, plain fp32 rows, enough publications for k (<= 4096 of them).
    // flags & 2048: never.  (The 128-query tile of a one-query-tile launch was tried: 100 k rows x 128 queries 0.66 -> 0.38 ms
    // where no sample pass exists, but the code in that build cost its other launches 2-14 % -- 1.25 M rows 2.80 -> 3.20 ms;
    // such batches are searched as two 64-query pieces instead, see search_keys_impl.)
    int pub_rounds = 0, pub_m = 1;
    const bool pub_shape = !(h->flags & (8 | 16 | 128 | 2048)) && !h->approx16 && pl.qt <= 64 && pl.cap >= 2 * pl.dt && pl.tiles_base >= (pl.npairs ? dev_knob("KNN355_PUB_MIN_TILES_PAIRED", 2) : 4);
    if (seed && pub_shape) {
        for (int r = 2; r >= 1; r--) // (one round if it gives enough publications)
            if ((int64_t)r * pl.nchunks <= 2048 && (int64_t)r * pl.nchunks >= 2 * (int64_t)k + 64 && r < pl.tiles_base) pub_rounds = r;
        if (h->pub_rounds_force > 0 && (int64_t)h->pub_rounds_force * pl.nchunks <= 2048 && h->pub_rounds_force < pl.tiles_base &&
            (int64_t)h->pub_rounds_force * pl.nchunks >= (int64_t)k + 32)
            pub_rounds = h->pub_rounds_force;
    }
    if (!pub_rounds && pub_shape && !h->pub_rounds_force && nb >= 16 * (int64_t)k && k <= KNN_WAVE_SELECT_MAX_K) {
        // A k beyond what one key per workgroup supports (512 workgroups, two rounds: k <= 480): every WAVE publishes the
        // best key of its own rows of the first tile -- 4 keys per workgroup and query with the 32-query tile, 2 with the
        // 64-query one, all of different rows, no reduction across the waves.  The k-th smallest of P such keys sits near the
        // -P ln(1 - k / P) / (P x rows per wave) quantile (k = 1000, P = 2048: 1372 of 131 k sampled rows; 2 M rows x 32
        // queries: 2.30 ms unseeded -- every workgroup warming up its own 1000 best -- against 1.45 at k = 100).
        const int wm = (pl.qt == 32 || pl.qt == 48) ? 4 : 2;
        for (int r = 2; r >= 1; r--) { // (one round if it gives enough publications; 4096 keys are selected by probing them in memory)
            const int64_t P = (int64_t)r * wm * pl.nchunks;
            if (P <= 4096 && P >= (int64_t)k + k / 4 + 32 && r < pl.tiles_base) {
                pub_rounds = r;
                pub_m = wm;
            }
        }
    }
    if (pub_rounds) {
        sstride = 0;
        // the bound sits near the k / (publications x tile rows) quantile; the first tile(s) of every chunk are filtered
        // again at the end of the chunk
        expect_n = 2.0 * (double)k * (double)nb / ((double)pub_rounds * pl.nchunks * pl.dt) + 2.0 * k + 64;
        if (pub_m > 1) {
            const double P = (double)pub_rounds * pub_m * pl.nchunks;
            expect_n = 1.3 * (-P * log(1.0 - (double)k / P)) * (double)nb / ((double)pub_rounds * pl.nchunks * pl.dt) + 2.0 * k + 64;
        }
    }
    if (!sstride && !pub_rounds) make_plan(h, nb, nq, k, level > 0, pl, allow_pairs && level == 0, allow_big); // a seed sample is small: parallelism over warm-up
    // Most keys a chunk hands on per query.  Chunks of a single tile (a seed sample, a tiny database)
    // hand on ALL their candidates: cutting 32 lists of one tile down to 1.25 k at the end of the only
    // tile is serial work per workgroup that the final selection does anyway, one workgroup per query.
    int kslot = knn_kslot(k);
    if (pl.chunk_rows == pl.dt && pl.dt > kslot && pl.dt <= pl.cap - pl.dt && k <= KNN_WAVE_SELECT_MAX_K) kslot = pl.dt;
    const int qcap = pl.nchunks * kslot + (sstride ? k + std::max(k >> 2, 32) : 0); // (a seed sample hands on up to kmax keys)
    LevelBufs &lb = h->ws_level[level];
    uint64_t reset_sig = 0; // != 0: this search's final selection resets the level's state for a successor of the same shape
    const size_t nslots = (size_t)pl.nqtiles * pl.qt;
    if (lb.qlist.ensure((size_t)nq * qcap * 8, h->done, s) || lb.qcnt.ensure((size_t)nq * 4, h->done, s) || lb.gthr.ensure(nslots * 4, h->done, s) || lb.qthr.ensure((size_t)nq * 4, h->done, s))
        return set_err(KNN_ERR_HIP, "search: out of device memory");
    uint64_t *qlist = (uint64_t *)lb.qlist.p;
    uint32_t *qcnt = (uint32_t *)lb.qcnt.p, *gthr = (uint32_t *)lb.gthr.p, *qthr = (uint32_t *)lb.qthr.p;
    int rc;
    if (sstride) {
        lb.clean_sig = 0;
        // the sample's sorted top-k opens every query's candidate array; its seed_j-th score is the
        // running threshold the main pass starts from
        SearchOut so;
        so.keys = qlist; so.keys_stride = qcap; so.keys_fill = 0;
        so.seed_cnt = qcnt; so.seed_gthr = gthr; so.seed_qthr = qthr; so.seed_j = seed_j; so.seed_stat = seed_stat;
        so.seed_nslots = (int64_t)nslots;
        // a statistical seed needs the sample's best ~1.25 j rows only (its bound has rank <= 1.25 j, and only rows that
        // beat the bound are handed on): the sample is searched with that k, not the caller's
        const int k_sample = seed_stat ? std::min(k, seed_j + std::max(seed_j >> 2, 8) + 8) : k;
        rc = search_view(h, q_dev, xn, nq, k_sample, id_base, row_mul * sstride, svshift, level + 1, so, false, s, reset_flag);
        if (rc) return rc;
    } else {
        const int64_t nn = std::max<int64_t>((int64_t)nslots, nq);
        uint64_t *pub = nullptr;
        uint32_t *arrive = nullptr;
        const int64_t npub = (int64_t)nslots * pub_rounds * pl.nchunks * pub_m;
        if (pub_rounds) {
            if (lb.pub.ensure((size_t)npub * 8, h->done, s) || lb.arrive.ensure((size_t)pl.nqtiles * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
            pub = (uint64_t *)lb.pub.p;
            arrive = (uint32_t *)lb.arrive.p;
        }
        // (the first launch of a search: it also clears the verification flag)
        if (pl.npairs && lb.pair_ctr.ensure(((size_t)pl.npairs + 1) * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        // A streaming search whose predecessor on this handle had exactly this shape finds the state already reset: that
        // search's final selection did it (SelectParams::rs_*) -- one launch and a ~10 us launch gap less per step.
        if (pub_rounds && pl.npairs && !(h->flags & 64)) {
            uint64_t sig = 0x9E3779B97F4A7C15ull;
            const uint64_t parts[] = {(uint64_t)nslots, (uint64_t)nq, (uint64_t)npub, (uint64_t)pl.nqtiles, (uint64_t)pl.npairs, (uint64_t)(uintptr_t)gthr,
                                      (uint64_t)(uintptr_t)qcnt, (uint64_t)(uintptr_t)qthr, (uint64_t)(uintptr_t)pub, (uint64_t)(uintptr_t)arrive,
                                      (uint64_t)(uintptr_t)lb.pair_ctr.p, (uint64_t)(uintptr_t)h->ws_flag.p};
            for (uint64_t v : parts) sig = (sig ^ v) * 0x100000001B3ull;
            reset_sig = sig ? sig : 1;
        }
        if (!(reset_sig && lb.clean_sig == reset_sig)) {
            hipLaunchKernelGGL(init_level_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s, gthr, (int64_t)nslots, qcnt, qthr, nq,
                               reset_flag, pub, npub, arrive, (int64_t)pl.nqtiles, pl.npairs ? (uint32_t *)lb.pair_ctr.p : nullptr, (int64_t)pl.npairs);
            HIP_TRY(hipGetLastError());
        }
        lb.clean_sig = 0; // (until this search's own selection has been enqueued)
    }
    if (pl.npairs && sstride) { // (the sample's own search initialised this level: the pairs' ticket counters are left)
        if (lb.pair_ctr.ensure(((size_t)pl.npairs + 1) * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)lb.pair_ctr.p, 2, (size_t)pl.npairs, s));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)((uint32_t *)lb.pair_ctr.p + pl.npairs), 0, 1, s));
    }
    if (h->ws_lists.ensure((size_t)pl.grid * pl.qt * pl.cap * 8, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory (candidate lists)");
    ScanParams p = {};
    p.xb = h->xb; p.yn = h->yn; p.xq = q_dev; p.xn = xn;
    p.nb = nb; p.nq = nq; p.dp = h->dp; p.k = k; p.cap = pl.cap;
    if (h->approx16) { // bf16 rows and queries: a row is dp / 2 four-byte units
        p.xb = (const float *)h->xb16.p;
        p.dp = h->dp / 2;
    }
    p.nqtiles = pl.nqtiles; p.nchunks = pl.nchunks; p.tiles_base = pl.tiles_base; p.tiles_rem = pl.tiles_rem;
    p.lists = (uint64_t *)h->ws_lists.p; p.gthr = gthr;
    p.qlist = qlist; p.qcnt = qcnt; p.qcap = qcap;
    p.id_base = id_base;
    p.row_mul = row_mul;
    p.vshift = sstride ? svshift : vshift;
    p.skip_mask = sstride ? sstride - 1 : -1;
    p.kslot = kslot;
    p.sparse_epi = !(sstride && !seed_stat) && !pub_rounds; // (not exactly seeded: a statistical estimate, or no seed at all)
    p.pub = pub_rounds ? (uint64_t *)lb.pub.p : nullptr;
    p.arrive = pub_rounds ? (uint32_t *)lb.arrive.p : nullptr;
    p.pub_rounds = pub_rounds;
    p.pub_n = pub_rounds * pl.nchunks * pub_m;
    p.pub_m = pub_m;
    p.pair_ctr = pl.npairs ? (uint32_t *)lb.pair_ctr.p : nullptr;
    p.npairs = pl.npairs;
    if (pl.diff) {
        // (the build launch_scan_cfg picks: up to 4, 12 or 20 queries)
        const int dnq = diff_build_width(nq);
        if (h->ws_qdiff.ensure((size_t)dnq * h->dp * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        const int64_t tot = (int64_t)dnq * h->dp;
        hipLaunchKernelGGL(diff_interleave_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, q_dev, nq, h->dp, dnq, (float *)h->ws_qdiff.p);
        HIP_TRY(hipGetLastError());
        p.xq_diff = (const float *)h->ws_qdiff.p;
    }
    if (pl.nqtiles > 1 && !h->approx16 && pl.tiles_base >= 2 && pl.qt != 256 && !(h->flags & 256)) { // (turn taking: batch launches with real chunks, two workgroups per CU)
        if (h->ws_turn.ensure(2048 * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        if (h->turn_zeroed != h->ws_turn.p) { // (once per allocation: a fill in front of every launch was 10 us of idle GPU per CATH-sized search)
            HIP_TRY(hipMemsetAsync(h->ws_turn.p, 0, 2048 * 4, s));
            h->turn_zeroed = h->ws_turn.p;
        }
        p.cu_turn = (uint32_t *)h->ws_turn.p;
    }
    // the pool: about a tenth of every pair's tiles (none with flags & 2)
    p.pool_tiles = pl.npairs && !(h->flags & 2) ? std::min(std::max(1, (pl.tiles_base + 5) / 10), pl.tiles_base / 4) : 0;
#ifdef KNN355_DEV
    if (p.pool_tiles && getenv("KNN355_POOL_PCT")) // (developer build: the pool's share of every pair's range, in percent)
        p.pool_tiles = std::min(std::max(1, pl.tiles_base * atoi(getenv("KNN355_POOL_PCT")) / 100), pl.tiles_base - 2);
#endif
    if (pub_rounds) {
        if (h->ws_defer.ensure((size_t)pl.grid * pl.qt * pl.dt * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        p.defer = (float *)h->ws_defer.p;
    }
    const bool top = level == 0;
#ifdef KNN355_TRACE
    if (level == (getenv("KNN355_TRACE_LEVEL") ? atoi(getenv("KNN355_TRACE_LEVEL")) : 0)) { // (developer build: which seed level's launch is stamped)
        if (g_trace_buf.ensure((size_t)pl.grid * 128 * 8)) return set_err(KNN_ERR_HIP, "trace: out of device memory");
        HIP_TRY(hipMemsetAsync(g_trace_buf.p, 0, (size_t)pl.grid * 128 * 8, s));
        p.trace = (unsigned long long *)g_trace_buf.p;
        p.ablate = getenv("KNN355_ABLATE") ? atoi(getenv("KNN355_ABLATE")) : 0;
        g_trace_grid = pl.grid;
    }
#endif
    if (top) {
        const int slot = (int)(h->nlaunches % knn_index_s::RING);
        if (!h->ring0[slot]) {
            HIP_TRY(hipEventCreate(&h->ring0[slot]));
            HIP_TRY(hipEventCreate(&h->ring1[slot]));
        }
        h->ev0 = h->ring0[slot];
        h->ev1 = h->ring1[slot];
        h->nlaunches++;
        HIP_TRY(hipEventRecord(h->ev0, s));
    }
    if (pl.qt == 256) rc = launch_scan_big(h, p, pl, s);
    else if (pl.qt == 128) rc = launch_scan_cfg<2, 2, 2, 2>(h, p, pl, s);
    else if (pl.qt == 96) rc = launch_scan16<2, 2, 2, 3>(h, p, pl, s);
    else if (pl.qt == 48) rc = launch_scan16<4, 1, 2, 3>(h, p, pl, s);
    else if (pl.qt == 64) rc = launch_scan_cfg<2, 2, 2, 1>(h, p, pl, s);
    else rc = launch_scan_cfg<4, 1, 2, 1>(h, p, pl, s);
    if (rc) return rc;
    if (top) {
        HIP_TRY(hipEventRecord(h->ev1, s));
        h->last_kernel = pl.name; h->last_qt = pl.qt; h->last_dt = pl.dt; h->last_chunks = pl.nchunks; h->last_grid = pl.grid;
        h->last_seed_stride = pub_rounds ? -pub_rounds * pub_m : sstride; // (negative: tile-minimum seed, keys per workgroup and query)
        h->last_seed_stat = seed_stat ? seed_j : 0;
        h->last_sample_rows = sstride ? view_rows(nb, sstride, p.vshift) : 0;
    }
    // final selection: every query's candidates (seed list + the chunks' survivors) -> sorted top-k
    SelectParams sp = {};
    sp.in = qlist; sp.in_stride = qcap; sp.cnt = qcnt; sp.cap = qcap;
    if (expect_n <= 0 && sstride && !seed_stat) {
        // exact seed: the sample's k-th score admits about k rows per sample-sized slice of the rest of the view
        const double S = (double)view_rows(nb, sstride, p.vshift);
        expect_n = 1.5 * (double)k * (double)nb / std::max(1.0, S) + 1.25 * k;
    }
    if (expect_n <= 0) expect_n = (double)qcap;
    sp.n_expect = (int)std::min<double>(std::min<double>((double)qcap, expect_n), (double)nb); // (never more than the view has rows)
    sp.nq = nq; sp.k = k; sp.metric = h->metric;
    sp.out_keys = out.keys; sp.out_stride = out.keys ? out.keys_stride : 0; sp.out_fill = out.keys_fill;
    sp.D = out.D; sp.I = out.I;
    sp.seed_cnt = out.seed_cnt; sp.seed_gthr = out.seed_gthr; sp.seed_qthr = out.seed_qthr; sp.seed_j = out.seed_j; sp.seed_stat = out.seed_stat;
    sp.seed_nslots = out.seed_nslots;
    sp.qthr = qthr; sp.fail = (int *)h->ws_flag.p;
    if (reset_sig && out.seed_cnt == nullptr) {
        sp.rs_gthr = gthr; sp.rs_qcnt = qcnt; sp.rs_qthr = qthr; sp.rs_nslots = (int)nslots;
        sp.rs_pub = (uint64_t *)lb.pub.p; sp.rs_pub_n = pub_rounds * pl.nchunks * pub_m;
        sp.rs_arrive = (uint32_t *)lb.arrive.p; sp.rs_narrive = pl.nqtiles;
        sp.rs_pair = (uint32_t *)lb.pair_ctr.p; sp.rs_npairs = pl.npairs;
    }
    rc = launch_select(sp, s, &h->ws_tmp);
    if (!rc && sp.rs_gthr) lb.clean_sig = reset_sig;
    if (top && h->done) (void)hipEventRecord(h->done, s); // (everything this search enqueued: see DevBuf::ensure)
    return rc;
}

// queries [nq][dp] already on device (padded); writes sorted keys [nq][k] and/or D/I.
// allow_stat: the caller reads the fail flag once the stream has drained (search_failed) and repeats
// the search with allow_stat = false if it is set.
// The batch a caller handed over, for as long as the search works through it in pieces (the outermost scope wins).
struct BatchScope {
    knn_index_s *h;
    int64_t prev;
    BatchScope(knn_index_s *h_, int64_t nq) : h(h_), prev(h_->batch_nq)
    {
        if (!prev) h->batch_nq = h->batch_hint > 0 ? std::max(nq, h->batch_hint) : nq;
    }
    ~BatchScope() { h->batch_nq = prev; }
};

static int search_keys_impl(knn_index_s *h, const float *q_dev, int64_t nq, int k, uint32_t id_base,
                            uint64_t *keys_out, float *D_out, int64_t *I_out, bool allow_stat, hipStream_t s)
{
    BatchScope scope(h, nq);
    const float *xn = nullptr;
    if (h->metric == KNN_METRIC_L2) {
        if (h->ws_qn.ensure((size_t)nq * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        int rc = norms_dev_impl(q_dev, nq, h->d, h->dp, (float *)h->ws_qn.p, s);
        if (rc) return rc;
        xn = (const float *)h->ws_qn.p;
    }
    if (h->ws_flag.ensure(64, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
    if (h->approx16) {
        // the scan reads bf16 queries (the fp32 ones above gave the exact norms)
        if (h->ws_q16.ensure((size_t)nq * h->dp * 2, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        const int64_t total = nq * h->dp;
        hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)std::min<int64_t>((total / 4 + 255) / 256, 65535)), dim3(256), 0, s, q_dev, total,
                           (__bf16 *)h->ws_q16.p);
        HIP_TRY(hipGetLastError());
        q_dev = (const float *)h->ws_q16.p;
    }
    // Many queries: blocks of 16384 (128 query tiles), one launch each.  A launch keeps two workgroups per CU in flight;
    // with 128 query tiles those share 4 chunks of rows and their 64 MB of query tiles stay in the Infinity Cache, with
    // 1563 query tiles (200 k queries at once) 512 different query tiles are streamed beside ONE chunk and the last of
    // 3.05 rounds of workgroups runs almost alone (200 k x 200 k x 1024: 0.79 -> 0.83 of the fp32 MFMA peak).  The
    // verification flag of a statistically seeded search is cleared by the first block only: it accumulates.
    const int64_t QB = 16384;
    const int64_t nblocks = nq > QB + QB / 2 ? (nq + QB - 1) / QB : 1;
    // The remainder behind the full 128-query tiles.  A query tile costs its whole width whatever it holds: on a database
    // that is streamed from HBM 129 queries took two passes of the 128-query build (10 M rows: 39.2 ms; 128 queries: 20.9).
    // The remainder is searched on its own with the narrowest build that holds it -- <= 32 queries: the streaming build
    // (7.1 ms per 10 M rows), <= 64: the 64-query build (11.6), a batch of 65..96: both (18.4-18.9 ms against 20.2-20.5 for
    // one 128-query pass).  Queries are independent, so the pieces return what one launch would.  Small databases keep one launch (the
    // pieces' own seeds and selections cost more than a padded tile saves).
    struct Piece {
        int64_t q0, m;
    };
    std::vector<Piece> pieces;
    const bool split = h->ntotal >= (1 << 18) && !h->force_qt && !(h->flags & 16384);
    // A database of 32 k .. 262 k rows has no sample pass to seed a 128-query launch with, and the 64-query build seeds
    // itself: a batch of 65..128 queries goes as two 64-query pieces (100 k rows x 100 queries: 0.56 -> 0.41 ms) -- unless the
    // statistical seed can serve the one launch (synchronous callers) and k is large: from k ~ 200 on the one seeded
    // 128-query launch is ahead (100 k rows x 128 queries, k = 1000: 0.53 ms against 0.85 in pieces; k = 100: 0.49 against 0.44).
    const bool split_small = !split && h->ntotal >= (1 << 15) && nq > 64 && nq <= 128 && !h->force_qt && !h->force_chunks && !h->approx16 &&
                             !(h->flags & (8 | 16 | 128 | 2048 | 16384)) && (k <= 200 || !allow_stat || (h->flags & 512));
    for (int64_t b = 0; b < nblocks; b++) {
        int64_t q0 = b * QB, m = nblocks == 1 ? nq : std::min(QB, nq - q0);
        if (split_small) {
            pieces.push_back({q0, 64});
            pieces.push_back({q0 + 64, m - 64});
            continue;
        }
        int64_t full = m / 128 * 128;
        const int64_t r = m - full;
        // A batch the 256 x 256 tile serves (big_tile_pays; synchronous callers: it needs the statistical seed) whose full tiles
        // end in half a 256-query tile: that half goes with the remainder, on the 128 x 128 tile -- 10 M rows x 640 queries:
        // 512 on the wide tile + 128 on the narrow one, not three wide query tiles of which one is half empty.
        const bool may_big = (allow_stat && !(h->flags & (8 | 16 | 512))) || (h->flags & 524288);
        if (split && may_big && !h->force_chunks && full % 256 == 128 && full >= 384 && big_tile_pays(h, h->ntotal, full - 128)) {
            pieces.push_back({q0, full - 128});
            q0 += full - 128;
            m -= full - 128;
            full = 128;
        }
        // (round 4: 33..48 and 65..96 queries have builds of their own width -- 16-query blocks, make_plan -- so a remainder of
        // up to 96 queries is one piece: 10 M rows x 80 queries: 18.7 ms as 64 + 16, ~16 as one 96-query pass)
        const bool q96 = !h->approx16 && !(h->flags & 131072);
        if (!split || m <= 64 || r == 0 || r > 96 || (r > 64 && full && !q96) || (!full && q96)) { // (without the 96-query build: 65..96 behind full tiles cost two narrow passes what the padded tile does)
            pieces.push_back({q0, m});
            continue;
        }
        if (full) pieces.push_back({q0, full});
        if (r > 64 && !q96) {
            pieces.push_back({q0 + full, 64});
            pieces.push_back({q0 + full + 64, r - 64});
        } else {
            pieces.push_back({q0 + full, r});
        }
    }
    for (size_t b = 0; b < pieces.size(); b++) {
        const int64_t q0 = pieces[b].q0, m = pieces[b].m;
        SearchOut out;
        out.keys = keys_out ? keys_out + q0 * k : nullptr; out.keys_stride = k; out.keys_fill = 0;
        out.D = D_out ? D_out + q0 * k : nullptr; out.I = I_out ? I_out + q0 * k : nullptr;
        int rc = search_view(h, q_dev + q0 * (h->approx16 ? h->dp / 2 : h->dp), xn ? xn + q0 : nullptr, m, k, id_base, 1, 3, 0, out, allow_stat, s,
                             b == 0 ? (int *)h->ws_flag.p : nullptr);
        if (rc) return rc;
    }
    return 0;
}

static int check_search_args(knn_index_s *h, const void *q, int64_t nq, int64_t k, const void *D, const void *I)
{
    if (!h) return set_err(KNN_ERR_INVALID, "search: null handle");
    if (nq < 0) return set_err(KNN_ERR_INVALID, "search: negative nq");
    if (k < 1) return set_err(KNN_ERR_INVALID, "search: k must be >= 1");
    if (k > KNN_MAX_K) return set_err(KNN_ERR_UNSUPPORTED, "search: k > 2048 is not supported by the fused top-k");
    if (nq > 0 && (!q || !D || !I)) return set_err(KNN_ERR_INVALID, "search: null pointer");
    if (h->is_view && h->storage_gen->load() != h->view_gen)
        return set_err(KNN_ERR_INVALID, "search: this view is stale (its parent index was grown, reset or freed after the view was made)");
    return 0;
}

// index with no rows: every slot is unfilled
__global__ void fill_empty_kernel(int64_t total, int metric, uint64_t *__restrict__ keys, float *__restrict__ D, int64_t *__restrict__ I)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (keys) keys[i] = KEY_PAD;
    if (D) {
        D[i] = metric == KNN_METRIC_INNER_PRODUCT ? -FLT_MAX : FLT_MAX;
        I[i] = -1;
    }
}

// q_dev: [nq][d] contiguous on device. D_dev/I_dev on device.
static int search_dev_impl(knn_index_s *h, const float *q_dev, int64_t nq, int k, float *D_dev, int64_t *I_dev,
                           uint64_t *keys_dev, uint32_t id_base, bool allow_stat, hipStream_t s)
{
    if (nq == 0) return 0;
    const int64_t total = nq * k;
    if (h->ntotal == 0) {
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, h->metric, keys_dev, D_dev, I_dev);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const float *qp = q_dev;
    if (h->dp != h->d) {
        if (h->ws_q.ensure((size_t)nq * h->dp * 4, h->done, s)) return set_err(KNN_ERR_HIP, "search: out of device memory");
        int64_t tot = nq * h->dp;
        unsigned grid = (unsigned)std::min<int64_t>((tot + 255) / 256, 65535);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, s, q_dev, nq, h->d, (float *)h->ws_q.p, h->dp);
        HIP_TRY(hipGetLastError());
        qp = (const float *)h->ws_q.p;
    }
    return search_keys_impl(h, qp, nq, k, id_base, keys_dev, D_dev, I_dev, allow_stat, s);
}

// after the stream has drained: did a statistically seeded search fail its verification?
static int search_failed(knn_index_s *h, bool *failed)
{
    *failed = false;
    if (!h->ws_flag.p || h->ntotal == 0) return 0;
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, h->ws_flag.p, 4, hipMemcpyDeviceToHost));
    *failed = flag != 0;
    if (flag) {
        h->stat_redo++;
        // read = cleared: a search whose predecessor left the level's state reset (LevelBufs::clean_sig) skips the launch
        // that clears the flag, and a repeat that returns early with an error never reaches its own
        // (on the handle's stream, like every other access to the flag: a NULL-stream memset is not ordered against a
        // non-blocking stream, and completing late it could erase the flag of the repeat search -- ADVICE r4)
        HIP_TRY(hipMemsetAsync(h->ws_flag.p, 0, 4, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int knn_flat_search_dev(knn_handle h, const float *q_dev, int64_t nq, int64_t k, float *D_dev,
                                   int64_t *I_dev, void *stream)
{
    int rc = check_search_args(h, q_dev, nq, k, D_dev, I_dev);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    h->last_ms = -1.f;
    // a caller's stream means "enqueue and return": no way to look at the verification flag, so
    // only the synchronous form may use the statistical seed
    const bool sync = stream == nullptr;
    rc = search_dev_impl(h, q_dev, nq, (int)k, D_dev, I_dev, nullptr, 0, sync, s);
    if (rc) return rc;
    if (sync) {
        HIP_TRY(hipStreamSynchronize(s));
        bool failed = false;
        rc = search_failed(h, &failed);
        if (rc) return rc;
        if (failed) {
            rc = search_dev_impl(h, q_dev, nq, (int)k, D_dev, I_dev, nullptr, 0, false, s);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(s));
        }
    }
    return 0;
}

extern "C" int knn_flat_search_keys_dev(knn_handle h, const float *q_dev, int64_t nq, int64_t k, uint32_t id_base,
                                        uint64_t *keys_dev, void *stream)
{
    int rc = check_search_args(h, q_dev, nq, k, keys_dev, keys_dev);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    h->last_ms = -1.f;
    rc = search_dev_impl(h, q_dev, nq, (int)k, nullptr, nullptr, keys_dev, id_base, false, s);
    if (rc) return rc;
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int knn_merge_keys_dev(knn_handle h, const uint64_t *keys_dev, int32_t nlists, int64_t nq, int64_t k,
                                  float *D_dev, int64_t *I_dev, void *stream)
{
    if (!h) return set_err(KNN_ERR_INVALID, "merge_keys: null handle");
    if (nlists < 1 || nq < 0 || k < 1 || k > KNN_MAX_K) return set_err(KNN_ERR_INVALID, "merge_keys: bad shape");
    if (nq == 0) return 0;
    if (!keys_dev || !D_dev || !I_dev) return set_err(KNN_ERR_INVALID, "merge_keys: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    // one launch whatever nlists * k is, no scratch memory: the selection reads the all-gather buffer in place
    SelectParams sp = {};
    sp.in = keys_dev; sp.lm_lists = nlists; sp.lm_k = (int)k;
    sp.nq = nq; sp.k = (int)k; sp.metric = h->metric;
    sp.D = D_dev; sp.I = I_dev;
    int rc = launch_select(sp, s);
    if (rc) return rc;
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// Symmetric all-vs-all: index.search(x, k) for x = ALL rows of the index (cath/search.py:22-24,
// pfam/proteins_search.py:37,49, pfam/slices/slices_search.py:22-28 -- every flat search of the reference is one).
// Only the score tiles on and above the diagonal are multiplied (flat_scan_kernel<..., SYM>); needs the statistical
// seed (the rows of a database tile have no candidate list of their own in that kernel, only their compact arrays)
// and k <= 1536.  Returns 1 when it ran (D_dev / I_dev hold all n x k results, the verification flag is still to be
// read by the caller), 0 when the plain path should be used, < 0 on error.
// will self_search_symmetric take this search?  (asked BEFORE the caller sets aside device memory for the whole n x k result)
// The statistical sample of a symmetric self-search: every 64th row from 131 072 rows on (Pfam-sized k = 100 / 1000 340.0 / 356.8 ms
// against 348.7 / 366.2 with every 32nd -- the sample pass is 6 % of that search --, 100 k rows 89.2 against 90.8, 60 k rows and
// fewer: within 1 % either way; CATH-sized: 3.06 against 2.87 ms; until late in round 5 the step to 64 came at 2^20 rows).  ONE
// function: the rank j and the sample it is taken from must belong together (a rank worked out for a sparser sample is too tight
// a bound on a denser one: every verification fails and the plain path repeats the search).
static int sym_stat_stride(int64_t n)
{
    int st = n >= (1 << 17) ? 64 : 32;
#ifdef KNN355_DEV
    if (getenv("KNN355_STAT_STRIDE")) st = atoi(getenv("KNN355_STAT_STRIDE"));
#endif
    return st;
}

static bool self_search_symmetric_eligible(const knn_index_s *h, int k, int *j_out = nullptr, int *qcap_out = nullptr)
{
    const int64_t n = h->ntotal;
    if (n < dev_knob("KNN355_SYM_MIN_N", 3000) || k > KNN_REGISTER_SELECT_MAX_K || k >= n || (h->flags & (8 | 16 | 512 | 1024)) || h->force_qt || h->force_chunks || h->approx16) return false;
    const int st = sym_stat_stride(n);
    const int64_t S = view_rows(n, st, 0);
    const int j = stat_seed_rank(S, n, k);
    if (j <= 0) return false;
    const double expect = 1.3 * (double)j * (double)n / (double)S + 1.25 * k;
    const int qcap = (int)std::min<double>(((int64_t)(2.0 * expect) + 1024 + 63) / 64 * 64, 1 << 20);
    if ((double)n * qcap * 8.0 > 24.0 * (1u << 30)) return false;
    if (j_out) *j_out = j;
    if (qcap_out) *qcap_out = qcap;
    return true;
}

// D_host / I_host / d2h given and the result large: the query tiles are served in SYM_GROUPS launches of consecutive tiles, each
// followed by the final selection of ITS rows -- a row's candidates are complete once every query tile up to its own has been
// served (tile (I, J), J >= I, scores rows I against J and J against I) -- and by their download on the copy stream d2h, which
// then overlaps the launches of the later groups (Pfam-sized k = 1000: 2.4 GB of results, 0.15 s behind a 0.38 s search;
// CATH-sized k = 301: 52 MB, 1.3 ms on the wire behind a 2.5 ms search, in four groups).
// Returns 2 then (the caller waits for d2h and checks the verification flag), 1 when the results are in D_dev / I_dev only.
static int self_search_symmetric(knn_index_s *h, int k, float *D_dev, int64_t *I_dev, hipStream_t s, float *D_host = nullptr,
                                 int64_t *I_host = nullptr, hipStream_t d2h = nullptr)
{
    const int64_t n = h->ntotal;
    int j = 0, qcap = 0;
    if (!self_search_symmetric_eligible(h, k, &j, &qcap)) return 0;
    const int st = sym_stat_stride(n);
    const int64_t S = view_rows(n, st, 0);
    const double expect = 1.3 * (double)j * (double)n / (double)S + 1.25 * k;
    ScanPlan pl;
    // (tile shape, list capacity, LDS.  256-row tiles only on demand, flags & 524288: the symmetric launch filters every tile
    // twice and executes half the flops per row pair, its epilogue weighs twice as much beside the K loop -- Pfam-sized
    // k = 100 / 1000: 372.7 / 392.6 ms on 256-row tiles against 349.7 / 377.4 on 128-row tiles, one box)
    make_plan(h, n, n, k, true, pl, false, (h->flags & 524288) != 0);
    if (pl.qt != pl.dt || (pl.qt != 128 && pl.qt != 256)) return 0;
    const int TS = pl.qt; // square tiles of 128 rows (two workgroups per CU) or 256 rows (one: flat_scan_kernel<2, 2, 4, 4>, large indexes)
    const int T = (int)((n + TS - 1) / TS);
    const size_t nslots = (size_t)T * TS;
    LevelBufs &lb = h->ws_level[0];
    lb.clean_sig = 0; // (this search leaves the level's state in its own shape)
    if (lb.qlist.ensure((size_t)n * qcap * 8) || lb.qcnt.ensure((size_t)n * 4) || lb.gthr.ensure(nslots * 4) || lb.qthr.ensure((size_t)n * 4) ||
        h->ws_flag.ensure(64))
        return set_err(KNN_ERR_HIP, "search_self: out of device memory");
    uint64_t *qlist = (uint64_t *)lb.qlist.p;
    uint32_t *qcnt = (uint32_t *)lb.qcnt.p, *gthr = (uint32_t *)lb.gthr.p, *qthr = (uint32_t *)lb.qthr.p;
    const float *xn = h->metric == KNN_METRIC_L2 ? h->yn : nullptr; // the queries are the rows: their norms are the row norms
    // 0. work table: workgroup = (query tile I, a run of database tiles J >= I).  The run length that minimises
    //    rounds x (tiles + half a tile of fixed work), two workgroups per CU.  The table depends on (tiles, CUs) only: it
    //    stays on the device between searches (uploading it between the sample pass and the main launch cost a host
    //    round trip -- 35 us of idle GPU -- per search).
    const int64_t slots = (TS == 256 ? 1 : 2) * (int64_t)std::max(1, h->num_cus);
    constexpr int SYM_GROUPS = 8;
    // (eight groups for gigabytes of result; four from 32 MB on -- CATH-sized, 113 query tiles, 52 MB: cath.search end to end
    // 5.03 -> 4.43 ms with four, 4.62 with five, 4.94 with eight: every group is a launch with its own tail and selection)
    const int want_groups = std::min(SYM_GROUPS, std::max(2, dev_knob("KNN355_SELF_GROUPS", (size_t)n * k * 12 >= ((size_t)256 << 20) && T >= 16 * SYM_GROUPS ? SYM_GROUPS : 4)));
    const bool stream_out = D_host && I_host && d2h && (size_t)n * k * 12 >= ((size_t)dev_knob("KNN355_SELF_STREAM_MIN_MB", 32) << 20) && T >= 16 * want_groups;
    const int groups = stream_out ? want_groups : 1;
    if (h->sym_tiles != T || h->sym_ts != TS || h->sym_groups != groups || !h->ws_sym.p) {
        // (the run length of each group by itself: a group is a launch, and a short one -- the later groups of a CATH-sized
        // index -- fills the slots only with short runs)
        std::vector<SymItem> items;
        h->sym_gstart.assign(1, 0);
        int first_tp = 16;
        for (int g = 0; g < groups; g++) { // (query tiles [T g / groups, T (g + 1) / groups): equal shares of the RESULT; the first group is the longest)
            const int I0 = (int)((int64_t)T * g / groups), I1 = (int)((int64_t)T * (g + 1) / groups);
            int best_tp = 16;
            int64_t best_cost = INT64_MAX;
            for (int tp = dev_knob("KNN355_SYM_MIN_TP", 1); tp <= 96; tp++) {
                int64_t wgs = 0;
                for (int I = I0; I < I1; I++) wgs += (T - I + tp - 1) / tp;
                const int64_t rounds = (wgs + slots - 1) / slots;
                const int64_t cost = rounds * (2 * tp + 1);
                if (cost < best_cost || (cost == best_cost && tp > best_tp)) { best_cost = cost; best_tp = tp; }
            }
            if (g == 0) first_tp = best_tp;
            const size_t at = items.size();
            for (int I = I0; I < I1; I++)
                for (int j0 = I; j0 < T; j0 += best_tp) items.push_back({I, j0, std::min(best_tp, T - j0)});
            // long runs first: the short tails of every query tile fill the last round
            std::stable_sort(items.begin() + at, items.end(), [](const SymItem &a, const SymItem &b) { return a.jcount > b.jcount; });
            h->sym_gstart.push_back((int64_t)items.size());
        }
        const int best_tp = first_tp;
        h->sym_tiles = -1;
        if (h->ws_sym.ensure(items.size() * sizeof(SymItem), h->done, s)) return set_err(KNN_ERR_HIP, "search_self: out of device memory");
        HIP_TRY(hipMemcpyAsync(h->ws_sym.p, items.data(), items.size() * sizeof(SymItem), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s)); // (the table is a local vector)
        h->sym_tiles = T;
        h->sym_ts = TS;
        h->sym_groups = groups;
        h->sym_run = best_tp;
        h->sym_items = (int64_t)items.size();
    }
    const int best_tp = h->sym_run;
    const size_t nitems = (size_t)h->sym_items;
    // 1. the sample pass (every row is a query, the strided sample is the database): thresholds + the sample's own candidates
    SearchOut so;
    so.keys = qlist; so.keys_stride = qcap; so.keys_fill = 0;
    so.seed_cnt = qcnt; so.seed_gthr = gthr; so.seed_qthr = qthr; so.seed_j = j; so.seed_stat = 1; so.seed_nslots = (int64_t)nslots;
    const int k_sample = std::min(k, j + std::max(j >> 2, 8) + 8);
    int rc = search_view(h, h->xb, xn, n, k_sample, 0, st, 0, 1, so, false, s, (int *)h->ws_flag.p);
    if (rc) return rc;
    const size_t per_wg = (size_t)pl.qt * pl.cap * 8;
    const int64_t max_wgs = std::max<int64_t>(slots, (int64_t)((2ull << 30) / per_wg));
    if (h->ws_lists.ensure((size_t)std::min<int64_t>((int64_t)nitems, max_wgs) * per_wg)) return set_err(KNN_ERR_HIP, "search_self: out of device memory (candidate lists)");
    ScanParams p = {};
    p.xb = h->xb; p.yn = h->yn; p.xq = h->xb; p.xn = xn;
    p.nb = n; p.nq = n; p.dp = h->dp; p.k = k; p.cap = pl.cap;
    p.nqtiles = T; p.nchunks = 1; p.tiles_base = 0; p.tiles_rem = 0;
    p.lists = (uint64_t *)h->ws_lists.p; p.gthr = gthr;
    p.qlist = qlist; p.qcnt = qcnt; p.qcap = qcap;
    p.id_base = 0; p.row_mul = 1; p.vshift = 0; p.skip_mask = st - 1;
    p.kslot = knn_kslot(k);
    p.fail = (int *)h->ws_flag.p;
    if (!(h->flags & 256) && TS == 128) {
        if (h->ws_turn.ensure(2048 * 4)) return set_err(KNN_ERR_HIP, "search_self: out of device memory");
        if (h->turn_zeroed != h->ws_turn.p) {
            HIP_TRY(hipMemsetAsync(h->ws_turn.p, 0, 2048 * 4, s));
            h->turn_zeroed = h->ws_turn.p;
        }
        p.cu_turn = (uint32_t *)h->ws_turn.p;
    }
    // + thresholds, per-half counts and bases of the tile's rows (+ 8 KB of slots for the 128-row tile's sparse epilogue, see lds_main in the kernel)
    const size_t lds = pl.lds + (size_t)(2 + 2) * pl.dt * 4 + (TS == 128 ? (size_t)73728 - std::max((size_t)2 * (pl.dt + pl.qt) * 128, (size_t)pl.cap * 8) : 0)
                       + (size_t)dev_knob("KNN355_LDS_PAD", 0); // (developer build: more LDS than a second workgroup leaves room for = one workgroup per CU)
    void (*kern)(ScanParams) = h->metric == KNN_METRIC_L2 ? flat_scan_kernel<2, 2, 2, 2, true, false, true> : flat_scan_kernel<2, 2, 2, 2, false, false, true>;
    if (TS == 256) kern = h->metric == KNN_METRIC_L2 ? flat_scan_kernel<2, 2, 4, 4, true, false, true> : flat_scan_kernel<2, 2, 4, 4, false, false, true>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        const int slot = (int)(h->nlaunches % knn_index_s::RING);
        if (!h->ring0[slot]) {
            HIP_TRY(hipEventCreate(&h->ring0[slot]));
            HIP_TRY(hipEventCreate(&h->ring1[slot]));
        }
        h->ev0 = h->ring0[slot];
        h->ev1 = h->ring1[slot];
        h->nlaunches++;
        HIP_TRY(hipEventRecord(h->ev0, s));
    }
#ifdef KNN355_TRACE
    if (g_trace_buf.ensure((size_t)nitems * 128 * 8)) return set_err(KNN_ERR_HIP, "trace: out of device memory");
    HIP_TRY(hipMemsetAsync(g_trace_buf.p, 0, (size_t)nitems * 128 * 8, s));
    p.trace = (unsigned long long *)g_trace_buf.p;
    p.ablate = getenv("KNN355_ABLATE") ? atoi(getenv("KNN355_ABLATE")) : 0;
    g_trace_grid = (int)std::min<size_t>(nitems, (size_t)max_wgs);
#endif
    // 2 + 3. the launches, each followed by the final selection of the rows it completes (verified against the sample's bound)
    hipEvent_t gev[SYM_GROUPS] = {};
    struct GroupEvents { // (the events of this call go with it, whichever way it returns -- ADVICE r4)
        hipEvent_t *e;
        ~GroupEvents() { for (int i = 0; i < SYM_GROUPS; i++) if (e[i]) (void)hipEventDestroy(e[i]); }
    } gev_guard{gev};
    for (int g = 0; g < groups; g++) {
        const size_t g0 = (size_t)h->sym_gstart[(size_t)g], g1 = (size_t)h->sym_gstart[(size_t)g + 1];
        for (size_t i0 = g0; i0 < g1; i0 += (size_t)max_wgs) {
            const size_t cnt = std::min<size_t>((size_t)max_wgs, g1 - i0);
            p.sym_items = (const SymItem *)h->ws_sym.p + i0;
            hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(256), lds, s, p);
            HIP_TRY(hipGetLastError());
        }
        if (g == groups - 1) HIP_TRY(hipEventRecord(h->ev1, s));
        const int64_t r0 = std::min<int64_t>(n, (int64_t)T * g / groups * TS), r1 = std::min<int64_t>(n, (int64_t)T * (g + 1) / groups * TS);
        SelectParams sp = {};
        sp.in = qlist + (size_t)r0 * qcap; sp.in_stride = qcap; sp.cnt = qcnt + r0; sp.cap = qcap;
        sp.n_expect = (int)std::min<double>((double)qcap, expect);
        sp.nq = r1 - r0; sp.k = k; sp.metric = h->metric;
        sp.D = D_dev + (size_t)r0 * k; sp.I = I_dev + (size_t)r0 * k;
        sp.qthr = qthr + r0; sp.fail = (int *)h->ws_flag.p;
        rc = launch_select(sp, s, &h->ws_tmp);
        if (rc) return rc;
        if (stream_out) {
            HIP_TRY(hipEventCreateWithFlags(&gev[g], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(gev[g], s));
        }
    }
    h->last_kernel = TS == 256 ? "flat_scan_q256_d256_sym" : "flat_scan_q128_d128_sym"; h->last_qt = TS; h->last_dt = TS; h->last_chunks = best_tp; h->last_grid = (int)nitems;
    h->last_seed_stride = st; h->last_seed_stat = j; h->last_sample_rows = S;
    if (h->done) (void)hipEventRecord(h->done, s); // (everything this search enqueued on the handle's buffers: see DevBuf::ensure)
    h->sym_searches++;
    if (!stream_out) return 1;
    // the downloads, group by group behind their selections (everything above is enqueued: a copy into pageable memory may
    // block this thread as long as it likes)
    hipError_t e = hipSuccess;
    for (int g = 0; g < groups; g++) {
        const int64_t r0 = std::min<int64_t>(n, (int64_t)T * g / groups * TS), r1 = std::min<int64_t>(n, (int64_t)T * (g + 1) / groups * TS);
        if (e == hipSuccess) e = hipStreamWaitEvent(d2h, gev[g], 0);
        if (e == hipSuccess && r1 > r0) e = hipMemcpyAsync(D_host + (size_t)r0 * k, D_dev + (size_t)r0 * k, (size_t)(r1 - r0) * k * 4, hipMemcpyDeviceToHost, d2h);
        if (e == hipSuccess && r1 > r0) e = hipMemcpyAsync(I_host + (size_t)r0 * k, I_dev + (size_t)r0 * k, (size_t)(r1 - r0) * k * 8, hipMemcpyDeviceToHost, d2h);
    }
    // (always: a copy already queued writes into the caller's arrays)
    const hipError_t es = hipStreamSynchronize(d2h);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return set_err(KNN_ERR_HIP, hipGetErrorString(e));
    return 2;
}

// Copy streams of the pipelined host search: one set per device for the whole process, created on
// first use (a HIP stream costs milliseconds to create and callers such as cath/search.py build a
// fresh index per file).  Whoever holds `mu` pipelines; a concurrent host search on the same
// device simply runs its batches one after the other on its own stream.
struct CopyPipes {
    std::mutex mu;
    hipStream_t h2d = nullptr, d2h = nullptr;
    hipEvent_t ev_query[2] = {nullptr, nullptr}, ev_batch[2] = {nullptr, nullptr};
};
static CopyPipes g_pipes[64];

// Search with results (and, unless q_host is null, queries) in host memory.  q_host == nullptr:
// the queries are the index's own rows [self_row0, self_row0 + nq) -- already on the device and
// padded, nothing to upload.  Caller holds h->mu.
static int host_search(knn_index_s *h, const float *q_host, int64_t self_row0, int64_t nq, int64_t k, float *D_host,
                       int64_t *I_host)
{
    int rc = 0;
    BatchScope scope(h, nq);
    HIP_TRY(hipSetDevice(h->device));
    // Query batches bound the device workspace; 16384 queries keep >= 128 query tiles in flight.
    // Batches are pipelined: while batch b is scanned, a helper thread downloads the results of
    // batch b-1 (pageable host memory: the copy blocks its caller) and this thread uploads the
    // queries of batch b+1, each on its own stream with its own staging buffers.
    int64_t QB = 16384;
    // A single batch with a large result (CATH20-sized all-vs-all with 300 hits: 52 MB, a millisecond on
    // the wire) is split in two so that half of its download overlaps the scan; small results are not
    // worth the extra launches and the extra streams of the pipelined path.
    // (two halves: measured 5.1-5.2 ms for the CATH-sized k=301 search_self against 5.4 in one piece and 5.4 in four --
    // every piece pays its own sample pass and selections)
    if (nq <= QB && (size_t)nq * k * 12 >= ((size_t)24 << 20) && nq >= 4096)
        QB = std::max<int64_t>(2048, ((nq + 1) / 2 + 127) / 128 * 128);
    const int64_t bq = std::min(nq, QB);
    const int64_t nbatches = (nq + QB - 1) / QB;
    DevBuf *qbuf[2] = {&h->ws_tmp2, &h->ws_tmp3}, *dbuf[2] = {&h->ws_D, &h->ws_D1}, *ibuf[2] = {&h->ws_I, &h->ws_I1};
    const int nslots = nbatches > 1 ? 2 : 1;
    for (int i = 0; i < nslots; i++)
        if ((q_host && qbuf[i]->ensure((size_t)bq * h->d * 4)) || dbuf[i]->ensure((size_t)bq * k * 4) || ibuf[i]->ensure((size_t)bq * k * 8))
            return set_err(KNN_ERR_HIP, "search: out of device memory");
    if (!h->flag_host) HIP_TRY(hipHostMalloc((void **)&h->flag_host, 64, hipHostMallocDefault));
    // one batch of queries on the device: uploaded rows ([m][d]) or the index's own rows ([m][dp]).
    // The verification flag of a statistically seeded batch travels to flag_host[slot] behind it.
    auto scan = [&](int slot, int64_t b0, int64_t m, bool allow_stat) -> int {
        int r;
        if (q_host)
            r = search_dev_impl(h, (const float *)qbuf[slot]->p, m, (int)k, (float *)dbuf[slot]->p, (int64_t *)ibuf[slot]->p, nullptr, 0, allow_stat, h->stream);
        else
            r = search_keys_impl(h, h->xb + (size_t)(self_row0 + b0) * h->dp, m, (int)k, 0, nullptr, (float *)dbuf[slot]->p,
                                 (int64_t *)ibuf[slot]->p, allow_stat, h->stream);
        if (r) return r;
        h->flag_host[slot] = 0;
        if (allow_stat && h->ntotal > 0) HIP_TRY(hipMemcpyAsync(&h->flag_host[slot], h->ws_flag.p, 4, hipMemcpyDeviceToHost, h->stream));
        return 0;
    };
    // plain form: one batch at a time on the handle's stream (nothing to overlap with, or the
    // device's copy streams are taken)
    auto plain = [&](int64_t b0, int64_t m, bool allow_stat) -> int {
        if (q_host) HIP_TRY(hipMemcpyAsync(qbuf[0]->p, q_host + b0 * h->d, (size_t)m * h->d * 4, hipMemcpyHostToDevice, h->stream));
        int r = scan(0, b0, m, allow_stat);
        if (r) return r;
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->flag_host[0]) { // the statistical threshold was too tight for some query: once more without it
            h->stat_redo++;
            r = scan(0, b0, m, false);
            if (r) return r;
        }
        HIP_TRY(hipMemcpyAsync(D_host + b0 * k, dbuf[0]->p, (size_t)m * k * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(I_host + b0 * k, ibuf[0]->p, (size_t)m * k * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        float ms = 0.f;
        if (h->ntotal > 0 && h->ev0 && hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_ms += ms;
        return 0;
    };
    CopyPipes &cp = g_pipes[h->device & 63];
    std::unique_lock<std::mutex> pipes(cp.mu, std::defer_lock);
    if (nbatches == 1 || !pipes.try_lock()) {
        h->last_ms = 0.f;
        for (int64_t b0 = 0; b0 < nq; b0 += QB) {
            rc = plain(b0, std::min(QB, nq - b0), true);
            if (rc) return rc;
        }
        return 0;
    }
    if (!cp.h2d) {
        HIP_TRY(hipStreamCreateWithFlags(&cp.h2d, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&cp.d2h, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&cp.ev_query[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&cp.ev_batch[i], hipEventDisableTiming));
        }
    }
    struct Download {
        std::thread th;
        hipError_t err = hipSuccess;
        hipEvent_t t0 = nullptr, t1 = nullptr; // scan events of the batch
        int64_t batch = -1;
        bool failed = false;
    } dl[2];
    float ms_total = 0.f;
    hipError_t dl_err = hipSuccess;
    std::vector<int64_t> redo; // batches whose statistical threshold failed its verification
    auto join = [&](int slot) {
        if (!dl[slot].th.joinable()) return;
        dl[slot].th.join();
        if (dl[slot].err != hipSuccess) dl_err = dl[slot].err;
        if (dl[slot].failed) redo.push_back(dl[slot].batch);
        float ms = 0.f;
        if (dl[slot].t0 && hipEventElapsedTime(&ms, dl[slot].t0, dl[slot].t1) == hipSuccess) ms_total += ms;
    };
    auto upload = [&](int64_t b) -> hipError_t {
        if (!q_host) return hipSuccess;
        const int slot = (int)(b & 1);
        const int64_t b0 = b * QB, m = std::min(QB, nq - b0);
        hipError_t e = hipMemcpyAsync(qbuf[slot]->p, q_host + b0 * h->d, (size_t)m * h->d * 4, hipMemcpyHostToDevice, cp.h2d);
        if (e == hipSuccess) e = hipEventRecord(cp.ev_query[slot], cp.h2d);
        return e;
    };
    hipError_t e = upload(0);
    for (int64_t b = 0; b < nbatches && e == hipSuccess && rc == 0; b++) {
        const int slot = (int)(b & 1);
        const int64_t b0 = b * QB, m = std::min(QB, nq - b0);
        if (q_host) e = hipStreamWaitEvent(h->stream, cp.ev_query[slot], 0);
        if (e != hipSuccess) break;
        rc = scan(slot, b0, m, true);
        if (rc) break;
        e = hipEventRecord(cp.ev_batch[slot], h->stream);
        if (e != hipSuccess) break;
        dl[slot].t0 = h->ntotal > 0 ? h->ev0 : nullptr;
        dl[slot].t1 = h->ev1;
        dl[slot].err = hipSuccess;
        dl[slot].batch = b;
        dl[slot].failed = false;
        CopyPipes *cpp = &cp;
        dl[slot].th = std::thread([=, &dl]() {
            hipError_t r = hipSetDevice(h->device);
            if (r == hipSuccess) r = hipEventSynchronize(cpp->ev_batch[slot]);
            if (r == hipSuccess && h->flag_host[slot]) {
                dl[slot].failed = true; // repeated below, after the pipeline has drained
                dl[slot].err = hipSuccess;
                return;
            }
            if (r == hipSuccess) r = hipMemcpyAsync(D_host + b0 * k, dbuf[slot]->p, (size_t)m * k * 4, hipMemcpyDeviceToHost, cpp->d2h);
            if (r == hipSuccess) r = hipMemcpyAsync(I_host + b0 * k, ibuf[slot]->p, (size_t)m * k * 8, hipMemcpyDeviceToHost, cpp->d2h);
            if (r == hipSuccess) r = hipStreamSynchronize(cpp->d2h);
            dl[slot].err = r;
        });
        if (b + 1 < nbatches) {
            // the other slot's buffers: its scan (batch b-1) must be over and downloaded
            join(slot ^ 1);
            e = upload(b + 1);
        }
    }
    join(0);
    join(1);
    if (rc) return rc;
    if (e != hipSuccess) return set_err(KNN_ERR_HIP, std::string("search: ") + hipGetErrorString(e));
    if (dl_err != hipSuccess) return set_err(KNN_ERR_HIP, std::string("search: result download failed: ") + hipGetErrorString(dl_err));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->last_ms = ms_total;
    for (int64_t b : redo) {
        h->stat_redo++;
        const int64_t b0 = b * QB;
        rc = plain(b0, std::min(QB, nq - b0), false);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int knn_flat_search(knn_handle h, const float *q_host, int64_t nq, int64_t k, float *D_host,
                               int64_t *I_host)
{
    int rc = check_search_args(h, q_host, nq, k, D_host, I_host);
    if (rc) return rc;
    if (nq == 0) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    return host_search(h, q_host, 0, nq, k, D_host, I_host);
}

extern "C" int knn_flat_search_self(knn_handle h, int64_t row0, int64_t nrows, int64_t k, float *D_host, int64_t *I_host)
{
    if (!h) return set_err(KNN_ERR_INVALID, "search_self: null handle");
    if (nrows == 0) return 0;
    int rc = check_search_args(h, D_host, nrows, k, D_host, I_host);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    if (row0 < 0 || nrows < 0 || row0 + nrows > h->ntotal) return set_err(KNN_ERR_INVALID, "search_self: row range out of bounds");
    if (row0 == 0 && nrows == h->ntotal && (size_t)nrows * k * 12 <= ((size_t)16 << 30) && self_search_symmetric_eligible(h, (int)k)) {
        // every row against every row: half the score tiles suffice (self_search_symmetric)
        HIP_TRY(hipSetDevice(h->device));
        if (h->ws_D.ensure((size_t)nrows * k * 4) || h->ws_I.ensure((size_t)nrows * k * 8)) return set_err(KNN_ERR_HIP, "search_self: out of device memory");
        h->last_ms = -1.f;
        // (a large result leaves in pieces while the later query tiles are still being served: the copy stream of the device,
        // if no other host search of this process is using it)
        CopyPipes &cp = g_pipes[h->device & 63];
        std::unique_lock<std::mutex> pipes(cp.mu, std::defer_lock);
        hipStream_t d2h = nullptr;
        static const bool stream_off = getenv("KNN355_SELF_STREAM") && atoi(getenv("KNN355_SELF_STREAM")) == 0; // (A/B: the result in one piece behind the search)
        if (!stream_off && (size_t)nrows * k * 12 >= ((size_t)dev_knob("KNN355_SELF_STREAM_MIN_MB", 32) << 20) && pipes.try_lock()) {
            if (!cp.h2d) {
                HIP_TRY(hipStreamCreateWithFlags(&cp.h2d, hipStreamNonBlocking));
                HIP_TRY(hipStreamCreateWithFlags(&cp.d2h, hipStreamNonBlocking));
                for (int i = 0; i < 2; i++) {
                    HIP_TRY(hipEventCreateWithFlags(&cp.ev_query[i], hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&cp.ev_batch[i], hipEventDisableTiming));
                }
            }
            d2h = cp.d2h;
        }
        rc = self_search_symmetric(h, (int)k, (float *)h->ws_D.p, (int64_t *)h->ws_I.p, h->stream, D_host, I_host, d2h);
        if (pipes.owns_lock()) pipes.unlock();
        if (rc < 0) return rc;
        if (rc == 2) { // (the rows went out behind their groups' selections)
            HIP_TRY(hipStreamSynchronize(h->stream));
            bool failed = false;
            rc = search_failed(h, &failed);
            if (rc) return rc;
            if ((size_t)nrows * k * 12 > ((size_t)1 << 30)) { // (gigabytes of result staging are not kept with the handle)
                h->ws_D.release();
                h->ws_I.release();
            }
            if (!failed) return 0;
            // a threshold estimate was too tight or a candidate array overflowed: the plain path repeats the search
        } else if (rc == 1) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            bool failed = false;
            rc = search_failed(h, &failed);
            if (rc) return rc;
            if (!failed) {
                HIP_TRY(hipMemcpyAsync(D_host, h->ws_D.p, (size_t)nrows * k * 4, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipMemcpyAsync(I_host, h->ws_I.p, (size_t)nrows * k * 8, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
                if ((size_t)nrows * k * 12 > ((size_t)1 << 30)) { // (gigabytes of result staging are not kept with the handle)
                    h->ws_D.release();
                    h->ws_I.release();
                }
                return 0;
            }
            // a threshold estimate was too tight or a candidate array overflowed: the plain path repeats the search
        }
        h->ws_D.release(); // (host_search works in batches: it takes what it needs)
        h->ws_I.release();
    }
    return host_search(h, nullptr, row0, nrows, k, D_host, I_host);
}

// the same with the results left on the device (D_dev [ntotal][k], I_dev [ntotal][k]); synchronous
extern "C" int knn_flat_search_self_dev(knn_handle h, int64_t k, float *D_dev, int64_t *I_dev)
{
    if (!h) return set_err(KNN_ERR_INVALID, "search_self: null handle");
    if (h->ntotal == 0) return 0;
    int rc = check_search_args(h, D_dev, h->ntotal, k, D_dev, I_dev);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    h->last_ms = -1.f;
    rc = self_search_symmetric(h, (int)k, D_dev, I_dev, s);
    if (rc < 0) return rc;
    if (rc == 1) {
        HIP_TRY(hipStreamSynchronize(s));
        bool failed = false;
        rc = search_failed(h, &failed);
        if (rc) return rc;
        if (!failed) return 0;
    }
    const int64_t QB = 16384;
    BatchScope scope(h, h->ntotal);
    for (int64_t b0 = 0; b0 < h->ntotal; b0 += QB) {
        const int64_t m = std::min(QB, h->ntotal - b0);
        for (int attempt = 0; attempt < 2; attempt++) { // (statistical seed first, plain if its verification fails)
            rc = search_keys_impl(h, h->xb + (size_t)b0 * h->dp, m, (int)k, 0, nullptr, D_dev + b0 * k, I_dev + b0 * k, attempt == 0, s);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(s));
            bool failed = false;
            rc = search_failed(h, &failed);
            if (rc) return rc;
            if (!failed) break;
        }
    }
    return 0;
}

extern "C" int knn_flat_normalize_rows(knn_handle h)
{
    if (!h) return set_err(KNN_ERR_INVALID, "normalize_rows: null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->is_view) return set_err(KNN_ERR_INVALID, "normalize_rows: a view is read-only");
    if (h->ntotal == 0) return 0;
    HIP_TRY(hipSetDevice(h->device));
    int rc = normalize_dev_impl(h->xb, h->ntotal, h->d, h->dp, h->stream);
    if (rc) return rc;
    rc = norms_dev_impl(h->xb, h->ntotal, h->d, h->dp, h->yn, h->stream);
    if (rc) return rc;
    rc = approx16_sync_rows(h, 0, h->ntotal, h->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int knn_last_scan_info(knn_handle h, char *name, int32_t name_len, int32_t *query_tile, int32_t *db_tile,
                                  int32_t *nchunks, int32_t *grid)
{
    if (!h) return set_err(KNN_ERR_INVALID, "null handle");
    if (name && name_len > 0) {
        strncpy(name, h->last_kernel.c_str(), (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (query_tile) *query_tile = h->last_qt;
    if (db_tile) *db_tile = h->last_dt;
    if (nchunks) *nchunks = h->last_chunks;
    if (grid) *grid = h->last_grid;
    return 0;
}

extern "C" float knn_last_scan_ms(knn_handle h)
{
    if (!h) return -1.f;
    std::lock_guard<std::mutex> lk(h->mu);
    float ms = 0.f;
    if (h->last_ms >= 0.f) return h->last_ms;
    if (!h->ev0 || hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.f;
    return ms;
}

extern "C" int32_t knn_scan_times(knn_handle h, float *out_ms, int32_t max_n)
{
    if (!h || !out_ms || max_n < 0) return set_err(KNN_ERR_INVALID, "scan_times: bad arguments");
    std::lock_guard<std::mutex> lk(h->mu);
    int64_t n = std::min<int64_t>(std::min<int64_t>(h->nlaunches, knn_index_s::RING), max_n);
    // oldest first among the n most recent launches; launches whose events have not completed read -1
    for (int64_t i = 0; i < n; i++) {
        int64_t idx = h->nlaunches - n + i;
        int slot = (int)(idx % knn_index_s::RING);
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, h->ring0[slot], h->ring1[slot]) != hipSuccess) ms = -1.f;
        out_ms[i] = ms;
    }
    return (int32_t)n;
}

// Page-locked host memory for result arrays: the download of a search result into it is one DMA at PCIe
// line rate (into pageable memory the runtime stages through its own pinned buffer at a third of that).
extern "C" void *knn_host_alloc(int64_t bytes)
{
    if (bytes <= 0) return nullptr;
    if (ensure_device(g_device)) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable) != hipSuccess) {
        set_err(KNN_ERR_HIP, "host_alloc: hipHostMalloc failed");
        return nullptr;
    }
    return p;
}
extern "C" void knn_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int knn_last_seed_info(knn_handle h, int32_t *seed_stride, int32_t *stat_rank, int64_t *stat_redo, int64_t *sample_rows)
{
    if (!h) return set_err(KNN_ERR_INVALID, "null handle");
    if (sample_rows) *sample_rows = h->last_sample_rows;
    if (seed_stride) *seed_stride = h->last_seed_stride;
    if (stat_rank) *stat_rank = h->last_seed_stat;
    if (stat_redo) *stat_redo = h->stat_redo;
    return 0;
}

extern "C" int knn_flat_reserve(knn_handle h, int64_t nrows)
{
    if (!h) return set_err(KNN_ERR_INVALID, "reserve: null handle");
    if (nrows < 0 || nrows > 0xFFFFFFF0ll) return set_err(KNN_ERR_INVALID, "reserve: bad row count");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->is_view) return set_err(KNN_ERR_INVALID, "reserve: a view is read-only");
    HIP_TRY(hipSetDevice(h->device));
    if (nrows <= h->cap_rows) return 0;
    // exact-size allocation (grow_index over-allocates only when it has to guess)
    const size_t row_bytes = (size_t)h->dp * 4;
    size_t xb_got = 0, yn_got = 0;
    float *nxb = (float *)pool_alloc((size_t)nrows * row_bytes, h->device, &xb_got);
    if (!nxb) return set_err(KNN_ERR_HIP, "reserve: out of device memory");
    float *nyn = (float *)pool_alloc(((size_t)nrows + 64) * 4, h->device, &yn_got);
    if (!nyn) {
        g_pool.give(nxb, xb_got, h->device);
        return set_err(KNN_ERR_HIP, "reserve: out of device memory");
    }
    if (h->ntotal > 0) {
        HIP_TRY(hipMemcpyAsync(nxb, h->xb, (size_t)h->ntotal * row_bytes, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(nyn, h->yn, (size_t)h->ntotal * 4, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->xb) {
        h->storage_gen->fetch_add(1);
        (void)hipDeviceSynchronize();
        g_pool.give(h->xb, h->xb_bytes, h->device);
    }
    if (h->yn) g_pool.give(h->yn, h->yn_bytes, h->device);
    h->xb = nxb;
    h->yn = nyn;
    h->xb_bytes = xb_got;
    h->yn_bytes = yn_got;
    h->cap_rows = nrows;
    return 0;
}

extern "C" int knn_set_tuning(knn_handle h, int32_t query_tile, int32_t nchunks, int32_t flags)
{
    if (!h) return set_err(KNN_ERR_INVALID, "null handle");
    if (query_tile != 0 && query_tile != 32 && query_tile != 48 && query_tile != 64 && query_tile != 96 && query_tile != 128 && query_tile != 256)
        return set_err(KNN_ERR_INVALID, "set_tuning: query_tile must be 0, 32, 48, 64, 96, 128 or 256");
    std::lock_guard<std::mutex> lk(h->mu);
    h->force_qt = query_tile;
    h->force_chunks = nchunks;
    h->flags = flags & ~(3 << 12); // (bit 17 = 131072: plans without the 16-query-block builds, see make_plan)
    h->pub_rounds_force = (flags >> 12) & 3; // bits 12-13: publication rounds of the tile-minimum seed (0: the host's choice)
    return 0;
}

// ---- measurement aid: the box's read rate over the index's own rows -----------------------------------------------
__global__ __launch_bounds__(256) void read_rate_kernel(const f32x4 *__restrict__ p, size_t n4, float *__restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (; i + 3 * stride < n4; i += 4 * stride) { // four independent 16-byte loads in flight per lane
        const f32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a + b + c + d;
    }
    for (; i < n4; i += stride) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0]; // (keeps the loads alive)
}

extern "C" int knn_flat_read_rate(knn_handle h, int32_t reps, float *best_ms, int64_t *bytes_read)
{
    if (!h || !best_ms || !bytes_read) return set_err(KNN_ERR_INVALID, "read_rate: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = (size_t)h->ntotal * h->dp * 4;
    *bytes_read = (int64_t)bytes;
    *best_ms = 0.0f;
    if (!bytes) return 0;
    if (h->ws_flag.ensure(64, h->done, h->stream)) return set_err(KNN_ERR_HIP, "read_rate: out of device memory");
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float best = FLT_MAX;
    for (int grid : {2048, 4096, 8192})
        for (int r = 0; r < std::max(1, reps); r++) {
            HIP_TRY(hipEventRecord(e0, h->stream));
            hipLaunchKernelGGL(read_rate_kernel, dim3(grid), dim3(256), 0, h->stream, (const f32x4 *)h->xb, bytes / 16, (float *)h->ws_flag.p + 8);
            HIP_TRY(hipEventRecord(e1, h->stream));
            HIP_TRY(hipEventSynchronize(e1));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *best_ms = best;
    return 0;
}

// ---- measurement aid: the box's fp32 MFMA rate -----------------------------------------------------------------
// every wave issues back-to-back v_mfma_f32_32x32x2_f32 on four independent accumulators, operands in registers, no
// memory traffic: what the matrix pipes of THIS box sustain under load (the chip lowers its clock: the in-kernel clock is
// d(s_memtime) / d(s_memrealtime) x 100 MHz), to set beside the data sheet's 157.3 TFLOP/s
__global__ __launch_bounds__(256) void mfma_rate_kernel(float *out, int iters, unsigned long long *clk)
{
    float x[8], y[8];
    unsigned s0 = (blockIdx.x * 256u + threadIdx.x) * 16u;
    for (int i = 0; i < 16; i++) { // (random operands: constant ones toggle fewer wires and run at a higher clock than real data)
        unsigned s = (s0 + i) * 747796405u + 2891336453u;
        s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
        s = (s >> 22) ^ s;
        const float v = (float)(s & 0xFFFFFF) / 8388608.0f - 1.0f;
        if (i < 8) x[i] = v;
        else y[i - 8] = v;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u + 1], y[u], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u + 1], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u + 1], y[u + 1], a3, 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) out[0] = s; // (keeps the accumulators alive)
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

extern "C" int knn_mfma_rate(int32_t warm_ms, float *tflops, float *clock_mhz)
{
    if (!tflops) return set_err(KNN_ERR_INVALID, "mfma_rate: null pointer");
    int rc = ensure_device(g_device);
    if (rc) return rc;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g_device);
    const int grid = 2 * std::max(1, cus), iters = 8000; // (two workgroups per CU: two waves per SIMD, as in the scan)
    size_t got_o = 0, got_c = 0;
    float *o = (float *)pool_alloc(64, g_device, &got_o);
    unsigned long long *c = (unsigned long long *)pool_alloc((size_t)grid * 16, g_device, &got_c);
    hipStream_t s = g_streams.take(g_device);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (s) { (void)hipStreamSynchronize(s); g_streams.give(g_device, s); }
        if (o) g_pool.give(o, got_o, g_device);
        if (c) g_pool.give(c, got_c, g_device);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    if (!o || !c || !s || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        cleanup();
        return set_err(KNN_ERR_HIP, "mfma_rate: out of resources");
    }
    // back-to-back launches until warm_ms of them have run (the clock settles under load), then the best of eight
    const double flop = (double)grid * 4 /* waves */ * iters * 16.0 * (32.0 * 32 * 2 * 2);
    float best = FLT_MAX, total = 0.0f;
    double mhz = 0.0;
    std::vector<unsigned long long> hc((size_t)grid * 2);
    int timed = 0;
    for (int r = 0; r < 4000 && timed < 8; r++) {
        hipError_t e = hipEventRecord(e0, s);
        hipLaunchKernelGGL(mfma_rate_kernel, dim3(grid), dim3(256), 0, s, o, iters, c);
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            cleanup();
            return set_err(KNN_ERR_HIP, std::string("mfma_rate: ") + hipGetErrorString(e));
        }
        total += ms;
        if (total < (float)warm_ms) continue;
        timed++;
        if (ms < best) {
            best = ms;
            if (hipMemcpy(hc.data(), c, hc.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                double acc = 0.0;
                for (int b = 0; b < grid; b++) acc += (double)hc[2 * b] / (double)std::max<unsigned long long>(1, hc[2 * b + 1]) * 100.0;
                mhz = acc / grid;
            }
        }
    }
    cleanup();
    *tflops = (float)(flop / (best * 1e-3) / 1e12);
    if (clock_mhz) *clock_mhz = (float)mhz;
    return 0;
}

extern "C" int knn_flat_set_batch(knn_handle h, int64_t nq_whole)
{
    if (!h) return set_err(KNN_ERR_INVALID, "null handle");
    if (nq_whole < 0) return set_err(KNN_ERR_INVALID, "set_batch: negative batch size");
    std::lock_guard<std::mutex> lk(h->mu);
    h->batch_hint = nq_whole;
    return 0;
}

// ---- gather distances -------------------------------------------------------
extern "C" int knn_gather_distances(knn_handle h, const float *q_host, int64_t nq, const int64_t *cand_ids,
                                    const int64_t *cand_offsets, float *out_host)
{
    if (!h) return set_err(KNN_ERR_INVALID, "gather_distances: null handle");
    if (nq < 0) return set_err(KNN_ERR_INVALID, "gather_distances: negative nq");
    if (nq == 0) return 0;
    if (!q_host || !cand_offsets) return set_err(KNN_ERR_INVALID, "gather_distances: null pointer");
    const int64_t np = cand_offsets[nq];
    if (np == 0) return 0;
    if (!cand_ids || !out_host) return set_err(KNN_ERR_INVALID, "gather_distances: null pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    std::vector<int32_t> pq((size_t)np);
    for (int64_t i = 0; i < nq; i++) {
        if (cand_offsets[i + 1] < cand_offsets[i]) return set_err(KNN_ERR_INVALID, "gather_distances: offsets not monotone");
        for (int64_t p = cand_offsets[i]; p < cand_offsets[i + 1]; p++) {
            if (cand_ids[p] < 0 || cand_ids[p] >= h->ntotal) return set_err(KNN_ERR_INVALID, "gather_distances: candidate id out of range");
            pq[(size_t)p] = (int32_t)i;
        }
    }
    hipStream_t s = h->stream;
    if (h->ws_tmp2.ensure((size_t)nq * h->d * 4) || h->ws_q.ensure((size_t)nq * h->dp * 4) || h->ws_qn.ensure((size_t)nq * 4) ||
        h->ws_I.ensure((size_t)np * 8) || h->ws_tmp.ensure((size_t)np * 4) || h->ws_D.ensure((size_t)np * 4))
        return set_err(KNN_ERR_HIP, "gather_distances: out of device memory");
    HIP_TRY(hipMemcpyAsync(h->ws_tmp2.p, q_host, (size_t)nq * h->d * 4, hipMemcpyHostToDevice, s));
    {
        int64_t tot = nq * h->dp;
        unsigned grid = (unsigned)std::min<int64_t>((tot + 255) / 256, 65535);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, s, (const float *)h->ws_tmp2.p, nq, h->d, (float *)h->ws_q.p, h->dp);
        HIP_TRY(hipGetLastError());
    }
    int rc = norms_dev_impl((const float *)h->ws_q.p, nq, h->d, h->dp, (float *)h->ws_qn.p, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->ws_I.p, cand_ids, (size_t)np * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->ws_tmp.p, pq.data(), (size_t)np * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(pair_distance_kernel, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, s, h->xb, h->yn,
                       (const float *)h->ws_q.p, (const float *)h->ws_qn.p, h->dp, h->metric, np,
                       (const int32_t *)h->ws_tmp.p, (const int64_t *)h->ws_I.p, (float *)h->ws_D.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, h->ws_D.p, (size_t)np * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------
// Several GPUs without torch: RCCL inside the library.  One process per GPU; rank 0 makes a
// unique id (knn_comm_unique_id), the caller hands it to the other ranks by whatever means it
// has (a file, MPI, a socket), every rank creates its communicator, and a sharded search is
// local scan -> ncclAllGather of the [nq][k] packed keys -> selection, all on one stream.
// librccl is opened on first use (dlopen): a process that shards through torch.distributed
// instead (bench.py, sharded.py) never loads a second copy.
// ---------------------------------------------------------------------------
#include <dlfcn.h>
struct Id128 { char b[128]; };
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value: 128 bytes */ Id128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;
static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *L = nullptr;
    for (const char *nm : names)
        if ((L = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!L) return set_err(KNN_ERR_UNSUPPORTED, std::string("comm: librccl not found (") + dlerror() + ")");
    RcclApi a;
    a.GetUniqueId = (int (*)(void *))dlsym(L, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(L, "ncclCommInitRank");
    a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(L, "ncclAllGather");
    a.CommDestroy = (int (*)(void *))dlsym(L, "ncclCommDestroy");
    a.GetErrorString = (const char *(*)(int))dlsym(L, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.CommDestroy || !a.GetErrorString)
        return set_err(KNN_ERR_UNSUPPORTED, "comm: librccl lacks a required symbol");
    a.lib = L;
    g_rccl = a;
    return 0;
}
#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        int r_ = (expr);                                                                                      \
        if (r_ != 0) return set_err(KNN_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));      \
    } while (0)

// how long a rank waits for its peers (communicator creation, a synchronous sharded search): KNN355_COMM_TIMEOUT_S, default 120 s
static double comm_timeout_s()
{
    const char *e = getenv("KNN355_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 120.0;
    return v > 0 ? v : 120.0;
}

struct knn_comm_s {
    void *comm = nullptr;
    int world = 1, rank = 0, device = 0;
    DevBuf keys, gathered;    // this rank's keys / everybody's: one search at a time uses them (see `done`)
    hipEvent_t done = nullptr; // recorded behind the last search's selection: a call on ANOTHER stream waits for it
    std::mutex mu;
};

extern "C" int knn_comm_unique_id(uint8_t *id128)
{
    if (!id128) return set_err(KNN_ERR_INVALID, "comm_unique_id: null pointer");
    int rc = rccl_load();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id128));
    return 0;
}

extern "C" int knn_comm_create(const uint8_t *id128, int32_t world, int32_t rank, int32_t device, knn_comm_s **out)
{
    if (!id128 || !out) return set_err(KNN_ERR_INVALID, "comm_create: null pointer");
    if (world < 1 || rank < 0 || rank >= world) return set_err(KNN_ERR_INVALID, "comm_create: bad world / rank");
    int rc = rccl_load();
    if (rc) return rc;
    rc = ensure_device(device);
    if (rc) return rc;
    knn_comm_s *c = new knn_comm_s();
    c->world = world;
    c->rank = rank;
    c->device = device;
    Id128 id;
    memcpy(id.b, id128, 128);
    // ncclCommInitRank returns when EVERY rank has called it -- with a peer that never arrives, never.  It runs on a helper
    // thread and this call waits for it at most KNN355_COMM_TIMEOUT_S seconds (120): the first contact with a multi-GPU node
    // fails with the rank named, not as a hang.  (A helper that is still inside RCCL when the wait runs out is left to itself:
    // it owns its state and destroys the communicator should the call ever return.)
    struct InitState {
        std::mutex mu;
        std::condition_variable cv;
        bool finished = false, abandoned = false;
        int r = 0;
        void *comm = nullptr;
    };
    auto st = std::make_shared<InitState>();
    std::thread([st, id, world, rank, device]() {
        int r = hipSetDevice(device) == hipSuccess ? 0 : -1; // (the communicator belongs to the calling thread's current device)
        void *comm = nullptr;
        if (r == 0) r = g_rccl.CommInitRank(&comm, world, id, rank);
        std::unique_lock<std::mutex> lk(st->mu);
        st->r = r;
        st->comm = comm;
        st->finished = true;
        if (st->abandoned && r == 0 && comm) (void)g_rccl.CommDestroy(comm);
        st->cv.notify_all();
    }).detach();
    const double wait_s = comm_timeout_s();
    {
        std::unique_lock<std::mutex> lk(st->mu);
        if (!st->cv.wait_for(lk, std::chrono::duration<double>(wait_s), [&] { return st->finished; })) {
            st->abandoned = true;
            delete c;
            char msg[256];
            snprintf(msg, sizeof msg, "comm_create: rank %d of %d: ncclCommInitRank did not return within %.0f s (KNN355_COMM_TIMEOUT_S) -- "
                                      "a peer never arrived, or the ranks do not share the unique id", rank, world, wait_s);
            return set_err(KNN_ERR_TIMEOUT, msg);
        }
    }
    if (st->r != 0) {
        delete c;
        return set_err(KNN_ERR_HIP, st->r == -1 ? std::string("comm_create: hipSetDevice failed on the helper thread")
                                               : std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(st->r));
    }
    c->comm = st->comm;
    *out = c;
    return 0;
}

extern "C" void knn_comm_free(knn_comm_s *c)
{
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
        (void)hipDeviceSynchronize();
        c->keys.release();
        c->gathered.release();
        if (c->done) (void)hipEventDestroy(c->done);
    }
    delete c;
}

// h holds THIS rank's rows (global id of local row r = id_base + r); q_dev [nq][d] is the same on every rank;
// D_dev / I_dev receive the global result on every rank.  Every rank must call with the same nq and k.
extern "C" int knn_sharded_search_dev(knn_handle h, knn_comm_s *c, const float *q_dev, int64_t nq, int64_t k, uint32_t id_base,
                                      float *D_dev, int64_t *I_dev, void *stream)
{
    if (!c) return set_err(KNN_ERR_INVALID, "sharded_search: null communicator");
    int rc = check_search_args(h, q_dev, nq, k, D_dev, I_dev);
    if (rc) return rc;
    if (nq == 0) return 0;
    if (h->device != c->device) return set_err(KNN_ERR_INVALID, "sharded_search: index and communicator live on different devices");
    std::lock_guard<std::mutex> lc(c->mu);
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    const size_t nkeys = (size_t)nq * k;
    // The key and gather buffers belong to the communicator: a search on another stream than the previous one must not
    // overwrite them while that one's all-gather or selection is still running -- it waits for the previous
    // search's `done` event (on the GPU; the host does not block).  RCCL wants the collectives of one communicator
    // issued in one order anyway.
    if (!c->done) HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    HIP_TRY(hipStreamWaitEvent(s, c->done, 0));
    if (c->keys.ensure(nkeys * 8) || c->gathered.ensure(nkeys * 8 * c->world))
        return set_err(KNN_ERR_HIP, "sharded_search: out of device memory (before the collective: the other ranks will wait for this one -- "
                                    "treat a failure on any rank as fatal for the communicator)");
    h->last_ms = -1.f;
    rc = search_dev_impl(h, q_dev, nq, (int)k, nullptr, nullptr, (uint64_t *)c->keys.p, id_base, false, s);
    std::string local_err;
    if (rc) {
        // A rank that fails locally still enters the collective -- with "no rows" for keys -- so that its peers do not
        // hang in ncclAllGather; it returns its error afterwards.  The peers' result then lacks this shard and they
        // cannot know: callers must treat a failure on ANY rank as a failure of the search (exchange the return codes).
        local_err = g_err;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((nkeys + 255) / 256)), dim3(256), 0, s, (int64_t)nkeys, h->metric,
                           (uint64_t *)c->keys.p, (float *)nullptr, (int64_t *)nullptr);
    }
    RCCL_TRY(g_rccl.AllGather(c->keys.p, c->gathered.p, nkeys, /* ncclUint64 */ 5, c->comm, s));
    if (rc) {
        (void)hipEventRecord(c->done, s);
        return set_err(rc, local_err);
    }
    SelectParams sp = {};
    sp.in = (const uint64_t *)c->gathered.p; sp.lm_lists = c->world; sp.lm_k = (int)k;
    sp.nq = nq; sp.k = (int)k; sp.metric = h->metric;
    sp.D = D_dev; sp.I = I_dev;
    rc = launch_select(sp, s);
    (void)hipEventRecord(c->done, s);
    if (rc) return rc;
    if (!stream) {
        if (c->world == 1) {
            HIP_TRY(hipStreamSynchronize(s));
        } else {
            // bounded: a peer that never enters the all-gather would hold hipStreamSynchronize for ever
            const double wait_s = comm_timeout_s();
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            for (;;) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return set_err(KNN_ERR_HIP, std::string("sharded_search: ") + hipGetErrorString(q));
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) {
                    char msg[256];
                    snprintf(msg, sizeof msg, "sharded_search: rank %d of %d: the search (local scan, ncclAllGather, selection) did not complete within "
                                              "%.0f s (KNN355_COMM_TIMEOUT_S) -- a peer never entered the all-gather; the communicator is unusable", c->rank, c->world, wait_s);
                    return set_err(KNN_ERR_TIMEOUT, msg);
                }
                if (++spins < 2000) std::this_thread::yield(); // (a healthy step takes a millisecond: stay close for the first ones)
                else std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        }
    }
    return 0;
}

#include "hnsw.inc"
#include "lsh.inc"
#include "eval.inc"
