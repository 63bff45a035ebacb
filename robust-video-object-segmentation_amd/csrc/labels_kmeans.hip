// Label preparation, segmented k-means (bit-identical to scipy kmeans2) and adaptive-proxy
// construction.  Reference: AEM:252-286 (adaptive_embedding_for_matching.py) + scipy.cluster.vq.
#include "aoc_common.h"
#include <stdlib.h>
#include <algorithm>
#include <string.h>

namespace {

constexpr int LP_BLOCK = 256;

// ------------------------------------------------------------------------------------------
// Label prep, pass 1: per-row bit masks + per-block counts of (kept) and (kept & right_o).
__global__ __launch_bounds__(LP_BLOCK) void label_flags_kernel(const float *__restrict__ labels, int n, int n_obj,
                                                                uint32_t *__restrict__ right_bits,
                                                                uint32_t *__restrict__ wrong_bits,
                                                                int32_t *__restrict__ block_counts, int n_blocks) {
    __shared__ int32_t wave_cnt[LP_BLOCK / 64][AOC_MAX_OBJECTS + 1];
    const int row = blockIdx.x * LP_BLOCK + threadIdx.x;
    uint32_t right = 0, wrong = 0;
    if (row < n) {
        const float *l = labels + (size_t)row * n_obj;
        float sum = 0.0f;
        for (int o = 0; o < n_obj; ++o) {
            float v = l[o];
            sum += v;                              // AEM:585 torch.sum(labels, dim=1)
            if (v > 0.9f) right |= 1u << o;        // AEM:252
            if (v < 0.1f) wrong |= 1u << o;        // AEM:197
        }
        if (sum > 0.9f) right |= AOC_ROW_KEPT_BIT; // AEM:585
        right_bits[row] = right;
        wrong_bits[row] = wrong;
    }
    const bool kept = (right & AOC_ROW_KEPT_BIT) != 0;
    const int wave = threadIdx.x >> 6;
    for (int c = 0; c <= n_obj; ++c) {
        bool f = kept && (c == n_obj || ((right >> c) & 1u));
        unsigned long long m = __ballot(f);
        if (aoc_lane() == 0) wave_cnt[wave][c] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x <= n_obj) {
        int s = 0;
        for (int w = 0; w < LP_BLOCK / 64; ++w) s += wave_cnt[w][threadIdx.x];
        block_counts[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s;
    }
}

// Bits only (local matching needs no index lists).
__global__ __launch_bounds__(LP_BLOCK) void label_bits_kernel(const float *__restrict__ labels, int64_t n, int n_obj,
                                                               uint32_t *__restrict__ right_bits, uint32_t *__restrict__ wrong_bits) {
    const int64_t row = (int64_t)blockIdx.x * LP_BLOCK + threadIdx.x;
    if (row >= n) return;
    const float *l = labels + (size_t)row * n_obj;
    uint32_t right = 0, wrong = 0;
    float sum = 0.0f;
    for (int o = 0; o < n_obj; ++o) {
        float v = l[o];
        sum += v;
        if (v > 0.9f) right |= 1u << o;
        if (v < 0.1f) wrong |= 1u << o;
    }
    if (sum > 0.9f) right |= AOC_ROW_KEPT_BIT;
    right_bits[row] = right;
    if (wrong_bits) wrong_bits[row] = wrong;
}

// Pass 2 (one block): exclusive scan of the per-block counts of every counter, totals, offsets.
__global__ __launch_bounds__(1024) void label_scan_kernel(int32_t *__restrict__ block_counts, int n_blocks, int n_obj,
                                                           int32_t *__restrict__ counts, int32_t *__restrict__ obj_offsets) {
    __shared__ int32_t wave_sum[16];
    __shared__ int32_t carry_s;
    __shared__ int32_t totals[AOC_MAX_OBJECTS + 1];
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    for (int c = 0; c <= n_obj; ++c) {
        int32_t *bc = block_counts + (size_t)c * n_blocks;
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
        for (int base = 0; base < n_blocks; base += 1024) {
            int i = base + threadIdx.x;
            int v = (i < n_blocks) ? bc[i] : 0;
            int incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) wave_sum[wave] = incl;
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += wave_sum[w];
            int carry = carry_s;
            if (i < n_blocks) bc[i] = carry + woff + incl - v;
            __syncthreads();
            if (threadIdx.x == 1023) carry_s = carry + woff + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) totals[c] = carry_s;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int off = 0;
        for (int c = 0; c < n_obj; ++c) {
            counts[c] = totals[c];
            obj_offsets[c] = off;
            off += totals[c];
        }
        counts[n_obj] = totals[n_obj];
        obj_offsets[n_obj] = off;
    }
}

// Pass 3: stable scatter of row ids into fg_rows and the per-object lists.
__global__ __launch_bounds__(LP_BLOCK) void label_scatter_kernel(const uint32_t *__restrict__ right_bits, int n, int n_obj,
                                                                  const int32_t *__restrict__ block_offsets, int n_blocks,
                                                                  const int32_t *__restrict__ obj_offsets,
                                                                  int32_t *__restrict__ fg_rows, int32_t *__restrict__ obj_rows) {
    __shared__ int32_t wave_cnt[LP_BLOCK / 64][AOC_MAX_OBJECTS + 1];
    const int row = blockIdx.x * LP_BLOCK + threadIdx.x;
    const uint32_t right = (row < n) ? right_bits[row] : 0u;
    const bool kept = (right & AOC_ROW_KEPT_BIT) != 0;
    const int wave = threadIdx.x >> 6, lane = aoc_lane();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int my_rank[AOC_MAX_OBJECTS + 1];
#pragma unroll 1
    for (int c = 0; c <= n_obj; ++c) {
        bool f = kept && (c == n_obj || ((right >> c) & 1u));
        unsigned long long m = __ballot(f);
        my_rank[c] = __popcll(m & lt);
        if (lane == 0) wave_cnt[wave][c] = __popcll(m);
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c <= n_obj; ++c) {
        bool f = kept && (c == n_obj || ((right >> c) & 1u));
        if (!f) continue;
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w][c];
        int pos = block_offsets[(size_t)c * n_blocks + blockIdx.x] + woff + my_rank[c];
        if (c == n_obj) fg_rows[pos] = row;
        else obj_rows[obj_offsets[c] + pos] = row;
    }
}

__global__ void kmeans_plan_kernel(const int32_t *__restrict__ counts, int n_seg, int cluster_num, int32_t *__restrict__ seg_k) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int k = cluster_num;
        for (int s = 0; s < n_seg; ++s) {       // AEM:268: cluster_num = min(cluster_num, n_i) -- sticky
            k = min(k, counts[s]);
            seg_k[s] = k;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k-means.  All arithmetic below is ordered exactly as scipy's _vq.pyx (see aoc_oracle.c).

// sequential |x|^2: multiply, then add (two roundings per term), t = 0..C-1
__device__ __forceinline__ float sqnorm_seq(const float *__restrict__ p, int C) {
    float s = 0.0f;
    for (int t = 0; t < C; ++t) {
        float prod = p[t] * p[t];
        s = s + prod;
    }
    return s;
}

// centroids[s,j,:] = pool[rows[seg_off[s] + init_rows[s,j]], :], plus their norms.
__global__ __launch_bounds__(64) void km_init_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                      const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                      const int32_t *__restrict__ init_rows, int kmax,
                                                      float *__restrict__ centroids, float *__restrict__ cnorm,
                                                      int32_t *__restrict__ cluster_counts) {
    const int s = blockIdx.y, j = blockIdx.x;
    const int k = seg_k[s];
    float *dst = centroids + ((size_t)s * kmax + j) * C;
    if (j >= k) {
        for (int t = threadIdx.x; t < C; t += 64) dst[t] = 0.0f;
        if (threadIdx.x == 0) { cnorm[s * kmax + j] = INFINITY; cluster_counts[s * kmax + j] = 0; }
        return;
    }
    const int len = seg_off[s + 1] - seg_off[s];
    int local = init_rows[s * kmax + j];
    local = min(max(local, 0), len - 1);
    const float *src = pool + (size_t)rows[seg_off[s] + local] * C;
    for (int t = threadIdx.x; t < C; t += 64) dst[t] = src[t];
    if (threadIdx.x == 0) { cnorm[s * kmax + j] = sqnorm_seq(src, C); cluster_counts[s * kmax + j] = 0; }
}

// Assignment step (scipy _vq.vq).  One thread per row, the row held in registers (C4MAX float4),
// the segment's code book in LDS (broadcast reads).  dist = (-2*dot + |x|^2) + |c|^2, strict <.
template <int C4MAX>
__global__ __launch_bounds__(256) void km_assign_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                         const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                         const float *__restrict__ centroids, const float *__restrict__ cnorm,
                                                         int kmax, int32_t *__restrict__ labels, float *__restrict__ rownorm,
                                                         int first_iter) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const int k = seg_k[s];
    if (k <= 0) return;
    const int beg = seg_off[s];
    const int len = seg_off[s + 1] - beg;
    if ((int)(blockIdx.x * blockDim.x) >= len) return;
    const int c4 = C >> 2;
    float *lc = lds;                 // [k][C]
    float *lcn = lds + (size_t)k * C;  // [k]
    const float *csrc = centroids + (size_t)s * kmax * C;
    for (int i = threadIdx.x; i < k * C; i += blockDim.x) lc[i] = csrc[i];
    for (int i = threadIdx.x; i < k; i += blockDim.x) lcn[i] = cnorm[s * kmax + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= len) return;
    const float4 *xr = reinterpret_cast<const float4 *>(pool + (size_t)rows[beg + p] * C);
    float4 x[C4MAX];
#pragma unroll
    for (int t = 0; t < C4MAX; ++t) x[t] = (t < c4) ? xr[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    float xs;
    if (first_iter) {
        xs = 0.0f;
#pragma unroll
        for (int t = 0; t < C4MAX; ++t) {
            if (t < c4) {
                float p0 = x[t].x * x[t].x; xs = xs + p0;
                float p1 = x[t].y * x[t].y; xs = xs + p1;
                float p2 = x[t].z * x[t].z; xs = xs + p2;
                float p3 = x[t].w * x[t].w; xs = xs + p3;
            }
        }
        rownorm[beg + p] = xs;
    } else {
        xs = rownorm[beg + p];
    }
    float low = INFINITY;
    int best = 0;
    for (int j = 0; j < k; ++j) {
        const float4 *cj = reinterpret_cast<const float4 *>(lc + (size_t)j * C);
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < C4MAX; ++t) {
            if (t < c4) {
                float4 c = cj[t];
                acc = __builtin_fmaf(x[t].x, c.x, acc);
                acc = __builtin_fmaf(x[t].y, c.y, acc);
                acc = __builtin_fmaf(x[t].z, c.z, acc);
                acc = __builtin_fmaf(x[t].w, c.w, acc);
            }
        }
        float m = -2.0f * acc;
        float dist = (m + xs) + lcn[j];
        if (dist < low) { low = dist; best = j; }
    }
    labels[beg + p] = best;
}

// Generic-width variant (C not a multiple of 4, or C > 128): the row is re-read per centroid.
__global__ __launch_bounds__(256) void km_assign_generic_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                                 const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                                 const float *__restrict__ centroids, const float *__restrict__ cnorm,
                                                                 int kmax, int32_t *__restrict__ labels, float *__restrict__ rownorm,
                                                                 int first_iter) {
    const int s = blockIdx.y;
    const int k = seg_k[s];
    if (k <= 0) return;
    const int beg = seg_off[s];
    const int len = seg_off[s + 1] - beg;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= len) return;
    const float *x = pool + (size_t)rows[beg + p] * C;
    float xs;
    if (first_iter) { xs = sqnorm_seq(x, C); rownorm[beg + p] = xs; } else { xs = rownorm[beg + p]; }
    float low = INFINITY;
    int best = 0;
    for (int j = 0; j < k; ++j) {
        const float *c = centroids + ((size_t)s * kmax + j) * C;
        float acc = 0.0f;
        for (int t = 0; t < C; ++t) acc = __builtin_fmaf(x[t], c[t], acc);
        float m = -2.0f * acc;
        float dist = (m + xs) + cnorm[s * kmax + j];
        if (dist < low) { low = dist; best = j; }
    }
    labels[beg + p] = best;
}

// Ordered per-cluster accumulation.  One wave per (cluster j, segment s); lanes own features
// t = lane + 64 f.  The wave scans the segment's labels 64 at a time, queues the member rows in an
// LDS ring (stable order) and adds them strictly in row order, G rows of loads in flight at a time.
// MODE 0: update step of k-means (scipy _vq.update_cluster_means + vq.py:820-823).
// MODE 1: proxy construction, AEM:280-282 (rows come from the global kept-row list at
//         segment-LOCAL indices; see aoc_build_proxies).
constexpr int KU_G = 16;
constexpr int KU_RING = 256;

template <int NF, int MODE>
__global__ __launch_bounds__(64) void km_accumulate_kernel(const float *__restrict__ pool, int C,
                                                            const int32_t *__restrict__ rows,   // MODE 0: packed obj rows; MODE 1: fg_rows
                                                            const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                            const int32_t *__restrict__ labels, int kmax,
                                                            float *__restrict__ centroids, float *__restrict__ cnorm,
                                                            int32_t *__restrict__ cluster_counts,
                                                            float *__restrict__ proxies, float *__restrict__ proxy_sqnorm) {
    __shared__ int32_t ring[KU_RING];
    __shared__ float stage[NF * 64];
    const int s = blockIdx.y, j = blockIdx.x;
    const int k = seg_k[s];
    const int lane = threadIdx.x;
    if (j >= k) {
        if (MODE == 1) {
            float *p0 = proxies + (((size_t)s * 2 + 0) * kmax + j) * C;
            float *p1 = proxies + (((size_t)s * 2 + 1) * kmax + j) * C;
            for (int t = lane; t < C; t += 64) { p0[t] = 0.0f; p1[t] = 0.0f; }
            if (lane == 0) {
                proxy_sqnorm[((size_t)s * 2 + 0) * kmax + j] = INFINITY;
                proxy_sqnorm[((size_t)s * 2 + 1) * kmax + j] = INFINITY;
            }
        }
        return;
    }
    const int beg = seg_off[s];
    const int len = seg_off[s + 1] - beg;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    float acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = 0.0f;
    int qhead = 0, qtail = 0, cnt = 0;

    auto drain = [&](int nmem) {   // nmem <= KU_G, wave-uniform
        int r[KU_G];
#pragma unroll
        for (int u = 0; u < KU_G; ++u) {
            int idx = (qhead + (u < nmem ? u : 0)) & (KU_RING - 1);
            r[u] = __builtin_amdgcn_readfirstlane(ring[idx]);
        }
        float v[KU_G][NF];
#pragma unroll
        for (int u = 0; u < KU_G; ++u) {
            const float *base = pool + (size_t)r[u] * C;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                int t = lane + 64 * f;
                v[u][f] = (t < C) ? base[t] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < KU_G; ++u) {
            const bool on = u < nmem;   // padded slots add +0.0f, which is exact (acc is never -0)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = acc[f] + (on ? v[u][f] : 0.0f);
        }
        qhead += nmem;
    };

    for (int base = 0; base < len; base += 64) {
        const int p = base + lane;
        const bool valid = p < len;
        const int lab = valid ? labels[beg + p] : -1;
        int row;
        if (MODE == 0) row = valid ? rows[beg + p] : 0;
        else row = valid ? rows[p] : 0;            // AEM:280: global kept-row array at LOCAL index p
        const bool mine = lab == j;
        const unsigned long long m = __ballot(mine);
        const int nm = __popcll(m);
        if (nm == 0) continue;
        if (mine) ring[(qtail + __popcll(m & lt)) & (KU_RING - 1)] = row;
        qtail += nm;
        cnt += nm;
        while (qtail - qhead >= KU_G) drain(KU_G);
    }
    while (qtail - qhead > 0) drain(min(KU_G, qtail - qhead));

    if (MODE == 0) {
        if (lane == 0) cluster_counts[s * kmax + j] = cnt;
        if (cnt == 0) return;                      // vq.py:820-823: keep the previous centroid (and norm)
        const float fc = (float)cnt;
        float *dst = centroids + ((size_t)s * kmax + j) * C;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            int t = lane + 64 * f;
            float q = acc[f] / fc;
            if (t < C) { dst[t] = q; stage[t] = q; }
        }
        __syncthreads();
        if (lane == 0) cnorm[s * kmax + j] = sqnorm_seq(stage, C);
    } else {
        // proxy set 0 = centroid (copied), set 1 = centroid_avg; norms in any order (torch .pow(2).sum(1))
        const float *csrc = centroids + ((size_t)s * kmax + j) * C;
        float *p0 = proxies + (((size_t)s * 2 + 0) * kmax + j) * C;
        float *p1 = proxies + (((size_t)s * 2 + 1) * kmax + j) * C;
        float n0 = 0.0f, n1 = 0.0f;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            int t = lane + 64 * f;
            if (t < C) {
                float c = csrc[t];
                p0[t] = c;
                n0 += c * c;
                float a = (cnt > 0) ? acc[f] / (float)cnt : 0.0f;
                p1[t] = a;
                n1 += a * a;
            }
        }
        n0 = aoc_wave_sum(n0);
        n1 = aoc_wave_sum(n1);
        if (lane == 0) {
            proxy_sqnorm[((size_t)s * 2 + 0) * kmax + j] = n0;
            proxy_sqnorm[((size_t)s * 2 + 1) * kmax + j] = (cnt > 0) ? n1 : INFINITY;   // np.unique drops empty clusters
        }
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t KU_INVALID_OFF = 0xFFFFFF00u;   // beyond the buffer descriptor's range: raw buffer loads return 0

// ==========================================================================================
// Fast exact k-means update ("scan-sum" pipeline).
//
// scipy's update step adds the member rows of a cluster one after another in float32.  For the
// non-negative embeddings of this model (ReLU outputs and their bilinear blends) that sequential sum has
// structure: while the running sum s stays inside one binade [2^E, 2^(E+1)), every addition
// fl(s + x) = u * RNE(n + x/u)  (u = ulp(s) = 2^(E-23), n = s/u an integer in [2^23, 2^24)) adds the
// INTEGER rne(x/u) -- exact integer arithmetic, hence associative -- except for exact ties
// (frac(x/u) == 1/2), which round to even and therefore only need the running PARITY of n, and after a tie
// the parity is even whatever it was.  So 64 members are folded per step with lanes = members:
//   r_l = floor(y_l) + [frac(y_l) > 1/2],  y_l = x_l / u      (exact: power-of-two scaling)
//   ties fixed up from ballots (rare), n += sum_l r_l         (exact), accepted iff n stays < 2^24.
// Anything else -- binade crossing, s == 0 / tiny, negative or non-finite x -- falls back to the literal
// serial float additions for that 64-member block, so the result is bit-identical to the sequential sum
// in every case (tests compare with scipy bit for bit).
//
// Per Lloyd iteration: km_assign_rank (labels + stable rank of every row inside its 256-row block per
// cluster + block histogram) -> km_blockscan (block offsets, cluster sizes and bases) -> km_scatter
// (ordered member lists, as byte offsets) -> km_sum_scan (one wave per (cluster, 4 features)).
#ifdef AOC_KS_STATS
__device__ unsigned long long aoc_ks_stats[8];   // attempts, folded feature-attempts, redo feature-attempts, careful ok, serial blocks, ties
#define KS_STAT(i, n) do { if (threadIdx.x == 0) atomicAdd(&aoc_ks_stats[i], (unsigned long long)(n)); } while (0)
#else
#define KS_STAT(i, n) do { } while (0)
#endif
#ifdef AOC_KS_STATS
#define KS_CLK() ((long long)__builtin_readcyclecounter())
#else
#define KS_CLK() 0ll
#endif
constexpr int KS_T = 8;          // blocks (of 64 members) folded between two cross-lane reductions

// wave-wide integer sum, uniform result: 4 DPP row_shr adds (lane 15 of each 16-lane row ends up with the
// row total) + 4 readlanes + scalar adds -- no LDS crossbar round trips on the serial chain.
__device__ __forceinline__ int ks_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1, out-of-row lanes read 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) +
           __builtin_amdgcn_readlane(v, 63);
}

// One block of 64 members of one feature, folded in the integer domain of the current binade.
//   y = x / u (exact), r = RNE(y) as an integer; a tie (|RNE(y) - y| == 1/2) is re-rounded from the running
//   parity; `badmask` collects lanes whose value is not a plain non-negative number below 2^24 ulps
//   (negative, NaN/inf, x >> s): the caller then redoes the block with literal float additions.
__device__ __forceinline__ void ks_fold_block(float x, float inv_u, int lane, int &accr, int &par, unsigned long long &badmask) {
    const float y = x * inv_u;
    const float rn = rintf(y);                       // v_rndne_f32
    int r = (int)rn;
    badmask |= __ballot(!(__float_as_uint(y) < 0x4B800000u));   // y in [+0, 2^24) <=> bits(y) < bits(2^24)
    const unsigned long long ties = __ballot(fabsf(rn - y) == 0.5f);
    if (ties) {                                       // rare: resolve round-half-even from the running parity
        const int fl = (int)floorf(y);
        if ((ties >> lane) & 1ull) r = fl;            // start from floor(y); the bump below re-rounds
        const unsigned long long odd = __ballot((r & 1) != 0);
        int base_par = par, from = 0;
        unsigned long long tm = ties;
        while (tm) {
            const int t = __builtin_ctzll(tm);
            tm &= tm - 1;
            const unsigned long long below_t = (t == 0) ? 0ull : (~0ull >> (64 - t));
            const unsigned long long below_from = (from == 0) ? 0ull : (~0ull >> (64 - from));
            const int pb = base_par ^ (__popcll(odd & below_t & ~below_from) & 1);   // parity of n before lane t
            const int bump = pb ^ (int)((odd >> t) & 1ull);                           // n + floor(y) odd -> round up
            if (lane == t) r += bump;
            base_par = 0;                                                              // a tie always leaves n even
            from = t + 1;
        }
        const unsigned long long rest = (from >= 64) ? 0ull : (~0ull << from);
        par = base_par ^ (__popcll(odd & rest) & 1);
    } else {
        par ^= __popcll(__ballot((r & 1) != 0)) & 1;
    }
    accr += r;
}

struct KsBinade {
    float u, inv_u;
    int n_in;
    bool ok;
};
__device__ __forceinline__ KsBinade ks_binade(float s) {
    KsBinade b;
    const uint32_t bits = __float_as_uint(s);
    const int e = (int)((bits >> 23) & 0xff) - 127;
    b.ok = (s > 0.0f) && e >= -100 && e <= 100;
    const int ee = b.ok ? e : 0;
    b.u = __uint_as_float((uint32_t)(ee - 23 + 127) << 23);
    b.inv_u = __uint_as_float((uint32_t)(23 - ee + 127) << 23);
    b.n_in = (int)(s * b.inv_u);
    return b;
}

// literal serial additions of one 64-member block, members [from, 64) (x of absent members is +0, which is exact)
__device__ __forceinline__ float ks_serial_block(float s, float x, int from = 0) {
    if (__ballot(x != 0.0f) == 0ull) return s;   // all zeros: s + 0.0f == s (s is never -0)
    for (int kk = from; kk < 64; ++kk) s = s + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), kk));
    return s;
}

// wave-wide inclusive prefix sum (DPP: Hillis-Steele inside each 16-lane row, then row_bcast:15 / :31)
__device__ __forceinline__ int ks_wave_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// One 64-member block, exactly, from running sum s.  Members are folded in the integer domain of the current
// binade; if n would leave the binade, a prefix scan locates the member whose addition crosses, that one
// addition is done in floating point, and the rest of the block is folded in the new binade.  Anything
// unusual (s == 0 or tiny, negative / non-finite values, more than a few crossings) takes the serial additions.
__device__ __forceinline__ float ks_block_exact(float s, float x, int lane) {
    int from = 0;
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const KsBinade bb = ks_binade(s);
        if (!bb.ok) break;
        int r = 0, pr = bb.n_in & 1;
        unsigned long long bd = 0ull;
        ks_fold_block((lane >= from) ? x : 0.0f, bb.inv_u, lane, r, pr, bd);     // r = this lane's integer (ties resolved)
        if (bd != 0ull) break;
        const int pre = ks_wave_scan(r);
        const int total = __builtin_amdgcn_readlane(pre, 63);
        if ((long long)bb.n_in + total <= 0xFFFFFF) return (float)(bb.n_in + total) * bb.u;
        // first member whose addition takes n to 2^24 or beyond
        const unsigned long long over = __ballot((long long)bb.n_in + pre > 0xFFFFFF);
        const int mstar = __builtin_ctzll(over);
        const int before = (mstar == 0) ? 0 : __builtin_amdgcn_readlane(pre, mstar - 1);
        const float s_before = (float)(bb.n_in + before) * bb.u;                  // exact state in front of member m*
        s = s_before + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), mstar));   // the crossing addition itself
        from = mstar + 1;
        if (from >= 64) return s;
    }
    return ks_serial_block(s, x, from);
}

template <int C4MAX>
__global__ __launch_bounds__(256) void km_assign_rank_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                              const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                              const float *__restrict__ centroids, int kmax, int32_t *__restrict__ labels,
                                                              uint16_t *__restrict__ rank16, int32_t *__restrict__ hist, int nb_max,
                                                              float *__restrict__ rownorm, int first_iter) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const int k = seg_k[s];
    if (k <= 0) return;
    const int beg = seg_off[s];
    const int len = seg_off[s + 1] - beg;
    if ((int)(blockIdx.x * 256) >= len) return;
    const int c4 = C >> 2;
    float *lc = lds;                                    // [k][C]
    float *lcn = lds + (size_t)k * C;                   // [k]
    int32_t *wcnt = reinterpret_cast<int32_t *>(lcn + kmax);   // [4][kmax]
    const float *csrc = centroids + (size_t)s * kmax * C;
    for (int i = threadIdx.x; i < k * C; i += 256) lc[i] = csrc[i];
    __syncthreads();
    if ((int)threadIdx.x < k) lcn[threadIdx.x] = sqnorm_seq(lc + (size_t)threadIdx.x * C, C);   // scipy code_sqr, sequential
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < len;
    int best = -1;
    if (valid) {
        const float4 *xr = reinterpret_cast<const float4 *>(pool + (size_t)rows[beg + p] * C);
        float4 x[C4MAX];
#pragma unroll
        for (int t = 0; t < C4MAX; ++t) x[t] = (t < c4) ? xr[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        float xs;
        if (first_iter) {
            xs = 0.0f;
#pragma unroll
            for (int t = 0; t < C4MAX; ++t) {
                if (t < c4) {
                    float p0 = x[t].x * x[t].x; xs = xs + p0;
                    float p1 = x[t].y * x[t].y; xs = xs + p1;
                    float p2 = x[t].z * x[t].z; xs = xs + p2;
                    float p3 = x[t].w * x[t].w; xs = xs + p3;
                }
            }
            rownorm[beg + p] = xs;
        } else {
            xs = rownorm[beg + p];
        }
        float low = INFINITY;
        best = 0;
        for (int j = 0; j < k; ++j) {
            const float4 *cj = reinterpret_cast<const float4 *>(lc + (size_t)j * C);
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < C4MAX; ++t) {
                if (t < c4) {
                    float4 c = cj[t];
                    acc = __builtin_fmaf(x[t].x, c.x, acc);
                    acc = __builtin_fmaf(x[t].y, c.y, acc);
                    acc = __builtin_fmaf(x[t].z, c.z, acc);
                    acc = __builtin_fmaf(x[t].w, c.w, acc);
                }
            }
            float m = -2.0f * acc;
            float dist = (m + xs) + lcn[j];
            if (dist < low) { low = dist; best = j; }
        }
        labels[beg + p] = best;
    }
    // stable rank of the row among the rows of its block that share its label
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank = 0;
    for (int kk = 0; kk < k; ++kk) {
        const unsigned long long m = __ballot(best == kk);
        if (best == kk) rank = __popcll(m & lt);
        if (lane == 0) wcnt[wave * kmax + kk] = __popcll(m);
    }
    __syncthreads();
    if (valid) {
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wcnt[w * kmax + best];
        rank16[beg + p] = (uint16_t)(woff + rank);
    }
    if ((int)threadIdx.x < k)
        hist[((size_t)s * nb_max + blockIdx.x) * kmax + threadIdx.x] =
            wcnt[threadIdx.x] + wcnt[kmax + threadIdx.x] + wcnt[2 * kmax + threadIdx.x] + wcnt[3 * kmax + threadIdx.x];
}

// |x|^2 of every listed row, scipy's order (sequential multiply-then-add over the channels; the library is compiled with -ffp-contract=off),
// for the rows [0, seg_off[n_seg_limit]) of the packed lists: with it the matrix-pipe assignment below also serves the FIRST Lloyd iteration
// (km_assign_rank_kernel computed the norms on the fly there, one pass over the rows per replica).
__global__ __launch_bounds__(256) void km_rownorm_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ seg_off, int n_seg_limit, float *__restrict__ rownorm) {
    const int total = seg_off[n_seg_limit];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float4 *xr = reinterpret_cast<const float4 *>(pool + (size_t)rows[i] * C);
    const int c4 = C >> 2;
    float xs = 0.0f;
    for (int t0 = 0; t0 < c4; t0 += 5) {                     // five 16-byte loads in flight, then their 20 additions in channel order
        float4 v[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) v[u] = xr[min(t0 + u, c4 - 1)];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (t0 + u < c4) {
                float p0 = v[u].x * v[u].x; xs = xs + p0;
                float p1 = v[u].y * v[u].y; xs = xs + p1;
                float p2 = v[u].z * v[u].z; xs = xs + p2;
                float p3 = v[u].w * v[u].w; xs = xs + p3;
            }
        }
    }
    rownorm[i] = xs;
}

// Assignment on the fp32 matrix pipe (iterations 2..20; the first one also produces the row norms and uses the
// kernel above).  v_mfma_f32_16x16x4_f32 accumulates each output as one k-ordered fmaf chain -- the OpenBLAS order
// scipy's vq sees -- so labels stay bit-identical while the rows are fetched with coalesced 16-byte loads instead of
// one row per lane.  Block = 4 waves x 64 rows (the 256-row blocks the rank/histogram logic is built on).  Clusters
// are the A operand (code book, 25 VGPR per 16 clusters, resident), rows the B operand: each wave stages its 16-row
// tiles in a private, double-buffered, k-permuted LDS image (lane (j, kq) reads x[j][4t + kq] with ds_read_b128).
// D: lane holds row j = lane & 15 and clusters (lane >> 4) * 4 + r: the argmin is 4 in-lane compares and two
// cross-group exchanges, ties to the lowest index like scipy's strict <.
#ifndef AOC_KA_TPF4
#define AOC_KA_TPF4 1
#endif
template <int TMAX, int KT>
__global__ __launch_bounds__(256, KT == 1 ? 3 : 2) void km_assign_mfma_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                              const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k, int n_seg,
                                                              const float *__restrict__ centroids, int kmax, int32_t *__restrict__ labels,
                                                              uint16_t *__restrict__ rank16, int32_t *__restrict__ hist, int nb_max,
                                                              const float *__restrict__ rownorm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TP = (TMAX + 3) / 4 * 4;             // floats per kq-stream (padded to float4)
    constexpr int RS = 4 * TP + 4;                     // row stride of the k-permuted image
    constexpr int NB4 = TP / 4;
    constexpr int PIECES = (16 * TMAX + 63) / 64;      // float4 pieces per lane per 16-row tile
    constexpr int c4 = TMAX;                           // launched for C == 4 TMAX only: piece indices divide by a constant (a runtime divisor costs ~20 VALU per division, ~1000 per 64 rows)
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;

    float *cimg = lds;                                           // [KT*16][RS] code book, k-permuted
    float *lcn = cimg + (size_t)KT * 16 * RS;                    // [KT*16] |c|^2 (+inf beyond k)
    float *wimg = lcn + KT * 16 + (size_t)wave * 16 * RS;        // this wave's row-tile image
    int32_t *wcnt = reinterpret_cast<int32_t *>(lcn + KT * 16 + (size_t)4 * 16 * RS);       // [4][kmax]

    // zero the stream padding of this wave's images and of the code book (never overwritten afterwards)
    if (TP > c4) {
        for (int idx = lane; idx < 16 * 4 * (TP - c4); idx += 64) {
            const int rr = idx / (4 * (TP - c4)), rem = idx - rr * 4 * (TP - c4);
            wimg[(size_t)rr * RS + (rem / (TP - c4)) * TP + c4 + rem % (TP - c4)] = 0.0f;
        }
        for (int idx = threadIdx.x; idx < KT * 16 * 4 * (TP - c4); idx += 256) {
            const int rr = idx / (4 * (TP - c4)), rem = idx - rr * 4 * (TP - c4);
            cimg[(size_t)rr * RS + (rem / (TP - c4)) * TP + c4 + rem % (TP - c4)] = 0.0f;
        }
    }

    // segment table in LDS (work item -> segment is a search over it; from global memory every probe is a dependent round trip)
    constexpr int SEG_LDS = 128;
    __shared__ int32_t lseg_off[SEG_LDS + 1], lseg_nb[SEG_LDS];
    const bool seg_in_lds = n_seg <= SEG_LDS;
    if (seg_in_lds) {
        for (int i = threadIdx.x; i <= n_seg; i += 256) lseg_off[i] = seg_off[i];
        __syncthreads();
        for (int i = threadIdx.x; i < n_seg; i += 256) lseg_nb[i] = (seg_k[i] > 0) ? (lseg_off[i + 1] - lseg_off[i] + 255) / 256 : 0;
        __syncthreads();
    }

    float4 ca[KT][NB4];
    float cn[KT][4];
    int cur_seg = -1, k = 0, beg = 0, len = 0;
    // persistent blocks walk the list of (segment, 256-row block) work items: no empty workgroups, and the code book is
    // staged once per block and segment instead of once per 256 rows
    for (int w = blockIdx.x;; w += gridDim.x) {
        int s = 0, bx = w;
        if (seg_in_lds) {
            for (; s < n_seg; ++s) {
                const int nbs = lseg_nb[s];
                if (bx < nbs) break;
                bx -= nbs;
            }
        } else {
            for (; s < n_seg; ++s) {
                const int ls = seg_off[s + 1] - seg_off[s];
                const int nbs = (seg_k[s] > 0) ? (ls + 255) / 256 : 0;
                if (bx < nbs) break;
                bx -= nbs;
            }
        }
        if (s >= n_seg) break;
        // ---- this item's rows: all loads of TPF tiles are issued before anything waits (ids -> pieces in registers)
        constexpr int TPF = KT >= 4 ? AOC_KA_TPF4 : 2;       // tiles in flight per wave (K > 48: one -- the second set of staging registers spilled 36 VGPRs)
        const int ibeg = seg_in_lds ? lseg_off[s] : seg_off[s], ilen = (seg_in_lds ? lseg_off[s + 1] : seg_off[s + 1]) - ibeg;
        const int wave_row0 = bx * 256 + wave * 64;
        float4 pv[TPF][PIECES];
        // the pool row of every row of this wave: ONE id load per lane (rows past the end take the segment's last row and are zeroed on
        // the way into the image), handed to the lanes that fetch the pieces by a cross-lane read -- so the id round trip happens once per
        // item and all piece loads of a tile are in flight together, with no exec-masked regions in between
        const int my_p = min(wave_row0 + lane, ilen - 1);
        const int my_id = rows[ibeg + max(my_p, 0)];
        const float my_xs = rownorm[ibeg + max(my_p, 0)];      // |x|^2 of row (wave_row0 + lane), fetched with the ids instead of once per tile
        auto issue_tile = [&](int tile, float4 (&v)[PIECES]) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = min(i * 64 + lane, 16 * c4 - 1);
                const int rr = idx / c4, t = idx - rr * c4;
                const int id = __shfl(my_id, tile * 16 + rr);
                v[i] = reinterpret_cast<const float4 *>(pool + (size_t)id * C)[t];
            }
        };
        auto write_tile = [&](int tile, const float4 (&v)[PIECES]) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = i * 64 + lane;
                if (idx < 16 * c4) {
                    const int rr = idx / c4, t = idx - rr * c4;
                    const bool in = wave_row0 + tile * 16 + rr < ilen;
                    float *d = wimg + (size_t)rr * RS + t;
                    d[0] = in ? v[i].x : 0.f; d[TP] = in ? v[i].y : 0.f; d[2 * TP] = in ? v[i].z : 0.f; d[3 * TP] = in ? v[i].w : 0.f;
                }
            }
        };
#pragma unroll
        for (int t = 0; t < TPF; ++t) issue_tile(t, pv[t]);
        if (s != cur_seg) {
            cur_seg = s;
            k = seg_k[s];
            beg = seg_off[s];
            len = seg_off[s + 1] - beg;
            __syncthreads();                                   // previous users of cimg / lcn are done
            const float *csrc = centroids + (size_t)s * kmax * C;
            for (int idx = threadIdx.x; idx < KT * 16 * c4; idx += 256) {
                const int cc = idx / c4, t = idx - cc * c4;
                const float4 v = (cc < k) ? reinterpret_cast<const float4 *>(csrc + (size_t)cc * C)[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                float *d = cimg + (size_t)cc * RS + t;
                d[0] = v.x; d[TP] = v.y; d[2 * TP] = v.z; d[3 * TP] = v.w;
            }
            __syncthreads();
            // |c|^2 from the staged image (scipy code_sqr order: k = 0..C-1, multiply then add)
            if ((int)threadIdx.x < KT * 16) {
                float nrm = INFINITY;
                if ((int)threadIdx.x < k) {
                    const float *im = cimg + (size_t)threadIdx.x * RS;
                    nrm = 0.0f;
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
                        if (t < c4) {
#pragma unroll
                            for (int kq = 0; kq < 4; ++kq) {
                                const float v = im[kq * TP + t];
                                const float prod = v * v;
                                nrm = nrm + prod;
                            }
                        }
                    }
                }
                lcn[threadIdx.x] = nrm;
            }
            __syncthreads();
            // A operands: lane (i = j, kq = g) holds c[16 kt + i][4t + kq]
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const float *st = cimg + (size_t)(kt * 16 + j) * RS + g * TP;
#pragma unroll
                for (int u = 0; u < NB4; ++u) ca[kt][u] = *reinterpret_cast<const float4 *>(st + 4 * u);
#pragma unroll
                for (int r = 0; r < 4; ++r) cn[kt][r] = lcn[kt * 16 + g * 4 + r];
            }
        }

        int best = -1;                                     // label of row (wave_row0 + lane) once all four tiles are done
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            write_tile(tile, pv[tile % TPF]);
            if (tile + TPF < 4) issue_tile(tile + TPF, pv[tile % TPF]);
            // B operand: lane (j, kq = g) reads its stream of row j
            const float *bs = wimg + (size_t)j * RS + g * TP;
            float4 xb[NB4];
#pragma unroll
            for (int u = 0; u < NB4; ++u) xb[u] = *reinterpret_cast<const float4 *>(bs + 4 * u);
            const int prow = wave_row0 + tile * 16 + j;
            const float xs_l = __shfl(my_xs, tile * 16 + j);
            const float xs = (prow < len) ? xs_l : 0.0f;
            float low = INFINITY;
            int arg = 0;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NB4; ++u) {
                    const float aa[4] = {ca[kt][u].x, ca[kt][u].y, ca[kt][u].z, ca[kt][u].w};
                    const float bb[4] = {xb[u].x, xb[u].y, xb[u].z, xb[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * u + e < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[e], bb[e], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float mm = -2.0f * acc[r];
                    const float dist = (mm + xs) + cn[kt][r];
                    if (dist < low) { low = dist; arg = kt * 16 + g * 4 + r; }
                }
            }
            // the four lane groups hold disjoint cluster ranges of the same row: lowest distance, then lowest index
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float d2 = __shfl_xor(low, off);
                const int a2 = __shfl_xor(arg, off);
                if (d2 < low || (d2 == low && a2 < arg)) { low = d2; arg = a2; }
            }
            const int mine = __shfl(arg, lane & 15);       // every lane: label of row (lane & 15) of this tile
            if ((lane >> 4) == tile) best = mine;
        }
        const int p = wave_row0 + lane;
        const bool valid = p < len;
        if (!valid) best = -1;
        if (valid) labels[beg + p] = best;
        // stable rank of the row among the rows of its block that share its label
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        int rank = 0;
        for (int kk = 0; kk < k; ++kk) {
            const unsigned long long mk = __ballot(best == kk);
            if (best == kk) rank = __popcll(mk & lt);
            if (lane == 0) wcnt[wave * kmax + kk] = __popcll(mk);
        }
        __syncthreads();
        if (valid) {
            int woff = 0;
            for (int ww = 0; ww < wave; ++ww) woff += wcnt[ww * kmax + best];
            rank16[beg + p] = (uint16_t)(woff + rank);
        }
        if ((int)threadIdx.x < k)
            hist[((size_t)s * nb_max + bx) * kmax + threadIdx.x] =
                wcnt[threadIdx.x] + wcnt[kmax + threadIdx.x] + wcnt[2 * kmax + threadIdx.x] + wcnt[3 * kmax + threadIdx.x];
        __syncthreads();                                       // wcnt is reused by the next work item
    }
}

// The same assignment for REPLICATED segment lists (aoc_kmeans_replicate[_levels]: the k-means of several frames and / or cluster levels
// that see the same pool advance as n_rep replicas of n_base segments with identical row lists, segment f * n_base + s0 = replica f of
// base segment s0).  A work item = (group of up to n_grp replicas, base segment, 256-row block): the rows are fetched and staged ONCE
// and multiplied against the code books of all replicas of the group, which sit side by side in LDS (the A operands are read from there
// per (tile, replica) instead of living in registers).  Per replica the arithmetic -- and therefore every label, rank and histogram --
// is exactly that of km_assign_mfma_kernel; what changes is the traffic: with F frames x L levels in a chain the single-replica kernel
// streams the pool rows F * L times per Lloyd iteration.
template <int TMAX, int KT>
__global__ __launch_bounds__(256, KT == 1 ? 3 : 2) void km_assign_mfma_rep_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ rows,
                                                                  const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k, int n_base,
                                                                  int n_rep, int n_grp, const float *__restrict__ centroids, int kmax,
                                                                  int32_t *__restrict__ labels, uint16_t *__restrict__ rank16,
                                                                  int32_t *__restrict__ hist, int nb_max, const float *__restrict__ rownorm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TP = (TMAX + 3) / 4 * 4;
    constexpr int RS = 4 * TP + 4;
    constexpr int NB4 = TP / 4;
    constexpr int PIECES = (16 * TMAX + 63) / 64;
    constexpr int c4 = TMAX;
    constexpr int CB = KT * 16 * RS;                             // floats of one replica's code book image
    constexpr int MAXG = 16;                                     // replicas per group (host: n_grp <= MAXG)
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;

    float *cimg = lds;                                           // [n_grp][KT*16][RS]
    float *lcn = cimg + (size_t)n_grp * CB;                      // [n_grp][KT*16]
    float *wimg = lcn + (size_t)n_grp * KT * 16 + (size_t)wave * 16 * RS;
    int32_t *wcnt = reinterpret_cast<int32_t *>(lcn + (size_t)n_grp * KT * 16 + (size_t)4 * 16 * RS);       // [4][kmax]
    uint8_t *lab = reinterpret_cast<uint8_t *>(wcnt + 4 * kmax);                                             // [n_grp][256]
    __shared__ int32_t lk[MAXG], lbeg[MAXG];

    if (TP > c4) {
        for (int idx = lane; idx < 16 * 4 * (TP - c4); idx += 64) {
            const int rr = idx / (4 * (TP - c4)), rem = idx - rr * 4 * (TP - c4);
            wimg[(size_t)rr * RS + (rem / (TP - c4)) * TP + c4 + rem % (TP - c4)] = 0.0f;
        }
        for (int idx = threadIdx.x; idx < n_grp * KT * 16 * 4 * (TP - c4); idx += 256) {
            const int rr = idx / (4 * (TP - c4)), rem = idx - rr * 4 * (TP - c4);
            cimg[(size_t)rr * RS + (rem / (TP - c4)) * TP + c4 + rem % (TP - c4)] = 0.0f;
        }
    }

    constexpr int SEG_LDS = 32;        // (static LDS small enough for three workgroups of three code books per CU, KT = 1)
    __shared__ int32_t lseg_off[SEG_LDS + 1], lseg_nb[SEG_LDS];
    const bool seg_in_lds = n_base <= SEG_LDS;
    // a base segment is walked if ANY replica clusters it (the cluster counts may differ per replica; a replica with K = 0 is skipped below)
    auto seg_live = [&](int i) -> bool {
        bool any = false;
        for (int f = 0; f < n_rep; ++f) any |= seg_k[f * n_base + i] > 0;
        return any;
    };
    if (seg_in_lds) {
        for (int i = threadIdx.x; i <= n_base; i += 256) lseg_off[i] = seg_off[i];
        __syncthreads();
        for (int i = threadIdx.x; i < n_base; i += 256) lseg_nb[i] = seg_live(i) ? (lseg_off[i + 1] - lseg_off[i] + 255) / 256 : 0;
        __syncthreads();
    }
    int items_per_group = 0;
    for (int s = 0; s < n_base; ++s)
        items_per_group += seg_in_lds ? lseg_nb[s] : (seg_live(s) ? (seg_off[s + 1] - seg_off[s] + 255) / 256 : 0);
    const int n_groups = (n_rep + n_grp - 1) / n_grp;

    int cur_seg = -1, cur_grp = -1, ng = 0, len = 0;
    for (int w = blockIdx.x; items_per_group > 0; w += gridDim.x) {
        const int gi = w / items_per_group;
        if (gi >= n_groups) break;
        int s = 0, bx = w - gi * items_per_group;
        for (; s < n_base; ++s) {
            const int nbs = seg_in_lds ? lseg_nb[s] : (seg_live(s) ? (seg_off[s + 1] - seg_off[s] + 255) / 256 : 0);
            if (bx < nbs) break;
            bx -= nbs;
        }
        if (s >= n_base) break;
        // ---- this item's rows (replica 0's lists: every replica lists the same rows in the same order)
        constexpr int TPF = 2;
        const int ibeg = seg_in_lds ? lseg_off[s] : seg_off[s], ilen = (seg_in_lds ? lseg_off[s + 1] : seg_off[s + 1]) - ibeg;
        const int wave_row0 = bx * 256 + wave * 64;
        float4 pv[TPF][PIECES];
        const int my_p = min(wave_row0 + lane, ilen - 1);
        const int my_id = rows[ibeg + max(my_p, 0)];
        const float my_xs = rownorm[ibeg + max(my_p, 0)];
        auto issue_tile = [&](int tile, float4 (&v)[PIECES]) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = min(i * 64 + lane, 16 * c4 - 1);
                const int rr = idx / c4, t = idx - rr * c4;
                const int id = __shfl(my_id, tile * 16 + rr);
                v[i] = reinterpret_cast<const float4 *>(pool + (size_t)id * C)[t];
            }
        };
        auto write_tile = [&](int tile, const float4 (&v)[PIECES]) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = i * 64 + lane;
                if (idx < 16 * c4) {
                    const int rr = idx / c4, t = idx - rr * c4;
                    const bool in = wave_row0 + tile * 16 + rr < ilen;
                    float *d = wimg + (size_t)rr * RS + t;
                    d[0] = in ? v[i].x : 0.f; d[TP] = in ? v[i].y : 0.f; d[2 * TP] = in ? v[i].z : 0.f; d[3 * TP] = in ? v[i].w : 0.f;
                }
            }
        };
#pragma unroll
        for (int t = 0; t < TPF; ++t) issue_tile(t, pv[t]);
        if (s != cur_seg || gi != cur_grp) {
            cur_seg = s;
            cur_grp = gi;
            ng = min(n_grp, n_rep - gi * n_grp);
            len = ilen;
            __syncthreads();                                   // previous users of cimg / lcn / lk are done
            for (int r = 0; r < ng; ++r) {
                const int sr = (gi * n_grp + r) * n_base + s;
                const int kr = seg_k[sr];
                const float *csrc = centroids + (size_t)sr * kmax * C;
                float *ci = cimg + (size_t)r * CB;
                for (int idx = threadIdx.x; idx < KT * 16 * c4; idx += 256) {
                    const int cc = idx / c4, t = idx - cc * c4;
                    const float4 v = (cc < kr) ? reinterpret_cast<const float4 *>(csrc + (size_t)cc * C)[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float *d = ci + (size_t)cc * RS + t;
                    d[0] = v.x; d[TP] = v.y; d[2 * TP] = v.z; d[3 * TP] = v.w;
                }
                if (threadIdx.x == 0) { lk[r] = kr; lbeg[r] = seg_off[sr]; }
            }
            __syncthreads();
            // |c|^2 from the staged images (scipy code_sqr order: k = 0..C-1, multiply then add)
            for (int idx = threadIdx.x; idx < ng * KT * 16; idx += 256) {
                const int r = idx / (KT * 16), cc = idx - r * (KT * 16);
                float nrm = INFINITY;
                if (cc < lk[r]) {
                    const float *im = cimg + (size_t)r * CB + (size_t)cc * RS;
                    nrm = 0.0f;
#pragma unroll
                    for (int t = 0; t < TMAX; ++t) {
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) {
                            const float v = im[kq * TP + t];
                            const float prod = v * v;
                            nrm = nrm + prod;
                        }
                    }
                }
                lcn[idx] = nrm;
            }
            __syncthreads();
        }

#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            write_tile(tile, pv[tile % TPF]);
            if (tile + TPF < 4) issue_tile(tile + TPF, pv[tile % TPF]);
            const float *bs = wimg + (size_t)j * RS + g * TP;
            float4 xb[NB4];
#pragma unroll
            for (int u = 0; u < NB4; ++u) xb[u] = *reinterpret_cast<const float4 *>(bs + 4 * u);
            const int prow = wave_row0 + tile * 16 + j;
            const float xs_l = __shfl(my_xs, tile * 16 + j);
            const float xs = (prow < len) ? xs_l : 0.0f;
#pragma unroll 1
            for (int r = 0; r < ng; ++r) {
                float low = INFINITY;
                int arg = 0;
                // cluster tiles this replica really has (levels 8 / 16 next to 32 in one chain: kmax = 32 pads every code book to two tiles);
                // the padding rows carry the norm +inf and can never be the minimum, so leaving their MFMAs out changes nothing
                const int ktr = (lk[r] + 15) >> 4;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    if (kt >= ktr) continue;
                    // A operands of replica r: lane (i = j, kq = g) holds c[16 kt + i][4t + kq]
                    const float *st = cimg + (size_t)r * CB + (size_t)(kt * 16 + j) * RS + g * TP;
                    float4 ca[NB4];
#pragma unroll
                    for (int u = 0; u < NB4; ++u) ca[u] = *reinterpret_cast<const float4 *>(st + 4 * u);
                    const float4 cn4 = *reinterpret_cast<const float4 *>(lcn + (size_t)r * KT * 16 + kt * 16 + g * 4);
                    const float cn[4] = {cn4.x, cn4.y, cn4.z, cn4.w};
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < NB4; ++u) {
                        const float aa[4] = {ca[u].x, ca[u].y, ca[u].z, ca[u].w};
                        const float bb[4] = {xb[u].x, xb[u].y, xb[u].z, xb[u].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * u + e < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[e], bb[e], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float mm = -2.0f * acc[q];
                        const float dist = (mm + xs) + cn[q];
                        if (dist < low) { low = dist; arg = kt * 16 + g * 4 + q; }
                    }
                }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float d2 = __shfl_xor(low, off);
                    const int a2 = __shfl_xor(arg, off);
                    if (d2 < low || (d2 == low && a2 < arg)) { low = d2; arg = a2; }
                }
                const int mine = __shfl(arg, lane & 15);       // every lane: label of row (lane & 15) of this tile
                if ((lane >> 4) == tile) lab[r * 256 + threadIdx.x] = (uint8_t)mine;      // read back by the same thread below
            }
        }
        const int p = wave_row0 + lane;
        const bool valid = p < len;
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        for (int r = 0; r < ng; ++r) {
            const int kr = lk[r], beg = lbeg[r];
            if (kr <= 0) continue;                             // this replica does not cluster the segment (wave- and block-uniform)
            const int sr = (gi * n_grp + r) * n_base + s;
            const int best = valid ? (int)lab[r * 256 + threadIdx.x] : -1;
            if (valid) labels[beg + p] = best;
            int rank = 0;
            for (int kk = 0; kk < kr; ++kk) {
                const unsigned long long mk = __ballot(best == kk);
                if (best == kk) rank = __popcll(mk & lt);
                if (lane == 0) wcnt[wave * kmax + kk] = __popcll(mk);
            }
            __syncthreads();
            if (valid) {
                int woff = 0;
                for (int ww = 0; ww < wave; ++ww) woff += wcnt[ww * kmax + best];
                rank16[beg + p] = (uint16_t)(woff + rank);
            }
            if ((int)threadIdx.x < kr)
                hist[((size_t)sr * nb_max + bx) * kmax + threadIdx.x] =
                    wcnt[threadIdx.x] + wcnt[kmax + threadIdx.x] + wcnt[2 * kmax + threadIdx.x] + wcnt[3 * kmax + threadIdx.x];
            __syncthreads();                                   // wcnt is reused by the next replica / work item
        }
    }
}

// Segment lists replicated n_rep times (k-means of several frames that see the same pool, advanced together in the same
// launches): rows_out[f * total + i] = rows[i], seg_off_out[f * n_seg + s] = f * total + seg_off[s], total = seg_off[n_seg].
__global__ __launch_bounds__(256) void km_replicate_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ seg_off,
                                                            const int32_t *__restrict__ seg_k, int n_seg, int n_rep, int64_t capacity,
                                                            int32_t *__restrict__ rows_out, int32_t *__restrict__ seg_off_out,
                                                            int32_t *__restrict__ seg_k_out) {
    const int64_t total = seg_off[n_seg];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (i < total && i < capacity) rows_out[f * total + i] = rows[i];
    if (i <= n_seg && (i < n_seg || f == n_rep - 1)) seg_off_out[f * n_seg + i] = (int32_t)(f * total + seg_off[i]);
    if (i < n_seg) seg_k_out[f * n_seg + i] = seg_k[i];
}

// The same with a per-replica cluster count: replica f clusters at level klev.k[f % klev.n] (multi-level proxies, K in {8, 16, 32};
// several frames x levels in one chain), with the sticky rule of AEM:268 applied per replica from the segment sizes.
struct KmLevels {
    int32_t k[8];
    int32_t n;
};
__global__ __launch_bounds__(256) void km_replicate_levels_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ seg_off,
                                                                   int n_seg, int n_rep, KmLevels klev, int64_t capacity,
                                                                   int32_t *__restrict__ rows_out, int32_t *__restrict__ seg_off_out,
                                                                   int32_t *__restrict__ seg_k_out) {
    const int64_t total = seg_off[n_seg];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (i < total && i < capacity && rows_out != rows) rows_out[f * total + i] = rows[i];
    if (i <= n_seg && (i < n_seg || f == n_rep - 1)) seg_off_out[f * n_seg + i] = (int32_t)(f * total + seg_off[i]);
    if (i < n_seg) {
        int k = klev.k[f % klev.n];
        for (int s = 0; s <= (int)i; ++s) k = min(k, seg_off[s + 1] - seg_off[s]);     // AEM:268, sticky
        seg_k_out[f * n_seg + i] = k;
    }
}

// rank + histogram from EXISTING labels (proxy construction after the last iteration)
__global__ __launch_bounds__(256) void km_rank_only_kernel(const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                            const int32_t *__restrict__ labels, int kmax, uint16_t *__restrict__ rank16,
                                                            int32_t *__restrict__ hist, int nb_max) {
    __shared__ int32_t wcnt[4 * AOC_MAX_CLUSTERS];
    const int s = blockIdx.y;
    const int k = seg_k[s];
    if (k <= 0) return;
    const int beg = seg_off[s];
    const int len = seg_off[s + 1] - beg;
    if ((int)(blockIdx.x * 256) >= len) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < len;
    const int best = valid ? labels[beg + p] : -1;
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank = 0;
    for (int kk = 0; kk < k; ++kk) {
        const unsigned long long m = __ballot(best == kk);
        if (best == kk) rank = __popcll(m & lt);
        if (lane == 0) wcnt[wave * kmax + kk] = __popcll(m);
    }
    __syncthreads();
    if (valid) {
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wcnt[w * kmax + best];
        rank16[beg + p] = (uint16_t)(woff + rank);
    }
    if ((int)threadIdx.x < k)
        hist[((size_t)s * nb_max + blockIdx.x) * kmax + threadIdx.x] =
            wcnt[threadIdx.x] + wcnt[kmax + threadIdx.x] + wcnt[2 * kmax + threadIdx.x] + wcnt[3 * kmax + threadIdx.x];
}

// Chunk = KS_CHUNK consecutive members of one cluster's ordered list.  Segment s owns the chunk ids
// [ks_seg_chunk_base(s), +ks_seg_chunk_region(s)): enough for sum_j ceil(cnt_j / KS_CHUNK) whatever the split.
constexpr int KS_CHUNK = 64 * KS_T;
__host__ __device__ __forceinline__ int ks_seg_chunk_base(int seg_beg, int s, int kmax) { return seg_beg / KS_CHUNK + s * (kmax + 1); }
__host__ __device__ __forceinline__ int ks_seg_chunk_region(int len, int kmax) { return len / KS_CHUNK + kmax + 1; }

// Block offsets, cluster sizes, chunk table AND the ordered member lists in ONE launch (round 3: km_blockscan_kernel on one workgroup per
// segment, then km_scatter_kernel -- two latency floors per Lloyd iteration).  A workgroup owns KSS_BLOCKS consecutive 256-row blocks of
// one segment.  The per-block histograms of the WHOLE segment are final when this kernel starts (the assignment kernel wrote them), so every
// workgroup adds up, per cluster, the blocks in front of its own range (its exclusive offsets) and all blocks (cluster sizes -> cluster bases)
// itself: nb x kmax integers per workgroup from L2 (38 KB at 600 blocks x 16 clusters; the quadratic total stays below 2 % of one pass over the
// pool rows because a workgroup covers 2048 rows), lanes = clusters x block strides, no inter-workgroup dependency.  Workgroup 0 of a
// segment also publishes what the sum kernels read: cluster sizes, cluster bases, and the segment's chunk table (chunk base per cluster; owner
// cluster / local index per chunk id, -1 = unused).  Member position = segment begin + cluster base + offset of the row's block + stable rank
// inside the block -- the same formula as before, so the lists are identical.
//   moff[pos] = byte offset of the member's pool row.  rows_local != nullptr: proxy construction, AEM:280 -- the GLOBAL kept-row array at
//   the LOCAL index p.
constexpr int KSS_BLOCKS = 8;
__global__ __launch_bounds__(256) void km_scan_scatter_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ rows_local,
                                                               const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k, int n_seg,
                                                               const int32_t *__restrict__ labels, const uint16_t *__restrict__ rank16,
                                                               const int32_t *__restrict__ hist, int nb_max, int kmax, uint32_t row_bytes,
                                                               uint32_t *__restrict__ moff, int32_t *__restrict__ counts, int32_t *__restrict__ cbase,
                                                               int32_t *__restrict__ cchunk, int32_t *__restrict__ owner_cluster,
                                                               int32_t *__restrict__ owner_local, int nch_cap, uint32_t *__restrict__ cflag) {
    __shared__ int32_t part[2][256];                               // [before my range | whole segment][stride group x cluster]
    __shared__ int32_t lpre[AOC_MAX_CLUSTERS], ltot[AOC_MAX_CLUSTERS], lbase[AOC_MAX_CLUSTERS];
    __shared__ int32_t loff[KSS_BLOCKS][AOC_MAX_CLUSTERS];
    __shared__ int32_t lchunk[AOC_MAX_CLUSTERS + 1];
    const int s = blockIdx.y;
    const int k = seg_k[s];
    const int beg = seg_off[s];
    const int len = seg_off[s + 1] - beg;
    const int b0 = blockIdx.x * KSS_BLOCKS;
    const bool first = blockIdx.x == 0;
    if (!first && (k <= 0 || b0 * 256 >= len)) return;
    const int nb = (k > 0) ? (len + 255) / 256 : 0;
    // ---- per cluster: blocks in front of this workgroup's range, and all blocks.  kp = clusters rounded up to a power of two (<= 64)
    int kp = 1;
    while (kp < kmax) kp <<= 1;
    const int groups = 256 / kp;
    {
        const int kk = threadIdx.x & (kp - 1), g = threadIdx.x / kp;
        int pre = 0, tot = 0;
        if (kk < k) {
            const int32_t *hp = hist + (size_t)s * nb_max * kmax + kk;
            // eight independent loads in flight per round (clamped index, select afterwards: no exec-masked regions)
            for (int bb = g; bb < nb; bb += 8 * groups) {
                int v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = hp[(size_t)min(bb + u * groups, nb - 1) * kmax];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int b = bb + u * groups;
                    const int w = (b < nb) ? v[u] : 0;
                    tot += w;
                    pre += (b < b0) ? w : 0;
                }
            }
        }
        part[0][threadIdx.x] = pre;
        part[1][threadIdx.x] = tot;
    }
    __syncthreads();
    if ((int)threadIdx.x < kmax) {
        int pre = 0, tot = 0;
        for (int g = 0; g < groups; ++g) {
            pre += part[0][g * kp + threadIdx.x];
            tot += part[1][g * kp + threadIdx.x];
        }
        lpre[threadIdx.x] = pre;
        ltot[threadIdx.x] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int cb = ks_seg_chunk_base(beg, s, kmax);
        int off = 0, ch = 0;
        for (int kk = 0; kk < kmax; ++kk) {
            lbase[kk] = off;
            lchunk[kk] = ch;
            if (first) {
                counts[s * kmax + kk] = ltot[kk];
                cbase[s * kmax + kk] = off;
                if (cchunk) cchunk[s * kmax + kk] = cb + ch;
            }
            off += ltot[kk];
            ch += (ltot[kk] + KS_CHUNK - 1) / KS_CHUNK;
        }
        lchunk[kmax] = ch;
    }
    // offsets of this workgroup's blocks: a serial walk over at most KSS_BLOCKS histograms per cluster
    if ((int)threadIdx.x < k) {
        int hv[KSS_BLOCKS];
#pragma unroll
        for (int i = 0; i < KSS_BLOCKS; ++i) hv[i] = hist[((size_t)s * nb_max + min(b0 + i, nb - 1)) * kmax + threadIdx.x];
        int run = lpre[threadIdx.x];
#pragma unroll
        for (int i = 0; i < KSS_BLOCKS; ++i) {
            loff[i][threadIdx.x] = run;
            run += (b0 + i < nb) ? hv[i] : 0;
        }
    }
    __syncthreads();
    if (first && owner_cluster) {
        // this segment initialises every chunk id up to the next segment's base (the last one up to the capacity)
        const int cb = ks_seg_chunk_base(beg, s, kmax);
        const int region = ((s + 1 < n_seg) ? ks_seg_chunk_base(seg_off[s + 1], s + 1, kmax) : nch_cap) - cb;
        for (int i = threadIdx.x; i < region; i += 256) {
            int oc = -1, ol = 0;
            if (i < lchunk[kmax]) {
                int kk = 0;
                while (kk + 1 < kmax && lchunk[kk + 1] <= i) ++kk;
                oc = s * kmax + kk;
                ol = i - lchunk[kk];
            }
            owner_cluster[cb + i] = oc;
            owner_local[cb + i] = ol;
            if (cflag) {
#pragma unroll
                for (int g = 0; g < 8; ++g) cflag[(size_t)(cb + i) * 8 + g] = 0u;      // nothing published yet in this pass
            }
        }
    }
    if (k <= 0) return;
    // ---- the member lists of this workgroup's rows: all loads of the KSS_BLOCKS rounds are independent
    int lab[KSS_BLOCKS], rk[KSS_BLOCKS], row[KSS_BLOCKS];
    const int32_t *rsrc = rows_local ? rows_local : rows + beg;
#pragma unroll
    for (int i = 0; i < KSS_BLOCKS; ++i) {
        const int p = min((b0 + i) * 256 + (int)threadIdx.x, len - 1);
        lab[i] = labels[beg + p];
        rk[i] = (int)rank16[beg + p];
        row[i] = rsrc[p];
    }
#pragma unroll
    for (int i = 0; i < KSS_BLOCKS; ++i) {
        const int p = (b0 + i) * 256 + threadIdx.x;
        if (p < len) moff[beg + lbase[lab[i]] + loff[i][lab[i]] + rk[i]] = (uint32_t)row[i] * row_bytes;
    }
}

// ------------------------------------------------------------------------------------------
// Chunk-parallel part of the exact update.  Within a binade the fold is integer addition, so a chunk's
// contribution is a pair of integers (one per parity of the incoming n, they differ only if the chunk
// contains ties) that can be computed by ANY workgroup once the binade is known.  The binade is PREDICTED
// from an any-order prefix of chunk sums; the serial stitch (km_sum_scan_kernel) verifies every prediction
// against the exact running sum (same exponent, no overflow of n) and otherwise recomputes the chunk from the
// rows, so a wrong prediction costs time, never exactness.
constexpr int KC_TILE_LD = AOC_MAX_CHANNELS / 2 + 1;   // 129: row stride of the 64 x C LDS tile (C <= 128), conflict-free columns

// Staging of 64-member blocks (full rows, coalesced 16-byte pieces) with memory-level parallelism: the chunk's
// offsets sit in LDS, every thread issues all of its row loads for block b+1 before block b is consumed, and
// writes them to the tile afterwards.  KC_STAGE_MAX float4 per thread cover 64 rows x C <= 128 floats.
constexpr int KC_STAGE_MAX = (64 * (AOC_MAX_CHANNELS / 2 / 4) + 255) / 256;   // 8
struct KcStage {
    float4 v[KC_STAGE_MAX];
};
__device__ __forceinline__ void kc_issue_block(KcStage &st, const float *__restrict__ pool, const uint32_t *__restrict__ loffs, int blk,
                                               int members_in_chunk, int c4) {
#pragma unroll
    for (int it = 0; it < KC_STAGE_MAX; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int mloc = idx / c4, piece = idx - mloc * c4;
        const int m = blk * 64 + mloc;
        st.v[it] = (idx < 64 * c4 && m < members_in_chunk)
                       ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pool) + loffs[m] + piece * 16)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void kc_write_block(const KcStage &st, float *__restrict__ tile, int c4) {
#pragma unroll
    for (int it = 0; it < KC_STAGE_MAX; ++it) {
        const int idx = it * 256 + threadIdx.x;
        if (idx < 64 * c4) {
            const int mloc = idx / c4, piece = idx - mloc * c4;
            float *d = tile + mloc * KC_TILE_LD + piece * 4;
            d[0] = st.v[it].x; d[1] = st.v[it].y; d[2] = st.v[it].z; d[3] = st.v[it].w;
        }
    }
}

// P0: csum[chunk, f] = any-order float sum of the chunk's members (only used to predict binades).  Thread t owns the float4 piece
// t % c4 of the rows t / c4, t / c4 + RPP, ... of the chunk (RPP = 256 / c4 rows per pass; consecutive threads read one row's consecutive
// 16 bytes): private float4 partial sums, eight independent loads in flight per thread, no LDS tile and no barrier in the loop; the RPP
// partial sums of a piece are combined through LDS at the end.
__device__ __forceinline__ void kc_chunk_sum_body(int chunk, uint32_t *__restrict__ loffs, float4 *__restrict__ part,
                                                  const float *__restrict__ pool, int C, const int32_t *__restrict__ seg_off,
                                                  const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                  const uint32_t *__restrict__ moff, int kmax,
                                                  const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                  float *__restrict__ csum, int start_chunk) {
    const int oc = owner_cluster[chunk];
    if (oc < 0) return;
    if (owner_local[chunk] < start_chunk) return;       // head chunks are summed literally (km_ordered_sum_kernel)
    const int s = oc / kmax;
    const int cnt = counts[oc];
    const uint32_t *list = moff + seg_off[s] + cbase[oc];
    const int first = owner_local[chunk] * KS_CHUNK;
    const int members = min(KS_CHUNK, cnt - first);
    for (int i = threadIdx.x; i < KS_CHUNK; i += 256) loffs[i] = list[first + min(i, members - 1)];     // 256 threads run this body
    __syncthreads();
    const int c4 = C >> 2;                              // float4 pieces per row (C % 4 == 0 on this path)
    const int rpp = 256 / c4;                           // rows per pass
    const int r0 = threadIdx.x / c4, piece = threadIdx.x - r0 * c4;
    const bool worker = r0 < rpp;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (worker) {
        const char *base = reinterpret_cast<const char *>(pool) + piece * 16;
        for (int m0 = r0; m0 < members; m0 += 8 * rpp) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(base + loffs[min(m0 + u * rpp, KS_CHUNK - 1)]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (m0 + u * rpp < members) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < c4) {
        float4 t = part[threadIdx.x];
        for (int r = 1; r < rpp; ++r) {
            const float4 x = part[r * c4 + threadIdx.x];
            t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
        }
        *reinterpret_cast<float4 *>(csum + (size_t)chunk * C + 4 * threadIdx.x) = t;
    }
}
__global__ __launch_bounds__(256) void km_chunk_sum_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ seg_off,
                                                            const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                            const uint32_t *__restrict__ moff, int kmax,
                                                            const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                            float *__restrict__ csum, int start_chunk) {
    __shared__ uint32_t loffs[KS_CHUNK];
    __shared__ __attribute__((aligned(16))) float4 part[256];
    kc_chunk_sum_body(blockIdx.x, loffs, part, pool, C, seg_off, counts, cbase, moff, kmax, owner_cluster, owner_local, csum, start_chunk);
}

// P1: per cluster, thread = feature: running any-order prefix over the chunks -> predicted binade of each chunk
// (KC_UNSAFE when the chunk may touch a binade boundary, starts at 0, or is not plainly positive)
constexpr int KC_UNSAFE = -128;
// binade of a chunk whose any-order running sum goes from `pre` to `end`, or KC_UNSAFE
__device__ __forceinline__ int kc_predict_binade(float pre, float end) {
    int e = KC_UNSAFE;
    if (pre > 1e-30f && end < 1e30f && end >= pre) {
        const int e_lo = (int)((__float_as_uint(pre * 0.9995f) >> 23) & 0xff) - 127;
        const int e_hi = (int)((__float_as_uint(end * 1.0005f) >> 23) & 0xff) - 127;
        if (e_lo == e_hi && e_lo >= -100 && e_lo <= 100) e = e_lo;
    }
    return e;
}
// Round 5: the binade of a tail chunk PREDICTED FROM THE PREVIOUS LLOYD ITERATION.  The stitch of iteration i walks every cluster's tail
// with the exact running sum; it leaves, per (cluster, feature), the exponent in front of the tail (e0) and the local indices of the chunks
// inside which the exponent changed, with the exponent behind them (at most KP_CROSS: a running sum of non-negative values doubles once per
// doubling of the member count).  Between two iterations few rows change cluster, so chunk c of iteration i + 1 almost always sits in the
// binade chunk c of iteration i sat in: the fold of iteration i + 1 takes that binade WITHOUT the any-order chunk sums (one pass over the
// tail rows and one launch less per iteration).  As before the stitch verifies every summary against the exact running sum (same exponent,
// n stays below 2^24) and folds a chunk from its rows when the check fails: a wrong prediction costs time, never exactness.  Clusters
// without a tail get e0 from their literal head sum (a tail that appears in the next iteration starts near it).
constexpr int KP_CROSS = 4;
struct KsPred {
    int8_t *e0;          // [n_seg * kmax * C]
    uint16_t *cidx;      // [n_seg * kmax * C * KP_CROSS]  local chunk index of the j-th crossing, 0xFFFF = none
    int8_t *ce;          // [n_seg * kmax * C * KP_CROSS]  exponent behind it (KC_UNSAFE: more crossings than slots -- nothing predicted from there on)
};
__device__ __forceinline__ int kp_predict(const KsPred &p, size_t idx, int c) {
    int e = p.e0[idx];
#pragma unroll
    for (int i = 0; i < KP_CROSS; ++i) {
        const int ci = p.cidx[idx * KP_CROSS + i];
        if (e == KC_UNSAFE || ci == 0xFFFF || c < ci) break;
        e = (c == ci) ? KC_UNSAFE : (int)p.ce[idx * KP_CROSS + i];      // the chunk that crossed last time is left to the stitch
    }
    return e;
}
__global__ __launch_bounds__(128) void km_chunk_predict_kernel(const int32_t *__restrict__ seg_k, const int32_t *__restrict__ counts,
                                                                const int32_t *__restrict__ cchunk, const float *__restrict__ csum, int kmax,
                                                                int C, int8_t *__restrict__ cexp, int start_chunk,
                                                                const float *__restrict__ head_state) {
    const int s = blockIdx.y, j = blockIdx.x;
    if (j >= seg_k[s]) return;
    const int oc = s * kmax + j;
    const int nch = (counts[oc] + KS_CHUNK - 1) / KS_CHUNK;
    if (nch <= start_chunk) return;
    const int base = cchunk[oc];
    for (int f = threadIdx.x; f < C; f += blockDim.x) {
        float pre = (start_chunk > 0) ? head_state[(size_t)oc * C + f] : 0.0f;     // exact sum of the head chunks
        for (int c0 = start_chunk; c0 < nch; c0 += 8) {
          float cs[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) cs[u] = (c0 + u < nch) ? csum[(size_t)(base + c0 + u) * C + f] : 0.0f;   // 8 loads in flight
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            if (c >= nch) break;
            const float end = pre + cs[u];
            cexp[(size_t)(base + c) * C + f] = (int8_t)kc_predict_binade(pre, end);
            pre = end;
          }
        }
    }
}

// Fold with both incoming parities tracked.  The two variants differ only at ties: every lane adds
// floor-or-rne(y) to ONE accumulator and each tie contributes a 0/1 bump that depends on the running parity,
// so the variants are two scalar bump counters plus two scalar parities.
struct KcFold {
    int acc;                 // per lane
    int par0, par1;          // uniform: parity of n after the members folded so far, for incoming parity 0 / 1
    int bump0, bump1;        // uniform: number of ties rounded up so far
    int bad;                 // uniform
};
__device__ __forceinline__ void kc_fold_block2(float x, float inv_u, KcFold &k) {
    const float y = x * inv_u;
    const float rn = rintf(y);
    int r = (int)rn;
    k.bad |= (__ballot(!(__float_as_uint(y) < 0x4B800000u)) != 0ull) ? 1 : 0;
    const unsigned long long ties = __ballot(fabsf(rn - y) == 0.5f);
    if (ties) {
        const int lane = aoc_lane();
        if ((ties >> lane) & 1ull) r = (int)floorf(y);
        const unsigned long long odd = __ballot((r & 1) != 0);
        int pp[2] = {k.par0, k.par1}, bb[2] = {0, 0};
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            int base_par = pp[v], from = 0;
            unsigned long long tm = ties;
            while (tm) {
                const int t = __builtin_ctzll(tm);
                tm &= tm - 1;
                const unsigned long long below_t = (t == 0) ? 0ull : (~0ull >> (64 - t));
                const unsigned long long below_from = (from == 0) ? 0ull : (~0ull >> (64 - from));
                const int pb = base_par ^ (__popcll(odd & below_t & ~below_from) & 1);
                bb[v] += pb ^ (int)((odd >> t) & 1ull);          // n + floor(y) odd -> this tie rounds up
                base_par = 0;                                      // a tie always leaves n even
                from = t + 1;
            }
            const unsigned long long rest = (from >= 64) ? 0ull : (~0ull << from);
            pp[v] = base_par ^ (__popcll(odd & rest) & 1);
        }
        k.par0 = pp[0]; k.par1 = pp[1];
        k.bump0 += bb[0]; k.bump1 += bb[1];
    } else {
        const int p = __popcll(__ballot((r & 1) != 0)) & 1;
        k.par0 ^= p; k.par1 ^= p;
    }
    k.acc += r;
}

// P2: cinc0/cinc1[chunk, f] = integer increment of n over the chunk for incoming parity 0 / 1, in the predicted
// binade.  grid = (chunk ids, feature groups of KC_FG); block = 4 waves x KC_FW features; only the group's
// columns of each row are staged (KC_FG * 4 bytes per row).
constexpr int KC_FW = 5;                 // features per wave
constexpr int KC_FG = 4 * KC_FW;         // features per workgroup (20 = 5 float4)
constexpr int KC_G_LD = KC_FG + 1;       // LDS row stride of the group tile
constexpr int KC_FOLD_LDS_FLOATS = 2 * 64 * KC_G_LD + KS_CHUNK + KC_FG;      // tile[2][64 x KC_G_LD] | loffs[KS_CHUNK] | lexp[KC_FG]
// b = workgroup index inside the fold role's 1-D grid; lds = KC_FOLD_LDS_FLOATS floats; 256 threads run it.
// pred != nullptr: binades from the previous Lloyd iteration (KsPred) instead of csum / km_chunk_predict_kernel.
__device__ __forceinline__ void kc_chunk_fold_body(int b, float *__restrict__ lds, const float *__restrict__ pool, int C, const int32_t *__restrict__ seg_off,
                                                   const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                   const uint32_t *__restrict__ moff, int kmax,
                                                   const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                   int8_t *__restrict__ cexp, int32_t *__restrict__ cinc0, int32_t *__restrict__ cinc1,
                                                   int start_chunk, const float *__restrict__ csum, const int32_t *__restrict__ cchunk,
                                                   const float *__restrict__ head_state, int n_chunks_grid, int n_groups, int xcd_aware,
                                                   const KsPred *__restrict__ pred) {
    float (*tile)[64 * KC_G_LD] = reinterpret_cast<float (*)[64 * KC_G_LD]>(lds);
    uint32_t *loffs = reinterpret_cast<uint32_t *>(lds + 2 * 64 * KC_G_LD);
    int *lexp = reinterpret_cast<int *>(lds + 2 * 64 * KC_G_LD + KS_CHUNK);
    // 1-D grid, XCD-aware: the feature groups of ONE chunk read neighbouring 80-byte pieces of the same member rows; with ids that differ
    // by 8 they run on the same XCD at about the same time and share the 64-byte sectors in its L2 (workgroups are dealt round-robin to
    // the XCDs).  Blocks of 8 chunks x n_groups; the last block may hold fewer chunks.
    int chunk, grp;
    {
        const int blk = b / (8 * n_groups), rem = b - blk * (8 * n_groups);
        const int pc = min(8, n_chunks_grid - blk * 8);
        grp = rem / pc;
        chunk = blk * 8 + rem - grp * pc;
        if (!xcd_aware) { chunk = b % n_chunks_grid; grp = b / n_chunks_grid; }
    }
    const int oc = owner_cluster[chunk];
    if (oc < 0) return;
    if (owner_local[chunk] < start_chunk) return;
    const int f0 = grp * KC_FG;
    const int s = oc / kmax;
    const int cnt = counts[oc];
    const uint32_t *list = moff + seg_off[s] + cbase[oc];
    const int first = owner_local[chunk] * KS_CHUNK;
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int members = min(KS_CHUNK, cnt - first);
    const int nblk = (members + 63) / 64;

    KcFold k[KC_FW];
    float inv_u[KC_FW];
    bool live[KC_FW];
    bool any_live = false;
    // csum != nullptr: the binade prediction of km_chunk_predict_kernel happens here (one launch less per Lloyd iteration): the any-order
    // prefix of the cluster's earlier tail chunks (lanes = predecessor chunks) on top of the exact head sum
    const int pbase = csum ? cchunk[oc] : 0, plocal = owner_local[chunk];
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        const int f = f0 + wave * KC_FW + i;
        int e = KC_UNSAFE;
        if (f < C) {
            if (pred) {
                e = kp_predict(*pred, (size_t)oc * C + f, plocal);
                if (lane == 0) cexp[(size_t)chunk * C + f] = (int8_t)e;
            } else if (csum) {
                float part = 0.0f;
                for (int c = start_chunk + lane; c < plocal; c += 64) part += csum[(size_t)(pbase + c) * C + f];
                part = aoc_wave_sum(part);
                const float pre = ((start_chunk > 0) ? head_state[(size_t)oc * C + f] : 0.0f) + part;
                e = kc_predict_binade(pre, pre + csum[(size_t)chunk * C + f]);
                if (lane == 0) cexp[(size_t)chunk * C + f] = (int8_t)e;
            } else {
                e = cexp[(size_t)chunk * C + f];
            }
        }
        if (lane == 0) lexp[wave * KC_FW + i] = e;
        live[i] = e != KC_UNSAFE;
        inv_u[i] = __uint_as_float((uint32_t)(23 - (live[i] ? e : 0) + 127) << 23);
        k[i].acc = 0; k[i].par0 = 0; k[i].par1 = 1; k[i].bump0 = 0; k[i].bump1 = 0; k[i].bad = 0;
        any_live |= live[i];
    }
    __syncthreads();
    // does any wave of the block have a live feature?  (uniform per block)
    bool block_live = false;
    for (int i = 0; i < KC_FG; ++i) block_live |= (lexp[i] != KC_UNSAFE);
    if (!block_live) return;

    for (int i = threadIdx.x; i < KS_CHUNK; i += 256) loffs[i] = (i < members) ? list[first + i] : 0u;
    __syncthreads();
    // staging: 64 rows x (KC_FG / 4) float4 = 320 pieces per block; thread t takes pieces t and t + 256
    const int npiece = KC_FG / 4;
    auto issue = [&](int blk, float4 (&v)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + threadIdx.x;
            const int mloc = idx / npiece, piece = idx - mloc * npiece;
            const int m = blk * 64 + mloc;
            const bool in = idx < 64 * npiece && m < members && f0 + piece * 4 < C;
            v[it] = in ? *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pool) + loffs[in ? m : 0] + (f0 + piece * 4) * 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf, const float4 (&v)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + threadIdx.x;
            if (idx < 64 * npiece) {
                const int mloc = idx / npiece, piece = idx - mloc * npiece;
                float *d = tile[buf] + mloc * KC_G_LD + piece * 4;
                d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
            }
        }
    };
    float4 v[2];
    issue(0, v);
    commit(0, v);
    for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) issue(b + 1, v);            // in flight under this block's folds
        __syncthreads();                              // tile[b & 1] complete
        if (any_live) {
#pragma unroll
            for (int i = 0; i < KC_FW; ++i)
                if (live[i]) kc_fold_block2(tile[b & 1][lane * KC_G_LD + wave * KC_FW + i], inv_u[i], k[i]);
        }
        if (b + 1 < nblk) commit((b + 1) & 1, v);     // the other buffer was last read two iterations ago
    }
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        if (live[i]) {
            const int f = f0 + wave * KC_FW + i;
            const int t = ks_wave_sum(k[i].acc);
            if (lane == 0) {
                if (k[i].bad) cexp[(size_t)chunk * C + f] = (int8_t)KC_UNSAFE;
                cinc0[(size_t)chunk * C + f] = t + k[i].bump0;
                cinc1[(size_t)chunk * C + f] = t + k[i].bump1;
            }
        }
    }
}
// The fold of ONE chunk x feature group in the predicted binades (KsPred), for the heads launch: that launch's workgroups own 43 KB of LDS each
// (the heads' ring), so only three of them fit a CU and the block-by-block pipeline of kc_chunk_fold_body -- one memory round trip per
// 64-member block, hidden by eight resident workgroups in its own launch -- would be exposed.  Here the workgroup requests ALL its rows at once
// (512 members x 80 bytes: ten 16-byte loads per thread, offsets first), parks them in the LDS it owns anyway and folds the eight blocks
// back to back: two round trips per chunk instead of eight.  Same arithmetic, same summaries.
constexpr int KC_WIDE_LDS_FLOATS = KS_CHUNK * KC_G_LD + KC_FG;
__device__ __forceinline__ void kc_chunk_fold_wide_body(int b, float *__restrict__ lds, const float *__restrict__ pool, int C, const int32_t *__restrict__ seg_off,
                                                        const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                        const uint32_t *__restrict__ moff, int kmax,
                                                        const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                        int8_t *__restrict__ cexp, int32_t *__restrict__ cinc0, int32_t *__restrict__ cinc1,
                                                        int start_chunk, int n_chunks_grid, int n_groups, int xcd_aware, const KsPred &pred) {
    float *tile = lds;                                                  // [KS_CHUNK][KC_G_LD]
    int *lexp = reinterpret_cast<int *>(lds + KS_CHUNK * KC_G_LD);
    int chunk, grp;
    {
        const int blk = b / (8 * n_groups), rem = b - blk * (8 * n_groups);
        const int pc = min(8, n_chunks_grid - blk * 8);
        grp = rem / pc;
        chunk = blk * 8 + rem - grp * pc;
        if (!xcd_aware) { chunk = b % n_chunks_grid; grp = b / n_chunks_grid; }
    }
    const int oc = owner_cluster[chunk];
    if (oc < 0) return;
    const int plocal = owner_local[chunk];
    if (plocal < start_chunk) return;
    const int f0 = grp * KC_FG;
    const int s = oc / kmax;
    const int cnt = counts[oc];
    const uint32_t *list = moff + seg_off[s] + cbase[oc];
    const int first = plocal * KS_CHUNK;
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int members = min(KS_CHUNK, cnt - first);
    const int nblk = (members + 63) / 64;
    // the member offsets this thread needs (its ten pieces belong to members idx / 5): requested before anything else
    constexpr int NP = KC_FG / 4;                                       // 5 float4 pieces per member
    constexpr int NIT = KS_CHUNK * NP / 256;                            // 10 pieces per thread
    uint32_t off[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) off[it] = list[first + min((it * 256 + (int)threadIdx.x) / NP, members - 1)];

    KcFold k[KC_FW];
    float inv_u[KC_FW];
    bool live[KC_FW];
    bool any_live = false;
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        const int f = f0 + wave * KC_FW + i;
        int e = KC_UNSAFE;
        if (f < C) {
            e = kp_predict(pred, (size_t)oc * C + f, plocal);
            if (lane == 0) cexp[(size_t)chunk * C + f] = (int8_t)e;
        }
        if (lane == 0) lexp[wave * KC_FW + i] = e;
        live[i] = e != KC_UNSAFE;
        inv_u[i] = __uint_as_float((uint32_t)(23 - (live[i] ? e : 0) + 127) << 23);
        k[i].acc = 0; k[i].par0 = 0; k[i].par1 = 1; k[i].bump0 = 0; k[i].bump1 = 0; k[i].bad = 0;
        any_live |= live[i];
    }
    __syncthreads();
    bool block_live = false;
    for (int i = 0; i < KC_FG; ++i) block_live |= (lexp[i] != KC_UNSAFE);
    if (!block_live) return;                                            // (uniform per workgroup)
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int mloc = idx / NP, piece = idx - mloc * NP;
        const bool in = f0 + piece * 4 < C;
        v[it] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pool) + off[it] + (in ? (f0 + piece * 4) * 4 : 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int mloc = idx / NP, piece = idx - mloc * NP;
        const bool in = mloc < members && f0 + piece * 4 < C;
        float *d = tile + mloc * KC_G_LD + piece * 4;
        d[0] = in ? v[it].x : 0.0f; d[1] = in ? v[it].y : 0.0f; d[2] = in ? v[it].z : 0.0f; d[3] = in ? v[it].w : 0.0f;
    }
    __syncthreads();
    if (!any_live) return;
    for (int bb = 0; bb < nblk; ++bb) {
#pragma unroll
        for (int i = 0; i < KC_FW; ++i)
            if (live[i]) kc_fold_block2(tile[(bb * 64 + lane) * KC_G_LD + wave * KC_FW + i], inv_u[i], k[i]);
    }
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        if (live[i]) {
            const int f = f0 + wave * KC_FW + i;
            const int t = ks_wave_sum(k[i].acc);
            if (lane == 0) {
                if (k[i].bad) cexp[(size_t)chunk * C + f] = (int8_t)KC_UNSAFE;
                cinc0[(size_t)chunk * C + f] = t + k[i].bump0;
                cinc1[(size_t)chunk * C + f] = t + k[i].bump1;
            }
        }
    }
}
__global__ __launch_bounds__(256) void km_chunk_fold_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ seg_off,
                                                             const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                             const uint32_t *__restrict__ moff, int kmax,
                                                             const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                             int8_t *__restrict__ cexp, int32_t *__restrict__ cinc0, int32_t *__restrict__ cinc1,
                                                             int start_chunk, const float *__restrict__ csum, const int32_t *__restrict__ cchunk,
                                                             const float *__restrict__ head_state, int n_chunks_grid, int n_groups, int xcd_aware,
                                                             KsPred pred) {
    __shared__ __attribute__((aligned(16))) float lds[KC_FOLD_LDS_FLOATS];
    kc_chunk_fold_body(blockIdx.x, lds, pool, C, seg_off, counts, cbase, moff, kmax, owner_cluster, owner_local, cexp, cinc0, cinc1, start_chunk, csum, cchunk,
                       head_state, n_chunks_grid, n_groups, xcd_aware, pred.e0 ? &pred : nullptr);
}

// P0 + P1 + P2 in one pass over the rows ("scan-fold").  Same grid and roles as km_chunk_fold_kernel, but the workgroup keeps all
// KS_T member blocks of its chunk (its feature group's columns) in LDS and
//   1. sums them in any order and PUBLISHES the chunk's local sums (csum + a flag per (chunk, feature group)),
//   2. adds up the local sums of the cluster's earlier tail chunks (lanes = predecessor chunks; a chunk id is dispatched after every
//      smaller one, so the flags it waits for belong to workgroups that are running or done) on top of the exact head sum: the
//      any-order prefix the binade prediction needs,
//   3. predicts the binade exactly as km_chunk_predict_kernel does and folds the staged rows in it.
// A flag that does not arrive within the polling budget only makes the wave mark its features KC_UNSAFE: the stitch then folds that
// chunk from its rows -- time, never exactness (and no dependence on the dispatch order for progress).
constexpr int KC_POLL_BUDGET = 1 << 16;
__global__ __launch_bounds__(256) void km_chunk_scanfold_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ seg_off,
                                                                 const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                                 const uint32_t *__restrict__ moff, int kmax,
                                                                 const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                                 const int32_t *__restrict__ cchunk, const float *__restrict__ head_state,
                                                                 float *__restrict__ csum, uint32_t *__restrict__ cflag,
                                                                 int8_t *__restrict__ cexp, int32_t *__restrict__ cinc0, int32_t *__restrict__ cinc1,
                                                                 int start_chunk) {
    __shared__ float tile[2][64 * KC_G_LD];
    __shared__ uint32_t loffs[KS_CHUNK];
    const int chunk = blockIdx.x, grp = blockIdx.y;
    const int oc = owner_cluster[chunk];
    if (oc < 0) return;
    const int ol = owner_local[chunk];
    if (ol < start_chunk) return;
    const int f0 = grp * KC_FG;
    const int s = oc / kmax;
    const int cnt = counts[oc];
    const uint32_t *list = moff + seg_off[s] + cbase[oc];
    const int first = ol * KS_CHUNK;
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int members = min(KS_CHUNK, cnt - first);
    const int nblk = (members + 63) / 64;

    for (int i = threadIdx.x; i < KS_CHUNK; i += blockDim.x) loffs[i] = (i < members) ? list[first + i] : 0u;
    __syncthreads();
    // staging of one member block: 64 rows x (KC_FG / 4) float4 = 320 pieces, thread t takes pieces t and t + 256 (branch-free: every
    // lane loads from a valid address and selects afterwards).  The chunk is streamed twice through a double-buffered tile: once for
    // the local sums, once for the folds (the second pass hits L2); keeping all KS_T blocks in LDS instead costs occupancy (45 KB).
    const int npiece = KC_FG / 4;
    auto issue = [&](int blk, float4 (&v)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + threadIdx.x;
            const int mloc = idx / npiece, piece = idx - mloc * npiece;
            const int m = blk * 64 + mloc;
            const bool in = idx < 64 * npiece && m < members && f0 + piece * 4 < C;
            const float4 x = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(pool) + loffs[in ? m : 0] + (in ? (f0 + piece * 4) * 4 : 0));
            v[it].x = in ? x.x : 0.f; v[it].y = in ? x.y : 0.f; v[it].z = in ? x.z : 0.f; v[it].w = in ? x.w : 0.f;
        }
    };
    auto commit = [&](int buf, const float4 (&v)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = it * 256 + threadIdx.x;
            if (idx < 64 * npiece) {
                const int mloc = idx / npiece, piece = idx - mloc * npiece;
                float *d = tile[buf] + mloc * KC_G_LD + piece * 4;
                d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
            }
        }
    };
    float4 v[2];
    float local[KC_FW];
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) local[i] = 0.0f;
    issue(0, v);
    commit(0, v);
    for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) issue(b + 1, v);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KC_FW; ++i) local[i] += tile[b & 1][lane * KC_G_LD + wave * KC_FW + i];
        if (b + 1 < nblk) commit((b + 1) & 1, v);
    }
    issue(0, v);                                       // first block of the second pass: in flight under the publication and the look-back

    // ---- 1. local any-order sums (wave = KC_FW features, lanes = members), published for the later chunks of the cluster
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        local[i] = aoc_wave_sum(local[i]);
        const int f = f0 + wave * KC_FW + i;
        if (lane == 0 && f < C) __hip_atomic_store(csum + (size_t)chunk * C + f, local[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (device-scope stores go through to the memory all XCDs see; the wave waits for theirs to be acknowledged, then the flag -- no
    // release / acquire fences: across XCDs those write back and invalidate whole L2s)
    __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(cflag + (size_t)chunk * 8 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- 2. any-order prefix: exact head sum + the local sums of the earlier tail chunks (chunk ids of a cluster are contiguous)
    const int cc = cchunk[oc];
    const int npred = ol - start_chunk;
    float pre[KC_FW];
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        const int f = f0 + wave * KC_FW + i;
        pre[i] = (start_chunk > 0 && f < C) ? head_state[(size_t)oc * C + f] : 0.0f;
    }
    bool late = false;
    for (int p0 = 0; p0 < npred; p0 += 64) {
        const int pc = cc + start_chunk + p0 + lane;
        const bool have = p0 + lane < npred;
        if (have) {
            int polls = 0;
            while (__hip_atomic_load(cflag + (size_t)pc * 8 + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++polls > KC_POLL_BUDGET) { late = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
#pragma unroll
        for (int i = 0; i < KC_FW; ++i) {
            const int f = f0 + wave * KC_FW + i;
            float x = 0.0f;
            if (have && f < C) x = __hip_atomic_load(csum + (size_t)pc * C + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pre[i] += aoc_wave_sum(x);
        }
    }
    late = __builtin_amdgcn_ballot_w64(late) != 0ull;

    // ---- 3. binade prediction (the same test as km_chunk_predict_kernel) and the integer folds of the staged rows
    KcFold k[KC_FW];
    float inv_u[KC_FW];
    int ebin[KC_FW];
    bool live[KC_FW];
    bool any_live = false;
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        const int f = f0 + wave * KC_FW + i;
        const float end = pre[i] + local[i];
        int e = KC_UNSAFE;
        if (!late && f < C && pre[i] > 1e-30f && end < 1e30f && end >= pre[i]) {
            const int e_lo = (int)((__float_as_uint(pre[i] * 0.9995f) >> 23) & 0xff) - 127;
            const int e_hi = (int)((__float_as_uint(end * 1.0005f) >> 23) & 0xff) - 127;
            if (e_lo == e_hi && e_lo >= -100 && e_lo <= 100) e = e_lo;
        }
        live[i] = e != KC_UNSAFE;
        if (lane == 0 && f < C && !live[i]) cexp[(size_t)chunk * C + f] = (int8_t)KC_UNSAFE;
        inv_u[i] = __uint_as_float((uint32_t)(23 - (live[i] ? e : 0) + 127) << 23);
        k[i].acc = 0; k[i].par0 = 0; k[i].par1 = 1; k[i].bump0 = 0; k[i].bump1 = 0; k[i].bad = 0;
        any_live |= live[i];
        ebin[i] = e;
    }
    // does any wave of the workgroup fold anything?  (uniform per workgroup: the second pass has barriers)
    __shared__ int lany;
    __syncthreads();                                   // every wave is done with the tile of the first pass
    if (threadIdx.x == 0) lany = 0;
    __syncthreads();
    if (any_live && lane == 0) lany = 1;
    __syncthreads();
    if (!lany) return;
    commit(0, v);
    for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) issue(b + 1, v);
        __syncthreads();
        if (any_live) {
#pragma unroll
            for (int i = 0; i < KC_FW; ++i)
                if (live[i]) kc_fold_block2(tile[b & 1][lane * KC_G_LD + wave * KC_FW + i], inv_u[i], k[i]);
        }
        if (b + 1 < nblk) commit((b + 1) & 1, v);
    }
#pragma unroll
    for (int i = 0; i < KC_FW; ++i) {
        if (live[i]) {
            const int f = f0 + wave * KC_FW + i;
            const int t = ks_wave_sum(k[i].acc);
            if (lane == 0) {
                cexp[(size_t)chunk * C + f] = k[i].bad ? (int8_t)KC_UNSAFE : (int8_t)ebin[i];
                cinc0[(size_t)chunk * C + f] = t + k[i].bump0;
                cinc1[(size_t)chunk * C + f] = t + k[i].bump1;
            }
        }
    }
}

// Serial stitch, one wave per (cluster, 4 features).
// MODE 0: centroids[s,j,4q..4q+3] = ordered sum / count (empty cluster untouched, vq.py:820-823)
// MODE 1: proxies[s,1,j,4q..] = mean of the listed rows (AEM:280), zeros when empty
// For every chunk the wave first tries the chunk's summary (predicted binade e, integer increments for both
// parities): it applies iff the EXACT running sum is in binade e and n stays below 2^24 -- then all members
// of the chunk were added in that binade and the summary is the exact result.  Otherwise (first chunk, binade
// crossing, unusable summary) the chunk's rows are folded block by block, with the literal serial additions
// where a block itself crosses.  Summaries of 64 chunks sit in registers (v_readlane), rows of the next chunk
// that will need them are prefetched, so memory latency stays off the chain.
template <int NF>
struct KsChunkT {
    uint32_t x[KS_T][NF];
};

// One chunk (KS_T blocks of 64 members) of feature F, exactly, from running sum s.  All blocks are folded at once in
// the integer domain of the current binade -- they are independent, so the memory-free arithmetic pipelines instead
// of paying the dependent-latency of KS_T serial block steps -- and a scalar walk over the block totals finds the
// block (if any) where n would leave the binade; only that block takes the crossing-aware path (ks_block_exact),
// after which the remaining blocks are folded again in the new binade.  Ties, negative or non-finite values and
// s == 0 fall back to the per-block path, which ends in the literal serial additions.
template <int F, int NF>
__device__ __forceinline__ float ks_chunk_exact(float s, const KsChunkT<NF> &ch, int nblk, int lane, bool dbg = false) {
    int from = 0;
#pragma unroll 1
    for (int round = 0; round < 3 && from < nblk; ++round) {
        const KsBinade bb = ks_binade(s);
        if (!bb.ok) break;
        int r[KS_T];
        unsigned long long tie[KS_T], odd[KS_T];
        bool bad = false;
#pragma unroll
        for (int b = 0; b < KS_T; ++b) {
            const float y = __uint_as_float(ch.x[b][F]) * bb.inv_u;
            const float rn = rintf(y);
            const bool is_tie = fabsf(rn - y) == 0.5f;
            const int rr = is_tie ? (int)floorf(y) : (int)rn;             // ties start from floor(y); the walk re-rounds them
            r[b] = (b >= from) ? rr : 0;
            bad |= (b >= from) && !(__float_as_uint(y) < 0x4B800000u);    // y in [+0, 2^24)
            tie[b] = __ballot(is_tie && b >= from);
            odd[b] = __ballot((r[b] & 1) != 0);
        }
        if (__ballot(bad) != 0ull) break;
        int t[KS_T];
#pragma unroll
        for (int b = 0; b < KS_T; ++b) t[b] = ks_wave_sum(r[b]);
        // scalar walk over the block totals: ties are re-rounded from the running parity (a tie leaves n even)
        int n = bb.n_in, par = bb.n_in & 1, bcross = -1;
#pragma unroll
        for (int b = 0; b < KS_T; ++b) {
            if (b >= from && b < nblk && bcross < 0) {
                int bump = 0, base_par = par, lo = 0;
                unsigned long long tm = tie[b];
                while (tm) {
                    const int tl = __builtin_ctzll(tm);
                    tm &= tm - 1;
                    const unsigned long long below_t = (tl == 0) ? 0ull : (~0ull >> (64 - tl));
                    const unsigned long long below_lo = (lo == 0) ? 0ull : (~0ull >> (64 - lo));
                    const int pb = base_par ^ (__popcll(odd[b] & below_t & ~below_lo) & 1);   // parity of n in front of lane tl
                    bump += pb ^ (int)((odd[b] >> tl) & 1ull);                                 // n + floor(y) odd -> round up
                    base_par = 0;
                    lo = tl + 1;
                }
                const unsigned long long rest = (lo >= 64) ? 0ull : (~0ull << lo);
                const int inc = t[b] + bump;
                if (n + inc > 0xFFFFFF) {
                    bcross = b;
                } else {
                    n += inc;
                    par = base_par ^ (__popcll(odd[b] & rest) & 1);
                }
            }
        }
        s = (float)n * bb.u;
        if (bcross < 0) return s;
        float xc = 0.0f;
#pragma unroll
        for (int b = 0; b < KS_T; ++b)
            if (b == bcross) xc = __uint_as_float(ch.x[b][F]);
        s = ks_block_exact(s, xc, lane);
        from = bcross + 1;
    }
#pragma unroll
    for (int b = 0; b < KS_T; ++b)
        if (b >= from && b < nblk) s = ks_block_exact(s, __uint_as_float(ch.x[b][F]), lane);
    return s;
}

constexpr int KS_SCAN_WAVES = 4;      // waves (feature slices) per workgroup of the stitch
template <int MODE, int NF>
__global__ __launch_bounds__(KS_SCAN_WAVES * 64) void km_sum_scan_kernel(const float *__restrict__ pool, uint32_t pool_bytes, int C,
                                                          const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                          const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                          const uint32_t *__restrict__ moff, int kmax, float *__restrict__ dst,
                                                          const int32_t *__restrict__ cchunk, const int8_t *__restrict__ cexp,
                                                          const int32_t *__restrict__ cinc0, const int32_t *__restrict__ cinc1,
                                                          int start_chunk, const float *__restrict__ head_state, int n_seg_grid, int xcd_aware,
                                                          KsPred pred) {
    // 1-D grid of workgroups of KS_SCAN_WAVES independent waves (no barrier, no LDS: a workgroup is only the unit of dispatch -- most clusters
    // have no tail and their waves leave at once, and a launch of C x kmax x n_seg ONE-wave workgroups is bound by the dispatcher: 172 800 of them
    // at cfg3, F = 3, took 37 us with nothing to do).  The waves of a workgroup take neighbouring feature slices of ONE cluster.  XCD-aware: the
    // C / NF waves of a cluster each read 4 NF bytes of the same member rows (the chunks whose summaries do not apply); workgroups whose ids
    // differ by 8 run on one XCD and fetch every 64-byte sector once instead of once per XCD.  Blocks of 8 clusters x n_qg workgroups; the last
    // block may hold fewer clusters.
    int s, j, q;
    {
        const int n_q = C / NF, n_qg = (n_q + KS_SCAN_WAVES - 1) / KS_SCAN_WAVES, n_cl = kmax * n_seg_grid;
        const int b = blockIdx.x, blk = b / (8 * n_qg), rem = b - blk * (8 * n_qg);
        const int pc = min(8, n_cl - blk * 8);
        int qg = rem / pc;
        int c = blk * 8 + rem - qg * pc;
        if (!xcd_aware) { qg = b % n_qg; c = b / n_qg; }
        q = qg * KS_SCAN_WAVES + (int)(threadIdx.x >> 6);
        if (q >= n_q) return;
        j = c % kmax;
        s = c / kmax;
    }
    if (j >= seg_k[s]) return;
    const int cnt = counts[s * kmax + j];
    if (start_chunk > 0 && cnt <= start_chunk * KS_CHUNK) return;       // finished by km_ordered_sum_kernel
    const int lane = threadIdx.x & 63;
    float *out = (MODE == 0) ? dst + ((size_t)s * kmax + j) * C + NF * q : dst + (((size_t)s * 2 + 1) * kmax + j) * C + NF * q;
    if (cnt == 0) {
        if (MODE == 1 && lane < NF) out[lane] = 0.0f;
        return;
    }
    const bool have_summ = cexp != nullptr;
    const size_t srow = have_summ ? ((size_t)cchunk[s * kmax + j] * C + NF * q) : 0;
    const uint32_t *list = moff + seg_off[s] + cbase[s * kmax + j];
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pool), 0, pool_bytes, 0x00020000);
    const uint32_t qoff = (uint32_t)q * (NF * 4u);
    typedef KsChunkT<NF> KsChunk;
    const int n_chunks = (cnt + KS_CHUNK - 1) / KS_CHUNK;

    auto load_offs = [&](int c, uint32_t (&o)[KS_T]) {
        // branch-free: clamped indices, all eight offset loads issued before any is looked at, selects afterwards (the empty asm keeps
        // hipcc from sinking each load back under its condition, which serialises eight round trips)
        uint32_t v[KS_T];
#pragma unroll
        for (int b = 0; b < KS_T; ++b) v[b] = list[min((c * KS_T + b) * 64 + lane, cnt - 1)];
#pragma unroll
        for (int b = 0; b < KS_T; ++b) asm volatile("" : "+v"(v[b]));
#pragma unroll
        for (int b = 0; b < KS_T; ++b) {
            const int m = (c * KS_T + b) * 64 + lane;
            o[b] = (m < cnt) ? v[b] + qoff : KU_INVALID_OFF;
        }
    };
    auto load_rows = [&](const uint32_t (&o)[KS_T], KsChunk &ch) {
#pragma unroll
        for (int b = 0; b < KS_T; ++b) {
            if constexpr (NF == 4) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o[b], 0, 0);
                ch.x[b][0] = v.x; ch.x[b][1] = v.y; ch.x[b][2] = v.z; ch.x[b][3] = v.w;
            } else if constexpr (NF == 2) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, o[b], 0, 0);
                ch.x[b][0] = v.x; ch.x[b][1] = v.y;
            } else {
                ch.x[b][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, o[b], 0, 0);
            }
        }
    };
    auto fetch_rows = [&](int c, KsChunk &ch) {
        uint32_t o[KS_T];
        load_offs(c, o);
        load_rows(o, ch);
    };

    // summaries of chunks [batch0, batch0 + 64) in registers (lane l = chunk batch0 + l), next batch prefetched
    struct Summ { int e[NF]; int i0[NF], i1[NF]; };
    Summ sc, sx;
#pragma unroll
    for (int f = 0; f < NF; ++f) { sc.e[f] = sx.e[f] = KC_UNSAFE; sc.i0[f] = sc.i1[f] = sx.i0[f] = sx.i1[f] = 0; }
    int batch0 = 0;
    unsigned long long need_cur = ~0ull, need_next = ~0ull;
    auto unsafe_any = [](const Summ &m) -> bool {
        bool u = false;
#pragma unroll
        for (int f = 0; f < NF; ++f) u |= (m.e[f] == KC_UNSAFE);
        return u;
    };
    auto fetch_batch = [&](int b0, Summ &m) {
        const int c = b0 + lane;
#pragma unroll
        for (int f = 0; f < NF; ++f) m.e[f] = KC_UNSAFE;
        if (have_summ && c < n_chunks) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                m.e[f] = cexp[srow + (size_t)c * C + f];
                m.i0[f] = cinc0[srow + (size_t)c * C + f];
                m.i1[f] = cinc1[srow + (size_t)c * C + f];
            }
        }
    };
    if (have_summ) {
        fetch_batch(0, sc);
        fetch_batch(64, sx);
        need_cur = __ballot(unsafe_any(sc));
        need_next = __ballot(unsafe_any(sx));
    }
    // first chunk >= from (inside the two known batches) whose rows will be needed; n_chunks if none is known
    auto next_needed = [&](int from) -> int {
        int d = from - batch0;
        if (d < 64) {
            const unsigned long long m = need_cur & (~0ull << d);
            if (m) return min(n_chunks, batch0 + __builtin_ctzll(m));
            d = 64;
        }
        if (d < 128) {
            const unsigned long long m = need_next & (~0ull << (d - 64));
            if (m) return min(n_chunks, batch0 + 64 + __builtin_ctzll(m));
        }
        return n_chunks;
    };

    const bool ks_big = cnt > 30000 && q == 0;     // instrumentation target (AOC_KS_STATS builds only)
    long long ks_t0 = KS_CLK(), ks_tsum = 0, ks_tex = 0;
    (void)ks_big; (void)ks_t0; (void)ks_tsum; (void)ks_tex;
    // running sums: while a sum sits in a plain binade it is kept as (exponent, n = s / ulp) in integers, so a verified
    // chunk summary is applied with a handful of scalar operations; st[] holds the float whenever that form is not valid
    float st[NF];
    int se[NF], sn[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { st[f] = 0.0f; se[f] = KC_UNSAFE; sn[f] = 0; }
    auto to_int = [&](int f) {
        const KsBinade bb = ks_binade(st[f]);
        se[f] = bb.ok ? (int)((__float_as_uint(st[f]) >> 23) & 0xff) - 127 : KC_UNSAFE;
        sn[f] = bb.n_in;
    };
    auto to_float = [&](int f) -> float {
        return (se[f] != KC_UNSAFE) ? (float)sn[f] * __uint_as_float((uint32_t)(se[f] - 23 + 127) << 23) : st[f];
    };
    if (start_chunk > 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            st[f] = head_state[((size_t)s * kmax + j) * C + NF * q + f];
            to_int(f);
        }
    }
    // what the next Lloyd iteration's fold predicts its binades from (KsPred): exponent in front of the tail, chunks that changed it
    int pe0[NF], pn[NF], pci[NF][KP_CROSS], pce[NF][KP_CROSS];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        pe0[f] = se[f];
        pn[f] = 0;
#pragma unroll
        for (int i = 0; i < KP_CROSS; ++i) { pci[f][i] = 0xFFFF; pce[f][i] = KC_UNSAFE; }
    }
    auto note_crossing = [&](int f, int c, int e_before) {
        if (se[f] == e_before) return;
#pragma unroll
        for (int i = 0; i < KP_CROSS; ++i) {
            if (i == pn[f]) { pci[f][i] = c; pce[f][i] = (i + 1 < KP_CROSS) ? se[f] : KC_UNSAFE; }      // the last slot only says "unknown from here on"
        }
        if (pn[f] < KP_CROSS) pn[f] += 1;
    };
    // rows of the next chunk that will need them (pending) are in flight in nxt; the member offsets of the one after
    // that (pending2) are already in registers, so its rows cost one memory latency, not two, when their turn comes
    KsChunk cur, nxt;
    uint32_t onext[KS_T];
    int pending = next_needed(start_chunk), pending2 = n_chunks;
    if (pending < n_chunks) {
        fetch_rows(pending, nxt);
        pending2 = next_needed(pending + 1);
        if (pending2 < n_chunks) load_offs(pending2, onext);
    }

    for (int c = start_chunk; c < n_chunks; ++c) {
        if (have_summ && c - batch0 >= 64) {                  // advance to the prefetched batch, prefetch the one after
            batch0 += 64;
            sc = sx;
            need_cur = need_next;
            fetch_batch(batch0 + 64, sx);
            need_next = __ballot(unsafe_any(sx));
            if (pending >= n_chunks) {                        // nothing was known to be needed: look again
                pending = next_needed(c);
                if (pending < n_chunks) fetch_rows(pending, nxt);
            }
            if (pending < n_chunks && pending2 >= n_chunks) {
                pending2 = next_needed(pending + 1);
                if (pending2 < n_chunks) load_offs(pending2, onext);
            }
        }
        bool skip[NF];
        bool all_skip = true;
        int e_in[NF];
        const long long ks_ta = KS_CLK();
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            skip[f] = false;
            e_in[f] = se[f];
            if (have_summ) {
                const int l = c - batch0;
                const int e = __builtin_amdgcn_readlane(sc.e[f], l);
                if (e != KC_UNSAFE && e == se[f]) {                   // prediction verified against the exact sum
                    const int inc = (sn[f] & 1) ? __builtin_amdgcn_readlane(sc.i1[f], l) : __builtin_amdgcn_readlane(sc.i0[f], l);
                    const int n_out = sn[f] + inc;
                    if (n_out <= 0xFFFFFF && n_out >= sn[f]) {        // n never left the binade inside the chunk
                        sn[f] = n_out;
                        skip[f] = true;
                    }
                }
            }
            all_skip &= skip[f];
        }
        ks_tsum += KS_CLK() - ks_ta;
#ifdef AOC_KS_STATS
        if (lane == 0) {                                               // all waves: tail chunks visited / verified / refolded without a prefetch
            atomicAdd(&aoc_ks_stats[5], 1ull);
            if (all_skip) atomicAdd(&aoc_ks_stats[6], 1ull);
            else if (pending != c) atomicAdd(&aoc_ks_stats[7], 1ull);
        }
#endif
        if (all_skip) continue;                                        // (a verified summary never changes the exponent)
        const long long ks_tb = KS_CLK();
        // ---- this chunk needs its rows
        if (pending == c) {
            cur = nxt;
            pending = pending2;
            if (pending < n_chunks) load_rows(onext, nxt);             // offsets arrived long ago: one latency to go
            pending2 = (pending < n_chunks) ? next_needed(pending + 1) : n_chunks;
            if (pending2 < n_chunks) load_offs(pending2, onext);
        } else {
            fetch_rows(c, cur);                                        // not prefetched (mispredicted summary): rare
        }
        const int blocks_here = min(KS_T, (cnt - c * KS_CHUNK + 63) / 64);
#ifdef AOC_KS_STATS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ks_big) KS_STAT(3, KS_CLK() - ks_tb);      // big wave: cycles waiting for the rows
#endif
        if (!skip[0]) { st[0] = ks_chunk_exact<0, NF>(to_float(0), cur, blocks_here, lane, ks_big); to_int(0); }
        if constexpr (NF > 1) { if (!skip[1]) { st[1] = ks_chunk_exact<1, NF>(to_float(1), cur, blocks_here, lane, ks_big); to_int(1); } }
        if constexpr (NF > 2) {
            if (!skip[2]) { st[2] = ks_chunk_exact<2, NF>(to_float(2), cur, blocks_here, lane, ks_big); to_int(2); }
            if (!skip[3]) { st[3] = ks_chunk_exact<3, NF>(to_float(3), cur, blocks_here, lane, ks_big); to_int(3); }
        }
        ks_tex += KS_CLK() - ks_tb;
        if (ks_big) KS_STAT(2, 1);
#pragma unroll
        for (int f = 0; f < NF; ++f) note_crossing(f, c, e_in[f]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) st[f] = to_float(f);
    if (ks_big) { KS_STAT(0, KS_CLK() - ks_t0); KS_STAT(1, ks_tsum); KS_STAT(4, ks_tex); }
    if (lane == 0) {
        const float fc = (float)cnt;
#pragma unroll
        for (int f = 0; f < NF; ++f) out[f] = st[f] / fc;
        if (pred.e0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const size_t idx = ((size_t)s * kmax + j) * C + NF * q + f;
                pred.e0[idx] = (int8_t)pe0[f];
#pragma unroll
                for (int i = 0; i < KP_CROSS; ++i) {
                    pred.cidx[idx * KP_CROSS + i] = (uint16_t)pci[f][i];
                    pred.ce[idx * KP_CROSS + i] = (int8_t)pce[f][i];
                }
            }
        }
    }
}

// ==========================================================================================
// Ordered sums, literally: one workgroup per (cluster, group of 28 features), lanes = features.  The consumer wave
// adds the member rows one after another in float32 -- the scipy sequence itself, so nothing is assumed about the
// data (sign, range, NaN) -- while seven producer waves gather the group's 112 bytes of every member row through
// the ordered member list into a double-buffered LDS batch: member offsets OS_DEPTH + 1 batches ahead, row pieces
// OS_DEPTH batches ahead (all in flight across the barriers), so the chain only ever waits on LDS and costs a few
// cycles per member.  Splitting the features over workgroups keeps the per-CU gather rate low enough for that.
// MODE 0: centroids[s,j,:] = sum / count (empty cluster untouched, vq.py:820-823)
// MODE 1: proxies[s,1,j,:] = mean of the listed rows (AEM:280), zeros when empty
constexpr int OS_BATCH = 128;                   // rows per LDS buffer
constexpr int OS_FP = 7;                        // 16-byte pieces (4 features each) per feature group
constexpr int OS_LD = 32;                       // LDS row stride in floats
constexpr int OS_NPROD = 7;                     // producer waves (448 lanes: exactly 2 pieces per lane per batch)
constexpr int OS_PP = OS_BATCH * OS_FP / (OS_NPROD * 64);
constexpr int OS_DEPTH = 6;                     // batches of row pieces in flight per lane (even)
static_assert(OS_PP * OS_NPROD * 64 == OS_BATCH * OS_FP && OS_DEPTH % 2 == 0 && OS_FP * 4 <= OS_LD, "ordered-sum geometry");
inline int os_groups(int C) { return (C / 4 + OS_FP - 1) / OS_FP; }
__device__ __forceinline__ int os_groups_dev(int C) { return (C / 4 + OS_FP - 1) / OS_FP; }

template <int MODE>
__device__ __forceinline__ void os_ordered_sum_body(int j, int s, int grp, float *__restrict__ os_lds, const float *__restrict__ pool, uint32_t pool_bytes, int C,
                                                    const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                    const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                    const uint32_t *__restrict__ moff, int kmax, float *__restrict__ dst,
                                                    int member_cap, float *__restrict__ head_state) {
    if (j >= seg_k[s]) return;
    const int oc = s * kmax + j;
    const int cnt_all = counts[oc];
    const int f0 = grp * OS_FP * 4;
    const int nfeat = min(OS_FP * 4, C - f0);
    float *out = ((MODE == 0) ? dst + (size_t)oc * C : dst + (((size_t)s * 2 + 1) * kmax + j) * C) + f0;
    if (cnt_all == 0) {
        if (MODE == 1 && (int)threadIdx.x < nfeat) out[threadIdx.x] = 0.0f;
        return;
    }
    // head mode: a cluster larger than member_cap only gets its first member_cap members summed here (the running
    // sums go to head_state and the chunk-parallel stitch continues from there)
    const bool head_only = member_cap > 0 && cnt_all > member_cap;
    const int cnt = head_only ? member_cap : cnt_all;
    const int wave = threadIdx.x >> 6, lane = aoc_lane();
    const int nb = (cnt + OS_BATCH - 1) / OS_BATCH;

    if (wave == 0) {
        // ---- consumer: the sequential float32 sum (rows past the count are zeros: s + 0.0f == s)
        const bool active = lane < nfeat;
        const float *col = os_lds + (active ? lane : 0);
        float sum = 0.0f;
        for (int b = 0; b < nb; ++b) {
            __syncthreads();                                   // batch b is in buffer b & 1
            const float *t = col + (b & 1) * OS_BATCH * OS_LD;
            float xa[16], xb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) xa[u] = t[u * OS_LD];
#pragma unroll
            for (int g = 0; g < OS_BATCH / 16; g += 2) {
#pragma unroll
                for (int u = 0; u < 16; ++u) xb[u] = t[((g + 1) * 16 + u) * OS_LD];
#pragma unroll
                for (int u = 0; u < 16; ++u) sum = sum + xa[u];
                if (g + 2 < OS_BATCH / 16) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) xa[u] = t[((g + 2) * 16 + u) * OS_LD];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) sum = sum + xb[u];
            }
        }
        if (active) {
            if (head_only) head_state[(size_t)oc * C + f0 + lane] = sum;
            else out[lane] = sum / (float)cnt;
        }
        return;
    }

    // ---- producers
    const uint32_t *list = moff + seg_off[s] + cbase[oc];
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pool), 0, pool_bytes, 0x00020000);
    const int p = (wave - 1) * 64 + lane;
    const int c4 = C >> 2;
    int prow[OS_PP];
    uint32_t pbyte[OS_PP];
    bool pvalid[OS_PP];
#pragma unroll
    for (int i = 0; i < OS_PP; ++i) {
        const int idx = i * (OS_NPROD * 64) + p;
        prow[i] = idx / OS_FP;
        const int piece = idx - prow[i] * OS_FP;
        pvalid[i] = grp * OS_FP + piece < c4;
        pbyte[i] = (uint32_t)(grp * OS_FP + piece) * 16u;
    }
    auto load_offs = [&](int b, uint32_t (&o)[OS_PP]) {
#pragma unroll
        for (int i = 0; i < OS_PP; ++i) {
            const int m = b * OS_BATCH + prow[i];
            const uint32_t v = list[min(m, cnt - 1)];          // branch-free (clamped index, select afterwards): all offset loads of a batch in flight together
            o[i] = (b < nb && pvalid[i] && m < cnt) ? v + pbyte[i] : KU_INVALID_OFF;
        }
    };
    auto load_rows = [&](const uint32_t (&o)[OS_PP], u32x4 (&d)[OS_PP]) {
#pragma unroll
        for (int i = 0; i < OS_PP; ++i) d[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o[i], 0, 0);   // out of range -> zeros
    };
    auto write_rows = [&](int b, const u32x4 (&d)[OS_PP]) {
        float *base = os_lds + (b & 1) * OS_BATCH * OS_LD;
#pragma unroll
        for (int i = 0; i < OS_PP; ++i)
            *reinterpret_cast<u32x4 *>(base + prow[i] * OS_LD + ((pbyte[i] >> 2) - f0)) = d[i];
    };
    u32x4 d[OS_DEPTH][OS_PP];
    uint32_t o[2][OS_PP];
    {
        uint32_t po[OS_DEPTH + 2][OS_PP];
#pragma unroll
        for (int k = 0; k < OS_DEPTH + 2; ++k) load_offs(k, po[k]);
#pragma unroll
        for (int k = 0; k < OS_DEPTH; ++k) load_rows(po[k], d[k]);
        write_rows(0, d[0]);
        load_rows(po[OS_DEPTH], d[0]);                          // batch OS_DEPTH
#pragma unroll
        for (int i = 0; i < OS_PP; ++i) o[1][i] = po[OS_DEPTH + 1][i];
    }
    // at the top of iteration b: batches b + 1 .. b + DEPTH are in d[(b + 1) % DEPTH] ... (in flight), the offsets of
    // batch b + 1 + DEPTH in o[(b + 1) & 1]
    for (int b0 = 0; b0 < nb; b0 += OS_DEPTH) {
#pragma unroll
        for (int k = 0; k < OS_DEPTH; ++k) {
            const int b = b0 + k;
            if (b < nb) {
                __syncthreads();                               // the consumer is done with buffer (b + 1) & 1
                if (b + 1 < nb) write_rows(b + 1, d[(k + 1) % OS_DEPTH]);
                load_offs(b + 2 + OS_DEPTH, o[k & 1]);
                load_rows(o[(k + 1) & 1], d[(k + 1) % OS_DEPTH]);   // batch b + 1 + DEPTH
            }
        }
    }
}
// ------------------------------------------------------------------------------------------
// The same literal sums with the rows travelling by LDS-DMA (round 4): one consumer wave + two producer waves per (cluster, feature group).
// The eight-wave version above keeps its row pieces "in flight across the barriers" only on paper: hipcc cannot count vmcnt across the
// loop's branches and drains every load in front of every barrier (s_waitcnt vmcnt(0)), so each batch of 128 members costs one full memory
// round trip -- 6.4 ns per member, 66 us for a 10 240-member head at any depth, although the adding wave itself only needs ~3 ns (one
// ds_read2_b32 per two members and one dependent v_add_f32 per member, at ~4 cycles per issue slot).  Here nothing the compiler sees is a
// vector-memory operation: member offsets and row pieces travel by global_load_lds (asm statements hipcc neither counts nor drains) and
// the producers do their own vmcnt arithmetic.  Per step (128 members) a producer (64 of the members each):
//      s_waitcnt vmcnt((D-1) * 8)  -> what it issued D steps ago has landed: its half of batch b and its member offsets of batch b + D
//      s_barrier                   -> batch b belongs to the consumer, batch b - 1's ring slot is free
//      issue the offsets of batch b + 2D (one 256-byte transfer) and its rows of batch b + D (seven 1 KiB transfers whose source
//      addresses come from the landed offsets, read back from LDS)
// A transfer costs its issuing wave 60-100 cycles (MI355X_MICROARCH.md), i.e. ~700 per step and producer against the consumer's ~770
// (the single-wave form of this kernel, everything on the adding wave, was measured SLOWER than the old one for exactly that reason: 75
// against 66 us).  Indices past the cluster's end are clamped to its last member (valid addresses, uniform operation counts); the last
// batch adds +0.0f for them exactly as the zero rows of the version above did.  LDS rows are unpadded (112 B): transfer k of a producer
// fills slots [64 k, 64 k + 64) of its half lane-linearly, slot j = piece j % 7 of member j / 7.
#ifndef AOC_OD_NPROD
#define AOC_OD_NPROD 3
#endif
#ifndef AOC_OD_DEPTH
#define AOC_OD_DEPTH 1
#endif
constexpr int OD_NPROD = AOC_OD_NPROD;                         // producer waves
constexpr int OD_HALF = 64;                                    // members per producer and step
constexpr int OD_BATCH = OD_NPROD * OD_HALF;                   // members per step
constexpr int OD_DEPTH = AOC_OD_DEPTH;                         // batches of row pieces in flight
constexpr int OD_NB = OD_DEPTH + 1;                            // ring slots (rows and offsets)
constexpr int OD_LD = OS_FP * 4;                               // 28 floats per member
constexpr int OD_BATCH_BYTES = OD_BATCH * OD_LD * 4;           // 14336
constexpr int OD_RING_FLOATS = OD_NB * OD_BATCH * OD_LD;
constexpr int OD_OPS = OS_FP + 1;                              // vector-memory operations per producer and step
constexpr int OD_LDS_FLOATS = OD_RING_FLOATS + OD_NB * OD_BATCH;
static_assert(OD_DEPTH * OD_OPS < 64 && OD_NPROD >= 1 && OD_NPROD <= 3, "vmcnt is a 6-bit counter; the launch has four waves");

__device__ __forceinline__ void od_glds4(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void od_glds16(uint32_t byte_off, const void *base, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void od_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void od_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): this wave's LDS reads of the slot it gives up have returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int MODE>
__device__ __forceinline__ void os_head_dma_body(int j, int s, int grp, float *__restrict__ os_lds, const float *__restrict__ pool, int C,
                                                 const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                 const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                 const uint32_t *__restrict__ moff, int kmax, float *__restrict__ dst, int member_cap,
                                                 float *__restrict__ head_state, KsPred pred) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = aoc_lane();
    if (wave > OD_NPROD) return;                               // the launch is 256 wide for the chunk-sum role
    if (j >= seg_k[s]) return;
    const int oc = s * kmax + j;
    const int cnt_all = counts[oc];
    const int f0 = grp * OS_FP * 4;
    const int nfeat = min(OS_FP * 4, C - f0);
    float *out = ((MODE == 0) ? dst + (size_t)oc * C : dst + (((size_t)s * 2 + 1) * kmax + j) * C) + f0;
    if (cnt_all == 0) {
        if (MODE == 1 && (int)threadIdx.x < nfeat) out[threadIdx.x] = 0.0f;
        return;
    }
    const bool head_only = member_cap > 0 && cnt_all > member_cap;
    const int cnt = head_only ? member_cap : cnt_all;
    const int nb = (cnt + OD_BATCH - 1) / OD_BATCH;
    auto wrap = [](int x) { return x >= OD_NB ? x - OD_NB : x; };

    if (wave == 0) {
        // ---- consumer: the sequential float32 sum, lanes = features
        const bool active = lane < nfeat;
        const float *col = os_lds + (active ? lane : 0);
        float sum = 0.0f;
        int slot = 0;
        for (int b = 0; b < nb; ++b) {
            od_barrier();                                      // batch b has landed (the producers waited for it before arriving)
            const float *t = col + slot * (OD_BATCH * OD_LD);
            // groups of OD_G members, the next group's LDS reads issued in front of this group's additions (two register sets): the
            // adding chain never waits for an LDS round trip (left to itself hipcc issues 18 reads, waits, adds 18: 9 cycles per member)
            constexpr int OD_G = 32, NG = OD_BATCH / OD_G;
            const int rem = (b + 1 < nb) ? OD_BATCH : cnt - b * OD_BATCH;
            float xa[OD_G], xb[OD_G];
#pragma unroll
            for (int u = 0; u < OD_G; ++u) xa[u] = t[u * OD_LD];
#pragma unroll
            for (int g = 0; g < NG; g += 2) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int u = 0; u < OD_G; ++u) xb[u] = t[((g + 1) * OD_G + u) * OD_LD];
                }
                if (rem >= (g + 1) * OD_G) {
#pragma unroll
                    for (int u = 0; u < OD_G; ++u) sum = sum + xa[u];
                } else {
#pragma unroll
                    for (int u = 0; u < OD_G; ++u) sum = sum + (g * OD_G + u < rem ? xa[u] : 0.0f);
                }
                if (g + 2 < NG) {
#pragma unroll
                    for (int u = 0; u < OD_G; ++u) xa[u] = t[((g + 2) * OD_G + u) * OD_LD];
                }
                if (g + 1 < NG) {
                    if (rem >= (g + 2) * OD_G) {
#pragma unroll
                        for (int u = 0; u < OD_G; ++u) sum = sum + xb[u];
                    } else {
#pragma unroll
                        for (int u = 0; u < OD_G; ++u) sum = sum + ((g + 1) * OD_G + u < rem ? xb[u] : 0.0f);
                    }
                }
            }
            slot = wrap(slot + 1);
        }
        if (active) {
            if (head_only) head_state[(size_t)oc * C + f0 + lane] = sum;
            else out[lane] = sum / (float)cnt;
            if (!head_only && pred.e0) {
                // no tail this time: should the cluster grow one, its chunks start from the binade of this sum (KsPred)
                const size_t idx = (size_t)oc * C + f0 + lane;
                const KsBinade bb = ks_binade(sum);
                pred.e0[idx] = (int8_t)(bb.ok ? (int)((__float_as_uint(sum) >> 23) & 0xff) - 127 : KC_UNSAFE);
                pred.cidx[idx * KP_CROSS] = (uint16_t)0xFFFF;
            }
        }
        return;
    }

    // ---- producers: p = 0 / 1 owns members [64 p, 64 p + 64) of every batch
    const int p = wave - 1;
    const uint32_t *list = moff + seg_off[s] + cbase[oc];
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(os_lds);
    const uint32_t offs_base = lds_base + OD_RING_FLOATS * 4 + (uint32_t)p * (OD_HALF * 4);
    const uint32_t *offs = reinterpret_cast<const uint32_t *>(os_lds + OD_RING_FLOATS) + p * OD_HALF;
    // transfer plan of this lane: slot i * 64 + lane of the producer's half = (member, piece); pieces past the row's end (last group)
    // repeat the group's first
    const int npiece = min(OS_FP, (C >> 2) - grp * OS_FP);
    int prow[OS_FP];
    uint32_t pbyte[OS_FP];
#pragma unroll
    for (int i = 0; i < OS_FP; ++i) {
        const int idx = i * 64 + lane;
        prow[i] = idx / OS_FP;
        const int piece = idx - prow[i] * OS_FP;
        pbyte[i] = (uint32_t)(grp * OS_FP + (piece < npiece ? piece : 0)) * 16u;
    }
    auto issue_offs = [&](int b, int slot) {                   // this producer's member offsets of batch b -> offsets slot
        od_glds4(list + min(b * OD_BATCH + p * OD_HALF + lane, cnt - 1), offs_base + (uint32_t)slot * (OD_BATCH * 4));
    };
    auto issue_rows = [&](int slot) {                          // its rows of the batch whose offsets sit in `slot` -> ring slot `slot`
        const uint32_t *o = offs + slot * OD_BATCH;
        uint32_t src[OS_FP];
#pragma unroll
        for (int i = 0; i < OS_FP; ++i) src[i] = o[prow[i]] + pbyte[i];
        const uint32_t d = lds_base + (uint32_t)slot * OD_BATCH_BYTES + (uint32_t)p * (OD_HALF * OD_LD * 4);
#pragma unroll
        for (int i = 0; i < OS_FP; ++i) od_glds16(src[i], pool, d + (uint32_t)i * 1024u);
    };
    // prologue: offsets of batches 0 .. D-1 (drained once), then D steps' worth of transfers in the steady-state order
#pragma unroll
    for (int x = 0; x < OD_DEPTH; ++x) issue_offs(x, x);
    od_wait_vm<0>();
#pragma unroll
    for (int i = 0; i < OD_DEPTH; ++i) {
        issue_offs(OD_DEPTH + i, wrap(OD_DEPTH + i));
        issue_rows(i);
    }
    int slot = 0;                                              // ring slot of batch b
    for (int b = 0; b < nb; ++b) {
        od_wait_vm<(OD_DEPTH - 1) * OD_OPS>();                  // this producer's rows of batch b and offsets of batch b + D have landed
        od_barrier();
        const int slot_d = wrap(slot + OD_DEPTH);              // slot of batch b + D (= of batch b - 1: consumed)
        issue_offs(b + 2 * OD_DEPTH, wrap(slot_d + OD_DEPTH)); // = slot of batch b + D - 1, whose offsets the last step used
        issue_rows(slot_d);
        slot = wrap(slot + 1);
    }
    od_wait_vm<0>();                                           // nothing may still be travelling into this workgroup's LDS when it ends
}
template <int MODE>
__global__ __launch_bounds__((OS_NPROD + 1) * 64) void km_ordered_sum_kernel(const float *__restrict__ pool, uint32_t pool_bytes, int C,
                                                                              const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                                              const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                                              const uint32_t *__restrict__ moff, int kmax, float *__restrict__ dst,
                                                                              int member_cap, float *__restrict__ head_state) {
    __shared__ __attribute__((aligned(16))) float os_lds[2 * OS_BATCH * OS_LD];
    os_ordered_sum_body<MODE>(blockIdx.x, blockIdx.y, blockIdx.z, os_lds, pool, pool_bytes, C, seg_off, seg_k, counts, cbase, moff, kmax, dst, member_cap, head_state);
}
// The literal heads and the any-order sums of the tail chunks only depend on the member lists, not on each other: ONE launch, the
// first kmax * n_seg * groups workgroups take the heads (the long ones, dispatched first), the others one tail chunk each (on the
// first four of their eight waves; the LDS allocation of the head role is reused).
template <int MODE, bool DMA>
__global__ __launch_bounds__(DMA ? 256 : (OS_NPROD + 1) * 64) void km_heads_chunk_sums_kernel(const float *__restrict__ pool, uint32_t pool_bytes, int C,
                                                                                   const int32_t *__restrict__ seg_off, const int32_t *__restrict__ seg_k,
                                                                                   const int32_t *__restrict__ counts, const int32_t *__restrict__ cbase,
                                                                                   const uint32_t *__restrict__ moff, int kmax, int n_seg, float *__restrict__ dst,
                                                                                   int member_cap, float *__restrict__ head_state,
                                                                                   const int32_t *__restrict__ owner_cluster, const int32_t *__restrict__ owner_local,
                                                                                   float *__restrict__ csum, int start_chunk, int xcd_aware,
                                                                                   KsPred pred, int spec_fold, int8_t *__restrict__ cexp,
                                                                                   int32_t *__restrict__ cinc0, int32_t *__restrict__ cinc1,
                                                                                   const int32_t *__restrict__ cchunk, int n_chunks_grid, int n_fold_groups) {
    __shared__ __attribute__((aligned(16))) float os_lds[DMA ? (OD_LDS_FLOATS > 2 * OS_BATCH * OS_LD ? OD_LDS_FLOATS : 2 * OS_BATCH * OS_LD) : 2 * OS_BATCH * OS_LD];
    static_assert(sizeof(float) * 2 * OS_BATCH * OS_LD >= sizeof(uint32_t) * KS_CHUNK + sizeof(float4) * 256, "the chunk role's buffers fit the head role's");
    static_assert(2 * OS_BATCH * OS_LD >= KC_FOLD_LDS_FLOATS, "the fold role's buffers fit the head role's");
    const int n_head = kmax * n_seg * os_groups_dev(C);
    const int b = blockIdx.x;
    if (b < n_head) {
        // XCD-aware ids: workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  The feature groups of ONE cluster read
        // neighbouring 112-byte pieces of the same member rows; their ids differ by 8, so they run on the same XCD at about the same time
        // and find each other's lines in its L2 (bench: cfg2 +1.3 %, cfg3 +6 %; FETCH_SIZE of this kernel alone only drops 3 % -- at the
        // fabric it was already near one pass over the rows per replica, profiles/r03_pmc_kmeans_R6_F3.txt).  Blocks of 8 clusters x
        // groups; the last block may hold fewer clusters.
        const int n_cl = kmax * n_seg, groups = os_groups_dev(C);
        const int blk = b / (8 * groups), rem = b - blk * (8 * groups);
        const int pc = min(8, n_cl - blk * 8);
        int grp = rem / pc, c = blk * 8 + rem - grp * pc;
        if (!xcd_aware) { c = b % n_cl; grp = b / n_cl; }       // developer switch AOC_KM_XCD=0: the group-major order of before
        const int j = c % kmax, s = c / kmax;
        if (DMA) os_head_dma_body<MODE>(j, s, grp, os_lds, pool, C, seg_off, seg_k, counts, cbase, moff, kmax, dst, member_cap, head_state, pred);
        else os_ordered_sum_body<MODE>(j, s, grp, os_lds, pool, pool_bytes, C, seg_off, seg_k, counts, cbase, moff, kmax, dst, member_cap, head_state);
        return;
    }
    if (threadIdx.x >= 256) return;
    if (DMA && spec_fold) {
        // round 5: the tail chunks' integer folds in their PREDICTED binades (KsPred: the previous Lloyd iteration's) ride in this launch --
        // no any-order chunk sums, no separate fold launch; the stitch verifies every summary as before
        static_assert(!DMA || OD_LDS_FLOATS >= KC_WIDE_LDS_FLOATS, "the wide fold's tile fits the heads' ring");
        kc_chunk_fold_wide_body(b - n_head, os_lds, pool, C, seg_off, counts, cbase, moff, kmax, owner_cluster, owner_local, cexp, cinc0, cinc1, start_chunk,
                                n_chunks_grid, n_fold_groups, xcd_aware, pred);
        return;
    }
    uint32_t *loffs = reinterpret_cast<uint32_t *>(os_lds);
    float4 *part = reinterpret_cast<float4 *>(os_lds + KS_CHUNK);
    kc_chunk_sum_body(b - n_head, loffs, part, pool, C, seg_off, counts, cbase, moff, kmax, owner_cluster, owner_local, csum, start_chunk);
}

// proxy set 0 = centroids (copied) and the squared norms of both sets; one wave per (slot j, segment s)
__global__ __launch_bounds__(64) void km_proxy_finish_kernel(const float *__restrict__ centroids, const int32_t *__restrict__ seg_k,
                                                              const int32_t *__restrict__ counts, int kmax, int C,
                                                              float *__restrict__ proxies, float *__restrict__ proxy_sqnorm) {
    const int s = blockIdx.y, j = blockIdx.x, lane = threadIdx.x;
    const int k = seg_k[s];
    float *p0 = proxies + (((size_t)s * 2 + 0) * kmax + j) * C;
    float *p1 = proxies + (((size_t)s * 2 + 1) * kmax + j) * C;
    if (j >= k) {
        for (int t = lane; t < C; t += 64) { p0[t] = 0.0f; p1[t] = 0.0f; }
        if (lane == 0) {
            proxy_sqnorm[((size_t)s * 2 + 0) * kmax + j] = INFINITY;
            proxy_sqnorm[((size_t)s * 2 + 1) * kmax + j] = INFINITY;
        }
        return;
    }
    const float *c = centroids + ((size_t)s * kmax + j) * C;
    float n0 = 0.0f, n1 = 0.0f;
    for (int t = lane; t < C; t += 64) {
        const float cv = c[t];
        p0[t] = cv;
        n0 += cv * cv;
        const float av = p1[t];
        n1 += av * av;
    }
    n0 = aoc_wave_sum(n0);
    n1 = aoc_wave_sum(n1);
    if (lane == 0) {
        proxy_sqnorm[((size_t)s * 2 + 0) * kmax + j] = n0;
        proxy_sqnorm[((size_t)s * 2 + 1) * kmax + j] = (counts[s * kmax + j] > 0) ? n1 : INFINITY;   // np.unique drops empty clusters
    }
}

struct KsWorkspace {
    float *rownorm;
    uint16_t *rank16;
    int32_t *hist, *counts, *cbase;
    uint32_t *moff;
    int nb_max;
    // chunk summaries
    int nch_cap;
    int32_t *cchunk, *owner_cluster, *owner_local, *cinc0, *cinc1;
    float *csum, *head;
    int8_t *cexp;
    uint32_t *cflag;          // per (chunk, feature group): local sums published (km_chunk_scanfold_kernel)
    int seg_chunks_max;       // no segment (hence no cluster) has more chunks than this
    KsPred pred;              // binades of the tail chunks as the last stitch left them (round 5)
};
inline size_t ks_pred_bytes(int n_seg, int kmax) {
    const size_t n = (size_t)n_seg * kmax * (AOC_MAX_CHANNELS / 2);
    return aoc_align_up(n, 256) + aoc_align_up(n * KP_CROSS * 2, 256) + aoc_align_up(n * KP_CROSS, 256);
}
inline int ks_chunk_capacity(int64_t cap, int n_seg, int kmax) { return (int)(cap / KS_CHUNK) + n_seg * (kmax + 1) + 2; }
inline size_t ks_workspace_bytes(int64_t cap, int n_seg, int kmax) {
    const size_t nb = (size_t)(cap + 255) / 256 + 1;
    const size_t nch = (size_t)ks_chunk_capacity(cap, n_seg, kmax);
    return aoc_align_up((size_t)cap * 4, 256) + aoc_align_up((size_t)cap * 2, 256) + aoc_align_up((size_t)n_seg * nb * kmax * 4, 256) +
           3 * aoc_align_up((size_t)n_seg * kmax * 4, 256) + aoc_align_up(((size_t)cap + 64) * 4, 256) + 2 * aoc_align_up(nch * 4, 256) +
           3 * aoc_align_up(nch * AOC_MAX_CHANNELS / 2 * 4, 256) + aoc_align_up(nch * AOC_MAX_CHANNELS / 2, 256) +
           aoc_align_up((size_t)n_seg * kmax * (AOC_MAX_CHANNELS / 2) * 4, 256) + aoc_align_up(nch * 8 * 4, 256) + ks_pred_bytes(n_seg, kmax);
}
inline KsWorkspace ks_carve(void *workspace, int64_t cap, int n_seg, int kmax, int64_t seg_bound = 0) {
    KsWorkspace w;
    char *p = static_cast<char *>(workspace);
    const size_t nb = (size_t)(cap + 255) / 256 + 1;
    w.nb_max = (int)((seg_bound > 0 && seg_bound < cap) ? (size_t)(seg_bound + 255) / 256 + 1 : nb);   // stride of the per-block histograms
    w.seg_chunks_max = (int)(((seg_bound > 0 && seg_bound < cap) ? seg_bound : cap) / KS_CHUNK + 1);
    w.rownorm = reinterpret_cast<float *>(p); p += aoc_align_up((size_t)cap * 4, 256);
    w.rank16 = reinterpret_cast<uint16_t *>(p); p += aoc_align_up((size_t)cap * 2, 256);
    w.hist = reinterpret_cast<int32_t *>(p); p += aoc_align_up((size_t)n_seg * nb * kmax * 4, 256);
    w.counts = reinterpret_cast<int32_t *>(p); p += aoc_align_up((size_t)n_seg * kmax * 4, 256);
    w.cbase = reinterpret_cast<int32_t *>(p); p += aoc_align_up((size_t)n_seg * kmax * 4, 256);
    w.cchunk = reinterpret_cast<int32_t *>(p); p += aoc_align_up((size_t)n_seg * kmax * 4, 256);
    w.moff = reinterpret_cast<uint32_t *>(p); p += aoc_align_up(((size_t)cap + 64) * 4, 256);
    const size_t nch = (size_t)ks_chunk_capacity(cap, n_seg, kmax);
    w.nch_cap = (int)nch;
    w.owner_cluster = reinterpret_cast<int32_t *>(p); p += aoc_align_up(nch * 4, 256);
    w.owner_local = reinterpret_cast<int32_t *>(p); p += aoc_align_up(nch * 4, 256);
    w.csum = reinterpret_cast<float *>(p); p += aoc_align_up(nch * AOC_MAX_CHANNELS / 2 * 4, 256);
    w.cinc0 = reinterpret_cast<int32_t *>(p); p += aoc_align_up(nch * AOC_MAX_CHANNELS / 2 * 4, 256);
    w.cinc1 = reinterpret_cast<int32_t *>(p); p += aoc_align_up(nch * AOC_MAX_CHANNELS / 2 * 4, 256);
    w.cexp = reinterpret_cast<int8_t *>(p); p += aoc_align_up(nch * AOC_MAX_CHANNELS / 2, 256);
    w.head = reinterpret_cast<float *>(p); p += aoc_align_up((size_t)n_seg * kmax * (AOC_MAX_CHANNELS / 2) * 4, 256);
    w.cflag = reinterpret_cast<uint32_t *>(p); p += aoc_align_up(nch * 8 * 4, 256);
    {
        const size_t n = (size_t)n_seg * kmax * (AOC_MAX_CHANNELS / 2);
        w.pred.e0 = reinterpret_cast<int8_t *>(p); p += aoc_align_up(n, 256);
        w.pred.cidx = reinterpret_cast<uint16_t *>(p); p += aoc_align_up(n * KP_CROSS * 2, 256);
        w.pred.ce = reinterpret_cast<int8_t *>(p);
    }
    return w;
}

// Ordered per-cluster sums of the member lists in ws.moff -> dst (MODE 0 centroids / MODE 1 proxy set 1).
//   "hybrid" (default): the first KS_HEAD_CHUNKS chunks of every cluster -- where the running sum crosses a binade at
//             every doubling -- are summed literally, lanes = features (km_ordered_sum_kernel); larger clusters continue
//             with the chunk-parallel integer folds and the serial stitch (their crossings are rare from there on);
//   "ordered": literal sums only;   "scan": chunk-parallel pipeline only.   (AOC_KM_SUM, developer switch.)
// 20 chunks = 10240 members (12 until the sum kernels' XCD-aware ids; after them, three runs each at 12 / 20 / 28 chunks: cfg2 360.3 / 359.7 / 356.0,
// cfg3 208.5 / 212.6 / 211.1, closed evaluation loop 246 / 251 / 239).  Alone, a chain is fastest at 5-6 chunks (sweep 1 .. 8 at R = 6: 3.13, 3.03, 2.90, 2.89, 2.76, 2.78, 2.81,
// 2.84 ms per chain; 3.12 at 16): the heads share a launch with the tail's chunk sums and are off the critical path up to about
// there.  In the bench, where the chains share the GPU with the other streams, what counts is the work a chain puts on the CUs, and the
// literal heads (one adding wave per workgroup, a few cycles per member) are the cheapest way to sum a member: frames/s at
// 2 / 3 / 4 / 6 / 8 / 10 / 12 / 16 / 24 / 32 / 64 chunks: cfg2 287 / 305 / 310 / 326 / 327 / 337 / 334 / 329 / 323 / 304 / 271, cfg3
// 142 / 165 / 170 / 182 / 185 / 191 / 192 / 191 / 192 / 190 / 184.  AOC_KM_HEAD_CHUNKS: developer switch.  (A [4 members][feature] LDS layout
// with one ds_read_b128 per four members shortens the adding wave's chain -- 3.0 -> 2.9 ms alone at 12 chunks -- but costs the seven
// producer waves four ds_write_b32 per piece instead of one ds_write_b128: 1 % SLOWER in the bench, three runs each; not kept.)
inline int km_xcd_aware() {
    static const int on = AOC_DEV_ENV_INT("AOC_KM_XCD", 1) != 0;       // developer switch (counter comparisons)
    return on;
}
static const int KS_HEAD_CHUNKS = AOC_DEV_ENV_INT("AOC_KM_HEAD_CHUNKS", 20) > 0 ? AOC_DEV_ENV_INT("AOC_KM_HEAD_CHUNKS", 20) : 20;
constexpr int KC_INLINE_PREDICT_CHUNKS = 800;   // 409 600 rows per segment
inline int ks_sum_mode() {
    static const int mode = [] {
        const char *e = AOC_DEV_ENV("AOC_KM_SUM");
        if (e && strcmp(e, "scan") == 0) return 0;
        if (e && strcmp(e, "ordered") == 0) return 1;
        return 2;
    }();
    return mode;
}
// spec: MODE 0, Lloyd iteration >= 1 -- the tail chunks are folded in the binades the previous iteration's stitch recorded (ws.pred), inside the
// heads launch: FOUR launches per iteration (assignment, scan + scatter, heads + folds, stitch) and one pass over the tail rows instead of two.
// Iteration 0 and the proxy sums (MODE 1: other rows, another workspace) have no previous stitch and take the any-order chunk sums + fold launch.
template <int MODE>
inline void ks_launch_sums(hipStream_t st, const float *pool, uint32_t pool_bytes, int C, const int32_t *seg_offsets, const int32_t *seg_k,
                           const int32_t *counts, const KsWorkspace &ws, int kmax, int n_seg, float *dst, bool spec = false) {
    const KsPred no_pred = {nullptr, nullptr, nullptr};
    const KsPred pred = (MODE == 0) ? ws.pred : no_pred;             // MODE 0: every stitch (and every head without a tail) records for the next iteration
    const int n_fold_groups = (C + KC_FG - 1) / KC_FG;
    // Where the folds run: INSIDE the heads launch (4 launches per iteration; smode 1).  A heads workgroup owns 43 KB of LDS, so three fit a CU, a
    // cluster that fills its literal head keeps four of them for ~60 us, and the folds of the launch compete for the same slots; with the
    // whole-chunk staging of kc_chunk_fold_wide_body that still beats heads + folds as two launches (smode 2, development switch) and the
    // five-launch iteration of round 4 (smode 0) alone at every size measured (profiles/r05_kmeans_spec_fold_ab.txt: cfg2 R = 6, three frames per
    // chain 3.7-3.8 -> 3.1 ms; cfg3 R = 2 4.9 -> 4.1; cfg4 R = 3 7.4 -> 6.7) except cfg3 at R = 6 on one of two boxes (10.7 -> 11.1 ms there, 12.3 -> 11.2
    // on the other: the K = 8 level has ~200 clusters that fill their head, more head workgroups than slots).
    int smode = spec ? 1 : 0;
#ifdef AOC_DEV
    static const int spec_env = AOC_DEV_ENV_INT("AOC_KM_SPEC", -1);      // developer switch: 0 = the five-launch iteration of round 4, 1 = folds in the heads launch, 2 = folds apart
    if (spec && spec_env >= 0) smode = spec_env;
    if (!(ks_sum_mode() == 2 && AOC_DEV_ENV_INT("AOC_KM_FUSED", 0) != 1 && AOC_DEV_ENV_INT("AOC_KM_HEADS_DMA", 1) != 0 &&
          !(AOC_DEV_ENV("AOC_KM_HEADS") && strcmp(AOC_DEV_ENV("AOC_KM_HEADS"), "kernel") == 0)))
        smode = 0;
#endif
    if (!(MODE == 0 && C <= KC_FG * 8)) smode = 0;
    if (smode != 0) {
        const int start = KS_HEAD_CHUNKS;
        const unsigned n_head_wg = (unsigned)(kmax * n_seg * os_groups(C));
        hipLaunchKernelGGL((km_heads_chunk_sums_kernel<MODE, true>), dim3(n_head_wg + (smode == 1 ? (unsigned)ws.nch_cap * n_fold_groups : 0u)), dim3(256), 0, st, pool,
                           pool_bytes, C, seg_offsets, seg_k, counts, ws.cbase, ws.moff, kmax, n_seg, dst, KS_HEAD_CHUNKS * KS_CHUNK, ws.head, ws.owner_cluster,
                           ws.owner_local, ws.csum, start, km_xcd_aware(), pred, 1, ws.cexp, ws.cinc0, ws.cinc1, ws.cchunk, ws.nch_cap, n_fold_groups);
        if (smode == 2)
            hipLaunchKernelGGL(km_chunk_fold_kernel, dim3((unsigned)ws.nch_cap * n_fold_groups), dim3(256), 0, st, pool, C, seg_offsets, counts, ws.cbase, ws.moff,
                               kmax, ws.owner_cluster, ws.owner_local, ws.cexp, ws.cinc0, ws.cinc1, start, (const float *)nullptr, ws.cchunk, ws.head, ws.nch_cap,
                               n_fold_groups, km_xcd_aware(), pred);
        hipLaunchKernelGGL((km_sum_scan_kernel<MODE, 1>), dim3((unsigned)((C + KS_SCAN_WAVES - 1) / KS_SCAN_WAVES) * kmax * n_seg), dim3(KS_SCAN_WAVES * 64), 0, st, pool, pool_bytes, C, seg_offsets, seg_k, counts, ws.cbase,
                           ws.moff, kmax, dst, ws.cchunk, ws.cexp, ws.cinc0, ws.cinc1, start, ws.head, n_seg, km_xcd_aware(), pred);
        return;
    }
#ifndef AOC_DEV
    // release build: literal heads by LDS-DMA + any-order chunk sums in one launch, then the fold (binade prediction inside it while no cluster
    // can have more than KC_INLINE_PREDICT_CHUNKS chunks) -- the alternatives below only exist in the development build
    const int start = KS_HEAD_CHUNKS;
    hipLaunchKernelGGL((km_heads_chunk_sums_kernel<MODE, true>), dim3(kmax * n_seg * os_groups(C) + ws.nch_cap), dim3(256), 0, st, pool, pool_bytes, C,
                       seg_offsets, seg_k, counts, ws.cbase, ws.moff, kmax, n_seg, dst, KS_HEAD_CHUNKS * KS_CHUNK, ws.head, ws.owner_cluster, ws.owner_local,
                       ws.csum, start, 1, pred, 0, ws.cexp, ws.cinc0, ws.cinc1, ws.cchunk, ws.nch_cap, n_fold_groups);
    {
        const bool inline_predict = ws.seg_chunks_max <= KC_INLINE_PREDICT_CHUNKS;
        if (!inline_predict)
            hipLaunchKernelGGL(km_chunk_predict_kernel, dim3(kmax, n_seg), dim3(128), 0, st, seg_k, counts, ws.cchunk, ws.csum, kmax, C, ws.cexp, start, ws.head);
        hipLaunchKernelGGL(km_chunk_fold_kernel, dim3((unsigned)ws.nch_cap * ((C + KC_FG - 1) / KC_FG)), dim3(256), 0, st, pool, C, seg_offsets, counts, ws.cbase, ws.moff,
                           kmax, ws.owner_cluster, ws.owner_local, ws.cexp, ws.cinc0, ws.cinc1, start, inline_predict ? ws.csum : (const float *)nullptr,
                           ws.cchunk, ws.head, ws.nch_cap, (C + KC_FG - 1) / KC_FG, 1, no_pred);
    }
#else
    const int mode = ks_sum_mode();
    const int start = (mode == 2) ? KS_HEAD_CHUNKS : 0;
    static const bool fused = AOC_DEV_ENV_INT("AOC_KM_FUSED", 0) == 1;
    static const bool split_heads = AOC_DEV_ENV("AOC_KM_HEADS") && strcmp(AOC_DEV_ENV("AOC_KM_HEADS"), "kernel") == 0;   // developer switch: heads and chunk sums as two launches
    const bool merged = (mode == 2) && !(fused && C <= KC_FG * 8) && !split_heads;
    if (merged) {
        // literal heads on one wave per (cluster, feature group) with LDS-DMA row transfers (os_head_dma_body); developer switch
        // AOC_KM_HEADS_DMA=0: the eight-wave producer / consumer version of rounds 2-3
        static const bool dma = AOC_DEV_ENV_INT("AOC_KM_HEADS_DMA", 1) != 0;
        if (dma)
            hipLaunchKernelGGL((km_heads_chunk_sums_kernel<MODE, true>), dim3(kmax * n_seg * os_groups(C) + ws.nch_cap), dim3(256), 0, st, pool, pool_bytes, C,
                               seg_offsets, seg_k, counts, ws.cbase, ws.moff, kmax, n_seg, dst, KS_HEAD_CHUNKS * KS_CHUNK, ws.head, ws.owner_cluster, ws.owner_local,
                               ws.csum, start, km_xcd_aware(), pred, 0, ws.cexp, ws.cinc0, ws.cinc1, ws.cchunk, ws.nch_cap, n_fold_groups);
        else
            hipLaunchKernelGGL((km_heads_chunk_sums_kernel<MODE, false>), dim3(kmax * n_seg * os_groups(C) + ws.nch_cap), dim3((OS_NPROD + 1) * 64), 0, st, pool, pool_bytes, C,
                               seg_offsets, seg_k, counts, ws.cbase, ws.moff, kmax, n_seg, dst, KS_HEAD_CHUNKS * KS_CHUNK, ws.head, ws.owner_cluster, ws.owner_local,
                               ws.csum, start, km_xcd_aware(), pred, 0, ws.cexp, ws.cinc0, ws.cinc1, ws.cchunk, ws.nch_cap, n_fold_groups);
    } else if (mode != 0) {
        const int cap = (mode == 2) ? KS_HEAD_CHUNKS * KS_CHUNK : 0;
        hipLaunchKernelGGL(km_ordered_sum_kernel<MODE>, dim3(kmax, n_seg, os_groups(C)), dim3((OS_NPROD + 1) * 64), 0, st, pool, pool_bytes, C, seg_offsets,
                           seg_k, counts, ws.cbase, ws.moff, kmax, dst, cap, ws.head);
        if (mode == 1) return;
    }
    // developer switch AOC_KM_FUSED=1: km_chunk_scanfold_kernel instead of chunk sum -> predict -> fold.  Bit-identical (all k-means tests
    // pass in both modes); measured 3.5 vs 3.9 ms per 20-iteration chain at R = 6 with one frame per chain, but 6.6 vs 6.4 ms with three
    // frames per chain and 5.7 vs 5.6 at R = 12: workgroups that wait for their predecessors' sums hold CU slots, so the default stays
    // the three-kernel tail.
    if (fused && C <= KC_FG * 8) {
        hipLaunchKernelGGL(km_chunk_scanfold_kernel, dim3(ws.nch_cap, (C + KC_FG - 1) / KC_FG), dim3(256), 0, st, pool, C, seg_offsets, counts, ws.cbase,
                           ws.moff, kmax, ws.owner_cluster, ws.owner_local, ws.cchunk, ws.head, ws.csum, ws.cflag, ws.cexp, ws.cinc0, ws.cinc1, start);
    } else {
        if (!merged)
            hipLaunchKernelGGL(km_chunk_sum_kernel, dim3(ws.nch_cap), dim3(256), 0, st, pool, C, seg_offsets, counts, ws.cbase, ws.moff, kmax, ws.owner_cluster,
                               ws.owner_local, ws.csum, start);
        // the binade prediction runs inside the fold kernel (every workgroup re-adds its cluster's earlier chunk sums: quadratic in the
        // chunks of a cluster, so only while a cluster cannot have more than KC_INLINE_PREDICT_CHUNKS); AOC_KM_PREDICT=kernel: separate launch
        static const bool sep_predict = AOC_DEV_ENV("AOC_KM_PREDICT") && strcmp(AOC_DEV_ENV("AOC_KM_PREDICT"), "kernel") == 0;
        const bool inline_predict = !sep_predict && ws.seg_chunks_max <= KC_INLINE_PREDICT_CHUNKS;
        if (!inline_predict)
            hipLaunchKernelGGL(km_chunk_predict_kernel, dim3(kmax, n_seg), dim3(128), 0, st, seg_k, counts, ws.cchunk, ws.csum, kmax, C, ws.cexp, start, ws.head);
        hipLaunchKernelGGL(km_chunk_fold_kernel, dim3((unsigned)ws.nch_cap * ((C + KC_FG - 1) / KC_FG)), dim3(256), 0, st, pool, C, seg_offsets, counts, ws.cbase, ws.moff,
                           kmax, ws.owner_cluster, ws.owner_local, ws.cexp, ws.cinc0, ws.cinc1, start, inline_predict ? ws.csum : (const float *)nullptr,
                           ws.cchunk, ws.head, ws.nch_cap, (C + KC_FG - 1) / KC_FG, km_xcd_aware(), no_pred);
    }
#endif
    static const int nf = AOC_DEV_ENV_INT("AOC_KS_NF", 1);       // features per stitch wave (developer switch)
#define AOC_KSS(NF) hipLaunchKernelGGL((km_sum_scan_kernel<MODE, NF>), dim3((unsigned)((C / NF + KS_SCAN_WAVES - 1) / KS_SCAN_WAVES) * kmax * n_seg), dim3(KS_SCAN_WAVES * 64), 0, st, pool, pool_bytes, C, seg_offsets, seg_k, \
                                       counts, ws.cbase, ws.moff, kmax, dst, ws.cchunk, ws.cexp, ws.cinc0, ws.cinc1, start, ws.head, n_seg, km_xcd_aware(), pred)
#ifdef AOC_DEV
    if (nf == 4) AOC_KSS(4); else if (nf == 2) AOC_KSS(2); else AOC_KSS(1);
#else
    (void)nf;
    AOC_KSS(1);
#endif
#undef AOC_KSS
}

#ifndef AOC_KM_REP64
#define AOC_KM_REP64 0
#endif
// LDS a workgroup of the replica-fused assignment may use: half a CU's (two workgroups per CU; three at K <= 16) -- or, for K = 64 (four cluster tiles: two
// 30 KB code books do not fit half a CU), a whole CU's with AOC_KM_REP64 (one workgroup of four waves per CU then stages the rows once for up to four code books)
inline size_t km_rep_lds_budget(int kt) {
    static const bool rep64 = AOC_DEV_ENV_INT("AOC_KM_REP64", AOC_KM_REP64) != 0;
    return (kt == 4 && rep64) ? (size_t)150 * 1024 : (size_t)78 * 1024;
}

inline int km_assign_grid_cap() {
    static const int cap = AOC_DEV_ENV_INT("AOC_KM_ASSIGN_GRID", 512);      // developer switch (measured: 128 .. 512 within 3 %)
    return cap > 0 ? cap : 512;
}

inline int label_blocks(int64_t n) { return (int)((n + LP_BLOCK - 1) / LP_BLOCK); }

}  // namespace

extern "C" {

const char *aoc_version(void) { return "aoc_hip 0.2 (gfx950)"; }

size_t aoc_label_prep_workspace_bytes(int64_t n, int n_obj) {
    if (n < 0 || n_obj < 1) return 0;
    return aoc_align_up((size_t)(n_obj + 1) * (size_t)(label_blocks(n) > 0 ? label_blocks(n) : 1) * sizeof(int32_t), 256);
}

int aoc_label_prep(const float *labels, int64_t n, int n_obj, uint32_t *right_bits, uint32_t *wrong_bits,
                   int32_t *fg_rows, int32_t *obj_rows, int32_t *counts, int32_t *obj_offsets,
                   void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!labels || !right_bits || !wrong_bits || !fg_rows || !obj_rows || !counts || !obj_offsets || !workspace)
        return AOC_ERR_INVALID_ARG;
    if (n < 1 || n >= (1ll << 31) || n_obj < 1) return AOC_ERR_INVALID_ARG;
    if (n_obj > AOC_MAX_OBJECTS) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_label_prep_workspace_bytes(n, n_obj)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    const int nb = label_blocks(n);
    int32_t *bc = static_cast<int32_t *>(workspace);
    hipLaunchKernelGGL(label_flags_kernel, dim3(nb), dim3(LP_BLOCK), 0, st, labels, (int)n, n_obj, right_bits, wrong_bits, bc, nb);
    hipLaunchKernelGGL(label_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nb, n_obj, counts, obj_offsets);
    hipLaunchKernelGGL(label_scatter_kernel, dim3(nb), dim3(LP_BLOCK), 0, st, right_bits, (int)n, n_obj, bc, nb, obj_offsets, fg_rows, obj_rows);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_label_bits(const float *labels, int64_t n, int n_obj, uint32_t *right_bits, uint32_t *wrong_bits, aoc_stream_t stream) {
    if (!labels || !right_bits || n < 1 || n_obj < 1) return AOC_ERR_INVALID_ARG;
    if (n_obj > AOC_MAX_OBJECTS) return AOC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(label_bits_kernel, dim3((unsigned)((n + LP_BLOCK - 1) / LP_BLOCK)), dim3(LP_BLOCK), 0, aoc_hip_stream(stream),
                       labels, n, n_obj, right_bits, wrong_bits);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_kmeans_plan(const int32_t *counts, int n_seg, int cluster_num, int32_t *seg_k, aoc_stream_t stream) {
    if (!counts || !seg_k || n_seg < 1 || cluster_num < 0) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_plan_kernel, dim3(1), dim3(64), 0, aoc_hip_stream(stream), counts, n_seg, cluster_num, seg_k);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_kmeans_workspace_bytes(int64_t rows_capacity, int n_seg, int kmax, int C) {
    (void)C;
    if (rows_capacity < 0 || n_seg < 1 || kmax < 1) return 0;
    return aoc_align_up((size_t)n_seg * kmax * sizeof(float), 256) + aoc_align_up(ks_workspace_bytes(rows_capacity, n_seg, kmax), 256);
}

int aoc_kmeans_segmented(const float *pool, int C, const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                         const int32_t *init_rows, int n_seg, int kmax, int iters, int64_t rows_capacity,
                         float *centroids, int32_t *labels, int32_t *cluster_counts,
                         void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_kmeans_segmented_ex(pool, 0, C, rows, seg_offsets, seg_k, init_rows, n_seg, kmax, iters, rows_capacity, centroids, labels,
                                   cluster_counts, workspace, workspace_bytes, stream);
}

int aoc_kmeans_segmented_ex(const float *pool, int64_t pool_rows, int C, const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                            const int32_t *init_rows, int n_seg, int kmax, int iters, int64_t rows_capacity,
                            float *centroids, int32_t *labels, int32_t *cluster_counts,
                            void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_kmeans_segmented_rep(pool, pool_rows, C, rows, seg_offsets, seg_k, init_rows, n_seg, 1, kmax, iters, rows_capacity, centroids, labels,
                                    cluster_counts, workspace, workspace_bytes, stream);
}

int aoc_kmeans_segmented_rep(const float *pool, int64_t pool_rows, int C, const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                             const int32_t *init_rows, int n_seg, int n_rep, int kmax, int iters, int64_t rows_capacity,
                             float *centroids, int32_t *labels, int32_t *cluster_counts,
                             void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (n_rep < 1 || (n_seg > 0 && n_seg % n_rep != 0)) return AOC_ERR_INVALID_ARG;
    if (!pool || !rows || !seg_offsets || !seg_k || !init_rows || !centroids || !labels || !cluster_counts || !workspace)
        return AOC_ERR_INVALID_ARG;
    if (C < 1 || n_seg < 1 || kmax < 1 || iters < 1 || rows_capacity < 1 || rows_capacity >= (1ll << 31)) return AOC_ERR_INVALID_ARG;
    if (C > AOC_MAX_CHANNELS || kmax > AOC_MAX_CLUSTERS || n_seg > 65535) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_kmeans_workspace_bytes(rows_capacity, n_seg, kmax, C)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    float *cnorm = static_cast<float *>(workspace);
    // a segment lists distinct pool rows, so no segment is longer than the pool: bounds the per-segment grids
    const int64_t seg_bound = (pool_rows > 0 && pool_rows < rows_capacity) ? pool_rows : rows_capacity;
    KsWorkspace ws = ks_carve(static_cast<char *>(workspace) + aoc_align_up((size_t)n_seg * kmax * sizeof(float), 256), rows_capacity, n_seg, kmax,
                              seg_bound);
    float *rownorm = ws.rownorm;
    // scan-sum pipeline: rows addressed by 32-bit byte offsets through a bounds-checked buffer descriptor
    const bool fast = (C % 4) == 0 && C <= 128 && pool_rows > 0 && (uint64_t)pool_rows * C * 4 < 0xFFFFFF00ull;
    const uint32_t pool_bytes = fast ? (uint32_t)((uint64_t)pool_rows * C * 4) : 0u;
    static const bool mfma_assign = !(AOC_DEV_ENV("AOC_KM_ASSIGN") && strcmp(AOC_DEV_ENV("AOC_KM_ASSIGN"), "valu") == 0);   // developer switch

    hipLaunchKernelGGL(km_init_kernel, dim3(kmax, n_seg), dim3(64), 0, st, pool, C, rows, seg_offsets, seg_k, init_rows, kmax,
                       centroids, cnorm, cluster_counts);
    const dim3 agrid((unsigned)((seg_bound + 255) / 256), (unsigned)n_seg);
    const dim3 sgrid((unsigned)((seg_bound + 256 * KSS_BLOCKS - 1) / (256 * KSS_BLOCKS)), (unsigned)n_seg);
    const size_t lds = ((size_t)kmax * C + kmax) * sizeof(float);
    const size_t lds_fast = lds + (size_t)4 * kmax * sizeof(int32_t);
    const int nf = (C + 63) / 64;
    const bool use_mfma = fast && mfma_assign && C == 100 && kmax <= 64;
    if (use_mfma) {
        // row norms up front (replicated lists whose assignment is fused: replica 0's entries are the ones read), so that the first iteration
        // takes the matrix-pipe kernel as well
        const size_t per_r = ((size_t)((kmax + 15) / 16) * 16 * 116 + ((kmax + 15) / 16) * 16) * sizeof(float) + 256;
        const size_t fixed_r = (size_t)4 * 16 * 116 * sizeof(float) + (size_t)4 * kmax * sizeof(int32_t);
        const bool rep_path = n_rep > 1 && (km_rep_lds_budget((kmax + 15) / 16) - fixed_r) / per_r >= 2 && AOC_DEV_ENV_INT("AOC_KM_ASSIGN_REP", 1) != 0;
        const int lim = rep_path ? n_seg / n_rep : n_seg;
        const int64_t bound = rep_path ? std::min<int64_t>(rows_capacity / n_rep + 1, rows_capacity) : rows_capacity;
        hipLaunchKernelGGL(km_rownorm_kernel, dim3((unsigned)((bound + 255) / 256)), dim3(256), 0, st, pool, C, rows, seg_offsets, lim, rownorm);
    }
    for (int it = 0; it < iters; ++it) {
        const int first = (it == 0);
        if (use_mfma) {
            const int kt = (kmax + 15) / 16;
            // K <= 16: the kernels are built for three workgroups per CU (168 registers, <= 53 KB of LDS): 768 resident workgroups instead of 512
            const int64_t gcap = kt == 1 ? (int64_t)km_assign_grid_cap() * 3 / 2 : km_assign_grid_cap();
            bool rep_done = false;
            // replicated segment lists: the rows of a block are staged once for a group of replicas (km_assign_mfma_rep_kernel) as long as
            // at least two code books fit next to the row images in half a CU's LDS
            {
                const size_t fixed = (size_t)4 * 16 * 116 * sizeof(float) + (size_t)4 * kmax * sizeof(int32_t);
                const size_t per = ((size_t)kt * 16 * 116 + kt * 16) * sizeof(float) + 256;
                const int fit = (int)std::min<size_t>(16, (km_rep_lds_budget(kt) - fixed) / per);
                static const bool rep_off = AOC_DEV_ENV_INT("AOC_KM_ASSIGN_REP", 1) == 0;     // developer switch
                if (n_rep > 1 && fit >= 2 && !rep_off) {
                    const int n_groups = (n_rep + fit - 1) / fit;
                    const int n_grp = (n_rep + n_groups - 1) / n_groups;
                    const int n_base = n_seg / n_rep;
                    const size_t rlds = fixed + (size_t)n_grp * per;
                    const int64_t base_rows = std::min<int64_t>(rows_capacity / n_rep, seg_bound * n_base);
                    const unsigned rgrid = (unsigned)std::min<int64_t>((base_rows / 256 + n_base) * n_groups, gcap);
#define AOC_KAR(KT)                                                                                                                                        \
    do {                                                                                                                                                   \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(km_assign_mfma_rep_kernel<25, KT>),                                     \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)km_rep_lds_budget(KT) + 2048) == hipSuccess;           \
        if (!ok) return AOC_ERR_LAUNCH;                                                                                                                    \
        hipLaunchKernelGGL((km_assign_mfma_rep_kernel<25, KT>), dim3(rgrid), dim3(256), rlds, st, pool, C, rows, seg_offsets, seg_k, n_base, n_rep, n_grp, \
                           centroids, kmax, labels, ws.rank16, ws.hist, ws.nb_max, rownorm);                                                               \
    } while (0)
                    if (kt == 1) AOC_KAR(1); else if (kt == 2) AOC_KAR(2); else if (kt == 3) AOC_KAR(3); else AOC_KAR(4);
#undef AOC_KAR
                    rep_done = true;
                }
            }
            if (!rep_done) {
            const size_t alds = ((size_t)kt * 16 * 116 + kt * 16 + (size_t)4 * 16 * 116) * sizeof(float) + (size_t)4 * kmax * sizeof(int32_t);
            const unsigned pgrid = (unsigned)std::min<int64_t>(std::min<int64_t>(rows_capacity, seg_bound * n_seg) / 256 + n_seg, gcap);    // persistent workgroups, several 256-row items each: the code book staging is paid once per workgroup
#define AOC_KA(KT) hipLaunchKernelGGL((km_assign_mfma_kernel<25, KT>), dim3(pgrid), dim3(256), alds, st, pool, C, rows, seg_offsets, seg_k, n_seg, centroids, \
                                      kmax, labels, ws.rank16, ws.hist, ws.nb_max, rownorm)
            if (kt == 1) AOC_KA(1); else if (kt == 2) AOC_KA(2); else if (kt == 3) AOC_KA(3); else AOC_KA(4);
#undef AOC_KA
            }
        } else if (fast) {
            if (C <= 100)
                hipLaunchKernelGGL(km_assign_rank_kernel<25>, agrid, dim3(256), lds_fast, st, pool, C, rows, seg_offsets, seg_k, centroids, kmax, labels,
                                   ws.rank16, ws.hist, ws.nb_max, rownorm, first);
            else
                hipLaunchKernelGGL(km_assign_rank_kernel<32>, agrid, dim3(256), lds_fast, st, pool, C, rows, seg_offsets, seg_k, centroids, kmax, labels,
                                   ws.rank16, ws.hist, ws.nb_max, rownorm, first);
        }
        if (fast) {
            hipLaunchKernelGGL(km_scan_scatter_kernel, sgrid, dim3(256), 0, st, rows, (const int32_t *)nullptr, seg_offsets, seg_k, n_seg, labels, ws.rank16,
                               ws.hist, ws.nb_max, kmax, (uint32_t)C * 4u, ws.moff, cluster_counts, ws.cbase, ws.cchunk, ws.owner_cluster, ws.owner_local,
                               ws.nch_cap, ws.cflag);
            ks_launch_sums<0>(st, pool, pool_bytes, C, seg_offsets, seg_k, cluster_counts, ws, kmax, n_seg, centroids, it > 0);
            continue;
        }
        if ((C % 4) == 0 && C <= 100)
            hipLaunchKernelGGL(km_assign_kernel<25>, agrid, dim3(256), lds, st, pool, C, rows, seg_offsets, seg_k, centroids, cnorm, kmax, labels, rownorm, first);
        else if ((C % 4) == 0 && C <= 128)
            hipLaunchKernelGGL(km_assign_kernel<32>, agrid, dim3(256), lds, st, pool, C, rows, seg_offsets, seg_k, centroids, cnorm, kmax, labels, rownorm, first);
        else
            hipLaunchKernelGGL(km_assign_generic_kernel, agrid, dim3(256), 0, st, pool, C, rows, seg_offsets, seg_k, centroids, cnorm, kmax, labels, rownorm, first);
        const dim3 ugrid(kmax, n_seg);
#define AOC_KU(NF) hipLaunchKernelGGL((km_accumulate_kernel<NF, 0>), ugrid, dim3(64), 0, st, pool, C, rows, seg_offsets, seg_k, labels, kmax, centroids, cnorm, cluster_counts, (float *)nullptr, (float *)nullptr)
        if (nf == 1) AOC_KU(1); else if (nf == 2) AOC_KU(2); else if (nf == 3) AOC_KU(3); else AOC_KU(4);
#undef AOC_KU
    }
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_kmeans_replicate(const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k, int n_seg, int n_rep, int64_t rows_capacity,
                         int32_t *rows_out, int32_t *seg_offsets_out, int32_t *seg_k_out, aoc_stream_t stream) {
    if (!rows || !seg_offsets || !seg_k || !rows_out || !seg_offsets_out || !seg_k_out) return AOC_ERR_INVALID_ARG;
    if (n_seg < 1 || n_rep < 1 || rows_capacity < 1 || (int64_t)n_rep * rows_capacity >= (1ll << 31)) return AOC_ERR_INVALID_ARG;
    const int64_t span = rows_capacity > n_seg + 1 ? rows_capacity : n_seg + 1;
    hipLaunchKernelGGL(km_replicate_kernel, dim3((unsigned)((span + 255) / 256), (unsigned)n_rep), dim3(256), 0, aoc_hip_stream(stream), rows, seg_offsets,
                       seg_k, n_seg, n_rep, rows_capacity, rows_out, seg_offsets_out, seg_k_out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_kmeans_replicate_levels(const int32_t *rows, const int32_t *seg_offsets, int n_seg, int n_rep, const int32_t *levels_host, int n_levels,
                                int64_t rows_capacity, int32_t *rows_out, int32_t *seg_offsets_out, int32_t *seg_k_out, aoc_stream_t stream) {
    if (!rows || !seg_offsets || !levels_host || !rows_out || !seg_offsets_out || !seg_k_out) return AOC_ERR_INVALID_ARG;
    if (n_seg < 1 || n_rep < 1 || n_levels < 1 || rows_capacity < 1 || (int64_t)n_rep * rows_capacity >= (1ll << 31)) return AOC_ERR_INVALID_ARG;
    if (n_levels > 8) return AOC_ERR_UNSUPPORTED;
    if (rows_out == rows && n_rep != 1) return AOC_ERR_INVALID_ARG;
    KmLevels kl;
    kl.n = n_levels;
    for (int i = 0; i < 8; ++i) {
        kl.k[i] = i < n_levels ? levels_host[i] : 0;
        if (i < n_levels && (levels_host[i] < 0 || levels_host[i] > AOC_MAX_CLUSTERS)) return AOC_ERR_INVALID_ARG;
    }
    const int64_t span = rows_capacity > n_seg + 1 ? rows_capacity : n_seg + 1;
    hipLaunchKernelGGL(km_replicate_levels_kernel, dim3((unsigned)((span + 255) / 256), (unsigned)n_rep), dim3(256), 0, aoc_hip_stream(stream), rows,
                       seg_offsets, n_seg, n_rep, kl, rows_capacity, rows_out, seg_offsets_out, seg_k_out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_build_proxies_workspace_bytes(int64_t rows_capacity, int n_seg, int kmax) {
    if (rows_capacity < 0 || n_seg < 1 || kmax < 1) return 0;
    return ks_workspace_bytes(rows_capacity, n_seg, kmax);
}

int aoc_build_proxies(const float *pool, int64_t pool_rows, int C, const int32_t *fg_rows, const int32_t *seg_offsets, const int32_t *seg_k,
                      const int32_t *labels, const float *centroids, int n_seg, int kmax, int64_t rows_capacity,
                      float *proxies, float *proxy_sqnorm, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!pool || !fg_rows || !seg_offsets || !seg_k || !labels || !centroids || !proxies || !proxy_sqnorm) return AOC_ERR_INVALID_ARG;
    if (C < 1 || n_seg < 1 || kmax < 1) return AOC_ERR_INVALID_ARG;
    if (C > AOC_MAX_CHANNELS || kmax > AOC_MAX_CLUSTERS || n_seg > 65535) return AOC_ERR_UNSUPPORTED;
    hipStream_t st = aoc_hip_stream(stream);
    const dim3 grid(kmax, n_seg);
    const int nf = (C + 63) / 64;
    const bool fast = (C % 4) == 0 && C <= 128 && pool_rows > 0 && rows_capacity > 0 && (uint64_t)pool_rows * C * 4 < 0xFFFFFF00ull && workspace &&
                      workspace_bytes >= aoc_build_proxies_workspace_bytes(rows_capacity, n_seg, kmax);
    if (fast) {
        const int64_t seg_bound = (pool_rows < rows_capacity) ? pool_rows : rows_capacity;
        KsWorkspace ws = ks_carve(workspace, rows_capacity, n_seg, kmax, seg_bound);
        const dim3 agrid((unsigned)((seg_bound + 255) / 256), (unsigned)n_seg);
        hipLaunchKernelGGL(km_rank_only_kernel, agrid, dim3(256), 0, st, seg_offsets, seg_k, labels, kmax, ws.rank16, ws.hist, ws.nb_max);
        const dim3 sgrid((unsigned)((seg_bound + 256 * KSS_BLOCKS - 1) / (256 * KSS_BLOCKS)), (unsigned)n_seg);
        hipLaunchKernelGGL(km_scan_scatter_kernel, sgrid, dim3(256), 0, st, (const int32_t *)nullptr, fg_rows, seg_offsets, seg_k, n_seg, labels, ws.rank16,
                           ws.hist, ws.nb_max, kmax, (uint32_t)C * 4u, ws.moff, ws.counts, ws.cbase, ws.cchunk, ws.owner_cluster, ws.owner_local,
                           ws.nch_cap, ws.cflag);
        ks_launch_sums<1>(st, pool, (uint32_t)((uint64_t)pool_rows * C * 4), C, seg_offsets, seg_k, ws.counts, ws, kmax, n_seg, proxies);
        hipLaunchKernelGGL(km_proxy_finish_kernel, grid, dim3(64), 0, st, centroids, seg_k, ws.counts, kmax, C, proxies, proxy_sqnorm);
        AOC_RETURN_IF_LAUNCH_FAILED();
        return AOC_OK;
    }
#define AOC_KP(NF) hipLaunchKernelGGL((km_accumulate_kernel<NF, 1>), grid, dim3(64), 0, st, pool, C, fg_rows, seg_offsets, seg_k, labels, kmax, const_cast<float *>(centroids), (float *)nullptr, (int32_t *)nullptr, proxies, proxy_sqnorm)
    if (nf == 1) AOC_KP(1); else if (nf == 2) AOC_KP(2); else if (nf == 3) AOC_KP(3); else AOC_KP(4);
#undef AOC_KP
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

#ifdef AOC_KS_STATS
int aoc_debug_ks_stats(unsigned long long *out8_host, int reset) {
    if (hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(aoc_ks_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return AOC_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(aoc_ks_stats), z, sizeof(z)) != hipSuccess) return AOC_ERR_LAUNCH;
    }
    return AOC_OK;
}
#endif

}  // extern "C"
