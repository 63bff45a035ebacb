// Label preparation, segmented k-means (bit-identical to scipy kmeans2) and adaptive-proxy
// construction.  Reference: AEM:252-286 (adaptive_embedding_for_matching.py) + scipy.cluster.vq.
#include "aoc_common.h"

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

inline int label_blocks(int64_t n) { return (int)((n + LP_BLOCK - 1) / LP_BLOCK); }

}  // namespace

extern "C" {

const char *aoc_version(void) { return "aoc_hip 0.1 (gfx950, fp32-exact)"; }

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
    return aoc_align_up((size_t)n_seg * kmax * sizeof(float), 256) + aoc_align_up((size_t)rows_capacity * sizeof(float), 256);
}

int aoc_kmeans_segmented(const float *pool, int C, const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                         const int32_t *init_rows, int n_seg, int kmax, int iters, int64_t rows_capacity,
                         float *centroids, int32_t *labels, int32_t *cluster_counts,
                         void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!pool || !rows || !seg_offsets || !seg_k || !init_rows || !centroids || !labels || !cluster_counts || !workspace)
        return AOC_ERR_INVALID_ARG;
    if (C < 1 || n_seg < 1 || kmax < 1 || iters < 1 || rows_capacity < 1 || rows_capacity >= (1ll << 31)) return AOC_ERR_INVALID_ARG;
    if (C > AOC_MAX_CHANNELS || kmax > AOC_MAX_CLUSTERS || n_seg > 65535) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_kmeans_workspace_bytes(rows_capacity, n_seg, kmax, C)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    float *cnorm = static_cast<float *>(workspace);
    float *rownorm = reinterpret_cast<float *>(static_cast<char *>(workspace) + aoc_align_up((size_t)n_seg * kmax * sizeof(float), 256));

    hipLaunchKernelGGL(km_init_kernel, dim3(kmax, n_seg), dim3(64), 0, st, pool, C, rows, seg_offsets, seg_k, init_rows, kmax,
                       centroids, cnorm, cluster_counts);
    const dim3 agrid((unsigned)((rows_capacity + 255) / 256), (unsigned)n_seg);
    const size_t lds = ((size_t)kmax * C + kmax) * sizeof(float);
    const int nf = (C + 63) / 64;
    for (int it = 0; it < iters; ++it) {
        const int first = (it == 0);
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

int aoc_build_proxies(const float *pool, int C, const int32_t *fg_rows, const int32_t *seg_offsets, const int32_t *seg_k,
                      const int32_t *labels, const float *centroids, int n_seg, int kmax,
                      float *proxies, float *proxy_sqnorm, aoc_stream_t stream) {
    if (!pool || !fg_rows || !seg_offsets || !seg_k || !labels || !centroids || !proxies || !proxy_sqnorm) return AOC_ERR_INVALID_ARG;
    if (C < 1 || n_seg < 1 || kmax < 1) return AOC_ERR_INVALID_ARG;
    if (C > AOC_MAX_CHANNELS || kmax > AOC_MAX_CLUSTERS || n_seg > 65535) return AOC_ERR_UNSUPPORTED;
    hipStream_t st = aoc_hip_stream(stream);
    const dim3 grid(kmax, n_seg);
    const int nf = (C + 63) / 64;
#define AOC_KP(NF) hipLaunchKernelGGL((km_accumulate_kernel<NF, 1>), grid, dim3(64), 0, st, pool, C, fg_rows, seg_offsets, seg_k, labels, kmax, const_cast<float *>(centroids), (float *)nullptr, (int32_t *)nullptr, proxies, proxy_sqnorm)
    if (nf == 1) AOC_KP(1); else if (nf == 2) AOC_KP(2); else if (nf == 3) AOC_KP(3); else AOC_KP(4);
#undef AOC_KP
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

}  // extern "C"
