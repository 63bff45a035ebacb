// Pixel-to-proxy correlation (AEM:92-128, 316-319) and dense pixel-level matching (AEM:61-89,
// 178-227) on v_mfma_f32_16x16x4_f32 (exact fp32), fused with the proto-mask transform.
//
// MFMA operand map (16x16x4, f32): A lane l holds A[i = l & 15][k = l >> 4]; B lane l holds
// B[k = l >> 4][j = l & 15]; D reg r of lane l is D[row = (l >> 4) * 4 + r][col = l & 15].
// Query pixels are the A rows, proxies / reference pixels the B columns.  Step t of the K loop
// multiplies channel 4t + kq on both sides (kq = l >> 4), so the B image in LDS is "k-permuted"
// (see aoc_common.h) and each lane streams its channels with ds_read_b128.
#include <stdlib.h>

#include "aoc_common.h"
#include "correlation_shared.h"

namespace {

constexpr int PC_MAX_TILES = 20;

struct ProxyTile {
    int32_t proxy_begin;  // first proxy row of this 16-column tile
    int32_t ncols;        // valid columns (0..16)
    int32_t set;          // kind 0: set index; kind 1: set index of column 0 (indexes set_bias)
    int32_t flags;        // bit0: column-wise (every column is its own set), bit1: first tile of set, bit2: last tile of set
    int64_t out_offset;   // element offset of the set's output plane (kind 1: of column 0)
    int64_t col_stride;   // kind 1: output offset step between consecutive columns
    int32_t oc;           // first output column of this tile in the launch's output-column list
    int32_t pad_;
};
constexpr int PC_MAX_OUT = 64;   // output columns (sets) per launch that go through the per-wave transpose buffer
struct ProxyTileTable {
    ProxyTile t[PC_MAX_TILES];
    int32_t n;
    int32_t n_out;                     // output columns in this launch (0 = too many: direct stores)
    int64_t oc_offset[PC_MAX_OUT];     // element offset of each output column
    int32_t oc_bias[PC_MAX_OUT];       // index into set_bias
};

// Stage 16-column operand tiles into the k-permuted LDS image.  row_of(c) gives the source row
// pointer of tile column c (nullptr = zero fill).
template <bool F16, int C4C, typename RowFn>
__device__ __forceinline__ void stage_tile_rows(float *__restrict__ lds, int n_rows, int C, int TP, int RS, RowFn row_of) {
    const int c4 = C4C > 0 ? C4C : (C >> 2);          // compile-time when the caller knows it: the piece indices then divide by a constant
    const int total = n_rows * c4;
    constexpr int BATCH = 16;     // loads in flight per thread: a naive load->write loop would serialise on latency
    for (int base = 0; base < total; base += BATCH * (int)blockDim.x) {
        float4 v[BATCH];
        int dst[BATCH];
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            const int idx = base + it * (int)blockDim.x + (int)threadIdx.x;
            dst[it] = -1;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < total) {
                const int rr = idx / c4, t = idx - rr * c4;
                const float *src = row_of(rr);
                if (src) v[it] = reinterpret_cast<const float4 *>(src)[t];
                dst[it] = rr * RS + t;
            }
        }
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            if (dst[it] >= 0) {
                float *d = lds + dst[it];
                d[0] = aoc_hr<F16>(v[it].x); d[TP] = aoc_hr<F16>(v[it].y); d[2 * TP] = aoc_hr<F16>(v[it].z); d[3 * TP] = aoc_hr<F16>(v[it].w);
            }
        }
    }
    // zero the TP - T padding of every stream (read by the last ds_read_b128 of a stream)
    const int T = c4, padn = TP - T;
    if (padn > 0) {
        for (int idx = threadIdx.x; idx < n_rows * 4 * padn; idx += blockDim.x) {
            const int rr = idx / (4 * padn), rem = idx - rr * 4 * padn;
            const int kq = rem / padn, u = rem - kq * padn;
            lds[(size_t)rr * RS + kq * TP + T + u] = 0.0f;
        }
    }
}

// Load the A fragment (16 query rows) and the rows' squared norms.
// F16: the reference's `.half()` mode -- the operand is rounded to float16, |q|^2 = h(sum h(q_c^2)) (AEM:200 on a float16 tensor)
template <int TMAX, bool F16 = false>
__device__ __forceinline__ void load_a_fragment(const float *__restrict__ query, int64_t m, int C, int64_t row0, float (&a)[TMAX], float &q2) {
    const int lane = aoc_lane();
    const int i = lane & 15, kq = lane >> 4, T = C >> 2;
    int64_t row = row0 + i;
    if (row > m - 1) row = m - 1;
    const float *src = query + row * C + kq;
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        a[t] = aoc_hr<F16>((t < T) ? src[4 * t] : 0.0f);
        part += aoc_hr<F16>(a[t] * a[t]);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    q2 = aoc_hr<F16>(part);   // |q_i|^2 on every lane with (lane & 15) == i
}

template <int TMAX>
__device__ __forceinline__ f32x4 mfma_tile(const float (&a)[TMAX], const float *__restrict__ bstream, int TP) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (TMAX + 3) / 4; ++u) {
        if (4 * u >= TP) break;   // streams are TP floats long (a[t] = 0 beyond T anyway)
        const float4 b = *reinterpret_cast<const float4 *>(bstream + 4 * u);
        if (4 * u + 0 < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + 0 < TMAX ? 4 * u + 0 : 0], b.x, acc, 0, 0, 0);
        if (4 * u + 1 < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + 1 < TMAX ? 4 * u + 1 : 0], b.y, acc, 0, 0, 0);
        if (4 * u + 2 < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + 2 < TMAX ? 4 * u + 2 : 0], b.z, acc, 0, 0, 0);
        if (4 * u + 3 < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + 3 < TMAX ? 4 * u + 3 : 0], b.w, acc, 0, 0, 0);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------
// proxy_corr_min: one block per CU stages every proxy tile once; each wave walks 16-pixel row tiles.
// Inner loop as in dense_match: compile-time K loop (EXACT), two register B tiles ping-pong so the LDS reads
// of tile t+1 fly under tile t's MFMAs; the set minimum uses DPP row reductions (no LDS crossbar shuffles).
__device__ __forceinline__ float aoc_min16_dpp(float v) {     // lane 15 of each 16-lane row gets the row minimum
    const int inf = 0x7f800000;
    v = aoc_fmin_raw(v, __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), 0x111, 0xf, 0xf, false)));
    v = aoc_fmin_raw(v, __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), 0x112, 0xf, 0xf, false)));
    v = aoc_fmin_raw(v, __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), 0x114, 0xf, 0xf, false)));
    v = aoc_fmin_raw(v, __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v), 0x118, 0xf, 0xf, false)));
    return v;
}

// blockIdx.y = frame of a batched launch (frames share m, C and the set structure; pointers differ)
struct PcFrames {
    const float *query[AOC_CORR_MAX_FRAMES], *proxies[AOC_CORR_MAX_FRAMES], *sqnorm[AOC_CORR_MAX_FRAMES], *bias[AOC_CORR_MAX_FRAMES];
    float *out[AOC_CORR_MAX_FRAMES];
    int32_t n;
};
template <int TMAX, bool EXACT, bool F16>
__global__ __launch_bounds__(256) void proxy_corr_min_kernel(PcFrames frames, int64_t m, int C, ProxyTileTable tiles, int64_t pstride, int transform,
                                                              const int32_t *__restrict__ gate, int32_t gate_value) {
    if (gate && *gate != gate_value) return;   // the fp16-split kernel of correlation_batched.hip owns this launch
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // frames of the launch: one per blockIdx.y, or (gated take-over launch: a small grid that normally exits at once) all of them in turn
    for (int fi = blockIdx.y; fi < frames.n; fi += gridDim.y) {
    if (fi != (int)blockIdx.y) __syncthreads();
    const float *__restrict__ query = frames.query[fi];
    const float *__restrict__ proxies = frames.proxies[fi];
    const float *__restrict__ proxy_sqnorm = frames.sqnorm[fi];
    const float *__restrict__ set_bias = frames.bias[fi];
    float *__restrict__ out = frames.out[fi];
    constexpr int NB4 = (TMAX + 3) / 4;
    const int TP = EXACT ? NB4 * 4 : aoc_tile_tp(C);
    const int RS = EXACT ? (4 * NB4 * 4 + 4) : aoc_tile_row_stride(C);
    const int ncols_total = tiles.n * 16;
    float *lp2 = lds + (size_t)ncols_total * RS;
    float *wbuf_all = lp2 + ncols_total;                      // [4 waves][16 rows][n_out + 1] raw distances
    const int nout = tiles.n_out, wld = nout + 1;
    stage_tile_rows<F16, (EXACT ? TMAX : 0)>(lds, ncols_total, C, TP, RS, [&](int c) -> const float * {
        const ProxyTile &pt = tiles.t[c >> 4];
        return ((c & 15) < pt.ncols) ? proxies + (size_t)(pt.proxy_begin + (c & 15)) * C : nullptr;
    });
    __syncthreads();
    for (int c = threadIdx.x; c < ncols_total; c += blockDim.x) {
        const ProxyTile &pt = tiles.t[c >> 4];
        float v = INFINITY;
        if ((c & 15) < pt.ncols) {
            if (proxy_sqnorm) v = proxy_sqnorm[pt.proxy_begin + (c & 15)];
            if (!proxy_sqnorm || (F16 && v < INFINITY)) {   // |p|^2 from the staged image (AEM:150 .pow(2).sum(1); any order)
                const float *r = lds + (size_t)c * RS;
                v = 0.0f;
                for (int kq = 0; kq < 4; ++kq)
                    for (int t = 0; t < (C >> 2); ++t) v += aoc_hr<F16>(r[kq * TP + t] * r[kq * TP + t]);
                v = aoc_hr<F16>(v);
            }
        }
        lp2[c] = v;
    }
    __syncthreads();

    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int64_t n_row_tiles = (m + 15) / 16;
    float *wbuf = wbuf_all + (size_t)wave * 16 * wld;

    struct BTile {
        float4 b[NB4];
        float p2;
    };
    auto load_tile = [&](int ti, BTile &t) {
        const float *bstream = lds + (size_t)(ti * 16 + j) * RS + g * TP;
#pragma unroll
        for (int u = 0; u < NB4; ++u)
            t.b[u] = (EXACT || 4 * u < TP) ? *reinterpret_cast<const float4 *>(bstream + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        t.p2 = lp2[ti * 16 + j];
    };

    for (int64_t rt = (int64_t)blockIdx.x * 4 + wave; rt < n_row_tiles; rt += (int64_t)gridDim.x * 4) {
        const int64_t row0 = rt * 16;
        float a[TMAX], q2;
        load_a_fragment<TMAX, F16>(query, m, C, row0, a, q2);
        float q2r[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) q2r[r] = __shfl(q2, g * 4 + r);
        float setmin[4] = {INFINITY, INFINITY, INFINITY, INFINITY};

        auto step = [&](int ti, const BTile &t) {
            const ProxyTile pt = tiles.t[ti];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const float bb[4] = {t.b[u].x, t.b[u].y, t.b[u].z, t.b[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * u + e < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + e < TMAX ? 4 * u + e : 0], bb[e], acc, 0, 0, 0);
            }
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)   // AEM:43; in float16 mode every tensor-level result is a float16
                d[r] = F16 ? aoc_h(aoc_h(q2r[r] + t.p2) - 2.0f * aoc_h(acc[r])) : (q2r[r] + t.p2) - 2.0f * acc[r];
            if (pt.flags & 1) {   // column-wise: k = 1 proxies, no min (AEM:127)
                if (j < pt.ncols) {
                    if (nout > 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) wbuf[(g * 4 + r) * wld + pt.oc + j] = d[r];
                    } else {
                        const float b = set_bias ? set_bias[pt.set + j] : 0.0f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int64_t row = row0 + g * 4 + r;
                            if (row < m) out[row * pstride + pt.out_offset + j * pt.col_stride] = transform ? aoc_proto_transform(d[r], b) : d[r];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) setmin[r] = (pt.flags & 2) ? d[r] : aoc_fmin_raw(setmin[r], d[r]);
                if (pt.flags & 4) {
                    const float b = set_bias ? set_bias[pt.set] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = aoc_min16_dpp(setmin[r]);               // AEM:109 min over the set's proxies (valid in lane 15 of the row)
                        if (v == INFINITY) v = AOC_PAD_DISTANCE;          // absent object: AEM:310-313
                        if (nout > 0) {
                            if (j == 15) wbuf[(g * 4 + r) * wld + pt.oc] = v;
                        } else {
                            const int64_t row = row0 + g * 4 + r;
                            if (j == 15 && row < m) out[row * pstride + pt.out_offset] = transform ? aoc_proto_transform(v, b) : v;
                        }
                    }
                }
            }
        };

        BTile t0, t1;
        load_tile(0, t0);
        for (int ti = 0; ti < tiles.n; ti += 2) {
            if (ti + 1 < tiles.n) load_tile(ti + 1, t1);
            step(ti, t0);
            if (ti + 2 < tiles.n) load_tile(ti + 2, t0);
            if (ti + 1 < tiles.n) step(ti + 1, t1);
        }
        // transposed epilogue: every lane transforms and stores its share of the 16 x n_out raw distances, so the
        // expf work is spread over the wave and each output column gets one 64-byte store run
        for (int idx = lane; idx < 16 * nout; idx += 64) {
            const int oc = idx >> 4, r = idx & 15;
            const int64_t row = row0 + r;
            if (row < m) {
                const float v = wbuf[r * wld + oc];
                const float b = set_bias ? set_bias[tiles.oc_bias[oc]] : 0.0f;
                out[row * pstride + tiles.oc_offset[oc]] = transform ? aoc_proto_transform(v, b) : v;
            }
        }
    }
    }   // frames
}

// ------------------------------------------------------------------------------------------
// dense_match_min.  Block = 4 waves x NA A-tiles (16 query pixels each); reference pixels stream
// through LDS in chunks of DM_NB 16-row tiles; the m x n distance matrix only ever exists as MFMA
// accumulators.  Grid = (row blocks, n-splits); partial minima go to the workspace.
constexpr int DM_NB = 8;

template <bool F16>
__global__ __launch_bounds__(256) void gather_sqnorm_kernel(const float *__restrict__ pool, int C, const int32_t *__restrict__ fg_rows,
                                                             const int32_t *__restrict__ n_fg, float *__restrict__ r2,
                                                             const int32_t *__restrict__ gate) {
    if (gate && *gate == 0) return;            // the split-fp16 kernels own this call (dense_split.hip)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= *n_fg) return;
    const float *x = pool + (size_t)fg_rows[p] * C;
    float s = 0.0f;
    for (int t = 0; t < C; ++t) {
        const float v = aoc_hr<F16>(x[t]);
        s += aoc_hr<F16>(v * v);
    }
    r2[p] = aoc_hr<F16>(s);
}

// One B tile (16 reference pixels) held in registers: the lane's k-stream as TP/4 float4, plus the
// column's |r|^2 and wrong-label bits.
template <int NB4>
struct DenseBTile {
    float4 b[NB4];
    float r2;
    uint32_t wrong;
};

// NW waves per block (NW * 64 threads); block = NW * NA * 16 query pixels; one LDS chunk buffer shared by all
// waves; the NEXT chunk's row ids and rows are fetched into registers while the current chunk is multiplied
// (loads stay in flight across the barrier, written to LDS after it), so the matrix pipe only idles for the
// LDS write pass.
template <int NA, int OMAX, int TMAX, bool EXACT, int NW, bool F16>
__global__ __launch_bounds__(NW * 64, 2) void dense_match_partial_kernel(const float *__restrict__ query, int64_t m, int C,
                                                                          const float *__restrict__ pool, const int32_t *__restrict__ fg_rows,
                                                                          const int32_t *__restrict__ n_fg_ptr, const float *__restrict__ r2_all,
                                                                          const uint32_t *__restrict__ wrong_bits, int n_obj,
                                                                          float *__restrict__ partial, const int32_t *__restrict__ gate, int obj_base) {
    // objects [obj_base, obj_base + OMAX) of the n_obj: more than 16 objects take a second launch over the same pixels
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NT = NW * 64;
    constexpr int NB4 = (TMAX + 3) / 4;
    constexpr int ROWS = DM_NB * 16;                                   // reference pixels per chunk
    constexpr int STAGE_ITERS = (ROWS * TMAX + NT - 1) / NT;           // float4 loads per thread per chunk
    constexpr int META_ITERS = (ROWS + NT - 1) / NT;
    const int TP = EXACT ? NB4 * 4 : aoc_tile_tp(C);
    const int RS = EXACT ? (4 * NB4 * 4 + 4) : aoc_tile_row_stride(C);
    float *lr2 = lds + (size_t)ROWS * RS;
    uint32_t *lwrong = reinterpret_cast<uint32_t *>(lr2 + ROWS);
    int32_t *lrow = reinterpret_cast<int32_t *>(lwrong + ROWS);        // [2][ROWS] row ids of chunk c+1 / c+2

    const int n_fg = *n_fg_ptr;
    const int n_tiles = (n_fg + 15) / 16;
    const int tiles_per_split = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int tile_beg = blockIdx.y * tiles_per_split;
    const int tile_end = min(n_tiles, tile_beg + tiles_per_split);

    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int64_t block_row0 = (int64_t)blockIdx.x * (16 * NA * NW);
    const int64_t wave_row0 = block_row0 + (int64_t)wave * 16 * NA;
    const int c4 = EXACT ? TMAX : (C >> 2);

    float a[NA][TMAX], q2r[NA][4];
#pragma unroll
    for (int ia = 0; ia < NA; ++ia) {
        float q2;
        load_a_fragment<TMAX, F16>(query, m, C, wave_row0 + ia * 16, a[ia], q2);
#pragma unroll
        for (int r = 0; r < 4; ++r) q2r[ia][r] = __shfl(q2, g * 4 + r);
    }
    float mn[OMAX][NA][4];
#pragma unroll
    for (int o = 0; o < OMAX; ++o)
#pragma unroll
        for (int ia = 0; ia < NA; ++ia)
#pragma unroll
            for (int r = 0; r < 4; ++r) mn[o][ia][r] = INFINITY;

    auto load_tile = [&](int ti, DenseBTile<NB4> &t) {
        const float *bstream = lds + (size_t)(ti * 16 + j) * RS + g * TP;
#pragma unroll
        for (int u = 0; u < NB4; ++u)
            t.b[u] = (EXACT || 4 * u < TP) ? *reinterpret_cast<const float4 *>(bstream + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        t.r2 = lr2[ti * 16 + j];
        t.wrong = lwrong[ti * 16 + j];
    };
    auto step = [&](const DenseBTile<NB4> &t) {
        f32x4 acc[NA];
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) acc[ia] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            const float bb[4] = {t.b[u].x, t.b[u].y, t.b[u].z, t.b[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (4 * u + e < TMAX) {
#pragma unroll
                    for (int ia = 0; ia < NA; ++ia)
                        acc[ia] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ia][4 * u + e < TMAX ? 4 * u + e : 0], bb[e], acc[ia], 0, 0, 0);
                }
            }
        }
        float padv[OMAX];   // per-column padding of every object (AEM:84-86); objects >= n_obj are never written
#pragma unroll
        for (int o = 0; o < OMAX; ++o) padv[o] = ((t.wrong >> (o + obj_base)) & 1u) ? (F16 ? aoc_h(AOC_PAD_DISTANCE) : AOC_PAD_DISTANCE) : 0.0f;
#pragma unroll
        for (int ia = 0; ia < NA; ++ia)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // AEM:43; float16 mode (AEM:65-66, 801-803): dists, the padded sum and the min are float16 tensors
                const float d = F16 ? aoc_h(aoc_h(q2r[ia][r] + t.r2) - 2.0f * aoc_h(acc[ia][r])) : (q2r[ia][r] + t.r2) - 2.0f * acc[ia][r];
#pragma unroll
                for (int o = 0; o < OMAX; ++o) mn[o][ia][r] = aoc_fmin_raw(mn[o][ia][r], F16 ? aoc_h(d + padv[o]) : d + padv[o]);   // AEM:88
            }
    };

    // ---- staging pipeline state (registers)
    struct Staged {
        float4 v[STAGE_ITERS];
        float r2[META_ITERS];
        uint32_t wrong[META_ITERS];
    } stg;
    int32_t ids_next[META_ITERS];
    auto load_ids = [&](int chunk) {          // row ids of `chunk` -> registers (-1 = padding)
#pragma unroll
        for (int it = 0; it < META_ITERS; ++it) {
            const int c = it * NT + threadIdx.x;
            const int p = chunk * 16 + c;
            ids_next[it] = (c < ROWS && chunk < tile_end && p < min(n_fg, tile_end * 16)) ? fg_rows[p] : -1;
        }
    };
    auto store_ids = [&](int buf) {
#pragma unroll
        for (int it = 0; it < META_ITERS; ++it) {
            const int c = it * NT + threadIdx.x;
            if (c < ROWS) lrow[buf * ROWS + c] = ids_next[it];
        }
    };
    auto issue_rows = [&](int chunk, int buf) {   // rows + per-row metadata of `chunk` (ids in lrow[buf]) -> registers
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int idx = it * NT + threadIdx.x;
            const int rr = idx / c4, t = idx - rr * c4;
            const int row = (idx < ROWS * c4) ? lrow[buf * ROWS + rr] : -1;
            stg.v[it] = (row >= 0) ? reinterpret_cast<const float4 *>(pool + (size_t)row * C)[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < META_ITERS; ++it) {
            const int c = it * NT + threadIdx.x;
            const int row = (c < ROWS) ? lrow[buf * ROWS + c] : -1;
            stg.r2[it] = (row >= 0) ? r2_all[chunk * 16 + c] : INFINITY;
            stg.wrong[it] = (row >= 0) ? wrong_bits[row] : 0xffffffffu;
        }
    };
    auto commit_rows = [&]() {
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            const int idx = it * NT + threadIdx.x;
            if (idx < ROWS * c4) {
                const int rr = idx / c4, t = idx - rr * c4;
                float *d = lds + (size_t)rr * RS + t;
                d[0] = aoc_hr<F16>(stg.v[it].x); d[TP] = aoc_hr<F16>(stg.v[it].y); d[2 * TP] = aoc_hr<F16>(stg.v[it].z); d[3 * TP] = aoc_hr<F16>(stg.v[it].w);
            }
        }
#pragma unroll
        for (int it = 0; it < META_ITERS; ++it) {
            const int c = it * NT + threadIdx.x;
            if (c < ROWS) { lr2[c] = stg.r2[it]; lwrong[c] = stg.wrong[it]; }
        }
    };

    // zero the stream padding once (never overwritten: rows only write their T real entries per stream)
    {
        const int padn = TP - c4;
        for (int idx = threadIdx.x; idx < ROWS * 4 * padn; idx += NT) {
            const int rr = idx / (4 * padn), rem = idx - rr * 4 * padn;
            lds[(size_t)rr * RS + (rem / padn) * TP + c4 + (rem % padn)] = 0.0f;
        }
    }
    // prologue: ids(chunk0) -> LDS, rows(chunk0) -> regs -> LDS, ids(chunk1) -> LDS
    load_ids(tile_beg);
    store_ids(0);
    __syncthreads();
    issue_rows(tile_beg, 0);
    load_ids(tile_beg + DM_NB);
    commit_rows();
    store_ids(1);
    __syncthreads();

    int parity = 1;                                   // lrow[parity] holds the ids of the NEXT chunk
    for (int chunk = tile_beg; chunk < tile_end; chunk += DM_NB) {
        const int nt = min(DM_NB, tile_end - chunk);
        const bool more = chunk + DM_NB < tile_end;
        if (more) {
            issue_rows(chunk + DM_NB, parity);        // in flight under this chunk's MFMAs
            load_ids(chunk + 2 * DM_NB);
        }
        // two register tiles ping-pong so the LDS reads of tile t+1 are in flight under tile t's MFMAs
        DenseBTile<NB4> t0, t1;
        load_tile(0, t0);
        for (int ti = 0; ti < nt; ti += 2) {
            if (ti + 1 < nt) load_tile(ti + 1, t1);
            step(t0);
            if (ti + 2 < nt) load_tile(ti + 2, t0);
            if (ti + 1 < nt) step(t1);
        }
        __syncthreads();                              // every wave is done reading this chunk
        if (more) {
            commit_rows();
            store_ids(parity ^ 1);                    // ids of chunk+2 replace the ids of the chunk just consumed
            parity ^= 1;
        }
        __syncthreads();
    }
    // reduce over the 16 columns a lane group holds and write this split's partial minima
#pragma unroll
    for (int o = 0; o < OMAX; ++o) {
        if (o + obj_base < n_obj) {
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) {
                float v = 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float mr = aoc_min16(mn[o][ia][r]);
                    if (j == r) v = mr;
                }
                const int64_t row = wave_row0 + ia * 16 + g * 4 + j;
                if (j < 4 && row < m) partial[((int64_t)blockIdx.y * m + row) * n_obj + obj_base + o] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void dense_match_finalize_kernel(const float *__restrict__ partial, int n_split, int64_t m, int n_obj,
                                                                    const int32_t *__restrict__ n_fg_ptr, const float *__restrict__ obj_bias,
                                                                    float *__restrict__ out, int64_t pstride, int64_t ostride, int transform,
                                                                    const int32_t *__restrict__ gate) {
    if (gate && *gate == 0) return;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * n_obj) return;
    const int64_t row = idx / n_obj;
    const int o = (int)(idx - row * n_obj);
    float v;
    if (*n_fg_ptr == 0) {
        v = transform ? 1.0f : INFINITY;   // AEM:796-797: nothing labelled -> ones
    } else {
        v = INFINITY;
        for (int s = 0; s < n_split; ++s) v = fminf(v, partial[((int64_t)s * m + row) * n_obj + o]);
        if (transform) v = aoc_proto_transform(v, obj_bias ? obj_bias[o] : 0.0f);
    }
    out[row * pstride + o * ostride] = v;
}

constexpr int DM_NW = 8;   // waves per block
inline int dense_nsplit(int64_t m, int na) {
    // One 8-wave block per CU at a time (registers).  Pick the n-split so that row_blocks * nsplit fills whole
    // rounds of 256 CUs (tail effect); among near-equal fills prefer MORE, shorter blocks: CUs then free up
    // often, which lets the latency-bound kernels of other streams (k-means chain) interleave with this one.
    const int64_t row_blocks = (m + 16 * DM_NW * na - 1) / (16 * DM_NW * na);
    static const int max_rounds = AOC_DEV_ENV_INT("AOC_DENSE_ROUNDS", 4);
    int best = 1;
    double best_eff = 0.0;
    for (int k = 1; k <= max_rounds; ++k) {
        int64_t ns = (256 * k) / row_blocks;
        if (ns < 1) ns = 1;
        if (ns > 64) ns = 64;
        const int64_t blocks = row_blocks * ns;
        const int64_t rounds = (blocks + 255) / 256;
        const double eff = (double)blocks / (256.0 * rounds);
        if (eff >= best_eff - 0.005) { best_eff = eff > best_eff ? eff : best_eff; best = (int)ns; }
    }
    return best;
}
inline int dense_na(int n_obj) { return n_obj <= 4 ? 2 : 1; }

}  // namespace

extern "C" {

int aoc_proxy_corr_min(const float *query, int64_t m, int C, const float *proxies, const float *proxy_sqnorm, int n_proxy,
                       int n_set, const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                       const float *set_bias, float *out, int64_t out_pixel_stride, int transform, aoc_stream_t stream) {
    if (!query || !proxies || !out) return AOC_ERR_INVALID_ARG;
    const aoc_corr_frame fr = {query, proxies, proxy_sqnorm, set_bias, out};
    return aoc_corr_fp32_batched(&fr, 1, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, out_pixel_stride, transform,
                                 nullptr, stream, 0);
}

int aoc_proxy_corr_min_f16(const float *query, int64_t m, int C, const float *proxies, const float *proxy_sqnorm, int n_proxy,
                           int n_set, const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                           const float *set_bias, float *out, int64_t out_pixel_stride, int transform, aoc_stream_t stream) {
    if (!query || !proxies || !out) return AOC_ERR_INVALID_ARG;
    const aoc_corr_frame fr = {query, proxies, proxy_sqnorm, set_bias, out};
    return aoc_corr_fp32_batched(&fr, 1, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, out_pixel_stride, transform,
                                 nullptr, stream, 1);
}

size_t aoc_dense_match_workspace_bytes(int64_t m, int64_t n_fg_capacity, int n_obj) {
    if (m < 1 || n_fg_capacity < 0 || n_obj < 1) return 0;
    const int ns = dense_nsplit(m, dense_na(n_obj));
    return aoc_align_up((size_t)n_fg_capacity * sizeof(float) + 16, 256) + aoc_align_up((size_t)ns * m * n_obj * sizeof(float), 256);
}

int aoc_dense_match_min(const float *query, int64_t m, int C, const float *pool, const int32_t *fg_rows, const int32_t *n_fg,
                        int64_t n_fg_capacity, const uint32_t *wrong_bits, const float *obj_bias, int n_obj,
                        float *out, int64_t out_pixel_stride, int64_t out_obj_stride, int transform,
                        void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_dense_match_min_gated(query, m, C, pool, fg_rows, n_fg, n_fg_capacity, wrong_bits, obj_bias, n_obj, out, out_pixel_stride,
                                     out_obj_stride, transform, workspace, workspace_bytes, nullptr, stream, 0);
}

int aoc_dense_match_min_f16(const float *query, int64_t m, int C, const float *pool, const int32_t *fg_rows, const int32_t *n_fg,
                            int64_t n_fg_capacity, const uint32_t *wrong_bits, const float *obj_bias, int n_obj,
                            float *out, int64_t out_pixel_stride, int64_t out_obj_stride, int transform,
                            void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_dense_match_min_gated(query, m, C, pool, fg_rows, n_fg, n_fg_capacity, wrong_bits, obj_bias, n_obj, out, out_pixel_stride,
                                     out_obj_stride, transform, workspace, workspace_bytes, nullptr, stream, 1);
}

}  // extern "C"

int aoc_corr_fp32_batched(const aoc_corr_frame *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                          const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                          int64_t out_pixel_stride, int transform, const int32_t *gate, aoc_stream_t stream, int float16, int32_t gate_value) {
    if (!frames_host || !set_begin_host || !set_size_host || !set_out_offset_host) return AOC_ERR_INVALID_ARG;
    if (n_frames < 1 || m < 1 || C < 4 || n_set < 1 || n_proxy < 0) return AOC_ERR_INVALID_ARG;
    if ((C & 3) || C > AOC_MAX_CHANNELS) return AOC_ERR_UNSUPPORTED;
    for (int s = 0; s < n_set; ++s)
        if (set_size_host[s] < 0 || set_begin_host[s] < 0 || set_begin_host[s] + set_size_host[s] > n_proxy) return AOC_ERR_INVALID_ARG;
    hipStream_t st = aoc_hip_stream(stream);
    const int RS = aoc_tile_row_stride(C);
    const size_t tile_bytes = (size_t)16 * RS * sizeof(float) + 16 * sizeof(float);
    int max_tiles = (int)((size_t)130 * 1024 / tile_bytes);   // leaves room for the per-wave transpose buffer
    if (max_tiles > PC_MAX_TILES) max_tiles = PC_MAX_TILES;
    if (max_tiles < 4) return AOC_ERR_UNSUPPORTED;
    const int64_t n_row_tiles = (m + 15) / 16;
    int grid = (int)((n_row_tiles + 3) / 4);
    if (grid > 512) grid = 512;        // two blocks per CU (LDS permitting): tiles staged once per block, waves walk the row tiles
    for (int f0 = 0; f0 < n_frames; f0 += AOC_CORR_MAX_FRAMES) {
        const int nf = n_frames - f0 < AOC_CORR_MAX_FRAMES ? n_frames - f0 : AOC_CORR_MAX_FRAMES;
        PcFrames fr;
        fr.n = nf;
        for (int f = 0; f < AOC_CORR_MAX_FRAMES; ++f) {
            const aoc_corr_frame &src = frames_host[f0 + (f < nf ? f : 0)];
            if (!src.query || !src.proxies || !src.out) return AOC_ERR_INVALID_ARG;
            fr.query[f] = src.query; fr.proxies[f] = src.proxies; fr.sqnorm[f] = src.proxy_sqnorm; fr.bias[f] = src.set_bias; fr.out[f] = src.out;
        }
        ProxyTileTable tab;
        tab.n = 0;
        tab.n_out = 0;
        bool out_overflow = false;
        auto add_out = [&](int64_t off, int bias_idx) -> int {
            if (tab.n_out >= PC_MAX_OUT) { out_overflow = true; return 0; }
            tab.oc_offset[tab.n_out] = off;
            tab.oc_bias[tab.n_out] = bias_idx;
            return tab.n_out++;
        };
        auto flush = [&]() -> int {
            if (tab.n == 0) return AOC_OK;
            if (out_overflow) tab.n_out = 0;       // too many output columns for the transpose buffer: direct stores
            const size_t lds = (size_t)tab.n * tile_bytes + (size_t)4 * 16 * (tab.n_out + 1) * sizeof(float);
#define AOC_PC(TM, EX, F16) hipLaunchKernelGGL((proxy_corr_min_kernel<TM, EX, F16>), dim3(gate ? (grid < 32 ? grid : 32) : grid, gate ? 1 : nf), dim3(256), lds, st, fr, m, C, tab, out_pixel_stride, transform, gate, gate_value)
            if (float16) { if (C == 100) AOC_PC(25, true, true); else if (C <= 128) AOC_PC(32, false, true); else AOC_PC(64, false, true); }
            else if (C == 100) AOC_PC(25, true, false); else if (C <= 128) AOC_PC(32, false, false); else AOC_PC(64, false, false);
#undef AOC_PC
            tab.n = 0;
            tab.n_out = 0;
            out_overflow = false;
            return hipGetLastError() == hipSuccess ? AOC_OK : AOC_ERR_LAUNCH;
        };
        int s = 0;
        while (s < n_set) {
            const int size = set_size_host[s];
            if (size == 1) {
                // run of single-proxy sets over consecutive proxies with a constant output step -> one column-wise tile
                int run = 1;
                const int64_t step = (s + 1 < n_set) ? set_out_offset_host[s + 1] - set_out_offset_host[s] : 0;
                while (run < 16 && s + run < n_set && set_size_host[s + run] == 1 && set_begin_host[s + run] == set_begin_host[s] + run &&
                       set_out_offset_host[s + run] - set_out_offset_host[s + run - 1] == step)
                    ++run;
                if (tab.n + 1 > max_tiles) { int rc = flush(); if (rc) return rc; }
                const int oc0 = tab.n_out;
                for (int c = 0; c < run; ++c) add_out(set_out_offset_host[s + c], s + c);
                tab.t[tab.n++] = ProxyTile{set_begin_host[s], run, s, 1, set_out_offset_host[s], step, oc0, 0};
                s += run;
            } else {
                const int nt = size == 0 ? 1 : (size + 15) / 16;
                if (nt > max_tiles) return AOC_ERR_UNSUPPORTED;
                if (tab.n + nt > max_tiles) { int rc = flush(); if (rc) return rc; }
                const int oc0 = add_out(set_out_offset_host[s], s);
                for (int t = 0; t < nt; ++t) {
                    const int cols = size == 0 ? 0 : ((t == nt - 1) ? size - 16 * t : 16);
                    tab.t[tab.n++] = ProxyTile{set_begin_host[s] + 16 * t, cols, s, (t == 0 ? 2 : 0) | (t == nt - 1 ? 4 : 0), set_out_offset_host[s], 0, oc0, 0};
                }
                ++s;
            }
        }
        int rc = flush();
        if (rc) return rc;
    }
    return AOC_OK;
}

static thread_local AocDenseProbe g_dense_probe = {nullptr, nullptr};
AocDenseProbe aoc_take_dense_probe() {
    const AocDenseProbe p = g_dense_probe;
    g_dense_probe = AocDenseProbe{nullptr, nullptr};
    return p;
}
extern "C" int aoc_dense_match_set_probe(void *start_event, void *stop_event) {
    g_dense_probe = AocDenseProbe{static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)};
    return AOC_OK;
}

int aoc_dense_match_min_gated(const float *query, int64_t m, int C, const float *pool, const int32_t *fg_rows, const int32_t *n_fg,
                              int64_t n_fg_capacity, const uint32_t *wrong_bits, const float *obj_bias, int n_obj,
                              float *out, int64_t out_pixel_stride, int64_t out_obj_stride, int transform,
                              void *workspace, size_t workspace_bytes, const int32_t *gate, aoc_stream_t stream, int float16) {
    if (!query || !pool || !fg_rows || !n_fg || !wrong_bits || !out || !workspace) return AOC_ERR_INVALID_ARG;
    if (m < 1 || C < 4 || n_obj < 1 || n_fg_capacity < 1 || n_fg_capacity >= (1ll << 31)) return AOC_ERR_INVALID_ARG;
    if ((C & 3) || C > 128 || n_obj > AOC_MAX_OBJECTS) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_dense_match_workspace_bytes(m, n_fg_capacity, n_obj)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    float *r2 = static_cast<float *>(workspace);
    float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + aoc_align_up((size_t)n_fg_capacity * sizeof(float) + 16, 256));
    const int na = dense_na(n_obj);
    const int ns = dense_nsplit(m, na);
    const int row_blocks = (int)((m + 16 * DM_NW * na - 1) / (16 * DM_NW * na));
    const int RS = aoc_tile_row_stride(C);
    const size_t lds = (size_t)DM_NB * 16 * RS * sizeof(float) + DM_NB * 16 * (sizeof(float) + sizeof(uint32_t) + 2 * sizeof(int32_t));

    if (float16) hipLaunchKernelGGL(gather_sqnorm_kernel<true>, dim3((unsigned)((n_fg_capacity + 255) / 256)), dim3(256), 0, st, pool, C, fg_rows, n_fg, r2, gate);
    else hipLaunchKernelGGL(gather_sqnorm_kernel<false>, dim3((unsigned)((n_fg_capacity + 255) / 256)), dim3(256), 0, st, pool, C, fg_rows, n_fg, r2, gate);
    const dim3 grid(row_blocks, ns);
    const AocDenseProbe probe = gate ? AocDenseProbe{nullptr, nullptr} : aoc_take_dense_probe();
    if (probe.start) (void)hipEventRecord(probe.start, st);
    for (int obj_base = 0; obj_base < n_obj; obj_base += 16) {
    const int n_here = n_obj - obj_base;
#define AOC_DM(NA, OM, TM, EX, F16) hipLaunchKernelGGL((dense_match_partial_kernel<NA, OM, TM, EX, DM_NW, F16>), grid, dim3(DM_NW * 64), lds, st, query, m, C, pool, fg_rows, n_fg, r2, wrong_bits, n_obj, partial, gate, obj_base)
    if (float16) {
        if (C == 100) {
            if (n_obj <= 4 && obj_base == 0) AOC_DM(2, 4, 25, true, true); else if (n_here <= 8) AOC_DM(1, 8, 25, true, true); else AOC_DM(1, 16, 25, true, true);
        } else {
            if (n_obj <= 4 && obj_base == 0) AOC_DM(2, 4, 32, false, true); else if (n_here <= 8) AOC_DM(1, 8, 32, false, true); else AOC_DM(1, 16, 32, false, true);
        }
    } else if (C == 100) {
        if (n_obj <= 4 && obj_base == 0) AOC_DM(2, 4, 25, true, false); else if (n_here <= 8) AOC_DM(1, 8, 25, true, false); else AOC_DM(1, 16, 25, true, false);
    } else {
        if (n_obj <= 4 && obj_base == 0) AOC_DM(2, 4, 32, false, false); else if (n_here <= 8) AOC_DM(1, 8, 32, false, false); else AOC_DM(1, 16, 32, false, false);
    }
#undef AOC_DM
    }
    if (probe.stop) (void)hipEventRecord(probe.stop, st);
    const int64_t total = m * n_obj;
    hipLaunchKernelGGL(dense_match_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, ns, m, n_obj, n_fg,
                       obj_bias, out, out_pixel_stride, out_obj_stride, transform, gate);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}
