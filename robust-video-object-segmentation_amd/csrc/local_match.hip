// Local (windowed) matching, AEM:921-963 + 968-1060, without F.unfold: the [HW, (2R+1)^2] distance
// volume only exists as MFMA accumulators.  Also the three resize helpers of that path.
//
// Block = 4 waves = 4 consecutive query rows x 16 query columns.  For every candidate row cy the
// block stages the 16 + 2R candidate pixels of the previous frame (k-permuted LDS image, see
// aoc_common.h); each wave whose query row is within R of cy multiplies its 16 query pixels against
// them (v_mfma_f32_16x16x4_f32) and folds the masked distances into per-(pixel, ring, object)
// minima in LDS with ds_min_f32.  "ring" = max(|dy|,|dx|) bucketed by the nested window radii, so
// the nested-window minima of AEM:1036-1046 are a prefix-min over rings at the end.
#include <algorithm>

#include "aoc_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

__device__ __forceinline__ void lds_fmin(float *p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_ds_fminf((__attribute__((address_space(3))) float *)p, v, 0, 0, false);
#endif
}

constexpr int LM_MAX_RADII = 8;
struct LocalRadii {
    int32_t r[LM_MAX_RADII];
    int32_t n;
};

// radii are in units of the atrous rate (ring a = max(|dy|, |dx|) / rate; only offsets that are multiples of the rate exist,
// AEM:949-959 unfold with stride = atrous_rate); R = rate * radii.r[n-1] is the window half-size in pixels.
// f16 != 0: the reference's `.half()` mode (AEM:1002-1005): operands, norms, dot products and distances rounded to float16.
template <int TMAX>
__global__ __launch_bounds__(256) void local_window_kernel(const float *__restrict__ query, const float *__restrict__ prev,
                                                            const uint32_t *__restrict__ right_bits, int H, int W, int C,
                                                            LocalRadii radii, const float *__restrict__ obj_bias, int n_obj,
                                                            float *__restrict__ out, int transform, int rate, int f16) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int RA = radii.r[radii.n - 1];             // window half-size in atrous units
    const int R = RA * rate;
    const float padv = f16 ? aoc_h(AOC_PAD_DISTANCE) : AOC_PAD_DISTANCE;
    const int TP = aoc_tile_tp(C), RS = aoc_tile_row_stride(C);
    const int NC = 16 + 2 * R;
    const int NG = (NC + 15) / 16;
    float *ly2 = lds + (size_t)NG * 16 * RS;                          // [NG*16] candidate |y|^2
    uint32_t *lbits = reinterpret_cast<uint32_t *>(ly2 + NG * 16);     // [NG*16] candidate label bits
    int32_t *lcls = reinterpret_cast<int32_t *>(lbits + NG * 16);      // [R+1] ring -> class
    float *lacc = reinterpret_cast<float *>(lcls + 32);                // [4 waves][16][n_radii][n_obj]
    const int nr = radii.n;
    const int acc_per_wave = 16 * nr * n_obj;

    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * 16;
    const int y0 = blockIdx.y * 4;
    const int y = y0 + wave;

    for (int i = threadIdx.x; i < 4 * acc_per_wave; i += blockDim.x) lacc[i] = padv;   // AEM:1032 pad
    if ((int)threadIdx.x <= RA) {
        int c = 0;
        while (radii.r[c] < (int)threadIdx.x) ++c;
        lcls[threadIdx.x] = c;
    }

    float a[TMAX], q2 = 0.0f, q2r[4];
    {   // A fragment: 16 query pixels of row y (clamped when y >= H; such waves never accumulate)
        const int i = lane & 15, kq = lane >> 4, T = C >> 2;
        const int qx = min(x0 + i, W - 1), qy = min(y, H - 1);
        const float *src = query + ((size_t)qy * W + qx) * C + kq;
        float part = 0.0f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            a[t] = (t < T) ? src[4 * t] : 0.0f;
            if (f16) { a[t] = aoc_h(a[t]); part += aoc_h(a[t] * a[t]); }
            else part += a[t] * a[t];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        q2 = f16 ? aoc_h(part) : part;
#pragma unroll
        for (int r = 0; r < 4; ++r) q2r[r] = __shfl(q2, g * 4 + r);
    }
    float *my_acc = lacc + wave * acc_per_wave;

    const int cy_beg = max(0, y0 - R), cy_end = min(H - 1, y0 + 3 + R);
    for (int cy = cy_beg; cy <= cy_end; ++cy) {
        __syncthreads();
        {   // stage candidate pixels (cy, x0 - R + c), c = 0..NG*16-1
            const int c4 = C >> 2;
            for (int idx = threadIdx.x; idx < NG * 16 * c4; idx += blockDim.x) {
                const int c = idx / c4, t = idx - c * c4;
                const int cx = x0 - R + c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < NC && cx >= 0 && cx < W) v = reinterpret_cast<const float4 *>(prev + ((size_t)cy * W + cx) * C)[t];
                if (f16) { v.x = aoc_h(v.x); v.y = aoc_h(v.y); v.z = aoc_h(v.z); v.w = aoc_h(v.w); }
                float *d = lds + (size_t)c * RS + t;
                d[0] = v.x; d[TP] = v.y; d[2 * TP] = v.z; d[3 * TP] = v.w;
            }
            const int padn = TP - c4;
            for (int idx = threadIdx.x; idx < NG * 16 * 4 * padn; idx += blockDim.x) {
                const int c = idx / (4 * padn), rem = idx - c * 4 * padn;
                lds[(size_t)c * RS + (rem / padn) * TP + c4 + (rem % padn)] = 0.0f;
            }
            for (int c = threadIdx.x; c < NG * 16; c += blockDim.x) {
                const int cx = x0 - R + c;
                const bool in = c < NC && cx >= 0 && cx < W;
                lbits[c] = in ? (right_bits[(size_t)cy * W + cx] & ~AOC_ROW_KEPT_BIT) : 0u;   // AEM:1023-1028 (pad 0)
            }
        }
        __syncthreads();
        // per-candidate |y|^2 from the staged image (any summation order: tolerance-level)
        for (int c = threadIdx.x; c < NG * 16; c += blockDim.x) {
            const float *r = lds + (size_t)c * RS;
            float s = 0.0f;
            for (int kq = 0; kq < 4; ++kq)
                for (int t = 0; t < (C >> 2); ++t) s += f16 ? aoc_h(r[kq * TP + t] * r[kq * TP + t]) : r[kq * TP + t] * r[kq * TP + t];
            ly2[c] = f16 ? aoc_h(s) : s;
        }
        __syncthreads();
        const int dy = cy - y;
        const int ady = dy < 0 ? -dy : dy;
        if (y < H && ady <= R) {
            for (int gi = 0; gi < NG; ++gi) {
                const float *bstream = lds + (size_t)(gi * 16 + j) * RS + g * TP;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < (TMAX + 3) / 4; ++u) {
                    if (4 * u < TP) {
                        const float4 b = *reinterpret_cast<const float4 *>(bstream + 4 * u);
                        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (4 * u + e < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + e < TMAX ? 4 * u + e : 0], bb[e], acc, 0, 0, 0);
                    }
                }
                const int c = gi * 16 + j;
                const int cx = x0 - R + c;
                uint32_t bits = lbits[c];
                const float y2 = ly2[c];
                if (bits != 0u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qi = g * 4 + r;
                        const int qx = x0 + qi;
                        int dx = cx - qx;
                        dx = dx < 0 ? -dx : dx;
                        if (dx <= R && qx < W && (rate == 1 || (dx % rate == 0 && ady % rate == 0))) {
                            const float d = f16 ? aoc_h(aoc_h(q2r[r] + y2) - 2.0f * aoc_h(acc[r])) : (q2r[r] + y2) - 2.0f * acc[r];   // AEM:961
                            const int cls = lcls[max(ady, dx) / rate];
                            uint32_t b = bits;
                            while (b) {                                     // AEM:1032 where(mask, d, pad)
                                const int o = __builtin_ctz(b);
                                b &= b - 1;
                                if (o < n_obj) lds_fmin(&my_acc[(qi * nr + cls) * n_obj + o], d);
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // prefix-min over rings -> nested windows; channel order [max, r_0, r_1, ...] (AEM:1034-1046)
    if (y < H) {
        for (int idx = lane; idx < 16 * n_obj; idx += 64) {
            const int qi = idx / n_obj, o = idx - qi * n_obj;
            const int qx = x0 + qi;
            if (qx >= W) continue;
            const float bias = obj_bias ? obj_bias[o] : 0.0f;
            float run = INFINITY;
            for (int cls = 0; cls < nr; ++cls) {
                run = fminf(run, my_acc[(qi * nr + cls) * n_obj + o]);
                const int ch = (cls == nr - 1) ? 0 : cls + 1;
                out[(((size_t)o * nr + ch) * H + y) * W + qx] = transform ? aoc_proto_transform(run, bias) : run;   // AEM:1049
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Same computation, laid out for latency: block = ONE query row x 16 columns, and the block's four waves split the
// 2R + 1 candidate rows between them (wave w takes rows cy_beg + w, + 4, ...).  Each wave stages its candidate row
// in a wave-private LDS image (the next row's loads are in flight in registers while the current one is
// multiplied), so there is no workgroup barrier inside the loop, four times as many workgroups (427 instead of
// 112 on a 61 x 107 map) and a critical path of ~7 instead of ~28 candidate rows.  The four waves' per-(pixel,
// ring, object) minima are merged at the end.
template <int TMAX>
__global__ __launch_bounds__(256) void local_window_row_kernel(const float *__restrict__ query, const float *__restrict__ prev,
                                                                const uint32_t *__restrict__ right_bits, int H, int W, int C,
                                                                LocalRadii radii, const float *__restrict__ obj_bias, int n_obj,
                                                                float *__restrict__ out, int transform, int rate, int f16) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TP = (TMAX + 3) / 4 * 4, RS = 4 * TP + 4, NB4 = TP / 4;
    const int RA = radii.r[radii.n - 1];             // window half-size in atrous units
    const int R = RA * rate;
    const float padv = f16 ? aoc_h(AOC_PAD_DISTANCE) : AOC_PAD_DISTANCE;
    const int NC = 16 + 2 * R;                          // candidates per row
    const int NG = (NC + 15) / 16;
    const int NCP = NG * 16;
    const int nr = radii.n;
    const int acc_per_wave = 16 * nr * n_obj;
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    constexpr int c4 = TMAX;                         // launched for C == 4 TMAX only: the piece indices divide by a constant
    // per wave: [NCP][RS] image, [NCP] |y|^2, [NCP] label bits; then [R+1] ring classes and the accumulators
    const size_t wave_floats = (size_t)NCP * RS + 2 * NCP;
    float *wimg = lds + (size_t)wave * wave_floats;
    float *ly2 = wimg + (size_t)NCP * RS;
    uint32_t *lbits = reinterpret_cast<uint32_t *>(ly2 + NCP);
    int32_t *lcls = reinterpret_cast<int32_t *>(lds + 4 * wave_floats);
    float *lacc = reinterpret_cast<float *>(lcls + 32);             // [4 waves][16][n_radii][n_obj]
    float *my_acc = lacc + wave * acc_per_wave;

    const int x0 = blockIdx.x * 16;
    const int y = blockIdx.y;

    for (int i = lane; i < acc_per_wave; i += 64) my_acc[i] = padv;      // AEM:1032 pad
    if ((int)threadIdx.x <= RA) {
        int c = 0;
        while (radii.r[c] < (int)threadIdx.x) ++c;
        lcls[threadIdx.x] = c;
    }
    // stream padding of this wave's image, once
    if constexpr (TP > c4) {
        for (int idx = lane; idx < NCP * 4 * (TP - c4); idx += 64) {
            const int c = idx / (4 * (TP - c4)), rem = idx - c * 4 * (TP - c4);
            wimg[(size_t)c * RS + (rem / (TP - c4)) * TP + c4 + rem % (TP - c4)] = 0.0f;
        }
    }

    float a[TMAX], q2r[4];
    {   // A fragment: the 16 query pixels of row y
        const int qx = min(x0 + j, W - 1);
        const float *src = query + ((size_t)y * W + qx) * C + g;
        float part = 0.0f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            a[t] = (t < c4) ? src[4 * t] : 0.0f;
            if (f16) { a[t] = aoc_h(a[t]); part += aoc_h(a[t] * a[t]); }
            else part += a[t] * a[t];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (f16) part = aoc_h(part);
#pragma unroll
        for (int r = 0; r < 4; ++r) q2r[r] = __shfl(part, g * 4 + r);
    }
    __syncthreads();                                    // lcls

    constexpr int PIECES = (48 * TMAX + 63) / 64;      // float4 pieces per lane per candidate row (up to 48 candidates: R <= 16)
    float4 pv[PIECES];
    uint32_t pbits = 0u;
    auto issue_row = [&](int cy) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int idx = i * 64 + lane;
            const int c = idx / c4, t = idx - c * c4;
            const int cx = x0 - R + c;
            const bool ok = idx < NCP * c4 && c < NC && cx >= 0 && cx < W;
            pv[i] = ok ? reinterpret_cast<const float4 *>(prev + ((size_t)cy * W + cx) * C)[t] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (f16) { pv[i].x = aoc_h(pv[i].x); pv[i].y = aoc_h(pv[i].y); pv[i].z = aoc_h(pv[i].z); pv[i].w = aoc_h(pv[i].w); }
        }
        const int cx = x0 - R + lane;
        pbits = (lane < NC && cx >= 0 && cx < W) ? (right_bits[(size_t)cy * W + cx] & ~AOC_ROW_KEPT_BIT) : 0u;   // AEM:1023-1028 (pad 0)
    };
    auto write_row = [&]() {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int idx = i * 64 + lane;
            if (idx < NCP * c4) {
                const int c = idx / c4, t = idx - c * c4;
                float *d = wimg + (size_t)c * RS + t;
                d[0] = pv[i].x; d[TP] = pv[i].y; d[2 * TP] = pv[i].z; d[3 * TP] = pv[i].w;
            }
        }
        if (lane < NCP) lbits[lane] = pbits;
    };

    const int cy_beg = max(0, y - R), cy_end = min(H - 1, y + R);
    int cy = cy_beg + wave;
    if (cy <= cy_end) issue_row(cy);
    for (; cy <= cy_end; cy += 4) {
        write_row();
        if (cy + 4 <= cy_end) issue_row(cy + 4);       // in flight under this row's arithmetic
        // |y|^2 of the staged candidates (lane = candidate; four independent partial sums)
        if (lane < NCP) {
            const float *r = wimg + (size_t)lane * RS;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const float4 v0 = *reinterpret_cast<const float4 *>(r + 4 * u);
                const float4 v1 = *reinterpret_cast<const float4 *>(r + TP + 4 * u);
                const float4 v2 = *reinterpret_cast<const float4 *>(r + 2 * TP + 4 * u);
                const float4 v3 = *reinterpret_cast<const float4 *>(r + 3 * TP + 4 * u);
                if (f16) {
                    s0 += aoc_h(v0.x * v0.x) + aoc_h(v0.y * v0.y) + aoc_h(v0.z * v0.z) + aoc_h(v0.w * v0.w);
                    s1 += aoc_h(v1.x * v1.x) + aoc_h(v1.y * v1.y) + aoc_h(v1.z * v1.z) + aoc_h(v1.w * v1.w);
                    s2 += aoc_h(v2.x * v2.x) + aoc_h(v2.y * v2.y) + aoc_h(v2.z * v2.z) + aoc_h(v2.w * v2.w);
                    s3 += aoc_h(v3.x * v3.x) + aoc_h(v3.y * v3.y) + aoc_h(v3.z * v3.z) + aoc_h(v3.w * v3.w);
                } else {
                    s0 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
                    s1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
                    s2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
                    s3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
                }
            }
            ly2[lane] = f16 ? aoc_h((s0 + s1) + (s2 + s3)) : (s0 + s1) + (s2 + s3);
        }
        const int dy = cy - y;
        const int ady = dy < 0 ? -dy : dy;
        for (int gi = 0; gi < NG; ++gi) {
            const float *bstream = wimg + (size_t)(gi * 16 + j) * RS + g * TP;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const float4 b = *reinterpret_cast<const float4 *>(bstream + 4 * u);
                const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * u + e < TMAX) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * u + e < TMAX ? 4 * u + e : 0], bb[e], acc, 0, 0, 0);
            }
            const int c = gi * 16 + j;
            const int cx = x0 - R + c;
            const uint32_t bits = lbits[c];
            const float y2 = ly2[c];
            if (bits != 0u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = g * 4 + r;
                    const int qx = x0 + qi;
                    int dx = cx - qx;
                    dx = dx < 0 ? -dx : dx;
                    if (dx <= R && qx < W && (rate == 1 || (dx % rate == 0 && ady % rate == 0))) {
                        const float d = f16 ? aoc_h(aoc_h(q2r[r] + y2) - 2.0f * aoc_h(acc[r])) : (q2r[r] + y2) - 2.0f * acc[r];   // AEM:961
                        const int cls = lcls[max(ady, dx) / rate];
                        uint32_t b = bits;
                        while (b) {                                     // AEM:1032 where(mask, d, pad)
                            const int o = __builtin_ctz(b);
                            b &= b - 1;
                            if (o < n_obj) lds_fmin(&my_acc[(qi * nr + cls) * n_obj + o], d);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // merge the four waves, prefix-min over rings -> nested windows; channel order [max, r_0, r_1, ...] (AEM:1034-1046)
    for (int idx = threadIdx.x; idx < 16 * n_obj; idx += blockDim.x) {
        const int qi = idx / n_obj, o = idx - qi * n_obj;
        const int qx = x0 + qi;
        if (qx >= W) continue;
        const float bias = obj_bias ? obj_bias[o] : 0.0f;
        float run = INFINITY;
        for (int cls = 0; cls < nr; ++cls) {
            const int e = (qi * nr + cls) * n_obj + o;
            const float v = fminf(fminf(lacc[e], lacc[acc_per_wave + e]), fminf(lacc[2 * acc_per_wave + e], lacc[3 * acc_per_wave + e]));
            run = fminf(run, v);
            const int ch = (cls == nr - 1) ? 0 : cls + 1;
            out[(((size_t)o * nr + ch) * H + y) * W + qx] = transform ? aoc_proto_transform(run, bias) : run;   // AEM:1049
        }
    }
}

// ------------------------------------------------------------------------------------------
// Same computation with the MFMA operands fetched STRAIGHT INTO REGISTERS (no LDS image): the k dimension of a dot product may be
// permuted freely as long as both operands agree, so lane (j, g) takes the float4 pieces g, g + 4, g + 8, ... of pixel j's row (and
// channel 16 NP + g when C = 100) -- every load instruction is 16-byte loads of which four neighbouring lanes cover 64 contiguous
// bytes, and a candidate group needs NP (+1) of them instead of a transposing LDS round trip (76 ds_write_b32 + 25 ds_read_b128 per
// lane and candidate row in the row kernel above).  Without the 81 KB of wave-private images a workgroup needs 6.5 KB of LDS (the
// per-(pixel, ring, object) minima), so every wave of the launch is resident at once (the row kernel ran one workgroup per CU = one
// wave per SIMD, nothing to hide a latency behind, and 427 workgroups took two rounds on 256 CUs).
// The 16 query pixels of a workgroup are 2 rows x 8 columns: their windows cover 8 + 2R candidate columns -- exactly two groups of
// 16 at R = 12, 78 % of the computed products used (16 x 1: 16 + 2R = 40 columns = three groups, 52 %) -- and 2 + 2R candidate
// rows, each multiplied against both query rows at once.  The four waves take the candidate rows round-robin; a wave walks its
// (candidate row, group) items with the next item's loads in flight under the current item's MFMAs (two register buffers, loop
// unrolled by two).
template <int TMAX>
__global__ __launch_bounds__(256) void local_window_reg_kernel(const float *__restrict__ query, const float *__restrict__ prev_a,
                                                                const uint32_t *__restrict__ right_bits, int H, int W,
                                                                LocalRadii radii, const float *__restrict__ obj_bias, int n_obj,
                                                                float *__restrict__ out_a, int transform, int rate, int f16,
                                                                const float *__restrict__ prev_b, float *__restrict__ out_b) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // gridDim.z == 2 (aoc_local_window_match_pair): the same query against a second previous-frame map (aocnet.py:328, the per-pixel proxy
    // map) in the same launch
    const float *__restrict__ prev = blockIdx.z ? prev_b : prev_a;
    float *__restrict__ out = blockIdx.z ? out_b : out_a;
    constexpr int C = 4 * TMAX;
    constexpr int NP = TMAX / 4;                        // float4 pieces per lane
    constexpr bool TAIL = (TMAX % 4) != 0;              // TMAX = 25: one more channel per lane (16 NP + g)
    static_assert(TMAX % 4 == 0 || TMAX % 4 == 1, "lane g takes pieces g, g + 4, ... and at most one tail channel");
    const int nr = radii.n;
    const int RA = radii.r[nr - 1];                     // window half-size in atrous units
    const int R = RA * rate;
    const float padv = f16 ? aoc_h(AOC_PAD_DISTANCE) : AOC_PAD_DISTANCE;
    const int NC = 8 + 2 * R;                           // candidate columns of the block
    const int NG = (NC + 15) / 16;
    const int acc_per_wave = 16 * nr * n_obj;
    const int lane = aoc_lane(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    int32_t *lcls = reinterpret_cast<int32_t *>(lds);               // [RA + 1] ring -> class
    float *lacc = lds + 32;                                         // [4 waves][16 queries][n_radii][n_obj]
    float *my_acc = lacc + wave * acc_per_wave;
    const int x0 = blockIdx.x * 8;
    const int y0 = blockIdx.y * 2;

    for (int i = lane; i < acc_per_wave; i += 64) my_acc[i] = padv;      // AEM:1032 pad
    if ((int)threadIdx.x <= RA) {
        int c = 0;
        while (radii.r[c] < (int)threadIdx.x) ++c;
        lcls[threadIdx.x] = c;
    }

    // this lane's 4 NP (+1) channels of a pixel row
    auto load_row = [&](const float *__restrict__ p, float (&v)[TMAX]) {
        const float4 *p4 = reinterpret_cast<const float4 *>(p) + g;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const float4 x = p4[4 * u];
            v[4 * u] = x.x; v[4 * u + 1] = x.y; v[4 * u + 2] = x.z; v[4 * u + 3] = x.w;
        }
        if constexpr (TAIL) v[4 * NP] = p[16 * NP + g];
    };
    // sum of squares of the lane's channels, reduced over the four lanes (g) that share a pixel (any order: tolerance-level)
    auto sq_norm = [&](const float (&v)[TMAX]) -> float {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int t = 0; t + 3 < TMAX; t += 4) {
            if (f16) { s0 += aoc_h(v[t] * v[t]); s1 += aoc_h(v[t + 1] * v[t + 1]); s2 += aoc_h(v[t + 2] * v[t + 2]); s3 += aoc_h(v[t + 3] * v[t + 3]); }
            else {
                s0 = __builtin_fmaf(v[t], v[t], s0); s1 = __builtin_fmaf(v[t + 1], v[t + 1], s1);
                s2 = __builtin_fmaf(v[t + 2], v[t + 2], s2); s3 = __builtin_fmaf(v[t + 3], v[t + 3], s3);
            }
        }
        if constexpr (TAIL) s0 = f16 ? s0 + aoc_h(v[TMAX - 1] * v[TMAX - 1]) : __builtin_fmaf(v[TMAX - 1], v[TMAX - 1], s0);
        float s = (s0 + s1) + (s2 + s3);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        return f16 ? aoc_h(s) : s;
    };

    // query i = 0..15 is pixel (y0 + i / 8, x0 + i % 8); as an MFMA row, lane (j, g) supplies query j; as an MFMA result, register r of
    // lane (j, g) is query 4 g + r against candidate j: its row y0 + g / 2 is the same for the lane's four results
    float a[TMAX], q2r[4];
    {   // A operand (pixels beyond the map re-read the last row / column; they are never stored)
        load_row(query + ((size_t)min(y0 + (j >> 3), H - 1) * W + min(x0 + (j & 7), W - 1)) * C, a);
        if (f16) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) a[t] = aoc_h(a[t]);
        }
        const float q2 = sq_norm(a);
#pragma unroll
        for (int r = 0; r < 4; ++r) q2r[r] = __shfl(q2, g * 4 + r);
    }
    __syncthreads();                                    // lcls
    const int qy = y0 + (g >> 1);                       // query row of this lane's results
    const int qc0 = 4 * (g & 1);                        // ... and their first column (relative to x0)

    // candidate rows cy in [y0 - R, y0 + 1 + R] (clipped); this wave takes cy_beg + wave, + 4, ...
    const int cy_beg = max(0, y0 - R), cy_end = min(H - 1, y0 + 1 + R);
    const int cy_first = cy_beg + wave;
    const int n_items = cy_first <= cy_end ? ((cy_end - cy_first) / 4 + 1) * NG : 0;

    float b0[TMAX], b1[TMAX];
    uint32_t bits0 = 0u, bits1 = 0u;
    int cy_ld = cy_first, gi_ld = 0;                    // item -> (cy, gi), advanced incrementally (wave-uniform)
    auto issue = [&](float (&b)[TMAX], uint32_t &bits) {
        const int c = gi_ld * 16 + j, cx = x0 - R + c;
        const bool ok = c < NC && cx >= 0 && cx < W;
        const int cxc = min(max(cx, 0), W - 1);                       // clamped address, selected afterwards: the loads stay branch-free
        const size_t pix = (size_t)cy_ld * W + cxc;
        load_row(prev + pix * C, b);
        const uint32_t raw = right_bits[pix];
        bits = ok ? (raw & ~AOC_ROW_KEPT_BIT) : 0u;                    // AEM:1023-1028 (pad 0)
        if (++gi_ld == NG) { gi_ld = 0; cy_ld += 4; }
    };
    int cy_cur = cy_first, gi_cur = 0;
    auto compute = [&](float (&b)[TMAX], uint32_t bits) {
        if (f16) {
#pragma unroll
            for (int t = 0; t < TMAX; ++t) b[t] = aoc_h(b[t]);
        }
        const float y2 = sq_norm(b);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
        int ady = cy_cur - qy;
        ady = ady < 0 ? -ady : ady;
        bool row_on = ady <= R && qy < H;
        int aky = ady;
        if (rate != 1) { aky = ady / rate; row_on = row_on && aky * rate == ady; }
        const int cq = gi_cur * 16 + j - R - qc0;                      // cx - qx for r = 0
        if (bits != 0u && row_on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int dx = cq - r;
                dx = dx < 0 ? -dx : dx;
                bool on = dx <= R && x0 + qc0 + r < W;
                int akx = dx;
                if (rate != 1) { akx = dx / rate; on = on && akx * rate == dx; }
                if (on) {
                    const float d = f16 ? aoc_h(aoc_h(q2r[r] + y2) - 2.0f * aoc_h(acc[r])) : (q2r[r] + y2) - 2.0f * acc[r];   // AEM:961
                    const int cls = lcls[max(aky, akx)];
                    uint32_t bb = bits;
                    while (bb) {                                        // AEM:1032 where(mask, d, pad)
                        const int o = __builtin_ctz(bb);
                        bb &= bb - 1;
                        if (o < n_obj) lds_fmin(&my_acc[((g * 4 + r) * nr + cls) * n_obj + o], d);
                    }
                }
            }
        }
        if (++gi_cur == NG) { gi_cur = 0; cy_cur += 4; }
    };

    if (n_items > 0) issue(b0, bits0);
    for (int it = 0; it < n_items; it += 2) {
        if (it + 1 < n_items) issue(b1, bits1);
        compute(b0, bits0);
        if (it + 1 < n_items) {
            if (it + 2 < n_items) issue(b0, bits0);
            compute(b1, bits1);
        }
    }
    __syncthreads();
    // merge the four waves, prefix-min over rings -> nested windows; channel order [max, r_0, r_1, ...] (AEM:1034-1046)
    for (int idx = threadIdx.x; idx < 16 * n_obj; idx += blockDim.x) {
        const int qi = idx / n_obj, o = idx - qi * n_obj;
        const int oy = y0 + (qi >> 3), ox = x0 + (qi & 7);
        if (ox >= W || oy >= H) continue;
        const float bias = obj_bias ? obj_bias[o] : 0.0f;
        float run = INFINITY;
        for (int cls = 0; cls < nr; ++cls) {
            const int e = (qi * nr + cls) * n_obj + o;
            const float v = fminf(fminf(lacc[e], lacc[acc_per_wave + e]), fminf(lacc[2 * acc_per_wave + e], lacc[3 * acc_per_wave + e]));
            run = fminf(run, v);
            const int ch = (cls == nr - 1) ? 0 : cls + 1;
            out[(((size_t)o * nr + ch) * H + oy) * W + ox] = transform ? aoc_proto_transform(run, bias) : run;   // AEM:1049
        }
    }
}

// ------------------------------------------------------------------------------------------
// Resize helpers (torch semantics, fp32).
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in_size, int &i0, int &i1, float &l0, float &l1) {
    const float real = scale * (float)dst;                 // align_corners=True: area_pixel_compute_source_index
    i0 = (int)real;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    float lam = real - (float)i0;
    lam = fminf(fmaxf(lam, 0.0f), 1.0f);
    l1 = lam;
    l0 = 1.0f - lam;
}

__global__ __launch_bounds__(256) void resize_bilinear_hwc_kernel(const float *__restrict__ in, int h, int w, int C,
                                                                   float *__restrict__ out, int H, int W, float sh, float sw, int f16) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t pix = idx / C;
    const int X = (int)(pix % W), Y = (int)(pix / W);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    bilinear_src(Y, sh, h, y0, y1, hy0, hy1);
    bilinear_src(X, sw, w, x0, x1, wx0, wx1);
    const float v00 = in[((size_t)y0 * w + x0) * C + c], v01 = in[((size_t)y0 * w + x1) * C + c];
    const float v10 = in[((size_t)y1 * w + x0) * C + c], v11 = in[((size_t)y1 * w + x1) * C + c];
    if (f16) {
        // F.interpolate on a float16 tensor (AEM:938-941 after `.half()`): float16 samples, fp32 arithmetic, ONE rounding to float16.  The
        // fp32 value is a float16 tie surprisingly often (weights like 0.9 x 11-bit samples), so the association order matters: this is
        // torch-CPU's channels-last kernel term by term -- the four corner weights as fp32 products, then a fused multiply-add chain that
        // runs from the last corner to the first inside its vector body (32 float16 channels per AVX-512 vector) and from the first to the
        // last in its scalar tail (channels >= C - C % 32).  Found by enumerating the orders against F.interpolate: 0 of 27 300 differ.
        // NB this is the association order of torch-CPU on an AVX-512 host -- the machine the golden vectors were generated on (and the CPU
        // oracle runs on) -- not of the reference's CUDA run, which evaluates h0 (w0 a + w1 b) + h1 (w0 c + w1 d); an AVX2 host vectorises 16
        // float16 lanes and moves the body / tail boundary.  The float16 mode is therefore pinned to one float16 ulp of any of these orders
        // (tests: 2e-6 on the generating host's goldens, one ulp = 4e-3 stated in DESIGN 2 as the cross-host tolerance).
        const float a = aoc_h(v00), b = aoc_h(v01), cc = aoc_h(v10), d = aoc_h(v11);
        const float w00 = hy0 * wx0, w01 = hy0 * wx1, w10 = hy1 * wx0, w11 = hy1 * wx1;
        float acc;
        if (c < C - C % 32) {
            acc = w11 * d;
            acc = __builtin_fmaf(w10, cc, acc);
            acc = __builtin_fmaf(w01, b, acc);
            acc = __builtin_fmaf(w00, a, acc);
        } else {
            acc = w00 * a;
            acc = __builtin_fmaf(w01, b, acc);
            acc = __builtin_fmaf(w10, cc, acc);
            acc = __builtin_fmaf(w11, d, acc);
        }
        out[idx] = aoc_h(acc);
        return;
    }
    out[idx] = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
}

__global__ __launch_bounds__(256) void resize_bilinear_planes_kernel(const float *__restrict__ in, int P, int h, int w,
                                                                      float *__restrict__ out, int H, int W, float sh, float sw,
                                                                      int inner_count, int64_t outer_stride, int64_t plane_stride,
                                                                      int64_t pixel_stride, int outer_count, int64_t group_stride) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)P * H * W;
    if (idx >= total) return;
    const int X = (int)(idx % W);
    const int Y = (int)((idx / W) % H);
    const int p = (int)(idx / ((int64_t)W * H));
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    bilinear_src(Y, sh, h, y0, y1, hy0, hy1);
    bilinear_src(X, sw, w, x0, x1, wx0, wx1);
    const float *ip = in + (size_t)p * h * w;
    const float v00 = ip[(size_t)y0 * w + x0], v01 = ip[(size_t)y0 * w + x1];
    const float v10 = ip[(size_t)y1 * w + x0], v11 = ip[(size_t)y1 * w + x1];
    // plane p = (group, outer, inner): groups of outer_count * inner_count planes land group_stride apart (two local-matching results ->
    // two channel ranges of the proto-mask tensor in one launch)
    const int po = p / inner_count;
    out[(po / outer_count) * group_stride + (po % outer_count) * outer_stride + (p % inner_count) * plane_stride + ((int64_t)Y * W + X) * pixel_stride] = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
}

// out[y', x', :] = in[rate * y', rate * x', :]  (AEM:533-579: the atrous grid of the reference pool; X floats per pixel, X % 4 == 0 takes 16-byte moves)
__global__ __launch_bounds__(256) void atrous_subsample_kernel(const float *__restrict__ in, int w, int X, int rate, float *__restrict__ out, int H, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((X & 3) == 0) {
        const int x4 = X >> 2;
        if (idx >= (int64_t)H * W * x4) return;
        const int c = (int)(idx % x4);
        const int64_t pix = idx / x4;
        const int xo = (int)(pix % W), yo = (int)(pix / W);
        reinterpret_cast<float4 *>(out)[idx] = reinterpret_cast<const float4 *>(in)[((size_t)yo * rate * w + (size_t)xo * rate) * x4 + c];
    } else {
        if (idx >= (int64_t)H * W * X) return;
        const int c = (int)(idx % X);
        const int64_t pix = idx / X;
        const int xo = (int)(pix % W), yo = (int)(pix / W);
        out[idx] = in[((size_t)yo * rate * w + (size_t)xo * rate) * X + c];
    }
}

__global__ __launch_bounds__(256) void local_prep_kernel(const float *__restrict__ cur, const float *__restrict__ prev, const float *__restrict__ lab,
                                                          const float *__restrict__ rows, int h, int w, int C, int n_obj, float *__restrict__ q2,
                                                          float *__restrict__ p2, float *__restrict__ pm2, uint32_t *__restrict__ bits2, int H, int W,
                                                          float sh, float sw, float nsh, float nsw, const float *__restrict__ obj_bias, int n_pair_sets,
                                                          float *__restrict__ set_bias_out, const float *__restrict__ csa, float *__restrict__ cda, int nca,
                                                          const float *__restrict__ csb, float *__restrict__ cdb, int ncb) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the small tables first (any thread range will do: they only depend on this launch's inputs)
    if (set_bias_out && idx < n_pair_sets + n_obj) {
        const int o = idx < n_pair_sets ? (int)((idx >> 1) % n_obj) : (int)(idx - n_pair_sets);
        set_bias_out[idx] = obj_bias ? obj_bias[o] : 0.0f;
    }
    if (idx < nca) cda[idx] = csa[idx];
    if (idx < ncb) cdb[idx] = csb[idx];
    const int64_t total = (int64_t)H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int64_t pix = idx / C;
    const int X = (int)(pix % W), Y = (int)(pix / W);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    bilinear_src(Y, sh, h, y0, y1, hy0, hy1);
    bilinear_src(X, sw, w, x0, x1, wx0, wx1);
    const size_t p00 = (size_t)y0 * w + x0, p01 = (size_t)y0 * w + x1, p10 = (size_t)y1 * w + x0, p11 = (size_t)y1 * w + x1;
    {
        const float v00 = cur[p00 * C + c], v01 = cur[p01 * C + c], v10 = cur[p10 * C + c], v11 = cur[p11 * C + c];
        q2[idx] = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
    }
    {
        const float v00 = prev[p00 * C + c], v01 = prev[p01 * C + c], v10 = prev[p10 * C + c], v11 = prev[p11 * C + c];
        p2[idx] = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
    }
    {
        // label_mix_kernel's sum at the four corners (objects in order), then the bilinear expression
        float v00 = 0.0f, v01 = 0.0f, v10 = 0.0f, v11 = 0.0f;
        for (int o = 0; o < n_obj; ++o) {
            const float r = rows[(size_t)o * C + c];
            v00 += lab[p00 * n_obj + o] * r;
            v01 += lab[p01 * n_obj + o] * r;
            v10 += lab[p10 * n_obj + o] * r;
            v11 += lab[p11 * n_obj + o] * r;
        }
        pm2[idx] = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
    }
    if (c == 0) {
        // label_bits_kernel at the nearest source pixel (torch nearest: floor(dst * in / out))
        const int sy = min((int)floorf((float)Y * nsh), h - 1);
        const int sx = min((int)floorf((float)X * nsw), w - 1);
        const float *l = lab + ((size_t)sy * w + sx) * n_obj;
        uint32_t right = 0;
        float sum = 0.0f;
        for (int o = 0; o < n_obj; ++o) {
            const float v = l[o];
            sum += v;
            if (v > 0.9f) right |= 1u << o;
        }
        if (sum > 0.9f) right |= AOC_ROW_KEPT_BIT;
        bits2[pix] = right;
    }
}

__global__ __launch_bounds__(256) void resize_nearest_bits_kernel(const uint32_t *__restrict__ in, int h, int w,
                                                                   uint32_t *__restrict__ out, int H, int W, float sh, float sw) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int X = idx % W, Y = idx / W;
    const int sy = min((int)floorf((float)Y * sh), h - 1);   // torch nearest: floor(dst * in/out)
    const int sx = min((int)floorf((float)X * sw), w - 1);
    out[idx] = in[(size_t)sy * w + sx];
}

}  // namespace

extern "C" {

int aoc_local_window_match(const float *query, const float *prev, const uint32_t *right_bits, int H, int W, int C,
                           const int32_t *radii_host, int n_radii, const float *obj_bias, int n_obj,
                           float *out, int transform, aoc_stream_t stream) {
    return aoc_local_window_match_ex(query, prev, right_bits, H, W, C, radii_host, n_radii, obj_bias, n_obj, out, transform, 1, 0, stream);
}

static int local_window_match_impl(const float *query, const float *prev, const uint32_t *right_bits, int H, int W, int C,
                                   const int32_t *radii_host, int n_radii, const float *obj_bias, int n_obj,
                                   float *out, int transform, int atrous_rate, int float16, aoc_stream_t stream, const float *prev2, float *out2);

int aoc_local_window_match_ex(const float *query, const float *prev, const uint32_t *right_bits, int H, int W, int C,
                              const int32_t *radii_host, int n_radii, const float *obj_bias, int n_obj,
                              float *out, int transform, int atrous_rate, int float16, aoc_stream_t stream) {
    return local_window_match_impl(query, prev, right_bits, H, W, C, radii_host, n_radii, obj_bias, n_obj, out, transform, atrous_rate, float16, stream,
                                   nullptr, nullptr);
}

int aoc_local_window_match_pair(const float *query, const float *prev_a, const float *prev_b, const uint32_t *right_bits, int H, int W, int C,
                                const int32_t *radii_host, int n_radii, const float *obj_bias, int n_obj, float *out_a, float *out_b,
                                int transform, aoc_stream_t stream) {
    if (!prev_b || !out_b) return AOC_ERR_INVALID_ARG;
    if (C == 100 || C == 128)      // the register-operand kernel takes both maps as one launch (grid z)
        return local_window_match_impl(query, prev_a, right_bits, H, W, C, radii_host, n_radii, obj_bias, n_obj, out_a, transform, 1, 0, stream, prev_b, out_b);
    const int rc = local_window_match_impl(query, prev_a, right_bits, H, W, C, radii_host, n_radii, obj_bias, n_obj, out_a, transform, 1, 0, stream, nullptr, nullptr);
    if (rc != AOC_OK) return rc;
    return local_window_match_impl(query, prev_b, right_bits, H, W, C, radii_host, n_radii, obj_bias, n_obj, out_b, transform, 1, 0, stream, nullptr, nullptr);
}

static int local_window_match_impl(const float *query, const float *prev, const uint32_t *right_bits, int H, int W, int C,
                                   const int32_t *radii_host, int n_radii, const float *obj_bias, int n_obj,
                                   float *out, int transform, int atrous_rate, int float16, aoc_stream_t stream, const float *prev2, float *out2) {
    if (!query || !prev || !right_bits || !radii_host || !out) return AOC_ERR_INVALID_ARG;
    if (H < 1 || W < 1 || C < 4 || n_radii < 1 || n_obj < 1 || atrous_rate < 1) return AOC_ERR_INVALID_ARG;
    if ((C & 3) || C > 128 || n_radii > LM_MAX_RADII || n_obj > AOC_MAX_OBJECTS) return AOC_ERR_UNSUPPORTED;
    LocalRadii radii;
    radii.n = n_radii;
    // window radii in units of the atrous rate: AEM:949 pad_max_distance = max - max % rate, AEM:1039 local_dis // rate
    for (int i = 0; i < n_radii; ++i) {
        if (radii_host[i] < 0 || (i > 0 && radii_host[i] <= radii_host[i - 1])) return AOC_ERR_INVALID_ARG;
        radii.r[i] = radii_host[i] / atrous_rate;
    }
    for (int i = n_radii; i < LM_MAX_RADII; ++i) radii.r[i] = radii.r[n_radii - 1];
    const int rate = atrous_rate, f16 = float16 ? 1 : 0;
    const int R = radii.r[n_radii - 1] * atrous_rate;
    if (R > 31) return AOC_ERR_UNSUPPORTED;
    const int RS = aoc_tile_row_stride(C);
    const int NG = (16 + 2 * R + 15) / 16;
    const size_t lds = (size_t)NG * 16 * RS * sizeof(float) + (size_t)NG * 16 * 8 + 32 * sizeof(int32_t) +
                       (size_t)4 * 16 * n_radii * n_obj * sizeof(float);
    if (lds > 150 * 1024) return AOC_ERR_UNSUPPORTED;
    hipStream_t st = aoc_hip_stream(stream);
    static const char *which = AOC_DEV_ENV("AOC_LOCAL_KERNEL");            // developer switch: "row" / "block" = the LDS-image kernels
    // register-operand kernel (no LDS image): C == 100 / 128
    if ((C == 100 || C == 128) && (prev2 || !(which && (strcmp(which, "row") == 0 || strcmp(which, "block") == 0)))) {
        const dim3 rgrid((W + 7) / 8, (H + 1) / 2, prev2 ? 2 : 1);
        const size_t lds_reg = (32 + (size_t)4 * 16 * n_radii * n_obj) * sizeof(float);
        if (C == 100)
            hipLaunchKernelGGL(local_window_reg_kernel<25>, rgrid, dim3(256), lds_reg, st, query, prev, right_bits, H, W, radii, obj_bias, n_obj, out, transform, rate, f16, prev2, out2);
        else
            hipLaunchKernelGGL(local_window_reg_kernel<32>, rgrid, dim3(256), lds_reg, st, query, prev, right_bits, H, W, radii, obj_bias, n_obj, out, transform, rate, f16, prev2, out2);
        AOC_RETURN_IF_LAUNCH_FAILED();
        return AOC_OK;
    }
    // row-per-block layout (wave-private candidate images) whenever it fits: C == 100 / 128 tiles, R <= 16
    {
        const int TPc = (C == 100) ? 28 : 32, RSc = 4 * TPc + 4;
        const size_t lds_row = ((size_t)4 * ((size_t)NG * 16 * RSc + 2 * NG * 16) + 32 + (size_t)4 * 16 * n_radii * n_obj) * sizeof(float);
        static const bool use_row = !(AOC_DEV_ENV("AOC_LOCAL_KERNEL") && strcmp(AOC_DEV_ENV("AOC_LOCAL_KERNEL"), "block") == 0);   // developer switch
        if (use_row && (C == 100 || C == 128) && R <= 16 && lds_row <= 150 * 1024) {
            const dim3 rgrid((W + 15) / 16, H);
            if (C == 100)
                hipLaunchKernelGGL(local_window_row_kernel<25>, rgrid, dim3(256), lds_row, st, query, prev, right_bits, H, W, C, radii, obj_bias, n_obj, out, transform, rate, f16);
            else
                hipLaunchKernelGGL(local_window_row_kernel<32>, rgrid, dim3(256), lds_row, st, query, prev, right_bits, H, W, C, radii, obj_bias, n_obj, out, transform, rate, f16);
            AOC_RETURN_IF_LAUNCH_FAILED();
            return AOC_OK;
        }
    }
    const dim3 grid((W + 15) / 16, (H + 3) / 4);
    if (C == 100)
        hipLaunchKernelGGL(local_window_kernel<25>, grid, dim3(256), lds, st, query, prev, right_bits, H, W, C, radii, obj_bias, n_obj, out, transform, rate, f16);
    else
        hipLaunchKernelGGL(local_window_kernel<32>, grid, dim3(256), lds, st, query, prev, right_bits, H, W, C, radii, obj_bias, n_obj, out, transform, rate, f16);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

static inline float align_corners_scale(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
}

int aoc_resize_bilinear_hwc(const float *in, int h, int w, int C, float *out, int H, int W, aoc_stream_t stream) {
    return aoc_resize_bilinear_hwc_ex(in, h, w, C, out, H, W, 0, stream);
}

int aoc_resize_bilinear_hwc_ex(const float *in, int h, int w, int C, float *out, int H, int W, int float16, aoc_stream_t stream) {
    if (!in || !out || h < 1 || w < 1 || C < 1 || H < 1 || W < 1) return AOC_ERR_INVALID_ARG;
    const int64_t total = (int64_t)H * W * C;
    hipLaunchKernelGGL(resize_bilinear_hwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), in, h, w, C,
                       out, H, W, align_corners_scale(h, H), align_corners_scale(w, W), float16 ? 1 : 0);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_resize_bilinear_planes(const float *in, int P, int h, int w, float *out, int H, int W, int inner_count,
                               int64_t out_outer_stride, int64_t out_plane_stride, int64_t out_pixel_stride, aoc_stream_t stream) {
    return aoc_resize_bilinear_planes_grouped(in, P, h, w, out, H, W, inner_count, P > 0 && inner_count > 0 ? (P + inner_count - 1) / inner_count : 1, 0,
                                              out_outer_stride, out_plane_stride, out_pixel_stride, stream);
}

int aoc_resize_bilinear_planes_grouped(const float *in, int P, int h, int w, float *out, int H, int W, int inner_count, int outer_count,
                                       int64_t out_group_stride, int64_t out_outer_stride, int64_t out_plane_stride, int64_t out_pixel_stride,
                                       aoc_stream_t stream) {
    if (!in || !out || P < 1 || h < 1 || w < 1 || H < 1 || W < 1 || inner_count < 1 || outer_count < 1) return AOC_ERR_INVALID_ARG;
    const int64_t total = (int64_t)P * H * W;
    hipLaunchKernelGGL(resize_bilinear_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), in, P, h, w,
                       out, H, W, align_corners_scale(h, H), align_corners_scale(w, W), inner_count, out_outer_stride, out_plane_stride, out_pixel_stride,
                       outer_count, out_group_stride);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

// The half-resolution operands of both local matchings in ONE launch (aocnet.py:255, 325-337 with MODEL_LOCAL_DOWNSAMPLE; AEM:938-941):
// q2 / p2 = bilinear (align_corners) down-samples of the current and the previous frame's embedding, pm2 = the same of the per-pixel proxy
// map matmul(prev label, prev_head_pos) (never materialised at full resolution), bits2 = the nearest-neighbour down-sample of the previous
// frame's "right for object o" bit mask.  Every value is computed by the expressions of aoc_resize_bilinear_hwc / aoc_label_mix /
// aoc_label_bits / aoc_resize_nearest_bits, term by term: the outputs equal those four calls' bit for bit.  The launch also fills small
// per-frame tables when asked to: the per-set bias table of the correlation launch (set s < n_pair_sets belongs to object (s / 2) % n_obj,
// the others to s - n_pair_sets) and copies of `n_copy` floats (the pooled reference heads into the k = 1 rows of the proxy table).
int aoc_local_prep(const float *cur_emb, const float *prev_emb, const float *prev_labels, const float *prev_pos, int h, int w, int C, int n_obj,
                   float *q2, float *p2, float *pm2, uint32_t *bits2, int H2, int W2,
                   const float *obj_bias, int n_pair_sets, float *set_bias_out,
                   const float *copy_src_a, float *copy_dst_a, int n_copy_a, const float *copy_src_b, float *copy_dst_b, int n_copy_b,
                   aoc_stream_t stream) {
    if (!cur_emb || !prev_emb || !prev_labels || !prev_pos || !q2 || !p2 || !pm2 || !bits2) return AOC_ERR_INVALID_ARG;
    if (h < 1 || w < 1 || C < 1 || n_obj < 1 || H2 < 1 || W2 < 1 || n_pair_sets < 0 || n_copy_a < 0 || n_copy_b < 0) return AOC_ERR_INVALID_ARG;
    if (n_obj > AOC_MAX_OBJECTS) return AOC_ERR_UNSUPPORTED;
    // one thread per element of the half-resolution maps; the same threads also write the per-set bias table and the two piggy-backed copies
    // (idx < n): the launch has to be at least as wide as the longest of those tables (tiny maps with many objects)
    int64_t total = (int64_t)H2 * W2 * C;
    total = std::max<int64_t>(total, std::max<int64_t>(std::max(n_copy_a, n_copy_b), (int64_t)n_pair_sets + n_obj));
    hipLaunchKernelGGL(local_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), cur_emb, prev_emb, prev_labels,
                       prev_pos, h, w, C, n_obj, q2, p2, pm2, bits2, H2, W2, align_corners_scale(h, H2), align_corners_scale(w, W2),
                       (float)h / (float)H2, (float)w / (float)W2, obj_bias, n_pair_sets, set_bias_out, copy_src_a, copy_dst_a, n_copy_a, copy_src_b,
                       copy_dst_b, n_copy_b);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_atrous_subsample(const float *in, int h, int w, int X, int rate, float *out, aoc_stream_t stream) {
    if (!in || !out || h < 1 || w < 1 || X < 1 || rate < 1) return AOC_ERR_INVALID_ARG;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) && (X & 3) == 0) return AOC_ERR_INVALID_ARG;
    const int H = (h + rate - 1) / rate, W = (w + rate - 1) / rate;
    const int64_t total = (int64_t)H * W * ((X & 3) == 0 ? X >> 2 : X);
    hipLaunchKernelGGL(atrous_subsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), in, w, X, rate, out, H, W);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_resize_nearest_bits(const uint32_t *in, int h, int w, uint32_t *out, int H, int W, aoc_stream_t stream) {
    if (!in || !out || h < 1 || w < 1 || H < 1 || W < 1) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(resize_nearest_bits_kernel, dim3((unsigned)((H * W + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), in, h, w, out, H, W,
                       (float)h / (float)H, (float)w / (float)W);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

}  // extern "C"
