// Streaming kernels of the mask-calibration side: fg->bg min (AEM:9-23), k = 1 proxy pooling
// (ATT:134-189), FiLM gate (ATT:12-17, CLB:81-84) and the conditioning-layer gate + pool (CL:23-43).
// All of them are HBM-bound: one coalesced pass over the big operand, reductions in LDS/registers.
#include <stdlib.h>

#include "aoc_common.h"

namespace {

// ------------------------------------------------------------------------------------------ fg2bg
// dis [n_obj, n_ch, inner] -> out [n_obj, 1, inner]: the reference concatenates the other objects'
// maps along dim 1 and takes the min over that dim (AEM:18-20), i.e. over (other objects x channels).
__global__ __launch_bounds__(256) void fg2bg_kernel(const float *__restrict__ dis, int n_obj, int n_ch, int64_t inner, int64_t dis_obj_stride,
                                                    float *__restrict__ out, int64_t out_obj_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= inner) return;
    float m1 = INFINITY, m2 = INFINITY;
    int arg = -1;
    for (int o = 0; o < n_obj; ++o) {
        float v = INFINITY;
        for (int c = 0; c < n_ch; ++c) v = fminf(v, dis[(size_t)o * dis_obj_stride + (size_t)c * inner + i]);
        if (v < m1) { m2 = m1; m1 = v; arg = o; }
        else if (v < m2) { m2 = v; }
    }
    for (int o = 0; o < n_obj; ++o) out[(size_t)o * out_obj_stride + i] = (o == arg) ? m2 : m1;   // min over the OTHER objects
}

// The tail of a frame's proto-mask tensor in ONE launch (aocnet.py:349-358): the background maps of the local-matching channels and of
// the dense channel (two fg2bg_kernel passes: min over the OTHER objects; a single object keeps its own map, AEM:10-11), the previous-frame
// mask channel, and the attention head [O, 4C] = (ref_pos | ref_neg | prev_pos | prev_neg) (ATT:188).  feat is [O, n_ch, hw] with object
// stride obj_stride; channel indices < 0 switch a part off.
__global__ __launch_bounds__(256) void proto_finish_kernel(float *__restrict__ feat, int n_obj, int64_t hw, int64_t obj_stride, int ch_local, int n_local,
                                                           int ch_local_bg, int ch_global, int ch_global_bg, int ch_prev, const float *__restrict__ prev_labels,
                                                           const float *__restrict__ ref_pos, const float *__restrict__ ref_neg,
                                                           const float *__restrict__ prev_pos, const float *__restrict__ prev_neg, int C,
                                                           float *__restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (head && i < (int64_t)n_obj * 4 * C) {
        const int o = (int)(i / (4 * C)), r = (int)(i - (int64_t)o * 4 * C), part = r / C, c = r - part * C;
        const float *src = part == 0 ? ref_pos : part == 1 ? ref_neg : part == 2 ? prev_pos : prev_neg;
        head[i] = src[(size_t)o * C + c];
    }
    auto fg2bg = [&](const float *dis, float *out, int64_t idx) {
        if (n_obj == 1) { out[idx] = dis[idx]; return; }
        float m1 = INFINITY, m2 = INFINITY;
        int arg = -1;
        for (int o = 0; o < n_obj; ++o) {
            float v = INFINITY;
            v = fminf(v, dis[(size_t)o * obj_stride + idx]);
            if (v < m1) { m2 = m1; m1 = v; arg = o; }
            else if (v < m2) { m2 = v; }
        }
        for (int o = 0; o < n_obj; ++o) out[(size_t)o * obj_stride + idx] = (o == arg) ? m2 : m1;
    };
    if (ch_local_bg >= 0 && i < (int64_t)n_local * hw) fg2bg(feat + (size_t)ch_local * hw, feat + (size_t)ch_local_bg * hw, i);
    if (i < hw) {
        if (ch_global_bg >= 0) fg2bg(feat + (size_t)ch_global * hw, feat + (size_t)ch_global_bg * hw, i);
        if (ch_prev >= 0)
            for (int o = 0; o < n_obj; ++o) feat[(size_t)o * obj_stride + (size_t)ch_prev * hw + i] = prev_labels[(size_t)i * n_obj + o];
    }
}

// ------------------------------------------------------------------------------------------ pooling
constexpr int MP_PIX = 128;      // pixels per block
constexpr int MP_OMAX = 32;
constexpr int MP_SPLIT = 4;      // pixel sub-ranges per block (threads = MP_SPLIT x 64 channel lanes)

// partial[blk][o][c] = sum_{p in chunk} emb[p,c] * lab[o,p]  (o < O), partial[blk][O][c] = sum emb[p,c];
// pcount[blk][o] = sum lab[o,p].   emb [F, hw, C] channel-last, lab [F, O, hw] (or [F, hw, O] pixel-major).
// Block = 256 threads = MP_SPLIT pixel sub-ranges x 64 channel lanes (lane handles channels c, c+64, ...);
// loads are coalesced along channels and unrolled along pixels for memory-level parallelism.
// OM: compile-time bound of the object loop (4, 8 or MP_OMAX): the accumulators live in registers and the per-object work is not
// 32 predicated iterations for a 4-object frame
template <int OM>
__global__ __launch_bounds__(256) void masked_pool_partial_kernel(const float *__restrict__ emb, const float *__restrict__ lab,
                                                                   int64_t hw, int C, int n_obj, int chunks_per_frame, int pixel_major,
                                                                   float *__restrict__ partial, float *__restrict__ pcount) {
    extern __shared__ float llab[];   // [n_obj][MP_PIX] labels, then [MP_SPLIT][(n_obj+1)][64] combine buffer
    float *lcomb = llab + n_obj * MP_PIX;
    const int f = blockIdx.x / chunks_per_frame, chunk = blockIdx.x - f * chunks_per_frame;
    const int64_t p0 = (int64_t)chunk * MP_PIX;
    const int np = (int)min((int64_t)MP_PIX, hw - p0);
    for (int i = threadIdx.x; i < n_obj * MP_PIX; i += blockDim.x) {
        const int o = i / MP_PIX, p = i - o * MP_PIX;
        llab[i] = (p >= np) ? 0.0f : (pixel_major ? lab[((size_t)f * hw + p0 + p) * n_obj + o] : lab[((size_t)f * n_obj + o) * hw + p0 + p]);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int pb = sub * (MP_PIX / MP_SPLIT), pe = min(np, pb + MP_PIX / MP_SPLIT);
    const float *e = emb + ((size_t)f * hw + p0) * C;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float acc[OM + 1];
#pragma unroll
        for (int o = 0; o <= OM; ++o) acc[o] = 0.0f;
        if (c < C) {
            for (int p = pb; p < pe; p += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (p + u < pe) ? e[(size_t)(p + u) * C + c] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc[OM] += v[u];
#pragma unroll
                    for (int o = 0; o < OM; ++o)
                        if (o < n_obj) acc[o] += v[u] * llab[o * MP_PIX + min(p + u, MP_PIX - 1)];
                }
            }
        }
        // combine the MP_SPLIT sub-ranges in a fixed order (deterministic)
        __syncthreads();
#pragma unroll
        for (int o = 0; o < OM; ++o)
            if (o < n_obj) lcomb[(sub * (n_obj + 1) + o) * 64 + lane] = acc[o];
        lcomb[(sub * (n_obj + 1) + n_obj) * 64 + lane] = acc[OM];
        __syncthreads();
        if (sub == 0 && c < C) {
            for (int o = 0; o <= n_obj; ++o) {
                float t = 0.0f;
                for (int q = 0; q < MP_SPLIT; ++q) t += lcomb[(q * (n_obj + 1) + o) * 64 + lane];
                partial[((size_t)blockIdx.x * (n_obj + 1) + o) * C + c] = t;
            }
        }
    }
    if ((int)threadIdx.x < n_obj) {
        float s = 0.0f;
        for (int p = 0; p < np; ++p) s += llab[threadIdx.x * MP_PIX + p];
        pcount[(size_t)blockIdx.x * n_obj + threadIdx.x] = s;
    }
}

// The same partial sums with 16-byte loads (C % 4 == 0, up to 8 objects): thread t owns the float4 piece t % (C/4) of the pixels
// t / (C/4), + RPP, ... of the chunk (RPP = 256 / (C/4) pixels per pass; consecutive threads read one pixel's consecutive 16 bytes), four
// loads in flight per thread; the RPP partial sums of a piece are combined through LDS in a fixed order (deterministic).
template <int OM>
__global__ __launch_bounds__(256) void masked_pool_partial4_kernel(const float *__restrict__ emb, const float *__restrict__ lab,
                                                                    int64_t hw, int C, int n_obj, int chunks_per_frame, int cpb, int pixel_major,
                                                                    float *__restrict__ partial, float *__restrict__ pcount) {
    extern __shared__ __attribute__((aligned(16))) float llab4[];   // [n_obj][MP_PIX] labels, then [RPP][n_obj + 1][C / 4] float4 combine buffer
    float *llab = llab4;
    float4 *lcomb = reinterpret_cast<float4 *>(llab4 + n_obj * MP_PIX);
    // block = cpb consecutive chunks of one frame (large pools: fewer, fatter partial blocks for the final kernel to add up)
    const int bpf = (chunks_per_frame + cpb - 1) / cpb;
    const int f = blockIdx.x / bpf, cb = blockIdx.x - f * bpf;
    const int c4 = C >> 2, rpp = 256 / c4;
    const int r0 = threadIdx.x / c4, piece = threadIdx.x - r0 * c4;
    const bool worker = r0 < rpp;
    float4 acc[OM + 1];
#pragma unroll
    for (int o = 0; o <= OM; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    float cnt = 0.0f;
    for (int chunk = cb * cpb; chunk < min(chunks_per_frame, (cb + 1) * cpb); ++chunk) {
        const int64_t p0 = (int64_t)chunk * MP_PIX;
        const int np = (int)min((int64_t)MP_PIX, hw - p0);
        __syncthreads();                                   // the previous chunk's labels are no longer read
        for (int i = threadIdx.x; i < n_obj * MP_PIX; i += 256) {
            const int o = i / MP_PIX, p = i - o * MP_PIX;
            llab[i] = (p >= np) ? 0.0f : (pixel_major ? lab[((size_t)f * hw + p0 + p) * n_obj + o] : lab[((size_t)f * n_obj + o) * hw + p0 + p]);
        }
        __syncthreads();
        if (worker) {
            const float4 *e = reinterpret_cast<const float4 *>(emb + ((size_t)f * hw + p0) * C) + piece;
            for (int p = r0; p < np; p += 4 * rpp) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = e[(size_t)min(p + u * rpp, np - 1) * c4];       // clamped: the loads stay branch-free
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pp = p + u * rpp;
                    if (pp < np) {
                        acc[OM].x += v[u].x; acc[OM].y += v[u].y; acc[OM].z += v[u].z; acc[OM].w += v[u].w;
#pragma unroll
                        for (int o = 0; o < OM; ++o) {
                            if (o < n_obj) {
                                const float w = llab[o * MP_PIX + pp];
                                acc[o].x += v[u].x * w; acc[o].y += v[u].y * w; acc[o].z += v[u].z * w; acc[o].w += v[u].w * w;
                            }
                        }
                    }
                }
            }
        }
        if ((int)threadIdx.x < n_obj)
            for (int p = 0; p < np; ++p) cnt += llab[threadIdx.x * MP_PIX + p];
    }
    if (worker) {
#pragma unroll
        for (int o = 0; o < OM; ++o)
            if (o < n_obj) lcomb[(r0 * (n_obj + 1) + o) * c4 + piece] = acc[o];
        lcomb[(r0 * (n_obj + 1) + n_obj) * c4 + piece] = acc[OM];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (n_obj + 1) * c4; i += 256) {
        const int o = i / c4, pc = i - o * c4;
        float4 t = lcomb[o * c4 + pc];
        for (int r = 1; r < rpp; ++r) {
            const float4 x = lcomb[(r * (n_obj + 1) + o) * c4 + pc];
            t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
        }
        *reinterpret_cast<float4 *>(partial + ((size_t)blockIdx.x * (n_obj + 1) + o) * C + 4 * pc) = t;
    }
    if ((int)threadIdx.x < n_obj) pcount[(size_t)blockIdx.x * n_obj + threadIdx.x] = cnt;
}

// block = object; 1024 threads = 8 slices of the partial blocks x 128 channel lanes; fixed combine order
__global__ __launch_bounds__(1024) void masked_pool_final_kernel(const float *__restrict__ partial, const float *__restrict__ pcount,
                                                                  int n_blocks, int C, int n_obj, float total_pixels, float eps,
                                                                  float *__restrict__ out_pos, float *__restrict__ out_neg,
                                                                  float *__restrict__ out_pos_sqnorm) {
    __shared__ float lpos[8][128], ltot[8][128], lcnt[8], lsq[2];
    const int o = blockIdx.x;
    const int cl = threadIdx.x & 127, slice = threadIdx.x >> 7;
    const int per = (n_blocks + 7) / 8;
    const int b0 = slice * per, b1 = min(n_blocks, b0 + per);
    if (cl == 0) {
        float cnt = 0.0f;
        for (int b = b0; b < b1; ++b) cnt += pcount[(size_t)b * n_obj + o];
        lcnt[slice] = cnt;
    }
    float sq = 0.0f;
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int c = c0 + cl;
        float pos = 0.0f, tot = 0.0f;
        if (c < C) {
            int b = b0;
            for (; b + 16 <= b1; b += 16) {              // 32 loads in flight per thread; the additions keep their order
                float p[16], t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    p[u] = partial[((size_t)(b + u) * (n_obj + 1) + o) * C + c];
                    t[u] = partial[((size_t)(b + u) * (n_obj + 1) + n_obj) * C + c];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) { pos += p[u]; tot += t[u]; }
            }
            for (; b + 4 <= b1; b += 4) {
                float p[4], t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    p[u] = partial[((size_t)(b + u) * (n_obj + 1) + o) * C + c];
                    t[u] = partial[((size_t)(b + u) * (n_obj + 1) + n_obj) * C + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { pos += p[u]; tot += t[u]; }
            }
            for (; b < b1; ++b) {
                pos += partial[((size_t)b * (n_obj + 1) + o) * C + c];
                tot += partial[((size_t)b * (n_obj + 1) + n_obj) * C + c];
            }
        }
        __syncthreads();
        lpos[slice][cl] = pos;
        ltot[slice][cl] = tot;
        __syncthreads();
        if (slice == 0 && c < C) {
            float cnt = 0.0f, ps = 0.0f, ts = 0.0f;
            for (int q = 0; q < 8; ++q) { cnt += lcnt[q]; ps += lpos[q][cl]; ts += ltot[q][cl]; }
            const float pv = ps / (cnt + eps);                                         // ATT:173
            out_pos[(size_t)o * C + c] = pv;
            out_neg[(size_t)o * C + c] = (ts - ps) / ((total_pixels - cnt) + eps);     // ATT:166,174
            sq += pv * pv;
        }
    }
    if (slice == 0) {
        sq = aoc_wave_sum(sq);
        if (aoc_lane() == 0) lsq[threadIdx.x >> 6] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0 && out_pos_sqnorm) out_pos_sqnorm[o] = lsq[0] + lsq[1];
}

// ------------------------------------------------------------------------------------------ FiLM
// one wave per output channel; all objects at once.  gain[o,c] = 1 + tanh(head[o,:].W[c,:] + b[c])
__global__ __launch_bounds__(64) void film_gain_kernel(const float *__restrict__ head, const float *__restrict__ weight,
                                                        const float *__restrict__ bias, int n_obj, int D, int channels,
                                                        float *__restrict__ gain) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    const float *w = weight + (size_t)c * D;
    for (int o0 = 0; o0 < n_obj; o0 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = lane; d < D; d += 64) {
            const float wv = w[d];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (o0 + q < n_obj) acc[q] += wv * head[(size_t)(o0 + q) * D + d];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = aoc_wave_sum(acc[q]);
            if (lane == 0 && o0 + q < n_obj) gain[(size_t)(o0 + q) * channels + c] = 1.0f + tanhf(s + (bias ? bias[c] : 0.0f));
        }
    }
}

// y[p, :] = gain[p] * x[p, :]   (planes p = (object, channel)); grid.y = plane
__global__ __launch_bounds__(256) void channel_scale_kernel(const float *__restrict__ x, const float *__restrict__ gain, int64_t hw,
                                                             float *__restrict__ y) {
    const int64_t plane = blockIdx.y;
    const float g = gain[plane];
    const float *xp = x + plane * hw;
    float *yp = y + plane * hw;
    // 16-byte aligned body + scalar head/tail (hw is odd for the 16k+1 input sizes)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(xp);
    int64_t head = ((16 - (addr & 15)) & 15) / 4;
    if (head > hw) head = hw;
    const bool same_align = ((reinterpret_cast<uintptr_t>(yp) & 15) == (addr & 15));
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (same_align) {
        if (tid < head) yp[tid] = g * xp[tid];
        const int64_t body4 = (hw - head) / 4;
        const float4 *x4 = reinterpret_cast<const float4 *>(xp + head);
        float4 *y4 = reinterpret_cast<float4 *>(yp + head);
        for (int64_t i = tid; i < body4; i += nthreads) {
            float4 v = x4[i];
            v.x *= g; v.y *= g; v.z *= g; v.w *= g;
            y4[i] = v;
        }
        const int64_t tail0 = head + body4 * 4;
        if (tid < hw - tail0) yp[tail0 + tid] = g * xp[tail0 + tid];
    } else {
        for (int64_t i = tid; i < hw; i += nthreads) yp[i] = g * xp[i];
    }
}

#ifdef AOC_DEV
// FiLM gate in one launch: every block first computes its plane's gain 1 + tanh(head[o,:].W[c,:] + b[c]) (a D-long
// dot product, block-reduced), then streams its slice of the plane.  Saves the separate gain launch + round trip.
__global__ __launch_bounds__(256) void film_scale_kernel(const float *__restrict__ x, const float *__restrict__ head, const float *__restrict__ weight,
                                                          const float *__restrict__ bias, int D, int channels, int64_t hw, float *__restrict__ y, int nt) {
    __shared__ float wsum[4];
    const int64_t plane = blockIdx.y;
    const int o = (int)(plane / channels), c = (int)(plane - (int64_t)o * channels);
    const float *h = head + (size_t)o * D, *w = weight + (size_t)c * D;
    float acc = 0.0f;
    // four strides of the dot product at a time, branch-free (clamped index, select afterwards): the eight loads are in flight together
    // instead of one dependent round trip per stride, which is most of a workgroup's life when its slice of the plane is small
    for (int d0 = threadIdx.x; d0 < D; d0 += 4 * blockDim.x) {
        float wv[4], hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = min(d0 + u * (int)blockDim.x, D - 1);
            wv[u] = w[d];
            hv[u] = h[d];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (d0 + u * (int)blockDim.x < D) ? wv[u] * hv[u] : 0.0f;
    }
    acc = aoc_wave_sum(acc);
    if (aoc_lane() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float g = 1.0f + tanhf((wsum[0] + wsum[1] + wsum[2] + wsum[3]) + (bias ? bias[c] : 0.0f));
    const float *xp = x + plane * hw;
    float *yp = y + plane * hw;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(xp);
    int64_t headn = ((16 - (addr & 15)) & 15) / 4;
    if (headn > hw) headn = hw;
    const bool same_align = ((reinterpret_cast<uintptr_t>(yp) & 15) == (addr & 15));
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    if (same_align) {
        if (tid < headn) yp[tid] = g * xp[tid];
        const int64_t body4 = (hw - headn) / 4;
        const float4 *x4 = reinterpret_cast<const float4 *>(xp + headn);
        float4 *y4 = reinterpret_cast<float4 *>(yp + headn);
        for (int64_t i0 = tid; i0 < body4; i0 += 4 * nthreads) {     // four strides per trip, all loads in flight before the first store
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u * nthreads;
                v[u] = x4[i < body4 ? i : body4 - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u * nthreads;
                if (i < body4) {
                    v[u].x *= g; v[u].y *= g; v[u].z *= g; v[u].w *= g;
                    if (nt) {
                        typedef float f32x4_t __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(f32x4_t{v[u].x, v[u].y, v[u].z, v[u].w}, reinterpret_cast<f32x4_t *>(y4 + i));
                    } else {
                        y4[i] = v[u];
                    }
                }
            }
        }
        const int64_t tail0 = headn + body4 * 4;
        if (tid < hw - tail0) yp[tail0 + tid] = g * xp[tail0 + tid];
    } else {
        for (int64_t i = tid; i < hw; i += nthreads) yp[i] = g * xp[i];
    }
}
#endif  // AOC_DEV

// FiLM gate in one launch: y[o,c,:] = (1 + tanh(head[o,:].W[c,:] + b[c])) x[o,c,:]  (ATT:12-17, CLB:81-84), the gain computed by every workgroup
// of the plane (a D-long dot product, block-reduced) -- no separate gain launch.  The workgroup's whole slice of the plane is requested BEFORE
// the dot product: a workgroup lives for one memory round
// trip (its U float4 per thread, the weight row and the head row are all in flight together) instead of two dependent ones, and U x 256 float4
// per workgroup keeps the half-resolution planes (1 631 float4) at one or two workgroups per plane instead of seven that move one float4 per
// thread behind a 400-long dot product each.
template <int U>
__global__ __launch_bounds__(256) void film_scale_ahead_kernel(const float *__restrict__ x, const float *__restrict__ head, const float *__restrict__ weight,
                                                                const float *__restrict__ bias, int D, int channels, int64_t hw, int chunk, float *__restrict__ y) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    __shared__ float wsum[4];
    const int64_t plane = blockIdx.y;
    const int o = (int)(plane / channels), c = (int)(plane - (int64_t)o * channels);
    const float *xp = x + plane * hw;
    float *yp = y + plane * hw;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(xp);
    int64_t headn = ((16 - (addr & 15)) & 15) / 4;
    if (headn > hw) headn = hw;
    const int64_t body4 = (hw - headn) / 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(yp) & 15) == (addr & 15)) && body4 > 0;
    const f32x4_t *x4 = reinterpret_cast<const f32x4_t *>(xp + headn);
    // the plane's float4 split evenly over the workgroups of the row (the last one is not a mostly empty straggler)
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = min(body4, i0 + chunk);          // chunk = ceil((hw / 4) / gridDim.x) <= 256 U
    f32x4_t v[U];
    if (vec) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + threadIdx.x + u * 256;
            if (i < i1) v[u] = __builtin_nontemporal_load(x4 + i);
        }
    }
    const float *h = head + (size_t)o * D, *w = weight + (size_t)c * D;
    float acc = 0.0f;
    for (int d0 = threadIdx.x; d0 < D; d0 += 4 * 256) {
        float wv[4], hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = min(d0 + u * 256, D - 1);
            wv[u] = w[d];
            hv[u] = h[d];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (d0 + u * 256 < D) ? wv[u] * hv[u] : 0.0f;
    }
    acc = aoc_wave_sum(acc);
    if (aoc_lane() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    const float g = 1.0f + tanhf((wsum[0] + wsum[1] + wsum[2] + wsum[3]) + (bias ? bias[c] : 0.0f));
    if (vec) {
        f32x4_t *y4 = reinterpret_cast<f32x4_t *>(yp + headn);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + threadIdx.x + u * 256;
            if (i < i1) __builtin_nontemporal_store(v[u] * g, y4 + i);
        }
        if (blockIdx.x == 0) {
            const int64_t tail0 = headn + body4 * 4;
            if (threadIdx.x < headn) yp[threadIdx.x] = g * xp[threadIdx.x];
            if (threadIdx.x < hw - tail0) yp[tail0 + threadIdx.x] = g * xp[tail0 + threadIdx.x];
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) yp[i] = g * xp[i];
    }
}

// ------------------------------------------------------------------------------------------ conditioning layer
__device__ __forceinline__ uint32_t float_order_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);    // larger float <=> larger key
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- fused conditioning-layer streams (round 2) --------------------------------------------------------------------------
// One pass over z produces the phi scores (CL:27), the per-plane sums that conditioning_block needs for its inter-object code
// (CLB:68 avg_pool2d) and the first radix histogram of the k-th-largest selection (CL:33): a conditioning block reads its activation
// three times (this pass, the masked pooling, the FiLM scale) instead of four.  The selection itself is spread over the whole GPU:
// one small kernel per remaining radix digit instead of one 1024-thread block per sample walking its scores four times.
constexpr int CS_PIX = 256;                     // pixels per block of the fused scores pass and of the radix passes
struct CondSel { uint32_t prefix, k; };

__device__ __forceinline__ float wave_sum_dpp(float v) {      // lane 63 gets the sum of the wave (row scans + two row broadcasts)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));   // row_shr:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;
}

// histogram one digit of up to 4 keys per thread into the block's LDS bins: the lanes of a wave that hit the same bin are counted
// with a ballot first (scores of one map share their leading bits, so plain LDS atomics would serialise 64 ways)
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, bool act) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const unsigned long long mm = __ballot(act);
        if (mm == 0ull) break;
        const int leader = __builtin_ctzll(mm);
        const uint32_t lb = __shfl(bin, leader);
        const unsigned long long same = __ballot(act && bin == lb);
        if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
        act = act && bin != lb;
    }
    if (act) atomicAdd(&hist[bin], 1u);
}

// the bin (from the top) in which the `need`-th largest key of a 256-bin histogram falls, and how many of the keys of that bin
// are still to be skipped; one wave, every lane returns the result
__device__ __forceinline__ void select_bin(const uint32_t *__restrict__ hist, uint32_t need, uint32_t &bin, uint32_t &left) {
    const int lane = threadIdx.x & 63;
    // lane l owns bins 255 - 4l .. 252 - 4l (descending)
    uint32_t h[4], tot = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { h[u] = hist[255 - (4 * lane + u)]; tot += h[u]; }
    uint32_t inc = tot;                                       // inclusive scan over lanes = keys in this lane's bins and all higher ones
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    const uint32_t before = inc - tot;
    const bool mine = before < need && need <= inc;
    uint32_t b = 0, l = 0;
    if (mine) {
        uint32_t acc = before;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (acc < need && need <= acc + h[u]) { b = 255 - (4 * lane + u); l = need - acc; }
            acc += h[u];
        }
    }
    const unsigned long long mm = __ballot(mine);
    const int src = mm ? __builtin_ctzll(mm) : 0;
    bin = __shfl(b, src);
    left = __shfl(l, src);
}

// First pass over z, work item = (tile of CS_TILE pixels, chunk of CS_CH channels, sample): partial scores of the chunk
// (sum_c w[c] z[n,c,p], channels in order) and the tile's share of the chunk's plane sums.  A thread owns CS_PPT pixels, so the
// cross-lane reduction of a plane sum (6 DPP adds) is paid once per CS_PPT values; every wave-wide load is a 256-byte run.
// part_scores [n_chunks][N][hw]; plane_partial [n_tiles][N][C]
constexpr int CS_PPT = 8, CS_TILE = 256 * CS_PPT, CS_CH = 32;
__global__ __launch_bounds__(256) void cond_scores_part_kernel(const float *__restrict__ z, int C, int64_t hw, const float *__restrict__ phi_w,
                                                                float *__restrict__ part_scores, float *__restrict__ plane_partial,
                                                                uint32_t *__restrict__ hist) {
    __shared__ float lps[4][CS_CH];
    const int n = blockIdx.z, chunk = blockIdx.y, N = gridDim.z;
    // the radix histograms of this call start at zero: the launches that add to them come after this one on the stream, the last reader
    // of the previous call's came before it (no memset node per call)
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < 4 * 256; i += 256) hist[(size_t)n * 4 * 256 + i] = 0u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t p0 = (int64_t)blockIdx.x * CS_TILE + threadIdx.x;
    const int c0 = chunk * CS_CH, c1 = min(C, c0 + CS_CH);
    const float *zn = z + ((size_t)n * C + c0) * hw;
    float s[CS_PPT];
#pragma unroll
    for (int u = 0; u < CS_PPT; ++u) s[u] = 0.0f;
    // four channels' loads (32 per thread) in flight together: with one channel at a time the 32 channels of a chunk were 32 dependent
    // memory round trips on a launch of ~1.6 workgroups per CU (31 us for 105 MB); the additions keep the channel order
    constexpr int CS_CU = 4;
    for (int c = c0; c < c1; c += CS_CU) {
        float v[CS_CU][CS_PPT];
#pragma unroll
        for (int k = 0; k < CS_CU; ++k) {
            const int cc = min(c + k, c1 - 1);
#pragma unroll
            for (int u = 0; u < CS_PPT; ++u) {
                const int64_t p = p0 + 256 * u;
                v[k][u] = zn[(size_t)(cc - c0) * hw + (p < hw ? p : hw - 1)];
            }
        }
#pragma unroll
        for (int k = 0; k < CS_CU; ++k) {
            if (c + k < c1) {
                const float w = phi_w[c + k];
                float t = 0.0f;
#pragma unroll
                for (int u = 0; u < CS_PPT; ++u) {
                    const float x = (p0 + 256 * u < hw) ? v[k][u] : 0.0f;
                    s[u] += w * x;
                    t += x;
                }
                t = wave_sum_dpp(t);
                if (lane == 63) lps[wave][c + k - c0] = t;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CS_PPT; ++u) {
        const int64_t p = p0 + 256 * u;
        if (p < hw) part_scores[((size_t)chunk * N + n) * hw + p] = s[u];
    }
    __syncthreads();
    if ((int)threadIdx.x < c1 - c0)
        plane_partial[((size_t)blockIdx.x * N + n) * C + c0 + threadIdx.x] =
            (lps[0][threadIdx.x] + lps[1][threadIdx.x]) + (lps[2][threadIdx.x] + lps[3][threadIdx.x]);
}

// scores[n,p] = sum over the channel chunks (in order) of the partial scores + b   (CL:27), and the radix histogram of their top byte.
// grid (ceil(hw / CS_PIX), N); hist [N][4][256] (zeroed by the caller)
__global__ __launch_bounds__(CS_PIX) void cond_scores_reduce_kernel(const float *__restrict__ part_scores, int n_chunks, int64_t hw,
                                                                     const float *__restrict__ phi_b, float *__restrict__ scores,
                                                                     uint32_t *__restrict__ hist) {
    __shared__ uint32_t lh[256];
    const int n = blockIdx.y, N = gridDim.y;
    const int64_t p = (int64_t)blockIdx.x * CS_PIX + threadIdx.x;
    lh[threadIdx.x] = 0;
    const bool ok = p < hw;
    float s = 0.0f;
    if (ok)
        for (int k = 0; k < n_chunks; ++k) s += part_scores[((size_t)k * N + n) * hw + p];
    const float sc = s + phi_b[0];
    if (ok) scores[(size_t)n * hw + p] = sc;
    __syncthreads();
    hist_add(lh, float_order_key(sc) >> 24, ok);
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[((size_t)n * 4 + 0) * 256 + threadIdx.x], lh[threadIdx.x]);
}

// radix digit `pass` (1..3) of the k-th-largest selection: every block first resolves digit pass-1 from its complete histogram
// (block x = 0 records it in sel[n][pass-1]), then counts digit `pass` of the keys that still match.   grid (ceil(hw / 1024), N)
__global__ __launch_bounds__(256) void cond_select_pass_kernel(const float *__restrict__ scores, int64_t hw, int k_rank, int pass,
                                                                uint32_t *__restrict__ hist, CondSel *__restrict__ sel) {
    __shared__ uint32_t lh[256];
    __shared__ CondSel cur;
    const int n = blockIdx.y;
    lh[threadIdx.x] = 0;
    if (threadIdx.x < 64) {
        const CondSel prev = pass == 1 ? CondSel{0u, (uint32_t)k_rank} : sel[(size_t)n * 4 + pass - 2];
        uint32_t bin, left;
        select_bin(hist + ((size_t)n * 4 + pass - 1) * 256, prev.k, bin, left);
        if (threadIdx.x == 0) {
            cur = CondSel{prev.prefix | (bin << (8 * (4 - pass))), left};
            if (blockIdx.x == 0) sel[(size_t)n * 4 + pass - 1] = cur;
        }
    }
    __syncthreads();
    const uint32_t prefix = cur.prefix, mask = 0xffffffffu << (8 * (4 - pass)), shift = 8 * (3 - pass);
    {
        const int64_t p = (int64_t)blockIdx.x * CS_PIX + threadIdx.x;
        uint32_t key = 0;
        bool act = false;
        if (p < hw) {
            key = float_order_key(scores[(size_t)n * hw + p]);
            act = (key & mask) == prefix;
        }
        hist_add(lh, (key >> shift) & 255u, act);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[((size_t)n * 4 + pass) * 256 + threadIdx.x], lh[threadIdx.x]);
}

// Radix digits 1..3 of the selection in ONE launch, one 1024-thread workgroup per sample: the keys of the sample stay in registers (KPT per
// thread), digit 0 comes from the complete histogram the score reduction left in global memory, every later digit from the LDS histogram of
// the pass before.  Below the top byte the keys of a map are spread over the bins, so plain LDS atomics do (the ballot pre-count of hist_add
// is for the top byte, which the keys of one map share).  Leaves what cond_masked_gap_fused_kernel reads: hist[n][3] (plain stores) and
// sel[n][2].  Replaces three dependent launches of ~4.6 us each.
template <int KPT>
__global__ __launch_bounds__(1024) void cond_select_tail_kernel(const float *__restrict__ scores, int64_t hw, int k_rank, uint32_t *__restrict__ hist,
                                                                 CondSel *__restrict__ sel) {
    __shared__ uint32_t lh[2][256];
    __shared__ CondSel cur;
    const int n = blockIdx.x;
    const float *s = scores + (size_t)n * hw;
    uint32_t key[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int64_t i = (int64_t)threadIdx.x + u * 1024;
        key[u] = float_order_key(s[i < hw ? i : hw - 1]);
    }
    const int nv = (int)((hw - threadIdx.x + 1023) / 1024);           // valid keys of this thread
    if (threadIdx.x < 512) lh[threadIdx.x >> 8][threadIdx.x & 255] = 0u;
    CondSel prev = CondSel{0u, (uint32_t)k_rank};
#pragma unroll
    for (int pass = 1; pass <= 3; ++pass) {
        if (threadIdx.x < 64) {
            uint32_t bin, left;
            select_bin(pass == 1 ? hist + (size_t)n * 4 * 256 : lh[pass & 1], prev.k, bin, left);
            if (threadIdx.x == 0) cur = CondSel{prev.prefix | (bin << (8 * (4 - pass))), left};
        }
        __syncthreads();
        prev = cur;
        uint32_t *H = lh[(pass + 1) & 1];
        const uint32_t mask = 0xffffffffu << (8 * (4 - pass)), shift = 8 * (3 - pass);
#pragma unroll
        for (int u = 0; u < KPT; ++u)
            if (u < nv && (key[u] & mask) == prev.prefix) atomicAdd(&H[(key[u] >> shift) & 255u], 1u);
        __syncthreads();
        if (pass < 3 && threadIdx.x < 256) lh[pass & 1][threadIdx.x] = 0u;     // consumed by this pass's select_bin; the pass after next adds to it
    }
    if (threadIdx.x < 256) hist[((size_t)n * 4 + 3) * 256 + threadIdx.x] = lh[0][threadIdx.x];
    if (threadIdx.x == 0) sel[(size_t)n * 4 + 2] = prev;
}

// gap[n,c] = (1/HW) * sum_p z[n,c,p] * (scores[n,p] > threshold[n])    (CL:36-43), threshold = the k-th largest score (last radix digit
// resolved here); also plane_mean[n,c] from the block partials of the fused pass.  One block per plane.
__global__ __launch_bounds__(256) void cond_masked_gap_fused_kernel(const float *__restrict__ z, int C, int64_t hw, const float *__restrict__ scores,
                                                                     const uint32_t *__restrict__ hist, const CondSel *__restrict__ sel,
                                                                     const float *__restrict__ plane_partial, int n_part,
                                                                     float *__restrict__ threshold, float *__restrict__ gap, float *__restrict__ plane_mean) {
    __shared__ float wsum[4];
    __shared__ float lthr;
    const int c = blockIdx.x, n = blockIdx.y, N = gridDim.y;
    if (threadIdx.x < 64) {
        const CondSel prev = sel[(size_t)n * 4 + 2];
        uint32_t bin, left;
        select_bin(hist + ((size_t)n * 4 + 3) * 256, prev.k, bin, left);
        if (threadIdx.x == 0) {
            lthr = key_to_float(prev.prefix | bin);
            if (c == 0 && threshold) threshold[n] = lthr;
        }
    }
    if (threadIdx.x == 64 && plane_mean) {                    // fixed-order sum of the block partials: run-to-run reproducible
        float t = 0.0f;
        for (int b = 0; b < n_part; ++b) t += plane_partial[((size_t)b * N + n) * C + c];
        plane_mean[(size_t)n * C + c] = t / (float)hw;
    }
    __syncthreads();
    const float thr = lthr;
    const float *zp = z + ((size_t)n * C + c) * hw;
    const float *sp = scores + (size_t)n * hw;
    float acc = 0.0f;
    int64_t p = threadIdx.x;
    for (; p + 3 * (int64_t)blockDim.x < hw; p += 4 * (int64_t)blockDim.x) {
        float zv[4], sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { zv[u] = zp[p + u * blockDim.x]; sv[u] = sp[p + u * blockDim.x]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (sv[u] > thr) ? zv[u] : 0.0f;
    }
    for (; p < hw; p += blockDim.x) acc += (sp[p] > thr) ? zp[p] : 0.0f;
    acc = aoc_wave_sum(acc);
    if (aoc_lane() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gap[(size_t)n * C + c] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)hw;
}

// ---- GroupNorm (+ residual) + ReLU in two streams (the decoder Bottleneck, gct.py:69-90) ---------------------------
// GroupNorm as torch computes it (biased variance over the group's channels x HW), then y = relu(gn(x) [+ residual]).  PyTorch runs the
// normalisation, the residual add and the ReLU as separate passes; here x is read twice (statistics, apply) and y written once.
// Statistics: per (sample, group) block, per-thread fp32 partial sums of x and x^2 folded in fp64 (fixed order: reproducible).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, int group_channels, int64_t hw, float eps,
                                                        float *__restrict__ stats /* [N * groups][2] mean, rstd */) {
    __shared__ double sh[2][4];
    const float *xp = x + (size_t)blockIdx.x * group_channels * hw;
    const int64_t n = (int64_t)group_channels * hw;
    double s = 0.0, q = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 7 * 256 < n; i += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[i + u * 256];
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { ps += v[u]; pq += v[u] * v[u]; }
        s += ps; q += pq;
    }
    for (; i < n; i += 256) { const float v = xp[i]; s += v; q += (double)v * v; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (aoc_lane() == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ts = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]), tq = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        const double mean = ts / (double)n;
        double var = tq / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * blockIdx.x] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// grid (ceil(hw / 1024), N * C): y = [relu]( (x - mean) * rstd * gamma[c] + beta[c] [+ residual] )
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, int C, int group_channels, int64_t hw, const float *__restrict__ stats,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        const float *__restrict__ residual, int relu, float *__restrict__ y) {
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    const int g = n * (C / group_channels) + c / group_channels;
    const float mean = stats[2 * g], rstd = stats[2 * g + 1];
    const float a = rstd * (gamma ? gamma[c] : 1.0f), b = (beta ? beta[c] : 0.0f) - mean * a;
    const float *xp = x + (size_t)plane * hw;
    const float *rp = residual ? residual + (size_t)plane * hw : nullptr;
    float *yp = y + (size_t)plane * hw;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t p = (int64_t)blockIdx.x * 1024 + u * 256 + threadIdx.x;
        if (p < hw) {
            float v = xp[p] * a + b;
            if (rp) v += rp[p];
            yp[p] = relu ? fmaxf(v, 0.0f) : v;
        }
    }
}

// out[p, :] = sum_o lab[p, o] * rows[o, :]   (aocnet.py:325: matmul(prev label, prev_head_pos))
__global__ __launch_bounds__(256) void label_mix_kernel(const float *__restrict__ lab, const float *__restrict__ rows, int64_t n, int n_obj, int C,
                                                         float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int64_t p = idx / C;
    const int c = (int)(idx - p * C);
    float acc = 0.0f;
    for (int o = 0; o < n_obj; ++o) acc += lab[p * n_obj + o] * rows[(size_t)o * C + c];
    out[idx] = acc;
}

// y[n,o] = x[n,:] . W[o,:] + b[o]; one wave per output
__global__ __launch_bounds__(64) void linear_kernel(const float *__restrict__ x, const float *__restrict__ weight, const float *__restrict__ bias,
                                                     int in_dim, int out_dim, float *__restrict__ y) {
    const int o = blockIdx.x, n = blockIdx.y;
    const float *xr = x + (size_t)n * in_dim, *wr = weight + (size_t)o * in_dim;
    float acc = 0.0f;
    for (int d = threadIdx.x; d < in_dim; d += 64) acc += xr[d] * wr[d];
    acc = aoc_wave_sum(acc);
    if (threadIdx.x == 0) y[(size_t)n * out_dim + o] = acc + (bias ? bias[o] : 0.0f);
}

// float4 loads, eight per thread in flight (32 KB per workgroup: one workgroup per plane, 4-8 workgroups per CU, needs that much outstanding to
// reach the HBM rate); the 16-byte aligned body + scalar head / tail of the streaming kernels (hw is odd for the 16k + 1 input sizes)
__global__ __launch_bounds__(256) void plane_mean4_kernel(const float *__restrict__ x, int64_t hw, float *__restrict__ out) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    __shared__ float wsum[4];
    const float *xp = x + (size_t)blockIdx.x * hw;
    int64_t headn = ((16 - (reinterpret_cast<uintptr_t>(xp) & 15)) & 15) / 4;
    if (headn > hw) headn = hw;
    const int64_t body4 = (hw - headn) / 4, tail0 = headn + body4 * 4;
    const f32x4_t *x4 = reinterpret_cast<const f32x4_t *>(xp + headn);
    float acc = 0.0f;
    if (threadIdx.x < headn) acc += xp[threadIdx.x];
    if (threadIdx.x < hw - tail0) acc += xp[tail0 + threadIdx.x];
    for (int64_t i0 = threadIdx.x; i0 < body4; i0 += 8 * 256) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + u * 256;
            v[u] = __builtin_nontemporal_load(x4 + (i < body4 ? i : body4 - 1));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (i0 + u * 256 < body4) ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.0f;
    }
    acc = aoc_wave_sum(acc);
    if (aoc_lane() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)hw;
}

// out[o, :D] = head[o, :];  out[o, D + c] = sum_o' px[o', c] - px[o, c]   (decoding_module.py:126-130: the IA head extended with the
// inter-object code of the gate's own input; replaces a reduction, a subtraction and a concatenation of tiny tensors)
__global__ __launch_bounds__(256) void head_delta_kernel(const float *__restrict__ head, int D, const float *__restrict__ px, int n_obj, int C,
                                                          float *__restrict__ out) {
    const int o = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D + C) return;
    float v;
    if (i < D) {
        v = head[(size_t)o * D + i];
    } else {
        const int c = i - D;
        float s = 0.0f;
        for (int q = 0; q < n_obj; ++q) s += px[(size_t)q * C + c];          // the same left-to-right sum as torch.sum(dim=0) over a handful of objects
        v = s - px[(size_t)o * C + c];
    }
    out[(size_t)o * (D + C) + i] = v;
}

// ------------------------------------------------------------------------------------------
// Decoder-side streams that sit next to the FiLM gates.
// plane_reduce: out[plane] = sum over the plane of x (mode 0), x^2 (mode 1) or |x| (mode 2)  (gct.py:19,27-30)
__global__ __launch_bounds__(256) void plane_reduce_kernel(const float *__restrict__ x, int64_t hw, int mode, float *__restrict__ out) {
    __shared__ float wsum[4];
    const float *xp = x + (size_t)blockIdx.x * hw;
    float acc = 0.0f;
    int64_t p = threadIdx.x;
    for (; p + 7 * (int64_t)blockDim.x < hw; p += 8 * (int64_t)blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[p + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (mode == 1) ? v[u] * v[u] : (mode == 2 ? fabsf(v[u]) : v[u]);
    }
    for (; p < hw; p += blockDim.x) {
        const float v = xp[p];
        acc += (mode == 1) ? v * v : (mode == 2 ? fabsf(v) : v);
    }
    acc = aoc_wave_sum(acc);
    if (aoc_lane() == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// GCT gate (gct.py:17-36): per sample n, from the per-plane reductions s[n, c]:
//   l2: e_c = sqrt(s_c + eps) * alpha_c,  norm_c = gamma_c / sqrt(mean_c(e_c^2) + eps)
//   l1: e_c = s_c * alpha_c,              norm_c = gamma_c / (mean_c|e_c| + eps)
//   gate[n, c] = 1 + tanh(e_c * norm_c + beta_c)
__global__ __launch_bounds__(256) void gct_gate_kernel(const float *__restrict__ s, const float *__restrict__ alpha, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, int C, float eps, int l1, float *__restrict__ gate) {
    __shared__ float wsum[4];
    __shared__ float mean_s;
    const int n = blockIdx.x;
    float part = 0.0f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sv = s[(size_t)n * C + c];
        const float e = l1 ? sv * alpha[c] : sqrtf(sv + eps) * alpha[c];
        part += l1 ? fabsf(e) : e * e;
    }
    part = aoc_wave_sum(part);
    if (aoc_lane() == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) mean_s = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)C;
    __syncthreads();
    const float m = mean_s;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sv = s[(size_t)n * C + c];
        const float e = l1 ? sv * alpha[c] : sqrtf(sv + eps) * alpha[c];
        const float norm = l1 ? gamma[c] / (m + eps) : gamma[c] / sqrtf(m + eps);
        gate[(size_t)n * C + c] = 1.0f + tanhf(e * norm + beta[c]);
    }
}

// IA_logit (decoding_module.py:151-160): logit[n, p] = sum_c x[n, c, p] * weight[n, c] + bias[n]: a 1x1 grouped convolution
// whose weights are generated per object from the IA head
__global__ __launch_bounds__(256) void object_logit_kernel(const float *__restrict__ x, int C, int64_t hw, const float *__restrict__ weight,
                                                           int64_t weight_stride, const float *__restrict__ bias, int64_t bias_stride,
                                                           float *__restrict__ out) {
    const int n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const float *xp = x + (size_t)n * C * hw + p;
    const float *w = weight + (size_t)n * weight_stride;
    float s = 0.0f;
    int c = 0;
    for (; c + 8 <= C; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[(size_t)(c + u) * hw];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += w[c + u] * v[u];
    }
    for (; c < C; ++c) s += w[c] * xp[(size_t)c * hw];
    out[(size_t)n * hw + p] = s + bias[(size_t)n * bias_stride];
}

// conditioning_block codes in one launch (CLB:68-80): code[n] = [ W1 gap[n] + b1 | W2 (sum_m px[m] - px[n]) + b2 | W3 head[n] + b3 ]
// (the three mlp_layer products, the inter-object delta of CLB:69 and the concatenation of CLB:80); one wave per output
__global__ __launch_bounds__(64) void cond_codes_kernel(const float *__restrict__ gap, const float *__restrict__ px, const float *__restrict__ head,
                                                         const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w2,
                                                         const float *__restrict__ b2, const float *__restrict__ w3, const float *__restrict__ b3,
                                                         int N, int C, int D, float *__restrict__ code) {
    const int o = blockIdx.x, n = blockIdx.y;
    const int out_dim = 2 * C + D;
    float acc = 0.0f, bias;
    if (o < C) {
        const float *xr = gap + (size_t)n * C, *wr = w1 + (size_t)o * C;
        for (int d = threadIdx.x; d < C; d += 64) acc += xr[d] * wr[d];
        bias = b1[o];
    } else if (o < 2 * C) {
        const float *wr = w2 + (size_t)(o - C) * C;
        for (int d = threadIdx.x; d < C; d += 64) {
            float tot = 0.0f;
            for (int m = 0; m < N; ++m) tot += px[(size_t)m * C + d];          // CLB:69 px1.sum(dim=0) - px1
            acc += (tot - px[(size_t)n * C + d]) * wr[d];
        }
        bias = b2[o - C];
    } else {
        const float *xr = head + (size_t)n * D, *wr = w3 + (size_t)(o - 2 * C) * D;
        for (int d = threadIdx.x; d < D; d += 64) acc += xr[d] * wr[d];
        bias = b3[o - 2 * C];
    }
    acc = aoc_wave_sum(acc);
    if (threadIdx.x == 0) code[(size_t)n * out_dim + o] = acc + bias;
}

// ------------------------------------------------------------------------------------------
// DynamicPreHead (decoding_module.py:228-240: 1x1 conv -> GroupNorm -> ReLU on the [O, 24, h, w] proto-mask tensor) fused with
// the concatenation of aocnet.py:362 (current-frame embedding || pre-head output -> [O, C + E, h, w]).
constexpr int PH_MAX_IN = 32, PH_MAX_OUT = 128, PH_PIX = 256;
// pass 1: per (object, pixel chunk) partial sums of y and y^2 per GroupNorm group (y = W x + b), in double.
// N_IN > 0: the input channel count as a compile-time constant (24 / 26 / 28 = 22 + 2 x levels proto-mask channels): the 1x1 convolution is
// a fully unrolled FMA chain in the same order as the generic form (N_IN = 0: runtime count, every FMA behind a scalar branch -- 2 048
// branches per pixel at 28 -> 64 channels, which is what made this kernel and the next take ~100 us each on a 145 x 261 map).  The groups'
// wave sums are all formed first and combined behind ONE barrier (they were 16 x 2 barriers); the order of every addition is unchanged.
constexpr int PH_MAX_GROUPS = 32;
template <int N_IN>
__global__ __launch_bounds__(PH_PIX) void prehead_stats_kernel(const float *__restrict__ feat, int n_in_rt, int64_t hw, const float *__restrict__ w,
                                                                const float *__restrict__ b, int n_out, int group_size, int n_chunks,
                                                                double *__restrict__ partial) {
    __shared__ float lw[PH_MAX_OUT * PH_MAX_IN + PH_MAX_OUT];
    __shared__ double wred[PH_MAX_GROUPS][PH_PIX / 64][2];
    const int n_in = N_IN > 0 ? N_IN : n_in_rt;
    constexpr int NK = N_IN > 0 ? N_IN : PH_MAX_IN;
    const int o = blockIdx.y;
    for (int i = threadIdx.x; i < n_out * n_in; i += blockDim.x) lw[i] = w[i];
    for (int i = threadIdx.x; i < n_out; i += blockDim.x) lw[n_out * n_in + i] = b[i];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * PH_PIX + threadIdx.x;
    float x[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) x[k] = (k < n_in && p < hw) ? feat[((size_t)o * n_in + k) * hw + p] : 0.0f;
    const int n_groups = n_out / group_size;
    for (int g0 = 0; g0 < n_groups; g0 += PH_MAX_GROUPS) {
        const int g1 = min(n_groups, g0 + PH_MAX_GROUPS);
        for (int g = g0; g < g1; ++g) {
            float s1 = 0.0f, s2 = 0.0f;
            for (int c = g * group_size; c < (g + 1) * group_size; ++c) {
                float y = lw[n_out * n_in + c];
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    if (N_IN > 0 || k < n_in) y = __builtin_fmaf(lw[c * n_in + k], x[k], y);
                if (p < hw) { s1 += y; s2 += y * y; }
            }
            double d1 = (double)s1, d2 = (double)s2;
            for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
            if (aoc_lane() == 0) { wred[g - g0][threadIdx.x >> 6][0] = d1; wred[g - g0][threadIdx.x >> 6][1] = d2; }
        }
        __syncthreads();
        if ((int)threadIdx.x < g1 - g0) {
            double t1 = 0.0, t2 = 0.0;
            for (int wv = 0; wv < PH_PIX / 64; ++wv) { t1 += wred[threadIdx.x][wv][0]; t2 += wred[threadIdx.x][wv][1]; }
            double *dst = partial + (((size_t)o * n_groups + g0 + threadIdx.x) * n_chunks + blockIdx.x) * 2;
            dst[0] = t1; dst[1] = t2;
        }
        __syncthreads();
    }
}
// pass 2: mean / rstd per (object, group) from the chunk partials (fixed order: deterministic)
__global__ __launch_bounds__(64) void prehead_finalize_kernel(const double *__restrict__ partial, int n_chunks, double count, float eps,
                                                               float *__restrict__ stats) {
    const int og = blockIdx.x;
    double t1 = 0.0, t2 = 0.0;
    for (int c = threadIdx.x; c < n_chunks; c += 64) { t1 += partial[((size_t)og * n_chunks + c) * 2]; t2 += partial[((size_t)og * n_chunks + c) * 2 + 1]; }
    for (int off = 32; off > 0; off >>= 1) { t1 += __shfl_xor(t1, off); t2 += __shfl_xor(t2, off); }
    if (threadIdx.x == 0) {
        const double mean = t1 / count;
        double var = t2 / count - mean * mean;          // biased variance, like torch.nn.GroupNorm
        if (var < 0.0) var = 0.0;
        stats[og * 2] = (float)mean;
        stats[og * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
// pass 3: out[o, C + c, p] = relu((y - mean) * rstd * gamma_c + beta_c); out[o, k, p] = emb[p, k] for k < C
template <int N_IN>
__global__ __launch_bounds__(PH_PIX) void prehead_apply_kernel(const float *__restrict__ feat, int n_in_rt, int64_t hw, const float *__restrict__ w,
                                                                const float *__restrict__ b, int n_out, int group_size,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                const float *__restrict__ stats, const float *__restrict__ emb, int C,
                                                                float *__restrict__ out) {
    __shared__ float lw[PH_MAX_OUT * PH_MAX_IN + PH_MAX_OUT];
    const int n_in = N_IN > 0 ? N_IN : n_in_rt;
    constexpr int NK = N_IN > 0 ? N_IN : PH_MAX_IN;
    const int o = blockIdx.y;
    for (int i = threadIdx.x; i < n_out * n_in; i += blockDim.x) lw[i] = w[i];
    for (int i = threadIdx.x; i < n_out; i += blockDim.x) lw[n_out * n_in + i] = b[i];
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * PH_PIX + threadIdx.x;
    if (p >= hw) return;
    float *dst = out + (size_t)o * (C + n_out) * hw + p;
    if (emb) {
        const float *e = emb + (size_t)p * C;
        for (int k = 0; k < C; ++k) dst[(size_t)k * hw] = e[k];          // aocnet.py:188,362: the embedding, expanded over the objects
    }
    float x[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) x[k] = (k < n_in) ? feat[((size_t)o * n_in + k) * hw + p] : 0.0f;
    const int n_groups = n_out / group_size;
    for (int c = 0; c < n_out; ++c) {
        float y = lw[n_out * n_in + c];
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (N_IN > 0 || k < n_in) y = __builtin_fmaf(lw[c * n_in + k], x[k], y);
        const int g = c / group_size;
        const float mean = stats[((size_t)o * n_groups + g) * 2], rstd = stats[((size_t)o * n_groups + g) * 2 + 1];
        const float v = (y - mean) * rstd * gamma[c] + beta[c];
        dst[(size_t)(C + c) * hw] = v > 0.0f ? v : 0.0f;
    }
}

inline int pool_chunks(int64_t hw) { return (int)((hw + MP_PIX - 1) / MP_PIX); }

}  // namespace

extern "C" {

int aoc_fg2bg_min(const float *dis, int n_obj, int n_ch, int64_t inner, int64_t dis_obj_stride, float *out, int64_t out_obj_stride,
                  aoc_stream_t stream) {
    if (!dis || !out || n_obj < 2 || n_ch < 1 || inner < 1 || dis_obj_stride < n_ch * inner || out_obj_stride < inner) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(fg2bg_kernel, dim3((unsigned)((inner + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), dis, n_obj, n_ch, inner,
                       dis_obj_stride, out, out_obj_stride);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_proto_finish(float *feat, int n_obj, int64_t hw, int64_t obj_stride, int ch_local, int n_local, int ch_local_bg, int ch_global, int ch_global_bg,
                     int ch_prev_mask, const float *prev_labels, const float *ref_pos, const float *ref_neg, const float *prev_pos, const float *prev_neg,
                     int C, float *head, aoc_stream_t stream) {
    if (!feat || n_obj < 1 || hw < 1 || obj_stride < hw || n_local < 0) return AOC_ERR_INVALID_ARG;
    if (ch_prev_mask >= 0 && !prev_labels) return AOC_ERR_INVALID_ARG;
    if (head && (!ref_pos || !ref_neg || !prev_pos || !prev_neg || C < 1)) return AOC_ERR_INVALID_ARG;
    int64_t span = hw > (int64_t)n_local * hw ? hw : (int64_t)n_local * hw;
    if (head && span < (int64_t)n_obj * 4 * C) span = (int64_t)n_obj * 4 * C;
    hipLaunchKernelGGL(proto_finish_kernel, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), feat, n_obj, hw, obj_stride, ch_local,
                       n_local, ch_local_bg, ch_global, ch_global_bg, ch_prev_mask, prev_labels, ref_pos, ref_neg, prev_pos, prev_neg, C, head);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_masked_mean_pool_workspace_bytes(int n_frames, int64_t hw, int n_obj, int C) {
    if (n_frames < 1 || hw < 1 || n_obj < 1 || C < 1) return 0;
    const size_t nb = (size_t)n_frames * pool_chunks(hw);
    return aoc_align_up(nb * (n_obj + 1) * C * sizeof(float), 256) + aoc_align_up(nb * n_obj * sizeof(float), 256);
}

int aoc_masked_mean_pool(const float *emb, const float *labels, int n_frames, int64_t hw, int C, int n_obj, int labels_pixel_major, float epsilon,
                         float *out_pos, float *out_neg, float *out_pos_sqnorm, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!emb || !labels || !out_pos || !out_neg || !workspace) return AOC_ERR_INVALID_ARG;
    if (n_frames < 1 || hw < 1 || C < 1 || n_obj < 1) return AOC_ERR_INVALID_ARG;
    if (n_obj > MP_OMAX) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_masked_mean_pool_workspace_bytes(n_frames, hw, n_obj, C)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    const int cpf = pool_chunks(hw);
    const int nb = n_frames * cpf;
    float *partial = static_cast<float *>(workspace);
    float *pcount = reinterpret_cast<float *>(static_cast<char *>(workspace) + aoc_align_up((size_t)nb * (n_obj + 1) * C * sizeof(float), 256));
    const size_t plds = ((size_t)n_obj * MP_PIX + (size_t)MP_SPLIT * (n_obj + 1) * 64) * sizeof(float);
    const bool vec4 = (C & 3) == 0 && C >= 16 && C <= 256 && n_obj <= 8 && (reinterpret_cast<uintptr_t>(emb) & 15) == 0;
    const size_t plds4 = ((size_t)n_obj * MP_PIX + (size_t)(vec4 ? 256 / (C / 4) : 0) * (n_obj + 1) * C) * sizeof(float);
    if (vec4) {
        const int cpb = (nb + 767) / 768;                     // chunks per workgroup: at most ~768 partial blocks whatever the pool size
        const int nb4 = n_frames * ((cpf + cpb - 1) / cpb);
        if (n_obj <= 4)
            hipLaunchKernelGGL(masked_pool_partial4_kernel<4>, dim3(nb4), dim3(256), plds4, st, emb, labels, hw, C, n_obj, cpf, cpb, labels_pixel_major, partial, pcount);
        else
            hipLaunchKernelGGL(masked_pool_partial4_kernel<8>, dim3(nb4), dim3(256), plds4, st, emb, labels, hw, C, n_obj, cpf, cpb, labels_pixel_major, partial, pcount);
        hipLaunchKernelGGL(masked_pool_final_kernel, dim3(n_obj), dim3(1024), 0, st, partial, pcount, nb4, C, n_obj, (float)((double)hw * n_frames), epsilon, out_pos, out_neg, out_pos_sqnorm);
        AOC_RETURN_IF_LAUNCH_FAILED();
        return AOC_OK;
    }
    if (n_obj <= 4)
        hipLaunchKernelGGL(masked_pool_partial_kernel<4>, dim3(nb), dim3(256), plds, st, emb, labels, hw, C, n_obj, cpf, labels_pixel_major, partial, pcount);
    else if (n_obj <= 8)
        hipLaunchKernelGGL(masked_pool_partial_kernel<8>, dim3(nb), dim3(256), plds, st, emb, labels, hw, C, n_obj, cpf, labels_pixel_major, partial, pcount);
    else
        hipLaunchKernelGGL(masked_pool_partial_kernel<MP_OMAX>, dim3(nb), dim3(256), plds, st, emb, labels, hw, C, n_obj, cpf, labels_pixel_major, partial, pcount);
    hipLaunchKernelGGL(masked_pool_final_kernel, dim3(n_obj), dim3(1024), 0, st, partial, pcount, nb, C, n_obj, (float)((double)hw * n_frames), epsilon, out_pos, out_neg, out_pos_sqnorm);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_film_gain(const float *head, const float *weight, const float *bias, int n_obj, int head_dim, int channels, float *gain, aoc_stream_t stream) {
    if (!head || !weight || !gain || n_obj < 1 || head_dim < 1 || channels < 1) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(film_gain_kernel, dim3(channels), dim3(64), 0, aoc_hip_stream(stream), head, weight, bias, n_obj, head_dim, channels, gain);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_channel_scale(const float *x, const float *gain, int64_t planes, int64_t hw, float *y, aoc_stream_t stream) {
    if (!x || !gain || !y || planes < 1 || hw < 1) return AOC_ERR_INVALID_ARG;
    if (planes > 65535) return AOC_ERR_UNSUPPORTED;
    int bx = (int)((hw / 4 + 255) / 256);
    if (bx < 1) bx = 1;
    if (bx > 8) bx = 8;
    hipLaunchKernelGGL(channel_scale_kernel, dim3(bx, (unsigned)planes), dim3(256), 0, aoc_hip_stream(stream), x, gain, hw, y);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_film_scale(const float *x, const float *head, const float *weight, const float *bias, int n_obj, int head_dim, int channels,
                   int64_t hw, float *y, aoc_stream_t stream) {
    if (!x || !head || !weight || !y || n_obj < 1 || head_dim < 1 || channels < 1 || hw < 1) return AOC_ERR_INVALID_ARG;
    const int64_t planes = (int64_t)n_obj * channels;
    if (planes > 65535) return AOC_ERR_UNSUPPORTED;
    // U = 8 float4 per thread where that makes one workgroup per plane (the half-resolution maps), otherwise 4: measured equal within 2 % on
    // the full-resolution maps (tools/bench_gates.py).  Loads and stores are nontemporal: the activation is read once and the gated copy is a
    // pure stream-out -- every dirty line it would leave in the XCDs' L2s is written back at the next kernel boundary of ANY stream, and the
    // k-means chain on the side stream has ~100 of them per frame (bench: +2 % frames/s when the stores went nontemporal in round 3).
    static const int ahead = AOC_DEV_ENV_INT("AOC_FILM_AHEAD", -1);
    const int U = ahead == 4 || ahead == 8 ? ahead : (hw / 4 <= 2048 ? 8 : 4);
#ifdef AOC_DEV
    if (ahead == 0) {                                           // the kernel of rounds 2-4 (dot product first, then the stream)
        static const int nt = AOC_DEV_ENV_INT("AOC_FILM_NT", 1);
        const int bx = (int)std::min<int64_t>(8, std::max<int64_t>(1, (hw / 4 + 255) / 256));
        hipLaunchKernelGGL(film_scale_kernel, dim3(bx, (unsigned)planes), dim3(256), 0, aoc_hip_stream(stream), x, head, weight, bias, head_dim, channels, hw, y, nt);
        AOC_RETURN_IF_LAUNCH_FAILED();
        return AOC_OK;
    }
#endif
    const int64_t per_wg = 256 * (int64_t)U;
    const unsigned gx = (unsigned)std::max<int64_t>(1, (hw / 4 + per_wg - 1) / per_wg);
    const int chunk = (int)std::max<int64_t>(1, (hw / 4 + gx - 1) / gx);
    if (U == 4)
        hipLaunchKernelGGL(film_scale_ahead_kernel<4>, dim3(gx, (unsigned)planes), dim3(256), 0, aoc_hip_stream(stream), x, head, weight, bias, head_dim, channels, hw, chunk, y);
    else
        hipLaunchKernelGGL(film_scale_ahead_kernel<8>, dim3(gx, (unsigned)planes), dim3(256), 0, aoc_hip_stream(stream), x, head, weight, bias, head_dim, channels, hw, chunk, y);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

struct CondWs {
    float *scores, *threshold, *partial, *part_scores;
    uint32_t *hist;
    CondSel *sel;
    int n_part, n_chunks;
    size_t total;
};
static inline CondWs cond_carve(void *base, int N, int C, int64_t hw) {
    CondWs w;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += aoc_align_up(bytes, 256); return r; };
    w.n_part = (int)((hw + CS_TILE - 1) / CS_TILE);
    w.n_chunks = (C + CS_CH - 1) / CS_CH;
    w.hist = reinterpret_cast<uint32_t *>(take((size_t)N * 4 * 256 * sizeof(uint32_t)));       // zeroed by the first launch of a call
    w.sel = reinterpret_cast<CondSel *>(take((size_t)N * 4 * sizeof(CondSel)));
    w.scores = reinterpret_cast<float *>(take((size_t)N * hw * sizeof(float)));
    w.threshold = reinterpret_cast<float *>(take((size_t)N * sizeof(float)));
    w.partial = reinterpret_cast<float *>(take((size_t)w.n_part * N * C * sizeof(float)));
    w.part_scores = reinterpret_cast<float *>(take((size_t)w.n_chunks * N * hw * sizeof(float)));
    w.total = off;
    return w;
}

size_t aoc_cond_gate_pool_workspace_bytes(int N, int C, int64_t hw) {
    if (N < 1 || C < 1 || hw < 1) return 0;
    return cond_carve(nullptr, N, C, hw).total;
}

int aoc_cond_gate_pool(const float *z, int N, int C, int64_t hw, const float *phi_w, const float *phi_b, int k_rank,
                       float *gap, float *scores, float *threshold, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_cond_gate_pool_ex(z, N, C, hw, phi_w, phi_b, k_rank, gap, nullptr, scores, threshold, workspace, workspace_bytes, stream);
}

int aoc_cond_gate_pool_ex(const float *z, int N, int C, int64_t hw, const float *phi_w, const float *phi_b, int k_rank,
                          float *gap, float *plane_mean, float *scores, float *threshold, void *workspace, size_t workspace_bytes,
                          aoc_stream_t stream) {
    if (!z || !phi_w || !phi_b || !gap || !workspace) return AOC_ERR_INVALID_ARG;
    if (N < 1 || C < 1 || hw < 1 || k_rank < 1 || k_rank > hw) return AOC_ERR_INVALID_ARG;
    if (N > 65535 || C > 65535 * CS_CH) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_cond_gate_pool_workspace_bytes(N, C, hw)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    const CondWs w = cond_carve(workspace, N, C, hw);
    float *sc = scores ? scores : w.scores;
    const dim3 pgrid((unsigned)((hw + CS_PIX - 1) / CS_PIX), N);
    hipLaunchKernelGGL(cond_scores_part_kernel, dim3((unsigned)w.n_part, (unsigned)w.n_chunks, N), dim3(256), 0, st, z, C, hw, phi_w, w.part_scores, w.partial,
                       w.hist);
    hipLaunchKernelGGL(cond_scores_reduce_kernel, pgrid, dim3(CS_PIX), 0, st, w.part_scores, w.n_chunks, hw, phi_b, sc, w.hist);
    static const int tail = AOC_DEV_ENV_INT("AOC_COND_TAIL", 1);
    if (tail && hw <= 8 * 1024)
        hipLaunchKernelGGL(cond_select_tail_kernel<8>, dim3(N), dim3(1024), 0, st, sc, hw, k_rank, w.hist, w.sel);
    else if (tail && hw <= 32 * 1024)
        hipLaunchKernelGGL(cond_select_tail_kernel<32>, dim3(N), dim3(1024), 0, st, sc, hw, k_rank, w.hist, w.sel);
    else if (tail && hw <= 64 * 1024)
        hipLaunchKernelGGL(cond_select_tail_kernel<64>, dim3(N), dim3(1024), 0, st, sc, hw, k_rank, w.hist, w.sel);
    else
        for (int pass = 1; pass <= 3; ++pass)                     // larger maps: one launch per digit, spread over the GPU
            hipLaunchKernelGGL(cond_select_pass_kernel, pgrid, dim3(256), 0, st, sc, hw, k_rank, pass, w.hist, w.sel);
    hipLaunchKernelGGL(cond_masked_gap_fused_kernel, dim3(C, N), dim3(256), 0, st, z, C, hw, sc, w.hist, w.sel, w.partial, w.n_part, threshold, gap,
                       plane_mean);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_groupnorm_relu_workspace_bytes(int N, int groups) { return N < 1 || groups < 1 ? 0 : aoc_align_up((size_t)N * groups * 2 * sizeof(float), 256); }

int aoc_groupnorm_relu(const float *x, int N, int C, int64_t hw, int groups, const float *gamma, const float *beta, float eps,
                       const float *residual, int relu, float *y, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!x || !y || !workspace || N < 1 || C < 1 || hw < 1 || groups < 1) return AOC_ERR_INVALID_ARG;
    if (C % groups) return AOC_ERR_INVALID_ARG;
    if ((int64_t)N * C > 65535ll) return AOC_ERR_UNSUPPORTED;                  // gn_apply_kernel: one grid row per plane (grid.y limit); nothing is enqueued
    if (workspace_bytes < aoc_groupnorm_relu_workspace_bytes(N, groups)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    float *stats = static_cast<float *>(workspace);
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)(N * groups)), dim3(256), 0, st, x, C / groups, hw, eps, stats);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((hw + 1023) / 1024), (unsigned)(N * C)), dim3(256), 0, st, x, C, C / groups, hw, stats, gamma, beta,
                       residual, relu, y);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_linear(const float *x, const float *weight, const float *bias, int N, int in_dim, int out_dim, float *y, aoc_stream_t stream) {
    if (!x || !weight || !y || N < 1 || in_dim < 1 || out_dim < 1) return AOC_ERR_INVALID_ARG;
    if (N > 65535) return AOC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(linear_kernel, dim3(out_dim, N), dim3(64), 0, aoc_hip_stream(stream), x, weight, bias, in_dim, out_dim, y);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_label_mix(const float *labels, const float *rows, int64_t n, int n_obj, int C, float *out, aoc_stream_t stream) {
    if (!labels || !rows || !out || n < 1 || n_obj < 1 || C < 1) return AOC_ERR_INVALID_ARG;
    const int64_t total = n * C;
    hipLaunchKernelGGL(label_mix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), labels, rows, n, n_obj, C, out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_plane_mean(const float *x, int64_t planes, int64_t hw, float *out, aoc_stream_t stream) {
    if (!x || !out || planes < 1 || hw < 1) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(plane_mean4_kernel, dim3((unsigned)planes), dim3(256), 0, aoc_hip_stream(stream), x, hw, out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_head_delta(const float *head, int head_dim, const float *plane_means, int n_obj, int channels, float *out, aoc_stream_t stream) {
    if (!head || !plane_means || !out || head_dim < 1 || n_obj < 1 || channels < 1) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(head_delta_kernel, dim3((unsigned)((head_dim + channels + 255) / 256), (unsigned)n_obj), dim3(256), 0, aoc_hip_stream(stream), head,
                       head_dim, plane_means, n_obj, channels, out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_prehead_workspace_bytes(int n_obj, int n_out, int group_size, int64_t hw) {
    if (n_obj < 1 || n_out < 1 || group_size < 1 || hw < 1) return 0;
    const size_t n_chunks = (size_t)((hw + PH_PIX - 1) / PH_PIX), n_groups = (size_t)(n_out / group_size);
    return aoc_align_up((size_t)n_obj * n_groups * n_chunks * 2 * sizeof(double), 256) + aoc_align_up((size_t)n_obj * n_groups * 2 * sizeof(float), 256);
}

int aoc_prehead(const float *feat, int n_obj, int n_in, int64_t hw, const float *weight, const float *bias, int n_out, int n_groups,
                const float *gamma, const float *beta, float eps, const float *emb_hwc, int C, float *out, void *workspace,
                size_t workspace_bytes, aoc_stream_t stream) {
    if (!feat || !weight || !bias || !gamma || !beta || !out || !workspace || n_obj < 1 || n_in < 1 || n_out < 1 || n_groups < 1 || hw < 1 || C < 0)
        return AOC_ERR_INVALID_ARG;
    if (n_in > PH_MAX_IN || n_out > PH_MAX_OUT || n_out % n_groups != 0 || n_obj > 65535 || (emb_hwc == nullptr && C != 0)) return AOC_ERR_UNSUPPORTED;
    const int group_size = n_out / n_groups;
    if (workspace_bytes < aoc_prehead_workspace_bytes(n_obj, n_out, group_size, hw)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    const int n_chunks = (int)((hw + PH_PIX - 1) / PH_PIX);
    double *partial = static_cast<double *>(workspace);
    float *stats = reinterpret_cast<float *>(static_cast<char *>(workspace) + aoc_align_up((size_t)n_obj * n_groups * n_chunks * 2 * sizeof(double), 256));
#define AOC_PH(NI)                                                                                                                                        \
    do {                                                                                                                                                  \
        hipLaunchKernelGGL(prehead_stats_kernel<NI>, dim3(n_chunks, n_obj), dim3(PH_PIX), 0, st, feat, n_in, hw, weight, bias, n_out, group_size, n_chunks, partial); \
        hipLaunchKernelGGL(prehead_finalize_kernel, dim3(n_obj * n_groups), dim3(64), 0, st, partial, n_chunks, (double)group_size * (double)hw, eps, stats);         \
        hipLaunchKernelGGL(prehead_apply_kernel<NI>, dim3(n_chunks, n_obj), dim3(PH_PIX), 0, st, feat, n_in, hw, weight, bias, n_out, group_size, gamma, beta, stats, \
                           emb_hwc, C, out);                                                                                                               \
    } while (0)
    // the proto-mask tensor has 22 + 2 x levels channels (aocnet.py:341-358): compile-time channel counts for one to three levels
    if (n_in == 24) AOC_PH(24); else if (n_in == 26) AOC_PH(26); else if (n_in == 28) AOC_PH(28); else AOC_PH(0);
#undef AOC_PH
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_cond_codes(const float *gap, const float *plane_means, const float *head, const float *w1, const float *b1, const float *w2,
                   const float *b2, const float *w3, const float *b3, int N, int C, int D, float *code, aoc_stream_t stream) {
    if (!gap || !plane_means || !head || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !code || N < 1 || C < 1 || D < 1) return AOC_ERR_INVALID_ARG;
    if (N > 65535) return AOC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cond_codes_kernel, dim3(2 * C + D, N), dim3(64), 0, aoc_hip_stream(stream), gap, plane_means, head, w1, b1, w2, b2, w3, b3, N, C, D,
                       code);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_plane_reduce(const float *x, int64_t planes, int64_t hw, int mode, float *out, aoc_stream_t stream) {
    if (!x || !out || planes < 1 || hw < 1 || mode < 0 || mode > 2) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(plane_reduce_kernel, dim3((unsigned)planes), dim3(256), 0, aoc_hip_stream(stream), x, hw, mode, out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_gct_gate(const float *plane_sums, const float *alpha, const float *gamma, const float *beta, int N, int C, float eps, int l1_mode,
                 float *gate, aoc_stream_t stream) {
    if (!plane_sums || !alpha || !gamma || !beta || !gate || N < 1 || C < 1) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gct_gate_kernel, dim3((unsigned)N), dim3(256), 0, aoc_hip_stream(stream), plane_sums, alpha, gamma, beta, C, eps, l1_mode, gate);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_object_logit(const float *x, int N, int C, int64_t hw, const float *weight, int64_t weight_stride, const float *bias,
                     int64_t bias_stride, float *out, aoc_stream_t stream) {
    if (!x || !weight || !bias || !out || N < 1 || C < 1 || hw < 1) return AOC_ERR_INVALID_ARG;
    if (N > 65535) return AOC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(object_logit_kernel, dim3((unsigned)((hw + 255) / 256), N), dim3(256), 0, aoc_hip_stream(stream), x, C, hw, weight, weight_stride,
                       bias, bias_stride, out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

}  // extern "C"
