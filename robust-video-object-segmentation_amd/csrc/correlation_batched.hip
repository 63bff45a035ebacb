// Batched pixel-to-proxy correlation ("the correlation kernel", AEM:92-128 + 316-319 + the proto-mask transform of AEM:393/602/864)
// for several frames / sequences per launch, on the fp16 matrix pipe with fp32-equivalent products.
//
// Why: one 480p frame is 11.6 MB of algorithmic traffic -- a launch that small can never leave the latency regime, and exact-fp32
// MFMA makes the arithmetic (0.68 GF) slower than the HBM stream.  Here ONE persistent launch walks the 32-pixel tiles of up to 32
// frames; every fp32 value (scaled by 2^10) is split once into hi + lo fp16 in registers and
//     q.p = qh.ph + qh.pl + ql.ph          (the dropped ql.pl term is < 2^-22 |q.p|)
// is ONE K = 304 dot product per (pixel, proxy): 3 x 100 product slots + 3 slots that carry -|p|^2 / 2, i.e. 19 k-steps of
// v_mfma_f32_32x32x16_f16 instead of 3 x 7.  The accumulator holds 2^20 (q.p - |p|^2 / 2), so the min over a set's proxies is a
// max over raw accumulator registers and  d = |q|^2 - 2^-19 max.
//
// Operand roles: A = proxies (rows i of the 32x32 tile, read from an LDS image staged once per block and frame), B = query pixels
// (columns j; built in registers from coalescing-friendly 16-byte global loads).  D register r of lane l is row (r/4)*8 + (l/32)*4 + r%4,
// column l%32: a set that occupies whole 8-row groups is reduced in-lane over registers plus ONE exchange between the two lane halves,
// and the 32 lanes of a half then store 32 consecutive pixels of the set's output plane (128-byte runs).
//
// The k dimension may be permuted freely as long as both operands agree.  Lane half h (= lane / 32) owns channels 48h .. 48h+47 and
// 96+2h, 97+2h (twelve 16-byte loads + one 8-byte load per pixel row, all naturally aligned) and supplies, per k-step s, the dwords
// [4s, 4s+4) of the 76-dword sequence  [ hi(50 ch) | hi(50 ch) | lo(50 ch) | norm-slot constants ]  (query side)  against
// [ hi | lo | hi | pieces of -16 |p|^2 ]  (proxy side, LDS image, 624-byte rows: conflict-free ds_read_b128).
//
// Preconditions of the split arithmetic (|x| 2^10 <= 65000, |x|^2 <= 4000) are checked on the device for every value the kernel
// touches; a violation raises a flag and the exact-fp32 kernel of correlation.hip, gated on that flag, recomputes the launch.
#include <stdlib.h>

#include "aoc_common.h"
#include "correlation_shared.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CB_NW = 8;                         // waves per block (one block per CU: the proxy image fills most of the LDS)
constexpr int CB_NT = CB_NW * 64;
constexpr int CB_CH = 50;                        // channels per lane half (C = 100)
constexpr int CB_PK = CB_CH / 2;                 // packed fp16 pairs per plane
constexpr int CB_STEPS = 19;                     // k-steps: (3 * 100 + 3 + 1 pad) / 16
constexpr int CB_HALF_DW = CB_STEPS * 4;         // 76 dwords per (row, lane half)
constexpr int CB_ROW_DW = 2 * CB_HALF_DW + 4;    // 156 dwords = 624 B: 16 consecutive rows start on distinct 4-bank groups
constexpr int CB_TILE_DW = 32 * CB_ROW_DW;
constexpr float CB_SCALE = 1024.0f;              // 2^10
constexpr float CB_QCONST = 32768.0f;            // query-side value of the norm slots: 2^15 * (-16 |p|^2) = -2^19 |p|^2
constexpr float CB_UNSCALE = -1.0f / 524288.0f;  // d - |q|^2 = -2^-19 * accumulator
constexpr float CB_MAX_ABS = 65000.0f / 1024.0f;
constexpr float CB_MAX_SQ = 4000.0f;

__device__ __forceinline__ uint32_t pack_f16(_Float16 a, _Float16 b) {
    union { _Float16 h[2]; uint32_t u; } x;
    x.h[0] = a; x.h[1] = b;
    return x.u;
}

// 50 fp32 channels of one lane half -> hi / lo packed planes, |x|^2 partial and max |x|
struct SplitHalf {
    uint32_t hi[CB_PK], lo[CB_PK];
    float sq, amax;
};
__device__ __forceinline__ void split_half(const float (&v)[CB_CH], SplitHalf &o) {
    float sq = 0.0f, amax = 0.0f;
#pragma unroll
    for (int e = 0; e < CB_PK; ++e) {
        const float x0 = v[2 * e], x1 = v[2 * e + 1];
        sq = __builtin_fmaf(x0, x0, sq);
        sq = __builtin_fmaf(x1, x1, sq);
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)));
        const float s0 = x0 * CB_SCALE, s1 = x1 * CB_SCALE;
        const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
        o.hi[e] = pack_f16(h0, h1);
        o.lo[e] = pack_f16((_Float16)(s0 - (float)h0), (_Float16)(s1 - (float)h1));
    }
    o.sq = sq;
    o.amax = amax;
}

// the 50 channels of lane half h of one fp32 row: 12 x 16 B at float offset 48 h + 4 t, then 8 B at 96 + 2 h
__device__ __forceinline__ void load_half_row(const float *__restrict__ row, int h, float (&v)[CB_CH]) {
    const float4 *p4 = reinterpret_cast<const float4 *>(row + 48 * h);
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const float4 x = p4[t];
        v[4 * t] = x.x; v[4 * t + 1] = x.y; v[4 * t + 2] = x.z; v[4 * t + 3] = x.w;
    }
    const float2 y = *reinterpret_cast<const float2 *>(row + 96 + 2 * h);
    v[48] = y.x; v[49] = y.y;
}

struct CbWork {      // work range of one block: global tile index g = frame * tiles_per_frame + tile
    int64_t g0, g1;
};

__global__ __launch_bounds__(CB_NT) void proxy_corr_batched_kernel(AocCorrFrames frames, int64_t m, AocCorrTiles tiles, int transform,
                                                                    int32_t *__restrict__ gate, int n_blocks_virtual) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int n_rows = tiles.n * 32;
    uint32_t *limg = lds;                                                     // [n_rows][CB_ROW_DW]
    int32_t *lsrc = reinterpret_cast<int32_t *>(limg + (size_t)tiles.n * CB_TILE_DW);   // [n_rows] proxy row feeding each image row (-1: none)
    float *lnorm = reinterpret_cast<float *>(lsrc + n_rows);                  // [n_rows] |p|^2 of that proxy
    int32_t *lfirst = reinterpret_cast<int32_t *>(lnorm + n_rows);            // [AOC_CORR_MAX_OUT] first valid image row of each output column (-1: absent)
    float *lbias = reinterpret_cast<float *>(lfirst + AOC_CORR_MAX_OUT);      // [AOC_CORR_MAX_OUT]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t T = (m + 31) >> 5;                                          // 32-pixel tiles per frame
    const int64_t total = T * frames.n;
    // XCD-aware virtual block id: workgroups are dealt round-robin to the 8 XCDs; give every XCD a contiguous range of the work list so
    // the blocks that share a frame's proxy table (and write neighbouring output planes) share one L2
    int vb;
    {
        const int nb = gridDim.x, lin = blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3, q = nb >> 3, r = nb & 7;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int64_t g0 = total * vb / gridDim.x, g1 = total * (vb + 1) / gridDim.x;
    if (g0 >= g1) return;
    const int f_beg = (int)(g0 / T), f_end = (int)((g1 - 1) / T) + 1;

    // this wave's tiles: within frame f the tiles [lo_f, hi_f) of the block's range, taken round-robin by the 8 waves
    auto seg_lo = [&](int f) -> int64_t { return f == f_beg ? g0 - (int64_t)f * T : 0; };
    auto seg_hi = [&](int f) -> int64_t { return f == f_end - 1 ? g1 - (int64_t)f * T : T; };
    // first tile of this wave at or after frame f (returns frame in nf, tile in nt; nf = f_end when there is none)
    auto first_tile_from = [&](int f, int &nf, int64_t &nt) {
        for (; f < f_end; ++f) {
            const int64_t t = seg_lo(f) + wave;
            if (t < seg_hi(f)) { nf = f; nt = t; return; }
        }
        nf = f_end; nt = 0;
    };

    float raw[CB_CH];
    auto issue_pixel_loads = [&](int f, int64_t tile) {
        int64_t pix = tile * 32 + j;
        if (pix > m - 1) pix = m - 1;
        load_half_row(frames.f[f].query + pix * 100, h, raw);
    };

    int nf;
    int64_t nt;
    first_tile_from(f_beg, nf, nt);
    if (nf < f_end) issue_pixel_loads(nf, nt);          // in flight under the first staging pass

    bool bad = false;
    for (int f = f_beg; f < f_end; ++f) {
        const AocCorrFrame fr = frames.f[f];
        // ---- stage this frame's proxy image -----------------------------------------------------------------
        // phase 1: which proxy feeds each image row, its norm
        for (int r = threadIdx.x; r < n_rows; r += CB_NT) {
            const AocCorrTile &tl = tiles.t[r >> 5];
            const int rr = r & 31;
            int src = -1;
            if (tl.kind == 1) {
                if (rr < tl.cnt[0]) src = tl.begin[0] + rr;
            } else {
                const int g = rr >> 3, e = rr & 7;
                if (e < tl.cnt[g]) src = tl.begin[g] + e;
            }
            float nrm = INFINITY;
            if (src >= 0) {
                if (fr.sqnorm) {
                    nrm = fr.sqnorm[src];
                } else {
                    const float4 *p = reinterpret_cast<const float4 *>(fr.proxies + (size_t)src * 100);
                    float s = 0.0f;
                    for (int t = 0; t < 25; ++t) {
                        const float4 x = p[t];
                        s = __builtin_fmaf(x.x, x.x, s); s = __builtin_fmaf(x.y, x.y, s); s = __builtin_fmaf(x.z, x.z, s); s = __builtin_fmaf(x.w, x.w, s);
                    }
                    nrm = s;
                }
                if (!(nrm < INFINITY)) src = -1;                              // +inf (or NaN) norm: proxy absent (AEM:271-273, 283-286)
                else if (nrm > CB_MAX_SQ) bad = true;
            }
            lsrc[r] = src;
            lnorm[r] = nrm;
        }
        __syncthreads();
        // phase 2: per output column the first valid row of its set (pads of the set's row groups are filled with a copy of it:
        // a duplicate never changes a min); columns of a column-wise tile are their own set
        for (int oc = threadIdx.x; oc < tiles.n_out; oc += CB_NT) {
            const int r0 = tiles.oc_row0[oc], nr = tiles.oc_rows[oc];
            int first = -1;
            for (int r = r0; r < r0 + nr; ++r)
                if (lsrc[r] >= 0) { first = r; break; }
            lfirst[oc] = first;
            lbias[oc] = fr.bias ? fr.bias[tiles.oc_bias[oc]] : 0.0f;
        }
        __syncthreads();
        // phase 3: (row, half) items -> image
        for (int it = threadIdx.x; it < n_rows * 2; it += CB_NT) {
            const int r = it >> 1, hh = it & 1;
            const AocCorrTile &tl = tiles.t[r >> 5];
            int srow = r;                                                     // image row whose proxy is copied here
            if (lsrc[r] < 0 && tl.kind == 0) {
                const int first = lfirst[tl.oc[(r & 31) >> 3]];
                srow = first;
            }
            uint32_t *dst = limg + (size_t)r * CB_ROW_DW + hh * CB_HALF_DW;
            const int src = srow >= 0 ? lsrc[srow] : -1;
            if (src < 0) {
#pragma unroll
                for (int s = 0; s < CB_STEPS; ++s) reinterpret_cast<uint4 *>(dst)[s] = make_uint4(0, 0, 0, 0);
                continue;
            }
            float v[CB_CH];
            load_half_row(fr.proxies + (size_t)src * 100, hh, v);
            SplitHalf sp;
            split_half(v, sp);
            if (sp.amax > CB_MAX_ABS || !(sp.amax == sp.amax)) bad = true;
            const float a = -16.0f * lnorm[srow];
            const _Float16 n0 = (_Float16)a;
            const _Float16 n1 = (_Float16)(a - (float)n0);
            const _Float16 n2 = (_Float16)((a - (float)n0) - (float)n1);
            uint32_t seq[CB_HALF_DW];
#pragma unroll
            for (int e = 0; e < CB_PK; ++e) { seq[e] = sp.hi[e]; seq[CB_PK + e] = sp.lo[e]; seq[2 * CB_PK + e] = sp.hi[e]; }
            seq[3 * CB_PK] = hh == 0 ? pack_f16(n0, n1) : pack_f16(n2, (_Float16)0.0f);
#pragma unroll
            for (int s = 0; s < CB_STEPS; ++s) reinterpret_cast<uint4 *>(dst)[s] = make_uint4(seq[4 * s], seq[4 * s + 1], seq[4 * s + 2], seq[4 * s + 3]);
        }
        __syncthreads();

        // ---- this wave's pixel tiles of frame f ----------------------------------------------------------------
        const int64_t hi_f = seg_hi(f);
        for (int64_t tile = seg_lo(f) + wave; tile < hi_f; tile += CB_NW) {
            // raw holds this tile (issued one tile ago): split it into the B operand sequence
            SplitHalf sp;
            split_half(raw, sp);
            if (sp.amax > CB_MAX_ABS || !(sp.amax == sp.amax)) bad = true;
            const float q2 = sp.sq + __shfl_xor(sp.sq, 32);
            if (!(q2 <= CB_MAX_SQ)) bad = true;
            uint32_t seq[CB_HALF_DW];
#pragma unroll
            for (int e = 0; e < CB_PK; ++e) { seq[e] = sp.hi[e]; seq[CB_PK + e] = sp.hi[e]; seq[2 * CB_PK + e] = sp.lo[e]; }
            seq[3 * CB_PK] = h == 0 ? pack_f16((_Float16)CB_QCONST, (_Float16)CB_QCONST) : pack_f16((_Float16)CB_QCONST, (_Float16)0.0f);
            // next tile's loads: in flight under this tile's MFMAs
            if (tile + CB_NW < hi_f) { nf = f; nt = tile + CB_NW; }
            else first_tile_from(f + 1, nf, nt);
            if (nf < f_end) issue_pixel_loads(nf, nt);

            const int64_t pix = tile * 32 + j;
            const bool live = pix < m;
            float carry = -INFINITY;
            for (int ti = 0; ti < tiles.n; ++ti) {
                const AocCorrTile tl = tiles.t[ti];
                const uint4 *arow = reinterpret_cast<const uint4 *>(limg + (size_t)(ti * 32 + j) * CB_ROW_DW + h * CB_HALF_DW);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                f16x8 a0 = __builtin_bit_cast(f16x8, arow[0]);
                f16x8 a1 = __builtin_bit_cast(f16x8, arow[1]);
#pragma unroll
                for (int s = 0; s < CB_STEPS; ++s) {
                    f16x8 a2 = a1;
                    if (s + 2 < CB_STEPS) a2 = __builtin_bit_cast(f16x8, arow[s + 2]);
                    const f16x8 b = __builtin_bit_cast(f16x8, make_uint4(seq[4 * s], seq[4 * s + 1], seq[4 * s + 2], seq[4 * s + 3]));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc, 0, 0, 0);
                    a0 = a1; a1 = a2;
                }
                if (tl.kind == 1) {
                    // column-wise: every row is its own output (k = 1 proxies, no min: AEM:127)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r >> 2) * 8 + h * 4 + (r & 3);
                        if (row < tl.cnt[0] && live) {
                            const int oc = tl.oc[0] + row;
                            float d = lfirst[oc] >= 0 ? q2 + CB_UNSCALE * acc[r] : AOC_PAD_DISTANCE;
                            if (transform) d = aoc_proto_transform(d, lbias[oc]);
                            fr.out[tiles.oc_offset[oc] + pix] = d;
                        }
                    }
                } else {
                    // grouped: group g = rows 8g .. 8g+7 = registers 4g .. 4g+3 of both lane halves
                    float gm[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        gm[g] = __builtin_fmaxf(__builtin_fmaxf(acc[4 * g], acc[4 * g + 1]), __builtin_fmaxf(acc[4 * g + 2], acc[4 * g + 3]));
                    if (tl.gs == 4) {
                        float v = __builtin_fmaxf(__builtin_fmaxf(gm[0], gm[1]), __builtin_fmaxf(gm[2], gm[3]));
                        v = __builtin_fmaxf(v, __shfl_xor(v, 32));
                        if (!tl.first) v = __builtin_fmaxf(v, carry);
                        carry = v;
                        if (tl.last && h == 0 && live) {
                            const int oc = tl.oc[0];
                            float d = lfirst[oc] >= 0 ? q2 + CB_UNSCALE * v : AOC_PAD_DISTANCE;
                            if (transform) d = aoc_proto_transform(d, lbias[oc]);
                            fr.out[tiles.oc_offset[oc] + pix] = d;
                        }
                    } else if (tl.gs == 2) {
                        float v0 = __builtin_fmaxf(gm[0], gm[1]), v1 = __builtin_fmaxf(gm[2], gm[3]);
                        v0 = __builtin_fmaxf(v0, __shfl_xor(v0, 32));
                        v1 = __builtin_fmaxf(v1, __shfl_xor(v1, 32));
                        const float v = h == 0 ? v0 : v1;                     // lane half h stores set h
                        const int oc = tl.oc[2 * h];
                        if (oc >= 0 && live) {
                            float d = lfirst[oc] >= 0 ? q2 + CB_UNSCALE * v : AOC_PAD_DISTANCE;
                            if (transform) d = aoc_proto_transform(d, lbias[oc]);
                            fr.out[tiles.oc_offset[oc] + pix] = d;
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) gm[g] = __builtin_fmaxf(gm[g], __shfl_xor(gm[g], 32));
#pragma unroll
                        for (int u = 0; u < 2; ++u) {                         // lane half h stores sets 2h, 2h + 1
                            const float v = h == 0 ? gm[u] : gm[2 + u];
                            const int oc = tl.oc[2 * h + u];
                            if (oc >= 0 && live) {
                                float d = lfirst[oc] >= 0 ? q2 + CB_UNSCALE * v : AOC_PAD_DISTANCE;
                                if (transform) d = aoc_proto_transform(d, lbias[oc]);
                                fr.out[tiles.oc_offset[oc] + pix] = d;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();                                                      // every wave is done with this frame's image
    }
    if (bad) atomicOr(gate, 1);
}

inline int cb_n_cus() {
    static int n = 0;
    if (n == 0) {
        const char *e = getenv("AOC_CORR_CUS");                               // CUs the launching stream may use (HIP CU mask)
        if (e && atoi(e) > 0) n = atoi(e);
        else {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
            else n = 256;
        }
    }
    return n;
}

}  // namespace

extern "C" {

size_t aoc_proxy_corr_min_batched_workspace_bytes(void) { return 256; }

int aoc_proxy_corr_min_batched(const aoc_corr_frame *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                               const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                               int transform, int precision, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!frames_host || !set_begin_host || !set_size_host || !set_out_offset_host) return AOC_ERR_INVALID_ARG;
    if (n_frames < 1 || m < 1 || n_set < 1 || n_proxy < 0 || C < 4) return AOC_ERR_INVALID_ARG;
    if ((C & 3) || C > AOC_MAX_CHANNELS) return AOC_ERR_UNSUPPORTED;
    if (precision != AOC_CORR_SPLIT && precision != AOC_CORR_FP32) return AOC_ERR_INVALID_ARG;
    for (int s = 0; s < n_set; ++s)
        if (set_size_host[s] < 0 || set_begin_host[s] < 0 || set_begin_host[s] + set_size_host[s] > n_proxy) return AOC_ERR_INVALID_ARG;
    for (int f = 0; f < n_frames; ++f)
        if (!frames_host[f].query || !frames_host[f].proxies || !frames_host[f].out) return AOC_ERR_INVALID_ARG;
    hipStream_t st = aoc_hip_stream(stream);

    bool split_ok = precision == AOC_CORR_SPLIT && C == 100 && workspace && workspace_bytes >= aoc_proxy_corr_min_batched_workspace_bytes();
    for (int f = 0; f < n_frames && split_ok; ++f)
        if ((reinterpret_cast<uintptr_t>(frames_host[f].query) | reinterpret_cast<uintptr_t>(frames_host[f].proxies)) & 15) split_ok = false;
    if (!split_ok) {
        if (precision == AOC_CORR_SPLIT && (!workspace || workspace_bytes < aoc_proxy_corr_min_batched_workspace_bytes())) return AOC_ERR_WORKSPACE;
        return aoc_corr_fp32_batched(frames_host, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, 1, transform,
                                     nullptr, stream);
    }

    int32_t *gate = static_cast<int32_t *>(workspace);
    if (hipMemsetAsync(gate, 0, 16, st) != hipSuccess) return AOC_ERR_LAUNCH;

    // ---- pack the sets into 32-row tiles: single-proxy sets -> column-wise tiles, the others by row-group class
    const size_t tile_bytes = (size_t)CB_TILE_DW * 4 + 32 * 8;
    int max_tiles = (int)(((size_t)160 * 1024 - AOC_CORR_MAX_OUT * 8 - 64) / tile_bytes);
    if (max_tiles > AOC_CORR_MAX_TILES) max_tiles = AOC_CORR_MAX_TILES;
    const int n_cu = cb_n_cus();
    const int64_t T = (m + 31) / 32;

    for (int f0 = 0; f0 < n_frames; f0 += AOC_CORR_MAX_FRAMES) {
        AocCorrFrames fr;
        fr.n = n_frames - f0 < AOC_CORR_MAX_FRAMES ? n_frames - f0 : AOC_CORR_MAX_FRAMES;
        for (int f = 0; f < fr.n; ++f) {
            fr.f[f].query = frames_host[f0 + f].query;
            fr.f[f].proxies = frames_host[f0 + f].proxies;
            fr.f[f].sqnorm = frames_host[f0 + f].proxy_sqnorm;
            fr.f[f].bias = frames_host[f0 + f].set_bias;
            fr.f[f].out = frames_host[f0 + f].out;
        }
        AocCorrTiles tab;
        auto reset = [&]() { tab.n = 0; tab.n_out = 0; };
        auto flush = [&]() -> int {
            if (tab.n == 0) return AOC_OK;
            const size_t lds = (size_t)tab.n * tile_bytes + AOC_CORR_MAX_OUT * 8;
            int64_t grid = T * fr.n;
            if (grid > n_cu) grid = n_cu;
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(proxy_corr_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return AOC_ERR_LAUNCH;
            hipLaunchKernelGGL(proxy_corr_batched_kernel, dim3((unsigned)grid), dim3(CB_NT), lds, st, fr, m, tab, transform, gate, 0);
            reset();
            return hipGetLastError() == hipSuccess ? AOC_OK : AOC_ERR_LAUNCH;
        };
        auto add_out = [&](int64_t off, int set, int row0, int rows) -> int {
            tab.oc_offset[tab.n_out] = off;
            tab.oc_bias[tab.n_out] = set;
            tab.oc_row0[tab.n_out] = (int16_t)row0;
            tab.oc_rows[tab.n_out] = (int16_t)rows;
            return tab.n_out++;
        };
        reset();
        // pass over the row-group classes: 1 = single proxies (column-wise), then sets of <= 8, <= 16, > 16 proxies
        for (int cls = 0; cls < 4; ++cls) {
            int open_tile = -1, open_groups = 0;                              // grouped tile being filled (classes 1, 2)
            for (int s = 0; s < n_set; ++s) {
                const int size = set_size_host[s];
                const int c = size == 1 ? 0 : size <= 8 ? 1 : size <= 16 ? 2 : 3;
                if (c != cls) continue;
                if (cls == 0) {
                    // runs of single-proxy sets over consecutive proxies share a column-wise tile
                    const bool cont = open_tile >= 0 && tab.t[open_tile].cnt[0] < 32 && set_begin_host[s] == tab.t[open_tile].begin[0] + tab.t[open_tile].cnt[0];
                    if (!cont || tab.n_out + 1 > AOC_CORR_MAX_OUT) {
                        if (tab.n + 1 > max_tiles || tab.n_out + 1 > AOC_CORR_MAX_OUT) { int rc = flush(); if (rc) return rc; }
                        open_tile = tab.n++;
                        AocCorrTile &tl = tab.t[open_tile];
                        tl = AocCorrTile{};
                        tl.kind = 1; tl.gs = 0; tl.first = 1; tl.last = 1;
                        tl.begin[0] = set_begin_host[s];
                        tl.cnt[0] = 0;
                        tl.oc[0] = (int16_t)tab.n_out;
                    }
                    AocCorrTile &tl = tab.t[open_tile];
                    add_out(set_out_offset_host[s], s, open_tile * 32 + tl.cnt[0], 1);
                    tl.cnt[0]++;
                } else if (cls == 1 || cls == 2) {
                    const int gs = cls;                                       // row groups per set
                    if (open_tile < 0 || open_groups + gs > 4 || tab.n_out + 1 > AOC_CORR_MAX_OUT) {
                        if (tab.n + 1 > max_tiles || tab.n_out + 1 > AOC_CORR_MAX_OUT) { int rc = flush(); if (rc) return rc; }
                        open_tile = tab.n++;
                        open_groups = 0;
                        AocCorrTile &tl = tab.t[open_tile];
                        tl = AocCorrTile{};
                        tl.kind = 0; tl.gs = gs; tl.first = 1; tl.last = 1;
                        for (int g = 0; g < 4; ++g) { tl.oc[g] = -1; tl.cnt[g] = 0; tl.begin[g] = 0; }
                    }
                    AocCorrTile &tl = tab.t[open_tile];
                    const int oc = add_out(set_out_offset_host[s], s, open_tile * 32 + open_groups * 8, gs * 8);
                    for (int g = 0; g < gs; ++g) {
                        const int left = size - 8 * g;
                        tl.begin[open_groups + g] = set_begin_host[s] + 8 * g;
                        tl.cnt[open_groups + g] = (int16_t)(left < 0 ? 0 : left > 8 ? 8 : left);
                        tl.oc[open_groups + g] = (int16_t)oc;
                    }
                    open_groups += gs;
                } else {
                    const int nt = size == 0 ? 1 : (size + 31) / 32;
                    if (nt > max_tiles) return AOC_ERR_UNSUPPORTED;
                    if (tab.n + nt > max_tiles || tab.n_out + 1 > AOC_CORR_MAX_OUT) { int rc = flush(); if (rc) return rc; }
                    const int oc = add_out(set_out_offset_host[s], s, tab.n * 32, nt * 32);
                    for (int t = 0; t < nt; ++t) {
                        AocCorrTile &tl = tab.t[tab.n++];
                        tl = AocCorrTile{};
                        tl.kind = 0; tl.gs = 4; tl.first = t == 0; tl.last = t == nt - 1;
                        for (int g = 0; g < 4; ++g) {
                            const int left = size - 32 * t - 8 * g;
                            tl.begin[g] = set_begin_host[s] + 32 * t + 8 * g;
                            tl.cnt[g] = (int16_t)(left < 0 ? 0 : left > 8 ? 8 : left);
                            tl.oc[g] = (int16_t)oc;
                        }
                    }
                    open_tile = -1;
                }
            }
        }
        int rc = flush();
        if (rc) return rc;
    }
    // exact-fp32 kernel: runs only when a precondition of the split arithmetic failed somewhere in the launch
    return aoc_corr_fp32_batched(frames_host, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, 1, transform,
                                 gate, stream);
}

}  // extern "C"
