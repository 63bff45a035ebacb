// Batched pixel-to-proxy correlation ("the correlation kernel", AEM:92-128 + 316-319 + the proto-mask transform of AEM:393/602/864)
// for several frames / sequences per launch, on the fp16 matrix pipe with fp32-equivalent products.
//
// Why: one 480p frame is 11.6 MB of algorithmic traffic -- a launch that small can never leave the latency regime, and exact-fp32
// MFMA makes the arithmetic (0.68 GF) slower than the HBM stream.  Here ONE persistent launch walks the 32-pixel tiles of up to 32
// frames; every fp32 value (scaled by 2^10) is split once into hi + lo fp16 in registers and
//     q.p = qh.ph + qh.pl + ql.ph          (the dropped ql.pl term is < 2^-22 |q.p|)
// is three chained products of K = 112 each (100 channels + 3 slots that carry -|p|^2 / 2 + padding): 21 k-steps of
// v_mfma_f32_32x32x16_f16 on ONE accumulator, the hi plane of each side serving two products from the same registers / LDS bytes.
// The accumulator holds 2^20 (q.p - |p|^2 / 2), so the min over a set's proxies is a max over raw accumulator registers and
// d = |q|^2 - 2^-19 max.
//
// Operand roles: A = proxies (rows i of the 32x32 tile, read from an LDS image staged once per block and frame), B = query pixels
// (columns j; built in registers from coalescing-friendly 16-byte global loads).  D register r of lane l is row (r/4)*8 + (l/32)*4 + r%4,
// column l%32: a set that occupies whole 8-row groups is reduced in-lane over registers plus ONE exchange between the two lane halves,
// and the 32 lanes of a half then store 32 consecutive pixels of the set's output plane (128-byte runs).
//
// The k dimension may be permuted freely as long as both operands agree.  Lane half h (= lane / 32) owns channels 48h .. 48h+47 and
// 96+2h, 97+2h and holds two 28-dword planes  hi(50 ch) + norm slots + pad | lo(50 ch) + pad  (query side: 56 registers; proxy side:
// LDS image, 464-byte rows: conflict-free ds_read_b128).
//
// Preconditions of the split arithmetic (|x| 2^10 <= 65000, |x|^2 <= 4000) are checked on the device for every value the kernel
// touches; a violation raises a flag and the exact-fp32 kernel of correlation.hip, gated on that flag, recomputes the launch.
#include <stdlib.h>

#include <atomic>

#include <cstring>
#include <functional>

#include "aoc_common.h"
#include "correlation_shared.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CB_NW = 12;                        // waves per block (one block per CU, three waves per SIMD: 168 VGPRs each)
constexpr int CB_CH = 50;                        // channels per lane half (C = 100)
constexpr int CB_PK = CB_CH / 2;                 // packed fp16 pairs per plane
constexpr int CB_SEG_STEPS = 7;                  // k-steps per plane: 56 fp16 slots = 50 channels + norm slots + pad
constexpr int CB_SEG_DW = CB_SEG_STEPS * 4;      // 28 dwords per plane
constexpr int CB_HALF_DW = 2 * CB_SEG_DW;        // [hi | lo] planes of one (row, lane half): 56 dwords
constexpr int CB_ROW_DW = 2 * CB_HALF_DW + 4;    // 116 dwords = 464 B: 16 consecutive rows start on distinct 4-bank groups
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

// (x0, x1) -> packed fp16 pairs hi = fp16(2^10 x), lo = fp16(2^10 x - hi) in FOUR instructions: the mixed-precision FMA takes the fp32
// value, the fp32 scale and the fp16 hi piece (read from its half of the packed register) directly, rounds once to fp16 and writes one
// half of the destination -- the same values as scale / convert / convert back / subtract / convert / pack (15 instructions per pair as
// hipcc compiles the scalar casts): 2^10 x is exact, and so is 2^10 x - hi in fp32 (the low bits of a 24-bit significand)
__device__ __forceinline__ void cb_split_pair(float x0, float x1, uint32_t &hi, uint32_t &lo) {
    const float scale = CB_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "s"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "s"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "s"(scale), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "s"(scale), "v"(hi));
}

// the 50 channels of lane half h of one fp32 row in global memory: 12 x 16 B at float offset 48 h + 4 t, then 8 B at 96 + 2 h
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

struct PixelSeq {            // B operands of one pixel tile: hi plane (7 MFMA operands; dword 25 = norm-slot constants) and lo plane + |q|^2
    u32x4 bh[CB_SEG_STEPS], bl[CB_SEG_STEPS];
    float q2part;            // this lane half's share of |q|^2
};

// the 50 fp32 channels of this lane (pixel j, channel half h) -> packed hi / hi / lo planes of the operand sequence, |q|^2 share, max |x|
__device__ __forceinline__ void convert_raw(const float (&x)[CB_CH], PixelSeq &o) {
    // |x| 2^10 <= 65000 needs no test of its own on the query side: |q|^2 <= 4000 (checked per pixel) already implies |x| <= 63.25
    float sq = 0.0f, sq1 = 0.0f;
#pragma unroll
    for (int e = 0; e < CB_PK; ++e) {
        const float x0 = x[2 * e], x1 = x[2 * e + 1];
        sq = __builtin_fmaf(x0, x0, sq);
        sq1 = __builtin_fmaf(x1, x1, sq1);
        uint32_t hi, lo;
        cb_split_pair(x0, x1, hi, lo);
        o.bh[e >> 2][e & 3] = hi;
        o.bl[e >> 2][e & 3] = lo;
    }
    o.bh[6][2] = 0u; o.bh[6][3] = 0u;                                      // dwords 26, 27: padding (dword 25 holds the norm-slot constants)
    o.bl[6][1] = 0u; o.bl[6][2] = 0u; o.bl[6][3] = 0u;
    o.q2part = sq + sq1;
}

// One 32-pixel query tile is 32 x 400 = 12 800 CONTIGUOUS bytes: a wave fetches it with 12.5 fully coalesced 16-byte loads per lane
// (1 KiB per instruction) and transposes it through its private 6.4 KB LDS buffer in two halves of 16 pixel rows: the lanes that own
// pixels 0..15 read their (pixel, channel half) piece back after the first half is parked, the others after the second.
constexpr int CB_FLATB = 7;                      // 16-byte chunks per lane of ONE half tile (the 7th only for lanes 0..15)
constexpr int CB_TILE_BYTES = 32 * 400;
constexpr int CB_CHUNKS = CB_TILE_BYTES / 16;    // 800
constexpr int CB_HALF_CHUNKS = CB_CHUNKS / 2;    // 400: chunks of the first 16 pixel rows
constexpr int CB_WBUF_BYTES = CB_TILE_BYTES / 2;
constexpr int CB_TABLE_DW = (int)((sizeof(AocCorrTiles) + 15) / 16 * 4);      // the tile tables' copy at the start of the LDS allocation

struct CbSlotDesc {          // epilogue constants of one (proxy tile, lane half), staged per frame: two store slots
    int32_t off_a, off_b;    // element offset of the slot's output plane in the frame's `out` (-1: no store)
    float bias_a, bias_b;
    int32_t valid_a, valid_b;   // the set has at least one proxy (else the constant 5e4, AEM:310-313)
    int32_t pad0, pad1;
};
struct CbColDesc {           // one row of the column-wise tile
    int32_t off;
    float bias;
    int32_t valid, pad;
};

// (sigmoid(d + bias) - 0.5) * 2 with the hardware exp2 / rcp (about 1e-7 absolute on the output; d + bias is a squared distance plus a
// small bias, so the exponent argument is accurate where the result is not saturated)
__device__ __forceinline__ float cb_transform(float d, float bias) {
    const float e = __builtin_amdgcn_exp2f((d + bias) * -1.44269504088896341f);
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + e) - 1.0f;
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to the LDS bytes [lds_dst + 16 lane, +16): no staging registers.  The
// transfer is an asm statement hipcc does not count; the kernel only relies on loads completing in order (see convert_tile).
__device__ __forceinline__ void cb_glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// lanes 0..31 get max over both lane halves of a, lanes 32..63 of b (one v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ float cb_halfmax2(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// every lane gets the sum over both lane halves
__device__ __forceinline__ float cb_halfsum(float a) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct CbImage {            // the block's LDS image of one frame's proxies and its epilogue tables
    AocCorrTiles *tiles;
    uint32_t *img;           // [n_rows][CB_ROW_DW]
    int32_t *src;            // [n_rows] proxy row feeding each image row (-1: none)
    float *norm;             // [n_rows] |p|^2 of that proxy
    int32_t *first;          // [AOC_CORR_MAX_OUT] first valid image row of each output column (-1: absent)
    CbSlotDesc *slot;        // [NT][2]
    CbColDesc *col;          // [32]
};

// Stage one frame's proxy image (all threads of the block; ends with a barrier).  REC: k order of the split records (the query side is
// consumed straight from aoc_split_rows_tiled records); else the lane-half order of proxy_corr_batched_kernel.
template <int NTH, int NT, bool COL0, bool REC>
__device__ __forceinline__ void cb_stage_frame(const CbImage &L, const float *__restrict__ proxies, const float *__restrict__ sqnorm,
                                               const float *__restrict__ bias, bool &bad) {
    constexpr int n_rows = NT * 32;
    // phase 1: which proxy feeds each image row, its norm
    for (int r = threadIdx.x; r < n_rows; r += NTH) {
        const AocCorrTile &tl = L.tiles->t[r >> 5];
        const int rr = r & 31;
        int src = -1;
        if (tl.kind == 1) {
            if (rr < tl.cnt[0]) src = tl.begin[0] + rr;
            else if (tl.gs == 1 && rr >= 16 && rr - 16 < tl.cnt[0]) src = tl.begin[0] + rr - 16;   // stacked: row 16 + i = the lo pieces of proxy i
        } else {
            const int g = rr >> 3, e = rr & 7;
            if (e < tl.cnt[g]) src = tl.begin[g] + e;
        }
        float nrm = INFINITY;
        if (src >= 0) {
            if (sqnorm) {
                nrm = sqnorm[src];
            } else {
                const float4 *p = reinterpret_cast<const float4 *>(proxies + (size_t)src * 100);
                float sacc = 0.0f;
                for (int t = 0; t < 25; ++t) {
                    const float4 x = p[t];
                    sacc = __builtin_fmaf(x.x, x.x, sacc); sacc = __builtin_fmaf(x.y, x.y, sacc);
                    sacc = __builtin_fmaf(x.z, x.z, sacc); sacc = __builtin_fmaf(x.w, x.w, sacc);
                }
                nrm = sacc;
            }
            if (!(nrm < INFINITY)) src = -1;                              // +inf (or NaN) norm: proxy absent (AEM:271-273, 283-286)
            else if (nrm > CB_MAX_SQ) bad = true;
        }
        L.src[r] = src;
        L.norm[r] = nrm;
    }
    __syncthreads();
    // phase 2: per output column the first valid row of its set (pads of the set's row groups are filled with a copy of it:
    // a duplicate never changes a min); columns of a column-wise tile are their own set
    for (int oc = threadIdx.x; oc < L.tiles->n_out; oc += NTH) {
        const int r0 = L.tiles->oc_row0[oc], nr = L.tiles->oc_rows[oc];
        int first = -1;
        for (int r = r0; r < r0 + nr; ++r)
            if (L.src[r] >= 0) { first = r; break; }
        L.first[oc] = first;
    }
    __syncthreads();
    // phase 3: one item per (image row, float4 of the proxy row): consecutive threads read consecutive 16 bytes; the two packed
    // hi pairs land at dwords e, e+1 of the half's hi plane, the lo pairs at the same place of its lo plane
    for (int it0 = threadIdx.x; it0 < n_rows * 25; it0 += 4 * NTH) {
        // four items per trip: their global loads are all in flight before the first is converted
        float4 xs[4];
        int srcs[4], srows[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = it0 + u * NTH;
            xs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            srcs[u] = -1; srows[u] = -1;
            if (it < n_rows * 25) {
                const int r = it / 25, t = it - r * 25;
                const AocCorrTile &tl = L.tiles->t[r >> 5];
                int srow = r;                                             // image row whose proxy is copied here
                if (L.src[r] < 0) {
                    const int oc = tl.kind == 0 ? tl.oc[(r & 31) >> 3] : -1;
                    srow = oc >= 0 ? L.first[oc] : -1;
                }
                srows[u] = srow;
                srcs[u] = srow >= 0 ? L.src[srow] : -1;
                if (srcs[u] >= 0) xs[u] = reinterpret_cast<const float4 *>(proxies + (size_t)srcs[u] * 100)[t];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = it0 + u * NTH;
            if (it >= n_rows * 25) continue;
            const int r = it / 25, t = it - r * 25;
            const float4 x = xs[u];
            const int src = srcs[u], srow = srows[u];
            const float am = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y)), __builtin_fmaxf(__builtin_fabsf(x.z), __builtin_fabsf(x.w)));
            if (am > CB_MAX_ABS) bad = true;
            uint32_t hi0, hi1, lo0, lo1;
            cb_split_pair(x.x, x.y, hi0, lo0);
            cb_split_pair(x.z, x.w, hi1, lo1);
            // stacked column-wise tile: rows 16 .. 31 carry the lo pieces of proxies 0 .. 15 in the HI plane (and no norm pieces), so that
            // [ph ; pl] x qh and [ph ; pl] x ql -- 14 MFMAs -- give all four partial products in two row groups of one accumulator
            const AocCorrTile &tl3 = L.tiles->t[r >> 5];
            const bool lo_row = tl3.kind == 1 && tl3.gs == 1 && (r & 31) >= 16;
            if (lo_row) { hi0 = lo0; hi1 = lo1; lo0 = 0u; lo1 = 0u; }
            uint32_t *row = L.img + (size_t)r * CB_ROW_DW;
            if (REC) {
                // record order (dense_split.hip): k-step s holds slots 16 s .. 16 s + 15, lane half hh the slots 16 s + 8 hh .. + 7; channels
                // 4 t .. 4 t + 3 are slots of k-step t / 4, half (t / 2) % 2, dwords 2 (t % 2), + 1 of that chunk; norm pieces in slots 100 .. 102
                uint32_t *d = row + ((t >> 1) & 1) * CB_HALF_DW + (t >> 2) * 4 + (t & 1) * 2;
                d[0] = hi0; d[1] = hi1;
                d[CB_SEG_DW] = lo0; d[CB_SEG_DW + 1] = lo1;
                if (t == 24) {
                    const float a = (src >= 0 && !lo_row) ? -16.0f * L.norm[srow] : 0.0f;
                    const _Float16 n0 = (_Float16)a;
                    const _Float16 n1 = (_Float16)(a - (float)n0);
                    const _Float16 n2 = (_Float16)((a - (float)n0) - (float)n1);
                    uint32_t *d1 = row + CB_HALF_DW;
                    d[2] = pack_f16(n0, n1); d[3] = pack_f16(n2, (_Float16)0.0f);
                    d[CB_SEG_DW + 2] = 0u; d[CB_SEG_DW + 3] = 0u;
#pragma unroll
                    for (int z = 24; z < 28; ++z) { d1[z] = 0u; d1[CB_SEG_DW + z] = 0u; }
                }
            } else if (t < 24) {
                uint32_t *d = row + (t >= 12 ? CB_HALF_DW : 0) + 2 * (t >= 12 ? t - 12 : t);
                d[0] = hi0; d[1] = hi1;
                d[CB_SEG_DW] = lo0; d[CB_SEG_DW + 1] = lo1;
            } else {
                // channels 96, 97 -> pair 24 of half 0; 98, 99 -> pair 24 of half 1; then the norm slots (hi plane, dword 25) and the padding
                const float a = (src >= 0 && !lo_row) ? -16.0f * L.norm[srow] : 0.0f;
                const _Float16 n0 = (_Float16)a;
                const _Float16 n1 = (_Float16)(a - (float)n0);
                const _Float16 n2 = (_Float16)((a - (float)n0) - (float)n1);
                uint32_t *d0 = row, *d1 = row + CB_HALF_DW;
                d0[24] = hi0; d0[25] = pack_f16(n0, n1); d0[26] = 0u; d0[27] = 0u;
                d0[CB_SEG_DW + 24] = lo0; d0[CB_SEG_DW + 25] = 0u; d0[CB_SEG_DW + 26] = 0u; d0[CB_SEG_DW + 27] = 0u;
                d1[24] = hi1; d1[25] = pack_f16(n2, (_Float16)0.0f); d1[26] = 0u; d1[27] = 0u;
                d1[CB_SEG_DW + 24] = lo1; d1[CB_SEG_DW + 25] = 0u; d1[CB_SEG_DW + 26] = 0u; d1[CB_SEG_DW + 27] = 0u;
            }
        }
    }
    // phase 4: epilogue constants per (tile, lane half): which output plane each store slot writes, its bias, whether its set exists.
    // Slot a of half hh: gs 4 -> the set (half 0 of its last tile only); gs 2 -> set hh; gs 1 -> set 2 hh.  Slot b: gs 1 -> set 2 hh + 1.
    for (int it = threadIdx.x; it < NT * 2; it += NTH) {
        const int ti = it >> 1, hh = it & 1;
        const AocCorrTile &tl = L.tiles->t[ti];
        int oca = -1, ocb = -1;
        if (tl.kind == 0) {
            if (tl.gs == 4) oca = (tl.last && hh == 0) ? tl.oc[0] : -1;
            else if (tl.gs == 2) oca = tl.oc[2 * hh];
            else { oca = tl.oc[2 * hh]; ocb = tl.oc[2 * hh + 1]; }
        }
        CbSlotDesc dsc;
        dsc.off_a = oca >= 0 ? (int32_t)L.tiles->oc_offset[oca] : -1;
        dsc.off_b = ocb >= 0 ? (int32_t)L.tiles->oc_offset[ocb] : -1;
        dsc.bias_a = (oca >= 0 && bias) ? bias[L.tiles->oc_bias[oca]] : 0.0f;
        dsc.bias_b = (ocb >= 0 && bias) ? bias[L.tiles->oc_bias[ocb]] : 0.0f;
        dsc.valid_a = oca >= 0 ? (L.first[oca] >= 0) : 0;
        dsc.valid_b = ocb >= 0 ? (L.first[ocb] >= 0) : 0;
        dsc.pad0 = dsc.pad1 = 0;
        L.slot[it] = dsc;
    }
    if (COL0) {
        for (int row = threadIdx.x; row < 32; row += NTH) {
            const AocCorrTile &tl = L.tiles->t[0];
            CbColDesc c;
            const bool on = row < tl.cnt[0];
            const int oc = tl.oc[0] + (on ? row : 0);
            c.off = on ? (int32_t)L.tiles->oc_offset[oc] : -1;
            c.bias = (on && bias) ? bias[L.tiles->oc_bias[oc]] : 0.0f;
            c.valid = on ? (L.first[oc] >= 0) : 0;
            c.pad = 0;
            L.col[row] = c;
        }
    }
    __syncthreads();

}

// One 32-pixel tile against the NT proxy tiles of the image: 21 chained MFMAs per proxy tile, the in-register min (a max over raw
// accumulators), the proto-mask transform and the stores.  `after_last_chain` runs once the last chain has consumed `seq`.
template <int NT, bool COL0, typename Hook>
__device__ __forceinline__ void cb_tile_compute(const CbImage &L, const AocCorrTiles &tiles, const PixelSeq &seq, float q2, float *out_pix,
                                                bool live, float *dump, int transform, int j, int h, Hook after_last_chain) {
    float carry = -INFINITY;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        const uint4 *arow = reinterpret_cast<const uint4 *>(L.img + (size_t)(ti * 32 + j) * CB_ROW_DW + h * CB_HALF_DW);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;           // folds into the first MFMA's constant C operand (no unconditional path around the chain)
        const bool stacked = COL0 && ti == 0 && tiles.t[0].gs == 1;           // rows [ph ; pl]: no separate lo-plane product
#pragma unroll
        for (int s = 0; s < CB_SEG_STEPS; ++s) {              // q.p = qh.ph + qh.pl + ql.ph (the norm slots ride in the hi x hi product)
            const f16x8 ah = __builtin_bit_cast(f16x8, arow[s]);
            const f16x8 bh = __builtin_bit_cast(f16x8, seq.bh[s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            if (!stacked) {
                const f16x8 al = __builtin_bit_cast(f16x8, arow[CB_SEG_STEPS + s]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, seq.bl[s]), acc, 0, 0, 0);
        }
        if (ti == NT - 1) after_last_chain();                     // `seq` is free from here on
        if (COL0 && ti == 0) {
            // column-wise tile: every row is its own output (k = 1 proxies, no min: AEM:127).  Register r of lane half h is
            // row (r / 4) * 8 + 4 h + r % 4; row groups beyond the tile's rows are skipped with uniform branches.
            const int cnt = tiles.t[0].cnt[0];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (8 * g < cnt) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const CbColDesc c = L.col[8 * g + 4 * h + u];
                        // stacked (cnt <= 16, g < 2): row r holds ph.(qh + ql), row 16 + r = register + 8 of the same lane holds pl.(qh + ql)
                        const float raw = (stacked && g < 2) ? acc[4 * g + u] + acc[4 * g + 8 + u] : acc[4 * g + u];
                        float d = c.valid ? q2 + CB_UNSCALE * raw : AOC_PAD_DISTANCE;
                        const float td = cb_transform(d, c.bias);
                        d = transform ? td : d;
                        float *p = (c.off >= 0 && live) ? out_pix + c.off : dump;
                        *p = d;
                    }
                }
            }
            continue;
        }
        // grouped tile: group g = rows 8g .. 8g+7 = registers 4g .. 4g+3 of both lane halves
        const CbSlotDesc sd = L.slot[ti * 2 + h];
        const int gs = tiles.t[ti].gs;
        float gm[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            gm[g] = __builtin_fmaxf(__builtin_fmaxf(acc[4 * g], acc[4 * g + 1]), __builtin_fmaxf(acc[4 * g + 2], acc[4 * g + 3]));
        float va, vb = 0.0f;
        if (gs == 4) {                                                // one set spanning the tile (possibly continued from / into neighbours)
            float v = __builtin_fmaxf(__builtin_fmaxf(gm[0], gm[1]), __builtin_fmaxf(gm[2], gm[3]));
            v = cb_halfmax2(v, v);
            if (!tiles.t[ti].first) v = __builtin_fmaxf(v, carry);
            carry = v;
            va = v;
        } else if (gs == 2) {                                         // two sets: lane half h ends up with set h
            va = cb_halfmax2(__builtin_fmaxf(gm[0], gm[1]), __builtin_fmaxf(gm[2], gm[3]));
        } else {                                                      // four sets: lane half h ends up with sets 2h (slot a), 2h + 1 (slot b)
            va = cb_halfmax2(gm[0], gm[2]);
            vb = cb_halfmax2(gm[1], gm[3]);
        }
        {
            float d = sd.valid_a ? q2 + CB_UNSCALE * va : AOC_PAD_DISTANCE;
            const float td = cb_transform(d, sd.bias_a);
            d = transform ? td : d;
            float *p = (sd.off_a >= 0 && live) ? out_pix + sd.off_a : dump;
            *p = d;
        }
        if (gs == 1) {
            float d = sd.valid_b ? q2 + CB_UNSCALE * vb : AOC_PAD_DISTANCE;
            const float td = cb_transform(d, sd.bias_b);
            d = transform ? td : d;
            float *p = (sd.off_b >= 0 && live) ? out_pix + sd.off_b : dump;
            *p = d;
        }
    }
}

// NT = proxy tiles of the launch; COL0: tile 0 is the column-wise tile (k = 1 proxies).  Block = NW waves, three per SIMD: while one
// wave converts its next pixel tile (VALU) or waits for LDS, its partner's MFMA chain keeps the matrix pipe busy.
template <int NW, int NT, bool COL0>
__global__ __launch_bounds__(NW * 64) void proxy_corr_batched_kernel(AocCorrFrames frames, int64_t m, AocCorrTiles tiles, int transform,
                                                                      int32_t *__restrict__ gate, int dbg, int call_seq) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // the tile tables in LDS: the staging phases index them per thread, and from the kernel-argument segment every such read is a
    // dependent global round trip (hipcc cannot use scalar loads for a per-lane index): ~30 of them per frame before
    AocCorrTiles &ltiles = *reinterpret_cast<AocCorrTiles *>(lds);          // first in the dynamic allocation (8-byte members: 16-byte aligned base)
    for (int i = threadIdx.x; i < (int)(sizeof(AocCorrTiles) / 4); i += NW * 64)
        reinterpret_cast<uint32_t *>(&ltiles)[i] = reinterpret_cast<const uint32_t *>(&tiles)[i];
    __syncthreads();
    constexpr int NTH = NW * 64;
    constexpr int n_rows = NT * 32;
    uint32_t *limg = lds + CB_TABLE_DW;                                       // [n_rows][CB_ROW_DW]
    int32_t *lsrc = reinterpret_cast<int32_t *>(limg + (size_t)NT * CB_TILE_DW);   // [n_rows] proxy row feeding each image row (-1: none)
    float *lnorm = reinterpret_cast<float *>(lsrc + n_rows);                  // [n_rows] |p|^2 of that proxy
    int32_t *lfirst = reinterpret_cast<int32_t *>(lnorm + n_rows);            // [AOC_CORR_MAX_OUT] first valid image row of each output column (-1: absent)
    CbSlotDesc *lslot = reinterpret_cast<CbSlotDesc *>(lfirst + AOC_CORR_MAX_OUT);   // [NT][2]
    CbColDesc *lcol = reinterpret_cast<CbColDesc *>(lslot + NT * 2);          // [32]
    uint32_t *wbuf_all = reinterpret_cast<uint32_t *>(lcol + 32);             // [NW][6 400 B]
    const CbImage img = {&ltiles, limg, lsrc, lnorm, lfirst, lslot, lcol};
    float *dump = reinterpret_cast<float *>(gate) + 16 + (threadIdx.x & 31);  // where masked-off lanes store

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform: keeps the tile bookkeeping and pointer loads scalar
    const int j = lane & 31, h = lane >> 5;
    uint32_t *wbuf = wbuf_all + (size_t)wave * (CB_WBUF_BYTES / 4);
    const uint32_t *wbuf_lane = wbuf + (j & 15) * 100 + h * 48;               // this lane's channels 48 h .. 48 h + 47 of pixel row j % 16
    const uint32_t *wbuf_tail = wbuf + (j & 15) * 100 + 96 + 2 * h;           // ... and 96 + 2 h, 97 + 2 h
    const int64_t T = (m + 31) >> 5;                                          // 32-pixel tiles per frame
    const int64_t total = T * frames.n;
    const int64_t frame_chunks = m * 25;                                      // 16-byte chunks of one frame's query
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

    // this wave's tiles: within frame f the tiles [lo_f, hi_f) of the block's range, taken round-robin by the NW waves
    auto seg_lo = [&](int f) -> int64_t { return f == f_beg ? g0 - (int64_t)f * T : 0; };
    auto seg_hi = [&](int f) -> int64_t { return f == f_end - 1 ? g1 - (int64_t)f * T : T; };
    // the tile after (f, t) in this wave's list (f = f_end: none)
    auto advance = [&](int &f, int64_t &t) {
        t += NW;
        while (f < f_end && t >= seg_hi(f)) {
            ++f;
            if (f < f_end) t = seg_lo(f) + wave;
        }
    };

    // ---- the wave's pipeline: tile i in `seq` (MFMA operand), tile i+1 in flight from HBM: its first 16 pixel rows by LDS-DMA straight
    // into the wave's transposition buffer (free during the MFMA phase), its last 16 into `flatb` (28 registers instead of 52: twelve
    // waves per CU fit the register file)
    u32x4 flatb[CB_FLATB];
    const uint32_t wbuf_lds = (uint32_t)reinterpret_cast<uintptr_t>(wbuf);
    auto issue_flat = [&](int f_, int64_t tile_) {
        // wave-uniform by construction; say so, so that the pointer comes from a scalar load instead of queueing behind the stores
        const int f = __builtin_amdgcn_readfirstlane(f_);
        const int64_t tile = ((int64_t)__builtin_amdgcn_readfirstlane((int)(tile_ >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_);
        const u32x4 *q = reinterpret_cast<const u32x4 *>(frames.f[f].query);
        // 32-bit chunk indices (the host checks m * 25 < 2^31): one add, one min and one 64-bit shift-add per load
        const int c0 = (int)tile * CB_CHUNKS + lane, c_last = (int)frame_chunks - 1;
#pragma unroll
        for (int t = 0; t < CB_FLATB; ++t) {                                  // chunks [0, 400): DMA (issued first: they complete first)
            const int c = min(c0 + t * 64, c_last);                           // last tile of a frame: re-read valid data, never stored
            if (t < CB_FLATB - 1 || lane < 16) cb_glds16(q + c, wbuf_lds + (uint32_t)t * 1024u);
        }
#pragma unroll
        for (int t = 0; t < CB_FLATB; ++t) {                                  // chunks [400, 800): registers
            const int c = min(c0 + CB_HALF_CHUNKS + t * 64, c_last);
            flatb[t] = q[c];
        }
    };
    const uint32_t nconst = h == 0 ? pack_f16((_Float16)CB_QCONST, (_Float16)CB_QCONST) : pack_f16((_Float16)CB_QCONST, (_Float16)0.0f);
    PixelSeq seq;
    // flat -> LDS -> seq, in two halves of 16 pixel rows (chunks [0, 400) and [400, 800)); DS operations of a wave execute in order,
    // so the second half may overwrite the buffer right behind the first half's reads
    auto convert_tile = [&]() {
        float raw[CB_CH];
        seq.bh[6][1] = nconst;
        // memory reads return in order: once the LAST register load of the tile has arrived (hipcc's own wait for flatb[6]), the DMA
        // transfers issued in front of it have landed in the buffer
        asm volatile("" : "+v"(flatb[CB_FLATB - 1]) :: "memory");
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1) {
#pragma unroll
                for (int t = 0; t < CB_FLATB; ++t) {
                    const int c = t * 64 + lane;                              // chunk of the second half held in flatb[t]
                    if (c < CB_HALF_CHUNKS) reinterpret_cast<u32x4 *>(wbuf)[c] = flatb[t];
                }
            }
            if ((j >> 4) == half) {                                           // only the LDS reads are predicated; the arithmetic runs once
#pragma unroll
                for (int c = 0; c < 12; ++c) {
                    const float4 v = *reinterpret_cast<const float4 *>(wbuf_lane + 4 * c);
                    raw[4 * c] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
                }
                const float2 v = *reinterpret_cast<const float2 *>(wbuf_tail);
                raw[48] = v.x; raw[49] = v.y;
            }
        }
        convert_raw(raw, seq);
    };

    bool bad = false;
    int fa, fb;                        // frames of tiles i, i+1 of this wave (f_end: none)
    int64_t ta, tb;
    fa = f_beg; ta = seg_lo(fa) + wave - NW; advance(fa, ta);
    fb = fa; tb = ta; if (fa < f_end) advance(fb, tb);
    if (fa < f_end) issue_flat(fa, ta);                // in flight under the first staging pass
    bool primed = false;

    for (int f = f_beg; f < f_end; ++f) {
        const AocCorrFrame fr = frames.f[f];
        cb_stage_frame<NTH, NT, COL0, false>(img, fr.proxies, fr.sqnorm, fr.bias, bad);

        // ---- this wave's pixel tiles of frame f ----------------------------------------------------------------
        if (!primed && fa < f_end) {
            convert_tile();                                                   // tile i: flat -> seq
            if (fb < f_end) issue_flat(fb, tb);                               // tile i+1 requested
            primed = true;
        }
        while (fa == f) {
            const int64_t pix = ta * 32 + j;
            const bool live = pix < m;
            const float q2 = cb_halfsum(seq.q2part);
            if (!(q2 <= CB_MAX_SQ)) bad = true;                              // also catches NaN / inf
            float *const out_pix = fr.out + pix;
            cb_tile_compute<NT, COL0>(img, tiles, seq, q2, out_pix, live, dump, transform, j, h, [] {});
            // next tile: flat -> seq (its loads were issued one tile ago), then request the one after it
            fa = fb; ta = tb;
            if (fa < f_end) {
                if (dbg != 4) convert_tile();
                advance(fb, tb);
                if (fb < f_end && dbg != 2) issue_flat(fb, tb);
            }
        }
        __syncthreads();                                                      // every wave is done with this frame's image
    }
    if (bad && dbg == 0) atomicExch(gate, call_seq);      // raised for THIS call only: no memset between calls
}

// ------------------------------------------------------------------------------------------
// The same correlation with the query side read as split records in tile-major order (aoc_split_rows_tiled: what the dense kernel of
// dense_split.hip consumes for the same frame).  A B operand is ONE coalesced 1 KiB wave load straight into the MFMA register layout:
// no transposition buffer, no fp32 -> hi | lo conversion, |q|^2 from the records' norm array -- the wave's work per pixel tile is 14 loads,
// the MFMA chains and the epilogue.  The next tile's loads are issued as soon as the last chain has consumed the operand registers and
// fly under the epilogue and the other waves' chains.  Two blocks of 8 waves per CU: one block's staging of a frame's proxy image runs
// under the other's MFMAs.
struct AocCorrRecFrame {
    const uint4 *qrec;          // tile-major split records of the frame's query
    const float *q2;            // |q|^2 per pixel
    const float *proxies, *sqnorm, *bias;
    float *out;
};
struct AocCorrRecFrames {
    AocCorrRecFrame f[AOC_CORR_MAX_FRAMES];
    int32_t n;
};
constexpr int CR_NW = 8;
constexpr int CR_TILE_CHUNKS = 2 * CB_SEG_STEPS * 64;                         // 16-byte chunks of one 32-pixel tile: [plane][k-step][lane]

// `tiles`: the pass's tile table -- the kernel-argument copy (one pass per launch) or its copy in device memory (all passes in one launch)
template <int NW, int NT, bool COL0>
__device__ __forceinline__ void cr_body(uint32_t *lds, const AocCorrRecFrames &frames, int64_t m, const AocCorrTiles &tiles, int transform,
                                        int32_t *__restrict__ gate, int dbg, int call_seq) {
    AocCorrTiles &ltiles = *reinterpret_cast<AocCorrTiles *>(lds);
    for (int i = threadIdx.x; i < (int)(sizeof(AocCorrTiles) / 4); i += NW * 64)
        reinterpret_cast<uint32_t *>(&ltiles)[i] = reinterpret_cast<const uint32_t *>(&tiles)[i];
    __syncthreads();
    constexpr int NTH = NW * 64;
    constexpr int n_rows = NT * 32;
    uint32_t *limg = lds + CB_TABLE_DW;
    int32_t *lsrc = reinterpret_cast<int32_t *>(limg + (size_t)NT * CB_TILE_DW);
    float *lnorm = reinterpret_cast<float *>(lsrc + n_rows);
    int32_t *lfirst = reinterpret_cast<int32_t *>(lnorm + n_rows);
    CbSlotDesc *lslot = reinterpret_cast<CbSlotDesc *>(lfirst + AOC_CORR_MAX_OUT);
    CbColDesc *lcol = reinterpret_cast<CbColDesc *>(lslot + NT * 2);
    const CbImage img = {&ltiles, limg, lsrc, lnorm, lfirst, lslot, lcol};
    float *dump = reinterpret_cast<float *>(gate) + 16 + (threadIdx.x & 31);  // where masked-off lanes store

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t T = (m + 31) >> 5;
    const int64_t total = T * frames.n;
    int vb;                                                                   // XCD-aware virtual block id (see proxy_corr_batched_kernel)
    {
        const int nb = gridDim.x, lin = blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3, q = nb >> 3, r = nb & 7;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int64_t g0 = total * vb / gridDim.x, g1 = total * (vb + 1) / gridDim.x;
    if (g0 >= g1) return;
    const int f_beg = (int)(g0 / T), f_end = (int)((g1 - 1) / T) + 1;
    auto seg_lo = [&](int f) -> int64_t { return f == f_beg ? g0 - (int64_t)f * T : 0; };
    auto seg_hi = [&](int f) -> int64_t { return f == f_end - 1 ? g1 - (int64_t)f * T : T; };
    auto advance = [&](int &f, int64_t &t) {
        t += NW;
        while (f < f_end && t >= seg_hi(f)) {
            ++f;
            if (f < f_end) t = seg_lo(f) + wave;
        }
    };

    PixelSeq seq;
    float q2_next = 0.0f;
    auto issue = [&](int f_, int64_t tile_) {
        const int f = __builtin_amdgcn_readfirstlane(f_);
        const int tile = __builtin_amdgcn_readfirstlane((int)tile_);
        const u32x4 *q = reinterpret_cast<const u32x4 *>(frames.f[f].qrec) + (size_t)tile * CR_TILE_CHUNKS + lane;
#pragma unroll
        for (int s = 0; s < CB_SEG_STEPS; ++s) seq.bh[s] = q[s * 64];
#pragma unroll
        for (int s = 0; s < CB_SEG_STEPS; ++s) seq.bl[s] = q[(CB_SEG_STEPS + s) * 64];
        const int64_t pix = (int64_t)tile * 32 + j;
        q2_next = frames.f[f].q2[pix < m ? pix : m - 1];
    };
    // query-side value of the norm slots 100 .. 102 (k-step 6, lane half 0, halves 4 .. 6) and zero for slot 103 (the record keeps the
    // pixel's own norm pieces there)
    const uint32_t nc2 = pack_f16((_Float16)CB_QCONST, (_Float16)CB_QCONST), nc3 = pack_f16((_Float16)CB_QCONST, (_Float16)0.0f);

    bool bad = false;
    int fa = f_beg;
    int64_t ta = seg_lo(fa) + wave - NW;
    advance(fa, ta);
    if (fa < f_end) issue(fa, ta);                     // in flight under the first staging pass

    for (int f = f_beg; f < f_end; ++f) {
        const AocCorrRecFrame fr = frames.f[f];
        cb_stage_frame<NTH, NT, COL0, true>(img, fr.proxies, fr.sqnorm, fr.bias, bad);
        while (fa == f) {
            // the image is loop-invariant and nothing in this loop writes LDS: without this, hipcc hoists all NT x 14 operand reads out of
            // the loop (280 registers at NT = 5) and spills
            asm volatile("" ::: "memory");
            const int64_t pix = ta * 32 + j;
            const bool live = pix < m;
            const float q2 = q2_next;
            if (!(q2 <= CB_MAX_SQ)) bad = true;                              // also catches NaN / inf; implies |x| 2^10 <= 65000
            if (h == 0) { seq.bh[CB_SEG_STEPS - 1][2] = nc2; seq.bh[CB_SEG_STEPS - 1][3] = nc3; }
            int fn = fa;
            int64_t tn = ta;
            advance(fn, tn);
            cb_tile_compute<NT, COL0>(img, tiles, seq, q2, fr.out + pix, live, dump, transform, j, h, [&] {
                if (fn < f_end && dbg != 2) issue(fn, tn);
            });
            fa = fn; ta = tn;
        }
        __syncthreads();                                                      // every wave is done with this frame's image
    }
    if (bad && dbg == 0) atomicExch(gate, call_seq);
}

template <int NW, int NT, bool COL0>
__global__ __launch_bounds__(NW * 64, 4) void proxy_corr_records_kernel(AocCorrRecFrames frames, int64_t m, AocCorrTiles tiles, int transform,
                                                                         int32_t *__restrict__ gate, int dbg, int call_seq) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    cr_body<NW, NT, COL0>(lds, frames, m, tiles, transform, gate, dbg, call_seq);
}

// Round 5: ALL passes of a frame (a pass = up to AOC_CORR_MAX_TILES proxy tiles, what one workgroup's LDS image holds) in ONE launch:
// blockIdx.y = pass, its tile table read from device memory (`passes`, written once per set structure by cb_table_write_kernel).  A pass of
// its own launch costs ~30-40 us at cfg3 / cfg4 sizes although its MFMAs take ~2 us -- launch, staging of the proxy image (dependent loads
// and LDS phases) and one stream over the query records; the passes are independent, so side by side they take the time of ONE (cfg4: eight
// launches of 38 us -> one).  Every workgroup runs the specialisation of its pass (uniform switch): same code, same results as pass by pass.
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void proxy_corr_records_multi_kernel(AocCorrRecFrames frames, int64_t m, const AocCorrTiles *__restrict__ passes,
                                                                               int transform, int32_t *__restrict__ gate, int dbg, int call_seq) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const AocCorrTiles &tiles = passes[blockIdx.y];
    const int key = tiles.n * 2 + (tiles.t[0].kind == 1 ? 1 : 0);
    switch (key) {
        case 2: cr_body<NW, 1, false>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 3: cr_body<NW, 1, true>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 4: cr_body<NW, 2, false>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 5: cr_body<NW, 2, true>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 6: cr_body<NW, 3, false>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 7: cr_body<NW, 3, true>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 8: cr_body<NW, 4, false>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 9: cr_body<NW, 4, true>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 10: cr_body<NW, 5, false>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        case 11: cr_body<NW, 5, true>(lds, frames, m, tiles, transform, gate, dbg, call_seq); break;
        default: break;
    }
}
// one pass's tile table from the kernel arguments into the workspace (once per set structure: a sequence keeps it across its frames)
__global__ __launch_bounds__(256) void cb_table_write_kernel(AocCorrTiles t, AocCorrTiles *__restrict__ dst) {
    for (int i = threadIdx.x; i < (int)(sizeof(AocCorrTiles) / 4); i += 256)
        reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(&t)[i];
}

inline int cb_n_cus() {
    const int set = aoc_stream_cus();                                         // CUs the launching stream may use (aoc_set_stream_cus: HIP CU mask)
    if (set > 0) return set;
    static int n = 0;
    if (n == 0) {
        {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
            else n = 256;
        }
    }
    return n;
}

constexpr size_t CB_WS_BYTES = 256;
constexpr int CB_MAX_PASSES = 16;                                              // pass tables a workspace keeps for the one-launch form
static_assert(sizeof(AocCorrTiles) % 8 == 0, "the pass tables are an array in the workspace");
constexpr size_t CB_WS_CACHED_BYTES = (CB_WS_BYTES + (size_t)CB_MAX_PASSES * sizeof(AocCorrTiles) + 255) / 256 * 256;
inline AocCorrTiles *cb_pass_table(void *workspace, int i) {                  // an ARRAY: the kernel indexes it with the pass
    return reinterpret_cast<AocCorrTiles *>(static_cast<char *>(workspace) + CB_WS_BYTES) + i;
}

// rec_host != NULL: the frames' queries as tile-major split records (proxy_corr_records_kernel); frames_host always carries the fp32 rows
// (the exact-fp32 take-over reads them)
// tables_key != NULL (records form, one chunk of frames, workspace of CB_WS_CACHED_BYTES): ALL passes in one launch, their tile tables kept in
// the workspace and rewritten only when *tables_key (caller-owned host word; 0 = nothing cached) does not match this call's tables
int cb_run(const aoc_corr_frame *frames_host, const AocCorrRecFrame *rec_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
           const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
           int transform, int precision, void *workspace, size_t workspace_bytes, aoc_stream_t stream, int64_t *tables_key = nullptr) {
    if (!frames_host || !set_begin_host || !set_size_host || !set_out_offset_host) return AOC_ERR_INVALID_ARG;
    if (n_frames < 1 || m < 1 || n_set < 1 || n_proxy < 0 || C < 4) return AOC_ERR_INVALID_ARG;
    if (m > (int64_t)80000000) return AOC_ERR_UNSUPPORTED;                    // 32-bit chunk indices inside the kernels (m * 25 < 2^31)
    if ((C & 3) || C > AOC_MAX_CHANNELS) return AOC_ERR_UNSUPPORTED;
    if (precision != AOC_CORR_SPLIT && precision != AOC_CORR_FP32) return AOC_ERR_INVALID_ARG;
    for (int s = 0; s < n_set; ++s)
        if (set_size_host[s] < 0 || set_begin_host[s] < 0 || set_begin_host[s] + set_size_host[s] > n_proxy) return AOC_ERR_INVALID_ARG;
    for (int f = 0; f < n_frames; ++f)
        if (!frames_host[f].query || !frames_host[f].proxies || !frames_host[f].out) return AOC_ERR_INVALID_ARG;
    hipStream_t st = aoc_hip_stream(stream);

    bool split_ok = precision == AOC_CORR_SPLIT && C == 100 && workspace && workspace_bytes >= CB_WS_BYTES;
    for (int f = 0; f < n_frames && split_ok; ++f)
        if ((reinterpret_cast<uintptr_t>(frames_host[f].query) | reinterpret_cast<uintptr_t>(frames_host[f].proxies)) & 15) split_ok = false;
    if (!split_ok) {
        if (precision == AOC_CORR_SPLIT && (!workspace || workspace_bytes < CB_WS_BYTES)) return AOC_ERR_WORKSPACE;
        return aoc_corr_fp32_batched(frames_host, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, 1, transform,
                                     nullptr, stream);
    }

    // take-over flag: the kernel stores this call's sequence number in it when a precondition fails and the gated fp32 kernel runs
    // iff it finds that number there; a workspace is used by one stream at a time, so no reset is needed between calls
    int32_t *gate = static_cast<int32_t *>(workspace);
    static std::atomic<int32_t> g_seq{0};
    int32_t call_seq = g_seq.fetch_add(1) + 1;
    if (call_seq <= 0) { g_seq.store(1); call_seq = 1; }

    // ---- pack the sets into 32-row tiles: single-proxy sets -> column-wise tiles, the others by row-group class
    const size_t tile_bytes = (size_t)CB_TILE_DW * 4 + 32 * 8;
    constexpr int NW = CB_NW;
    // records kernel: no transposition buffers, two blocks per CU (80 KB each)
    const size_t lds_fixed = (size_t)CB_TABLE_DW * 4 + AOC_CORR_MAX_OUT * 4 + (size_t)AOC_CORR_MAX_TILES * 2 * 32 + 32 * 16 + (rec_host ? 0 : (size_t)NW * CB_WBUF_BYTES);
    int max_tiles = (int)(((size_t)(rec_host ? 80 : 160) * 1024 - lds_fixed) / tile_bytes);
    if (max_tiles > AOC_CORR_MAX_TILES) max_tiles = AOC_CORR_MAX_TILES;
    const int n_cu = cb_n_cus();
    const int64_t T = (m + 31) / 32;
    // a set whose proxies do not fit the LDS image of one launch (more than max_tiles * 32 = 160 of them at C = 100, e.g. the union of the
    // per-frame code books of hotpath.IncrementalProxyBank beyond 10 pool frames): the whole call takes the exact-fp32 kernel, which
    // handles any set size -- decided before anything is enqueued
    for (int s = 0; s < n_set; ++s)
        if ((set_size_host[s] + 31) / 32 > max_tiles)
            return aoc_corr_fp32_batched(frames_host, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, 1, transform,
                                         nullptr, stream);

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
        const bool multi = rec_host && tables_key && n_frames <= AOC_CORR_MAX_FRAMES && workspace_bytes >= CB_WS_CACHED_BYTES;
        AocCorrTiles *passes_host = multi ? static_cast<AocCorrTiles *>(malloc(sizeof(AocCorrTiles) * CB_MAX_PASSES)) : nullptr;
        int n_pass = 0;
        bool multi_ok = passes_host != nullptr;
        struct FreeGuard { void *p; ~FreeGuard() { free(p); } } free_guard{passes_host};
        std::function<int()> launch_multi = [&]() -> int {
            if (!multi_ok || n_pass == 0) return AOC_OK;
            // the tables are a function of the set structure alone: a sequence writes them once (key = a hash of their bytes)
            uint64_t key = 0xcbf29ce484222325ull ^ (uint64_t)n_pass;
            for (int i = 0; i < n_pass; ++i) {
                // every byte is initialised: tiles are value-initialised, the per-output arrays and the unused tiles are zeroed below
                AocCorrTiles &t = passes_host[i];
                for (int k = t.n; k < AOC_CORR_TABLE_TILES; ++k) t.t[k] = AocCorrTile{};
                for (int k = t.n_out; k < AOC_CORR_MAX_OUT; ++k) { t.oc_offset[k] = 0; t.oc_bias[k] = 0; t.oc_row0[k] = 0; t.oc_rows[k] = 0; }
                const uint64_t *w = reinterpret_cast<const uint64_t *>(&t);
                for (size_t k = 0; k < sizeof(AocCorrTiles) / 8; ++k) key = (key ^ w[k]) * 0x100000001b3ull + (key >> 29);
            }
            key |= 1ull;
            if ((uint64_t)*tables_key != key) {
                for (int i = 0; i < n_pass; ++i)
                    hipLaunchKernelGGL(cb_table_write_kernel, dim3(1), dim3(256), 0, st, passes_host[i], cb_pass_table(workspace, i));
                *tables_key = (int64_t)key;
            }
            int nt_max = 0;
            for (int i = 0; i < n_pass; ++i) nt_max = passes_host[i].n > nt_max ? passes_host[i].n : nt_max;
            const size_t lds = (size_t)nt_max * tile_bytes + lds_fixed;
            int64_t bpp = (2 * (int64_t)n_cu) / n_pass;                       // the passes' workgroups are resident together: two per CU in all
            if (bpp < 1) bpp = 1;
            if (bpp > T * fr.n) bpp = T * fr.n;
            AocCorrRecFrames rf;
            rf.n = fr.n;
            for (int f = 0; f < fr.n; ++f) rf.f[f] = rec_host[f0 + f];
            static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(proxy_corr_records_multi_kernel<CR_NW>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
            if (!lds_ok) return AOC_ERR_LAUNCH;
            static const int dbg_m = AOC_DEV_ENV_INT("AOC_CORR_DEBUG", 0);
            hipLaunchKernelGGL((proxy_corr_records_multi_kernel<CR_NW>), dim3((unsigned)bpp, (unsigned)n_pass), dim3(CR_NW * 64), lds, st, rf, m,
                               cb_pass_table(workspace, 0), transform, gate, dbg_m, call_seq);
            if (hipGetLastError() != hipSuccess) return AOC_ERR_LAUNCH;
            n_pass = 0;
            return AOC_OK;
        };
        auto flush = [&]() -> int {
            if (tab.n == 0) return AOC_OK;
            if (multi_ok) {
                // one-launch form: the pass is only recorded here; everything is launched after the packing loop (or when the table is full:
                // more than CB_MAX_PASSES passes go out as several launches, whose tables then replace each other in the workspace)
                if (n_pass == CB_MAX_PASSES) { const int rcm = launch_multi(); if (rcm) return rcm; }
                memcpy(static_cast<void *>(&passes_host[n_pass]), &tab, sizeof(tab));
                ++n_pass;
                reset();
                return AOC_OK;
            }
            const size_t lds = (size_t)tab.n * tile_bytes + lds_fixed;
            int64_t grid = T * fr.n;
            if (rec_host) {
                AocCorrRecFrames rf;
                rf.n = fr.n;
                for (int f = 0; f < fr.n; ++f) rf.f[f] = rec_host[f0 + f];
                static const int dbg_r = AOC_DEV_ENV_INT("AOC_CORR_DEBUG", 0);
                const bool col0_r = tab.t[0].kind == 1;
                if (grid > 2 * (int64_t)n_cu) grid = 2 * (int64_t)n_cu;
#define AOC_CR(N, COL)                                                                                                                  \
    do {                                                                                                                                \
        static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(proxy_corr_records_kernel<CR_NW, N, COL>),         \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;             \
        if (!lds_ok) return AOC_ERR_LAUNCH;                                                                                              \
        hipLaunchKernelGGL((proxy_corr_records_kernel<CR_NW, N, COL>), dim3((unsigned)grid), dim3(CR_NW * 64), lds, st, rf, m, tab, transform, gate, dbg_r, call_seq); \
    } while (0)
                switch (tab.n * 2 + (col0_r ? 1 : 0)) {
                    case 2: AOC_CR(1, false); break;
                    case 3: AOC_CR(1, true); break;
                    case 4: AOC_CR(2, false); break;
                    case 5: AOC_CR(2, true); break;
                    case 6: AOC_CR(3, false); break;
                    case 7: AOC_CR(3, true); break;
                    case 8: AOC_CR(4, false); break;
                    case 9: AOC_CR(4, true); break;
                    case 10: AOC_CR(5, false); break;
                    case 11: AOC_CR(5, true); break;
                    default: return AOC_ERR_UNSUPPORTED;
                }
#undef AOC_CR
                reset();
                return hipGetLastError() == hipSuccess ? AOC_OK : AOC_ERR_LAUNCH;
            }
            if (grid > n_cu) grid = n_cu;
            static const int dbg = AOC_DEV_ENV_INT("AOC_CORR_DEBUG", 0);   // developer switch: 2 = no tile loads, 4 = no conversion (timing experiments; wrong results)
            const bool col0 = tab.t[0].kind == 1;
#define AOC_CB(N, COL)                                                                                                                  \
    do {                                                                                                                                \
        static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(proxy_corr_batched_kernel<NW, N, COL>),            \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;            \
        if (!lds_ok) return AOC_ERR_LAUNCH;                                                                                              \
        hipLaunchKernelGGL((proxy_corr_batched_kernel<NW, N, COL>), dim3((unsigned)grid), dim3(NW * 64), lds, st, fr, m, tab, transform, gate, dbg, call_seq); \
    } while (0)
            switch (tab.n * 2 + (col0 ? 1 : 0)) {
                case 2: AOC_CB(1, false); break;
                case 3: AOC_CB(1, true); break;
                case 4: AOC_CB(2, false); break;
                case 5: AOC_CB(2, true); break;
                case 6: AOC_CB(3, false); break;
                case 7: AOC_CB(3, true); break;
                case 8: AOC_CB(4, false); break;
                case 9: AOC_CB(4, true); break;
                case 10: AOC_CB(5, false); break;
                case 11: AOC_CB(5, true); break;
                default: return AOC_ERR_UNSUPPORTED;
            }
#undef AOC_CB
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
                    // runs of single-proxy sets over consecutive proxies whose output planes are a constant step apart share a column-wise tile
                    bool cont = open_tile >= 0 && tab.t[open_tile].cnt[0] < 32 && set_begin_host[s] == tab.t[open_tile].begin[0] + tab.t[open_tile].cnt[0];
                    if (cont) {
                        AocCorrTile &tl = tab.t[open_tile];
                        const int64_t step = set_out_offset_host[s] - tab.oc_offset[tl.oc[0] + tl.cnt[0] - 1];
                        if (tl.cnt[0] == 1) tl.step = step;
                        else if (tl.step != step) cont = false;
                    }
                    if (!cont || tab.n_out + 1 > AOC_CORR_MAX_OUT) {
                        // the kernel takes at most one column-wise tile per launch, as tile 0
                        if (tab.n > 0) { int rc = flush(); if (rc) return rc; }
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
                    tl.gs = tl.cnt[0] <= 16 ? 1 : 0;                          // <= 16 proxies: rows 16 .. 31 take their lo pieces (stacked tile, 14 MFMAs)
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
        { const int rcm = launch_multi(); if (rcm) return rcm; }
    }
    // exact-fp32 kernel: runs only when a precondition of the split arithmetic failed somewhere in the launch
    return aoc_corr_fp32_batched(frames_host, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, 1, transform,
                                 gate, stream, 0, call_seq);
}

}  // namespace

extern "C" {

size_t aoc_proxy_corr_min_batched_workspace_bytes(void) { return CB_WS_BYTES; }

int aoc_proxy_corr_min_batched(const aoc_corr_frame *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                               const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                               int transform, int precision, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return cb_run(frames_host, nullptr, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, transform, precision,
                  workspace, workspace_bytes, stream);
}

int aoc_proxy_corr_min_records(const aoc_corr_frame_rec *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                               const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                               int transform, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_proxy_corr_min_records_cached(frames_host, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, transform,
                                             workspace, workspace_bytes, nullptr, stream);
}

size_t aoc_proxy_corr_min_records_cached_workspace_bytes(void) { return CB_WS_CACHED_BYTES; }

int aoc_proxy_corr_min_records_cached(const aoc_corr_frame_rec *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                                      const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                                      int transform, void *workspace, size_t workspace_bytes, int64_t *tables_key, aoc_stream_t stream) {
    if (!frames_host || n_frames < 1 || n_frames > 4096) return AOC_ERR_INVALID_ARG;
    if (C != 100) return AOC_ERR_UNSUPPORTED;                                 // the records of dense_split.hip at the kernel's channel count
    if (!workspace || workspace_bytes < (tables_key ? CB_WS_CACHED_BYTES : CB_WS_BYTES)) return AOC_ERR_WORKSPACE;
    aoc_corr_frame *plain = static_cast<aoc_corr_frame *>(malloc(sizeof(aoc_corr_frame) * (size_t)n_frames));
    AocCorrRecFrame *rec = static_cast<AocCorrRecFrame *>(malloc(sizeof(AocCorrRecFrame) * (size_t)n_frames));
    int rc = (plain && rec) ? AOC_OK : AOC_ERR_LAUNCH;
    for (int f = 0; f < n_frames && rc == AOC_OK; ++f) {
        const aoc_corr_frame_rec &s = frames_host[f];
        if (!s.query || !s.query_rec || !s.query_sqnorm || (reinterpret_cast<uintptr_t>(s.query_rec) & 15)) { rc = AOC_ERR_INVALID_ARG; break; }
        plain[f].query = s.query; plain[f].proxies = s.proxies; plain[f].proxy_sqnorm = s.proxy_sqnorm; plain[f].set_bias = s.set_bias; plain[f].out = s.out;
        rec[f].qrec = static_cast<const uint4 *>(s.query_rec); rec[f].q2 = s.query_sqnorm;
        rec[f].proxies = s.proxies; rec[f].sqnorm = s.proxy_sqnorm; rec[f].bias = s.set_bias; rec[f].out = s.out;
    }
    if (rc == AOC_OK)
        rc = cb_run(plain, rec, n_frames, m, C, n_proxy, n_set, set_begin_host, set_size_host, set_out_offset_host, transform, AOC_CORR_SPLIT,
                    workspace, workspace_bytes, stream, tables_key);
    free(plain);
    free(rec);
    return rc;
}

}  // extern "C"
