// Dense pixel-level matching (AEM:61-89, 178-227) on the fp16 matrix pipe with fp32-equivalent products.
//
// Every fp32 value x (scaled by 2^10) is split into hi = fp16(x') and lo = fp16(x' - hi): hi + lo represents x'
// to 2^-22 relative (two 11-bit significands; typically 2^-23), and q.r = qh.rh + qh.rl + ql.rh (+ a ql.rl term < 2^-22 |q||r| that is
// dropped) accumulates in
// fp32 inside v_mfma_f32_32x32x16_f16 -- three matrix instructions at 16x the fp32 MFMA rate.  The reference
// pixel's -|r|^2/2 rides along in three spare k-slots (K = 100 pads to 112 anyway), so one accumulator holds
// 2^20 * (q.r - |r|^2/2) and the min over reference pixels becomes a max over raw accumulators: the epilogue
// is one v_max3 per two outputs.  |q|^2 and the 5e4 wrong-label padding (AEM:84-88) are applied per query
// pixel at the end: with one-hot labels min_j(d_j + 5e4 wrong[j,o]) = min(own_o, 5e4 + min_{o' != o} own_o').
//
// This kernel only runs when (a) every scaled value fits fp16 and (b) every kept reference pixel is right for
// exactly one object; both facts are device flags, and the exact-fp32 kernels of correlation.hip take over on
// the same stream otherwise (each side checks the flag itself: no host round trip).
#include "aoc_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SP_KS = 7;                       // k-steps of 16 halves: 112 slots = 100 channels + 3 norm slots + pad
constexpr int SP_K = SP_KS * 16;
constexpr int SP_REC = SP_KS * 4;              // 16-byte chunks per record: per k-step [hi k0-7][hi k8-15][lo k0-7][lo k8-15]
constexpr int SP_LDS_ROW = SP_REC + 1;         // +16 B: rows land on distinct bank quads for ds_read_b128
constexpr int SP_NORM_SLOT = 100;              // slots 100..102 of the hi plane: the three fp16 pieces of -16 |r|^2
constexpr float SP_SCALE = 1024.0f;            // 2^10
constexpr float SP_QCONST = 32768.0f;          // query-side value of the norm slots: 2^15 * (-16 |r|^2) = -2^19 |r|^2
constexpr float SP_UNSCALE = -1.0f / 524288.0f;   // d - |q|^2 = -2^-19 * acc
constexpr int SP_TILE = 32;                    // reference pixels per MFMA tile
constexpr int SP_NB = 2;                       // tiles per staged chunk
constexpr int SP_NW = 8;                       // waves per block
constexpr int SP_NQ = 2;                       // 32-pixel query tiles per wave (stationary B operands in registers)
constexpr int SP_ROWS_PER_BLOCK = SP_NW * SP_NQ * 32;

static_assert(SP_NORM_SLOT + 3 <= SP_K && SP_NORM_SLOT / 16 == SP_KS - 1 && (SP_NORM_SLOT % 16) + 3 <= 8, "norm slots live in the low half of the last k-step");

// ------------------------------------------------------------------------------------------
// fp32 rows -> split records (+ |x|^2).  One thread per (row, k-step).
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, int64_t n, int C, uint4 *__restrict__ rec,
                                                          float *__restrict__ sqnorm, int32_t *__restrict__ overflow) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = idx / SP_KS;
    const int ks = (int)(idx - row * SP_KS);
    if (row >= n) return;
    const float *xr = x + (size_t)row * C;
    _Float16 hi[16], lo[16];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = ks * 16 + e;
        const float v = (k < C) ? xr[k] * SP_SCALE : 0.0f;
        bad |= !(fabsf(v) <= 65000.0f);
        hi[e] = (_Float16)v;
        lo[e] = (_Float16)(v - (float)hi[e]);
    }
    if (ks == SP_KS - 1) {
        float s = 0.0f;
        for (int t = 0; t < C; ++t) s += xr[t] * xr[t];
        if (sqnorm) sqnorm[row] = s;
        bad |= !(s <= 4000.0f);
        const float p = -16.0f * s;
        const _Float16 p1 = (_Float16)p;
        const _Float16 p2 = (_Float16)(p - (float)p1);
        const _Float16 p3 = (_Float16)((p - (float)p1) - (float)p2);
        hi[SP_NORM_SLOT % 16] = p1;
        hi[SP_NORM_SLOT % 16 + 1] = p2;
        hi[SP_NORM_SLOT % 16 + 2] = p3;
    }
    if (bad) atomicOr(overflow, 1);
    union { _Float16 h[32]; uint4 q[4]; } u;
#pragma unroll
    for (int e = 0; e < 16; ++e) { u.h[e] = hi[e]; u.h[16 + e] = lo[e]; }
    uint4 *dst = rec + (size_t)row * SP_REC + ks * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = u.q[c];
}

// ------------------------------------------------------------------------------------------
// Plan: the per-object row lists of label prep cut into 32-row tiles (object-pure, -1 padded), the tile count, and
// the one-hot check.  gate[0] |= overflow | (some kept row is not right for exactly one object).
__global__ __launch_bounds__(256) void split_plan_kernel(const int32_t *__restrict__ obj_rows, const int32_t *__restrict__ counts,
                                                          const int32_t *__restrict__ obj_offsets, int n_obj, int64_t n,
                                                          const uint32_t *__restrict__ right_bits, const uint32_t *__restrict__ wrong_bits,
                                                          const int32_t *__restrict__ overflow, int64_t tile_capacity,
                                                          int32_t *__restrict__ tile_rows, int32_t *__restrict__ tile_obj,
                                                          int32_t *__restrict__ n_tiles, int32_t *__restrict__ gate) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) {
        const uint32_t mask = (n_obj >= 32) ? 0xffffffffu : ((1u << n_obj) - 1u);
        const uint32_t right = right_bits[e];
        if (right & AOC_ROW_KEPT_BIT) {
            const uint32_t r = right & mask, nw = ~wrong_bits[e] & mask;
            if (__popc(r) != 1 || nw != r) atomicOr(gate, 1);
        }
    }
    if (e == 0 && overflow && *overflow) atomicOr(gate, 1);
    const int64_t t = e / SP_TILE;
    const int i = (int)(e - t * SP_TILE);
    if (t >= tile_capacity) return;
    int64_t base = 0;
    int obj = -1, local = 0;
    for (int o = 0; o < n_obj; ++o) {
        const int64_t nt = (counts[o] + SP_TILE - 1) / SP_TILE;
        if (obj < 0 && t < base + nt) { obj = o; local = (int)(t - base); }
        base += nt;
    }
    if (e == 0) *n_tiles = (int32_t)base;
    int32_t id = -1;
    if (obj >= 0) {
        const int pos = local * SP_TILE + i;
        if (pos < counts[obj]) id = obj_rows[obj_offsets[obj] + pos];
    }
    tile_rows[e] = id;
    if (i == 0) tile_obj[t] = obj;
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float max16(const f32x16 &a) {
    float m0 = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]);
    float m1 = __builtin_fmaxf(__builtin_fmaxf(a[3], a[4]), a[5]);
    float m2 = __builtin_fmaxf(__builtin_fmaxf(a[6], a[7]), a[8]);
    float m3 = __builtin_fmaxf(__builtin_fmaxf(a[9], a[10]), a[11]);
    float m4 = __builtin_fmaxf(__builtin_fmaxf(a[12], a[13]), a[14]);
    m0 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
    m3 = __builtin_fmaxf(__builtin_fmaxf(m3, m4), a[15]);
    return __builtin_fmaxf(m0, m3);
}

// Block = 8 waves x 2 query tiles (512 query pixels, B operands, resident in registers); the object-sorted
// reference tiles stream through a double-buffered LDS chunk (A operands).  Grid = (query blocks, tile splits).
// partial[split][pixel][object] = max over the split's tiles of that object of 2^20 (q.r - |r|^2/2); a split
// that holds no tile of an object writes nothing there (the finalize kernel knows the tile ranges).
__global__ __launch_bounds__(SP_NW * 64, 1) void dense_split_kernel(const uint4 *__restrict__ qrec, int64_t m, const uint4 *__restrict__ prec,
                                                                     const int32_t *__restrict__ tile_rows, const int32_t *__restrict__ tile_obj,
                                                                     const int32_t *__restrict__ n_tiles_ptr, const int32_t *__restrict__ gate,
                                                                     int n_obj, float *__restrict__ partial) {
    if (*gate) return;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    constexpr int NT = SP_NW * 64;
    constexpr int ROWS = SP_NB * SP_TILE;
    constexpr int CHUNKS = ROWS * SP_REC;                       // 16-byte pieces per staged chunk
    constexpr int ITERS = (CHUNKS + NT - 1) / NT;
    int32_t *lobj = reinterpret_cast<int32_t *>(lds4 + 2 * ROWS * SP_LDS_ROW);   // [2][SP_NB]

    // XCD-aware block -> (query block, tile split) map: workgroups are dealt round-robin to the 8 XCDs, each with its own
    // L2; give every XCD a contiguous range of the (split-major) work list so that the ~gridDim.x blocks that stream the
    // same reference tiles share one L2 instead of pulling them through all eight
    int bx, by;
    {
        const int nb = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3, q = nb >> 3, r = nb & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        by = v / gridDim.x;
        bx = v - by * gridDim.x;
    }
    const int n_tiles = *n_tiles_ptr;
    const int tps = (n_tiles + gridDim.y - 1) / gridDim.y;
    const int tile_beg = by * tps;
    const int tile_end = min(n_tiles, tile_beg + tps);
    if (tile_beg >= tile_end) return;

    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int64_t wave_row0 = (int64_t)bx * SP_ROWS_PER_BLOCK + (int64_t)wave * (SP_NQ * 32);

    // ---- stationary query operands
    f16x8 bh[SP_NQ][SP_KS], bl[SP_NQ][SP_KS];
#pragma unroll
    for (int iq = 0; iq < SP_NQ; ++iq) {
        const int64_t row = wave_row0 + iq * 32 + col;
        const bool valid = row < m;
        const uint4 *r = qrec + (size_t)(valid ? row : 0) * SP_REC;
#pragma unroll
        for (int ks = 0; ks < SP_KS; ++ks) {
            uint4 u = r[ks * 4 + h], v = r[ks * 4 + 2 + h];
            if (!valid) { u = make_uint4(0, 0, 0, 0); v = make_uint4(0, 0, 0, 0); }
            bh[iq][ks] = __builtin_bit_cast(f16x8, u);
            bl[iq][ks] = __builtin_bit_cast(f16x8, v);
        }
        // norm slots: the query side holds the constant 2^15 (its own norm pieces sit in the record for when the
        // frame later joins the pool); everything else past the channels is zero on both planes
        if (h == 0) {
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16] = (_Float16)SP_QCONST;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 1] = (_Float16)SP_QCONST;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 2] = (_Float16)SP_QCONST;
        }
    }

    // ---- staging pipeline (registers -> the other LDS buffer)
    uint4 sv[ITERS];
    int32_t ids[ITERS];
    int32_t obj_next = -1, obj_commit = -1;
    // padding rows: all-zero channels and the most negative norm pieces, so they never win the max
    const _Float16 NEG = (_Float16)(-65504.0f);
    union { _Float16 hh[8]; uint4 q; } padu;
#pragma unroll
    for (int e = 0; e < 8; ++e) padu.hh[e] = (_Float16)0.0f;
    padu.hh[SP_NORM_SLOT % 16] = NEG; padu.hh[SP_NORM_SLOT % 16 + 1] = NEG; padu.hh[SP_NORM_SLOT % 16 + 2] = NEG;
    const uint4 pad_chunk = padu.q;

    auto load_ids = [&](int t0) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = it * NT + threadIdx.x;
            const int rr = idx / SP_REC;
            ids[it] = (idx < CHUNKS && t0 + rr / SP_TILE < tile_end) ? tile_rows[(size_t)t0 * SP_TILE + rr] : -1;
        }
        obj_next = (threadIdx.x < SP_NB && t0 + (int)threadIdx.x < tile_end) ? tile_obj[t0 + threadIdx.x] : -1;
    };
    auto issue_rows = [&]() {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = it * NT + threadIdx.x;
            const int rr = idx / SP_REC, c = idx - rr * SP_REC;
            const int id = ids[it];
            if (id >= 0) sv[it] = prec[(size_t)id * SP_REC + c];
            else sv[it] = (c == (SP_KS - 1) * 4) ? pad_chunk : make_uint4(0, 0, 0, 0);
        }
        obj_commit = obj_next;
    };
    auto commit_rows = [&](int buf) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = it * NT + threadIdx.x;
            if (idx < CHUNKS) {
                const int rr = idx / SP_REC, c = idx - rr * SP_REC;
                lds4[(size_t)(buf * ROWS + rr) * SP_LDS_ROW + c] = sv[it];
            }
        }
        if (threadIdx.x < SP_NB) lobj[buf * SP_NB + threadIdx.x] = obj_commit;
    };

    float best[SP_NQ];
#pragma unroll
    for (int iq = 0; iq < SP_NQ; ++iq) best[iq] = -INFINITY;
    int cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int iq = 0; iq < SP_NQ; ++iq) {
            const float v = __builtin_fmaxf(best[iq], __shfl_xor(best[iq], 32));
            const int64_t row = wave_row0 + iq * 32 + col;
            if (h == 0 && row < m) partial[((size_t)by * m + row) * n_obj + cur] = v;
            best[iq] = -INFINITY;
        }
    };

    load_ids(tile_beg);
    issue_rows();
    load_ids(tile_beg + SP_NB);
    commit_rows(0);
    __syncthreads();

    int p = 0;
    for (int t0 = tile_beg; t0 < tile_end; t0 += SP_NB) {
        const bool more = t0 + SP_NB < tile_end;
        if (more) {
            issue_rows();                      // chunk t0 + NB: in flight under this chunk's MFMAs
            load_ids(t0 + 2 * SP_NB);
        }
#pragma unroll
        for (int ti = 0; ti < SP_NB; ++ti) {
            if (t0 + ti < tile_end) {
                const int o = lobj[p * SP_NB + ti];
                if (o != cur) { flush(); cur = o; }
                const uint4 *arow = lds4 + (size_t)(p * ROWS + ti * SP_TILE + col) * SP_LDS_ROW;
                f32x16 acc[SP_NQ];
#pragma unroll
                for (int iq = 0; iq < SP_NQ; ++iq)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[iq][r] = 0.0f;
                // A fragments one k-step ahead of the MFMAs that consume them (norm slots first: partial sums stay small)
                f16x8 ah = __builtin_bit_cast(f16x8, arow[(SP_KS - 1) * 4 + h]);
                f16x8 al = __builtin_bit_cast(f16x8, arow[(SP_KS - 1) * 4 + 2 + h]);
#pragma unroll
                for (int kk = 0; kk < SP_KS; ++kk) {
                    const int ks = (kk == 0) ? SP_KS - 1 : kk - 1;
                    f16x8 nh = ah, nl = al;
                    if (kk + 1 < SP_KS) {
                        nh = __builtin_bit_cast(f16x8, arow[kk * 4 + h]);          // k-step kk (the next one in this order)
                        nl = __builtin_bit_cast(f16x8, arow[kk * 4 + 2 + h]);
                    }
                    __builtin_amdgcn_sched_barrier(0);                             // keep the reads ahead of this k-step's MFMAs
#pragma unroll
                    for (int iq = 0; iq < SP_NQ; ++iq) acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[iq][ks], acc[iq], 0, 0, 0);
#pragma unroll
                    for (int iq = 0; iq < SP_NQ; ++iq) acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[iq][ks], acc[iq], 0, 0, 0);
#pragma unroll
                    for (int iq = 0; iq < SP_NQ; ++iq) acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[iq][ks], acc[iq], 0, 0, 0);
                    ah = nh; al = nl;
                }
#pragma unroll
                for (int iq = 0; iq < SP_NQ; ++iq) best[iq] = __builtin_fmaxf(best[iq], max16(acc[iq]));
            }
        }
        if (more) commit_rows(p ^ 1);
        __syncthreads();
        p ^= 1;
    }
    flush();
}

// out[i,o] = f( min(own_o, 5e4 + min_{o' != o} own_o') ), own_o = |q_i|^2 - 2^-19 max-accumulator (+inf: no pixel of o)
__global__ __launch_bounds__(256) void dense_split_finalize_kernel(const float *__restrict__ partial, int n_split, int64_t m, int n_obj,
                                                                    const int32_t *__restrict__ counts, const int32_t *__restrict__ gate,
                                                                    const float *__restrict__ q2, const float *__restrict__ obj_bias,
                                                                    float *__restrict__ out, int64_t pstride, int64_t ostride, int transform) {
    if (*gate) return;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    int n_tiles = 0;
    for (int o = 0; o < n_obj; ++o) n_tiles += (counts[o] + SP_TILE - 1) / SP_TILE;
    const int tps = (n_tiles + n_split - 1) / max(n_split, 1);
    const float qq = q2[row];
    float own[16];
    int base = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        own[o] = INFINITY;
        if (o < n_obj) {
            const int nt = (counts[o] + SP_TILE - 1) / SP_TILE;
            if (nt > 0) {
                float v = -INFINITY;
                const int s0 = base / tps, s1 = (base + nt - 1) / tps;
                for (int s = s0; s <= s1; ++s) v = __builtin_fmaxf(v, partial[((size_t)s * m + row) * n_obj + o]);
                own[o] = qq + SP_UNSCALE * v;
            }
            base += nt;
        }
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        if (o < n_obj) {
            float others = INFINITY;
#pragma unroll
            for (int o2 = 0; o2 < 16; ++o2)
                if (o2 != o && o2 < n_obj) others = fminf(others, own[o2]);
            float v = fminf(own[o], others + AOC_PAD_DISTANCE);
            if (n_tiles == 0) v = transform ? 1.0f : INFINITY;            // AEM:796-797
            else if (transform) v = aoc_proto_transform(v, obj_bias ? obj_bias[o] : 0.0f);
            out[row * pstride + o * ostride] = v;
        }
    }
}

inline int split_nsplit(int64_t m) {
    const int64_t row_blocks = (m + SP_ROWS_PER_BLOCK - 1) / SP_ROWS_PER_BLOCK;
    static const int max_rounds = getenv("AOC_DENSE_ROUNDS") ? atoi(getenv("AOC_DENSE_ROUNDS")) : 4;
    // CUs the launching stream may use (256 unless the caller runs it under a HIP CU mask and says so)
    static const int n_cu = (getenv("AOC_DENSE_CUS") && atoi(getenv("AOC_DENSE_CUS")) > 0) ? atoi(getenv("AOC_DENSE_CUS")) : 256;
    int best = 1;
    double best_eff = 0.0;
    for (int k = 1; k <= max_rounds; ++k) {
        int64_t ns = ((int64_t)n_cu * k) / row_blocks;
        if (ns < 1) ns = 1;
        if (ns > 64) ns = 64;
        const int64_t blocks = row_blocks * ns;
        const int64_t rounds = (blocks + n_cu - 1) / n_cu;
        const double eff = (double)blocks / ((double)n_cu * rounds);
        if (eff >= best_eff - 0.005) { best_eff = eff > best_eff ? eff : best_eff; best = (int)ns; }
    }
    return best;
}

struct SplitWs {
    int32_t *gate, *n_tiles, *tile_rows, *tile_obj;
    float *partial;
    void *fp32_ws;
    size_t fp32_bytes, total;
    int64_t tile_capacity;
};
inline SplitWs split_carve(void *base, int64_t m, int64_t n, int n_obj) {
    SplitWs w;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += aoc_align_up(bytes, 256); return r; };
    w.tile_capacity = (n + SP_TILE - 1) / SP_TILE + n_obj + 2 * SP_NB;
    w.gate = reinterpret_cast<int32_t *>(take(16));
    w.n_tiles = w.gate ? w.gate + 2 : nullptr;
    w.tile_rows = reinterpret_cast<int32_t *>(take((size_t)w.tile_capacity * SP_TILE * sizeof(int32_t)));
    w.tile_obj = reinterpret_cast<int32_t *>(take((size_t)w.tile_capacity * sizeof(int32_t)));
    w.partial = reinterpret_cast<float *>(take((size_t)split_nsplit(m) * m * n_obj * sizeof(float)));
    w.fp32_bytes = aoc_dense_match_workspace_bytes(m, n, n_obj);
    w.fp32_ws = take(w.fp32_bytes);
    w.total = off;
    return w;
}

}  // namespace

extern "C" {

size_t aoc_split_record_bytes(int C) { return (C >= 4 && (C & 3) == 0 && C <= SP_NORM_SLOT) ? (size_t)SP_REC * 16 : 0; }

int aoc_split_rows(const float *x, int64_t n, int C, void *records, float *sqnorm, int32_t *overflow_flag, aoc_stream_t stream) {
    if (!x || !records || !overflow_flag || n < 0) return AOC_ERR_INVALID_ARG;
    if (aoc_split_record_bytes(C) == 0) return AOC_ERR_UNSUPPORTED;
    if (n == 0) return AOC_OK;
    const int64_t total = n * SP_KS;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), x, n, C,
                       static_cast<uint4 *>(records), sqnorm, overflow_flag);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_dense_match_split_workspace_bytes(int64_t m, int64_t n, int n_obj) {
    if (m < 1 || n < 1 || n_obj < 1) return 0;
    return split_carve(nullptr, m, n, n_obj).total;
}

int aoc_dense_match_min_split(const float *query, const void *query_rec, const float *query_sqnorm, int64_t m, int C, const float *pool,
                              const void *pool_rec, const int32_t *overflow_flag, int64_t n, const uint32_t *right_bits,
                              const uint32_t *wrong_bits, const int32_t *fg_rows, const int32_t *obj_rows, const int32_t *counts,
                              const int32_t *obj_offsets, const float *obj_bias, int n_obj, float *out, int64_t out_pixel_stride,
                              int64_t out_obj_stride, int transform, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!query || !query_rec || !query_sqnorm || !pool || !pool_rec || !overflow_flag || !right_bits || !wrong_bits || !fg_rows ||
        !obj_rows || !counts || !obj_offsets || !out || !workspace)
        return AOC_ERR_INVALID_ARG;
    if (m < 1 || n < 1 || n >= (1ll << 31) - 4096 || n_obj < 1) return AOC_ERR_INVALID_ARG;
    if (aoc_split_record_bytes(C) == 0 || n_obj > 16) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_dense_match_split_workspace_bytes(m, n, n_obj)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    const SplitWs w = split_carve(workspace, m, n, n_obj);
    if (hipMemsetAsync(w.gate, 0, 16, st) != hipSuccess) return AOC_ERR_LAUNCH;
    const int64_t plan_threads = w.tile_capacity * SP_TILE > n ? w.tile_capacity * SP_TILE : n;
    hipLaunchKernelGGL(split_plan_kernel, dim3((unsigned)((plan_threads + 255) / 256)), dim3(256), 0, st, obj_rows, counts, obj_offsets, n_obj, n,
                       right_bits, wrong_bits, overflow_flag, w.tile_capacity, w.tile_rows, w.tile_obj, w.n_tiles, w.gate);
    const int ns = split_nsplit(m);
    const dim3 grid((unsigned)((m + SP_ROWS_PER_BLOCK - 1) / SP_ROWS_PER_BLOCK), ns);
    const size_t lds = (size_t)2 * SP_NB * SP_TILE * SP_LDS_ROW * 16 + 2 * SP_NB * sizeof(int32_t);
    const AocDenseProbe probe = aoc_take_dense_probe();
    if (probe.start) (void)hipEventRecord(probe.start, st);
    hipLaunchKernelGGL(dense_split_kernel, grid, dim3(SP_NW * 64), lds, st, static_cast<const uint4 *>(query_rec), m,
                       static_cast<const uint4 *>(pool_rec), w.tile_rows, w.tile_obj, w.n_tiles, w.gate, n_obj, w.partial);
    if (probe.stop) (void)hipEventRecord(probe.stop, st);
    hipLaunchKernelGGL(dense_split_finalize_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w.partial, ns, m, n_obj, counts, w.gate,
                       query_sqnorm, obj_bias, out, out_pixel_stride, out_obj_stride, transform);
    AOC_RETURN_IF_LAUNCH_FAILED();
    // exact-fp32 kernels: run only when the gate is set
    return aoc_dense_match_min_gated(query, m, C, pool, fg_rows, counts + n_obj, n, wrong_bits, obj_bias, n_obj, out, out_pixel_stride,
                                     out_obj_stride, transform, w.fp32_ws, w.fp32_bytes, w.gate, stream);
}

}  // extern "C"
