// Dense pixel-level matching (AEM:61-89, 178-227) on the fp16 matrix pipe with fp32-equivalent products.
//
// Every fp32 value x (scaled by 2^10) is split into hi = fp16(x') and lo = fp16(x' - hi): hi + lo represents x' to 2^-22 relative
// (two 11-bit significands; typically 2^-23), and q.r = qh.rh + qh.rl + ql.rh (+ a ql.rl term < 2^-22 |q||r| that is dropped)
// accumulates in fp32 inside v_mfma_f32_32x32x16_f16.  The reference pixel's -|r|^2/2 rides along in three spare k-slots of the hi
// plane (K = 100 pads to 112 anyway), so one accumulator holds 2^20 * (q.r - |r|^2/2) and the min over reference pixels becomes a max
// over raw accumulators.  |q|^2 and the 5e4 wrong-label padding (AEM:84-88) are applied per query pixel at the end: with one-hot
// labels min_j(d_j + 5e4 wrong[j,o]) = min(own_o, 5e4 + min_{o' != o} own_o').
//
// dense_prune_kernel evaluates the qh.rh product everywhere and the two cross products only where they can matter (see its header):
// about 40 % of the matrix instructions of evaluating all three everywhere, for the same result.
//
// This kernel only runs when (a) every scaled value fits fp16 and (b) every kept reference pixel is right for
// exactly one object; both facts are device flags, and the exact-fp32 kernels of correlation.hip take over on
// the same stream otherwise (each side checks the flag itself: no host round trip).
#include <atomic>
#include <type_traits>

#include "aoc_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SP_KS = 7;                       // k-steps of 16 halves: 112 slots = 100 channels + 3 norm slots + pad
constexpr int SP_K = SP_KS * 16;
constexpr int SP_HALF = SP_KS * 2;             // 16-byte chunks per plane: per k-step [k0-7][k8-15]
constexpr int SP_REC = 2 * SP_HALF;            // 16-byte chunks per record: the hi plane (14 chunks), then the lo plane
constexpr int SP_NORM_SLOT = 100;              // slots 100..102 of the hi plane: the three fp16 pieces of -16 |r|^2
constexpr int SP_REST_SLOT = 104;              // slots 104, 105 of the hi plane: upper bounds of the Euclidean norm of the row's hi plane over the k-steps
                                               // 2..5 / 3..5 (channels 32..95 / 48..95): the checkpoint bound of dense_prune_kernel<NW, 3 / 4>
constexpr float SP_SCALE = 1024.0f;            // 2^10
constexpr float SP_QCONST = 32768.0f;          // query-side value of the norm slots: 2^15 * (-16 |r|^2) = -2^19 |r|^2
constexpr float SP_UNSCALE = -1.0f / 524288.0f;   // d - |q|^2 = -2^-19 * acc
constexpr int SP_TILE = 32;                    // reference pixels per MFMA tile
constexpr int SP_NB = 4;                       // tiles per staged chunk (the product; dense_prune_kernel<4, 0, 2> stages two)
constexpr int SP_NQ = 2;                       // 32-pixel query tiles per wave (stationary B operands in registers)

static_assert(SP_NORM_SLOT + 4 <= SP_K && SP_NORM_SLOT / 16 == SP_KS - 1 && (SP_NORM_SLOT % 16) + 4 <= 8, "norm slots live in the low half of the last k-step");
static_assert(SP_REST_SLOT / 16 == SP_KS - 1 && SP_REST_SLOT % 16 == 8, "the rest-norm slots are the first two halves of the last k-step's high chunk");

// ------------------------------------------------------------------------------------------
// fp32 rows -> split records (+ |x|^2).  One thread per (row, k-step).  Record = hi plane (14 x 16 B: per k-step [k0-7][k8-15]), then the
// lo plane in the same order; hi-plane slots 100..102 = the three fp16 pieces of -16 |x|^2, slot 103 = an upper bound of the lo plane's norm.
//
// TILED: the same chunks in the order a wave consumes them as MFMA B operands -- per 32-row tile, per plane, per k-step the 64 chunks
// [k-half h][row j] (1 KiB: ONE fully coalesced wave load per operand, no transposition on the consumer's side).  The buffer holds whole
// tiles; rows past n are written as zeros.  Threads: j fastest, so that a half wave writes 512 consecutive bytes.
template <bool TILED>
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, int64_t n, int C, uint4 *__restrict__ rec,
                                                          float *__restrict__ sqnorm, int32_t *__restrict__ overflow) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row;
    int ks;
    if (TILED) {
        const int64_t tile = idx / (SP_KS * 32);
        const int rem = (int)(idx - tile * (SP_KS * 32));
        ks = rem >> 5;
        row = tile * 32 + (rem & 31);
        if (tile * 32 >= n) return;
    } else {
        row = idx / SP_KS;
        ks = (int)(idx - row * SP_KS);
        if (row >= n) return;
    }
    const bool live = row < n;                    // TILED: the last tile's rows past n are zero records
    const float *xr = x + (size_t)(live ? row : n - 1) * C;
    _Float16 hi[16], lo[16];
    bool bad = false;
    {
        // the k-step's 16 channels as four float4 loads issued together (clamped index, zero selected past the channels)
        const int c4k = C >> 2;
        float4 kv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kv[q] = reinterpret_cast<const float4 *>(xr)[min(ks * 4 + q, c4k - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = ks * 4 + q < c4k;
            const float e4[4] = {kv[q].x, kv[q].y, kv[q].z, kv[q].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = in ? e4[u] * SP_SCALE : 0.0f;
                bad |= !(fabsf(v) <= 65000.0f);
                hi[q * 4 + u] = (_Float16)v;
                lo[q * 4 + u] = (_Float16)(v - (float)hi[q * 4 + u]);
            }
        }
    }
    if (ks == SP_KS - 1) {
        float s = 0.0f, sl = 0.0f, hr3 = 0.0f, hr4 = 0.0f;
        // the whole row as 25 float4 loads issued together (a scalar loop is one dependent round trip per channel); the additions keep
        // the sequential order t = 0 .. C-1
        float4 rv[SP_NORM_SLOT / 4];
        const int c4n = C >> 2;
#pragma unroll
        for (int t4 = 0; t4 < SP_NORM_SLOT / 4; ++t4) rv[t4] = reinterpret_cast<const float4 *>(xr)[t4 < c4n ? t4 : c4n - 1];
#pragma unroll
        for (int t4 = 0; t4 < SP_NORM_SLOT / 4; ++t4) {
            if (t4 < c4n) {
                const float e[4] = {rv[t4].x, rv[t4].y, rv[t4].z, rv[t4].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s += e[u] * e[u];
                    const float v = e[u] * SP_SCALE;
                    const float hv = (float)(_Float16)v;                            // the hi value of the channel, as its own thread stores it
                    const float l = (float)(_Float16)(v - hv);                      // ... and the lo value
                    sl += l * l;
                    if (t4 >= 8 && t4 < 24) hr3 += hv * hv;                         // k-steps 2..5
                    if (t4 >= 12 && t4 < 24) hr4 += hv * hv;                        // k-steps 3..5
                }
            }
        }
        if (sqnorm && live) sqnorm[row] = s;
        bad |= !(s <= 4000.0f);
        const float p = -16.0f * s;
        const _Float16 p1 = (_Float16)p;
        const _Float16 p2 = (_Float16)(p - (float)p1);
        const _Float16 p3 = (_Float16)((p - (float)p1) - (float)p2);
        hi[SP_NORM_SLOT % 16] = p1;
        hi[SP_NORM_SLOT % 16 + 1] = p2;
        hi[SP_NORM_SLOT % 16 + 2] = p3;
        // slot 103: an upper bound of the Euclidean norm of the row's lo plane (the rescoring margin of the dense kernel); the
        // query side multiplies it by zero
        hi[SP_NORM_SLOT % 16 + 3] = (_Float16)(sqrtf(sl) * 1.002f + 1e-6f);
        // slots 104, 105: upper bounds of |hi plane| over the channels the dense kernel has NOT yet accumulated at its checkpoint (after the
        // k-steps 6, 0, 1 / 6, 0, 1, 2): Cauchy-Schwarz turns the two sides' values into an upper bound of the rest of the hi x hi product.
        // Every consumer multiplies them by zero on one side except that kernel (the query side's other slot is masked there; the proxy image of
        // the correlation kernel has zeros in slots 103..111)
        // Development build only (the checkpoint is not in the product): release records keep zeros there, so no product of two records depends on them
#ifdef AOC_DEV
        hi[SP_REST_SLOT % 16] = (_Float16)(sqrtf(hr3) * 1.002f + 1e-6f);
        hi[SP_REST_SLOT % 16 + 1] = (_Float16)(sqrtf(hr4) * 1.002f + 1e-6f);
#else
        (void)hr3;
        (void)hr4;
#endif
    }
    if (bad && live) atomicOr(overflow, 1);
    union { _Float16 h[32]; uint4 q[4]; } u;
#pragma unroll
    for (int e = 0; e < 16; ++e) { u.h[e] = hi[e]; u.h[16 + e] = lo[e]; }
    if (TILED) {
        if (!live) u.q[0] = u.q[1] = u.q[2] = u.q[3] = make_uint4(0, 0, 0, 0);
        uint4 *dst = rec + ((size_t)(row >> 5) * 2 * SP_KS + ks) * 64 + (row & 31);
        dst[0] = u.q[0];
        dst[32] = u.q[1];
        dst[SP_KS * 64] = u.q[2];
        dst[SP_KS * 64 + 32] = u.q[3];
    } else {
        uint4 *dst = rec + (size_t)row * SP_REC + ks * 2;
        dst[0] = u.q[0];
        dst[1] = u.q[1];
        dst[SP_HALF] = u.q[2];
        dst[SP_HALF + 1] = u.q[3];
    }
}

// ------------------------------------------------------------------------------------------
// Plan: the per-object row lists of label prep cut into 32-row tiles (object-pure, -1 padded), the tile count, and
// the one-hot check.  gate[0] |= overflow | (some kept row is not right for exactly one object).
__global__ __launch_bounds__(256) void split_plan_kernel(const int32_t *__restrict__ obj_rows, const int32_t *__restrict__ counts,
                                                          const int32_t *__restrict__ obj_offsets, int n_obj, int64_t n,
                                                          const uint32_t *__restrict__ right_bits, const uint32_t *__restrict__ wrong_bits,
                                                          const int32_t *__restrict__ overflow, const uint4 *__restrict__ prec,
                                                          int64_t tile_capacity, int32_t *__restrict__ tile_rows,
                                                          int32_t *__restrict__ tile_obj, int32_t *__restrict__ n_tiles,
                                                          int32_t *__restrict__ gate, uint32_t *__restrict__ pmax_bits) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float sq = 0.0f, ln = 0.0f;
    if (e < n) {
        const uint32_t mask = (n_obj >= 32) ? 0xffffffffu : ((1u << n_obj) - 1u);
        const uint32_t right = right_bits[e];
        if (right & AOC_ROW_KEPT_BIT) {
            const uint32_t r = right & mask, nw = ~wrong_bits[e] & mask;
            if (__popc(r) != 1 || nw != r) atomicOr(gate, 1);
            // |r|^2 back from the record's norm slots (three fp16 pieces of -16 |r|^2)
            union { uint4 q; _Float16 hh[8]; } u;
            u.q = prec[(size_t)e * SP_REC + (SP_KS - 1) * 2];
            sq = -((float)u.hh[SP_NORM_SLOT % 16] + (float)u.hh[SP_NORM_SLOT % 16 + 1] + (float)u.hh[SP_NORM_SLOT % 16 + 2]) * 0.0625f;
            ln = (float)u.hh[SP_NORM_SLOT % 16 + 3];
        }
    }
    // largest squared norm among the kept reference pixels (non-negative floats order like their bit patterns)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sq = __builtin_fmaxf(sq, __shfl_xor(sq, d));
        ln = __builtin_fmaxf(ln, __shfl_xor(ln, d));
    }
    // one atomic per wave only while the wave can still raise the maximum (thousands of atomics on two addresses serialise: most of this
    // kernel's 64 us): a relaxed device-scope read first, the atomic only if the wave's value is larger than what is already there
    if (aoc_lane() == 0 && sq > 0.0f && __float_as_uint(sq) > __hip_atomic_load(pmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(pmax_bits, __float_as_uint(sq));
    if (aoc_lane() == 0 && ln > 0.0f && __float_as_uint(ln) > __hip_atomic_load(pmax_bits - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(pmax_bits - 2, __float_as_uint(ln));      // gate[1]: largest lo-plane norm
    if (e == 0 && overflow && *overflow) atomicOr(gate, 1);
    const int64_t t = e / SP_TILE;
    const int i = (int)(e - t * SP_TILE);
    if (t >= tile_capacity) return;
    int64_t base = 0;
    int obj = -1, local = 0;
    for (int o = 0; o < n_obj; ++o) {
        const int64_t nt = (counts[o] + SP_TILE - 1) / SP_TILE;
        // an object's tiles in DESCENDING row order: the newest pool frame first -- that is where a video's best matches usually are, and an
        // early good match tightens the bounds for everything after it
        if (obj < 0 && t < base + nt) { obj = o; local = (int)(nt - 1 - (t - base)); }
        base += nt;
    }
    if (e == 0) *n_tiles = (int32_t)base;
    // a partial last tile is filled up with copies of its first row (a duplicate cannot change a maximum): no padding values anywhere
    int32_t id = -1;
    if (obj >= 0) {
        const int pos = local * SP_TILE + i;
        id = obj_rows[obj_offsets[obj] + (pos < counts[obj] ? pos : local * SP_TILE)];
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

// float <-> unsigned with the same order (atomicMax on the encoding = max on the floats); every encoded finite value or infinity
// is > 0, so a zeroed word means "nothing yet"
__device__ __forceinline__ uint32_t ord_enc(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(uint32_t u) {
    if (u == 0u) return -INFINITY;
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ uint32_t load_relaxed(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef AOC_DEV
// development build, AOC_DENSE_DEBUG bit 32768: per workgroup [start, prologue done, first step done, end] on the 100 MHz wall clock + [tiles, rescored pairs]
// of wave 0 (tools/dense_block_timeline.py reads the symbol through the HIP runtime)
__device__ unsigned long long aoc_dev_block_times[4096 * 6];
#endif
__device__ unsigned long long g_prune_stats[8];    // (tile, query tile) pairs tested / rescored, tiles with any rescoring, tiles, pairs stopped at the checkpoint

// LDS-DMA: 64 lanes x 16 bytes (or 4 bytes) from per-lane global addresses to the LDS bytes [lds_dst + 16 lane, +16).  Written in asm so
// that hipcc neither counts nor drains it: completion is the kernel's own vmcnt arithmetic (see the step loop).
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int SP_NBUF = 2;                                        // chunk buffers in LDS
// LDS layout of a workgroup of NW waves that stages NB tiles per chunk
__host__ __device__ constexpr int sp_chunk_bytes(int nb) { return nb * SP_TILE * SP_REC * 16; }     // NB = 4: 57344 (four tiles x 32 rows x 448 B, rows unpadded)
__host__ __device__ constexpr int sp_ids_off(int nb) { return SP_NBUF * sp_chunk_bytes(nb); }       // 2 slots x (up to) 128 row ids
__host__ __device__ constexpr int sp_obj_off(int nb) { return sp_ids_off(nb) + 2 * 512; }           // 4 slots x 64 tile objects (NB used)
__host__ __device__ constexpr int sp_bnd_off(int nb) { return sp_obj_off(nb) + 4 * 256; }           // per (wave, query tile): 64 published bounds
__host__ __device__ constexpr int sp_lds_bytes(int nw, int nb) { return sp_bnd_off(nb) + nw * SP_NQ * 256; }
constexpr int SP_TILE_SLACK = 2;                                  // the plan always holds an empty tile after the last one

// Coarse-then-rescore.  Block = 8 waves x 2 query tiles (512 query pixels; both planes of their records are the stationary B
// operands, 112 VGPR); the object-sorted reference tiles stream through two LDS chunk buffers (4 tiles each, both planes, A
// operands).  Grid = (query blocks, tile splits).  Per (reference tile, query tile) ONE pass of 7 MFMAs gives
// coarse = 2^20 (qh.rh - |r|^2/2); the exact three-product value differs from it by the qh.rl + ql.rh terms, bounded by
// eps(q) = 2^10 |q| max|r| (1 + margins): a pair whose coarse value plus eps is below the best EXACT value already known for that
// (query pixel, object) cannot hold the maximum and is skipped; any other pair gets the 14 MFMAs of the two cross terms added onto the
// same accumulators (everything is on chip: no memory latency on that path), which is then exactly the three-product value.  The best
// exact values live in gbest[pixel][object] (atomicMax on an order-preserving encoding) and are shared by all splits and by the
// workgroups of later rounds, so the bound tightens after the first few tiles anywhere on the chip.  The true maximum always survives
// (coarse + eps >= exact >= every bound) and its value does not depend on what else was evaluated: the result is deterministic
// although the set of rescored tiles is not.
//
// Data movement: the chunk of step s + 1 is fetched by LDS-DMA while step s computes (no staging registers, no ds_write pass); its row
// ids (and the tiles' objects) were themselves DMA'd one step earlier, and the bounds other workgroups published come in the same
// way.  LDS rows are unpadded (448 B); chunk c of row r sits at position c ^ ((r >> 3) & 3), which makes every ds_read_b128 of an A
// fragment conflict-free -- the swizzle is applied on the SOURCE address of the DMA, whose destination is lane-linear.  The transfers
// are asm statements that hipcc neither counts nor drains; one vmcnt(0) + barrier per step (4 tiles) publishes them.
//
// Checkpoint (CKPT = 3 or 4, development build only; 0 = off = the product: built and measured in round 5, correct and SLOWER, see split_ckpt()).  After CKPT of the 7 k-steps -- order 6, 0, 1, 2, ...: the norm slots first -- the
// accumulator holds 2^20 (P - |r|^2 / 2) with P the hi x hi product over the channels seen so far.  What the remaining k-steps can add is at most
// |qh_rest| |rh_rest| (Cauchy-Schwarz on the hi planes).  Both norms ride in the records (slot 104: rest = k-steps 2..5, slot 105: rest =
// k-steps 3..5, rounded up), so the FIRST MFMA of the tile already adds their product: at the checkpoint the accumulator is an UPPER BOUND of the
// final coarse value of every pair, and a (reference tile, query tile) pair whose largest bound plus eps is below what is already known for
// its pixels stops there -- 4 (3) k-steps, its final test and any rescoring skipped.  A surviving pair takes the product out again with one more
// MFMA whose A fragment is zero except for the negated norm slot, then continues as before.  The bound is rigorous (the products of the fp16
// values are exact, the norms are rounded up, the accumulations' roundings are inside eps), so the set of discarded pairs can never contain
// the maximum: same results as CKPT = 0, deterministic as before (a pair's value does not depend on what else was evaluated).
template <int NW, int CKPT, int NB = SP_NB>
__global__ __launch_bounds__(NW * 64, NB == 2 ? 2 : 1) void dense_prune_kernel(const uint4 *__restrict__ qrec, const float *__restrict__ q2, int64_t m,
                                                                     const uint4 *__restrict__ prec, const int32_t *__restrict__ tile_rows,
                                                                     const int32_t *__restrict__ tile_obj, const int32_t *__restrict__ n_tiles_ptr,
                                                                     const int32_t *__restrict__ gate, const uint32_t *__restrict__ pmax_bits,
                                                                     int n_obj, uint32_t *__restrict__ gbest, int dbg_arg, int q_tiled) {
    if (*gate) return;
    // developer bits (timing experiments that give WRONG results on purpose) exist in the development build only: in the release library `dbg` is
    // the constant 0 and every test on it folds away
#ifdef AOC_DEV
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;
    (void)dbg_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    static_assert(SP_NQ == 2 && ((NB == 4 && (NW == 8 || NW == 4)) || (NB == 2 && NW == 4)),
                  "the step structure below is written for 4 tiles x 2 query tiles x 8 (or 4) waves, or 2 tiles x 2 query tiles x 4 waves (two workgroups per CU)");
    constexpr int SP_CHUNK_BYTES = sp_chunk_bytes(NB), SP_IDS_OFF = sp_ids_off(NB), SP_OBJ_OFF = sp_obj_off(NB), SP_BND_OFF = sp_bnd_off(NB);
    static_assert(CKPT == 0 || CKPT == 3 || CKPT == 4, "checkpoint after 3 or 4 k-steps (rest norms in slots 104 / 105), or none");
    constexpr int SP_DMA_PER_WAVE = SP_CHUNK_BYTES / 1024 / NW;      // 7 (14) wave-wide 1 KiB transfers per wave and chunk
    constexpr int W_ID0 = NW / 2, W_ID1 = NW - 1, W_OBJ = 1;         // the waves that also fetch the row ids / the tiles' objects
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds4);
    const char *lds_bytes = reinterpret_cast<const char *>(lds4);

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
    // split `by` owns the tiles by, by + ns, by + 2 ns, ...: every split sees the same mix of objects (small objects rescore far more often
    // than the background; contiguous ranges would leave all of that work to the last splits, i.e. to one XCD)
    const int n_tiles = *n_tiles_ptr;
    const int ns = gridDim.y;
    if (by >= n_tiles) return;
    const int n_mine = (n_tiles - by + ns - 1) / ns;
    const int n_chunks = (n_mine + NB - 1) / NB;

    const int lane = aoc_lane(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, h = lane >> 5;
    const int64_t wave_row0 = (int64_t)bx * (NW * SP_NQ * 32) + (int64_t)wave * (SP_NQ * 32);

    // ---- stationary query operands (both planes) and the per-pixel rescoring margin
    const float pmax = sqrtf(__uint_as_float(*pmax_bits)) * 1.001f;
    const float plmax = __uint_as_float(*(pmax_bits - 2));
    f16x8 bh[SP_NQ][SP_KS], bl[SP_NQ][SP_KS];
    float eps[SP_NQ];
    bool valid[SP_NQ];
#pragma unroll
    for (int iq = 0; iq < SP_NQ; ++iq) {
        const int64_t row = wave_row0 + iq * 32 + col;
        valid[iq] = row < m;
        // row-major records: chunk (plane, ks, h) of row r at r * 28 + plane * 14 + ks * 2 + h; tile-major (aoc_split_rows_tiled): per
        // (32-row tile, plane, ks) the 64 chunks [h][row % 32] -- one coalesced 1 KiB wave load per operand
        const int64_t rr = valid[iq] ? row : 0;
        const uint4 *r = q_tiled ? qrec + (size_t)(rr >> 5) * (2 * SP_KS * 64) + h * 32 + (rr & 31) : qrec + (size_t)rr * SP_REC + h;
        const int ks_step = q_tiled ? 64 : 2, plane_step = q_tiled ? SP_KS * 64 : SP_HALF;
#pragma unroll
        for (int ks = 0; ks < SP_KS; ++ks) {
            uint4 u = r[ks * ks_step], v = r[plane_step + ks * ks_step];
            if (!valid[iq]) { u = make_uint4(0, 0, 0, 0); v = make_uint4(0, 0, 0, 0); }
            bh[iq][ks] = __builtin_bit_cast(f16x8, u);
            bl[iq][ks] = __builtin_bit_cast(f16x8, v);
        }
        // norm slots: the query side holds the constant 2^15 (its own norm pieces sit in the record for when the
        // frame later joins the pool); everything else past the channels is zero on both planes
        const float ql_own = (float)bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 3];      // lanes h == 0: the norm of the query pixel's own lo plane
        if (h == 0) {
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 3] = (_Float16)0.0f;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16] = (_Float16)SP_QCONST;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 1] = (_Float16)SP_QCONST;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 2] = (_Float16)SP_QCONST;
        } else {
            // slots 104 / 105: the query pixel's own rest norms.  The checkpoint's one stays (x the reference row's = the bound's rank-1 term),
            // the other one -- both without a checkpoint -- is multiplied by zero
            if (CKPT != 3) bh[iq][SP_KS - 1][0] = (_Float16)0.0f;
            if (CKPT != 4) bh[iq][SP_KS - 1][1] = (_Float16)0.0f;
        }
        // |exact - coarse| = |qh.rl + ql.rh| <= |qh| |rl| + |ql| |rh| (Euclidean norms of the planes, Cauchy-Schwarz) with |qh| <= 2^10 |q|
        // (1 + 2^-11), the lo norms as the records carry them (rounded up) and the maxima over the kept reference pixels, plus the
        // roundings of 14 more accumulations
        const float qn = valid[iq] ? sqrtf(q2[row]) : 0.0f;
        const float ql = valid[iq] ? __shfl(ql_own, col) : 0.0f;
        // (with a checkpoint the partial sums also carry the bound's term, at most 2^20 |q| max|r|: 16 |q| max|r| covers the roundings at that size)
        eps[iq] = (1026.0f * (qn * plmax + ql * pmax) + 8.0f * pmax * pmax + (CKPT != 0 ? 16.0f * qn * pmax : 0.0f) + 8.0f) * ((dbg & 64) ? 0.4f : 1.0f);
    }

    // ---- DMA plan of this wave: transfer k of a chunk fills the LDS slots [64 (7 wave + k), +64); slot j holds row j / 28, position j % 28.
    // Packed per transfer: row of the chunk (0..127) in the high half, source byte offset of that position's chunk in the low half.
    uint32_t dma_plan[SP_DMA_PER_WAVE];
#pragma unroll
    for (int k = 0; k < SP_DMA_PER_WAVE; ++k) {
        const int j = (wave * SP_DMA_PER_WAVE + k) * 64 + lane;
        const int r = j / SP_REC, pos = j - r * SP_REC;
        dma_plan[k] = ((uint32_t)r << 16) | (uint32_t)((pos ^ ((r >> 3) & 3)) * 16);
    }
    const char *prec_bytes = reinterpret_cast<const char *>(prec);
    auto dma_meta = [&](int chunk) {            // row ids (two waves, two tiles each) and tile objects (one wave) of a chunk -> their rings
        // tile i of the split is tile by + i ns of the plan; past the end everything reads the (always present) empty tile n_tiles
        const int i0 = chunk * NB;
        if (wave == W_ID0 || (NB == 4 && wave == W_ID1)) {
            const int i = i0 + (wave == W_ID1 ? 2 : 0) + (lane >> 5);
            const int t = min(by + i * ns, n_tiles);
            glds4(tile_rows + (size_t)t * SP_TILE + (lane & 31), lds_base + SP_IDS_OFF + (chunk & 1) * 512 + (wave == W_ID1 ? 256 : 0));
        }
        if (wave == W_OBJ) glds4(tile_obj + min(by + (i0 + (lane & (NB - 1))) * ns, n_tiles), lds_base + SP_OBJ_OFF + (chunk & 3) * 256);
    };
    auto dma_rows = [&](int chunk) {            // the chunk's records -> buffer chunk % 2 (its ids must have landed and been published)
        const int32_t *ids = reinterpret_cast<const int32_t *>(lds_bytes + SP_IDS_OFF + (chunk & 1) * 512);
        const uint32_t dst = lds_base + (uint32_t)(chunk % SP_NBUF) * SP_CHUNK_BYTES + (uint32_t)(wave * SP_DMA_PER_WAVE) * 1024u;
        int id[SP_DMA_PER_WAVE];
#pragma unroll
        for (int k = 0; k < SP_DMA_PER_WAVE; ++k) id[k] = ids[dma_plan[k] >> 16];      // all LDS reads before the first (ordering) asm statement
#pragma unroll
        for (int k = 0; k < SP_DMA_PER_WAVE; ++k)
            glds16(prec_bytes + (size_t)(uint32_t)max(id[k], 0) * (SP_REC * 16) + (dma_plan[k] & 0xffffu), dst + (uint32_t)k * 1024u);
    };

    // best[iq]: best exact value this lane has produced for the current object; shared[iq]: the best anybody has published
    float best[SP_NQ], shared[SP_NQ];
    uint32_t grow[SP_NQ];                          // grow: word index of (pixel, object 0) in gbest (pixel 0 for rows past the end)
#pragma unroll
    for (int iq = 0; iq < SP_NQ; ++iq) {
        grow[iq] = valid[iq] ? (uint32_t)(wave_row0 + iq * 32 + col) * (uint32_t)n_obj : 0u;
        best[iq] = INFINITY;
        shared[iq] = INFINITY;
    }
    int cur = -1;
    unsigned n_rescored = 0, n_any = 0, n_seen = 0, n_dead = 0;
    // development build, dbg 4096: core-clock stamps (s_memtime) around the pieces of a tile / a step, summed per wave.  dbg 8192 selects the second
    // triple.  [0] a tile's 14 coarse MFMAs (issue), [1] decision, [2] rescoring | [3] between tiles (object switch, loop), [4] step head (DMA
    // issue), [5] step tail (vmcnt(0), bound read-back, barrier)
    unsigned long long cyc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    auto stamp = [&]() -> unsigned long long { return (dbg & 4096) ? __builtin_amdgcn_s_memtime() : 0ull; };
    // what the other workgroups have published for the current object: one 4-byte transfer per query tile into this wave's own LDS
    // words, read back one step later (device-scope load: the values come from L2, not from this CU's vector cache)
    auto dma_bound = [&]() {
#pragma unroll
        for (int iq = 0; iq < SP_NQ; ++iq) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gbest + grow[iq] + max(cur, 0)), "s"(lds_base + SP_BND_OFF + (uint32_t)(wave * SP_NQ + iq) * 256u) : "memory");
        }
    };
    auto switch_object = [&](int o) {
        cur = o;
        uint32_t u[SP_NQ];
#pragma unroll
        for (int iq = 0; iq < SP_NQ; ++iq) u[iq] = load_relaxed(gbest + grow[iq] + cur);
#pragma unroll
        for (int iq = 0; iq < SP_NQ; ++iq) {
            best[iq] = valid[iq] ? -INFINITY : INFINITY;         // rows past the end never ask for a rescoring
            shared[iq] = valid[iq] ? ord_dec(u[iq]) : INFINITY;
        }
    };

#ifdef AOC_DEV
    const int blk_lin = blockIdx.y * gridDim.x + blockIdx.x;
    const bool blk_rec = (dbg & 32768) && threadIdx.x == 0 && blk_lin < 4096;
    if (blk_rec) aoc_dev_block_times[blk_lin * 6 + 0] = wall_clock64();
#endif
    // ---- prologue: meta of chunks 0 and 1, rows of chunk 0 (drained: once per workgroup)
    dma_meta(0);
    dma_meta(1);
    __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    dma_rows(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();

    // A fragment addressing: chunk index e + h (e even, compile time) of row (tile, col) sits at position (e & ~3) + (((e & 2) + h) ^ x),
    // x = (col >> 3) & 3: two per-lane byte offsets cover it
    const int xs = (col >> 3) & 3;
    const uint32_t row_off = (uint32_t)col * (SP_REC * 16);
    const uint32_t sw0 = row_off + (uint32_t)((h ^ xs) * 16), sw2 = row_off + (uint32_t)(((2 + h) ^ xs) * 16);
    auto frag = [&](const char *tile_base, int e) -> f16x8 {       // e: even chunk index (2 ks for the hi plane, 14 + 2 ks for the lo plane)
        const uint32_t off = ((e & 2) ? sw2 : sw0) + (uint32_t)((e & ~3) * 16);
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(tile_base + off));
    };
    auto ks_of = [&](int kk) { return (kk + SP_KS - 1) % SP_KS; };  // norm slots first: partial sums stay small
#ifdef AOC_DEV
    if (blk_rec) aoc_dev_block_times[blk_lin * 6 + 1] = wall_clock64();
#endif

    for (int s = 0; s < n_chunks; ++s) {
        // (a) prefetches for the next step: published bounds of the current object, meta of chunk s + 2, rows of chunk s + 1 (whose ids
        // the barrier that ended step s - 1 published)
        const unsigned long long t_s0 = stamp();
        const int4 objs = *reinterpret_cast<const int4 *>(lds_bytes + SP_OBJ_OFF + (s & 3) * 256);
        const int bound_obj = (dbg & 32) ? -3 : cur;
        if (!(dbg & 32)) dma_bound();
        dma_meta(s + 2);
        if (!(dbg & 4)) dma_rows(s + 1);

        // (b) the four tiles of chunk s
        const char *chunk_base = lds_bytes + (s % SP_NBUF) * SP_CHUNK_BYTES;
        const int n_here = min(NB, n_mine - s * NB);
        // the first two A fragments of a tile are requested while the previous tile's epilogue runs
        f16x8 pre0 = frag(chunk_base, 2 * ks_of(0)), pre1 = frag(chunk_base, 2 * ks_of(1));
        // the four tiles' objects in one scalar (objects are < 256): two SALU operations per tile instead of a chain of selects
        const uint32_t objs_packed = (uint32_t)__builtin_amdgcn_readfirstlane((objs.x & 0xff) | ((objs.y & 0xff) << 8) | ((objs.z & 0xff) << 16) | ((objs.w & 0xff) << 24));
        t_prev = stamp();
        cyc[4] += t_prev - t_s0;
#pragma unroll 1
        for (int t = 0; t < n_here; ++t) {
            const char *tile_base = chunk_base + t * (SP_TILE * SP_REC * 16);
            const int o = (int)((objs_packed >> (8 * t)) & 0xffu);
            if (o != cur) switch_object(o);
            const unsigned long long t_0 = stamp();
            cyc[3] += t_0 - t_prev;
            // coarse pass: 7 k-steps x 2 query tiles, A fragments two k-steps ahead through a ring of three
            f32x16 acc[SP_NQ];
#pragma unroll
            for (int iq = 0; iq < SP_NQ; ++iq)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[iq][r] = 0.0f;
            bool live[SP_NQ] = {true, true};
            if constexpr (CKPT == 0) {
                f16x8 af[3];
                af[0] = pre0;
                af[1] = pre1;
#pragma unroll
                for (int kk = 0; kk < SP_KS; ++kk) {
                    if (kk + 2 < SP_KS && !(dbg & 512)) af[(kk + 2) % 3] = frag(tile_base, 2 * ks_of(kk + 2));   // dbg 512: stale fragments, no LDS reads
#pragma unroll
                    for (int iq = 0; iq < SP_NQ; ++iq)
                        acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk % 3], bh[iq][ks_of(kk)], acc[iq], 0, 0, 0);
                }
                if (t + 1 < n_here) {
                    pre0 = frag(tile_base + SP_TILE * SP_REC * 16, 2 * ks_of(0));
                    pre1 = frag(tile_base + SP_TILE * SP_REC * 16, 2 * ks_of(1));
                }
            } else {
                // phase 1: the first CKPT k-steps (k-step 6 carries the norm slots AND the bound's rank-1 term)
                // (only the FIRST fragment of the next tile is requested ahead here: a second one would be live across phase 2 and the rescoring,
                // where the register file is full)
                f16x8 af[3];
                af[0] = pre0;
                af[1] = frag(tile_base, 2 * ks_of(1));
                const uint32_t a6d0 = __builtin_bit_cast(uint4, pre0).x;        // lanes h == 1: the reference row's slots 104 | 105
#pragma unroll
                for (int kk = 0; kk < CKPT; ++kk) {
                    if (kk + 2 < CKPT) af[(kk + 2) % 3] = frag(tile_base, 2 * ks_of(kk + 2));
#pragma unroll
                    for (int iq = 0; iq < SP_NQ; ++iq)
                        acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk % 3], bh[iq][ks_of(kk)], acc[iq], 0, 0, 0);
                }
                if (t + 1 < n_here) pre0 = frag(tile_base + SP_TILE * SP_REC * 16, 2 * ks_of(0));
                // checkpoint: can any pair of this (reference tile, query tile) still reach what is known for its pixel?  (wave-uniform)
#pragma unroll
                for (int iq = 0; iq < SP_NQ; ++iq) {
                    const float cm = max16(acc[iq]);
                    live[iq] = (__builtin_amdgcn_ballot_w64(cm + eps[iq] >= __builtin_fmaxf(best[iq], shared[iq])) != 0ull) || (dbg & 128);
                    if (!live[iq]) n_dead += 1;
                }
                if (live[0] || live[1]) {
                    // phase 2: take the rank-1 term out again (A = the negated norm slot, zero elsewhere), then the remaining k-steps.  Three
                    // straight-line variants (both query tiles / one of them): no branch between the MFMAs
                    uint4 c4 = make_uint4(0u, 0u, 0u, 0u);
                    if (h == 1) c4.x = (CKPT == 3) ? ((a6d0 & 0x0000ffffu) ^ 0x00008000u) : ((a6d0 & 0xffff0000u) ^ 0x80000000u);
                    const f16x8 a6c = __builtin_bit_cast(f16x8, c4);
                    auto phase2 = [&](auto q0, auto q1) {
                        constexpr bool Q0 = decltype(q0)::value, Q1 = decltype(q1)::value;
                        f16x8 ar[3];
                        ar[0] = frag(tile_base, 2 * ks_of(CKPT));
                        ar[1] = frag(tile_base, 2 * ks_of(CKPT + 1));
                        if (Q0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a6c, bh[0][SP_KS - 1], acc[0], 0, 0, 0);
                        if (Q1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a6c, bh[1][SP_KS - 1], acc[1], 0, 0, 0);
#pragma unroll
                        for (int kk = CKPT; kk < SP_KS; ++kk) {
                            if (kk + 2 < SP_KS) ar[(kk + 2 - CKPT) % 3] = frag(tile_base, 2 * ks_of(kk + 2));
                            if (Q0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[(kk - CKPT) % 3], bh[0][ks_of(kk)], acc[0], 0, 0, 0);
                            if (Q1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[(kk - CKPT) % 3], bh[1][ks_of(kk)], acc[1], 0, 0, 0);
                        }
                    };
                    if (live[0] && live[1]) phase2(std::true_type{}, std::true_type{});
                    else if (live[0]) phase2(std::true_type{}, std::false_type{});
                    else phase2(std::false_type{}, std::true_type{});
                }
            }
            const unsigned long long t_1 = stamp();
            cyc[0] += t_1 - t_0;
            // which query tiles may hold a new maximum (wave-uniform)
            bool want[SP_NQ];
#pragma unroll
            for (int iq = 0; iq < SP_NQ; ++iq) {
                want[iq] = false;
                if (live[iq] && !(dbg & 256)) {                    // dbg 256: no decision (the accumulators are never read)
                    const float cm = max16(acc[iq]);
                    want[iq] = __builtin_amdgcn_ballot_w64(cm + eps[iq] >= __builtin_fmaxf(best[iq], shared[iq])) != 0ull;
                }
            }
            n_seen += 1;
            const unsigned long long t_2 = stamp();
            cyc[1] += t_2 - t_1;
            t_prev = t_2;
            if ((want[0] || want[1]) && !(dbg & 2)) {
                n_any += 1;
#pragma unroll
                for (int iq = 0; iq < SP_NQ; ++iq) {
                    if (want[iq]) {
                        // cross terms: acc += rh.ql + rl.qh (one chain of 14 MFMAs), then the exact maximum
                        n_rescored += 1;
                        if (dbg & 65536) {                                 // rescored pairs of the objects 0, 1, 2 (read back like the stamps).  STATIC indices:
                            if (cur == 0) cyc[0] += 1;                     // `cyc[cur]` made hipcc index the register array dynamically and the kernel -- whose
                            else if (cur == 1) cyc[1] += 1;                // DMA statements are hand-written asm around m0 -- returned wrong minima for nine
                            else if (cur == 2) cyc[2] += 1;                // objects even with the bit off (caught by test_kmeans_bit_exact_with_the_single_pass_tail)
                        }
                        f16x8 ah[3], al[3];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            ah[kk] = frag(tile_base, 2 * ks_of(kk));
                            al[kk] = frag(tile_base, 2 * SP_KS + 2 * ks_of(kk));
                        }
#pragma unroll
                        for (int kk = 0; kk < SP_KS; ++kk) {
                            if (kk + 2 < SP_KS) {
                                ah[(kk + 2) % 3] = frag(tile_base, 2 * ks_of(kk + 2));
                                al[(kk + 2) % 3] = frag(tile_base, 2 * SP_KS + 2 * ks_of(kk + 2));
                            }
                            acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk % 3], bl[iq][ks_of(kk)], acc[iq], 0, 0, 0);
                            acc[iq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk % 3], bh[iq][ks_of(kk)], acc[iq], 0, 0, 0);
                        }
                        const float ex = max16(acc[iq]);
                        if (ex > best[iq]) {
                            best[iq] = ex;
                            if (ex > shared[iq] && !(dbg & 1)) atomicMax(gbest + grow[iq] + cur, ord_enc(ex));      // fire and forget
                        }
                    }
                }
                t_prev = stamp();
                cyc[2] += t_prev - t_2;
            }
        }
        const unsigned long long t_s1 = stamp();

        // (c) this step's transfers have landed (they had four tiles of time).  The wave reads back its own bound words right away (its
        // own vmcnt(0) covers them) so that the lgkmcnt(0) below also retires those reads before the next step's transfer can overwrite
        // the words; then the barrier publishes the chunk and the row ids to the other waves.
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
        uint32_t seen[SP_NQ];
#pragma unroll
        for (int iq = 0; iq < SP_NQ; ++iq)
            seen[iq] = *reinterpret_cast<const volatile uint32_t *>(lds_bytes + SP_BND_OFF + (wave * SP_NQ + iq) * 256 + lane * 4);
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
        if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
        if (bound_obj == cur && cur >= 0) {
#pragma unroll
            for (int iq = 0; iq < SP_NQ; ++iq)
                if (valid[iq]) shared[iq] = __builtin_fmaxf(shared[iq], ord_dec(seen[iq]));
        }
        cyc[5] += stamp() - t_s1;
#ifdef AOC_DEV
        if (blk_rec && s == 0) aoc_dev_block_times[blk_lin * 6 + 2] = wall_clock64();
#endif
    }
#ifdef AOC_DEV
    if (blk_rec) {
        aoc_dev_block_times[blk_lin * 6 + 3] = wall_clock64();
        aoc_dev_block_times[blk_lin * 6 + 4] = n_seen;
        aoc_dev_block_times[blk_lin * 6 + 5] = n_rescored;
    }
#endif
    if (lane == 0) {
        atomicAdd(&g_prune_stats[0], (unsigned long long)n_seen * SP_NQ);
        atomicAdd(&g_prune_stats[1], (unsigned long long)n_rescored);
        atomicAdd(&g_prune_stats[2], (unsigned long long)n_any);
        atomicAdd(&g_prune_stats[3], (unsigned long long)n_seen);
        atomicAdd(&g_prune_stats[4], (unsigned long long)n_dead);
        if (dbg & (4096 | 65536)) {
            const int b = (dbg & 8192) ? 3 : 0;
            atomicAdd(&g_prune_stats[5], cyc[b]);
            atomicAdd(&g_prune_stats[6], cyc[b + 1]);
            atomicAdd(&g_prune_stats[7], cyc[b + 2]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Round 6: the same algorithm with ONE wave per SIMD.  Workgroup = 4 waves x 4 query tiles (the same 512 query pixels, the same grid, the same LDS ring
// and DMA plan geometry as dense_prune_kernel<8, 0>); a wave holds both planes of FOUR query tiles (224 registers of the 512 a lone wave may have) and
// TWO accumulator sets, so that
//   * every A fragment read from LDS feeds four MFMAs instead of two (half the LDS read traffic per product),
//   * the A fragments of tile t + 1 are requested before tile t's 28 MFMAs are issued (a whole tile of cover for the LDS latency),
//   * the decision of tile t - 1 (max over its 4 x 16 accumulators, bound test, ballot) sits in program order behind tile t's MFMAs and executes in their
//     shadow -- with one wave per SIMD nothing else would hide it (profiles/r06_dense_experiments.txt, section 3: a wave of the 8 x 2 kernel spends ~2 000
//     cycles per tile of which 448 are matrix-pipe time; tools/probe/mfma_probe3.hip prices this shape at 25 ns per MFMA and SIMD against 37.5).
// The pipeline drains at the end of every step (the last tile's LDS buffer is the DMA target of the step after next) and in front of an object switch.
// Same values as the 8 x 2 kernel (a pair's exact value does not depend on what else was evaluated; a bound that is one tile stale only rescoring more).
#ifndef AOC_DENSE_Q4
#define AOC_DENSE_Q4 0           /* compile-time default of the kernel choice; the kernel itself is only compiled with -DAOC_DEV or -DAOC_DENSE_Q4=1 */
#endif
#if defined(AOC_DEV) || AOC_DENSE_Q4
constexpr int Q4_NW = 4, Q4_NQ = 4;
#ifndef AOC_Q4_DBG
#define AOC_Q4_DBG 0           /* timing experiments (tools/build_variant.sh ... "-DAOC_Q4_DBG=n"): 2 no rescoring, 4 no row DMA, 8 no step barrier -- WRONG results */
#endif
__host__ __device__ constexpr int q4_lds_bytes() { return sp_bnd_off(SP_NB) + Q4_NW * Q4_NQ * 256; }

__global__ __launch_bounds__(Q4_NW * 64, 1) void dense_prune_q4_kernel(const uint4 *__restrict__ qrec, const float *__restrict__ q2, int64_t m,
                                                                      const uint4 *__restrict__ prec, const int32_t *__restrict__ tile_rows,
                                                                      const int32_t *__restrict__ tile_obj, const int32_t *__restrict__ n_tiles_ptr,
                                                                      const int32_t *__restrict__ gate, const uint32_t *__restrict__ pmax_bits,
                                                                      int n_obj, uint32_t *__restrict__ gbest, int q_tiled) {
    if (*gate) return;
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    constexpr int NW = Q4_NW, NQ = Q4_NQ, NB = SP_NB;
    constexpr int SP_CHUNK_BYTES = sp_chunk_bytes(NB), SP_IDS_OFF = sp_ids_off(NB), SP_OBJ_OFF = sp_obj_off(NB), SP_BND_OFF = sp_bnd_off(NB);
    constexpr int DMA_PER_WAVE = SP_CHUNK_BYTES / 1024 / NW;          // 14 wave-wide 1 KiB transfers per wave and chunk
    constexpr int W_ID0 = 2, W_ID1 = 3, W_OBJ = 1;
    constexpr int TILE_BYTES = SP_TILE * SP_REC * 16;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds4);
    const char *lds_bytes = reinterpret_cast<const char *>(lds4);

    int bx, by;                                                       // XCD-aware block -> (query block, tile split) map, as in dense_prune_kernel
    {
        const int nb = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3, q = nb >> 3, r = nb & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        by = v / gridDim.x;
        bx = v - by * gridDim.x;
    }
    const int n_tiles = *n_tiles_ptr;
    const int ns = gridDim.y;
    if (by >= n_tiles) return;
    const int n_mine = (n_tiles - by + ns - 1) / ns;
    const int n_chunks = (n_mine + NB - 1) / NB;

    const int lane = aoc_lane(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 31, h = lane >> 5;
    const int64_t wave_row0 = (int64_t)bx * (NW * NQ * 32) + (int64_t)wave * (NQ * 32);

    const float pmax = sqrtf(__uint_as_float(*pmax_bits)) * 1.001f;
    const float plmax = __uint_as_float(*(pmax_bits - 2));
    f16x8 bh[NQ][SP_KS], bl[NQ][SP_KS];
    float eps[NQ];
    bool valid[NQ];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
        const int64_t row = wave_row0 + iq * 32 + col;
        valid[iq] = row < m;
        const int64_t rr = valid[iq] ? row : 0;
        const uint4 *r = q_tiled ? qrec + (size_t)(rr >> 5) * (2 * SP_KS * 64) + h * 32 + (rr & 31) : qrec + (size_t)rr * SP_REC + h;
        const int ks_step = q_tiled ? 64 : 2, plane_step = q_tiled ? SP_KS * 64 : SP_HALF;
#pragma unroll
        for (int ks = 0; ks < SP_KS; ++ks) {
            uint4 u = r[ks * ks_step], v = r[plane_step + ks * ks_step];
            if (!valid[iq]) { u = make_uint4(0, 0, 0, 0); v = make_uint4(0, 0, 0, 0); }
            bh[iq][ks] = __builtin_bit_cast(f16x8, u);
            bl[iq][ks] = __builtin_bit_cast(f16x8, v);
        }
        const float ql_own = (float)bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 3];
        if (h == 0) {
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 3] = (_Float16)0.0f;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16] = (_Float16)SP_QCONST;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 1] = (_Float16)SP_QCONST;
            bh[iq][SP_KS - 1][SP_NORM_SLOT % 16 + 2] = (_Float16)SP_QCONST;
        } else {
            bh[iq][SP_KS - 1][0] = (_Float16)0.0f;                   // (the rest-norm slots of a development build's records)
            bh[iq][SP_KS - 1][1] = (_Float16)0.0f;
        }
        const float qn = valid[iq] ? sqrtf(q2[row]) : 0.0f;
        const float ql = valid[iq] ? __shfl(ql_own, col) : 0.0f;
        eps[iq] = 1026.0f * (qn * plmax + ql * pmax) + 8.0f * pmax * pmax + 8.0f;
    }

    uint32_t dma_plan[DMA_PER_WAVE];
#pragma unroll
    for (int k = 0; k < DMA_PER_WAVE; ++k) {
        const int j = (wave * DMA_PER_WAVE + k) * 64 + lane;
        const int r = j / SP_REC, pos = j - r * SP_REC;
        dma_plan[k] = ((uint32_t)r << 16) | (uint32_t)((pos ^ ((r >> 3) & 3)) * 16);
    }
    const char *prec_bytes = reinterpret_cast<const char *>(prec);
    auto dma_meta = [&](int chunk) {
        const int i0 = chunk * NB;
        if (wave == W_ID0 || wave == W_ID1) {
            const int i = i0 + (wave == W_ID1 ? 2 : 0) + (lane >> 5);
            const int t = min(by + i * ns, n_tiles);
            glds4(tile_rows + (size_t)t * SP_TILE + (lane & 31), lds_base + SP_IDS_OFF + (chunk & 1) * 512 + (wave == W_ID1 ? 256 : 0));
        }
        if (wave == W_OBJ) glds4(tile_obj + min(by + (i0 + (lane & 3)) * ns, n_tiles), lds_base + SP_OBJ_OFF + (chunk & 3) * 256);
    };
    auto dma_rows = [&](int chunk) {
        const int32_t *ids = reinterpret_cast<const int32_t *>(lds_bytes + SP_IDS_OFF + (chunk & 1) * 512);
        const uint32_t dst = lds_base + (uint32_t)(chunk % SP_NBUF) * SP_CHUNK_BYTES + (uint32_t)(wave * DMA_PER_WAVE) * 1024u;
        int id[DMA_PER_WAVE];
#pragma unroll
        for (int k = 0; k < DMA_PER_WAVE; ++k) id[k] = ids[dma_plan[k] >> 16];
#pragma unroll
        for (int k = 0; k < DMA_PER_WAVE; ++k)
            glds16(prec_bytes + (size_t)(uint32_t)max(id[k], 0) * (SP_REC * 16) + (dma_plan[k] & 0xffffu), dst + (uint32_t)k * 1024u);
    };

    float best[NQ], shared[NQ];
    uint32_t grow[NQ];
#pragma unroll
    for (int iq = 0; iq < NQ; ++iq) {
        grow[iq] = valid[iq] ? (uint32_t)(wave_row0 + iq * 32 + col) * (uint32_t)n_obj : 0u;
        best[iq] = INFINITY;
        shared[iq] = INFINITY;
    }
    int cur = -1;
    unsigned n_rescored = 0, n_seen = 0;
    auto dma_bound = [&]() {
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gbest + grow[iq] + max(cur, 0)), "s"(lds_base + SP_BND_OFF + (uint32_t)(wave * NQ + iq) * 256u) : "memory");
        }
    };
    auto switch_object = [&](int o) {
        cur = o;
        uint32_t u[NQ];
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq) u[iq] = load_relaxed(gbest + grow[iq] + cur);
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq) {
            best[iq] = valid[iq] ? -INFINITY : INFINITY;
            shared[iq] = valid[iq] ? ord_dec(u[iq]) : INFINITY;
        }
    };

    dma_meta(0);
    dma_meta(1);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();
    dma_rows(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();

    const int xs = (col >> 3) & 3;
    const uint32_t row_off = (uint32_t)col * (SP_REC * 16);
    const uint32_t sw0 = row_off + (uint32_t)((h ^ xs) * 16), sw2 = row_off + (uint32_t)(((2 + h) ^ xs) * 16);
    auto frag = [&](const char *tile_base, int e) -> f16x8 {
        const uint32_t off = ((e & 2) ? sw2 : sw0) + (uint32_t)((e & ~3) * 16);
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(tile_base + off));
    };
    auto ks_of = [&](int kk) { return (kk + SP_KS - 1) % SP_KS; };
    // BOTH planes of a reference tile travel to registers a tile ahead: a rescoring then needs no LDS round trip of its own (with one wave per SIMD nobody
    // would hide it: profiles/r06_dense_experiments.txt, section 5) -- 14 reads per 28 coarse MFMAs, the LDS traffic per MFMA of the 8 x 2 kernel
    auto load_frags = [&](f16x8 (&af)[SP_KS], f16x8 (&al)[SP_KS], const char *tile_base) {
#pragma unroll
        for (int kk = 0; kk < SP_KS; ++kk) af[kk] = frag(tile_base, 2 * ks_of(kk));
#pragma unroll
        for (int kk = 0; kk < SP_KS; ++kk) al[kk] = frag(tile_base, 2 * SP_KS + 2 * ks_of(kk));
    };
    // one chain of 7 dependent MFMAs: the hi x hi pass of the reference tile against query tile IQ
    auto chain = [&](f32x16 &acc, const f16x8 (&af)[SP_KS], auto iq_c) {
        constexpr int IQ = decltype(iq_c)::value;
#pragma unroll
        for (int kk = 0; kk < SP_KS; ++kk)
            acc = kk == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk], bh[IQ][ks_of(kk)], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk], bh[IQ][ks_of(kk)], acc, 0, 0, 0);
    };
    // settle one (reference tile, query tile) pair: the bound test on its coarse values and, if it may hold a new maximum, the cross terms onto the same
    // accumulator and the exact maximum.  Program-ordered BEHIND the next chain's MFMAs: the test executes in their shadow.
    auto settle = [&](f32x16 &acc, auto iq_c, const f16x8 (&ah)[SP_KS], const f16x8 (&al)[SP_KS]) {
        constexpr int IQ = decltype(iq_c)::value;
        const float cm = max16(acc);
        if (__builtin_amdgcn_ballot_w64(cm + eps[IQ] >= __builtin_fmaxf(best[IQ], shared[IQ])) == 0ull) return;
        n_rescored += 1;
        if (AOC_Q4_DBG & 2) return;                                  // (the count keeps the decision alive)
#pragma unroll
        for (int kk = 0; kk < SP_KS; ++kk) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[IQ][ks_of(kk)], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[IQ][ks_of(kk)], acc, 0, 0, 0);
        }
        const float ex = max16(acc);
        if (ex > best[IQ]) {
            best[IQ] = ex;
            if (ex > shared[IQ]) atomicMax(gbest + grow[IQ] + cur, ord_enc(ex));
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    f16x8 afA[SP_KS], afB[SP_KS], alA[SP_KS], alB[SP_KS];
    f32x16 accX, accY;                                               // the chains of the query tiles 0 / 2 and 1 / 3 (ping-pong)
    for (int s = 0; s < n_chunks; ++s) {
        const int4 objs = *reinterpret_cast<const int4 *>(lds_bytes + SP_OBJ_OFF + (s & 3) * 256);
        const int bound_obj = cur;
        dma_bound();
        dma_meta(s + 2);
        if (!(AOC_Q4_DBG & 4)) dma_rows(s + 1);

        const char *chunk_base = lds_bytes + (s % SP_NBUF) * SP_CHUNK_BYTES;
        const int n_here = min(NB, n_mine - s * NB);
        const uint32_t objs_packed = (uint32_t)__builtin_amdgcn_readfirstlane((objs.x & 0xff) | ((objs.y & 0xff) << 8) | ((objs.z & 0xff) << 16) | ((objs.w & 0xff) << 24));
        load_frags(afA, alA, chunk_base);
        bool pending = false;                                        // accY still holds (previous tile, query tile 3), unsettled
        // tiles 0 and 2 read their fragments from set A, tiles 1 and 3 from set B (fully unrolled: register arrays)
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            if (t < n_here) {
                const char *tile_base = chunk_base + t * TILE_BYTES;
                const int o = (int)((objs_packed >> (8 * t)) & 0xffu);
                f16x8(&af_cur)[SP_KS] = (t & 1) ? afB : afA;
                f16x8(&al_cur)[SP_KS] = (t & 1) ? alB : alA;
                f16x8(&af_prev)[SP_KS] = (t & 1) ? afA : afB;         // (= the next tile's: requested only after the previous tile's last pair is settled)
                f16x8(&al_prev)[SP_KS] = (t & 1) ? alA : alB;
                if (pending) {
                    if (o != cur) {                                  // an object switch: the previous tile is settled under ITS object first
                        settle(accY, I3{}, af_prev, al_prev);
                        pending = false;
                    }
                }
                if (o != cur) switch_object(o);
                chain(accX, af_cur, I0{});
                if (pending) settle(accY, I3{}, af_prev, al_prev);
                if (t + 1 < n_here) load_frags(af_prev, al_prev, tile_base + TILE_BYTES);      // the other set is free now: the next tile's fragments
                chain(accY, af_cur, I1{});
                settle(accX, I0{}, af_cur, al_cur);
                chain(accX, af_cur, I2{});
                settle(accY, I1{}, af_cur, al_cur);
                chain(accY, af_cur, I3{});
                settle(accX, I2{}, af_cur, al_cur);
                n_seen += 1;
                pending = true;
            }
        }
        // drain: the last pair of the step (its LDS buffer is overwritten by the DMA of the step after next)
        if (pending) {
            if (n_here & 1) settle(accY, I3{}, afA, alA);
            else settle(accY, I3{}, afB, alB);
        }

        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
        uint32_t seen[NQ];
#pragma unroll
        for (int iq = 0; iq < NQ; ++iq)
            seen[iq] = *reinterpret_cast<const volatile uint32_t *>(lds_bytes + SP_BND_OFF + (wave * NQ + iq) * 256 + lane * 4);
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
        if (!(AOC_Q4_DBG & 8)) __builtin_amdgcn_s_barrier();
        if (bound_obj == cur && cur >= 0) {
#pragma unroll
            for (int iq = 0; iq < NQ; ++iq)
                if (valid[iq]) shared[iq] = __builtin_fmaxf(shared[iq], ord_dec(seen[iq]));
        }
    }
    if (lane == 0) {
        atomicAdd(&g_prune_stats[0], (unsigned long long)n_seen * NQ);
        atomicAdd(&g_prune_stats[1], (unsigned long long)n_rescored);
        atomicAdd(&g_prune_stats[3], (unsigned long long)n_seen);
    }
}
#endif  // AOC_DEV || AOC_DENSE_Q4

// ------------------------------------------------------------------------------------------
// Seeds of the shared bounds (round 6).  The pruning only bites once gbest[pixel][object] holds a good exact value, and until then almost every pair
// is rescored (at R = 1 a quarter of all pairs; profiles/r06_dense_experiments.txt, section 6).  A video's best match of query pixel i is, more often than
// not, the SAME pixel of the newest reference frame -- pool row n - m + i when the pool is whole frames of m rows.  This kernel evaluates that one pair per
// query pixel in plain fp32 and publishes  seed = 2^20 (q.r - |r|^2 / 2) - margin  for the object the row is labelled with, BEFORE the matrix kernel
// starts.  Whatever row that is, it is a real (pixel, kept row) pair of that object, and the margin keeps the seed at or below the value the matrix kernel
// itself computes for that pair, in the WORST case of every rounding: with M = 2^20 (|q||r| + |r|^2 / 2) >= every partial sum on either side,
//   the matrix kernel's 21 MFMAs round at most 21 x 16 times by half an ulp <= 2^-24 M each                          336 x 2^-24 M
//   this kernel's two fp32 dot products of C <= 100 terms (sequential adds) and the final subtraction                  ~210 x 2^-24 M
//   the ql.rl product the three-product value leaves out: <= 2^20 (2^-11 |q|)(2^-11 |r|)                              0.25 |q||r|
// margin = 4e-5 M + |q||r| + 16 (4e-5 > 546 x 2^-24 = 3.3e-5; tests/test_host_logic.py::test_dense_seed_margin_covers_the_worst_case replays the
// arithmetic in numpy).  Hence  seed <= exact(seed pair) <= exact(best pair) <= coarse(best pair) + eps : the best pair is still evaluated and the
// published maximum is the same number as without seeds -- the seeds only spare pairs that could never have held it.  (~5e-4 in squared-distance units
// on the bench's embeddings, below the kernel's own rescoring margin eps = 1.4e-3.)
#ifndef AOC_DENSE_SEED
#define AOC_DENSE_SEED 1
#endif
#ifndef AOC_DENSE_SEED_FRAMES
#define AOC_DENSE_SEED_FRAMES 1
#endif
__global__ __launch_bounds__(256) void dense_seed_kernel(const float *__restrict__ query, const float *__restrict__ q2, int64_t m, int C,
                                                          const float *__restrict__ pool, int64_t n, const uint32_t *__restrict__ right_bits, int n_obj,
                                                          const int32_t *__restrict__ gate, uint32_t *__restrict__ gbest) {
    if (*gate) return;
    // eight lanes per query pixel, a wave = eight consecutive pixels: their rows (and the candidate rows, consecutive as well) are read as whole 128-byte
    // lines; partial sums meet over three shuffles.  (One thread per pixel walked its two 400-byte rows alone: 37 us per launch on the frame's critical path.)
    const int sub = threadIdx.x & 7;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool live = i < m;
    const int64_t ii = live ? i : m - 1;
    const uint32_t mask = (n_obj >= 32) ? 0xffffffffu : ((1u << n_obj) - 1u);
    const float4 *qr = reinterpret_cast<const float4 *>(query + (size_t)ii * C);
    const float qq = q2[ii];
    const int c4 = C >> 2;
    // the same pixel of the newest AOC_DENSE_SEED_FRAMES reference frames: where an object has moved over the pixel, the older frames seed another object
    for (int f = 0; f < AOC_DENSE_SEED_FRAMES && (int64_t)(f + 1) * m <= n; ++f) {
        const int64_t j = n - (int64_t)(f + 1) * m + ii;
        const uint32_t right = right_bits[j];
        const bool ok = live && (right & AOC_ROW_KEPT_BIT) && __popc(right & mask) == 1;      // (uniform over the pixel's eight lanes)
        const float4 *rr = reinterpret_cast<const float4 *>(pool + (size_t)j * C);
        float dot = 0.0f, r2 = 0.0f;
        for (int t = sub; t < c4; t += 8) {
            const float4 a = qr[t], b = rr[t];
            dot = dot + a.x * b.x; dot = dot + a.y * b.y; dot = dot + a.z * b.z; dot = dot + a.w * b.w;
            r2 = r2 + b.x * b.x; r2 = r2 + b.y * b.y; r2 = r2 + b.z * b.z; r2 = r2 + b.w * b.w;
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            dot = dot + __shfl_xor(dot, d);
            r2 = r2 + __shfl_xor(r2, d);
        }
        if (ok && sub == 0) {
            const int o = __ffs((int)(right & mask)) - 1;
            const float value = 1048576.0f * (dot - 0.5f * r2);
            const float qr_n = sqrtf(qq * r2);
            const float seed = value - (4e-5f * (1048576.0f * (qr_n + 0.5f * r2)) + qr_n + 16.0f);
            atomicMax(gbest + (size_t)ii * n_obj + o, ord_enc(seed));
        }
    }
}

// out[i,o] = f( min(own_o, 5e4 + min_{o' != o} own_o') ), own_o = |q_i|^2 - 2^-19 max-accumulator (+inf: no pixel of o)
// (every word of gbest it reads is zeroed again: a caller that keeps the workspace across the frames of one pool state -- aoc_dense_match_min_split_cached
// -- starts the next frame without a memset)
__global__ __launch_bounds__(256) void dense_split_finalize_kernel(uint32_t *__restrict__ gbest, int64_t m, int n_obj,
                                                                    const int32_t *__restrict__ counts, const int32_t *__restrict__ gate,
                                                                    const float *__restrict__ q2, const float *__restrict__ obj_bias,
                                                                    float *__restrict__ out, int64_t pstride, int64_t ostride, int transform) {
    if (*gate) return;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    const float qq = q2[row];
    float own[16];
    int n_kept = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        own[o] = INFINITY;
        if (o < n_obj && counts[o] > 0) {
            own[o] = qq + SP_UNSCALE * ord_dec(gbest[(size_t)row * n_obj + o]);
            gbest[(size_t)row * n_obj + o] = 0u;
            n_kept += counts[o];
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
            if (n_kept == 0) v = transform ? 1.0f : INFINITY;            // AEM:796-797
            else if (transform) v = aoc_proto_transform(v, obj_bias ? obj_bias[o] : 0.0f);
            out[row * pstride + o * ostride] = v;
        }
    }
}

// reused plan (aoc_dense_match_min_split_cached): the plan's gate already holds "a kept row is not one-hot | overflow at plan time"; the
// overflow word is sticky and may have been raised by a later query
__global__ void split_gate_refresh_kernel(const int32_t *__restrict__ overflow, int32_t *__restrict__ gate) {
    if (*overflow) atomicOr(gate, 1);
}

inline int split_waves() {
    // developer switch AOC_DENSE_WAVES=4: workgroups of 4 waves (256 query pixels), one wave per SIMD -- half of every CU's register file stays
    // free for the other streams' kernels, and no CU mask is needed; the default (8) fills the CUs it runs on
    static const int nw = AOC_DEV_ENV_INT("AOC_DENSE_WAVES", 8) == 4 ? 4 : 8;
    return nw;
}
#if defined(AOC_DEV) || AOC_DENSE_Q4
inline bool split_q4() {
    // dense_prune_q4_kernel instead of dense_prune_kernel<8, 0>: compile-time default AOC_DENSE_Q4, developer switch AOC_DENSE_Q4=0/1 (the workgroup
    // covers the same 512 query pixels, so the grid and the split count are those of the eight-wave kernel: AOC_DENSE_WAVES must stay 8)
    static const bool q4 = AOC_DEV_ENV_INT("AOC_DENSE_Q4", AOC_DENSE_Q4) != 0 && split_waves() == 8;
    return q4;
}
#endif
inline int split_tiles_per_chunk() {
    // developer switch AOC_DENSE_NB=2 (with AOC_DENSE_WAVES=4): chunks of two tiles, 60 KB of LDS per workgroup -> TWO workgroups of four waves per
    // CU, i.e. two waves per SIMD as in the product but from different workgroups: their barriers and their phases are independent
    static const int nb = (AOC_DEV_ENV_INT("AOC_DENSE_NB", SP_NB) == 2 && split_waves() == 4) ? 2 : SP_NB;
    return nb;
}
inline int split_nsplit(int64_t m) {
    const int64_t rpb = (int64_t)split_waves() * SP_NQ * 32;
    const int64_t row_blocks = (m + rpb - 1) / rpb;
    const int wg_per_cu = split_tiles_per_chunk() == 2 ? 2 : 1;
    // at most two rounds of workgroups (developer switch AOC_DENSE_ROUNDS): fewer splits share their bounds sooner (in-run launch 1.43 /
    // 1.50 / 1.56 ms at 1 / 2 / 4 rounds), but with one round the other streams' kernels wait for a whole dense launch before a CU
    // comes free: bench 320 / 324 / 320 frames/s
    static const int max_rounds = AOC_DEV_ENV_INT("AOC_DENSE_ROUNDS", 2);
    // CUs the launching stream may use (256 unless the caller runs it under a HIP CU mask and says so)
    const int n_cu = (aoc_stream_cus() > 0 ? aoc_stream_cus() : 256) * wg_per_cu;      // resident workgroups
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

inline int split_ckpt() {
    // checkpoint of dense_prune_kernel after this many of the 7 k-steps; developer switch AOC_DENSE_CKPT = 3 / 4 (development build only).
    // Default 0 = none: measured in round 5 (profiles/r05_dense_experiments.txt) the checkpoint stops 26 % (after 3 k-steps) / 40 % (after 4) of
    // the pairs of the bench's R = 6 pools, same results -- and the kernel is 16 % / 13 % SLOWER: the wave has to wait for its own MFMA results
    // in the middle of every tile, and with two waves per SIMD nothing hides that wait.
    static const int c = AOC_DEV_ENV_INT("AOC_DENSE_CKPT", 0);
    return (c == 3 || c == 4) ? c : 0;
}

struct SplitWs {
    int32_t *gate, *n_tiles, *tile_rows, *tile_obj;
    uint32_t *pmax, *gbest;
    void *fp32_ws;
    size_t fp32_bytes, total;
    int64_t tile_capacity;
};
inline SplitWs split_carve(void *base, int64_t m, int64_t n, int n_obj) {
    SplitWs w;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += aoc_align_up(bytes, 256); return r; };
    w.tile_capacity = (n + SP_TILE - 1) / SP_TILE + n_obj + SP_TILE_SLACK;
    w.gate = reinterpret_cast<int32_t *>(take(16));
    w.n_tiles = w.gate ? w.gate + 2 : nullptr;
    w.pmax = w.gate ? reinterpret_cast<uint32_t *>(w.gate + 3) : nullptr;
    w.tile_rows = reinterpret_cast<int32_t *>(take((size_t)w.tile_capacity * SP_TILE * sizeof(int32_t)));
    w.tile_obj = reinterpret_cast<int32_t *>(take((size_t)w.tile_capacity * sizeof(int32_t)));
    w.gbest = reinterpret_cast<uint32_t *>(take((size_t)m * n_obj * sizeof(uint32_t)));
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
    hipLaunchKernelGGL(split_rows_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), x, n, C,
                       static_cast<uint4 *>(records), sqnorm, overflow_flag);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_split_rows_tiled_bytes(int64_t n, int C) {
    if (n < 0 || aoc_split_record_bytes(C) == 0) return 0;
    return (size_t)((n + 31) / 32) * 32 * SP_REC * 16;
}

int aoc_split_rows_tiled(const float *x, int64_t n, int C, void *records, float *sqnorm, int32_t *overflow_flag, aoc_stream_t stream) {
    if (!x || !records || !overflow_flag || n < 0) return AOC_ERR_INVALID_ARG;
    if (aoc_split_record_bytes(C) == 0) return AOC_ERR_UNSUPPORTED;
    if (n == 0) return AOC_OK;
    const int64_t total = (n + 31) / 32 * 32 * SP_KS;
    hipLaunchKernelGGL(split_rows_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), x, n, C,
                       static_cast<uint4 *>(records), sqnorm, overflow_flag);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

size_t aoc_dense_match_split_workspace_bytes(int64_t m, int64_t n, int n_obj) {
    if (m < 1 || n < 1 || n_obj < 1) return 0;
    return split_carve(nullptr, m, n, n_obj).total;
}

int aoc_dense_match_min_split(const float *query, const void *query_rec, const float *query_sqnorm, int query_rec_tiled, int64_t m, int C, const float *pool,
                              const void *pool_rec, const int32_t *overflow_flag, int64_t n, const uint32_t *right_bits,
                              const uint32_t *wrong_bits, const int32_t *fg_rows, const int32_t *obj_rows, const int32_t *counts,
                              const int32_t *obj_offsets, const float *obj_bias, int n_obj, float *out, int64_t out_pixel_stride,
                              int64_t out_obj_stride, int transform, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    return aoc_dense_match_min_split_cached(query, query_rec, query_sqnorm, query_rec_tiled, m, C, pool, pool_rec, overflow_flag, n, right_bits, wrong_bits,
                                            fg_rows, obj_rows, counts, obj_offsets, obj_bias, n_obj, out, out_pixel_stride, out_obj_stride, transform, workspace,
                                            workspace_bytes, 0, stream);
}

int aoc_dense_match_min_split_cached(const float *query, const void *query_rec, const float *query_sqnorm, int query_rec_tiled, int64_t m, int C, const float *pool,
                                     const void *pool_rec, const int32_t *overflow_flag, int64_t n, const uint32_t *right_bits,
                                     const uint32_t *wrong_bits, const int32_t *fg_rows, const int32_t *obj_rows, const int32_t *counts,
                                     const int32_t *obj_offsets, const float *obj_bias, int n_obj, float *out, int64_t out_pixel_stride,
                                     int64_t out_obj_stride, int transform, void *workspace, size_t workspace_bytes, int reuse_plan, aoc_stream_t stream) {
    if (!query || !query_rec || !query_sqnorm || !pool || !pool_rec || !overflow_flag || !right_bits || !wrong_bits || !fg_rows ||
        !obj_rows || !counts || !obj_offsets || !out || !workspace)
        return AOC_ERR_INVALID_ARG;
    if (m < 1 || n < 1 || n >= (1ll << 31) - 4096 || n_obj < 1) return AOC_ERR_INVALID_ARG;
    if (aoc_split_record_bytes(C) == 0 || n_obj > 16) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_dense_match_split_workspace_bytes(m, n, n_obj)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    const SplitWs w = split_carve(workspace, m, n, n_obj);
    if (reuse_plan) {
        // same pool rows, labels and records as the call that built the plan in this workspace: tile lists, norm maxima and the one-hot check
        // stand; gbest was left clean by that call's finalize kernel
        hipLaunchKernelGGL(split_gate_refresh_kernel, dim3(1), dim3(1), 0, st, overflow_flag, w.gate);
    } else {
        if (hipMemsetAsync(w.gate, 0, 16, st) != hipSuccess) return AOC_ERR_LAUNCH;
        if (hipMemsetAsync(w.gbest, 0, (size_t)m * n_obj * sizeof(uint32_t), st) != hipSuccess) return AOC_ERR_LAUNCH;
        const int64_t plan_threads = w.tile_capacity * SP_TILE > n ? w.tile_capacity * SP_TILE : n;
        hipLaunchKernelGGL(split_plan_kernel, dim3((unsigned)((plan_threads + 255) / 256)), dim3(256), 0, st, obj_rows, counts, obj_offsets, n_obj, n,
                           right_bits, wrong_bits, overflow_flag, static_cast<const uint4 *>(pool_rec), w.tile_capacity, w.tile_rows, w.tile_obj,
                           w.n_tiles, w.gate, w.pmax);
    }
    static const bool seeds = AOC_DEV_ENV_INT("AOC_DENSE_SEED", AOC_DENSE_SEED) != 0;     // developer switch (A / B)
    if (seeds && n >= m)
        hipLaunchKernelGGL(dense_seed_kernel, dim3((unsigned)((m * 8 + 255) / 256)), dim3(256), 0, st, query, query_sqnorm, m, C, pool, n, right_bits, n_obj, w.gate,
                           w.gbest);
    const int ns = split_nsplit(m);
    const int nw = split_waves();
    const int64_t rpb = (int64_t)nw * SP_NQ * 32;
    const dim3 grid((unsigned)((m + rpb - 1) / rpb), ns);
    const int nb = split_tiles_per_chunk();
    const size_t lds = sp_lds_bytes(nw, nb);
#if defined(AOC_DEV) || AOC_DENSE_Q4
    if (split_q4()) {
        // one wave per SIMD, four query tiles per wave (dense_prune_q4_kernel): same grid, same plan, same gbest
        static const bool q4_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_prune_q4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      q4_lds_bytes()) == hipSuccess;
        if (!q4_ok) return AOC_ERR_LAUNCH;
        const AocDenseProbe probe4 = aoc_take_dense_probe();
        if (probe4.start) (void)hipEventRecord(probe4.start, st);
        hipLaunchKernelGGL(dense_prune_q4_kernel, grid, dim3(Q4_NW * 64), q4_lds_bytes(), st, static_cast<const uint4 *>(query_rec), query_sqnorm, m,
                           static_cast<const uint4 *>(pool_rec), w.tile_rows, w.tile_obj, w.n_tiles, w.gate, w.pmax, n_obj, w.gbest, query_rec_tiled ? 1 : 0);
        if (probe4.stop) (void)hipEventRecord(probe4.stop, st);
        hipLaunchKernelGGL(dense_split_finalize_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w.gbest, m, n_obj, counts, w.gate,
                           query_sqnorm, obj_bias, out, out_pixel_stride, out_obj_stride, transform);
        AOC_RETURN_IF_LAUNCH_FAILED();
        return aoc_dense_match_min_gated(query, m, C, pool, fg_rows, counts + n_obj, n, wrong_bits, obj_bias, n_obj, out, out_pixel_stride,
                                         out_obj_stride, transform, w.fp32_ws, w.fp32_bytes, w.gate, stream);
    }
#endif
    const AocDenseProbe probe = aoc_take_dense_probe();
    static const int dbg = AOC_DEV_ENV_INT("AOC_DENSE_DEBUG", 0);       // developer switch: timing experiments only
    const int ckpt = split_ckpt();
    int launched = 0;
#define AOC_DENSE_LAUNCH(NW_, CK_, NB_)                                                                                                                     \
    if (!launched && nw == NW_ && ckpt == CK_ && nb == NB_) {                                                                                               \
        static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_prune_kernel<NW_, CK_, NB_>),                                  \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;                                \
        if (!lds_ok) return AOC_ERR_LAUNCH;                                                                                                                 \
        if (probe.start) (void)hipEventRecord(probe.start, st);                                                                                             \
        hipLaunchKernelGGL((dense_prune_kernel<NW_, CK_, NB_>), grid, dim3(NW_ * 64), lds, st, static_cast<const uint4 *>(query_rec), query_sqnorm, m,           \
                           static_cast<const uint4 *>(pool_rec), w.tile_rows, w.tile_obj, w.n_tiles, w.gate, w.pmax, n_obj, w.gbest, dbg,                   \
                           query_rec_tiled ? 1 : 0);                                                                                                        \
        launched = 1;                                                                                                                                       \
    }
    AOC_DENSE_LAUNCH(8, 0, 4)
#ifdef AOC_DEV
    AOC_DENSE_LAUNCH(8, 3, 4)
    AOC_DENSE_LAUNCH(8, 4, 4)
    AOC_DENSE_LAUNCH(4, 0, 4)
    AOC_DENSE_LAUNCH(4, 3, 4)
    AOC_DENSE_LAUNCH(4, 0, 2)
#endif
#undef AOC_DENSE_LAUNCH
    if (!launched) return AOC_ERR_UNSUPPORTED;
    if (probe.stop) (void)hipEventRecord(probe.stop, st);
    hipLaunchKernelGGL(dense_split_finalize_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w.gbest, m, n_obj, counts, w.gate,
                       query_sqnorm, obj_bias, out, out_pixel_stride, out_obj_stride, transform);
    AOC_RETURN_IF_LAUNCH_FAILED();
    // exact-fp32 kernels: run only when the gate is set
    return aoc_dense_match_min_gated(query, m, C, pool, fg_rows, counts + n_obj, n, wrong_bits, obj_bias, n_obj, out, out_pixel_stride,
                                     out_obj_stride, transform, w.fp32_ws, w.fp32_bytes, w.gate, stream);
}

static std::atomic<int> g_stream_cus{0};
int aoc_set_stream_cus(int n_cus) {
    if (n_cus < 0) return AOC_ERR_INVALID_ARG;
    g_stream_cus.store(n_cus, std::memory_order_relaxed);
    return AOC_OK;
}

// Developer counters of the coarse-then-rescore kernel, summed over all launches since the last reset:
// out[0] (reference tile, query tile) pairs tested, out[1] pairs rescored, out[2] reference tiles with a rescoring, out[3] reference tiles.
int aoc_dense_prune_stats(uint64_t *out4, int reset) {
    if (!out4) return AOC_ERR_INVALID_ARG;
    uint64_t v[8];
    const int rc = aoc_dense_prune_stats_ex(v, reset);
    for (int i = 0; i < 4 && rc == AOC_OK; ++i) out4[i] = v[i];
    return rc;
}

#ifdef AOC_DEV
// development build only (not part of the C ABI of include/aoc_hip.h): the per-workgroup stamps of the last launches with AOC_DENSE_DEBUG bit 32768
int aoc_dev_dense_block_times(uint64_t *out, int n_words) {
    if (!out || n_words < 1 || n_words > 4096 * 6) return AOC_ERR_INVALID_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(aoc_dev_block_times), (size_t)n_words * sizeof(uint64_t)) == hipSuccess ? AOC_OK : AOC_ERR_LAUNCH;
}
#endif

// out8[0..3] as aoc_dense_prune_stats; out8[4] = (reference tile, query tile) pairs that stopped at the checkpoint; out8[5..7] = 0.
int aoc_dense_prune_stats_ex(uint64_t *out8, int reset) {
    if (!out8) return AOC_ERR_INVALID_ARG;
    unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_prune_stats), sizeof(v)) != hipSuccess) return AOC_ERR_LAUNCH;
    for (int i = 0; i < 8; ++i) out8[i] = v[i];
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_prune_stats), z, sizeof(z)) != hipSuccess) return AOC_ERR_LAUNCH;
    }
    return AOC_OK;
}

}  // extern "C"

// the budget of the call in progress on this thread (aoc_frame_enqueue: aoc_frame_desc.stream_cus), else the process-wide setting
static thread_local int t_stream_cus = 0;
int aoc_stream_cus() { return t_stream_cus > 0 ? t_stream_cus : g_stream_cus.load(std::memory_order_relaxed); }
int aoc_stream_cus_scope(int n_cus) {
    const int before = t_stream_cus;
    t_stream_cus = n_cus > 0 ? n_cus : 0;
    return before;
}
