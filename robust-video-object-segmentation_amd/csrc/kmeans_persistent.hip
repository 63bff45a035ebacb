// The 20-iteration k-means chain of one call (scipy.cluster.vq.kmeans2, AEM:276) as ONE persistent launch: every Lloyd iteration is five
// phases of the same resident grid, separated by grid barriers --
//   A  assignment on the fp32 matrix pipe (bit-identical to scipy's vq, see labels_kmeans.hip) + per-(64-row mini-block, cluster)
//      any-order sums of |x| and their running prefix inside the workgroup's block range;
//   X  prefix of the workgroups' range sums along each segment: with A's prefixes, the PREDICTION of the exact sequential sum in
//      front of every chunk;
//   B  fold: every chunk (the members of one cluster inside one mini-block, in row order) becomes a record per feature -- integer
//      increments in the predicted binade(s), the members around a predicted binade crossing kept as literals (km_exact_core.h);
//   C  merge: consecutive pure-integer records of a part (64 mini-blocks) collapse into runs;
//   S  stitch: one wave per (segment, cluster, 64 features) adds the first KP_HEAD members literally (the sum crosses a binade at
//      every doubling there), then walks parts -> runs -> records with the EXACT float state, verifies every assumption and replaces
//      what does not verify by the literal additions of that chunk's rows.
// The result equals scipy's sequential float32 sums bit for bit for ANY input; predictions only decide how much of it runs in
// parallel.  Chunks are tied to row blocks, not to member ranks: no rank / scan / scatter passes, no member lists.
// Geometry: 512 threads = 8 waves per workgroup, one workgroup per CU (two waves per SIMD).  A 256-row block is four mini-blocks of
// 64 rows, staged row-major in LDS (4 x 25.6 KB); a PAIR of waves shares a mini-block: wave 2q + h works on its features 64 h ..
// 64 h + 63 (lanes = features) and on 16-row tiles 2h, 2h + 1 of the assignment.  The rows of the next block travel in registers
// while the current one is worked on.
#include "aoc_common.h"
#include "km_exact_core.h"

#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int KP_C = 100;                 // embedding width served by this path
constexpr int KP_TM = 25;                 // float4 pieces per row
constexpr int KP_SEG_MAX = 128;
constexpr int KP_REC = 6;                 // words per (chunk, feature): hdr, A0, B0, literal | literal offset, run hdr, run R0
constexpr int KP_LITCAP = 8192;           // literal pool per workgroup (floats)
constexpr int KP_GMAX = 1024;
constexpr int KP_LIT = KX_MAX_LIT;        // literals a lane can hold per chunk
constexpr int KP_ROWF = 64 * KP_C;        // floats of a staged mini-block
constexpr int KP_EV = 2;                  // records per (part, feature) the stitch fetches one part ahead
constexpr int KP_HEAD = 512;              // members of every cluster that the stitch adds literally (chunks that start below it)
constexpr int KP_HEADCAP = KP_HEAD + 64;  // row ids a wave lists for its head
constexpr int KP_THREADS = 512;
constexpr int KP_WAVES = KP_THREADS / 64;
constexpr uint32_t KP_HDR_HEAD = 1u << 28;   // record header flag: the chunk belongs to the literal head (kind KX_UNSAFE)
constexpr long long KP_TIMEOUT_TICKS = 300000000ll;   // 3 s of the 100 MHz wall clock: a barrier that does not complete poisons the output instead of hanging

struct KpBar {
    uint32_t count;
    uint32_t error;
    uint32_t pad[62];
};

struct KpArgs {
    const float *pool;
    const int32_t *rows, *seg_off, *seg_k;
    int n_seg, kmax, iters, grid, prof;
    float *centroids;
    int32_t *labels, *cluster_counts;
    float *rownorm;
    unsigned long long *pres;     // [mini-block] bit kk: cluster kk has members there
    float *headbuf;               // [segment][kmax][KP_HEADCAP][C] the first members of every cluster, in row order (written by the fold phase)
    int32_t *ccnt;                // [slot] members of the chunk
    float *bsum;                  // [slot][C] any-order sum of |x| over the chunk
    uint32_t *rec;                // [slot][KP_REC][C]
    float *PB;                    // [block][kmax][C] prefix of bsum over the earlier blocks of this workgroup's range (same segment)
    int32_t *PBC;                 // [block][kmax]
    float *T;                     // [workgroup][kmax][C] sum over the workgroup's blocks of its last segment
    int32_t *TC;
    float *WB;                    // [workgroup][kmax][C] sum over the EARLIER workgroups' blocks of this workgroup's first segment (phase X)
    int32_t *WBC;
    float *litpool;               // [workgroup][KP_LITCAP]
    uint32_t *part_post;          // [part][kmax][2][C] last run of the part
    unsigned long long *part_np;  // [part][kmax][C] positions of the records the stitch has to look at
    int32_t *part_cnt;            // [part][kmax] members
    KpBar *bar;
};

struct KpTables {
    int32_t seg_off[KP_SEG_MAX + 1];
    int32_t seg_k[KP_SEG_MAX];
    int32_t blk_base[KP_SEG_MAX + 1];     // 256-row blocks in front of the segment
    int32_t part_base[KP_SEG_MAX + 1];    // parts (64 mini-blocks = 16 blocks) in front of the segment
    int32_t error;
    int32_t litcnt;
    int32_t pad[2];
};

// developer counters (aoc_kmeans_chain_profile): what one workgroup spends where, in ticks of the 100 MHz wall clock
__device__ unsigned long long g_kp_prof[32];
__device__ unsigned int g_kp_wgprof[KP_GMAX * 16];      // per workgroup: ticks per phase slot, summed over the iterations of the LAST chain
#define KP_TICK(slot)                                                         \
    do {                                                                      \
        if (prof && threadIdx.x == 0) {                                       \
            const long long now_ = wall_clock64();                            \
            wgsec[slot] += (unsigned)(now_ - tprev);                          \
            if (wg == a.prof - 1) g_kp_prof[slot] += (unsigned long long)(now_ - tprev); \
            tprev = now_;                                                     \
        }                                                                     \
    } while (0)
// finer developer timing inside a phase: accumulators live in registers of the profiled thread, flushed once per phase
#define KP_SEC(i)                                                  \
    do {                                                           \
        if (pf) {                                                  \
            const long long now_ = wall_clock64();                 \
            sec[i] += (unsigned)(now_ - tsec);                     \
            tsec = now_;                                           \
        }                                                          \
    } while (0)

__device__ __forceinline__ unsigned long long kp_below(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
__device__ __forceinline__ unsigned long long kp_readlane64(unsigned long long v, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, l), hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------ grid barrier
// One monotonic counter; release fence by the arriving lane, relaxed polling, one acquire fence after the match (guide: §6 G16).
__device__ __forceinline__ bool kp_grid_barrier(KpBar *bar, KpTables *tab, uint32_t &epoch, int grid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t target = epoch * (uint32_t)grid;
        uint32_t v = __hip_atomic_fetch_add(&bar->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        const long long t0 = wall_clock64();
        int err = 0;
        while (v < target) {
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(&bar->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_load(&bar->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { err = 1; break; }
            if (wall_clock64() - t0 > KP_TIMEOUT_TICKS) {
                __hip_atomic_store(&bar->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                err = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        tab->error = err;
    }
    __syncthreads();
    return tab->error == 0;
}

// sequential |x|^2 of a row: multiply, then add (scipy's order)
__device__ __forceinline__ float kp_sqnorm_row(const float4 *__restrict__ xr) {
    float xs = 0.0f;
#pragma unroll 5
    for (int t = 0; t < KP_TM; ++t) {
        const float4 v = xr[t];
        float p0 = v.x * v.x; xs = xs + p0;
        float p1 = v.y * v.y; xs = xs + p1;
        float p2 = v.z * v.z; xs = xs + p2;
        float p3 = v.w * v.w; xs = xs + p3;
    }
    return xs;
}

// The 64 rows of a mini-block, staged row-major in LDS (64 x 100 floats = 1600 float4) by the PAIR of waves that works on it: lane l of
// wave half h carries float4 number i * 128 + 64 h + l, i = 0..12.  Row-major serves every consumer without bank conflicts: lanes =
// features read consecutive words, and the MFMA operand (lane (j, g) reads x[16 tile + j][4 t + g]) has 100 j + g distinct modulo 64
// for j < 16, g < 4.  (13 NAMED float4 values: an array that lives across the block loop is left in scratch memory by hipcc.)
#define KP_R13(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12)
#define KP_ROW_DECL(i) float4 rw##i = make_float4(0.f, 0.f, 0.f, 0.f);
#define KP_ROW_ISSUE1(i)                                                                       \
    {                                                                                          \
        const int idx_ = min(i * 128 + 64 * h + lane, 64 * KP_TM - 1);                         \
        const int row_ = idx_ / KP_TM, t_ = idx_ - row_ * KP_TM;                               \
        const int id_ = __shfl(my_id, row_);                                                   \
        rw##i = reinterpret_cast<const float4 *>(a.pool + (size_t)id_ * KP_C)[t_];             \
    }
#define KP_ROW_COMMIT1(i)                                                                      \
    if (i * 128 + 64 * h + lane < 64 * KP_TM) reinterpret_cast<float4 *>(wrow)[i * 128 + 64 * h + lane] = rw##i;
#define KP_ROWS_DECL() KP_R13(KP_ROW_DECL)
#define KP_ROWS_ISSUE() KP_R13(KP_ROW_ISSUE1)          /* uses a.pool, my_id, lane, h */
#define KP_ROWS_COMMIT() KP_R13(KP_ROW_COMMIT1)        /* uses wrow, lane, h */

// Rows of a mini-block ordered by (cluster, row): lane i of the result holds the row (0..63) of the i-th member in that order, so that the
// members of a cluster are a counted loop over lanes (v_readlane) instead of a scan over a 64-bit mask.  lab: cluster of row `lane`
// (-1: none), pm: clusters present.  Rows without a cluster end up behind all members.
__device__ __forceinline__ int kp_order_rows(int lab, unsigned long long pm, int lane) {
    int pos = 63, off = 0;
    unsigned long long rem = pm;
    unsigned long long none = __ballot(lab < 0);
    while (rem) {
        const int kk = __builtin_ctzll(rem);
        rem &= rem - 1;
        const unsigned long long m = __ballot(lab == kk);
        if (lab == kk) pos = off + __popcll(m & kp_below(lane));
        off += __popcll(m);
    }
    if (lab < 0) pos = off + __popcll(none & kp_below(lane));
    return __builtin_amdgcn_ds_permute(pos << 2, lane);
}

__device__ __forceinline__ int kp_locate(const KpTables *tab, int blk, int s) {
    while (blk >= tab->blk_base[s + 1]) ++s;
    return s;
}

// ------------------------------------------------------------------------------------------ phase A
template <int KT>
__device__ __forceinline__ void kp_phase_assign(const KpArgs &a, KpTables *tab, float *lds, int blk_begin, int blk_end, int wg) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int q = wave >> 1, h = wave & 1;
    const int j = lane & 15, g = lane >> 4;
    const int kmax = a.kmax;
    float *lc = lds;                                             // [KT*16][C] code book, row-major (rows >= k are zero)
    float *lcn = lc + (size_t)KT * 16 * KP_C;                    // [KT*16] |c|^2 (+inf beyond k)
    float *wrow = lcn + KT * 16 + (size_t)q * KP_ROWF;           // this pair's rows
    float *run = lcn + KT * 16 + (size_t)4 * KP_ROWF;            // [kmax][C] prefix of bsum over this workgroup's earlier blocks of the segment
    float *blks = run + (size_t)kmax * KP_C;                     // [kmax][C] sums of the current block
    int32_t *runc = reinterpret_cast<int32_t *>(blks + (size_t)kmax * KP_C);   // [kmax]
    int32_t *blkc = runc + kmax;                                 // [kmax]
    int32_t *labx = blkc + kmax + (size_t)q * 64;                // [4][64] labels of the mini-blocks (exchange between the waves of a pair)

    const bool pf = a.prof != 0 && wg == a.prof - 1 && threadIdx.x == 0;
    unsigned sec[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    long long tsec = pf ? wall_clock64() : 0ll;
    for (int i = threadIdx.x; i < kmax * KP_C; i += KP_THREADS) { run[i] = 0.0f; blks[i] = 0.0f; }
    if ((int)threadIdx.x < kmax) { runc[threadIdx.x] = 0; blkc[threadIdx.x] = 0; }
    __syncthreads();
    if (blk_begin < blk_end) {
        float ca[KT][KP_TM];
        float cn[KT][4];
        int cur_seg = -1, k = 0, beg = 0, len = 0;
        int s = kp_locate(tab, blk_begin, 0);
        KP_ROWS_DECL()
        int my_id, id_next = 0;
        float my_xs, xs_next = 0.0f;
        {
            const int ibeg = tab->seg_off[s], ilen = tab->seg_off[s + 1] - ibeg;
            const int p = max(min((blk_begin - tab->blk_base[s]) * 256 + q * 64 + lane, ilen - 1), 0);
            my_id = a.rows[ibeg + p];
            my_xs = a.rownorm[ibeg + p];
            KP_ROWS_ISSUE()
        }
        const int f = 64 * h + lane;
        const bool fvalid = f < KP_C;
        const int fc = fvalid ? f : KP_C - 1;
        for (int blk = blk_begin; blk < blk_end; ++blk) {
            s = kp_locate(tab, blk, s);
            const int bx = blk - tab->blk_base[s];
            const int ibeg = tab->seg_off[s], ilen = tab->seg_off[s + 1] - ibeg;
            const int mb_row0 = bx * 256 + q * 64;
            if (blk + 1 < blk_end) {                               // ids of the next block: one round trip ahead of its rows
                const int s1 = kp_locate(tab, blk + 1, s);
                const int b1 = tab->seg_off[s1], l1n = tab->seg_off[s1 + 1] - b1;
                const int p1 = max(min((blk + 1 - tab->blk_base[s1]) * 256 + q * 64 + lane, l1n - 1), 0);
                id_next = a.rows[b1 + p1];
                xs_next = a.rownorm[b1 + p1];
            }
            if (s != cur_seg) {
                cur_seg = s;
                k = tab->seg_k[s];
                beg = ibeg;
                len = ilen;
                for (int i = threadIdx.x; i < kmax * KP_C; i += KP_THREADS) run[i] = 0.0f;      // (the previous block ended behind a barrier)
                if ((int)threadIdx.x < kmax) runc[threadIdx.x] = 0;
                const float *csrc = a.centroids + (size_t)s * kmax * KP_C;
                for (int idx = threadIdx.x; idx < KT * 16 * KP_TM; idx += KP_THREADS) {
                    const int cc = idx / KP_TM, t = idx - cc * KP_TM;
                    const float4 v = (cc < k) ? reinterpret_cast<const float4 *>(csrc + (size_t)cc * KP_C)[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                    reinterpret_cast<float4 *>(lc + (size_t)cc * KP_C)[t] = v;
                }
                __syncthreads();
                if ((int)threadIdx.x < KT * 16) {              // |c|^2 in scipy's order: t = 0..C-1, multiply then add
                    float nrm = INFINITY;
                    if ((int)threadIdx.x < k) {
                        const float *im = lc + (size_t)threadIdx.x * KP_C;
                        nrm = 0.0f;
                        for (int t = 0; t < KP_C; ++t) {
                            const float v = im[t];
                            const float prod = v * v;
                            nrm = nrm + prod;
                        }
                    }
                    lcn[threadIdx.x] = nrm;
                }
                __syncthreads();
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const float *st = lc + (size_t)(kt * 16 + j) * KP_C + g;
#pragma unroll
                    for (int t = 0; t < KP_TM; ++t) ca[kt][t] = st[4 * t];      // lane (i = j, kq = g): c[16 kt + i][4 t + kq]
#pragma unroll
                    for (int r = 0; r < 4; ++r) cn[kt][r] = lcn[kt * 16 + g * 4 + r];
                }
            }
            KP_SEC(0);
            KP_ROWS_COMMIT()
            const float cur_xs = my_xs;
            __syncthreads();                                       // S1: the rows of the block are in LDS
            KP_SEC(1);
            if (blk + 1 < blk_end) {
                my_id = id_next;
                my_xs = xs_next;
                KP_ROWS_ISSUE()                                    // in flight under this block's work
            }
            // ---- assignment of tiles 2h, 2h + 1 of the pair's mini-block
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int tile = 2 * h + tt;
                const float *br = wrow + (size_t)(16 * tile + j) * KP_C + g;
                float xb[KP_TM];
#pragma unroll
                for (int t = 0; t < KP_TM; ++t) xb[t] = br[4 * t];
                const int prow = mb_row0 + tile * 16 + j;
                const float xs_l = __shfl(cur_xs, tile * 16 + j);
                const float xs = (prow < len) ? xs_l : 0.0f;
                float low = INFINITY;
                int arg = 0;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < KP_TM; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[kt][t], xb[t], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float mm = -2.0f * acc[r];
                        const float dist = (mm + xs) + cn[kt][r];
                        if (dist < low) { low = dist; arg = kt * 16 + g * 4 + r; }
                    }
                }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float d2 = __shfl_xor(low, off);
                    const int a2 = __shfl_xor(arg, off);
                    if (d2 < low || (d2 == low && a2 < arg)) { low = d2; arg = a2; }
                }
                if (g == 0) labx[16 * tile + j] = (prow < len) ? arg : -1;
            }
            __syncthreads();                                       // S2: the labels of all four mini-blocks
            KP_SEC(2);
            const int best = labx[lane];                           // lane = row of the pair's mini-block
            const int p = mb_row0 + lane;
            if (h == 0 && p < len) a.labels[beg + p] = best;

            // ---- chunks of this mini-block: presence mask, member masks and counts, any-order sums of |x| of this wave's features
            const int mb = 4 * blk + q;
            unsigned long long pm = 0ull;
            for (int kk = 0; kk < k; ++kk)
                if (__ballot(best == kk) != 0ull) pm |= 1ull << kk;
            if (h == 0 && lane == 0) a.pres[mb] = pm;
            const int order = kp_order_rows(best, pm, lane);
            unsigned long long rem = pm;
            int off = 0;
            while (rem) {
                const int kk = __builtin_ctzll(rem);
                rem &= rem - 1;
                const int cnt = __popcll(__ballot(best == kk));
                float a0 = 0.0f;
                for (int i0 = 0; i0 < cnt; i0 += 8) {
                    float x0[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int b = __builtin_amdgcn_readlane(order, off + min(i0 + u, cnt - 1));
                        x0[u] = wrow[b * KP_C + fc];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) a0 += (i0 + u < cnt) ? fabsf(x0[u]) : 0.0f;
                }
                off += cnt;
                const size_t slot = (size_t)mb * kmax + kk;
                if (fvalid) {
                    a.bsum[slot * KP_C + f] = a0;
                    atomicAdd(&blks[kk * KP_C + f], a0);
                }
                if (h == 0 && lane == 0) {
                    a.ccnt[slot] = cnt;
                    atomicAdd(&blkc[kk], cnt);
                }
            }
            KP_SEC(3);
            __syncthreads();                                       // S3: the block's sums
            // ---- running prefix at block granularity (present clusters only)
            for (int i = threadIdx.x; i < k * KP_C; i += KP_THREADS) {
                const int kk = i / KP_C;
                if (blkc[kk] > 0) {
                    a.PB[((size_t)blk * kmax + kk) * KP_C + (i - kk * KP_C)] = run[i];
                    run[i] += blks[i];
                    blks[i] = 0.0f;
                }
            }
            KP_SEC(4);
            __syncthreads();                                       // S4 (also: everybody is done with the rows and the labels of this block)
            if ((int)threadIdx.x < k) {
                const int c = blkc[threadIdx.x];
                if (c > 0) { a.PBC[(size_t)blk * kmax + threadIdx.x] = runc[threadIdx.x]; runc[threadIdx.x] += c; blkc[threadIdx.x] = 0; }
            }
            KP_SEC(5);
        }
        __syncthreads();
    }
    if (pf) for (int i = 0; i < 6; ++i) g_kp_prof[24 + i] += (unsigned long long)sec[i];
    // range sums of this workgroup (its last segment)
    for (int i = threadIdx.x; i < kmax * KP_C; i += KP_THREADS) a.T[(size_t)wg * kmax * KP_C + i] = run[i];
    if ((int)threadIdx.x < kmax) a.TC[(size_t)wg * kmax + threadIdx.x] = runc[threadIdx.x];
}

// ------------------------------------------------------------------------------------------ phase X
// WB[w] = sum of T over the earlier workgroups whose blocks belong to the segment workgroup w starts in.  One WORKGROUP per (segment,
// cluster, feature half): its eight waves take an eighth of the segment's workgroups each (one round trip for all loads), exchange their
// sums through LDS and write the prefixes from the values they still hold.
constexpr int KP_XW = 16;     // workgroups a wave handles per pass
__device__ __forceinline__ void kp_phase_prefix(const KpArgs &a, KpTables *tab, float *lds, int wg, int q) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax;
    const int ntask = a.n_seg * kmax * 2;
    float *xs = lds;                                                  // [KP_WAVES][64] sums of the waves' shares
    int32_t *xc = reinterpret_cast<int32_t *>(lds + KP_WAVES * 64);    // [KP_WAVES]
    for (int task = wg; task < ntask; task += a.grid) {
        const int h = task & 1, kk = (task >> 1) % kmax, s = (task >> 1) / kmax;
        const int b0 = tab->blk_base[s], b1 = tab->blk_base[s + 1];
        if (kk >= tab->seg_k[s] || b1 <= b0) continue;                 // (uniform over the workgroup)
        const int w_first = b0 / q, w_last = (b1 - 1) / q;
        const int f = 64 * h + lane;
        const bool fvalid = f < KP_C;
        const int fc = fvalid ? f : KP_C - 1;
        float carry = 0.0f;
        int carryc = 0;
        for (int wbase = w_first; wbase < w_last; wbase += KP_WAVES * KP_XW) {
            const int w0 = wbase + wave * KP_XW;
            float t[KP_XW];
            int tc[KP_XW];
#pragma unroll
            for (int u = 0; u < KP_XW; ++u) {
                const int w = min(w0 + u, w_last - 1);
                t[u] = a.T[((size_t)w * kmax + kk) * KP_C + fc];
                tc[u] = a.TC[(size_t)w * kmax + kk];
            }
            float mine = 0.0f;
            int minec = 0;
#pragma unroll
            for (int u = 0; u < KP_XW; ++u)
                if (w0 + u < w_last) { mine += t[u]; minec += tc[u]; }
            xs[wave * 64 + lane] = mine;
            if (lane == 0) xc[wave] = minec;
            __syncthreads();
            float running = carry, total = carry;
            int runningc = carryc, totalc = carryc;
            for (int v = 0; v < KP_WAVES; ++v) {
                const float sv = xs[v * 64 + lane];
                const int sc = xc[v];
                if (v < wave) { running += sv; runningc += sc; }
                total += sv;
                totalc += sc;
            }
#pragma unroll
            for (int u = 0; u < KP_XW; ++u) {
                const int w = w0 + u;
                if (w < w_last) {
                    running += t[u];
                    runningc += tc[u];
                    if (fvalid) a.WB[((size_t)(w + 1) * kmax + kk) * KP_C + f] = running;
                    if (h == 0 && lane == 0) a.WBC[(size_t)(w + 1) * kmax + kk] = runningc;
                }
            }
            carry = total;
            carryc = totalc;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------ phase B
__device__ __forceinline__ void kp_phase_fold(const KpArgs &a, KpTables *tab, float *lds, int blk_begin, int blk_end, int wg) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int q = wave >> 1, h = wave & 1;
    const int kmax = a.kmax;
    float *base = lds;                                              // [kmax][C]
    int32_t *basec = reinterpret_cast<int32_t *>(base + (size_t)kmax * KP_C);        // [kmax]
    float *wrow = reinterpret_cast<float *>(basec + ((kmax + 3) & ~3)) + (size_t)q * KP_ROWF;
    float *lits = reinterpret_cast<float *>(basec + ((kmax + 3) & ~3)) + (size_t)4 * KP_ROWF + (size_t)wave * KP_LIT * 64 + lane;   // [KP_LIT][64] per wave
    if (blk_begin >= blk_end) return;
    const bool pf = a.prof != 0 && wg == a.prof - 1 && threadIdx.x == 0;
    unsigned sec[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    long long tsec = pf ? wall_clock64() : 0ll;
    int s = kp_locate(tab, blk_begin, 0);
    // the earlier workgroups' share of this workgroup's first segment (phase X)
    {
        const bool inside = blk_begin > tab->blk_base[s];
        for (int i = threadIdx.x; i < kmax * KP_C; i += KP_THREADS) base[i] = inside ? a.WB[(size_t)wg * kmax * KP_C + i] : 0.0f;
        if ((int)threadIdx.x < kmax) basec[threadIdx.x] = inside ? a.WBC[(size_t)wg * kmax + threadIdx.x] : 0;
        if (threadIdx.x == 0) tab->litcnt = 0;
        __syncthreads();
    }
    KP_SEC(0);
    const int f = 64 * h + lane;
    const bool fvalid = f < KP_C;
    const int fc = fvalid ? f : KP_C - 1;
    int cur_seg = s;
    KP_ROWS_DECL()
    int my_id, id_next = 0, lab, lab_next = -1;
    unsigned long long pres4 = 0ull, pres4_next = 0ull;      // lane qq < 4: presence mask of mini-block qq of the block
    {
        const int beg = tab->seg_off[s], len = tab->seg_off[s + 1] - beg;
        const int p = (blk_begin - tab->blk_base[s]) * 256 + q * 64 + lane;
        my_id = a.rows[beg + max(min(p, len - 1), 0)];
        lab = (p < len) ? a.labels[beg + p] : -1;
        pres4 = a.pres[4 * blk_begin + (lane & 3)];
        KP_ROWS_ISSUE()
    }
    int n_steps = 0, n_generic = 0;
    for (int blk = blk_begin; blk < blk_end; ++blk) {
        s = kp_locate(tab, blk, s);
        if (s != cur_seg) {                                     // a new segment starts inside this workgroup's range: nothing in front of it
            cur_seg = s;
            for (int i = threadIdx.x; i < kmax * KP_C; i += KP_THREADS) base[i] = 0.0f;      // (the previous block ended behind a barrier)
            if ((int)threadIdx.x < kmax) basec[threadIdx.x] = 0;
        }
        if (blk + 1 < blk_end) {
            const int s1 = kp_locate(tab, blk + 1, s);
            const int b1 = tab->seg_off[s1], l1n = tab->seg_off[s1 + 1] - b1;
            const int p1 = (blk + 1 - tab->blk_base[s1]) * 256 + q * 64 + lane;
            id_next = a.rows[b1 + max(min(p1, l1n - 1), 0)];
            lab_next = (p1 < l1n) ? a.labels[b1 + p1] : -1;
            pres4_next = a.pres[4 * (blk + 1) + (lane & 3)];
        }
        const int mb = 4 * blk + q;
        const unsigned long long pm = kp_readlane64(pres4, q);
        unsigned long long pprev[3];
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) pprev[qq] = (qq < q) ? kp_readlane64(pres4, qq) : 0ull;
        KP_ROWS_COMMIT()
        const int cur_lab = lab;
        __syncthreads();                                           // the rows of the block are in LDS (and base[] is set)
        if (blk + 1 < blk_end) {
            my_id = id_next;
            lab = lab_next;
            pres4 = pres4_next;
            KP_ROWS_ISSUE()
        }
        const int order = kp_order_rows(cur_lab, pm, lane);
        int off_next = 0;
        unsigned long long rem = pm;
        KP_SEC(1);
        while (rem) {
            // ---- predictions of up to four chunks at once: every load is issued before any is used
            int kks[4], offs[4], cnts[4];
            float P0[4];
            int mbf[4];
            int ng = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool on = rem != 0ull;
                const int kk = on ? __builtin_ctzll(rem) : 0;
                if (on) { rem &= rem - 1; ++ng; }
                kks[u] = kk;
                cnts[u] = on ? __popcll(__ballot(cur_lab == kk)) : 0;
                offs[u] = off_next;
                off_next += cnts[u];
                float p0 = a.PB[((size_t)blk * kmax + kk) * KP_C + fc];
                int mc = a.PBC[(size_t)blk * kmax + kk];
                float b0[3];
                int cc[3];
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) {
                    const size_t sl = (size_t)(4 * blk + qq) * kmax + kk;
                    b0[qq] = a.bsum[sl * KP_C + fc];
                    cc[qq] = a.ccnt[sl];
                }
                p0 += base[kk * KP_C + fc];
                mc += basec[kk];
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) {
                    const bool here = ((pprev[qq] >> kk) & 1ull) != 0ull;
                    p0 += here ? b0[qq] : 0.0f;
                    mc += here ? cc[qq] : 0;
                }
                P0[u] = p0; mbf[u] = mc;
            }
            KP_SEC(2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < ng) {
                    const int kk = kks[u];
                    const size_t slot = (size_t)mb * kmax + kk;
                    uint32_t *r = a.rec + slot * KP_REC * KP_C;
                    const int total = cnts[u], off = offs[u];
                    if (mbf[u] < KP_HEAD) {                     // the stitch adds this chunk literally (head of the cluster): its rows, in order
                        if (fvalid) r[f] = kx_hdr(KX_UNSAFE, 0, 0, 0, 0, 0) | KP_HDR_HEAD;
                        float *hb = a.headbuf + (((size_t)s * kmax + kk) * KP_HEADCAP + mbf[u]) * KP_C + f;
                        for (int i = 0; i < total; ++i) {
                            const int b = __builtin_amdgcn_readlane(order, off + i);
                            if (fvalid) hb[(size_t)i * KP_C] = wrow[b * KP_C + f];
                        }
                    } else {
                        KxFold k0;
                        kx_fold_init(k0, P0[u], mbf[u]);
                        KP_SEC(3);
                        // Members in row order.  The tight loop only advances (acc, s): a member that needs more (a tie in some lane, a lane near
                        // the end of its binade or inside a literal window, a value that is not a plain number) leaves it for one general step.
                        int i = 0;
                        float xn = wrow[__builtin_amdgcn_readlane(order, off) * KP_C + fc];
                        float xnn = wrow[__builtin_amdgcn_readlane(order, off + min(1, total - 1)) * KP_C + fc];
                        while (i < total) {
                            int32_t acc = k0.acc;
                            float ss = k0.s;
                            const float inv_u = k0.inv_u;
                            const uint32_t lim = k0.lim;
                            const bool win = k0.mode == KXM_WIN;
                            float x = xn;
                            while (i < total) {
                                x = xn;
                                xn = xnn;
                                xnn = wrow[__builtin_amdgcn_readlane(order, off + min(i + 2, total - 1)) * KP_C + fc];      // two members ahead
                                const float t = __builtin_fmaf(x, inv_u, KX_MAGIC);
                                const uint32_t rr = kx_f2u(t) - KX_MAGIC_BITS;
                                const float rn = t - KX_MAGIC;
                                const float dd = __builtin_fmaf(x, inv_u, -rn);
                                const uint32_t cand = (uint32_t)acc + rr;
                                const bool special = __builtin_fabsf(dd) == 0.5f || rr >= 0x800000u || cand + 2u > lim || win;
                                if (__any(special)) break;
                                acc = (int32_t)cand;
                                ss = ss + x;
                                ++i;
                            }
                            k0.acc = acc;
                            k0.s = ss;
                            n_steps += i;
                            if (i < total) {
                                ++n_generic;
                                const KxFast s0 = kx_fold_fast(k0, x);
                                kx_fold_step(k0, s0, x, i, lits, 64);      // (lanes that are not over take their fast result, ties included)
                                ++i;
                            }
                        }
                        KP_SEC(4);
                        // record
                        int32_t A0, B0;
                        uint32_t hdr = kx_fold_finish(k0, total, A0, B0);
                        uint32_t w3 = 0u;
                        const int nlit = kx_hdr_nlit(hdr);
                        if (nlit == 1) w3 = kx_f2u(lits[0]);
                        if (nlit > 1 && fvalid) {
                            const int off = atomicAdd(&tab->litcnt, nlit);
                            if (off + nlit > KP_LITCAP) {
                                hdr = kx_hdr(KX_UNSAFE, 0, 0, 0, 0, 0);
                            } else {
                                float *dst = a.litpool + (size_t)wg * KP_LITCAP + off;
                                for (int i = 0; i < nlit; ++i) dst[i] = lits[i * 64];
                                w3 = (uint32_t)(wg * KP_LITCAP + off);
                            }
                        }
                        if (fvalid) {
                            r[f] = hdr;
                            r[KP_C + f] = (uint32_t)A0;
                            r[2 * KP_C + f] = (uint32_t)B0;
                            r[3 * KP_C + f] = w3;
                        }
                        KP_SEC(5);
                    }
                }
            }
        }
        __syncthreads();                                           // everybody is done with the rows of this block
    }
    if (pf) {
        g_kp_prof[12] += (unsigned long long)n_steps;
        g_kp_prof[13] += (unsigned long long)n_generic;
        for (int i = 0; i < 6; ++i) g_kp_prof[16 + i] += (unsigned long long)sec[i];
    }
}

// ------------------------------------------------------------------------------------------ phase C
__device__ __forceinline__ void kp_phase_merge(const KpArgs &a, KpTables *tab, int wg, int n_part) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax;
    const int ntask = n_part * kmax * 2;
    int s = 0;
    for (int task = wg + a.grid * wave; task < ntask; task += a.grid * KP_WAVES) {      // consecutive tasks on different CUs
        const int h = task & 1, kk = (task >> 1) % kmax, part = (task >> 1) / kmax;
        s = 0;
        while (part >= tab->part_base[s + 1]) ++s;
        if (kk >= tab->seg_k[s]) continue;
        const int pp = part - tab->part_base[s];
        const int nmb = 4 * (tab->blk_base[s + 1] - tab->blk_base[s]);
        const int mb0 = 4 * tab->blk_base[s] + 64 * pp;
        const int nm = min(64, nmb - 64 * pp);
        const unsigned long long pw = (lane < nm) ? a.pres[mb0 + lane] : 0ull;
        const bool here = ((pw >> kk) & 1ull) != 0ull;
        const unsigned long long pmask = __ballot(here);
        if (h == 0) {
            int c = here ? a.ccnt[(size_t)(mb0 + lane) * kmax + kk] : 0;
            for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
            if (lane == 0) a.part_cnt[(size_t)part * kmax + kk] = c;
        }
        const int f = 64 * h + lane;
        const bool fvalid = f < KP_C;
        const int fc = fvalid ? f : KP_C - 1;
        KxRun run{0, 0, 0};
        unsigned long long np = 0ull;
        unsigned long long mm = pmask;
        while (mm) {
            // sixteen positions at a time: all loads first
            int poss[16];
            uint32_t hd[16], w1[16];
            int ng = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool on = mm != 0ull;
                const int pos = on ? __builtin_ctzll(mm) : 0;
                if (on) { mm &= mm - 1; ++ng; }
                poss[u] = pos;
                const uint32_t *r = a.rec + ((size_t)(mb0 + pos) * kmax + kk) * KP_REC * KP_C + fc;
                hd[u] = r[0]; w1[u] = r[KP_C];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // head chunks (uniform over the lanes): the stitch takes them from the rows, nothing is in front of them
                if (u < ng && !(hd[u] & KP_HDR_HEAD)) {
                    if (!kx_run_merge(run, hd[u], (int32_t)w1[u])) {
                        if (fvalid) {
                            uint32_t *r = a.rec + ((size_t)(mb0 + poss[u]) * kmax + kk) * KP_REC * KP_C + fc;
                            r[4 * KP_C] = kx_run_hdr(run);
                            r[5 * KP_C] = (uint32_t)run.R0;
                        }
                        np |= 1ull << poss[u];
                        run = KxRun{0, 0, 0};
                    }
                }
            }
        }
        if (fvalid) {
            uint32_t *po = a.part_post + ((size_t)part * kmax + kk) * 2 * KP_C + f;
            po[0] = kx_run_hdr(run);
            po[KP_C] = (uint32_t)run.R0;
            a.part_np[((size_t)part * kmax + kk) * KP_C + f] = np;
        }
    }
}

// ------------------------------------------------------------------------------------------ phase S
__device__ __forceinline__ void kp_phase_stitch(const KpArgs &a, KpTables *tab, float *lds, int wg) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax;
    const int ntask = a.n_seg * kmax * 2;
    (void)lds;
    const bool pf = a.prof != 0 && wg == a.prof - 1 && threadIdx.x == 0;
    unsigned sec[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    long long tsec = pf ? wall_clock64() : 0ll;
    int n_parts_walked = 0, n_rec = 0, n_stuck = 0, n_expand = 0;
    for (int task = wg + a.grid * wave; task < ntask; task += a.grid * KP_WAVES) {        // consecutive tasks on different CUs
        const int h = task & 1, kk = (task >> 1) % kmax, s = (task >> 1) / kmax;
        if (kk >= tab->seg_k[s]) continue;
        const int f = 64 * h + lane;
        const bool fvalid = f < KP_C;
        const int fc = fvalid ? f : KP_C - 1;
        const int beg = tab->seg_off[s], len = tab->seg_off[s + 1] - beg;
        const int mb_base = 4 * tab->blk_base[s];
        const int nmb = 4 * (tab->blk_base[s + 1] - tab->blk_base[s]);
        const int nparts = tab->part_base[s + 1] - tab->part_base[s];
        float sv = 0.0f;
        int cnt_total = 0;

        // ---- head: the chunks that start below KP_HEAD members, added literally; the fold phase has put their rows in order into headbuf.
        // The first rows are fetched before anything is known about the head's length (they exist whenever the cluster is not empty).
        const float *hb = a.headbuf + ((size_t)s * kmax + kk) * KP_HEADCAP * KP_C + fc;
        constexpr int HB = 24;
        float hx[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) hx[u] = hb[(size_t)u * KP_C];
        int head_part = nparts, head_pos = 0;       // the first chunk behind the head: part, position inside it
        int n_head = 0;
        {
            // members per part (lane = part): a part whose members all lie below KP_HEAD belongs to the head as a whole
            bool found = false;
            int c_found = 0;
            for (int p0 = 0; p0 < nparts && !found; p0 += 64) {
                const int pc = (p0 + lane < nparts) ? a.part_cnt[(size_t)(tab->part_base[s] + p0 + lane) * kmax + kk] : 0;
                for (int l = 0; l < 64 && p0 + l < nparts; ++l) {
                    const int c = __builtin_amdgcn_readlane(pc, l);
                    if (n_head + c > KP_HEAD) { head_part = p0 + l; c_found = c; found = true; break; }
                    n_head += c;
                    cnt_total += c;
                }
            }
            if (found) {
                // inside head_part: chunks while the members in front of them are fewer than KP_HEAD
                const int mb0 = mb_base + 64 * head_part;
                const int nm = min(64, nmb - 64 * head_part);
                const unsigned long long pw = (lane < nm) ? a.pres[mb0 + lane] : 0ull;
                const bool here = ((pw >> kk) & 1ull) != 0ull;
                unsigned long long mmk = __ballot(here);
                const int cl = here ? a.ccnt[(size_t)(mb0 + lane) * kmax + kk] : 0;
                head_pos = 64;
                while (mmk) {
                    const int pos = __builtin_ctzll(mmk);
                    if (n_head >= KP_HEAD) { head_pos = pos; break; }
                    mmk &= mmk - 1;
                    n_head += __builtin_amdgcn_readlane(cl, pos);
                }
                if (head_pos == 64) { head_part += 1; head_pos = 0; cnt_total += c_found; }      // the part's last chunk straddles KP_HEAD
            }
        }
        KP_SEC(0);
        for (int i0 = 0; i0 < n_head; i0 += HB) {
            float cur[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) cur[u] = hx[u];
            if (i0 + HB < n_head) {
#pragma unroll
                for (int u = 0; u < HB; ++u) hx[u] = hb[(size_t)min(i0 + HB + u, KP_HEADCAP - 1) * KP_C];
            }
#pragma unroll
            for (int u = 0; u < HB; ++u)
                if (i0 + u < n_head) sv = sv + cur[u];
        }
        // members of the parts in front of head_part were counted above; a head that ends exactly at a part boundary starts the walk there
        (void)beg; (void)len;

        KP_SEC(1);
        // what the walk needs of a part, two parts ahead: np (positions of its records), its last run, its member count; one part ahead:
        // the first KP_EV records of every lane, gathered from the record array
        const size_t pk0 = (size_t)tab->part_base[s] * kmax + kk;
        unsigned long long np_c = 0ull, np_1 = 0ull, np_2 = 0ull;
        uint32_t ph_c = 0u, ph_1 = 0u, ph_2 = 0u, pR_c = 0u, pR_1 = 0u, pR_2 = 0u;
        int cn_c = 0, cn_1 = 0, cn_2 = 0;
        uint32_t ev_c[KP_EV][KP_REC], ev_1[KP_EV][KP_REC];
#pragma unroll
        for (int e = 0; e < KP_EV; ++e)
#pragma unroll
            for (int i = 0; i < KP_REC; ++i) { ev_c[e][i] = 0u; ev_1[e][i] = 0u; }
#define KP_PART_A(pp_, np_, ph_, pR_, cn_)                                                 \
        if ((pp_) < nparts) {                                                              \
            const size_t pk_ = pk0 + (size_t)(pp_) * kmax;                                 \
            np_ = a.part_np[pk_ * KP_C + fc];                                              \
            ph_ = a.part_post[pk_ * 2 * KP_C + fc];                                        \
            pR_ = a.part_post[pk_ * 2 * KP_C + KP_C + fc];                                 \
            cn_ = a.part_cnt[pk_];                                                         \
        }
#define KP_PART_E(pp_, np_, ev_, skip_)                                                    \
        if ((pp_) < nparts) {                                                              \
            unsigned long long m_ = (np_) & ~kp_below(skip_);                              \
            _Pragma("unroll") for (int e = 0; e < KP_EV; ++e) {                            \
                const int pos_ = m_ ? __builtin_ctzll(m_) : 0;                             \
                const uint32_t *r_ = a.rec + ((size_t)(mb_base + 64 * (pp_) + pos_) * kmax + kk) * KP_REC * KP_C + fc; \
                if (__any(m_ != 0ull)) {                                                   \
                    _Pragma("unroll") for (int i = 0; i < KP_REC; ++i) ev_[e][i] = r_[(size_t)i * KP_C]; \
                }                                                                          \
                m_ &= m_ - 1;                                                              \
            }                                                                              \
        }
        KP_PART_A(head_part, np_c, ph_c, pR_c, cn_c)
        KP_PART_A(head_part + 1, np_1, ph_1, pR_1, cn_1)
        KP_PART_E(head_part, np_c, ev_c, head_pos)
        for (int pp = head_part; pp < nparts; ++pp) {
            const int mb0 = mb_base + 64 * pp;
            KP_PART_A(pp + 2, np_2, ph_2, pR_2, cn_2)
            KP_PART_E(pp + 1, np_1, ev_1, 0)
            struct { unsigned long long np; uint32_t post_h, post_R; int cnt; } pd;
            pd.np = np_c; pd.post_h = ph_c; pd.post_R = pR_c; pd.cnt = cn_c;
            uint32_t evv[KP_EV][KP_REC];
#pragma unroll
            for (int e = 0; e < KP_EV; ++e)
#pragma unroll
                for (int i = 0; i < KP_REC; ++i) { evv[e][i] = ev_c[e][i]; ev_c[e][i] = ev_1[e][i]; }
            np_c = np_1; ph_c = ph_1; pR_c = pR_1; cn_c = cn_1;
            np_1 = np_2; ph_1 = ph_2; pR_1 = pR_2; cn_1 = cn_2;
            if (pd.cnt == 0) continue;                          // the cluster has no member in this part (uniform)
            cnt_total += pd.cnt;
            ++n_parts_walked;
            unsigned long long np = pd.np;
            unsigned long long pmask = 0ull;
            bool have_pmask = false;
            bool norun = false, stuck = false, post_done = false;
            int done = 0, spos = 64, evi = 0;
            if (pp == head_part) { np &= ~kp_below(head_pos); done = head_pos; }       // the head's chunks of this part are in the sum already
            for (;;) {
                // ---- every lane walks its own list of records until it is through or stuck
                for (;;) {
                    const int pos = (!stuck && np != 0ull) ? __builtin_ctzll(np) : 64;
                    const bool act = pos < 64;
                    if (!__any(act)) break;
                    ++n_rec;
                    // the record of this lane: inline copy (the first KP_EV of the part, while no run failed) or from the record array
                    const bool inl = act && !norun && evi < KP_EV;
                    bool renew = false;
                    uint32_t w[KP_REC];
#pragma unroll
                    for (int i = 0; i < KP_REC; ++i) w[i] = 0u;
                    if (__any(act && !inl)) {
                        if (act && !inl) {
                            const uint32_t *r = a.rec + ((size_t)(mb0 + pos) * kmax + kk) * KP_REC * KP_C + fc;
#pragma unroll
                            for (int i = 0; i < KP_REC; ++i) w[i] = r[(size_t)i * KP_C];
                        }
                    }
                    if (inl) {
#pragma unroll
                        for (int e = 0; e < KP_EV; ++e)
                            if (evi == e) {
#pragma unroll
                                for (int i = 0; i < KP_REC; ++i) w[i] = evv[e][i];
                            }
                    }
                    if (act) {
                        const uint32_t hdr = w[0];
                        const int32_t A0 = (int32_t)w[1], B0 = (int32_t)w[2];
                        const uint32_t w3 = w[3];
                        bool expand = false;
                        if (!norun) {
                            const KxRun run = kx_run_unpack(w[4], (int32_t)w[5]);
                            float t = sv;
                            if (kx_apply_run(t, run)) sv = t;
                            else expand = true;
                        }
                        if (expand) {
                            // a chunk of the run in front of this record did not happen as predicted: this lane takes the rest of the part record by record
                            norun = true;
                            renew = true;
                        } else {
                            done = pos;
                            float t = sv;
                            bool ok = true;
                            const int kind = kx_hdr_kind(hdr);
                            if (kind == KX_UNSAFE) ok = false;
                            else if (kind == KX_SET) {
                                ok = kx_f2u(t) == 0u;
                                t = kx_u2f((uint32_t)A0);
                            } else {
                                const int eA = kx_hdr_eA(hdr), eB = kx_hdr_eB(hdr), nlit = kx_hdr_nlit(hdr);
                                if (eA) ok = kx_apply_int(t, eA - 1, A0, kx_hdr_dA(hdr));
                                if (nlit == 1) t = t + kx_u2f(w3);
                                else
                                    for (int i = 0; i < nlit; ++i) t = t + a.litpool[w3 + i];
                                if (ok && eB) ok = kx_apply_int(t, eB - 1, B0, kx_hdr_dB(hdr));
                            }
                            if (ok) { sv = t; np &= np - 1; done = pos + 1; ++evi; }
                            else { stuck = true; spos = pos; }
                        }
                    }
                    // lanes that gave up on the runs need the part's presence mask: every present position from `done` on becomes a record to visit
                    if (__any(renew) && !have_pmask) {
                        const int nm = min(64, nmb - 64 * pp);
                        const unsigned long long pw = (lane < nm) ? a.pres[mb0 + lane] : 0ull;
                        pmask = __ballot(((pw >> kk) & 1ull) != 0ull);
                        have_pmask = true;
                    }
                    if (renew) np = pmask & ~kp_below(done);
                }
                // ---- stuck lanes: the lowest stuck position is summed literally from its rows (lanes = features, coalesced)
                if (__any(stuck)) {
                    ++n_stuck;
                    int pmin = stuck ? spos : 64;
                    for (int o = 32; o > 0; o >>= 1) pmin = min(pmin, __shfl_xor(pmin, o));
                    const int prow = (mb0 + pmin - mb_base) * 64 + lane;
                    const int lab = (prow < len) ? a.labels[beg + prow] : -1;
                    const int id = a.rows[beg + max(min(prow, len - 1), 0)];
                    unsigned long long mm = __ballot(lab == kk);
                    const bool mine = stuck && spos == pmin;
                    while (mm) {
                        float x[8];
                        bool on[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            on[u] = mm != 0ull;
                            const int b = on[u] ? __builtin_ctzll(mm) : 0;
                            if (on[u]) mm &= mm - 1;
                            x[u] = a.pool[(size_t)__builtin_amdgcn_readlane(id, b) * KP_C + fc];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (on[u] && mine) sv = sv + x[u];
                    }
                    if (mine) { stuck = false; np &= ~(1ull << pmin); done = pmin + 1; ++evi; }
                    continue;
                }
                // ---- the part's last run
                bool again = false;
                if (!norun && !post_done) {
                    post_done = true;
                    const KxRun run = kx_run_unpack(pd.post_h, (int32_t)pd.post_R);
                    float t = sv;
                    if (kx_apply_run(t, run)) sv = t;
                    else { norun = true; again = true; }
                }
                if (!__any(again)) break;
                ++n_expand;
                // a failed last run: the positions from `done` on, one by one
                if (!have_pmask) {
                    const int nm = min(64, nmb - 64 * pp);
                    const unsigned long long pw = (lane < nm) ? a.pres[mb0 + lane] : 0ull;
                    pmask = __ballot(((pw >> kk) & 1ull) != 0ull);
                    have_pmask = true;
                }
                if (again) np = pmask & ~kp_below(done);
            }
        }
        KP_SEC(2);
        if (h == 0 && lane == 0) a.cluster_counts[s * kmax + kk] = cnt_total;
        if (cnt_total > 0 && fvalid) a.centroids[((size_t)s * kmax + kk) * KP_C + f] = sv / (float)cnt_total;
    }
    if (pf) {
        g_kp_prof[14] += (unsigned long long)sec[0];
        g_kp_prof[15] += (unsigned long long)sec[1];
        g_kp_prof[22] += (unsigned long long)sec[2];
        g_kp_prof[23] += (unsigned long long)n_parts_walked;
        g_kp_prof[30] += (unsigned long long)n_rec;
        g_kp_prof[31] += (unsigned long long)(n_stuck * 1000 + n_expand);
    }
}

// ------------------------------------------------------------------------------------------ the chain
template <int KT>
__global__ __launch_bounds__(KP_THREADS, 1) void km_chain_kernel(KpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    KpTables *tab = reinterpret_cast<KpTables *>(lds_raw);
    float *lds = lds_raw + (sizeof(KpTables) + 15) / 16 * 4;
    const int wg = blockIdx.x;
    for (int i = threadIdx.x; i <= a.n_seg; i += KP_THREADS) tab->seg_off[i] = a.seg_off[i];
    for (int i = threadIdx.x; i < a.n_seg; i += KP_THREADS) tab->seg_k[i] = a.seg_k[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        int nb = 0, np = 0;
        for (int s = 0; s < a.n_seg; ++s) {
            tab->blk_base[s] = nb;
            tab->part_base[s] = np;
            const int len = tab->seg_off[s + 1] - tab->seg_off[s];
            const int b = (tab->seg_k[s] > 0) ? (len + 255) / 256 : 0;
            nb += b;
            np += (b + 15) / 16;
        }
        tab->blk_base[a.n_seg] = nb;
        tab->part_base[a.n_seg] = np;
        tab->error = 0;
        tab->litcnt = 0;
    }
    __syncthreads();
    const int nb_total = tab->blk_base[a.n_seg], n_part = tab->part_base[a.n_seg];
    const int q = max((nb_total + a.grid - 1) / a.grid, 1);
    const int blk_begin = min(wg * q, nb_total), blk_end = min(blk_begin + q, nb_total);

    const bool prof = a.prof != 0;
    long long tprev = prof ? wall_clock64() : 0ll;
    unsigned wgsec[12] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // row norms (sequential, scipy's order): every thread for rows of its own workgroup's blocks (no exchange between workgroups)
    {
        int s = 0;
        for (int blk = blk_begin; blk < blk_end; ++blk) {
            while (blk >= tab->blk_base[s + 1]) ++s;
            const int beg = tab->seg_off[s], len = tab->seg_off[s + 1] - beg;
            const int p = (blk - tab->blk_base[s]) * 256 + threadIdx.x;
            if (threadIdx.x < 256 && p < len) a.rownorm[beg + p] = kp_sqnorm_row(reinterpret_cast<const float4 *>(a.pool + (size_t)a.rows[beg + p] * KP_C));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    KP_TICK(0);
    uint32_t epoch = 0;
    bool ok = true;
    for (int it = 0; it < a.iters && ok; ++it) {
        kp_phase_assign<KT>(a, tab, lds, blk_begin, blk_end, wg);
        KP_TICK(1);
        ok = kp_grid_barrier(a.bar, tab, epoch, a.grid);
        KP_TICK(2);
        if (!ok) break;
        kp_phase_prefix(a, tab, lds, wg, q);
        KP_TICK(10);
        ok = kp_grid_barrier(a.bar, tab, epoch, a.grid);
        KP_TICK(11);
        if (!ok) break;
        kp_phase_fold(a, tab, lds, blk_begin, blk_end, wg);
        KP_TICK(3);
        ok = kp_grid_barrier(a.bar, tab, epoch, a.grid);
        KP_TICK(4);
        if (!ok) break;
        kp_phase_merge(a, tab, wg, n_part);
        KP_TICK(5);
        ok = kp_grid_barrier(a.bar, tab, epoch, a.grid);
        KP_TICK(6);
        if (!ok) break;
        kp_phase_stitch(a, tab, lds, wg);
        KP_TICK(7);
        ok = kp_grid_barrier(a.bar, tab, epoch, a.grid);
        KP_TICK(8);
    }
    if (prof && wg == a.prof - 1 && threadIdx.x == 0) g_kp_prof[9] += 1ull;
    if (prof && threadIdx.x == 0)
        for (int i = 0; i < 12; ++i) g_kp_wgprof[wg * 16 + i] = wgsec[i];
    if (!ok) {
        // a barrier timed out (the grid was not resident as a whole): poison the code books so that nothing downstream looks plausible
        for (int i = wg * KP_THREADS + threadIdx.x; i < a.n_seg * a.kmax * KP_C; i += a.grid * KP_THREADS) a.centroids[i] = __builtin_nanf("");
    }
}

size_t kp_lds_bytes(int kt, int kmax) {
    const size_t tabs = (sizeof(KpTables) + 15) / 16 * 16;
    const size_t pa = ((size_t)kt * 16 * KP_C + kt * 16 + (size_t)4 * KP_ROWF + (size_t)2 * kmax * KP_C + 2 * kmax + 4 * 64) * 4;
    const size_t pb = ((size_t)kmax * KP_C + ((kmax + 3) & ~3) + (size_t)4 * KP_ROWF + (size_t)KP_WAVES * KP_LIT * 64) * 4;
    const size_t ps = (size_t)KP_WAVES * KP_HEADCAP * 4;
    return tabs + std::max(std::max(pa, pb), ps) + 16;
}

struct KpLayout {
    size_t pres, headbuf, ccnt, bsum, rec, PB, PBC, T, TC, WB, WBC, litpool, part_post, part_np, part_cnt, bar, total;
};
KpLayout kp_layout(int64_t cap, int n_seg, int kmax) {
    const size_t nblk = (size_t)(cap / 256) + n_seg + 1, nmb = 4 * nblk, slots = nmb * kmax, nparts = nblk / 16 + n_seg + 1;
    KpLayout l;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += aoc_align_up(bytes, 256); return at; };
    l.bar = take(sizeof(KpBar));
    l.pres = take(nmb * 8);
    l.headbuf = take((size_t)n_seg * kmax * KP_HEADCAP * KP_C * 4);
    l.ccnt = take(slots * 4);
    l.bsum = take(slots * KP_C * 4);
    l.rec = take(slots * KP_REC * KP_C * 4);
    l.PB = take(nblk * kmax * KP_C * 4);
    l.PBC = take(nblk * kmax * 4);
    l.T = take((size_t)KP_GMAX * kmax * KP_C * 4);
    l.TC = take((size_t)KP_GMAX * kmax * 4);
    l.WB = take((size_t)(KP_GMAX + 1) * kmax * KP_C * 4);
    l.WBC = take((size_t)(KP_GMAX + 1) * kmax * 4);
    l.litpool = take((size_t)KP_GMAX * KP_LITCAP * 4);
    l.part_post = take(nparts * kmax * 2 * KP_C * 4);
    l.part_np = take(nparts * kmax * KP_C * 8);
    l.part_cnt = take(nparts * kmax * 4);
    l.total = o;
    return l;
}

int kp_grid_request() {
    static const int g = getenv("AOC_KM_GRID") ? atoi(getenv("AOC_KM_GRID")) : 0;
    return g;
}
int g_kp_grid_override = 0;

}  // namespace

// ---- internal interface (labels_kmeans.hip)
bool aoc_kp_supported(int C, int n_seg, int kmax) { return C == KP_C && n_seg <= KP_SEG_MAX && kmax <= 32 && kmax >= 1; }
size_t aoc_kp_workspace_bytes(int64_t rows_capacity, int n_seg, int kmax) { return kp_layout(rows_capacity, n_seg, kmax).total; }

// The chain after km_init_kernel (code books = initial rows): `iters` Lloyd iterations in one launch.  rownorm: [rows_capacity] scratch.
int aoc_kp_chain(const float *pool, const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k, int n_seg, int kmax, int iters,
                 int64_t rows_capacity, float *centroids, int32_t *labels, int32_t *cluster_counts, float *rownorm, void *workspace, hipStream_t st) {
    const int kt = (kmax + 15) / 16;
    const size_t lds = kp_lds_bytes(kt, kmax);
    // the grid has to be resident as a whole: workgroups per CU from the occupancy query (cached per code-book tile count)
    static int max_grid[3] = {0, 0, 0};
    static int n_cu = 0;
    if (max_grid[kt] == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return AOC_ERR_LAUNCH;
        n_cu = prop.multiProcessorCount;
        int per_cu = 0;
        hipError_t e = hipErrorUnknown;
        if (kt == 1) { (void)hipFuncSetAttribute((const void *)km_chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, km_chain_kernel<1>, KP_THREADS, lds); }
        if (kt == 2) { (void)hipFuncSetAttribute((const void *)km_chain_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, km_chain_kernel<2>, KP_THREADS, lds); }
        if (e != hipSuccess || per_cu < 1) return AOC_ERR_LAUNCH;
        max_grid[kt] = std::max(1, per_cu * n_cu);          // one workgroup per CU by construction (its LDS buffers take most of a CU's 160 KB)
    }
    int grid = g_kp_grid_override > 0 ? g_kp_grid_override : (kp_grid_request() > 0 ? kp_grid_request() : n_cu);
    grid = std::max(1, std::min(std::min(grid, max_grid[kt]), KP_GMAX));
    const KpLayout l = kp_layout(rows_capacity, n_seg, kmax);
    char *w = static_cast<char *>(workspace);
    KpArgs a;
    a.pool = pool; a.rows = rows; a.seg_off = seg_offsets; a.seg_k = seg_k;
    a.n_seg = n_seg; a.kmax = kmax; a.iters = iters; a.grid = grid;
    static const int prof = (getenv("AOC_KM_PROF") && atoi(getenv("AOC_KM_PROF")) > 0) ? atoi(getenv("AOC_KM_PROF")) : 0;   // workgroup + 1
    a.prof = prof;
    a.centroids = centroids; a.labels = labels; a.cluster_counts = cluster_counts; a.rownorm = rownorm;
    a.pres = reinterpret_cast<unsigned long long *>(w + l.pres);
    a.headbuf = reinterpret_cast<float *>(w + l.headbuf);
    a.ccnt = reinterpret_cast<int32_t *>(w + l.ccnt);
    a.bsum = reinterpret_cast<float *>(w + l.bsum);
    a.rec = reinterpret_cast<uint32_t *>(w + l.rec);
    a.PB = reinterpret_cast<float *>(w + l.PB);
    a.PBC = reinterpret_cast<int32_t *>(w + l.PBC);
    a.T = reinterpret_cast<float *>(w + l.T);
    a.TC = reinterpret_cast<int32_t *>(w + l.TC);
    a.WB = reinterpret_cast<float *>(w + l.WB);
    a.WBC = reinterpret_cast<int32_t *>(w + l.WBC);
    a.litpool = reinterpret_cast<float *>(w + l.litpool);
    a.part_post = reinterpret_cast<uint32_t *>(w + l.part_post);
    a.part_np = reinterpret_cast<unsigned long long *>(w + l.part_np);
    a.part_cnt = reinterpret_cast<int32_t *>(w + l.part_cnt);
    a.bar = reinterpret_cast<KpBar *>(w + l.bar);
    if (hipMemsetAsync(a.bar, 0, sizeof(KpBar), st) != hipSuccess) return AOC_ERR_LAUNCH;
    if (kt == 1) hipLaunchKernelGGL(km_chain_kernel<1>, dim3(grid), dim3(KP_THREADS), lds, st, a);
    else hipLaunchKernelGGL(km_chain_kernel<2>, dim3(grid), dim3(KP_THREADS), lds, st, a);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

extern "C" int aoc_kmeans_chain_profile(unsigned long long *out32_host, int reset) {
    if (!out32_host) return AOC_ERR_INVALID_ARG;
    if (hipMemcpyFromSymbol(out32_host, HIP_SYMBOL(g_kp_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return AOC_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_kp_prof), z, sizeof(z)) != hipSuccess) return AOC_ERR_LAUNCH;
    }
    return AOC_OK;
}

extern "C" int aoc_kmeans_chain_profile_workgroups(unsigned int *out_host, int n_workgroups) {
    if (!out_host || n_workgroups < 1 || n_workgroups > KP_GMAX) return AOC_ERR_INVALID_ARG;
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_kp_wgprof), (size_t)n_workgroups * 16 * sizeof(unsigned int)) != hipSuccess) return AOC_ERR_LAUNCH;
    return AOC_OK;
}

extern "C" int aoc_kmeans_set_grid(int workgroups) {
    if (workgroups < 0) return AOC_ERR_INVALID_ARG;
    g_kp_grid_override = workgroups;
    return AOC_OK;
}
