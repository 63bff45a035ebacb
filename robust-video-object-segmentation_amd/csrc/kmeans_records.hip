// Ordered per-cluster sums of the launch-per-phase k-means pipeline, "records" variant: the chunk-parallel exact sums of km_exact_core.h
// on the ordered member lists that km_blockscan / km_scatter produce (labels_kmeans.hip).  Per Lloyd iteration, after the scatter:
//   plan    chunk table: a chunk = 64 consecutive members of one cluster's list (every chunk but a cluster's last is full);
//   heads   the first 512 members of every cluster summed literally (the sum crosses a binade at every doubling there), next to the
//           any-order sums of |x| of all later chunks;
//   prefix  per (cluster, feature) the exclusive prefix of those sums: the PREDICTION of the exact running sum in front of every chunk;
//   fold    per (chunk, 64 features) a record: integer increments in the predicted binade(s), the members around a predicted binade
//           crossing as literals (lanes = features, rows staged in LDS, members folded one after another);
//   merge   consecutive pure-integer records of a part (64 chunks) collapse into runs;
//   stitch  one wave per (cluster, 64 features) walks head state -> parts -> runs -> records with the exact float state, verifies every
//           assumption and replaces what does not verify by the literal additions of that chunk's rows.  centroid = sum / count.
// Bit-identical to the sequential float32 sums for any input (tests: the k-means suite; CPU replay: tests/test_km_exact_core.py).
#include "aoc_common.h"
#include "km_exact_core.h"

#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int KR_C = 100;
constexpr int KR_CH = 64;             // members per chunk
constexpr int KR_HEADC = 8;           // chunks of the literal head (512 members)
constexpr int KR_REC = 6;             // words per (chunk, feature): hdr, A0, B0, literal | literal offset, run hdr, run R0
constexpr int KR_LITCAP = 4096;       // literal pool per fold workgroup (floats)
constexpr int KR_EV = 2;              // records per (part, feature) the stitch fetches one part ahead

__host__ __device__ __forceinline__ int kr_seg_chunk_base(int seg_beg, int s, int kmax) { return seg_beg / KR_CH + s * (kmax + 1); }
__device__ __forceinline__ unsigned long long kr_below(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }

struct KrArgs {
    const float *pool;
    const int32_t *seg_off, *seg_k, *counts, *cbase;
    const uint32_t *moff;             // byte offsets of the members' pool rows, per cluster in row order
    int n_seg, kmax, nch_cap, pmax, dbg;
    float *centroids;
    int32_t *cc64;                    // [cluster] first chunk id
    int4 *cdesc;                      // [chunk id] {first member (index into moff), members (0: unused id), cluster, chunk index inside the cluster}
    float *hstate, *habs;             // [cluster][C] exact sum / sum of |x| of the head
    float *PT;                        // [C][nch_cap] sum of |x| per tail chunk -> exclusive prefix within the cluster
    uint32_t *rec;                    // [chunk id][KR_REC][C]
    float *litpool;                   // [fold workgroup][KR_LITCAP]
    uint32_t *ppost;                  // [cluster][pmax][2][C] last run of the part
    unsigned long long *pnp;          // [cluster][pmax][C] chunks of the part whose records the stitch has to look at
};

// ---- plan: chunk ids of a segment's clusters (one workgroup per segment)
__global__ __launch_bounds__(256) void kr_plan_kernel(KrArgs a) {
    __shared__ int32_t lbase[AOC_MAX_CLUSTERS + 1];
    const int s = blockIdx.x, kmax = a.kmax;
    const int beg = a.seg_off[s];
    const int cb = kr_seg_chunk_base(beg, s, kmax);
    const int end = (s + 1 < a.n_seg) ? kr_seg_chunk_base(a.seg_off[s + 1], s + 1, kmax) : a.nch_cap;
    if (threadIdx.x == 0) {
        int ch = 0;
        const int k = a.seg_k[s];
        for (int j = 0; j < kmax; ++j) {
            lbase[j] = ch;
            a.cc64[s * kmax + j] = cb + ch;
            if (j < k) ch += (a.counts[s * kmax + j] + KR_CH - 1) / KR_CH;
        }
        lbase[kmax] = ch;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < end - cb; i += 256) {
        int4 d = make_int4(0, 0, -1, 0);
        if (i < lbase[kmax]) {
            int j = 0;
            while (j + 1 < kmax && lbase[j + 1] <= i) ++j;
            const int oc = s * kmax + j, c = i - lbase[j];
            d = make_int4(beg + a.cbase[oc] + c * KR_CH, min(KR_CH, a.counts[oc] - c * KR_CH), oc, c);
        }
        a.cdesc[cb + i] = d;
    }
}

// ---- heads + chunk sums: wave = task
__global__ __launch_bounds__(256) void kr_heads_sums_kernel(KrArgs a) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax, nc = a.n_seg * kmax;
    const int task = blockIdx.x * 4 + wave;
    const char *poolb = reinterpret_cast<const char *>(a.pool);
    if (task < nc * 2) {
        // literal head of (cluster, feature half): members 0 .. min(cnt, 512) - 1 in order
        const int oc = task >> 1, h = task & 1;
        const int s = oc / kmax, j = oc - s * kmax;
        if (j >= a.seg_k[s]) return;
        const int f = 64 * h + lane;
        const bool fvalid = f < KR_C;
        const int fc = fvalid ? f : KR_C - 1;
        const int cnt = a.counts[oc];
        const int nh = min(cnt, KR_HEADC * KR_CH);
        if (nh == 0) {
            if (fvalid) { a.hstate[(size_t)oc * KR_C + f] = 0.0f; a.habs[(size_t)oc * KR_C + f] = 0.0f; }
            return;
        }
        const uint32_t *list = a.moff + a.seg_off[s] + a.cbase[oc];
        float sv = 0.0f, sa = 0.0f;
        // member offsets of the whole head (lane = member, eight loads), then the rows 32 at a time, the next 32 in flight under the additions
        uint32_t offv[KR_HEADC];
#pragma unroll
        for (int b = 0; b < KR_HEADC; ++b) offv[b] = list[min(b * 64 + lane, max(nh - 1, 0))];
        const char *pb = poolb + fc * 4;
        float xa[32], xb[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) xa[u] = *reinterpret_cast<const float *>(pb + __builtin_amdgcn_readlane(offv[0], u));
#pragma unroll
        for (int b = 0; b < KR_HEADC; ++b) {
            if (b * 64 < nh) {
#pragma unroll
                for (int u = 0; u < 32; ++u) xb[u] = *reinterpret_cast<const float *>(pb + __builtin_amdgcn_readlane(offv[b], 32 + u));
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (b * 64 + u < nh) { sv = sv + xa[u]; sa += fabsf(xa[u]); }
                if (b + 1 < KR_HEADC) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) xa[u] = *reinterpret_cast<const float *>(pb + __builtin_amdgcn_readlane(offv[b + 1 < KR_HEADC ? b + 1 : b], u));
                }
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (b * 64 + 32 + u < nh) { sv = sv + xb[u]; sa += fabsf(xb[u]); }
            }
        }
        if (fvalid) { a.hstate[(size_t)oc * KR_C + f] = sv; a.habs[(size_t)oc * KR_C + f] = sa; }
        return;
    }
    // any-order sum of |x| of a tail chunk
    const int t2 = task - nc * 2;
    const int g = t2 >> 1, h = t2 & 1;
    if (g >= a.nch_cap) return;
    const int4 d = a.cdesc[g];
    const int n = d.y;
    if (n == 0 || d.w < KR_HEADC) return;
    const int f = 64 * h + lane;
    const bool fvalid = f < KR_C;
    const int fc = fvalid ? f : KR_C - 1;
    const uint32_t offl = a.moff[d.x + min(lane, n - 1)];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int u0 = 0; u0 < n; u0 += 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = *reinterpret_cast<const float *>(poolb + __builtin_amdgcn_readlane(offl, min(u0 + u, 63)) + fc * 4);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] += (u0 + u < n) ? fabsf(x[u]) : 0.0f;
    }
    if (fvalid) a.PT[(size_t)f * a.nch_cap + g] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// ---- prefix: wave per (cluster, feature), lanes = tail chunks
__global__ __launch_bounds__(256) void kr_prefix_kernel(KrArgs a) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax;
    const int task = blockIdx.x * 4 + wave;
    const int oc = task / KR_C, f = task - oc * KR_C;
    if (oc >= a.n_seg * kmax) return;
    const int s = oc / kmax, j = oc - s * kmax;
    if (j >= a.seg_k[s]) return;
    const int nch = (a.counts[oc] + KR_CH - 1) / KR_CH;
    float *p = a.PT + (size_t)f * a.nch_cap + a.cc64[oc];
    float carry = 0.0f;
    for (int c0 = KR_HEADC; c0 < nch; c0 += 64) {
        const int c = c0 + lane;
        const float v = (c < nch) ? p[c] : 0.0f;
        float incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const float t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (c < nch) p[c] = carry + (incl - v);
        carry += __shfl(incl, 63);
    }
}

// ---- fold: wave per tail chunk; lane l < 50 folds features l and 50 + l as one packed pair (v_pk_fma_f32 and friends).  Members come
// sixteen at a time into registers (the next sixteen in flight) and are folded FOUR at a time without looking: the integer increments are
// summed blindly while three maxima remember whether any of the four was special -- an increment out of range (a value that is not a plain
// number below the running sum), a remainder of exactly one half (a tie), the sum too close to the end of its binade; only then the four
// are taken again one by one through kx_fold_fast / kx_fold_step from the saved state.  Five packed operations per member instead of fifty.
typedef float kr_f2 __attribute__((ext_vector_type(2)));
constexpr int KR_HALF = KR_C / 2;
__global__ __launch_bounds__(256) void kr_fold_kernel(KrArgs a) {
    __shared__ int litcnt;
    __shared__ float lits_s[4][2][KX_MAX_LIT * 64];
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    float *lits0 = &lits_s[wave][0][lane], *lits1 = &lits_s[wave][1][lane];
    if (threadIdx.x == 0) litcnt = 0;
    __syncthreads();
    const int g = blockIdx.x * 4 + wave;
    if (g >= a.nch_cap) return;
    const int4 d = a.cdesc[g];
    const int n = d.y, oc = d.z, c = d.w;
    if (n == 0 || c < KR_HEADC) return;
    const bool lvalid = lane < KR_HALF;
    const int lc = lvalid ? lane : KR_HALF - 1;
    const int f0 = lc, f1 = KR_HALF + lc;
    const uint32_t offl = a.moff[d.x + min(lane, n - 1)];
    const float P0 = a.habs[(size_t)oc * KR_C + f0] + a.PT[(size_t)f0 * a.nch_cap + g];
    const float P1 = a.habs[(size_t)oc * KR_C + f1] + a.PT[(size_t)f1 * a.nch_cap + g];
    const char *pb0 = reinterpret_cast<const char *>(a.pool) + f0 * 4, *pb1 = reinterpret_cast<const char *>(a.pool) + f1 * 4;
    kr_f2 xa[16], xb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const uint32_t o = __builtin_amdgcn_readlane(offl, u);      // (lanes past n repeat member n - 1: never used)
        xa[u].x = *reinterpret_cast<const float *>(pb0 + o);
        xa[u].y = *reinterpret_cast<const float *>(pb1 + o);
    }
    KxFold k0, k1;
    kx_fold_init(k0, P0, c * KR_CH);
    kx_fold_init(k1, P1, c * KR_CH);
    const kr_f2 magic = {KX_MAGIC, KX_MAGIC};
#pragma unroll 1
    for (int b = 0; b < KR_CH / 16; ++b) {
        if (b * 16 >= n || (a.dbg & 1)) break;
        if ((b + 1) * 16 < n) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t o = __builtin_amdgcn_readlane(offl, min((b + 1) * 16 + u, 63));
                xb[u].x = *reinterpret_cast<const float *>(pb0 + o);
                xb[u].y = *reinterpret_cast<const float *>(pb1 + o);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i0 = b * 16 + q * 4;
            if (i0 < n) {
                const int nq = min(4, n - i0);
                // ---- blind: four members
                const kr_f2 inv_u = {k0.inv_u, k1.inv_u};
                uint32_t acc0 = (uint32_t)k0.acc, acc1 = (uint32_t)k1.acc, mx = 0u;
                kr_f2 ss = {k0.s, k1.s};
                float md = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kr_f2 xv = xa[q * 4 + e];
                    if (e >= nq) xv = (kr_f2){0.0f, 0.0f};              // (adds nothing: r = 0, no remainder)
                    const kr_f2 t = __builtin_elementwise_fma(xv, inv_u, magic);
                    const kr_f2 rn = t - magic;
                    const kr_f2 dd = __builtin_elementwise_fma(xv, inv_u, -rn);
                    const uint32_t r0 = kx_f2u(t.x) - KX_MAGIC_BITS, r1 = kx_f2u(t.y) - KX_MAGIC_BITS;
                    acc0 += r0;
                    acc1 += r1;
                    mx = max(mx, max(r0, r1));
                    md = __builtin_fmaxf(md, __builtin_fmaxf(__builtin_fabsf(dd.x), __builtin_fabsf(dd.y)));
                    ss = ss + xv;
                }
                const bool special = mx >= 0x800000u || md == 0.5f || acc0 + 2u > k0.lim || acc1 + 2u > k1.lim || k0.mode == KXM_WIN || k1.mode == KXM_WIN;
                if (!__any(special)) {
                    k0.acc = (int32_t)acc0; k1.acc = (int32_t)acc1;
                    k0.s = ss.x; k1.s = ss.y;
                } else {
                    // ---- the four again, one by one (ties, windows, literals, give-ups)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e >= nq) break;
                        const kr_f2 xv = xa[q * 4 + e];
                        const KxFast s0 = kx_fold_fast(k0, xv.x), s1 = kx_fold_fast(k1, xv.y);
                        if (!__any(s0.over || s1.over)) {
                            k0.acc = s0.acc; k0.dvar = s0.dvar; k0.s = k0.s + xv.x;
                            k1.acc = s1.acc; k1.dvar = s1.dvar; k1.s = k1.s + xv.y;
                        } else {
                            kx_fold_step(k0, s0, xv.x, i0 + e, lits0, 64);
                            kx_fold_step(k1, s1, xv.y, i0 + e, lits1, 64);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) xa[u] = xb[u];
    }
    // ---- records of the two features
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const float *lits = half ? lits1 : lits0;
        const int f = half ? f1 : f0;
        int32_t A0, B0;
        uint32_t hdr = half ? kx_fold_finish(k1, n, A0, B0) : kx_fold_finish(k0, n, A0, B0);
        uint32_t w3 = 0u;
        const int nlit = kx_hdr_nlit(hdr);
        if (nlit == 1) w3 = kx_f2u(lits[0]);
        if (nlit > 1 && lvalid) {
            const int off = atomicAdd(&litcnt, nlit);
            if (off + nlit > KR_LITCAP) {
                hdr = kx_hdr(KX_UNSAFE, 0, 0, 0, 0, 0);
            } else {
                float *dst = a.litpool + (size_t)blockIdx.x * KR_LITCAP + off;
                for (int q = 0; q < nlit; ++q) dst[q] = lits[q * 64];
                w3 = (uint32_t)(blockIdx.x * KR_LITCAP + off);
            }
        }
        if (lvalid) {
            uint32_t *r = a.rec + (size_t)g * KR_REC * KR_C + f;
            r[0] = hdr;
            r[KR_C] = (uint32_t)A0;
            r[2 * KR_C] = (uint32_t)B0;
            r[3 * KR_C] = w3;
        }
    }
}

// ---- merge: wave per (cluster, part, feature half)
__global__ __launch_bounds__(256) void kr_merge_kernel(KrArgs a) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax;
    const int task = blockIdx.x * 4 + wave;
    const int h = task & 1, pp = (task >> 1) % a.pmax, oc = (task >> 1) / a.pmax;
    if (oc >= a.n_seg * kmax) return;
    const int s = oc / kmax, j = oc - s * kmax;
    if (j >= a.seg_k[s]) return;
    const int nch = (a.counts[oc] + KR_CH - 1) / KR_CH;
    const int c_begin = max(64 * pp, KR_HEADC), c_end = min(64 * pp + 64, nch);
    if (c_begin >= c_end) return;
    const int f = 64 * h + lane;
    const bool fvalid = f < KR_C;
    const int fc = fvalid ? f : KR_C - 1;
    const int g0 = a.cc64[oc];
    KxRun run{0, 0, 0};
    unsigned long long np = 0ull;
    for (int c0 = c_begin; c0 < c_end; c0 += 32) {
        uint32_t hd[32], w1[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const uint32_t *r = a.rec + (size_t)(g0 + min(c0 + u, c_end - 1)) * KR_REC * KR_C + fc;
            hd[u] = r[0]; w1[u] = r[KR_C];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            if (c0 + u < c_end && !kx_run_merge(run, hd[u], (int32_t)w1[u])) {
                if (fvalid) {
                    uint32_t *r = a.rec + (size_t)(g0 + c0 + u) * KR_REC * KR_C + fc;
                    r[4 * KR_C] = kx_run_hdr(run);
                    r[5 * KR_C] = (uint32_t)run.R0;
                }
                np |= 1ull << (c0 + u - 64 * pp);
                run = KxRun{0, 0, 0};
            }
        }
    }
    if (fvalid) {
        uint32_t *po = a.ppost + ((size_t)oc * a.pmax + pp) * 2 * KR_C + f;
        po[0] = kx_run_hdr(run);
        po[KR_C] = (uint32_t)run.R0;
        a.pnp[((size_t)oc * a.pmax + pp) * KR_C + f] = np;
    }
}

// ---- stitch: wave per (cluster, feature half)
__global__ __launch_bounds__(256) void kr_stitch_kernel(KrArgs a) {
    const int lane = aoc_lane(), wave = threadIdx.x >> 6;
    const int kmax = a.kmax;
    const int task = blockIdx.x * 4 + wave;
    const int h = task & 1, oc = task >> 1;
    if (oc >= a.n_seg * kmax) return;
    const int s = oc / kmax, j = oc - s * kmax;
    if (j >= a.seg_k[s]) return;
    const int cnt = a.counts[oc];
    if (cnt == 0) return;                                           // vq.py:820-823: an empty cluster keeps its centroid
    const int f = 64 * h + lane;
    const bool fvalid = f < KR_C;
    const int fc = fvalid ? f : KR_C - 1;
    const int nch = (cnt + KR_CH - 1) / KR_CH;
    const int nparts = (nch + 63) / 64;
    const int g0 = a.cc64[oc];
    const uint32_t *list = a.moff + a.seg_off[s] + a.cbase[oc];
    const char *poolb = reinterpret_cast<const char *>(a.pool);
    float sv = a.hstate[(size_t)oc * KR_C + fc];
    const size_t pk0 = (size_t)oc * a.pmax;
    unsigned long long np_c = 0ull, np_1 = 0ull, np_2 = 0ull;
    uint32_t ph_c = 0u, ph_1 = 0u, ph_2 = 0u, pR_c = 0u, pR_1 = 0u, pR_2 = 0u;
    uint32_t ev_c[KR_EV][KR_REC], ev_1[KR_EV][KR_REC];
#pragma unroll
    for (int e = 0; e < KR_EV; ++e)
#pragma unroll
        for (int i = 0; i < KR_REC; ++i) { ev_c[e][i] = 0u; ev_1[e][i] = 0u; }
    const bool tail = nch > KR_HEADC;
#define KR_PART_A(pp_, np_, ph_, pR_)                                                      \
    if (tail && (pp_) < nparts) {                                                          \
        np_ = a.pnp[(pk0 + (pp_)) * KR_C + fc];                                            \
        ph_ = a.ppost[(pk0 + (pp_)) * 2 * KR_C + fc];                                      \
        pR_ = a.ppost[(pk0 + (pp_)) * 2 * KR_C + KR_C + fc];                               \
    }
#define KR_PART_E(pp_, np_, ev_)                                                           \
    if (tail && (pp_) < nparts) {                                                          \
        unsigned long long m_ = (np_);                                                     \
        _Pragma("unroll") for (int e = 0; e < KR_EV; ++e) {                                \
            const int pos_ = m_ ? __builtin_ctzll(m_) : 0;                                 \
            const uint32_t *r_ = a.rec + (size_t)(g0 + 64 * (pp_) + pos_) * KR_REC * KR_C + fc; \
            if (__any(m_ != 0ull)) {                                                       \
                _Pragma("unroll") for (int i = 0; i < KR_REC; ++i) ev_[e][i] = r_[(size_t)i * KR_C]; \
            }                                                                              \
            m_ &= m_ - 1;                                                                  \
        }                                                                                  \
    }
    KR_PART_A(0, np_c, ph_c, pR_c)
    KR_PART_A(1, np_1, ph_1, pR_1)
    KR_PART_E(0, np_c, ev_c)
    for (int pp = 0; tail && pp < nparts; ++pp) {
        KR_PART_A(pp + 2, np_2, ph_2, pR_2)
        KR_PART_E(pp + 1, np_1, ev_1)
        unsigned long long np = np_c;
        const uint32_t post_h = ph_c, post_R = pR_c;
        uint32_t evv[KR_EV][KR_REC];
#pragma unroll
        for (int e = 0; e < KR_EV; ++e)
#pragma unroll
            for (int i = 0; i < KR_REC; ++i) { evv[e][i] = ev_c[e][i]; ev_c[e][i] = ev_1[e][i]; }
        np_c = np_1; ph_c = ph_1; pR_c = pR_1;
        np_1 = np_2; ph_1 = ph_2; pR_1 = pR_2;
        const int c_lo = max(64 * pp, KR_HEADC) - 64 * pp, c_hi = min(64, nch - 64 * pp);      // positions of the part that hold tail chunks
        if (c_lo >= c_hi) continue;
        const unsigned long long pmask = kr_below(c_hi) & ~kr_below(c_lo);
        bool norun = false, stuck = false, post_done = false;
        int done = c_lo, spos = 64, evi = 0;
        for (;;) {
            for (;;) {
                const int pos = (!stuck && np != 0ull) ? __builtin_ctzll(np) : 64;
                const bool act = pos < 64;
                if (!__any(act)) break;
                const bool inl = act && !norun && evi < KR_EV;
                bool renew = false;
                uint32_t w[KR_REC];
#pragma unroll
                for (int i = 0; i < KR_REC; ++i) w[i] = 0u;
                if (__any(act && !inl)) {
                    if (act && !inl) {
                        const uint32_t *r = a.rec + (size_t)(g0 + 64 * pp + pos) * KR_REC * KR_C + fc;
#pragma unroll
                        for (int i = 0; i < KR_REC; ++i) w[i] = r[(size_t)i * KR_C];
                    }
                }
                if (inl) {
#pragma unroll
                    for (int e = 0; e < KR_EV; ++e)
                        if (evi == e) {
#pragma unroll
                            for (int i = 0; i < KR_REC; ++i) w[i] = evv[e][i];
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
                        norun = true;                               // a chunk of the run did not happen as predicted: the rest of the part record by record
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
                                for (int q = 0; q < nlit; ++q) t = t + a.litpool[w3 + q];
                            if (ok && eB) ok = kx_apply_int(t, eB - 1, B0, kx_hdr_dB(hdr));
                        }
                        if (ok) { sv = t; np &= np - 1; done = pos + 1; ++evi; }
                        else { stuck = true; spos = pos; }
                    }
                }
                if (renew) np = pmask & ~kr_below(done);
            }
            // stuck lanes: the lowest stuck chunk is summed literally from its rows (lanes = features, coalesced)
            if (__any(stuck)) {
                int pmin = stuck ? spos : 64;
                for (int o = 32; o > 0; o >>= 1) pmin = min(pmin, __shfl_xor(pmin, o));
                const int c = 64 * pp + pmin;
                const int n = min(KR_CH, cnt - c * KR_CH);
                const uint32_t offl = list[c * KR_CH + min(lane, n - 1)];
                const bool mine = stuck && spos == pmin;
#pragma unroll 1
                for (int u0 = 0; u0 < n; u0 += 16) {
                    float x[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) x[u] = *reinterpret_cast<const float *>(poolb + __builtin_amdgcn_readlane(offl, min(u0 + u, 63)) + fc * 4);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (u0 + u < n && mine) sv = sv + x[u];
                }
                if (mine) { stuck = false; np &= ~(1ull << pmin); done = pmin + 1; ++evi; }
                continue;
            }
            bool again = false;
            if (!norun && !post_done) {
                post_done = true;
                const KxRun run = kx_run_unpack(post_h, (int32_t)post_R);
                float t = sv;
                if (kx_apply_run(t, run)) sv = t;
                else { norun = true; again = true; }
            }
            if (!__any(again)) break;
            if (again) np = pmask & ~kr_below(done);
        }
    }
    if (fvalid) a.centroids[(size_t)oc * KR_C + f] = sv / (float)cnt;
}

struct KrLayout {
    size_t cc64, cdesc, hstate, habs, PT, rec, litpool, ppost, pnp, total;
    int nch_cap, pmax, fold_wgs;
};
KrLayout kr_layout(int64_t cap, int n_seg, int kmax, int64_t seg_bound) {
    KrLayout l;
    l.nch_cap = (int)(cap / KR_CH) + n_seg * (kmax + 1) + 2;
    const int64_t sb = (seg_bound > 0 && seg_bound < cap) ? seg_bound : cap;
    l.pmax = (int)((sb / KR_CH + 1 + 63) / 64) + 1;
    l.fold_wgs = (l.nch_cap + 3) / 4;
    const size_t nc = (size_t)n_seg * kmax;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += aoc_align_up(bytes, 256); return at; };
    l.cc64 = take(nc * 4);
    l.cdesc = take((size_t)l.nch_cap * 16);
    l.hstate = take(nc * KR_C * 4);
    l.habs = take(nc * KR_C * 4);
    l.PT = take((size_t)KR_C * l.nch_cap * 4);
    l.rec = take((size_t)l.nch_cap * KR_REC * KR_C * 4);
    l.litpool = take((size_t)l.fold_wgs * KR_LITCAP * 4);
    l.ppost = take(nc * l.pmax * 2 * KR_C * 4);
    l.pnp = take(nc * l.pmax * KR_C * 8);
    l.total = o;
    return l;
}

}  // namespace

bool aoc_kr_supported(int C, int kmax) { return C == KR_C && kmax >= 1 && kmax <= AOC_MAX_CLUSTERS; }
size_t aoc_kr_workspace_bytes(int64_t rows_capacity, int n_seg, int kmax) { return kr_layout(rows_capacity, n_seg, kmax, 0).total; }

// The ordered sums of one Lloyd iteration from the member lists (counts / cbase / moff of km_blockscan + km_scatter) -> centroids.
int aoc_kr_sums(const float *pool, const int32_t *seg_offsets, const int32_t *seg_k, const int32_t *counts, const int32_t *cbase, const uint32_t *moff,
                int n_seg, int kmax, int64_t rows_capacity, int64_t seg_bound, float *centroids, void *workspace, hipStream_t st) {
    const KrLayout l = kr_layout(rows_capacity, n_seg, kmax, 0);
    const KrLayout lb = kr_layout(rows_capacity, n_seg, kmax, seg_bound);
    char *w = static_cast<char *>(workspace);
    KrArgs a;
    a.pool = pool; a.seg_off = seg_offsets; a.seg_k = seg_k; a.counts = counts; a.cbase = cbase; a.moff = moff;
    a.n_seg = n_seg; a.kmax = kmax; a.nch_cap = l.nch_cap; a.pmax = std::min(l.pmax, lb.pmax);
    static const int dbg = getenv("AOC_KR_DEBUG") ? atoi(getenv("AOC_KR_DEBUG")) : 0;      // developer switch (timing experiments; wrong results)
    a.dbg = dbg;
    a.centroids = centroids;
    a.cc64 = reinterpret_cast<int32_t *>(w + l.cc64);
    a.cdesc = reinterpret_cast<int4 *>(w + l.cdesc);
    a.hstate = reinterpret_cast<float *>(w + l.hstate);
    a.habs = reinterpret_cast<float *>(w + l.habs);
    a.PT = reinterpret_cast<float *>(w + l.PT);
    a.rec = reinterpret_cast<uint32_t *>(w + l.rec);
    a.litpool = reinterpret_cast<float *>(w + l.litpool);
    a.ppost = reinterpret_cast<uint32_t *>(w + l.ppost);
    a.pnp = reinterpret_cast<unsigned long long *>(w + l.pnp);
    const int nc = n_seg * kmax;
    hipLaunchKernelGGL(kr_plan_kernel, dim3(n_seg), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_heads_sums_kernel, dim3((nc * 2 + l.nch_cap * 2 + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_prefix_kernel, dim3((nc * KR_C + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_fold_kernel, dim3(l.fold_wgs), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_merge_kernel, dim3((nc * a.pmax * 2 + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(kr_stitch_kernel, dim3((nc * 2 + 3) / 4), dim3(256), 0, st, a);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}
