// Exact sequential float32 sums, chunk-parallel: the per-(chunk, feature) arithmetic of the persistent k-means chain
// (kmeans_persistent.hip).  Plain scalar code, compiled for the device AND for the host: tests/csrc/km_core_sim.cpp replays
// whole chains through exactly these functions on the CPU and compares them bit for bit with the literal loop
//     s = 0; for x in members: s = s + x          (scipy _vq.update_cluster_means, float32, row order).
//
// Idea.  While the running sum s stays inside one binade [2^e, 2^(e+1)) an addition is  n <- rne(n + x/u)  on the integer
// mantissa n = s/u (u = 2^(e-23)): integer arithmetic, hence a CHUNK of members can be folded by anybody who knows e -- except
// for exact ties, which need the parity of n (both parities are tracked: they differ by at most one).  e is PREDICTED from an
// any-order prefix of chunk sums; where a prediction says "the sum leaves the binade inside this chunk", the members around the
// predicted crossing are kept as LITERALS (plain float additions for the stitch) and the rest of the chunk is folded in the
// next binade.  The stitch applies records to the exact state and VERIFIES every assumption (binade, no overflow of n); a
// record that does not verify is replaced by the literal additions of its chunk.  A wrong prediction therefore costs time,
// never exactness.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KX_HD __host__ __device__ __forceinline__
#else
#define KX_HD inline
#endif

KX_HD uint32_t kx_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
KX_HD float kx_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---- record header (one uint32 per (chunk, feature))
//   bits  0-1   kind
//   bits  2-9   eA + 128   binade of the A part (0: no A part)
//   bits 10-11  dA + 1     A1 - A0 (increment for odd incoming parity minus the one for even)
//   bits 12-13  dB + 1
//   bits 14-21  eB + 128   binade of the B part (0: no B part)
//   bits 22-27  nlit       literal members between A and B (0..KX_MAX_LIT)
enum { KX_PLAIN = 0, KX_CROSS = 1, KX_SET = 2, KX_UNSAFE = 3 };
constexpr int KX_MAX_LIT = 16;
constexpr int KX_E_MIN = -100, KX_E_MAX = 100;

// ebA / ebB: binade + 128, or 0 when the part is absent (binade -128 does not exist: KX_E_MIN)
KX_HD uint32_t kx_hdr(int kind, int ebA, int dA, int ebB, int dB, int nlit) {
    return (uint32_t)kind | ((uint32_t)ebA << 2) | ((uint32_t)(dA + 1) << 10) | ((uint32_t)(dB + 1) << 12) | ((uint32_t)ebB << 14) |
           ((uint32_t)nlit << 22);
}
KX_HD int kx_hdr_kind(uint32_t h) { return (int)(h & 3u); }
KX_HD int kx_hdr_eA(uint32_t h) { return (int)((h >> 2) & 0xffu); }      // biased (0 = none)
KX_HD int kx_hdr_dA(uint32_t h) { return (int)((h >> 10) & 3u) - 1; }
KX_HD int kx_hdr_dB(uint32_t h) { return (int)((h >> 12) & 3u) - 1; }
KX_HD int kx_hdr_eB(uint32_t h) { return (int)((h >> 14) & 0xffu); }     // biased (0 = none)
KX_HD int kx_hdr_nlit(uint32_t h) { return (int)((h >> 22) & 0x3fu); }

// ---- exact state: the float itself.  n += X in binade `eb` (biased exponent field), X = X0 or X0 + dX by the parity of n.
// Returns false (state untouched) unless s is a positive normal float of that binade and n stays below 2^24.
KX_HD bool kx_apply_int(float &s, int eb, int32_t X0, int dX) {
    const uint32_t bits = kx_f2u(s);
    if ((int)(bits >> 23) != eb) return false;                  // sign bit set or another binade
    const uint32_t n = (bits & 0x7fffffu) | 0x800000u;
    const int32_t X = X0 + ((n & 1u) ? dX : 0);
    if (X < 0) return false;
    const uint32_t n2 = n + (uint32_t)X;
    if (n2 > 0xffffffu || n2 < n) return false;
    s = kx_u2f((bits & 0xff800000u) | (n2 & 0x7fffffu));
    return true;
}

// One record applied to the exact running sum.  lits: the record's literal members, in order.
KX_HD bool kx_apply_record(float &s, uint32_t hdr, int32_t A0, int32_t B0, const float *lits) {
    const int kind = kx_hdr_kind(hdr);
    if (kind == KX_UNSAFE) return false;
    if (kind == KX_SET) {
        if (kx_f2u(s) != 0u) return false;                      // the chunk was summed literally from +0
        s = kx_u2f((uint32_t)A0);
        return true;
    }
    float t = s;
    const int eA = kx_hdr_eA(hdr), eB = kx_hdr_eB(hdr);
    if (eA && !kx_apply_int(t, eA - 128 + 127, A0, kx_hdr_dA(hdr))) return false;
    const int nlit = kx_hdr_nlit(hdr);
    for (int i = 0; i < nlit; ++i) t = t + lits[i];             // literal float additions: exact by definition
    if (eB && !kx_apply_int(t, eB - 128 + 127, B0, kx_hdr_dB(hdr))) return false;
    s = t;
    return true;
}

// ---- composition of consecutive pure-integer records of one binade: (R0, dR) then (A0, dA).
KX_HD void kx_compose(int32_t &R0, int &dR, int32_t A0, int dA) {
    const int32_t R1 = R0 + dR;
    const int32_t n0 = R0 + A0 + ((R0 & 1) ? dA : 0);           // even incoming parity: n parity after the run = R0 & 1
    const int32_t n1 = R1 + A0 + (((1 + R1) & 1) ? dA : 0);     // odd incoming parity
    R0 = n0;
    dR = (int)(n1 - n0);
}

// ---- the fold of one chunk (phase B): a per-lane state machine fed one member at a time.
struct KxFold {
    float inv_u;        // 2^(23 - e) of the part being folded (0: nothing is folded)
    int32_t acc;        // integer increment of the part so far, even incoming parity
    int32_t dvar;       // odd-parity increment minus acc
    uint32_t lim;       // acc + r may reach this without any chance of leaving the binade
    int32_t mode;       // KXM_*
    int32_t eA, eB;     // binades (unbiased; 0 is a legal binade, presence is tracked by nA / nB)
    int32_t A0, dA, nA; // nA: members in front of the window (set when the window opens; KXM_WIN from the start: 0)
    int32_t nlit;
    float n_lo, n_hi;   // plausible range of the exact n at the chunk start, in units of the A binade
    float trk;          // members of A and of the window so far, same units (float: only compared with margins)
    float s;            // KXM_SET: the literal sum
};
enum { KXM_A = 0, KXM_WIN = 1, KXM_B = 2, KXM_DEAD = 3, KXM_SET = 4 };
constexpr float KX_MAGIC = 8388608.0f;          // 2^23: fl(y + 2^23) = 2^23 + rne(y) for 0 <= y < 2^23
constexpr uint32_t KX_MAGIC_BITS = 0x4B000000u;

// a lane that gives up: it folds nothing from here on and never asks the wave for the general step
KX_HD void kx_kill(KxFold &k) { k.mode = KXM_DEAD; k.inv_u = 0.0f; k.lim = 0xffffffffu; }

// P: any-order sum of |x| over the members before this chunk (the prediction of the exact running sum); m_before: their number.
KX_HD void kx_fold_init(KxFold &k, float P, int32_t m_before) {
    k.acc = 0; k.dvar = 0; k.nA = 0; k.nlit = 0; k.A0 = 0; k.dA = 0; k.eA = 0; k.eB = 0; k.trk = 0.0f; k.s = 0.0f;
    k.inv_u = 0.0f; k.lim = 0xffffffffu; k.n_lo = 0.0f; k.n_hi = 0.0f;
    if (P == 0.0f) { k.mode = KXM_SET; return; }               // every earlier member was +-0: the exact state is +0
    const uint32_t pb = kx_f2u(P);
    const int eP = (int)(pb >> 23) - 127;                      // sign bit set (never: sum of |x|) or NaN/inf end up out of range
    if (!(P > 0.0f) || eP < KX_E_MIN + 1 || eP > KX_E_MAX) { kx_kill(k); return; }
    // how far the exact sequential sum may be from the prediction, in ulps of P: both are float sums of the same m_before
    // non-negative numbers; their roundings behave like a random walk.  4 sigma-ish; a miss only costs the slow path.
    float delta = 16.0f + 4.0f * __builtin_sqrtf((float)m_before);
    float nP = kx_u2f((pb & 0x7fffffu) | KX_MAGIC_BITS);       // mantissa as a float in [2^23, 2^24)
    int e0 = eP;
    if (nP - delta < 8388608.0f) { e0 = eP - 1; nP = nP * 2.0f; delta = delta * 2.0f; }   // the exact state may still be one binade below
    k.eA = e0;
    k.inv_u = kx_u2f((uint32_t)(23 - e0 + 127) << 23);
    k.n_lo = nP - delta;
    k.n_hi = nP + delta;
    k.trk = 0.0f;
    if (k.n_hi > 16777214.0f) { k.mode = KXM_WIN; k.lim = 0u; }   // the very first member may already cross
    else { k.mode = KXM_A; k.lim = (uint32_t)(int32_t)(16777215.0f - k.n_hi); }
}

// The branch-free common step: folds x into the part at hand unless something is special about it (value out of range, the
// binade may end here, a literal window is open).  Ties (frac(x/u) == 1/2) are part of it: both incoming parities are tracked.
// `over` lanes are left untouched; a wave commits only when NO lane is over and otherwise calls kx_fold_member for all lanes.
// Lanes that fold nothing ride along: KXM_DEAD / KXM_SET have inv_u = 0 and lim = ~0 (r = 0, no tie), KXM_SET also needs s += x.
struct KxFast {
    int32_t acc, dvar;
    bool over;
};
KX_HD KxFast kx_fold_fast(const KxFold &k, float x) {
    KxFast f;
    const float t = __builtin_fmaf(x, k.inv_u, KX_MAGIC);
    const uint32_t r = kx_f2u(t) - KX_MAGIC_BITS;              // rne(y) for 0 <= y < 2^23; anything else gives r >= 2^23 (as unsigned)
    const float rn = t - KX_MAGIC;
    const float dd = __builtin_fmaf(x, k.inv_u, -rn);          // y - rne(y), exact
    const bool tie = (dd == 0.5f) || (dd == -0.5f);
    const int32_t fl = (int32_t)r - ((dd == -0.5f) ? 1 : 0);   // floor(y) at a tie, rne(y) otherwise
    const int32_t b0 = k.acc + fl;
    const int32_t bump0 = tie ? (b0 & 1) : 0;                  // n + floor(y) odd -> the tie rounds up (even incoming parity)
    const int32_t b1 = b0 + k.dvar + 1;
    const int32_t bump1 = tie ? (b1 & 1) : 0;                  // odd incoming parity
    f.acc = b0 + bump0;
    f.dvar = k.dvar + bump1 - bump0;
    f.over = r >= 0x800000u || (uint32_t)b0 + 2u > k.lim || k.mode == KXM_WIN;
    return f;
}

// The general step, for a wave in which SOME lane is over: the other lanes take their fast result, an over lane opens, continues or
// closes its literal window, or gives up (a second binade end inside one chunk, a value that is not a plain non-negative number, more
// literals than a record holds).  f = kx_fold_fast(k, x); idx: ordinal of this member inside the chunk; lits: this lane's literal buffer
// (KX_MAX_LIT floats, stride lit_stride).
KX_HD void kx_fold_step(KxFold &k, const KxFast &f, float x, int idx, float *lits, int lit_stride) {
    if (!f.over) { k.acc = f.acc; k.dvar = f.dvar; k.s = k.s + x; return; }
    const uint32_t xb = kx_f2u(x);
    const bool plain_value = xb < 0x7f800000u || xb == 0x80000000u;       // +-0 .. largest finite positive
    if (k.mode == KXM_B || !plain_value) { kx_kill(k); return; }
    if (k.mode == KXM_A) {                                     // the binade may end at this member: it is the first literal
        k.A0 = k.acc; k.dA = k.dvar;
        k.nA = idx;
        k.trk = (float)k.acc;
        k.acc = 0; k.dvar = 0;
        k.mode = KXM_WIN;
    }
    if (k.nlit >= KX_MAX_LIT) { kx_kill(k); return; }
    lits[k.nlit * lit_stride] = x;
    ++k.nlit;
    k.trk = k.trk + x * k.inv_u;
    if (k.n_hi + k.trk >= 33554000.0f) { kx_kill(k); return; }   // two binades at once: not handled here
    if (k.n_lo + k.trk >= 16777216.0f) {
        // even the lowest plausible state is past 2^24 now: the rest of the chunk is folded one binade up
        k.eB = k.eA + 1;
        const float n1_hi = (k.n_hi + k.trk) * 0.5f + 2.0f;
        if (k.eB > KX_E_MAX || n1_hi > 16777214.0f) { kx_kill(k); return; }
        k.inv_u = k.inv_u * 0.5f;
        k.acc = 0; k.dvar = 0;
        k.lim = (uint32_t)(int32_t)(16777215.0f - n1_hi);
        k.mode = KXM_B;
    }
}
KX_HD void kx_fold_member(KxFold &k, float x, int idx, float *lits, int lit_stride) {
    const KxFast f = kx_fold_fast(k, x);
    kx_fold_step(k, f, x, idx, lits, lit_stride);
}

// total: members of the chunk.  -> header, A0 (KX_SET: the bits of the literal sum), B0.  The literals stay where kx_fold_member put them.
KX_HD uint32_t kx_fold_finish(const KxFold &k, int total, int32_t &A0, int32_t &B0) {
    A0 = 0; B0 = 0;
    const int nB = total - k.nA - k.nlit;
    switch (k.mode) {
    case KXM_SET: A0 = (int32_t)kx_f2u(k.s); return kx_hdr(KX_SET, 0, 0, 0, 0, 0);
    case KXM_A: A0 = k.acc; return kx_hdr(KX_PLAIN, k.eA + 128, k.dvar, 0, 0, 0);
    case KXM_WIN: A0 = k.A0; return kx_hdr(KX_CROSS, k.nA ? k.eA + 128 : 0, k.dA, 0, 0, k.nlit);
    case KXM_B: A0 = k.A0; B0 = k.acc; return kx_hdr(KX_CROSS, k.nA ? k.eA + 128 : 0, k.dA, nB > 0 ? k.eB + 128 : 0, k.dvar, k.nlit);
    default: return kx_hdr(KX_UNSAFE, 0, 0, 0, 0, 0);
    }
}

// ---- phase C: consecutive PLAIN records of one binade merge into a run (the stitch applies a run like one record)
struct KxRun {
    int32_t eb;         // binade + 128 (0: empty run)
    int32_t R0;
    int32_t dR;
};
KX_HD uint32_t kx_run_hdr(const KxRun &r) { return (uint32_t)r.eb | ((uint32_t)(r.dR + 1) << 8); }
KX_HD KxRun kx_run_unpack(uint32_t w, int32_t R0) {
    KxRun r;
    r.eb = (int32_t)(w & 0xffu);
    r.dR = (int32_t)((w >> 8) & 3u) - 1;
    r.R0 = R0;
    return r;
}
// true: the record went into the run.  false: the caller closes the run in front of this record.
KX_HD bool kx_run_merge(KxRun &run, uint32_t hdr, int32_t A0) {
    if (kx_hdr_kind(hdr) == KX_SET && A0 == 0) return true;    // a chunk of zeros in front of everything else: +0 stays +0
    if (kx_hdr_kind(hdr) != KX_PLAIN) return false;
    const int eb = kx_hdr_eA(hdr);
    if (run.eb == 0) { run.eb = eb; run.R0 = A0; run.dR = kx_hdr_dA(hdr); return true; }
    if (run.eb != eb || run.R0 > 0x3fffffff) return false;
    int d = run.dR;
    kx_compose(run.R0, d, A0, kx_hdr_dA(hdr));
    run.dR = d;
    return true;
}
KX_HD bool kx_apply_run(float &s, const KxRun &run) {
    if (run.eb == 0) return true;
    return kx_apply_int(s, run.eb - 1, run.R0, run.dR);
}
