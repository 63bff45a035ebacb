// Host side of the k-means initialisation (no device code): the initial rows scipy.cluster.vq.kmeans2(..., minit='points') draws for every
// object, level and frame (AEM:268-276: one kmeans2 call per object per level per frame).  scipy's _kpoints takes
// rng.choice(n, size=k, replace=False), which the legacy numpy.random.RandomState implements as permutation(n)[:k]: a Fisher-Yates shuffle of
// arange(n) from the top (_shuffle_raw: for i = n-1 .. 1: j = random_interval(i); swap(x[i], x[j])) on the MT19937 stream, random_interval =
// masked rejection sampling of 32-bit outputs.  An evaluation loop that wants the reference's draws has to run exactly that; numpy spends
// ~17 ns per element on it (2.5 ms for a 150 000-pixel object, once per object, level and frame), which made the closed evaluation loop
// host-bound.  This is the same generator and the same shuffle on an int32 array, 1.5-2x faster than numpy's generic item-size path, for all
// frames / levels / objects of a pool state in one call (the time goes into the rejection sampling itself: ~1.4 MT19937 outputs per element).  The stream is NumPy's frozen legacy stream (NEP 19): tests/test_host_logic.py compares
// rows and the final generator state with numpy itself.
#include <stdint.h>

#include <vector>

#include "aoc_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_MATRIX_A = 0x9908b0dfu, MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu;

struct Mt {
    uint32_t *key;
    int pos;
    void gen() {
        int i;
        uint32_t y;
        for (i = 0; i < MT_N - MT_M; ++i) {
            y = (key[i] & MT_UPPER) | (key[i + 1] & MT_LOWER);
            key[i] = key[i + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MT_MATRIX_A);
        }
        for (; i < MT_N - 1; ++i) {
            y = (key[i] & MT_UPPER) | (key[i + 1] & MT_LOWER);
            key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MT_MATRIX_A);
        }
        y = (key[MT_N - 1] & MT_UPPER) | (key[0] & MT_LOWER);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MT_MATRIX_A);
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos >= MT_N) gen();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    inline uint32_t interval(uint32_t max) {                 // numpy random_interval (max <= 2^32 - 1): masked rejection
        if (max == 0) return 0;
        const uint32_t mask = 0xffffffffu >> __builtin_clz(max);             // max smeared to the right (numpy: mask |= mask >> 1, 2, 4, 8, 16)
        uint32_t v;
        while ((v = next32() & mask) > max) {}
        return v;
    }
};

}  // namespace

extern "C" {

int aoc_kmeans_init_rows_draw(uint32_t *mt_key, int32_t *mt_pos, const int32_t *counts, int n_obj, const int32_t *levels, int n_levels,
                              int n_frames, int kmax, int32_t *rows_out, uint32_t *states_out) {
    if (!mt_key || !mt_pos || !counts || !levels || !rows_out || n_obj < 1 || n_levels < 1 || n_frames < 1 || kmax < 1) return AOC_ERR_INVALID_ARG;
    if (*mt_pos < 0 || *mt_pos > MT_N) return AOC_ERR_INVALID_ARG;
    int32_t cmax = 0;
    for (int i = 0; i < n_obj; ++i) {
        if (counts[i] < 0) return AOC_ERR_INVALID_ARG;
        cmax = counts[i] > cmax ? counts[i] : cmax;
    }
    for (int l = 0; l < n_levels; ++l)
        if (levels[l] < 1 || levels[l] > kmax) return AOC_ERR_INVALID_ARG;
    std::vector<int32_t> perm((size_t)cmax + 1);
    Mt mt{mt_key, *mt_pos};
    const size_t per_frame = (size_t)n_levels * n_obj * kmax;
    for (int f = 0; f < n_frames; ++f) {
        if (states_out) {                                    // the generator as this frame's first draw finds it
            uint32_t *s = states_out + (size_t)f * (MT_N + 1);
            for (int i = 0; i < MT_N; ++i) s[i] = mt.key[i];
            s[MT_N] = (uint32_t)mt.pos;
        }
        int32_t *rows = rows_out + (size_t)f * per_frame;
        for (size_t i = 0; i < per_frame; ++i) rows[i] = 0;
        for (int l = 0; l < n_levels; ++l) {
            int k = levels[l];
            for (int o = 0; o < n_obj; ++o) {
                k = k < counts[o] ? k : counts[o];           // AEM:268: the loop variable is overwritten (sticky within a level)
                if (k <= 0) continue;                        // no kmeans2 call, no draw
                const int32_t n = counts[o];
                for (int32_t i = 0; i < n; ++i) perm[i] = i;
                for (int32_t i = n - 1; i >= 1; --i) {
                    const uint32_t j = mt.interval((uint32_t)i);
                    const int32_t t = perm[j];
                    perm[j] = perm[i];
                    perm[i] = t;
                }
                int32_t *dst = rows + ((size_t)l * n_obj + o) * kmax;
                for (int q = 0; q < k; ++q) dst[q] = perm[q];
            }
        }
    }
    *mt_pos = mt.pos;
    return AOC_OK;
}

}  // extern "C"
