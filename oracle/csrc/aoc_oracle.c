/*
 * aoc_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Plain-C restatement of the one third-party algorithm on the hot path whose
 * *arithmetic order* matters for parity: scipy.cluster.vq.kmeans2 as called at
 *   /root/reference/AOC-Net/adaptive_embedding_for_matching.py:276
 *     kmeans2(X, K, minit='points', iter=20)
 * scipy is not vendored under /root/reference; the oracle is pinned against the
 * scipy 1.15.3 installed in this image (tests/test_oracle_kmeans.py compares this
 * file bit-for-bit with scipy.cluster.vq.vq / kmeans2, labels AND centroids).
 *
 * Published algorithm being restated (scipy/cluster/_vq.pyx, vq.py:646-826):
 *   _vq.vq (nfeat >= 5 path):
 *     obs_sqr[i]  = sum_t obs[i,t]^2        sequential float32, multiply then add
 *     code_sqr[j] = sum_t code[j,t]^2       idem
 *     M = -2 * obs @ code^T                 BLAS sgemm; OpenBLAS accumulates each
 *                                           element as ONE k-ordered fmaf chain from 0,
 *                                           alpha applied last (verified bit-exact here)
 *     dist = (M[i,j] + obs_sqr[i]) + code_sqr[j]
 *     label[i] = first j with strictly smaller dist (ties -> lowest j)
 *   _vq.update_cluster_means:
 *     per-cluster sums accumulated sequentially in observation order (float32),
 *     then divided by (float)count; empty cluster keeps its previous centroid
 *     (vq.py:820-823, missing='warn').
 *   kmeans2 returns the code book AFTER the last update and the labels of the
 *   last assignment (made against the code book BEFORE that update).
 *
 * Build: see oracle/Makefile  (-ffp-contract=off is REQUIRED: only the explicit
 * fmaf() calls may fuse; -mfma makes fmaf() a single instruction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float sq_norm_seq(const float *p, int d) {
    float s = 0.0f;
    for (int t = 0; t < d; ++t) {
        float prod = p[t] * p[t];
        s = s + prod;
    }
    return s;
}

/* scipy _vq.vq: labels + squared low distance (before scipy's sqrt/clamp). */
void aoc_oracle_vq(const float *obs, const float *code, int n, int k, int d,
                   int32_t *labels, float *low_dist_sq) {
    float *code_sqr = (float *)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    for (int j = 0; j < k; ++j) code_sqr[j] = sq_norm_seq(code + (size_t)j * d, d);
    for (int i = 0; i < n; ++i) {
        const float *x = obs + (size_t)i * d;
        float obs_sqr = sq_norm_seq(x, d);
        float low = INFINITY;
        int32_t best = 0;
        for (int j = 0; j < k; ++j) {
            const float *c = code + (size_t)j * d;
            float acc = 0.0f;
            for (int t = 0; t < d; ++t) acc = fmaf(x[t], c[t], acc);
            float m = -2.0f * acc;
            float dist = (m + obs_sqr) + code_sqr[j];
            if (dist < low) {
                low = dist;
                best = j;
            }
        }
        labels[i] = best;
        if (low_dist_sq) low_dist_sq[i] = low;
    }
    free(code_sqr);
}

/* scipy _vq.update_cluster_means + the empty-cluster rule of vq.py:820-823.
 * code is updated in place; counts[j] receives the member count. */
void aoc_oracle_update_means(const float *obs, const int32_t *labels, int n, int k, int d,
                             float *code, int32_t *counts) {
    float *sum = (float *)calloc((size_t)k * d + 1, sizeof(float));
    memset(counts, 0, sizeof(int32_t) * (size_t)k);
    for (int i = 0; i < n; ++i) {
        int32_t l = labels[i];
        const float *x = obs + (size_t)i * d;
        float *s = sum + (size_t)l * d;
        for (int t = 0; t < d; ++t) s[t] += x[t];
        counts[l] += 1;
    }
    for (int j = 0; j < k; ++j) {
        if (counts[j] > 0) {
            float cnt = (float)counts[j];
            for (int t = 0; t < d; ++t) code[(size_t)j * d + t] = sum[(size_t)j * d + t] / cnt;
        }
    }
    free(sum);
}

/* kmeans2(data, init_matrix, minit='matrix', iter=iters).  `code` holds the
 * initial code book on entry and the final one on return.  If trace_labels is
 * non-NULL it receives the labels of EVERY iteration ([iters, n]). */
int aoc_oracle_kmeans2(const float *obs, float *code, int n, int k, int d, int iters,
                       int32_t *labels, int32_t *counts, int32_t *trace_labels) {
    if (n < 1 || k < 1 || d < 1 || iters < 1) return -1;
    for (int it = 0; it < iters; ++it) {
        aoc_oracle_vq(obs, code, n, k, d, labels, NULL);
        if (trace_labels) memcpy(trace_labels + (size_t)it * n, labels, sizeof(int32_t) * (size_t)n);
        aoc_oracle_update_means(obs, labels, n, k, d, code, counts);
    }
    return 0;
}

/* Scalar single-thread pairwise-distance + per-object min, used only as the
 * "scalar port" arm of bench.py's cpu_baseline (cores = 1):
 *   out[i,o] = min_j ( (q2[i] + r2[j]) - 2 q_i.r_j + 5e4 * wrong[j,o] )
 * AEM:61-89.  wrong is a per-row bit mask (bit o set <=> label[j,o] < 0.1). */
void aoc_oracle_dense_match_min(const float *q, const float *r, const uint32_t *wrong,
                                int m, int n, int d, int n_obj, float *out) {
    float *r2 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; ++j) {
        float s = 0.0f;
        for (int t = 0; t < d; ++t) s += r[(size_t)j * d + t] * r[(size_t)j * d + t];
        r2[j] = s;
    }
    for (int i = 0; i < m; ++i) {
        const float *x = q + (size_t)i * d;
        float q2 = 0.0f;
        for (int t = 0; t < d; ++t) q2 += x[t] * x[t];
        for (int o = 0; o < n_obj; ++o) out[(size_t)i * n_obj + o] = INFINITY;
        for (int j = 0; j < n; ++j) {
            const float *y = r + (size_t)j * d;
            float acc = 0.0f;
            for (int t = 0; t < d; ++t) acc += x[t] * y[t];
            float dist = (q2 + r2[j]) - 2.0f * acc;
            for (int o = 0; o < n_obj; ++o) {
                float v = dist + (((wrong[j] >> o) & 1u) ? 5e4f : 0.0f);
                if (v < out[(size_t)i * n_obj + o]) out[(size_t)i * n_obj + o] = v;
            }
        }
    }
    free(r2);
}
