/*
 * aoc_hip.h -- C ABI of libaoc_hip.so: the MI355X (gfx950) implementation of the AOC-Net
 * adaptive-proxy matching + mask-calibration hot path.
 *
 * The reference has no FFI layer; its operator surface is the set of Python functions that
 * aocnet.py:6-7 and decoding_module.py:4,7 import.  Each entry point below names the reference
 * code it replaces (paths relative to /root/reference/AOC-Net; AEM =
 * adaptive_embedding_for_matching.py == complete_project/AOCNet/networks/layers/matching.py,
 * ATT = complete_project/AOCNet/networks/layers/attention.py, CL = conditioning_layer.py).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless the name ends in _host;
 *  - all buffers (outputs and workspaces included) are caller-owned; nothing is allocated;
 *  - work is enqueued on `stream` (a hipStream_t passed as void*) and NOT synchronised;
 *  - return value: 0 = enqueued, < 0 = aoc_status error (nothing enqueued); never throws;
 *  - fp32 data, row-major; "rows" are pixels of stride-4 feature maps; C = embedding width.
 */
#ifndef AOC_HIP_H
#define AOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *aoc_stream_t; /* hipStream_t */

enum aoc_status {
    AOC_OK = 0,
    AOC_ERR_INVALID_ARG = -1,   /* null pointer, negative size, unsupported width ... */
    AOC_ERR_WORKSPACE = -2,     /* workspace smaller than aoc_*_workspace_bytes()       */
    AOC_ERR_LAUNCH = -3,        /* hipGetLastError() != hipSuccess after a launch        */
    AOC_ERR_UNSUPPORTED = -4    /* e.g. more than AOC_MAX_OBJECTS objects                */
};

#define AOC_MAX_OBJECTS 30          /* label bits live in a uint32 (bit 31 = "row kept")      */
#define AOC_MAX_CHANNELS 256        /* embedding width handled by the matching kernels         */
#define AOC_MAX_CLUSTERS 64         /* proxies per object per level (cfg4)                     */
#define AOC_PAD_DISTANCE 5.0e4f     /* WRONG_LABEL_PADDING_DISTANCE, AEM:25                    */
#define AOC_ROW_KEPT_BIT 0x80000000u

/* Library / build identification (string is static). */
const char *aoc_version(void);

/* ------------------------------------------------------------------------------------------
 * Label preparation.  Replaces the masked_select / nonzero / index_select plumbing of
 * AEM:197-198 (wrong-label mask, label < 0.1), AEM:252,263-264 (per-object rows, label > 0.9)
 * and AEM:585-591 (keep rows whose label sum > 0.9).  Nothing is copied: the reference pool
 * stays in place and later kernels gather rows through the index lists produced here.
 *
 *  labels      [n, n_obj] float            (the concatenated [h,w,O] label maps of the pool)
 *  right_bits  [n] out: bit o = label[j,o] > 0.9 ; bit 31 = row kept (sum_o label[j,o] > 0.9)
 *  wrong_bits  [n] out: bit o = label[j,o] < 0.1
 *  fg_rows     [n] out: pool row of every kept row, ascending (first counts[n_obj] valid)
 *  obj_rows    [n_obj * n] out: for object o, at obj_offsets[o], the pool rows that are kept
 *              AND right for o, ascending  (the reference's reference_embeddings_flat_cur)
 *  counts      [n_obj + 1] out: rows per object; counts[n_obj] = kept rows
 *  obj_offsets [n_obj + 1] out: exclusive prefix sum of counts[0..n_obj)
 */
size_t aoc_label_prep_workspace_bytes(int64_t n, int n_obj);
int aoc_label_prep(const float *labels, int64_t n, int n_obj,
                   uint32_t *right_bits, uint32_t *wrong_bits,
                   int32_t *fg_rows, int32_t *obj_rows,
                   int32_t *counts, int32_t *obj_offsets,
                   void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* Bit masks only (same definition as above; wrong_bits may be NULL).  Used by local matching
 * (AEM:1023-1028: label > 0.9 of the previous frame). */
int aoc_label_bits(const float *labels, int64_t n, int n_obj, uint32_t *right_bits,
                   uint32_t *wrong_bits, aoc_stream_t stream);

/* Sticky cluster count of AEM:268 (the loop variable is overwritten): k[o] = min(k[o-1], counts[o]),
 * k[-1] = cluster_num.  Device-side so that a pipeline needs no host round trip. */
int aoc_kmeans_plan(const int32_t *counts, int n_seg, int cluster_num, int32_t *seg_k,
                    aoc_stream_t stream);

/* HOST function (no device work, no stream): the initial rows of the k-means calls of n_frames frames x n_levels levels x n_obj objects that see
 * one pool state, exactly as the reference draws them -- scipy kmeans2(minit='points') -> numpy.random.RandomState.choice(n, k, replace=False)
 * = permutation(n)[:k] on the global legacy MT19937 stream, in the order frame, level, object, with the sticky cluster count of AEM:268 and no
 * draw where it is zero (AEM:263-276).  mt_key [624] / *mt_pos: the generator's state (numpy get_state()[1:3]), advanced in place.
 * counts [n_obj] (host), levels [n_levels] (host, each <= kmax); rows_out [n_frames][n_levels * n_obj][kmax] int32 (host), unused entries 0;
 * states_out (may be NULL) [n_frames][625]: key + pos as each frame's first draw finds them (to hand unused frames' draws back). */
int aoc_kmeans_init_rows_draw(uint32_t *mt_key, int32_t *mt_pos, const int32_t *counts, int n_obj, const int32_t *levels, int n_levels,
                              int n_frames, int kmax, int32_t *rows_out, uint32_t *states_out);

/* Segment lists replicated n_rep times: the k-means of several frames that see the SAME pool (the reference re-clusters
 * the whole pool every frame with fresh initial rows, AEM:268-276; the pool only changes every MEM_EVERY frames,
 * eval_manager_mm.py:356-361) can then advance together in one aoc_kmeans_segmented_ex call with n_rep * n_seg segments.
 * rows_out [n_rep * rows_capacity], seg_offsets_out [n_rep * n_seg + 1], seg_k_out [n_rep * n_seg]; replica f's segment s
 * is rows_out[seg_offsets_out[f * n_seg + s] ...).  Device-side (the segment sizes never reach the host). */
int aoc_kmeans_replicate(const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k, int n_seg, int n_rep,
                         int64_t rows_capacity, int32_t *rows_out, int32_t *seg_offsets_out, int32_t *seg_k_out,
                         aoc_stream_t stream);

/* The same with a cluster count per replica: replica f clusters with K = levels_host[f % n_levels], made sticky per replica from
 * the segment sizes exactly like aoc_kmeans_plan (AEM:268).  This is the multi-level configuration (BASELINE.json configs[2],
 * K in {8, 16, 32}; the reference's `cluster_num` argument, AEM:231-232, one call per level): n_rep = frames x levels replicas
 * advance in ONE aoc_kmeans_segmented_ex chain with kmax = max level.  n_levels <= 8.  n_rep == 1 with rows_out == rows only
 * writes seg_offsets_out / seg_k_out. */
int aoc_kmeans_replicate_levels(const int32_t *rows, const int32_t *seg_offsets, int n_seg, int n_rep,
                                const int32_t *levels_host, int n_levels, int64_t rows_capacity, int32_t *rows_out,
                                int32_t *seg_offsets_out, int32_t *seg_k_out, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Segmented Lloyd k-means, bit-identical to scipy.cluster.vq.kmeans2(X, K, minit='matrix',
 * iter=iters) as called at AEM:276 (one "segment" = one object's rows; all segments advance in
 * the same launches).  Arithmetic order is scipy's (see oracle/csrc/aoc_oracle.c): sequential
 * mul-then-add row norms, one k-ordered fma chain per dot product, (M + |x|^2) + |c|^2, strict <
 * (ties -> lowest index), per-cluster sums in row order, division by (float)count, empty cluster
 * keeps its centroid.
 *
 *  pool         [*, C]        embeddings the row ids refer to (pool_rows of them in the _ex form: a segment lists DISTINCT
 *                             pool rows, so no segment is longer than pool_rows -- the per-segment grids rely on it)
 *  rows         packed row ids (aoc_label_prep's obj_rows); segment s = rows[seg_offsets[s] .. seg_offsets[s+1])
 *  seg_offsets  [n_seg + 1]   device
 *  seg_k        [n_seg]       device; 0 = segment skipped (reference: centroid None, AEM:271-273)
 *  init_rows    [n_seg, kmax] device; segment-local indices of the initial centroids
 *                             (scipy minit='points': permutation(n_i)[:K_i])
 *  rows_capacity              host-known upper bound of seg_offsets[n_seg] (sizes the grids)
 *  centroids    [n_seg, kmax, C] out   final code book
 *  labels       [rows_capacity]  out   segment-local cluster id of every packed row (last assignment)
 *  cluster_counts [n_seg, kmax]  out   members per cluster in the last update
 */
size_t aoc_kmeans_workspace_bytes(int64_t rows_capacity, int n_seg, int kmax, int C);
/* Same, with the number of rows of `pool` stated (pool_rows > 0): enables the pipelined ordered
 * accumulation, which addresses rows through a bounds-checked buffer descriptor.  pool_rows = 0
 * (or aoc_kmeans_segmented) selects the generic path.  Results are bit-identical either way. */
int aoc_kmeans_segmented_ex(const float *pool, int64_t pool_rows, int C,
                            const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                            const int32_t *init_rows, int n_seg, int kmax, int iters,
                            int64_t rows_capacity,
                            float *centroids, int32_t *labels, int32_t *cluster_counts,
                            void *workspace, size_t workspace_bytes, aoc_stream_t stream);
/* The same for REPLICATED segment lists (aoc_kmeans_replicate / aoc_kmeans_replicate_levels): n_seg = n_rep * n_base segments, segment
 * f * n_base + s lists the same rows, in the same order, as segment s (initial rows and cluster counts differ).  That is the k-means of
 * several frames that see one pool state (AEM:268-276, eval_manager_mm.py:356-361) and of the cluster levels of BASELINE.json configs[2]
 * in ONE chain; stating n_rep lets the assignment step fetch every pool row once per group of replicas instead of once per replica.
 * Results are bit-identical to n_rep = 1 (= aoc_kmeans_segmented_ex), which is always a valid way to run the same lists. */
int aoc_kmeans_segmented_rep(const float *pool, int64_t pool_rows, int C,
                             const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                             const int32_t *init_rows, int n_seg, int n_rep, int kmax, int iters,
                             int64_t rows_capacity,
                             float *centroids, int32_t *labels, int32_t *cluster_counts,
                             void *workspace, size_t workspace_bytes, aoc_stream_t stream);
int aoc_kmeans_segmented(const float *pool, int C,
                         const int32_t *rows, const int32_t *seg_offsets, const int32_t *seg_k,
                         const int32_t *init_rows, int n_seg, int kmax, int iters,
                         int64_t rows_capacity,
                         float *centroids, int32_t *labels, int32_t *cluster_counts,
                         void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* The reference's second proxy set, AEM:280: for every non-empty cluster j of segment s the mean of
 * the rows  fg[p], p in {segment-local indices with label == j}  of the GLOBAL kept-row array (the
 * reference indexes the wrong array; reproduced as is).  Also emits the squared norms of both
 * proxy sets (AEM:282); an empty / absent proxy gets norm = +inf so that a min skips it.
 *
 *  proxies      [n_seg, 2, kmax, C] out : [:,0] = centroids (copied), [:,1] = centroid_avg
 *  proxy_sqnorm [n_seg, 2, kmax]   out
 */
size_t aoc_build_proxies_workspace_bytes(int64_t rows_capacity, int n_seg, int kmax);
int aoc_build_proxies(const float *pool, int64_t pool_rows, int C, const int32_t *fg_rows,
                      const int32_t *seg_offsets, const int32_t *seg_k, const int32_t *labels,
                      const float *centroids, int n_seg, int kmax, int64_t rows_capacity,
                      float *proxies, float *proxy_sqnorm,
                      void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pixel-to-proxy correlation ("the correlation kernel"): AEM:92-110 + 316-319 (min over an
 * object's adaptive proxies), AEM:112-128 (k = 1 proxies, no min), fused with the proto-mask
 * transform (sigmoid(d + bias) - 0.5) * 2 of AEM:393/602/864.
 *
 *  query        [m, C]
 *  proxies      [n_proxy, C], proxy_sqnorm [n_proxy] (+inf = ignore; NULL = computed in-kernel)
 *  sets         HOST arrays of length n_set: set s = proxies[set_begin[s] .. set_begin[s] + set_size[s]);
 *               value = min over the set; an empty or all-ignored set yields AOC_PAD_DISTANCE
 *               (reference: absent object, AEM:310-313).  Single-proxy sets are the k = 1 proxies
 *               (no min).  The set structure is static (kmax slots per object; unused slots carry
 *               norm = +inf), so the host knows it without a device round trip.
 *  set_bias     [n_set] device (dis_bias of the set's object); may be NULL (= 0)
 *  out element (pixel i, set s) is written at out[i * out_pixel_stride + set_out_offset[s]], so a set
 *               can land in its channel of the [O, 24, h, w] proto-mask buffer or in [1,h,w,O,F].
 *  transform    1 = apply the proto-mask transform, 0 = raw squared distance
 */
int aoc_proxy_corr_min(const float *query, int64_t m, int C,
                       const float *proxies, const float *proxy_sqnorm, int n_proxy,
                       int n_set, const int32_t *set_begin_host, const int32_t *set_size_host,
                       const int64_t *set_out_offset_host, const float *set_bias,
                       float *out, int64_t out_pixel_stride, int transform, aoc_stream_t stream);

/* use_float16=True (AEM:388-392 / matching.py:2640-2646: `.half()` operands): same arguments; operands, |q|^2, |p|^2, the dot products and
 * the distances are rounded to float16 where the reference's float16 tensors round them (sums accumulate in fp32, as torch does);
 * proxy_sqnorm is only consulted for its +inf "absent" marks.  Outputs are fp32 (the bias is added and the sigmoid taken in fp32, as the
 * reference's type promotion does). */
int aoc_proxy_corr_min_f16(const float *query, int64_t m, int C,
                           const float *proxies, const float *proxy_sqnorm, int n_proxy,
                           int n_set, const int32_t *set_begin_host, const int32_t *set_size_host,
                           const int64_t *set_out_offset_host, const float *set_bias,
                           float *out, int64_t out_pixel_stride, int transform, aoc_stream_t stream);

/* Batched form: the same correlation for n_frames frames (of one or of several independent sequences) in ONE persistent launch.
 * One 480p frame is 11.6 MB of traffic: a launch that small is latency-bound by construction ("tiny working sets"), and
 * the reference runs 2 x O x n_chunks launches per frame (AEM:316-319).  All frames share m, C and the set structure (same number
 * of objects and proxy levels); their pointers differ.  Outputs are pixel-contiguous planes: element (pixel i, set s) of frame f is
 * written at frames[f].out[set_out_offset[s] + i].
 *
 *  precision AOC_CORR_SPLIT (C = 100 only; other widths run the fp32 kernel): every value x 2^10 is split in registers into hi + lo
 *            fp16 and q.p = qh.ph + qh.pl + ql.ph (dropped ql.pl term < 2^-22 |q.p|) is accumulated in fp32 by
 *            v_mfma_f32_32x32x16_f16 as ONE K = 304 dot product that also carries -|p|^2/2; fp32-equivalent (tests pin <= 5e-6 on the
 *            outputs).  Needs |x| 2^10 <= 65000 and |x|^2 <= 4000 for every query / proxy value; checked on the device, and when it
 *            fails the exact-fp32 kernel recomputes the launch inside the same call (no host round trip).
 *  precision AOC_CORR_FP32: exact fp32 MFMA (v_mfma_f32_16x16x4_f32), the arithmetic of aoc_proxy_corr_min.
 *  workspace: aoc_proxy_corr_min_batched_workspace_bytes() bytes (the device-side take-over flag: the kernel stores the call's sequence
 *            number there when a precondition fails and the gated fp32 kernel runs iff it finds it).  Zero it once after allocation and
 *            use it from one stream at a time; the calls themselves never reset it. */
typedef struct aoc_corr_frame {
    const float *query;         /* [m, C] */
    const float *proxies;       /* [n_proxy, C] */
    const float *proxy_sqnorm;  /* [n_proxy] (+inf = ignore) or NULL */
    const float *set_bias;      /* [n_set] or NULL */
    float *out;
} aoc_corr_frame;
enum aoc_corr_precision { AOC_CORR_SPLIT = 0, AOC_CORR_FP32 = 1 };
size_t aoc_proxy_corr_min_batched_workspace_bytes(void);
int aoc_proxy_corr_min_batched(const aoc_corr_frame *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                               const int32_t *set_begin_host, const int32_t *set_size_host,
                               const int64_t *set_out_offset_host, int transform, int precision,
                               void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* The same correlation (AEM:92-128 + 316-319, proto-mask transform of AEM:393/602/864) with the query side handed over as the
 * tile-major fp16 split records of aoc_split_rows_tiled -- what the dense kernel consumes for the same frame (aoc_dense_match_min_split
 * with query_rec_tiled = 1), so a frame's query is converted once and both kernels stream it in their MFMA operand layout.  Always the
 * fp16-split arithmetic (C = 100) with the same device-side take-over by the exact-fp32 kernel (which reads `query`); results equal
 * aoc_proxy_corr_min_batched(precision = AOC_CORR_SPLIT) up to the summation order of |q|^2 (query_sqnorm is the sequential sum).
 * Same workspace as aoc_proxy_corr_min_batched. */
typedef struct aoc_corr_frame_rec {
    const float *query;         /* [m, C] fp32 rows (take-over only) */
    const void *query_rec;      /* aoc_split_rows_tiled(query) */
    const float *query_sqnorm;  /* [m], from the same call */
    const float *proxies;       /* [n_proxy, C] */
    const float *proxy_sqnorm;  /* [n_proxy] (+inf = ignore) or NULL */
    const float *set_bias;      /* [n_set] or NULL */
    float *out;
} aoc_corr_frame_rec;
int aoc_proxy_corr_min_records(const aoc_corr_frame_rec *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                               const int32_t *set_begin_host, const int32_t *set_size_host,
                               const int64_t *set_out_offset_host, int transform,
                               void *workspace, size_t workspace_bytes, aoc_stream_t stream);
/* The same call as ONE launch whatever the number of proxy tiles (round 5).  A workgroup's LDS image holds 5 proxy tiles of 32 rows; a frame
 * with more (multi-level proxies: cfg3 has 22 tiles, cfg4 37) is cut into passes, and aoc_proxy_corr_min_records launches them one after
 * another -- ~30-40 us each for ~2 us of MFMA work (launch, image staging, one stream over the query records).  Here the passes are a grid
 * dimension: they run side by side and take the time of one.  Their tile tables live in the workspace
 * (aoc_proxy_corr_min_records_cached_workspace_bytes) and are rewritten only when *tables_key -- a caller-owned HOST word, 0 = nothing
 * cached, one per workspace -- does not match the call's set structure (a sequence writes them once).  tables_key = NULL: the plain call.
 * Results are identical to aoc_proxy_corr_min_records (every workgroup runs the same code on the same pass). */
size_t aoc_proxy_corr_min_records_cached_workspace_bytes(void);
int aoc_proxy_corr_min_records_cached(const aoc_corr_frame_rec *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                                      const int32_t *set_begin_host, const int32_t *set_size_host,
                                      const int64_t *set_out_offset_host, int transform,
                                      void *workspace, size_t workspace_bytes, int64_t *tables_key, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense pixel-level matching: AEM:178-227 + 61-89 without materialising [m, O, n]:
 *   out[i,o] = min_j ( (|q_i|^2 + |r_j|^2) - 2 q_i.r_j + 5e4 * wrong[j,o] ),  j over kept rows,
 * fused with the proto-mask transform (AEM:808).  fp32 MFMA (v_mfma_f32_16x16x4_f32).
 *
 *  pool [*, C]; fg_rows [n_fg_capacity] kept rows; n_fg device scalar (counts[n_obj] of label prep);
 *  wrong_bits [pool rows]; out element (i,o) at out[i*out_pixel_stride + o*out_obj_stride].
 *  n_fg == 0 yields 1.0 everywhere when transform != 0 (AEM:796-797), +inf otherwise.
 */
size_t aoc_dense_match_workspace_bytes(int64_t m, int64_t n_fg_capacity, int n_obj);
int aoc_dense_match_min(const float *query, int64_t m, int C,
                        const float *pool, const int32_t *fg_rows, const int32_t *n_fg,
                        int64_t n_fg_capacity, const uint32_t *wrong_bits,
                        const float *obj_bias, int n_obj,
                        float *out, int64_t out_pixel_stride, int64_t out_obj_stride,
                        int transform, void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* use_float16=True (AEM:801-803 `.half()`; AEM:65-66 the wrong-label mask becomes float16 too): same arguments and workspace as
 * aoc_dense_match_min; operands, norms, dot products, distances, the 5e4 padding (49984 in float16) and its sum are rounded to float16
 * where the reference's float16 tensors round them (sums accumulate in fp32).  Outputs are fp32. */
int aoc_dense_match_min_f16(const float *query, int64_t m, int C,
                            const float *pool, const int32_t *fg_rows, const int32_t *n_fg,
                            int64_t n_fg_capacity, const uint32_t *wrong_bits,
                            const float *obj_bias, int n_obj,
                            float *out, int64_t out_pixel_stride, int64_t out_obj_stride,
                            int transform, void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense pixel-level matching on the fp16 matrix pipe with fp32-equivalent products (same reference
 * lines as aoc_dense_match_min).  Each embedding row is converted ONCE into a "split record":
 * x * 2^10 = hi + lo with hi, lo fp16 (two 11-bit significands: |error| <= 2^-22 |x|, typically 2^-23), plus the three fp16 pieces of -16 |x|^2 in
 * spare k-slots (and, in a fourth one, an upper bound of the norm of the row's lo plane).  q.r is then qh.rh + qh.rl + ql.rh
 * accumulated in fp32 by v_mfma_f32_32x32x16_f16 (the dropped ql.rl term is < 2^-22 |q| |r|): a product carries up to about 4x the
 * rounding error of an fp32 product, and the deviations of the min-distances from the fp32 reference stay within a small multiple
 * of the reference's own fp32 rounding (a few 1e-7 on distances of O(1); tests pin <= 5e-6 on the outputs).
 *
 * The kernel evaluates the qh.rh product for every (32 reference pixels x 32 query pixels) pair and the two cross products only for
 * the pairs that can hold a minimum: |qh.rl + ql.rh| <= |qh||rl| + |ql||rh| (plane norms from the records), so a pair whose one-product
 * value lies further than that margin from the best three-product value already known for its (query pixel, object) is skipped.  The
 * minimum always survives and its value does not depend on what else was evaluated: same result as evaluating all three products
 * everywhere, run after run (tests/test_gpu_dense_split.py), at about 40 % of the matrix instructions.
 *
 * The fast kernels need (a) every |x| * 2^10 <= 65000 and |x|^2 <= 4000 and (b) every kept pool row right
 * for exactly one object (one-hot labels, as in the reference's eval loop).  Both are checked on the
 * device; when either fails the SAME call runs the exact-fp32 kernels of aoc_dense_match_min instead, on
 * the same stream, with no host round trip -- results are then bit-identical to aoc_dense_match_min.
 *
 *  aoc_split_record_bytes(C)   bytes per record (448) or 0 when C is unsupported (C % 4 != 0 or C > 100)
 *  aoc_split_rows              x [n, C] -> records [n * record_bytes], sqnorm [n] (may be NULL);
 *                              *overflow_flag |= 1 when a value does not fit (flag is sticky, caller zeroes it)
 *  aoc_split_rows_tiled        the same 16-byte chunks in tile-major order: per 32-row tile, per plane (hi, lo), per k-step the
 *                              64 chunks [k-half][row % 32] -- one coalesced 1 KiB wave load per MFMA B operand.  The buffer
 *                              holds whole tiles (aoc_split_rows_tiled_bytes(n, C)); rows past n are zero records.
 *  aoc_dense_match_min_split   query/pool [*, C] fp32 (only read by the fp32 take-over), query_rec/pool_rec
 *                              their records (query_rec_tiled != 0: query_rec is tile-major; the pool's are always row-major),
 *                              query_sqnorm [m]; n = pool rows; right_bits, wrong_bits, fg_rows,
 *                              obj_rows, counts, obj_offsets as produced by aoc_label_prep on the n pool rows;
 *                              out / strides / transform as aoc_dense_match_min.  n_obj <= 16.
 */
size_t aoc_split_record_bytes(int C);
int aoc_split_rows(const float *x, int64_t n, int C, void *records, float *sqnorm,
                   int32_t *overflow_flag, aoc_stream_t stream);
size_t aoc_split_rows_tiled_bytes(int64_t n, int C);
int aoc_split_rows_tiled(const float *x, int64_t n, int C, void *records, float *sqnorm,
                         int32_t *overflow_flag, aoc_stream_t stream);
size_t aoc_dense_match_split_workspace_bytes(int64_t m, int64_t n, int n_obj);
int aoc_dense_match_min_split(const float *query, const void *query_rec, const float *query_sqnorm, int query_rec_tiled,
                              int64_t m, int C, const float *pool, const void *pool_rec,
                              const int32_t *overflow_flag, int64_t n,
                              const uint32_t *right_bits, const uint32_t *wrong_bits,
                              const int32_t *fg_rows, const int32_t *obj_rows,
                              const int32_t *counts, const int32_t *obj_offsets,
                              const float *obj_bias, int n_obj,
                              float *out, int64_t out_pixel_stride, int64_t out_obj_stride,
                              int transform, void *workspace, size_t workspace_bytes, aoc_stream_t stream);
/* The same call for a caller that keeps `workspace` across the frames that see ONE pool state (same pool rows, labels and records):
 * reuse_plan = 0 builds the plan (object-pure tile lists, norm maxima, one-hot check: a function of the pool alone) and leaves it in the
 * workspace; reuse_plan = 1 skips the plan kernel and both memsets (the finalize kernel of every call leaves the per-pixel bounds zeroed)
 * and only refreshes the gate from the sticky overflow flag.  Results are identical to aoc_dense_match_min_split. */
int aoc_dense_match_min_split_cached(const float *query, const void *query_rec, const float *query_sqnorm, int query_rec_tiled,
                              int64_t m, int C, const float *pool, const void *pool_rec,
                              const int32_t *overflow_flag, int64_t n,
                              const uint32_t *right_bits, const uint32_t *wrong_bits,
                              const int32_t *fg_rows, const int32_t *obj_rows,
                              const int32_t *counts, const int32_t *obj_offsets,
                              const float *obj_bias, int n_obj,
                              float *out, int64_t out_pixel_stride, int64_t out_obj_stride,
                              int transform, void *workspace, size_t workspace_bytes, int reuse_plan, aoc_stream_t stream);

/* Developer counters of the coarse-then-rescore kernel behind aoc_dense_match_min_split, summed over all launches of the process
 * since the last reset (synchronises the device): out4[0] (reference tile, query tile) pairs tested with one fp16 product,
 * out4[1] pairs rescored with all three products, out4[2] reference tiles with at least one rescoring, out4[3] reference tiles. */
int aoc_dense_prune_stats(uint64_t *out4, int reset);
/* The same counters plus out8[4] = pairs that stopped at the kernel's checkpoint (after 3 of the 7 k-steps an upper bound of the pair's final
 * value -- the partial product plus the Cauchy-Schwarz bound of the rest, both rest norms carried by the records -- was already below what is
 * known for its pixels); out8[5..7] reserved (0). */
int aoc_dense_prune_stats_ex(uint64_t *out8, int reset);

/* How many CUs the streams that launch the matrix kernels (dense matching, batched correlation) may use -- a caller that runs them under a
 * HIP CU mask says so here and the kernels size their grids in whole rounds of that many CUs; 0 (default) = every CU of the device.
 * Process-wide, may be changed between calls.  (The library itself never reads the environment.) */
int aoc_set_stream_cus(int n_cus);

/* Measurement probe: the NEXT aoc_dense_match_min / aoc_dense_match_min_split call made by the calling thread records
 * `start` immediately before and `stop` immediately after its matrix kernel (dense_match_partial_kernel /
 * dense_prune_kernel) on the call's stream, then the probe is cleared.  Both are hipEvent_t created by the caller
 * (bench.py uses it to time that one kernel live, next to the rocprofv3 figure).  NULL, NULL disarms. */
int aoc_dense_match_set_probe(void *start_event, void *stop_event);

/* ------------------------------------------------------------------------------------------
 * Local (windowed) matching: AEM:921-963 + 968-1060 (and the identical local_matching_proxy,
 * AEM:1064-1156) without the F.unfold materialisation.  Works on maps already at matching
 * resolution (the bilinear down / up-sampling of AEM:938-941,1054-1056 are aoc_resize_*).
 *
 *  query, prev   [H, W, C]       right_bits [H*W] (bit o = prev label > 0.9 at that pixel)
 *  window radii  radii_host[n_radii] ascending, radii_host[n_radii-1] = max_distance (atrous 1)
 *  out           [n_obj, n_radii, H, W]; channel order [max, r_0, r_1, ...] (AEM:1034-1046);
 *                value = transform(min over the window of masked distances; 5e4 if none)
 */
int aoc_local_window_match(const float *query, const float *prev, const uint32_t *right_bits,
                           int H, int W, int C, const int32_t *radii_host, int n_radii,
                           const float *obj_bias, int n_obj, float *out, int transform,
                           aoc_stream_t stream);

/* The same with the two remaining arguments of the reference call (AEM:968-971):
 *   atrous_rate  window offsets are the multiples of the rate up to pad = max - max % rate (AEM:949-959 unfold with that stride); the
 *                nested windows are radii / rate rings (AEM:1039);
 *   float16      use_float16=True (AEM:1002-1005): operands, norms, dot products and distances are rounded to float16 exactly where the
 *                reference's float16 tensors round them (the sums accumulate in fp32 as torch does); outputs are fp32. */
int aoc_local_window_match_ex(const float *query, const float *prev, const uint32_t *right_bits,
                              int H, int W, int C, const int32_t *radii_host, int n_radii,
                              const float *obj_bias, int n_obj, float *out, int transform,
                              int atrous_rate, int float16, aoc_stream_t stream);

/* Bilinear (align_corners=True) resize of a channel-last map [h,w,C] -> [H,W,C]  (AEM:938-941).  _ex with float16 != 0: the resize
 * of a float16 tensor (float16 samples, fp32 arithmetic, float16 result), as F.interpolate does in the use_float16 mode. */
int aoc_resize_bilinear_hwc(const float *in, int h, int w, int C, float *out, int H, int W,
                            aoc_stream_t stream);
int aoc_resize_bilinear_hwc_ex(const float *in, int h, int w, int C, float *out, int H, int W,
                               int float16, aoc_stream_t stream);
/* Bilinear (align_corners=True) resize of planes [P,h,w]; plane p = (po, pi) = (p / inner_count,
 * p % inner_count); element (p,y,x) is written at
 *   out[po*out_outer_stride + pi*out_plane_stride + (y*W + x)*out_pixel_stride]
 * so one call can drop [O, F, h, w] results into their channel slice of the [O, 24, H, W] proto-mask
 * buffer or into the reference's [1, H, W, O, F] layout (AEM:604-607, 1054-1058).  An identity-size
 * resize is an exact copy. */
int aoc_resize_bilinear_planes(const float *in, int P, int h, int w, float *out, int H, int W,
                               int inner_count, int64_t out_outer_stride, int64_t out_plane_stride,
                               int64_t out_pixel_stride, aoc_stream_t stream);
/* Atrous sub-sampling of a channel-last map (AEM:533-579, the atrous_obj_pixel_num == 0 branch: pad to a multiple of the rate, view as
 * [h', rate, w', rate, X], keep [:, 0, :, 0]): out [h', w', X] = in[rate * y', rate * x', :] with h' = ceil(h / rate), w' = ceil(w / rate) -- the
 * padding is never selected.  For the embeddings AND the label maps of a pool frame (a device-side gather: no host-side pad / view arithmetic). */
int aoc_atrous_subsample(const float *in, int h, int w, int X, int rate, float *out, aoc_stream_t stream);
/* torch 'nearest' resize of per-pixel label bits [h,w] -> [H,W]  (AEM:1017-1018). */
int aoc_resize_nearest_bits(const uint32_t *in, int h, int w, uint32_t *out, int H, int W,
                            aoc_stream_t stream);

/* ---- round 4: fused per-frame launches (each replaces several of the calls above and computes the same expressions term by term, so
 * the outputs equal theirs bit for bit; hotpath.proto_mask_features and aoc_frame_enqueue both use them) ---------------------------- */

/* aoc_resize_bilinear_planes with one more level of output addressing: plane p = (group, outer, inner), groups of
 * outer_count * inner_count planes land out_group_stride apart (the two local-matching results of a frame -> two channel ranges of the
 * proto-mask tensor, aocnet.py:355, in one launch). */
int aoc_resize_bilinear_planes_grouped(const float *in, int P, int h, int w, float *out, int H, int W, int inner_count, int outer_count,
                                       int64_t out_group_stride, int64_t out_outer_stride, int64_t out_plane_stride, int64_t out_pixel_stride,
                                       aoc_stream_t stream);

/* local_matching AND local_matching_proxy (AEM:968-1156 as aocnet.py:255,328 calls them: same query, same label bits, fp32, atrous rate 1)
 * in one launch: out_a from prev_a (the previous frame's embedding), out_b from prev_b (its per-pixel proxy map). */
int aoc_local_window_match_pair(const float *query, const float *prev_a, const float *prev_b, const uint32_t *right_bits, int H, int W, int C,
                                const int32_t *radii_host, int n_radii, const float *obj_bias, int n_obj, float *out_a, float *out_b,
                                int transform, aoc_stream_t stream);

/* The half-resolution operands of both local matchings (AEM:938-941 with MODEL_LOCAL_DOWNSAMPLE) in one launch:
 *   q2, p2 [H2, W2, C]  = aoc_resize_bilinear_hwc of cur_emb / prev_emb [h, w, C]
 *   pm2    [H2, W2, C]  = aoc_resize_bilinear_hwc of aoc_label_mix(prev_labels [h*w, n_obj], prev_pos [n_obj, C])   (aocnet.py:325)
 *   bits2  [H2 * W2]    = aoc_resize_nearest_bits of aoc_label_bits(prev_labels)
 * Optional small tables filled by the same launch (NULL / 0 = off): set_bias_out [n_pair_sets + n_obj] = the per-set bias table of the
 * correlation launch (set s < n_pair_sets belongs to object (s / 2) % n_obj, set n_pair_sets + o to object o); two float copies
 * copy_dst_x[0 .. n_copy_x) = copy_src_x[..] (the pooled reference heads into the k = 1 rows of the proxy table). */
int aoc_local_prep(const float *cur_emb, const float *prev_emb, const float *prev_labels, const float *prev_pos, int h, int w, int C, int n_obj,
                   float *q2, float *p2, float *pm2, uint32_t *bits2, int H2, int W2,
                   const float *obj_bias, int n_pair_sets, float *set_bias_out,
                   const float *copy_src_a, float *copy_dst_a, int n_copy_a, const float *copy_src_b, float *copy_dst_b, int n_copy_b,
                   aoc_stream_t stream);

/* The tail of a frame's proto-mask tensor feat [n_obj, n_ch, hw] (object stride obj_stride floats) in one launch, aocnet.py:349-358:
 * channels [ch_local_bg, +n_local) = foreground2background of [ch_local, +n_local), channel ch_global_bg = foreground2background of
 * ch_global, channel ch_prev_mask = prev_labels [hw, n_obj] transposed; head [n_obj, 4 C] = (ref_pos | ref_neg | prev_pos | prev_neg),
 * ATT:188.  A channel index < 0 / head == NULL switches that part off. */
int aoc_proto_finish(float *feat, int n_obj, int64_t hw, int64_t obj_stride, int ch_local, int n_local, int ch_local_bg, int ch_global, int ch_global_bg,
                     int ch_prev_mask, const float *prev_labels, const float *ref_pos, const float *ref_neg, const float *prev_pos, const float *prev_neg,
                     int C, float *head, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * foreground2background, AEM:9-23: the reference concatenates the OTHER objects' maps along dim 1
 * and reduces that dim, so  out[o, 0, x] = min over o' != o and over channels c of dis[o', c, x].
 *  dis: object o at dis + o*dis_obj_stride holds [n_ch, inner]; out: object o at out + o*out_obj_stride
 *  holds [inner]; n_obj >= 2 (n_obj == 1 returns the input, AEM:10-11). */
int aoc_fg2bg_min(const float *dis, int n_obj, int n_ch, int64_t inner, int64_t dis_obj_stride,
                  float *out, int64_t out_obj_stride, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * k = 1 proxies / IA head: ATT:134-189 (calculate_attention_head[_for_eval]_p_m).
 *  Accumulates over n_frames maps  emb [n_frames, hw, C] (channel-last) with float label maps
 *  labels [n_frames, n_obj, hw] (the reference's one-hot [O,1,h,w] tensors; any float works), or
 *  [n_frames, hw, n_obj] when labels_pixel_major != 0 (the layout the matching functions take):
 *    pos[o] = sum_p emb[p] * label[o,p] / (sum_p label[o,p] + eps)
 *    neg[o] = (sum_p emb[p] - pos_sum[o]) / (sum_p (1 - label[o,p]) + eps)
 *  out_pos, out_neg [n_obj, C]; out_pos_sqnorm [n_obj] = |pos[o]|^2 (optional, may be NULL): the k = 1
 *  proxies go straight into aoc_proxy_corr_min's proxy table.
 */
size_t aoc_masked_mean_pool_workspace_bytes(int n_frames, int64_t hw, int n_obj, int C);
int aoc_masked_mean_pool(const float *emb, const float *labels, int n_frames, int64_t hw, int C,
                         int n_obj, int labels_pixel_major, float epsilon, float *out_pos, float *out_neg,
                         float *out_pos_sqnorm, void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FiLM gate: ATT:12-17 (IA_gate.forward) and CLB:81-84:
 *   gain[o,c] = 1 + tanh( sum_d head[o,d] * weight[c,d] + bias[c] );  y[o,c,:] = gain[o,c] * x[o,c,:]
 *  aoc_film_gain writes gain [n_obj, channels]; aoc_channel_scale streams x (in place allowed). */
int aoc_film_gain(const float *head, const float *weight, const float *bias, int n_obj,
                  int head_dim, int channels, float *gain, aoc_stream_t stream);
int aoc_channel_scale(const float *x, const float *gain, int64_t planes, int64_t hw, float *y,
                      aoc_stream_t stream);
/* Both steps in one launch (what IA_gate.forward does, ATT:12-17): x [n_obj, channels, hw]. */
int aoc_film_scale(const float *x, const float *head, const float *weight, const float *bias,
                   int n_obj, int head_dim, int channels, int64_t hw, float *y, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * conditioning_layer gate + pool: CL:23-43 (paper Eq. 7):
 *   s = phi(z) (1x1 conv C->1), t = k-th largest of s per sample (k = int(beta*H*W), exact radix
 *   select), mask = s > t (strict), gap[n,c] = mean over ALL HW of z[n,c,:] * mask.
 *  z [N, C, HW]; phi_w [C]; phi_b [1]; gap [N, C] out; scores [N, HW] / threshold [N] optional outs
 *  (may be NULL, then they live in the workspace).
 */
size_t aoc_cond_gate_pool_workspace_bytes(int N, int C, int64_t hw);
int aoc_cond_gate_pool(const float *z, int N, int C, int64_t hw, const float *phi_w,
                       const float *phi_b, int k_rank, float *gap, float *scores, float *threshold,
                       void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* The same, also emitting plane_mean [N, C] = mean over HW of z[n,c,:] (CLB:68 avg_pool2d, the input of conditioning_block's inter-object
 * code) from the SAME pass that computes the scores: a conditioning block then reads its activation three times (scores + plane sums,
 * masked pooling, FiLM scale) instead of four.  plane_mean may be NULL. */
int aoc_cond_gate_pool_ex(const float *z, int N, int C, int64_t hw, const float *phi_w,
                          const float *phi_b, int k_rank, float *gap, float *plane_mean, float *scores,
                          float *threshold, void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* Small dense layer used by the conditioning MLPs (CL:46, CLB:81): y[n,:] = x[n,:] W^T + b. */
int aoc_linear(const float *x, const float *weight, const float *bias, int N, int in_dim,
               int out_dim, float *y, aoc_stream_t stream);
/* out[p,:] = sum_o labels[p,o] * rows[o,:]: the per-pixel proxy map fed to local_matching_proxy
 * (aocnet.py:325).  labels [n, n_obj], rows [n_obj, C], out [n, C]. */
int aoc_label_mix(const float *labels, const float *rows, int64_t n, int n_obj, int C, float *out,
                  aoc_stream_t stream);
/* Global average pool of planes [planes, hw] -> [planes]  (CLB:68). */
int aoc_plane_mean(const float *x, int64_t planes, int64_t hw, float *out, aoc_stream_t stream);
/* out [n_obj, head_dim + channels] = head [n_obj, head_dim] extended with the inter-object code of the plane means [n_obj, channels]:
 * out[o, head_dim + c] = sum_o' px[o', c] - px[o, c]  (decoding_module.py:126-130, torch.cat([head, px.sum(0, keepdim=True) - px], 1)) */
int aoc_head_delta(const float *head, int head_dim, const float *plane_means, int n_obj, int channels, float *out, aoc_stream_t stream);

/* DynamicPreHead, decoding_module.py:228-240 (1x1 convolution n_in -> n_out, GroupNorm(n_groups), ReLU) applied to the
 * [n_obj, n_in, hw] proto-mask tensor, fused with the concatenation of aocnet.py:362: out [n_obj, C + n_out, hw] holds the
 * current-frame embedding emb_hwc [hw, C] (expanded over the objects, transposed to channel-first) in channels [0, C) and the
 * pre-head output in channels [C, C + n_out) -- the tensor CalibrationDecoding consumes.  emb_hwc may be NULL with C = 0.
 * weight [n_out, n_in], bias / gamma / beta [n_out]; GroupNorm statistics (biased variance) are reduced in a fixed order. */
size_t aoc_prehead_workspace_bytes(int n_obj, int n_out, int group_size, int64_t hw);
int aoc_prehead(const float *feat, int n_obj, int n_in, int64_t hw, const float *weight, const float *bias, int n_out,
                int n_groups, const float *gamma, const float *beta, float eps, const float *emb_hwc, int C, float *out,
                void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* conditioning_block codes in one launch, CLB:68-80: code[n] = [ W1 gap[n] + b1 | W2 (sum_m px[m] - px[n]) + b2 | W3 head[n] + b3 ]
 * with gap [N, C] (aoc_cond_gate_pool), px = plane means [N, C] (aoc_plane_mean), head [N, D]; W1, W2 [C, C], W3 [D, D] are
 * the mlp_layer weights of CL_1..CL_3 (row-major [out, in]); code [N, 2C + D] feeds aoc_film_scale. */
int aoc_cond_codes(const float *gap, const float *plane_means, const float *head, const float *w1, const float *b1,
                   const float *w2, const float *b2, const float *w3, const float *b3, int N, int C, int D,
                   float *code, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Decoder-side streams next to the FiLM gates.
 *
 * aoc_plane_reduce   out[plane] = sum over the plane of x (mode 0), x^2 (mode 1), |x| (mode 2)      (gct.py:19,27-30)
 * aoc_gct_gate       GCT gate of gct.py:17-36 from those sums [N, C]: l2 mode (l1_mode = 0, sums of squares) or l1 mode
 *                    (sums of x or |x|); gate [N, C] = 1 + tanh(embedding * norm + beta); y = x * gate is aoc_channel_scale
 * aoc_object_logit   IA_logit, decoding_module.py:151-160: out[n, p] = sum_c x[n, c, p] * weight[n * weight_stride + c] +
 *                    bias[n * bias_stride] (the per-object 1x1 grouped convolution with weights generated from the IA head;
 *                    weight / bias are usually views of one [N, C + 1] linear output: weight_stride = bias_stride = C + 1) */
int aoc_plane_reduce(const float *x, int64_t planes, int64_t hw, int mode, float *out, aoc_stream_t stream);
int aoc_gct_gate(const float *plane_sums, const float *alpha, const float *gamma, const float *beta, int N, int C,
                 float eps, int l1_mode, float *gate, aoc_stream_t stream);
int aoc_object_logit(const float *x, int N, int C, int64_t hw, const float *weight, int64_t weight_stride,
                     const float *bias, int64_t bias_stride, float *out, aoc_stream_t stream);

/* GroupNorm (+ residual) + ReLU of the decoder's Bottleneck (networks/layers/gct.py:69-90: bn1/bn2 followed by relu, bn3 followed by
 * `out += residual; relu`): y = [relu]( GroupNorm_groups(x) * gamma + beta [+ residual] ), x [N, C, hw], biased variance over each
 * group's channels x hw as torch.nn.GroupNorm.  Two streams over x (statistics, apply) and one write instead of the separate
 * normalisation / add / ReLU passes.  gamma, beta [C] or NULL; residual [N, C, hw] or NULL; y may alias x. */
size_t aoc_groupnorm_relu_workspace_bytes(int N, int groups);
int aoc_groupnorm_relu(const float *x, int N, int C, int64_t hw, int groups, const float *gamma, const float *beta,
                       float eps, const float *residual, int relu, float *y, void *workspace,
                       size_t workspace_bytes, aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Eval-loop memory policy (the caller of the matching path).
 *
 * aoc_confident_labels: the per-pixel decision of one frame, eval_manager_mm.py:253-265,300-326,339-346,357-361 with
 *   shannon_entropy.py:10-13.   probs [n_ch, n] = the (augmentation-averaged) class probabilities;
 *   exist_bits: bit c = label c has appeared in a ground-truth map so far (other channels are zeroed);
 *   join_label [n] or NULL: ground truth that introduces new objects (0 = keep the prediction, < 0 = unsure);
 *   labels_out [n] = argmax (first maximum) with the join override; confident_out [n] (may be NULL) = the same label or
 *   125 where the entropy -sum p log(p + 1e-6) over the seen channels exceeds unc_ratio; entropy_out [n] (may be NULL).
 * aoc_label_onehot_nearest: aocnet.py:128-133,151: nearest-neighbour resize of an int label map [H, W] to [h, w] and
 *   one-hot over n_obj ids -> float [h, w, n_obj] (label 125 or any id >= n_obj matches nothing: an all-zero row). */
int aoc_confident_labels(const float *probs, int n_ch, int64_t n, uint32_t exist_bits, const int32_t *join_label,
                         float unc_ratio, int32_t *labels_out, int32_t *confident_out, float *entropy_out,
                         aoc_stream_t stream);
int aoc_label_onehot_nearest(const int32_t *label, int H, int W, int h, int w, int n_obj, float *onehot_hwc,
                             aoc_stream_t stream);

/* J (region similarity) and F (boundary measure) of a predicted label map against the ground truth, summed over the foreground
 * objects 1 .. n_obj-1 and ACCUMULATED on the device: accum[0] += sum_o J_o, accum[1] += sum_o F_o,
 * accum[2] += n_obj - 1, accum[3] += 1.  The reference scores saved PNGs with the external DAVIS toolkit (README.md:110; its only
 * in-repo IoU is utils/metric.py:3-34): the definitions here are DAVIS-2017's db_eval_iou and db_eval_boundary (seg2bmap boundaries,
 * dilation by skimage's disk(bound_pix), bound_pix = ceil(0.008 * hypot(H, W)) chosen by the caller).  pred, gt [H, W] int32 labels;
 * workspace_is_clean != 0 skips the zeroing of the counters (the call leaves them zeroed).  n_obj <= 16. */
size_t aoc_mask_jf_workspace_bytes(int H, int W);
int aoc_mask_jf_accumulate(const int32_t *pred, const int32_t *gt, int H, int W, int n_obj, int bound_pix,
                           void *workspace, size_t workspace_bytes, int workspace_is_clean, double *accum,
                           aoc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One frame as ONE call: the counterpart of AOCNet.before_seghead_process (networks/aoc/aocnet.py:114-372, eval branch, batch 1) up to the
 * tensor handed to DynamicPreHead (aocnet.py:355-358) and the attention head (aocnet.py:297), enqueued on `stream` without any host
 * synchronisation.  It issues the kernels of the individual entry points above (results bit-identical to calling them one by one, which
 * is what hotpath.proto_mask_features does) out of ONE caller-owned device workspace per sequence, and keeps across the frames of a sequence
 * what only depends on the reference pool: its fp16 split records (append-only pool: only new frames are converted), the pooled reference
 * heads (ATT:155-170) and the dense kernel's plan -- keyed by pool_key.
 *
 * Covered configuration: C = 100, <= 30 objects (AOC_MAX_OBJECTS; beyond 16 the dense matching takes the exact-fp32 kernel in slices of 16 and
 * the correlation the non-record entry point, as hotpath.proto_mask_features does), and every switch the reference's evaluation CLI / config
 * exposes: MODEL_FLOAT16_MATCHING, MODEL_LOCAL_DOWNSAMPLE on / off, TEST_LOCAL_ATROUS_RATE, TEST_GLOBAL_ATROUS_RATE (round 5; the default
 * configuration -- fp32, down-sampled local matching, atrous rates 1, <= 16 objects -- keeps the fused launches of round 4, the others go
 * through the individual entry points the drop-in mirrors use, in the same order: results equal theirs).  Other widths return AOC_ERR_UNSUPPORTED.
 *
 * The adaptive proxies (k-means chain, AEM:252-286: aoc_label_prep, aoc_kmeans_replicate_levels, aoc_kmeans_segmented_rep,
 * aoc_build_proxies) only depend on the pool and are produced by the caller on ANOTHER stream, ahead of the frame; this call gets the label
 * prep arrays, the proxy table they are written to, and two hipEvent_t: it waits for prep_ready before reading the label prep and for
 * proxies_ready in front of the correlation launch (NULL = already ordered on `stream`).
 *
 * Output channels of feat [n_obj, n_ch, h, w], n_ch = aoc_frame_channels(): global(1) | cluster (centroid, centroid_avg) per level |
 * proxy(1) | local(n_radii) | local_proxy(n_radii) | prev_mask(1) [| local_bg(n_radii) | global_bg(1)]. */
typedef struct aoc_frame_desc {
    int32_t h, w, C, n_obj;          /* map size, embedding width, objects incl. background */
    int32_t R, R_capacity;           /* pool frames this frame sees / the workspace was sized for */
    int32_t n_radii, radii[8];       /* MODEL_MULTI_LOCAL_DISTANCE */
    int32_t n_levels, levels[8];     /* cluster_num per level (AEM:232; one level: {16}) */
    int32_t kmax;                    /* max(levels): proxy slots per (level, object, set) in the table */
    int32_t matching_background;     /* MODEL_MATCHING_BACKGROUND */
    int32_t n_adaptive;              /* = n_levels * n_obj * 2 * kmax rows of the proxy table in front of the n_obj k = 1 rows */
    float epsilon;                   /* MODEL_EPSILON */
    int32_t pool_prefix_frames;      /* how many leading pool frames are UNCHANGED since the previous call on this state: their split records are
                                        kept, the frames behind them are converted again.  Append-only pool: R (or anything >= the previous R);
                                        a pool whose frames were replaced: the first replaced frame's index; 0 converts everything */
    int32_t stream_cus;              /* CUs `stream` may use when the caller created it with a HIP CU mask: the matrix kernels of THIS call size their
                                        grids in whole rounds of that many CUs; 0 = the process-wide aoc_set_stream_cus value */
    /* the switches the reference's evaluation CLI / config exposes (tools/eval_net_mm_rpa.py:9-35, configs/resnet101_aocnet.py:71,78,125,126) */
    int32_t float16_matching;        /* MODEL_FLOAT16_MATCHING (--float16): the reference's `.half()` arithmetic for the dense, k = 1 proxy and local
                                        matchings; the cluster channels are the constant 1 the reference degrades to */
    int32_t local_downsample;        /* MODEL_LOCAL_DOWNSAMPLE: local matching at (h / 2 + 1, w / 2 + 1) (1) or at full resolution (0) */
    int32_t local_atrous_rate;       /* TEST_LOCAL_ATROUS_RATE (>= 1) */
    int32_t match_hw;                /* TEST_GLOBAL_ATROUS_RATE > 1 (--global_atrous_rate): rows per pool frame of match_emb (the pool sub-sampled on the
                                        atrous grid, AEM:533-579: every rate-th row and column); 0 = the matching pool is ref_emb itself */
    int64_t pool_key;                /* != 0; changes whenever the pool's content changes (append-only pool: R).  Keys the pooled reference
                                        heads and the dense kernel's plan */
    const float *ref_emb;            /* [R * h * w, C]   reference pool, resident, append-only */
    const float *ref_labels;         /* [R * h * w, n_obj] float 0 / 1 */
    const float *match_emb;          /* [R * match_hw, C] the pool the dense and cluster matchings see when match_hw > 0 (aoc_atrous_subsample of every
                                        pool frame; the label-prep arrays below then index ITS rows); NULL with match_hw = 0.  The pooled heads
                                        (attention head, k = 1 proxies) always come from ref_emb / ref_labels (aocnet.py:297) */
    const float *prev_emb, *prev_labels, *cur_emb;     /* [h * w, C], [h * w, n_obj], [h * w, C] */
    const float *dis_bias;           /* [n_obj] */
    const uint32_t *right_bits, *wrong_bits;           /* aoc_label_prep(ref_labels) */
    const int32_t *fg_rows, *obj_rows, *counts, *obj_offsets;
    float *proxy_table;              /* [n_adaptive + n_obj, C]: adaptive rows by the k-means chain, k = 1 rows by this call */
    float *proxy_sqnorm;             /* [n_adaptive + n_obj] */
    void *prep_ready, *proxies_ready;                  /* hipEvent_t or NULL */
    float *feat;                     /* [n_obj, n_ch, h, w] */
    float *head;                     /* [n_obj, 4 C] */
    void *probe[6];                  /* measurement only, hipEvent_t or NULL: recorded on `stream` immediately before / after the dense matching
                                        op [0][1], the correlation launch [2][3], the local-matching launch [4][5] */
} aoc_frame_desc;
/* Host-side record of what the workspace holds; caller-owned, zero-initialised when a sequence starts (the library keeps no state). */
typedef struct aoc_seq_state {
    int64_t initialised, records_frames, ref_pool_key, plan_key, plan_rows;
    int64_t corr_tables_key;         /* the correlation passes' tile tables the workspace holds (aoc_proxy_corr_min_records_cached) */
} aoc_seq_state;
int aoc_frame_channels(int n_radii, int n_levels, int matching_background);
size_t aoc_frame_workspace_bytes(int h, int w, int C, int n_obj, int R_capacity, int n_radii, int n_levels);
int aoc_frame_enqueue(const aoc_frame_desc *desc, aoc_seq_state *state, void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* The adaptive proxies of the frames that see ONE pool state as ONE call (round 5): what hotpath.launch_cluster_proxies_batch issued as
 * three C calls plus a dozen tensor allocations and 2 F table copies -- aoc_kmeans_replicate_levels, aoc_kmeans_segmented_rep (20 Lloyd
 * iterations, AEM:252-279), aoc_build_proxies (AEM:280-282) and the scatter of every frame's proxies into ITS proxy table -- out of one
 * caller-owned workspace, enqueued on `stream` (the caller's side stream) without host synchronisation.  Bit-identical to the three calls.
 * With aoc_frame_enqueue and aoc_gates_enqueue a frame of the orchestrated path is three C calls (one per stream it touches). */
typedef struct aoc_chain_desc {
    int32_t C, n_obj, n_frames;      /* embedding width, objects incl. background, frames F that see this pool state (<= 8) */
    int32_t n_levels, levels[8];     /* cluster_num per level (AEM:232) */
    int32_t kmax, iters;             /* max(levels); Lloyd iterations (20: scipy's kmeans2 default the reference uses) */
    int32_t reserved0;
    int64_t pool_rows;               /* rows of the pool the matchings see (R * h * w, or R * match_hw on the atrous grid) */
    int64_t rows_capacity;           /* entries of obj_rows (aoc_label_prep: pool_rows * n_obj) */
    const float *pool;               /* [pool_rows, C] */
    const int32_t *fg_rows, *obj_rows, *obj_offsets;    /* aoc_label_prep of that pool's labels */
    const int32_t *init_rows;        /* [n_frames * n_levels * n_obj, kmax] segment-local initial rows, (frame, level, object)-major */
    float *tables[8];                /* per frame: its proxy table [n_levels * n_obj * 2 * kmax + n_obj, C]; the adaptive rows are written */
    float *sqnorms[8];               /* per frame: [n_levels * n_obj * 2 * kmax + n_obj] */
} aoc_chain_desc;
size_t aoc_cluster_chain_workspace_bytes(const aoc_chain_desc *desc);
/* byte offsets (into the workspace) of what the chain leaves there, for callers that want to look at it: [0] centroids [S, kmax, C] float,
 * [1] labels [n_frames * n_levels * rows_capacity] int32, [2] cluster counts [S, kmax] int32, [3] proxies [S, 2, kmax, C] float,
 * [4] proxy squared norms [S, 2, kmax] float, [5] seg_k [S] int32, [6] seg_offsets [S + 1] int32;  S = n_frames * n_levels * n_obj */
int aoc_cluster_chain_layout(const aoc_chain_desc *desc, int64_t *offsets7);
int aoc_cluster_chain_enqueue(const aoc_chain_desc *desc, void *workspace, size_t workspace_bytes, aoc_stream_t stream);

/* The modulation gates of CalibrationDecoding.forward (decoding_module.py:96-149, 162-210) as ONE call: a list of gate descriptors, every
 * gate issuing exactly the launches of its module mirror (outputs bit-identical to attention.IA_gate / conditioning_layer.conditioning_block).
 *   kind 0  IA_gate (ATT:7-17):                       y = x * (1 + tanh(head W^T + b)),                     W [channels, head_dim]
 *   kind 1  IA gate with the extended head (decoding_module.py:126-130): head' = (head | sum_o GAP(x) - GAP(x)), W [channels, head_dim + channels]
 *   kind 2  conditioning_block (CLB:50-86):  W = mlp_layer.weight [channels, 2 channels + head_dim]; phi = CL_1.phi_layer; w1..b3 = the
 *           mlp_layer of CL_1 / CL_2 / CL_3; k_rank = int(beta * H * W) (CL:32)
 * x, y [n_obj, channels, hw]; head [n_obj, head_dim]; the scratch of all gates comes out of one workspace (stream-ordered reuse). */
typedef struct aoc_gate_desc {
    int32_t kind, channels, k_rank, reserved;
    int64_t hw;
    const float *x;
    float *y;
    const float *w, *b;
    const float *phi_w, *phi_b, *w1, *b1, *w2, *b2, *w3, *b3;
    void *probe[4];                  /* measurement only, hipEvent_t or NULL: recorded immediately before / after the gate's aoc_film_scale [0][1]
                                        and its aoc_cond_gate_pool_ex [2][3] */
} aoc_gate_desc;
size_t aoc_gates_workspace_bytes(const aoc_gate_desc *gates, int n_gates, int n_obj, int head_dim);
int aoc_gates_enqueue(const aoc_gate_desc *gates, int n_gates, const float *head, int n_obj, int head_dim, void *workspace, size_t workspace_bytes,
                      aoc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AOC_HIP_H */
