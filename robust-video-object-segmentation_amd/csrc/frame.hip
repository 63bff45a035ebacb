// One frame of the matching path as ONE call: the counterpart of AOCNet.before_seghead_process (aocnet.py:114-372) up to the proto-mask
// tensor and the attention head, on the caller's stream, out of one caller-owned workspace per sequence.
//
// The reference's own entry point for this work is a single Python method call per frame; until round 3 this library was driven through
// ~60 ctypes calls per frame (argument marshalling, ~40 torch allocations), 2.4-3.4 ms of host time per two-frame step.  aoc_frame_enqueue
// issues the same kernels the Python orchestrator (hotpath.proto_mask_features) issues, through the same C entry points, so the results
// are bit-identical by construction (tests/test_gpu_frame.py: torch.equal); what it adds is state kept across the frames of a sequence:
//   * the fp16 split records of the (append-only) reference pool: only frames that joined since the last call are converted;
//   * the pooled reference heads (ATT:155-170), a function of the pool alone;
//   * the dense kernel's plan (object-pure tile lists, norm maxima, one-hot check), also a function of the pool alone: 4 of 5 frames skip
//     the plan kernel and both memsets (aoc_dense_match_min_split_cached);
// keyed by desc->pool_key, which the caller changes whenever the pool's content changes.
//
// The adaptive proxies of the frame (k-means chain, AEM:252-286) are produced on ANOTHER stream (they only depend on the pool): the caller
// hands over the proxy table they are written to and two events; this call waits for `prep_ready` before it reads the label prep and for
// `proxies_ready` in front of the correlation launch.
#include <algorithm>

#include "aoc_common.h"

namespace {

struct FrameWs {
    // persistent across the frames of a sequence
    int32_t *overflow;            // sticky fp16-overflow flag of the split records (zeroed once)
    char *corr_ws;                // 256-byte flag workspace of the correlation launch (zeroed once)
    char *pool_rec;               // [R_capacity * hw, 448]
    float *pool_sq;               // [R_capacity * hw]
    float *ref_pos, *ref_neg, *ref_sq;
    char *dense_ws;
    size_t dense_bytes;
    // per frame (stream-ordered scratch)
    float *prev_pos, *prev_neg, *set_bias;
    char *q_rec;
    float *q_sq;
    float *q2, *p2, *pm2, *lf;
    uint32_t *bits2;
    char *pool_ws;
    size_t pool_ws_bytes;
    // the non-default modes (float16 matching, full-resolution local matching): full-resolution label bits and per-pixel proxy map
    uint32_t *bits_full;
    float *pmap;
    size_t total, init_bytes;
};

inline FrameWs frame_carve(void *base, int h, int w, int C, int n_obj, int R_cap, int n_radii, int n_set) {
    FrameWs f;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += aoc_align_up(bytes, 256); return r; };
    const int64_t hw = (int64_t)h * w, n_cap = hw * R_cap;
    const int H2 = h / 2 + 1, W2 = w / 2 + 1;
    f.overflow = reinterpret_cast<int32_t *>(take(256));
    f.corr_ws = take(aoc_proxy_corr_min_records_cached_workspace_bytes());
    f.init_bytes = off;                                     // [0, init_bytes) is zeroed when a sequence starts
    f.pool_rec = take((size_t)n_cap * aoc_split_record_bytes(C));
    f.pool_sq = reinterpret_cast<float *>(take((size_t)n_cap * sizeof(float)));
    f.ref_pos = reinterpret_cast<float *>(take((size_t)n_obj * C * sizeof(float)));
    f.ref_neg = reinterpret_cast<float *>(take((size_t)n_obj * C * sizeof(float)));
    f.ref_sq = reinterpret_cast<float *>(take((size_t)n_obj * sizeof(float)));
    // the split kernel's workspace (which contains the exact-fp32 take-over's) up to 16 objects, the exact-fp32 kernel's beyond
    f.dense_bytes = std::max(n_obj <= 16 ? aoc_dense_match_split_workspace_bytes(hw, n_cap, n_obj) : (size_t)0, aoc_dense_match_workspace_bytes(hw, n_cap, n_obj));
    f.dense_ws = take(f.dense_bytes);
    f.prev_pos = reinterpret_cast<float *>(take((size_t)n_obj * C * sizeof(float)));
    f.prev_neg = reinterpret_cast<float *>(take((size_t)n_obj * C * sizeof(float)));
    f.set_bias = reinterpret_cast<float *>(take((size_t)(n_set > 0 ? n_set : 1) * sizeof(float)));
    f.q_rec = take(aoc_split_rows_tiled_bytes(hw, C));
    f.q_sq = reinterpret_cast<float *>(take((size_t)hw * sizeof(float)));
    const size_t half = (size_t)H2 * W2 * C * sizeof(float);
    f.q2 = reinterpret_cast<float *>(take(half));
    f.p2 = reinterpret_cast<float *>(take(half));
    f.pm2 = reinterpret_cast<float *>(take(half));
    f.lf = reinterpret_cast<float *>(take((size_t)2 * n_obj * n_radii * hw * sizeof(float)));       // full resolution: MODEL_LOCAL_DOWNSAMPLE off
    f.bits2 = reinterpret_cast<uint32_t *>(take((size_t)H2 * W2 * sizeof(uint32_t)));
    f.bits_full = reinterpret_cast<uint32_t *>(take((size_t)hw * sizeof(uint32_t)));
    f.pmap = reinterpret_cast<float *>(take((size_t)hw * C * sizeof(float)));
    f.pool_ws_bytes = std::max(aoc_masked_mean_pool_workspace_bytes(R_cap, hw, n_obj, C), aoc_masked_mean_pool_workspace_bytes(1, hw, n_obj, C));
    f.pool_ws = take(f.pool_ws_bytes);
    f.total = off;
    return f;
}

inline int frame_sets(const aoc_frame_desc *d) { return 2 * d->n_levels * d->n_obj + d->n_obj; }

}  // namespace

extern "C" {

int aoc_frame_channels(int n_radii, int n_levels, int matching_background) {
    if (n_radii < 1 || n_levels < 1) return 0;
    return 2 + 2 * n_levels + 2 * n_radii + 1 + (matching_background ? n_radii + 1 : 0);       // aocnet.py:43-46 (+ 2 per further level)
}

size_t aoc_frame_workspace_bytes(int h, int w, int C, int n_obj, int R_capacity, int n_radii, int n_levels) {
    if (h < 1 || w < 1 || C < 1 || n_obj < 1 || R_capacity < 1 || n_radii < 1 || n_levels < 1) return 0;
    if (aoc_split_record_bytes(C) == 0) return 0;
    return frame_carve(nullptr, h, w, C, n_obj, R_capacity, n_radii, 2 * n_levels * n_obj + n_obj).total;
}

int aoc_frame_enqueue(const aoc_frame_desc *d, aoc_seq_state *state, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!d || !state || !workspace) return AOC_ERR_INVALID_ARG;
    if (!d->ref_emb || !d->ref_labels || !d->prev_emb || !d->prev_labels || !d->cur_emb || !d->dis_bias || !d->right_bits || !d->wrong_bits ||
        !d->fg_rows || !d->obj_rows || !d->counts || !d->obj_offsets || !d->proxy_table || !d->proxy_sqnorm || !d->feat || !d->head)
        return AOC_ERR_INVALID_ARG;
    if (d->h < 1 || d->w < 1 || d->n_obj < 1 || d->R < 1 || d->R > d->R_capacity || d->n_radii < 1 || d->n_radii > 8 || d->n_levels < 1 ||
        d->n_levels > 8 || d->kmax < 1 || d->pool_key == 0)
        return AOC_ERR_INVALID_ARG;
    // the widths this call covers (everything else goes through the individual entry points): C = 100 (split records), <= AOC_MAX_OBJECTS objects
    if (d->C != 100 || d->n_obj > AOC_MAX_OBJECTS || aoc_split_record_bytes(d->C) == 0) return AOC_ERR_UNSUPPORTED;
    const int h = d->h, w = d->w, C = d->C, O = d->n_obj, R = d->R, nl = d->n_radii, L = d->n_levels, kmax = d->kmax;
    const int64_t hw = (int64_t)h * w;
    const int n_set = frame_sets(d);
    const int n_ad = L * O * 2 * kmax;
    if (d->n_adaptive != n_ad) return AOC_ERR_INVALID_ARG;
    // everything a later stage would reject is rejected HERE, before the first launch and before the state record is touched: the cluster
    // levels (set sizes of the correlation launch), the window radii (aoc_local_window_match_pair: ascending, window inside the kernel's reach)
    if (d->pool_prefix_frames < 0 || d->stream_cus < 0 || d->local_atrous_rate < 0 || d->match_hw < 0 || d->match_hw > hw) return AOC_ERR_INVALID_ARG;
    if ((d->match_hw > 0) != (d->match_emb != nullptr)) return AOC_ERR_INVALID_ARG;
    for (int l = 0; l < L; ++l)
        if (d->levels[l] < 1 || d->levels[l] > kmax) return AOC_ERR_INVALID_ARG;
    for (int i = 0; i < nl; ++i)
        if (d->radii[i] < 0 || (i > 0 && d->radii[i] <= d->radii[i - 1])) return AOC_ERR_INVALID_ARG;
    if (d->radii[nl - 1] > 31) return AOC_ERR_UNSUPPORTED;
    if (kmax > AOC_MAX_CLUSTERS) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_frame_workspace_bytes(h, w, C, O, d->R_capacity, nl, L)) return AOC_ERR_WORKSPACE;
    const FrameWs f = frame_carve(workspace, h, w, C, O, d->R_capacity, nl, n_set);
    hipStream_t st = aoc_hip_stream(stream);
    const int H2 = h / 2 + 1, W2 = w / 2 + 1;
    const int n_ch = aoc_frame_channels(nl, L, d->matching_background);
    const int64_t obj_stride = (int64_t)n_ch * hw;
    // the reference's switches; `fused` = the default configuration, which keeps the fused launches of round 4
    const bool f16 = d->float16_matching != 0, down = d->local_downsample != 0;
    const int lrate = d->local_atrous_rate > 1 ? d->local_atrous_rate : 1;
    const bool split = !f16 && O <= 16;                                 // dense + correlation on the fp16-split matrix pipe with records
    const bool fused_local = !f16 && down && lrate == 1;
    // the pool the dense and cluster matchings see (the atrous-sub-sampled pool with TEST_GLOBAL_ATROUS_RATE > 1)
    const float *mpool = d->match_hw > 0 ? d->match_emb : d->ref_emb;
    const int64_t mhw = d->match_hw > 0 ? d->match_hw : hw, n = mhw * R;
    // channel layout, aocnet.py:355-358
    const int c2 = 2 * L;
    const int ch_global = 0, ch_cluster = 1, ch_proxy = 1 + c2, ch_local = 2 + c2, ch_local_proxy = 2 + c2 + nl, ch_prev = 2 + c2 + 2 * nl;
    const int ch_local_bg = d->matching_background ? 3 + c2 + 2 * nl : -1, ch_global_bg = d->matching_background ? 3 + c2 + 3 * nl : -1;
    int rc;
#define AOC_TRY(call) do { rc = (call); if (rc != AOC_OK) return rc; } while (0)
    auto mark = [&](int i) { if (d->probe[i]) (void)hipEventRecord(static_cast<hipEvent_t>(d->probe[i]), st); };
    // the CU budget of this call's stream (a caller that runs it under a HIP CU mask), for the duration of the call on this thread
    struct CuScope {
        int before;
        explicit CuScope(int n_cus) : before(aoc_stream_cus_scope(n_cus)) {}
        ~CuScope() { aoc_stream_cus_scope(before); }
    } cu_scope(d->stream_cus > 0 ? d->stream_cus : 0);

    if (!state->initialised) {
        if (hipMemsetAsync(workspace, 0, f.init_bytes, st) != hipSuccess) return AOC_ERR_LAUNCH;
        state->initialised = 1;
        state->records_frames = 0;
        state->ref_pool_key = 0;
        state->plan_key = 0;
        state->corr_tables_key = 0;
    }
    if (d->prep_ready && hipStreamWaitEvent(st, static_cast<hipEvent_t>(d->prep_ready), 0) != hipSuccess) return AOC_ERR_LAUNCH;

    // ---- k = 1 proxies (ATT:155-189): the pooled reference heads once per pool state, the previous frame's every frame
    if (state->ref_pool_key != d->pool_key) {
        AOC_TRY(aoc_masked_mean_pool(d->ref_emb, d->ref_labels, R, hw, C, O, 1, d->epsilon, f.ref_pos, f.ref_neg, f.ref_sq, f.pool_ws, f.pool_ws_bytes, stream));
        state->ref_pool_key = d->pool_key;
    }
    AOC_TRY(aoc_masked_mean_pool(d->prev_emb, d->prev_labels, 1, hw, C, O, 1, d->epsilon, f.prev_pos, f.prev_neg, nullptr, f.pool_ws, f.pool_ws_bytes, stream));

    // ---- half-resolution operands of the local matchings + the per-set bias table + the k = 1 rows of this frame's proxy table (the side
    // tables are needed in every mode; the half-resolution maps only by the fp32 down-sampled local matching)
    AOC_TRY(aoc_local_prep(d->cur_emb, d->prev_emb, d->prev_labels, f.prev_pos, h, w, C, O, f.q2, f.p2, f.pm2, f.bits2, H2, W2, d->dis_bias, 2 * L * O,
                           f.set_bias, f.ref_pos, d->proxy_table + (size_t)n_ad * C, O * C, f.ref_sq, d->proxy_sqnorm + n_ad, O, stream));

    // ---- split records: pool frames that joined since the last call, and the query (tile-major)
    if (split) {
        if (state->records_frames > R) state->records_frames = 0;             // the pool restarted: the caller should have reset the state
        // records behind the unchanged prefix belong to frames whose content was replaced (a pool that is not append-only): converted again
        if (state->records_frames > d->pool_prefix_frames) state->records_frames = d->pool_prefix_frames;
        if (state->records_frames < R) {
            const int64_t r0 = state->records_frames * mhw;
            AOC_TRY(aoc_split_rows(mpool + (size_t)r0 * C, n - r0, C, f.pool_rec + (size_t)r0 * aoc_split_record_bytes(C), f.pool_sq + r0, f.overflow, stream));
            state->records_frames = R;
        }
        AOC_TRY(aoc_split_rows_tiled(d->cur_emb, hw, C, f.q_rec, f.q_sq, f.overflow, stream));
    }

    // ---- dense pixel-level matching -> channel 0
    mark(0);
    if (split) {
        // (the plan is kept across the frames of one pool state)
        const int reuse = state->plan_key == d->pool_key && state->plan_rows == n;
        AOC_TRY(aoc_dense_match_min_split_cached(d->cur_emb, f.q_rec, f.q_sq, 1, hw, C, mpool, f.pool_rec, f.overflow, n, d->right_bits, d->wrong_bits,
                                                 d->fg_rows, d->obj_rows, d->counts, d->obj_offsets, d->dis_bias, O, d->feat + (size_t)ch_global * hw, 1,
                                                 obj_stride, 1, f.dense_ws, f.dense_bytes, reuse, stream));
        state->plan_key = d->pool_key;
        state->plan_rows = n;
    } else if (f16) {
        // the reference's `.half()` arithmetic (AEM:801-803), as matching.global_matching_for_eval(use_float16=True) runs it
        AOC_TRY(aoc_dense_match_min_f16(d->cur_emb, hw, C, mpool, d->fg_rows, d->counts + O, n, d->wrong_bits, d->dis_bias, O, d->feat + (size_t)ch_global * hw, 1,
                                        obj_stride, 1, f.dense_ws, f.dense_bytes, stream));
    } else {
        // more than 16 objects: the exact-fp32 kernel (slices of 16 objects), as ops.dense_match picks it
        AOC_TRY(aoc_dense_match_min(d->cur_emb, hw, C, mpool, d->fg_rows, d->counts + O, n, d->wrong_bits, d->dis_bias, O, d->feat + (size_t)ch_global * hw, 1,
                                    obj_stride, 1, f.dense_ws, f.dense_bytes, stream));
    }
    mark(1);

    // ---- both local matchings and their up-samples into the two channel ranges
    mark(4);
    if (fused_local) {
        AOC_TRY(aoc_local_window_match_pair(f.q2, f.p2, f.pm2, f.bits2, H2, W2, C, d->radii, nl, d->dis_bias, O, f.lf, f.lf + (size_t)O * nl * H2 * W2, 1, stream));
        mark(5);
        AOC_TRY(aoc_resize_bilinear_planes_grouped(f.lf, 2 * O * nl, H2, W2, d->feat + (size_t)ch_local * hw, h, w, nl, O, (int64_t)(ch_local_proxy - ch_local) * hw,
                                                   obj_stride, hw, 1, stream));
    } else {
        // matching.local_matching / local_matching_proxy step by step (AEM:968-1060): label bits, the per-pixel proxy map (aocnet.py:325), the
        // down-samples in the mode's arithmetic (float16: F.interpolate on a float16 tensor), the windows with the atrous rate
        const float *qm = d->cur_emb, *pa = d->prev_emb, *pb = f.pmap;
        const uint32_t *bits = f.bits_full;
        int Hm = h, Wm = w;
        AOC_TRY(aoc_label_bits(d->prev_labels, hw, O, f.bits_full, nullptr, stream));
        if (down) {
            Hm = H2; Wm = W2;
            bits = f.bits2;
            if (f16) {
                AOC_TRY(aoc_label_mix(d->prev_labels, f.prev_pos, hw, O, C, f.pmap, stream));
                AOC_TRY(aoc_resize_bilinear_hwc_ex(d->cur_emb, h, w, C, f.q2, H2, W2, 1, stream));
                AOC_TRY(aoc_resize_bilinear_hwc_ex(d->prev_emb, h, w, C, f.p2, H2, W2, 1, stream));
                AOC_TRY(aoc_resize_bilinear_hwc_ex(f.pmap, h, w, C, f.pm2, H2, W2, 1, stream));
                AOC_TRY(aoc_resize_nearest_bits(f.bits_full, h, w, f.bits2, H2, W2, stream));
            }                                   // fp32: aoc_local_prep above left the same q2 / p2 / pm2 / bits2 the separate calls give
            qm = f.q2; pa = f.p2; pb = f.pm2;
        } else {
            AOC_TRY(aoc_label_mix(d->prev_labels, f.prev_pos, hw, O, C, f.pmap, stream));
        }
        float *lf_b = f.lf + (size_t)O * nl * Hm * Wm;
        AOC_TRY(aoc_local_window_match_ex(qm, pa, bits, Hm, Wm, C, d->radii, nl, d->dis_bias, O, f.lf, 1, lrate, f16 ? 1 : 0, stream));
        AOC_TRY(aoc_local_window_match_ex(qm, pb, bits, Hm, Wm, C, d->radii, nl, d->dis_bias, O, lf_b, 1, lrate, f16 ? 1 : 0, stream));
        mark(5);
        AOC_TRY(aoc_resize_bilinear_planes_grouped(f.lf, 2 * O * nl, Hm, Wm, d->feat + (size_t)ch_local * hw, h, w, nl, O, (int64_t)(ch_local_proxy - ch_local) * hw,
                                                   obj_stride, hw, 1, stream));
    }

    // ---- correlation: cluster sets (2 per object and level) + the k = 1 set of every object
    {
        int32_t sb[2 * 8 * AOC_MAX_OBJECTS + AOC_MAX_OBJECTS], ss[2 * 8 * AOC_MAX_OBJECTS + AOC_MAX_OBJECTS];
        int64_t so[2 * 8 * AOC_MAX_OBJECTS + AOC_MAX_OBJECTS];
        int s = 0;
        for (int l = 0; l < L; ++l)
            for (int o = 0; o < O; ++o)
                for (int t = 0; t < 2; ++t, ++s) {
                    sb[s] = ((l * O + o) * 2 + t) * kmax;
                    ss[s] = d->levels[l];                                   // slots beyond the sticky K carry norm = +inf
                    so[s] = (int64_t)o * obj_stride + (int64_t)(ch_cluster + 2 * l + t) * hw;
                }
        for (int o = 0; o < O; ++o, ++s) {
            sb[s] = n_ad + o;
            ss[s] = 1;
            so[s] = (int64_t)o * obj_stride + (int64_t)ch_proxy * hw;
        }
        if (d->proxies_ready && hipStreamWaitEvent(st, static_cast<hipEvent_t>(d->proxies_ready), 0) != hipSuccess) return AOC_ERR_LAUNCH;
        mark(2);
        if (f16) {
            // use_float16: scipy's kmeans2 rejects float16 data, the reference's bare `except` turns every cluster distance into 5e4, i.e. the
            // feature 1.0: the 2 L cluster channels of every object are that constant; the k = 1 proxies run the `.half()` arithmetic
            for (int o = 0; o < O; ++o)
                if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d->feat + (size_t)o * obj_stride + (size_t)ch_cluster * hw), 0x3f800000, (size_t)c2 * hw, st) !=
                    hipSuccess)
                    return AOC_ERR_LAUNCH;
            AOC_TRY(aoc_proxy_corr_min_f16(d->cur_emb, hw, C, d->proxy_table, nullptr, n_ad + O, O, sb + 2 * L * O, ss + 2 * L * O, so + 2 * L * O,
                                           f.set_bias + 2 * L * O, d->feat, 1, 1, stream));
        } else if (split) {
            aoc_corr_frame_rec fr;
            fr.query = d->cur_emb;
            fr.query_rec = f.q_rec;
            fr.query_sqnorm = f.q_sq;
            fr.proxies = d->proxy_table;
            fr.proxy_sqnorm = d->proxy_sqnorm;
            fr.set_bias = f.set_bias;
            fr.out = d->feat;
            // one launch whatever the number of proxy tiles; the passes' tile tables are written into the workspace once per sequence
            AOC_TRY(aoc_proxy_corr_min_records_cached(&fr, 1, hw, C, n_ad + O, n_set, sb, ss, so, 1, f.corr_ws, aoc_proxy_corr_min_records_cached_workspace_bytes(),
                                                      &state->corr_tables_key, stream));
        } else {
            aoc_corr_frame fr;
            fr.query = d->cur_emb;
            fr.proxies = d->proxy_table;
            fr.proxy_sqnorm = d->proxy_sqnorm;
            fr.set_bias = f.set_bias;
            fr.out = d->feat;
            AOC_TRY(aoc_proxy_corr_min_batched(&fr, 1, hw, C, n_ad + O, n_set, sb, ss, so, 1, AOC_CORR_SPLIT, f.corr_ws, aoc_proxy_corr_min_batched_workspace_bytes(), stream));
        }
        mark(3);
    }

    // ---- background maps, previous-mask channel, attention head
    AOC_TRY(aoc_proto_finish(d->feat, O, hw, obj_stride, ch_local, nl, ch_local_bg, ch_global, ch_global_bg, ch_prev, d->prev_labels, f.ref_pos, f.ref_neg,
                             f.prev_pos, f.prev_neg, C, d->head, stream));
#undef AOC_TRY
    return AOC_OK;
}

// ------------------------------------------------------------------------------------------
// The k-means chain of the frames that see one pool state as ONE call (see include/aoc_hip.h): the three entry points the Python orchestrator
// called one by one, out of one workspace, plus one launch that scatters every frame's proxies into its own proxy table.
namespace {
struct ChainWs {
    int32_t *rows_f, *off_f, *k_f, *labels, *ccounts;
    float *centroids, *proxies, *psq;
    char *km_ws, *bp_ws;
    size_t km_bytes, bp_bytes, total;
    int64_t off[7];
};
inline ChainWs chain_carve(void *base, const aoc_chain_desc *d) {
    ChainWs w;
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes, int64_t *where) { char *r = p ? p + off : nullptr; if (where) *where = (int64_t)off; off += aoc_align_up(bytes, 256); return r; };
    const int FL = d->n_frames * d->n_levels, S = FL * d->n_obj;
    const int64_t cap = (int64_t)FL * d->rows_capacity;
    w.centroids = reinterpret_cast<float *>(take((size_t)S * d->kmax * d->C * sizeof(float), &w.off[0]));
    w.labels = reinterpret_cast<int32_t *>(take((size_t)cap * sizeof(int32_t), &w.off[1]));
    w.ccounts = reinterpret_cast<int32_t *>(take((size_t)S * d->kmax * sizeof(int32_t), &w.off[2]));
    w.proxies = reinterpret_cast<float *>(take((size_t)S * 2 * d->kmax * d->C * sizeof(float), &w.off[3]));
    w.psq = reinterpret_cast<float *>(take((size_t)S * 2 * d->kmax * sizeof(float), &w.off[4]));
    w.k_f = reinterpret_cast<int32_t *>(take((size_t)S * sizeof(int32_t), &w.off[5]));
    w.off_f = reinterpret_cast<int32_t *>(take((size_t)(S + 1) * sizeof(int32_t), &w.off[6]));
    w.rows_f = FL > 1 ? reinterpret_cast<int32_t *>(take((size_t)cap * sizeof(int32_t), nullptr)) : nullptr;      // one replica: the label prep's own list
    w.km_bytes = aoc_kmeans_workspace_bytes(cap, S, d->kmax, d->C);
    w.km_ws = take(w.km_bytes, nullptr);
    w.bp_bytes = aoc_build_proxies_workspace_bytes(cap, S, d->kmax);
    w.bp_ws = take(w.bp_bytes, nullptr);
    w.total = off;
    return w;
}
inline bool chain_desc_ok(const aoc_chain_desc *d) {
    if (!d || d->C < 1 || d->n_obj < 1 || d->n_frames < 1 || d->n_frames > 8 || d->n_levels < 1 || d->n_levels > 8 || d->kmax < 1 || d->iters < 1) return false;
    if (d->pool_rows < 1 || d->rows_capacity < 1 || (int64_t)d->n_frames * d->n_levels * d->rows_capacity >= (1ll << 31)) return false;
    for (int l = 0; l < d->n_levels; ++l)
        if (d->levels[l] < 1 || d->levels[l] > d->kmax) return false;
    return true;
}
struct ChainTables {
    float *table[8], *sqn[8];
};
// tables[f][row, :] = proxies[f * rows_per_frame + row, :], sqnorms[f][row] = psq[...]   (rows_per_frame = levels * objects * 2 * kmax)
__global__ __launch_bounds__(256) void chain_scatter_kernel(const float *__restrict__ proxies, const float *__restrict__ psq, ChainTables t, int n_frames,
                                                             int rows_per_frame, int C4) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)rows_per_frame * C4;
    if (idx >= per * n_frames) return;
    const int f = (int)(idx / per);
    const int64_t rem = idx - (int64_t)f * per;
    reinterpret_cast<float4 *>(t.table[f])[rem] = reinterpret_cast<const float4 *>(proxies)[idx];
    if (rem < rows_per_frame) t.sqn[f][rem] = psq[(int64_t)f * rows_per_frame + rem];
}
}  // namespace

size_t aoc_cluster_chain_workspace_bytes(const aoc_chain_desc *d) { return chain_desc_ok(d) ? chain_carve(nullptr, d).total : 0; }

int aoc_cluster_chain_layout(const aoc_chain_desc *d, int64_t *offsets7) {
    if (!chain_desc_ok(d) || !offsets7) return AOC_ERR_INVALID_ARG;
    const ChainWs w = chain_carve(nullptr, d);
    for (int i = 0; i < 7; ++i) offsets7[i] = w.off[i];
    return AOC_OK;
}

int aoc_cluster_chain_enqueue(const aoc_chain_desc *d, void *workspace, size_t workspace_bytes, aoc_stream_t stream) {
    if (!chain_desc_ok(d) || !workspace) return AOC_ERR_INVALID_ARG;
    if (!d->pool || !d->fg_rows || !d->obj_rows || !d->obj_offsets || !d->init_rows) return AOC_ERR_INVALID_ARG;
    if (d->C > AOC_MAX_CHANNELS || d->kmax > AOC_MAX_CLUSTERS || d->n_obj > AOC_MAX_OBJECTS || (d->C & 3)) return AOC_ERR_UNSUPPORTED;
    for (int f = 0; f < d->n_frames; ++f)
        if (!d->tables[f] || !d->sqnorms[f] || (reinterpret_cast<uintptr_t>(d->tables[f]) & 15)) return AOC_ERR_INVALID_ARG;
    if (workspace_bytes < aoc_cluster_chain_workspace_bytes(d)) return AOC_ERR_WORKSPACE;
    const ChainWs w = chain_carve(workspace, d);
    const int FL = d->n_frames * d->n_levels, S = FL * d->n_obj;
    const int64_t cap = (int64_t)FL * d->rows_capacity;
    int rc;
    // the replicated segment lists with the sticky K of every (frame, level) (AEM:268) -- one replica: offsets and K only, the rows stay where they are
    int32_t *rows_f = FL > 1 ? w.rows_f : const_cast<int32_t *>(d->obj_rows);
    rc = aoc_kmeans_replicate_levels(d->obj_rows, d->obj_offsets, d->n_obj, FL, d->levels, d->n_levels, d->rows_capacity, rows_f, w.off_f, w.k_f, stream);
    if (rc != AOC_OK) return rc;
    rc = aoc_kmeans_segmented_rep(d->pool, d->pool_rows, d->C, rows_f, w.off_f, w.k_f, d->init_rows, S, FL, d->kmax, d->iters, cap, w.centroids, w.labels, w.ccounts,
                                  w.km_ws, w.km_bytes, stream);
    if (rc != AOC_OK) return rc;
    rc = aoc_build_proxies(d->pool, d->pool_rows, d->C, d->fg_rows, w.off_f, w.k_f, w.labels, w.centroids, S, d->kmax, cap, w.proxies, w.psq, w.bp_ws, w.bp_bytes, stream);
    if (rc != AOC_OK) return rc;
    ChainTables t;
    for (int f = 0; f < 8; ++f) { t.table[f] = f < d->n_frames ? d->tables[f] : nullptr; t.sqn[f] = f < d->n_frames ? d->sqnorms[f] : nullptr; }
    const int rows_per_frame = d->n_levels * d->n_obj * 2 * d->kmax;
    const int64_t total = (int64_t)rows_per_frame * (d->C / 4) * d->n_frames;
    hipLaunchKernelGGL(chain_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), w.proxies, w.psq, t, d->n_frames,
                       rows_per_frame, d->C / 4);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

// ------------------------------------------------------------------------------------------
// The modulation gates of CalibrationDecoding.forward (decoding_module.py:96-149, 162-210) as ONE call: the reference applies them inside a
// single forward call; here a list of gate descriptors is walked and each gate issues exactly the launches of its module mirror
// (attention.IA_gate / hotpath's extended-head gate / conditioning_layer.conditioning_block), so the outputs are bit-identical to the modules'.
namespace {
struct GateWs { float *pm, *hx, *gap, *code; char *cond; size_t cond_bytes, total; };
inline GateWs gate_carve(void *base, const aoc_gate_desc *g, int n, int n_obj, int D) {
    GateWs w;
    int cmax = 1;
    size_t cond = 0;
    for (int i = 0; i < n; ++i) {
        cmax = std::max(cmax, (int)g[i].channels);
        if (g[i].kind == 2) cond = std::max(cond, aoc_cond_gate_pool_workspace_bytes(n_obj, g[i].channels, g[i].hw));
    }
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += aoc_align_up(bytes, 256); return r; };
    w.pm = reinterpret_cast<float *>(take((size_t)n_obj * cmax * sizeof(float)));
    w.gap = reinterpret_cast<float *>(take((size_t)n_obj * cmax * sizeof(float)));
    w.hx = reinterpret_cast<float *>(take((size_t)n_obj * (D + cmax) * sizeof(float)));
    w.code = reinterpret_cast<float *>(take((size_t)n_obj * (2 * cmax + D) * sizeof(float)));
    w.cond_bytes = cond;
    w.cond = take(cond > 0 ? cond : 16);
    w.total = off;
    return w;
}
}  // namespace

size_t aoc_gates_workspace_bytes(const aoc_gate_desc *gates, int n_gates, int n_obj, int head_dim) {
    if (!gates || n_gates < 1 || n_obj < 1 || head_dim < 1) return 0;
    return gate_carve(nullptr, gates, n_gates, n_obj, head_dim).total;
}

int aoc_gates_enqueue(const aoc_gate_desc *gates, int n_gates, const float *head, int n_obj, int head_dim, void *workspace, size_t workspace_bytes,
                      aoc_stream_t stream) {
    if (!gates || !head || !workspace || n_gates < 1 || n_obj < 1 || head_dim < 1) return AOC_ERR_INVALID_ARG;
    if (workspace_bytes < aoc_gates_workspace_bytes(gates, n_gates, n_obj, head_dim)) return AOC_ERR_WORKSPACE;
    const GateWs w = gate_carve(workspace, gates, n_gates, n_obj, head_dim);
    const int D = head_dim;
    hipStream_t st = aoc_hip_stream(stream);
    int rc;
    for (int i = 0; i < n_gates; ++i) {
        const aoc_gate_desc &g = gates[i];
        auto mark = [&](int k) { if (g.probe[k]) (void)hipEventRecord(static_cast<hipEvent_t>(g.probe[k]), st); };
        if (!g.x || !g.y || !g.w || g.channels < 1 || g.hw < 1) return AOC_ERR_INVALID_ARG;
        const int c = g.channels;
        if (g.kind == 0) {                          // IA_gate, ATT:7-17
            mark(0);
            rc = aoc_film_scale(g.x, head, g.w, g.b, n_obj, D, c, g.hw, g.y, stream);
            mark(1);
        } else if (g.kind == 1) {                   // IA gate whose head carries the inter-object code of its input (decoding_module.py:126-130)
            rc = aoc_plane_mean(g.x, (int64_t)n_obj * c, g.hw, w.pm, stream);
            if (rc == AOC_OK) rc = aoc_head_delta(head, D, w.pm, n_obj, c, w.hx, stream);
            mark(0);
            if (rc == AOC_OK) rc = aoc_film_scale(g.x, w.hx, g.w, g.b, n_obj, D + c, c, g.hw, g.y, stream);
            mark(1);
        } else if (g.kind == 2) {                   // conditioning_block, CLB:50-86
            if (!g.phi_w || !g.phi_b || !g.w1 || !g.b1 || !g.w2 || !g.b2 || !g.w3 || !g.b3) return AOC_ERR_INVALID_ARG;
            mark(2);
            rc = aoc_cond_gate_pool_ex(g.x, n_obj, c, g.hw, g.phi_w, g.phi_b, g.k_rank, w.gap, w.pm, nullptr, nullptr, w.cond, w.cond_bytes, stream);
            mark(3);
            if (rc == AOC_OK) rc = aoc_cond_codes(w.gap, w.pm, head, g.w1, g.b1, g.w2, g.b2, g.w3, g.b3, n_obj, c, D, w.code, stream);
            mark(0);
            if (rc == AOC_OK) rc = aoc_film_scale(g.x, w.code, g.w, g.b, n_obj, 2 * c + D, c, g.hw, g.y, stream);
            mark(1);
        } else {
            return AOC_ERR_INVALID_ARG;
        }
        if (rc != AOC_OK) return rc;
    }
    return AOC_OK;
}

}  // extern "C"
