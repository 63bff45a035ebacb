// Eval-loop memory policy: the per-pixel label decision of one frame and the label-map preparation
// of the matching path.  Reference: networks/engine/eval_manager_mm.py:253-265 (never-seen labels), :300-326 (argmax,
// new-object join), :305-306 + networks/layers/shannon_entropy.py:10-13 (Shannon entropy), :339-346 / :357-361
// (uncertain pixels get label 125); networks/aoc/aocnet.py:128-133,151 (nearest resize + one-hot of the label maps).
#include "aoc_common.h"

namespace {

// labels_out = argmax_c probs[c] over ALL channels with never-seen channels zeroed (first maximum, torch.argmax), then the
// join override; confident_out = the same label, or 125 where the entropy over the SEEN channels exceeds unc_ratio.
__global__ __launch_bounds__(256) void confident_labels_kernel(const float *__restrict__ probs, int n_ch, int64_t n, uint32_t exist_bits,
                                                                const int32_t *__restrict__ join_label, float unc_ratio,
                                                                int32_t *__restrict__ labels_out, int32_t *__restrict__ confident_out,
                                                                float *__restrict__ entropy_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = 0.0f, ent = 0.0f;
    int arg = 0;
    for (int c = 0; c < n_ch; ++c) {
        const bool seen = (exist_bits >> c) & 1u;
        const float p = seen ? probs[(size_t)c * n + i] : 0.0f;            // :257-259 zeros_like for unseen labels
        if (c == 0 || p > best) { best = p; arg = c; }
        if (seen) ent += p * logf(p + 1e-6f);                               // shannon_entropy.py:11
    }
    ent = -1.0f * ent;
    int label = arg;
    if (join_label) {
        const int jl = join_label[i];
        const int keep = (jl == 0) ? 1 : 0;                                 // :321-323
        label = label * keep + jl * (1 - keep);
        ent = ent * (float)keep + ((jl < 0) ? 1.0f : 0.0f) * (float)(1 - keep);   // :341-343
    }
    const int region = (ent > unc_ratio) ? 1 : 0;                           // :345
    labels_out[i] = label;
    if (confident_out) confident_out[i] = label * (1 - region) + 125 * region;   // :346
    if (entropy_out) entropy_out[i] = ent;
}

// out[y, x, o] = (label[nearest(y, x)] == o) ? 1 : 0   (aocnet.py:128-133 interpolate(mode='nearest') + :151 ==ref_obj_ids)
__global__ __launch_bounds__(256) void label_onehot_nearest_kernel(const int32_t *__restrict__ label, int H, int W, int h, int w, int n_obj,
                                                                    float sh, float sw, float *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= h * w) return;
    const int x = idx % w, y = idx / w;
    const int sy = min((int)floorf((float)y * sh), H - 1);   // torch nearest: floor(dst * in/out)
    const int sx = min((int)floorf((float)x * sw), W - 1);
    const int l = label[(size_t)sy * W + sx];
    float *o = out + (size_t)idx * n_obj;
    for (int c = 0; c < n_obj; ++c) o[c] = (l == c) ? 1.0f : 0.0f;
}

// ---- J (region similarity) and F (boundary measure) of a predicted label map against the ground truth, per foreground object,
// accumulated on the device so that an evaluation never reads a mask back (the reference saves PNGs and scores them with the external
// DAVIS toolkit, README.md:110; its only in-repo IoU is utils/metric.py:3-34).  Definitions: DAVIS-2017 db_eval_iou / db_eval_boundary
// (seg2bmap boundaries, dilation by a disk of bound_pix, F = 2 P R / (P + R)); restated in oracle/metrics.py.
//
// bits[y, x]: bit o = pixel is a boundary pixel of the binary mask (label == o), low 16 bits for `pred`, high 16 bits for `gt`.
// Counter updates: every wave counts its lanes per (counter, object) with ballots and adds the totals to the workgroup's LDS counters; one
// global atomic per non-zero (counter, object) and workgroup (a global atomic per pixel on a dozen addresses serialised the whole frame:
// 1.36 ms at 480 x 854 against ~10 us).
__device__ __forceinline__ void jf_count(int32_t *__restrict__ lds_counts, int row, int n_obj, uint32_t obj_bits) {
    // obj_bits: bit o set = this lane counts for object o in counter row `row`
    for (int o = 1; o < n_obj; ++o) {
        const unsigned long long m = __ballot((obj_bits >> o) & 1u);
        if (m != 0ull && aoc_lane() == 0) atomicAdd(&lds_counts[row * 16 + o], __popcll(m));
    }
}

__global__ __launch_bounds__(256) void jf_boundary_kernel(const int32_t *__restrict__ pred, const int32_t *__restrict__ gt, int H, int W, int n_obj,
                                                           uint32_t *__restrict__ bits, int32_t *__restrict__ counts) {
    __shared__ int32_t lc[3 * 16];
    if (threadIdx.x < 3 * 16) lc[threadIdx.x] = 0;
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < H * W;
    uint32_t in_pred = 0, in_gt = 0;
    if (live) {
        const int x = idx % W, y = idx / W;
        const bool last_row = y == H - 1, last_col = x == W - 1;
        uint32_t out = 0;
        for (int side = 0; side < 2; ++side) {
            const int32_t *m = side ? gt : pred;
            const int c = m[idx];
            const int e = last_col ? c : m[idx + 1];
            const int s = last_row ? c : m[idx + W];
            const int se = (last_row || last_col) ? c : m[idx + W + 1];
            for (int o = 1; o < n_obj; ++o) {
                const bool mc = c == o;
                bool b;
                if (last_row && last_col) b = false;                           // b[-1, -1] = 0
                else if (last_row) b = mc != (e == o);                          // b[-1, :] = seg ^ e
                else if (last_col) b = mc != (s == o);                          // b[:, -1] = seg ^ s
                else b = (mc != (e == o)) || (mc != (s == o)) || (mc != (se == o));
                if (b) out |= 1u << (o + 16 * side);
            }
            // region counts: |pred = o|, |gt = o|, |both|
            if (c >= 1 && c < n_obj) (side ? in_gt : in_pred) = 1u << c;
        }
        bits[idx] = out;
    }
    jf_count(lc, 0, n_obj, in_pred);
    jf_count(lc, 1, n_obj, in_gt);
    jf_count(lc, 2, n_obj, in_pred & in_gt);
    __syncthreads();
    if (threadIdx.x < 3 * 16 && lc[threadIdx.x] != 0) atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]);
}

// boundary matches within a disk of radius `r`: counts[3][o] += pred boundary pixels with a gt boundary pixel of o nearby,
// counts[4][o] the other way round, counts[5][o] / counts[6][o] the boundary pixel totals
__global__ __launch_bounds__(256) void jf_match_kernel(const uint32_t *__restrict__ bits, int H, int W, int n_obj, int r, int32_t *__restrict__ counts) {
    __shared__ int32_t lc[4 * 16];                  // rows 3..6 of `counts`
    if (threadIdx.x < 4 * 16) lc[threadIdx.x] = 0;
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t here = idx < H * W ? bits[idx] : 0u;
    uint32_t near = 0;
    if (here != 0) {
        const int x = idx % W, y = idx / W;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W || dx * dx + dy * dy > r * r) continue;   // skimage.morphology.disk(r)
                near |= bits[(size_t)yy * W + xx];
            }
        }
    }
    if (__ballot(here != 0) != 0ull) {               // wave-uniform: most waves hold no boundary pixel
        const uint32_t pb = here & 0xffffu, gb = here >> 16, ngt = near >> 16, npr = near & 0xffffu;
        jf_count(lc, 0, n_obj, pb & ngt);            // counts[3]: pred boundary pixels with a gt boundary pixel of the object nearby
        jf_count(lc, 1, n_obj, gb & npr);            // counts[4]: the other way round
        jf_count(lc, 2, n_obj, pb);                  // counts[5], counts[6]: boundary pixel totals
        jf_count(lc, 3, n_obj, gb);
    }
    __syncthreads();
    if (threadIdx.x < 4 * 16 && lc[threadIdx.x] != 0) atomicAdd(&counts[3 * 16 + threadIdx.x], lc[threadIdx.x]);
}

// The same on a 32 x 8 pixel tile whose bits (+ a halo of r pixels, zeros outside the image) are staged in LDS once: a boundary pixel's disk
// of (2 r + 1)^2 probes then reads LDS instead of global memory (r = 8 at 480p: 289 probes per boundary pixel).  r <= JF_R_LDS.
constexpr int JF_TW = 32, JF_TH = 8, JF_R_LDS = 32;
__global__ __launch_bounds__(256) void jf_match_tile_kernel(const uint32_t *__restrict__ bits, int H, int W, int n_obj, int r, int32_t *__restrict__ counts) {
    extern __shared__ uint32_t jf_tile[];            // [(JF_TH + 2 r)][(JF_TW + 2 r)]
    __shared__ int32_t lc[4 * 16];
    if (threadIdx.x < 4 * 16) lc[threadIdx.x] = 0;
    const int tw = JF_TW + 2 * r, th = JF_TH + 2 * r;
    const int x0 = blockIdx.x * JF_TW - r, y0 = blockIdx.y * JF_TH - r;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int yy = y0 + ty, xx = x0 + tx;
        jf_tile[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? bits[(size_t)yy * W + xx] : 0u;
    }
    __syncthreads();
    const int lx = threadIdx.x & (JF_TW - 1), ly = threadIdx.x / JF_TW;
    const int x = blockIdx.x * JF_TW + lx, y = blockIdx.y * JF_TH + ly;
    const uint32_t here = (x < W && y < H) ? jf_tile[(ly + r) * tw + lx + r] : 0u;
    uint32_t near = 0;
    if (here != 0) {
        for (int dy = -r; dy <= r; ++dy) {
            const uint32_t *row = jf_tile + (ly + r + dy) * tw + lx + r;
            for (int dx = -r; dx <= r; ++dx)
                if (dx * dx + dy * dy <= r * r) near |= row[dx];               // skimage.morphology.disk(r); outside the image: zeros
        }
    }
    if (__ballot(here != 0) != 0ull) {
        const uint32_t pb = here & 0xffffu, gb = here >> 16, ngt = near >> 16, npr = near & 0xffffu;
        jf_count(lc, 0, n_obj, pb & ngt);
        jf_count(lc, 1, n_obj, gb & npr);
        jf_count(lc, 2, n_obj, pb);
        jf_count(lc, 3, n_obj, gb);
    }
    __syncthreads();
    if (threadIdx.x < 4 * 16 && lc[threadIdx.x] != 0) atomicAdd(&counts[3 * 16 + threadIdx.x], lc[threadIdx.x]);
}

// One wave: lane o scores object o (its seven counters are seven independent loads instead of a single thread's serial walk: 67 -> ~5 us);
// the per-object values are then added in the order 1 .. n_obj - 1, as the oracle adds them, and the counters are zeroed for the next frame.
__global__ __launch_bounds__(64) void jf_finalize_kernel(int32_t *__restrict__ counts, int n_obj, double *__restrict__ accum) {
    const int o = threadIdx.x;
    double j = 0.0, f = 0.0;
    if (o >= 1 && o < n_obj) {
        const double ap = counts[o], ag = counts[16 + o], in = counts[32 + o];
        const double np_ = counts[5 * 16 + o], ng = counts[6 * 16 + o], mp = counts[3 * 16 + o], mg = counts[4 * 16 + o];
        const double un = ap + ag - in;
        j = un == 0.0 ? 1.0 : in / un;                                     // db_eval_iou: both empty -> 1
        double prec, rec;
        if (np_ == 0.0 && ng > 0.0) { prec = 1.0; rec = 0.0; }
        else if (np_ > 0.0 && ng == 0.0) { prec = 0.0; rec = 1.0; }
        else if (np_ == 0.0 && ng == 0.0) { prec = 1.0; rec = 1.0; }
        else { prec = mp / np_; rec = mg / ng; }
        f = (prec + rec == 0.0) ? 0.0 : 2.0 * prec * rec / (prec + rec);
    }
    // serial order 1 .. n_obj - 1 (as the oracle adds them), carried by lane 0 through readlane-style shuffles
    double sj = 0.0, sf = 0.0;
    for (int k = 1; k < n_obj; ++k) {
        sj += __shfl(j, k);
        sf += __shfl(f, k);
    }
    __syncthreads();                                                        // every lane has read its counters
    if (o == 0) {
        accum[0] += sj;
        accum[1] += sf;
        accum[2] += (double)(n_obj - 1);
        accum[3] += 1.0;
    }
    for (int i = o; i < 7 * 16; i += 64) counts[i] = 0;                     // ready for the next frame
}

}  // namespace

extern "C" {

size_t aoc_mask_jf_workspace_bytes(int H, int W) { return H < 1 || W < 1 ? 0 : aoc_align_up((size_t)7 * 16 * sizeof(int32_t), 256) + (size_t)H * W * sizeof(uint32_t); }

int aoc_mask_jf_accumulate(const int32_t *pred, const int32_t *gt, int H, int W, int n_obj, int bound_pix, void *workspace, size_t workspace_bytes,
                           int workspace_is_clean, double *accum, aoc_stream_t stream) {
    if (!pred || !gt || !workspace || !accum || H < 1 || W < 1 || n_obj < 1 || bound_pix < 0) return AOC_ERR_INVALID_ARG;
    if (n_obj > 16) return AOC_ERR_UNSUPPORTED;
    if (workspace_bytes < aoc_mask_jf_workspace_bytes(H, W)) return AOC_ERR_WORKSPACE;
    hipStream_t st = aoc_hip_stream(stream);
    int32_t *counts = static_cast<int32_t *>(workspace);
    uint32_t *bits = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + aoc_align_up((size_t)7 * 16 * sizeof(int32_t), 256));
    if (!workspace_is_clean && hipMemsetAsync(counts, 0, 7 * 16 * sizeof(int32_t), st) != hipSuccess) return AOC_ERR_LAUNCH;
    const unsigned nb = (unsigned)(((size_t)H * W + 255) / 256);
    hipLaunchKernelGGL(jf_boundary_kernel, dim3(nb), dim3(256), 0, st, pred, gt, H, W, n_obj, bits, counts);
    if (bound_pix <= JF_R_LDS)
        hipLaunchKernelGGL(jf_match_tile_kernel, dim3((W + JF_TW - 1) / JF_TW, (H + JF_TH - 1) / JF_TH), dim3(256),
                           (size_t)(JF_TW + 2 * bound_pix) * (JF_TH + 2 * bound_pix) * sizeof(uint32_t), st, bits, H, W, n_obj, bound_pix, counts);
    else
        hipLaunchKernelGGL(jf_match_kernel, dim3(nb), dim3(256), 0, st, bits, H, W, n_obj, bound_pix, counts);
    hipLaunchKernelGGL(jf_finalize_kernel, dim3(1), dim3(64), 0, st, counts, n_obj, accum);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_confident_labels(const float *probs, int n_ch, int64_t n, uint32_t exist_bits, const int32_t *join_label, float unc_ratio,
                         int32_t *labels_out, int32_t *confident_out, float *entropy_out, aoc_stream_t stream) {
    if (!probs || !labels_out || n < 1 || n_ch < 1) return AOC_ERR_INVALID_ARG;
    if (n_ch > 32) return AOC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(confident_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), probs, n_ch, n, exist_bits,
                       join_label, unc_ratio, labels_out, confident_out, entropy_out);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

int aoc_label_onehot_nearest(const int32_t *label, int H, int W, int h, int w, int n_obj, float *onehot_hwc, aoc_stream_t stream) {
    if (!label || !onehot_hwc || H < 1 || W < 1 || h < 1 || w < 1 || n_obj < 1) return AOC_ERR_INVALID_ARG;
    hipLaunchKernelGGL(label_onehot_nearest_kernel, dim3((unsigned)((h * w + 255) / 256)), dim3(256), 0, aoc_hip_stream(stream), label, H, W, h, w, n_obj,
                       (float)H / (float)h, (float)W / (float)w, onehot_hwc);
    AOC_RETURN_IF_LAUNCH_FAILED();
    return AOC_OK;
}

}  // extern "C"
