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

}  // namespace

extern "C" {

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
