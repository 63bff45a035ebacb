// Types shared by the two correlation translation units (correlation.hip: exact fp32; correlation_batched.hip: fp16-split).
#pragma once
#include "aoc_common.h"

constexpr int AOC_CORR_MAX_FRAMES = 32;   // frames per launch (kernel-argument table)
constexpr int AOC_CORR_MAX_TILES = 5;     // 32-row proxy tiles resident in LDS per launch (5 x 19.5 KB next to the waves' pixel-tile buffers)
constexpr int AOC_CORR_TABLE_TILES = 8;   // entries of a launch's tile table (the streaming records kernel keeps its proxy tiles in registers: up to 8)
constexpr int AOC_CORR_MAX_OUT = 64;      // output columns (sets) per launch

struct AocCorrFrame {
    const float *query, *proxies, *sqnorm, *bias;
    float *out;
};
struct AocCorrFrames {
    AocCorrFrame f[AOC_CORR_MAX_FRAMES];
    int32_t n;
};
// One 32-row tile of the proxy image.  kind 0 (grouped): row group g (rows 8g .. 8g+7) holds cnt[g] proxies starting at begin[g] of the
// set with output column oc[g] (-1: unused); gs = row groups per set (1, 2 or 4); first / last: position in a multi-tile set (gs = 4).
// kind 1 (column-wise): cnt[0] single-proxy sets starting at proxy begin[0], output columns oc[0] .. oc[0] + cnt[0] - 1, whose output
// planes are `step` elements apart.
struct AocCorrTile {
    int32_t begin[4];
    int16_t cnt[4];
    int16_t oc[4];
    int32_t kind, gs, first, last;
    int64_t step;
};
struct AocCorrTiles {
    AocCorrTile t[AOC_CORR_TABLE_TILES];
    int64_t oc_offset[AOC_CORR_MAX_OUT];   // element offset of each output column's plane in a frame's `out`
    int32_t oc_bias[AOC_CORR_MAX_OUT];     // index into the frame's set_bias
    int16_t oc_row0[AOC_CORR_MAX_OUT];     // image rows [row0, row0 + rows) belong to the column's set
    int16_t oc_rows[AOC_CORR_MAX_OUT];
    int32_t n, n_out;
};

// exact-fp32 batched correlation (correlation.hip).  gate != NULL: every kernel returns at once unless *gate == gate_value.
int aoc_corr_fp32_batched(const aoc_corr_frame *frames_host, int n_frames, int64_t m, int C, int n_proxy, int n_set,
                          const int32_t *set_begin_host, const int32_t *set_size_host, const int64_t *set_out_offset_host,
                          int64_t out_pixel_stride, int transform, const int32_t *gate, aoc_stream_t stream, int float16 = 0, int32_t gate_value = 1);
