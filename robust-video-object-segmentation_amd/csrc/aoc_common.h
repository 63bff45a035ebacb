// Shared device helpers for libaoc_hip.so (gfx950 only; wave = 64 lanes).
// The whole library is compiled with -ffp-contract=off: nothing fuses unless the code says
// __builtin_fmaf, because the k-means path must reproduce scipy's rounding sequence exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/aoc_hip.h"

#define AOC_RETURN_IF_LAUNCH_FAILED()                       \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return AOC_ERR_LAUNCH; \
    } while (0)

static inline hipStream_t aoc_hip_stream(aoc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t aoc_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Proto-mask transform of AEM:393/602/808/1049: (sigmoid(d + bias) - 0.5) * 2.
__device__ __forceinline__ float aoc_proto_transform(float d, float bias) {
    float s = 1.0f / (1.0f + expf(-(d + bias)));
    return (s - 0.5f) * 2.0f;
}

// single v_min_f32 (fminf would add a canonicalising v_max per operand); operands are never NaN here
__device__ __forceinline__ float aoc_fmin_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int aoc_lane() { return threadIdx.x & 63; }

// Round to the nearest float16 and back: the `.half()` matching mode of the reference (AEM:593-595, 801-803, 1002-1005) computes norms,
// dot products and distances as float16 TENSORS, i.e. every tensor-level result is rounded to float16 (torch accumulates the sums in
// fp32 and rounds once).  f16 == false: identity.
__device__ __forceinline__ float aoc_h(float x) { return (float)(_Float16)x; }
template <bool F16>
__device__ __forceinline__ float aoc_hr(float x) { return F16 ? (float)(_Float16)x : x; }

// min / sum across the 16 lanes that share (lane >> 4)  (one MFMA 16x16 output row group).
__device__ __forceinline__ float aoc_min16(float v) {
    v = fminf(v, __shfl_xor(v, 1));
    v = fminf(v, __shfl_xor(v, 2));
    v = fminf(v, __shfl_xor(v, 4));
    v = fminf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float aoc_wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LDS image of a 16-row operand tile for v_mfma_f32_16x16x4_f32, "k-permuted":
// lane (j = lane & 15, kq = lane >> 4) consumes x[j][4t + kq], t = 0..T-1.  Row j keeps its four
// kq-streams contiguous ([kq][t], each stream padded to TP = roundup(T,4) floats) so the lane reads
// its stream with ds_read_b128.  Row stride = 4*TP + 4 floats (the +4 staggers rows by 16 B).
__host__ __device__ __forceinline__ int aoc_tile_tp(int C) { return ((C / 4) + 3) / 4 * 4; }
__host__ __device__ __forceinline__ int aoc_tile_row_stride(int C) { return 4 * aoc_tile_tp(C) + 4; }

// aoc_dense_match_min with a device-side gate: every kernel of the call returns at once when gate != NULL and
// *gate == 0 (the split-fp16 kernels of dense_split.hip own the call then).  Defined in correlation.hip.
int aoc_dense_match_min_gated(const float *query, int64_t m, int C, const float *pool, const int32_t *fg_rows, const int32_t *n_fg,
                              int64_t n_fg_capacity, const uint32_t *wrong_bits, const float *obj_bias, int n_obj, float *out,
                              int64_t out_pixel_stride, int64_t out_obj_stride, int transform, void *workspace, size_t workspace_bytes,
                              const int32_t *gate, aoc_stream_t stream, int float16 = 0);

// measurement probe of aoc_dense_match_set_probe (thread-local; defined in correlation.hip)
struct AocDenseProbe { hipEvent_t start, stop; };
AocDenseProbe aoc_take_dense_probe();

// Developer switches (timing experiments, alternative kernels, some of which produce WRONG results on purpose) only exist in the
// development build (`make DEV=1` -> libaoc_hip_dev.so, -DAOC_DEV).  The release library never reads the environment: a stray
// variable cannot change a result or a kernel choice.  tests/test_host_logic.py checks that no switch name is in the release binary.
#include <stdlib.h>
#ifdef AOC_DEV
#define AOC_DEV_ENV(name) getenv(name)
#define AOC_DEV_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define AOC_DEV_ENV(name) (static_cast<const char *>(nullptr))      /* the switch's name is not even in the binary */
#define AOC_DEV_ENV_INT(name, dflt) (dflt)
#endif

// CUs the launching stream may use (aoc_set_stream_cus; 0 = all CUs of the device).  Defined in dense_split.hip.
int aoc_stream_cus();
// CU budget of the call in progress on THIS thread (0 = the process-wide aoc_set_stream_cus value); returns the previous one
int aoc_stream_cus_scope(int n_cus);
