// Developer probe (GPU box), round 6: what paces a register-fed stream of v_mfma_f32_32x32x16_f16 on the whole chip?
// Axes: waves per SIMD (1 / 2), independent accumulator chains per wave (1 / 2 / 4), workgroups (64 = a quarter of the CUs / 256 = all),
// operand data (zeros / small integers / random halves as in the dense kernel's hi planes).  Same instruction stream in every case: the differences
// are power / issue effects, not code.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe2 tools/probe/mfma_probe2.hip ; run: /tmp/mfma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// PRIO: 0 = none; 1 = static: the second wave of every SIMD (waves 4..7) runs at s_setprio 1; 2 = bursts: every wave alternates a burst of 14 MFMAs
// with ~60 dependent VALU instructions (the dense kernel's shape), no priorities; 3 = bursts, s_setprio 1 around each burst; 4 = bursts + static
template <int WAVES, int CHAINS, int PRIO = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(const uint4 *src, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    if ((PRIO == 1 || PRIO == 4) && (threadIdx.x >> 8)) __builtin_amdgcn_s_setprio(1);
    f16x8 b[CHAINS][7], a[7];
    for (int q = 0; q < CHAINS; ++q)
        for (int k = 0; k < 7; ++k) b[q][k] = __builtin_bit_cast(f16x8, src[((q * 7 + k) * 64 + lane) & 4095]);
    for (int k = 0; k < 7; ++k) a[k] = __builtin_bit_cast(f16x8, src[(2048 + k * 64 + lane) & 4095]);
    f32x16 acc[CHAINS];
    for (int q = 0; q < CHAINS; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float v = (float)lane;
    for (int it = 0; it < iters; ++it) {
        if (PRIO == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int q = 0; q < CHAINS; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[q][k], acc[q], 0, 0, 0);
        if (PRIO == 3) __builtin_amdgcn_s_setprio(0);
        if (PRIO >= 2) {
#pragma unroll
            for (int u = 0; u < 60; ++u) v = __builtin_fmaf(v, 1.0001f, 0.5f);      // a dependent VALU chain: ~60 x 4+ cycles of this wave's time, no MFMA
        }
    }
    float s = v;
    for (int q = 0; q < CHAINS; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES, int CHAINS, int PRIO = 0>
void run(const char *data, const uint4 *src, float *out, int blocks) {
    const int iters = 8000 / CHAINS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<WAVES, CHAINS, PRIO>), dim3(blocks), dim3(WAVES * 64), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double per_simd = (double)(WAVES / 4) * iters * 7 * CHAINS;          // MFMAs one SIMD issues
    const double total = (double)blocks * WAVES * iters * 7 * CHAINS;
    printf("data %-8s  workgroups %3d  waves/SIMD %d  chains/wave %d  prio-mode %d : %7.3f ms  %6.2f ns per MFMA per SIMD  %7.1f TFLOP/s\n", data, blocks, WAVES / 4, CHAINS, PRIO, best,
           best * 1e6 / per_simd, total * 32768 / best / 1e9);
}

int main() {
    uint4 *src;
    float *out;
    hipMalloc(&src, 4096 * 16);
    hipMalloc(&out, 256 * 1024 * 4);
    std::vector<unsigned short> h(4096 * 8);
    for (int mode = 0; mode < 3; ++mode) {
        unsigned x = 12345u;
        for (auto &v : h) {
            x = x * 1664525u + 1013904223u;
            if (mode == 0) v = 0;
            else if (mode == 1) v = (unsigned short)(0x3c00u + ((x >> 20) & 0x3u) * 0x400u);                 // 1, 2, 4, 8: one significand pattern
            else v = (unsigned short)(((x >> 16) & 0x83ffu) | (((x >> 8) & 1u) ? 0x3800u : 0x3c00u));   // random halves in +-[0.5, 2)
        }
        hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
        const char *name = mode == 0 ? "zeros" : mode == 1 ? "pow2" : "random";
        for (int blocks : {64, 256}) {
            run<4, 1>(name, src, out, blocks);
            run<4, 2>(name, src, out, blocks);
            run<4, 4>(name, src, out, blocks);
            run<8, 1>(name, src, out, blocks);
            run<8, 2>(name, src, out, blocks);
            run<8, 2, 1>(name, src, out, blocks);
            run<4, 2, 2>(name, src, out, blocks);
            run<8, 2, 2>(name, src, out, blocks);
            run<8, 2, 3>(name, src, out, blocks);
            run<8, 2, 4>(name, src, out, blocks);
        }
    }
    return 0;
}
