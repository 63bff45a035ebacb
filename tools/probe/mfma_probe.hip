// Developer probe (GPU box): what limits a loop of v_mfma_f32_32x32x16_f16 fed from LDS with 2 waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probe/mfma_probe.hip ; run: /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bits: 1 = A fragments from LDS (else registers), 2 = max16 epilogue per 14 MFMAs, 4 = one accumulator chain instead of two,
// 8 = barrier every 28 MFMAs
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void probe(const uint4 *src, float *out, int iters) {
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = src[i];
    __syncthreads();
    f16x8 b[2][7];
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < 7; ++k) b[q][k] = __builtin_bit_cast(f16x8, src[(q * 7 + k) * 64 + lane]);
    f32x16 acc[2];
    for (int q = 0; q < 2; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float best = -1e30f;
    f16x8 areg[7];
    for (int k = 0; k < 7; ++k) areg[k] = __builtin_bit_cast(f16x8, src[512 + k * 64 + lane]);
    for (int it = 0; it < iters; ++it) {
        const uint4 *row = lds + ((it & 31) * 64 + lane % 32) * 2;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            f16x8 a = (MODE & 1) ? __builtin_bit_cast(f16x8, row[k * 128 + (lane >> 5)]) : areg[k];
            if (MODE & 4) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[0][k], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[1][k], acc[0], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[0][k], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[1][k], acc[1], 0, 0, 0);
            }
        }
        if (MODE & 2) {
            float m = acc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[1][r]);
            if (__builtin_amdgcn_ballot_w64(m > best) != 0ull) best = fmaxf(best, m);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        }
        if ((MODE & 8) && (it & 1)) __syncthreads();
    }
    float s = best;
    for (int q = 0; q < 2; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES>
void run(const char *name, const uint4 *src, float *out) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 65536, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * WAVES * iters * 14;
    const double cyc_per_mfma_per_simd = ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4));
    printf("%-44s waves/CU %2d  %.3f ms  %.1f TFLOP/s  %.1f cycles(@2.4GHz)/MFMA/SIMD\n", name, WAVES, ms, mfma * 32768 / ms / 1e9, cyc_per_mfma_per_simd);
}

int main() {
    uint4 *src;
    float *out;
    hipMalloc(&src, 4096 * 16);
    hipMalloc(&out, 256 * 1024 * 4);
    std::vector<unsigned short> h(4096 * 8);
    unsigned x = 12345u;
    for (auto &v : h) {                                    // random halves in +-[0.5, 2): realistic toggling (the clock follows the power draw)
        x = x * 1664525u + 1013904223u;
        v = (unsigned short)(((x >> 16) & 0x83ffu) | (((x >> 8) & 1u) ? 0x3800u : 0x3c00u));
    }
    hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    run<0, 8>("regs, 2 chains", src, out);
    run<4, 8>("regs, 1 chain", src, out);
    run<0, 4>("regs, 2 chains", src, out);
    run<1, 8>("LDS A, 2 chains", src, out);
    run<3, 8>("LDS A, 2 chains, max16 epilogue", src, out);
    run<11, 8>("LDS A, 2 chains, epilogue, barrier/28", src, out);
    run<1, 4>("LDS A, 2 chains", src, out);
    run<3, 4>("LDS A, 2 chains, max16 epilogue", src, out);
    return 0;
}
