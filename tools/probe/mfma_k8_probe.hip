// Developer probe (GPU box), VERDICT r4 item 4a: does the legacy v_mfma_f32_32x32x8_f16 (K = 8) issue in HALF the time of
// v_mfma_f32_32x32x16_f16 (K = 16) on gfx950?  If it does, the dense kernel's K = 112 (7 x K16) could become 104 = 6 x K16 + 1 x K8.
// One wave per SIMD, four independent accumulators, back-to-back issue; cycles from s_memtime around the loop.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_k8_probe tools/probe/mfma_k8_probe.hip ; run: /tmp/mfma_k8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256, 1) void probe(const float *src, float *out, long long *cycles, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a8, b8;
    f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)src[lane + i]; b8[i] = (_Float16)src[64 + lane + i]; }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (KIND == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[q], 0, 0, 0);
                else acc[q] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[q], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, const float *src, float *out, long long *cyc, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[1];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 16;
    printf("%-28s %4d workgroups (1 wave per SIMD): %.3f ms, %.1f ns per MFMA per SIMD, s_memtime ticks per MFMA %.2f (100 MHz clock: x ~21-24 for core cycles)\n", name, blocks, ms,
           ms * 1e6 / n, (double)h[0] / n);
}

int main() {
    float *src, *out;
    long long *cyc;
    hipMalloc(&src, 4096);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&cyc, 4096 * 8);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 17) * 0.01f;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    for (int blocks : {1, 256}) {
        run<0>("v_mfma_f32_32x32x16_f16", src, out, cyc, blocks);
        run<1>("v_mfma_f32_32x32x8_f16", src, out, cyc, blocks);
    }
    return 0;
}
