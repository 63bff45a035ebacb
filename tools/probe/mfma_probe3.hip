// Developer probe (GPU box), round 6: ONE wave per SIMD, four stationary query tiles (4 accumulator chains), A fragments from LDS one tile ahead,
// and the decision of tile t - 1 (max over its 4 x 16 accumulators, bound test, ballot) program-ordered between the MFMAs of tile t (two accumulator sets).
// MODE 0: no decision; 1: decision after the tile's MFMAs (unpipelined); 2: decision of the previous tile interleaved by the compiler's own scheduling;
// 3: interleaved with sched_group_barrier (1 MFMA : 3 VALU).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe3 tools/probe/mfma_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float max16(const f32x16 &a) {
    float m0 = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]);
    float m1 = __builtin_fmaxf(__builtin_fmaxf(a[3], a[4]), a[5]);
    float m2 = __builtin_fmaxf(__builtin_fmaxf(a[6], a[7]), a[8]);
    float m3 = __builtin_fmaxf(__builtin_fmaxf(a[9], a[10]), a[11]);
    float m4 = __builtin_fmaxf(__builtin_fmaxf(a[12], a[13]), a[14]);
    m0 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
    m3 = __builtin_fmaxf(__builtin_fmaxf(m3, m4), a[15]);
    return __builtin_fmaxf(m0, m3);
}

template <int MODE, int NQ>
__global__ __launch_bounds__(256, 1) void probe(const uint4 *src, float *out, int iters) {
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    f16x8 b[NQ][7];
    for (int q = 0; q < NQ; ++q)
        for (int k = 0; k < 7; ++k) b[q][k] = __builtin_bit_cast(f16x8, src[((q * 7 + k) * 64 + lane) & 4095]);
    float best[NQ], eps[NQ];
    for (int q = 0; q < NQ; ++q) { best[q] = 1e30f; eps[q] = 1.0f + q; }
    unsigned hits = 0;
    f32x16 accA[NQ], accB[NQ];
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) { accA[q][r] = 0.f; accB[q][r] = 0.f; }
    f16x8 afA[7], afB[7];
    auto load = [&](f16x8 *af, int it) {
        const uint4 *row = lds + ((it & 31) * 64 + lane % 32) * 2;
#pragma unroll
        for (int k = 0; k < 7; ++k) af[k] = __builtin_bit_cast(f16x8, row[k * 128 + (lane >> 5)]);
    };
    auto mfmas = [&](f32x16 *acc, const f16x8 *af) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                acc[q] = k == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(af[k], b[q][k], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_f16(af[k], b[q][k], acc[q], 0, 0, 0);
    };
    auto decide = [&](const f32x16 *acc) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float m = max16(acc[q]);
            if (__builtin_amdgcn_ballot_w64(m + eps[q] >= best[q]) != 0ull) hits += 1;      // (never true with this data: no rescoring in the probe)
        }
    };
    load(afA, 0);
    for (int it = 0; it < iters; it += 2) {
        // tile `it`: fragments afA -> accA; prefetch afB; decide accB (tile it - 1)
        load(afB, it + 1);
        mfmas(accA, afA);
        if (MODE == 1) decide(accA);
        if (MODE >= 2) decide(accB);
        if (MODE == 3) {
#pragma unroll
            for (int g = 0; g < 7 * NQ; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three VALU
            }
        }
        load(afA, it + 2);
        mfmas(accB, afB);
        if (MODE == 1) decide(accB);
        if (MODE >= 2) decide(accA);
        if (MODE == 3) {
#pragma unroll
            for (int g = 0; g < 7 * NQ; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        }
    }
    float s = (float)hits;
    for (int q = 0; q < NQ; ++q)
        for (int r = 0; r < 16; ++r) s += accA[q][r] + accB[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NQ>
void run(const char *name, const uint4 *src, float *out, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE, NQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, NQ>), dim3(blocks), dim3(256), 65536, 0, src, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double per_simd = (double)iters * 7 * NQ;
    printf("%-58s NQ %d workgroups %3d : %7.3f ms  %6.2f ns per MFMA per SIMD  %7.1f TFLOP/s\n", name, NQ, blocks, best, best * 1e6 / per_simd,
           (double)blocks * 4 * per_simd * 32768 / best / 1e9);
}

int main() {
    uint4 *src;
    float *out;
    (void)hipMalloc(&src, 4096 * 16);
    (void)hipMalloc(&out, 256 * 1024 * 4);
    std::vector<unsigned short> h(4096 * 8);
    unsigned x = 12345u;
    for (auto &v : h) {
        x = x * 1664525u + 1013904223u;
        v = (unsigned short)(((x >> 16) & 0x83ffu) | (((x >> 8) & 1u) ? 0x3800u : 0x3c00u));
    }
    (void)hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    for (int blocks : {64, 256}) {
        run<0, 4>("1 wave/SIMD, LDS A one tile ahead, no decision", src, out, blocks);
        run<1, 4>("... decision after the tile (unpipelined)", src, out, blocks);
        run<2, 4>("... decision of the previous tile, compiler's order", src, out, blocks);
        run<3, 4>("... decision of the previous tile, sched_group_barrier", src, out, blocks);
        run<0, 2>("NQ = 2: no decision", src, out, blocks);
        run<3, 2>("NQ = 2: pipelined decision", src, out, blocks);
    }
    return 0;
}
