// Developer probe (GPU box): what does a hipGraph buy for a chain of ~100 small dependent kernels (the k-means chain: 20 Lloyd iterations x 5
// launches) -- host time per chain and GPU time per chain (launch gaps), against plain hipLaunchKernel on a stream?
// The kernels take ONE pointer to a device-side argument block (the chain's real arguments would sit there, rewritten by a small memcpy in
// front of every graph launch), so the instantiated graph never needs hipGraphExecKernelNodeSetParams.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_probe tools/probe/graph_probe.hip ; run: /tmp/graph_probe [kernels per chain=105] [spin cycles=2000]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Args {
    float *buf;
    int n;
    int spin;
};

__global__ __launch_bounds__(256) void step_kernel(const Args *a, int which) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const long long t0 = clock64();
    while (clock64() - t0 < a->spin) {}
    if (i < a->n) a->buf[i] += (float)which;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));               \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 105;
    const int spin = argc > 2 ? atoi(argv[2]) : 2000;
    const int grid = 512, chains = 50;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *buf;
    Args *dargs;
    CK(hipMalloc(&buf, grid * 256 * sizeof(float)));
    CK(hipMemset(buf, 0, grid * 256 * sizeof(float)));
    CK(hipMalloc(&dargs, sizeof(Args)));
    Args h{buf, grid * 256, spin};
    CK(hipMemcpy(dargs, &h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto launch_chain = [&]() {
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), 0, st, dargs, k);
    };
    // ---- plain stream launches
    for (int w = 0; w < 5; ++w) launch_chain();
    CK(hipStreamSynchronize(st));
    double host = 0;
    CK(hipEventRecord(e0, st));
    for (int c = 0; c < chains; ++c) {
        const double t0 = now_ms();
        launch_chain();
        host += now_ms() - t0;
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float gpu = 0;
    CK(hipEventElapsedTime(&gpu, e0, e1));
    printf("stream launches : %3d kernels per chain, host %.3f ms per chain (%.2f us per launch), GPU %.3f ms per chain (%.2f us per kernel)\n", K, host / chains,
           host / chains / K * 1e3, gpu / chains, gpu / chains / K * 1e3);

    // ---- one graph per chain, captured once
    hipGraph_t graph;
    hipGraphExec_t exec;
    double t0 = now_ms();
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    launch_chain();
    CK(hipStreamEndCapture(st, &graph));
    const double t_cap = now_ms() - t0;
    t0 = now_ms();
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const double t_inst = now_ms() - t0;
    for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    host = 0;
    CK(hipEventRecord(e0, st));
    for (int c = 0; c < chains; ++c) {
        const double t1 = now_ms();
        CK(hipMemcpyAsync(dargs, &h, sizeof(h), hipMemcpyHostToDevice, st));      // the argument block of this chain
        CK(hipGraphLaunch(exec, st));
        host += now_ms() - t1;
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&gpu, e0, e1));
    printf("graph launches  : capture %.3f ms, instantiate %.3f ms; host %.3f ms per chain (incl. the argument memcpy), GPU %.3f ms per chain (%.2f us per kernel)\n",
           t_cap, t_inst, host / chains, gpu / chains, gpu / chains / K * 1e3);

    // ---- capture + instantiate + launch EVERY chain (what a graph costs when the arguments cannot be kept in a device block)
    host = 0;
    for (int c = 0; c < 10; ++c) {
        const double t1 = now_ms();
        hipGraph_t g2;
        hipGraphExec_t x2;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch_chain();
        CK(hipStreamEndCapture(st, &g2));
        CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(x2, st));
        host += now_ms() - t1;
        CK(hipStreamSynchronize(st));
        CK(hipGraphExecDestroy(x2));
        CK(hipGraphDestroy(g2));
    }
    printf("capture per chain: host %.3f ms per chain\n", host / 10);
    return 0;
}
