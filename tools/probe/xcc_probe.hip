// Developer probe (GPU box): which XCDs do the workgroups of a stream with a HIP CU mask run on?
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/xcc_probe tools/probe/xcc_probe.hip ; run: /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(int *xcc, int spin) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(id & 0xf);
    for (int i = 0; i < spin; ++i) asm volatile("s_sleep 10");
}
static void run(const char *name, hipStream_t st) {
    const int n = 4096;
    int *d;
    hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(who, dim3(n), dim3(256), 0, st, d, 200);
    hipStreamSynchronize(st);
    std::vector<int> h(n);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int v : h) cnt[v & 15]++;
    printf("%-28s workgroups per XCC:", name);
    for (int i = 0; i < 8; ++i) printf(" %4d", cnt[i]);
    printf("   first 16 block -> xcc:");
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}
static hipStream_t masked(int lo, int hi, int step = 1) {
    unsigned mask[8] = {0};
    for (int cu = lo; cu < hi; cu += step) mask[cu / 32] |= 1u << (cu % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask failed\n"); return nullptr; }
    return s;
}
int main() {
    hipStream_t s0;
    hipStreamCreate(&s0);
    run("all CUs", s0);
    run("CUs 0..223", masked(0, 224));
    run("CUs 224..255", masked(224, 256));
    run("CUs 0..31", masked(0, 32));
    run("CUs 0..127", masked(0, 128));
    run("every 8th CU", masked(0, 256, 8));
    run("CUs 0..7", masked(0, 8));
    return 0;
}
