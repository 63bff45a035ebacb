// Developer probe (GPU box): what does rocprofv3's FETCH_SIZE report for the access patterns of the k-means sum kernels?  The micro-architecture
// guide calibrates the counter for wide coalesced streams only (a 128-byte request is tallied as 64 bytes: x2); the heads / fold kernels gather
// 112- and 80-byte pieces of 400-byte rows through member lists.  Every kernel here reads a KNOWN set of bytes of a 400 MB buffer of 400-byte rows
// (larger than the 256 MB Infinity Cache); the host prints the requested bytes and the bytes of the distinct 64-byte sectors / 128-byte lines
// touched, the counter comes from `rocprofv3 --kernel-trace --pmc FETCH_SIZE` around this binary (tools/profile_r05o.sh joins the two).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe tools/probe/fetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <random>
#include <set>
#include <vector>

constexpr int ROW_BYTES = 400;

__global__ void fill_kernel(float4 *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// wide coalesced stream: 16 bytes per lane, consecutive lanes consecutive addresses
__global__ void wide_kernel(const float4 *__restrict__ p, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc;
}

// pieces of rows through a member list: piece `g` (LANES x 16 bytes at byte offset g * LANES * 16) of every listed row, LANES lanes per row;
// n_groups > 1: blockIdx.x % n_groups... no -- the groups of one row block have workgroup ids that differ by 8 (they land on one XCD), like the
// k-means sum kernels' XCD-aware ids
template <int LANES>
__global__ void piece_kernel(const char *__restrict__ base, const int *__restrict__ rows, int n_rows, int g_first, int n_groups, int row_bytes_used, float *sink) {
    constexpr int RPB = 256 / LANES;                         // rows per pass of a 256-thread block
    const int per_block = RPB * 16;                          // rows per block
    int b = blockIdx.x;
    // blocks of 8 row blocks x n_groups: id = (blk8 * n_groups + grp) * 8 + (row block % 8)
    int grp = (b / 8) % n_groups, rb = (b / (8 * n_groups)) * 8 + (b % 8);
    int g = g_first + grp;
    int lane = threadIdx.x % LANES, sub = threadIdx.x / LANES;
    float acc = 0.f;
    if (sub < RPB) {
        for (int it = 0; it < 16; ++it) {
            int r = rb * per_block + it * RPB + sub;
            if (r < n_rows) {
                int off = g * LANES * 16 + lane * 16;
                if (off + 16 <= row_bytes_used) {
                    float4 v = *reinterpret_cast<const float4 *>(base + (size_t)rows[r] * ROW_BYTES + off);
                    acc += v.x + v.y + v.z + v.w;
                }
            }
        }
    }
    if (acc == 123.456f) *sink = acc;
}

static void expect(const char *name, const std::vector<int> &rows, int piece_bytes, int g_first, int n_groups) {
    std::set<uint64_t> s64, s128;
    uint64_t req = 0;
    for (int r : rows)
        for (int g = g_first; g < g_first + n_groups; ++g) {
            int lo = g * piece_bytes, hi = lo + piece_bytes;
            if (hi > ROW_BYTES) hi = ROW_BYTES;
            if (lo >= hi) continue;
            req += hi - lo;
            uint64_t a = (uint64_t)r * ROW_BYTES + lo, e = (uint64_t)r * ROW_BYTES + hi - 1;
            for (uint64_t x = a / 64; x <= e / 64; ++x) s64.insert(x);
            for (uint64_t x = a / 128; x <= e / 128; ++x) s128.insert(x);
        }
    printf("EXPECT %-34s requested_MB=%.3f sectors64_MB=%.3f lines128_MB=%.3f\n", name, req / 1e6, s64.size() * 64 / 1e6, s128.size() * 128 / 1e6);
}

int main() {
    const int n_rows_buf = 1 << 20;                          // 1 Mi rows x 400 B = 419 MB
    const size_t bytes = (size_t)n_rows_buf * ROW_BYTES;
    char *buf;
    float *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 4);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<float4 *>(buf), bytes / 16);
    hipDeviceSynchronize();
    std::mt19937 gen(7);
    // member lists: ascending row ids (a cluster's members keep the row order), every row / every ~4th row / every ~16th row at random
    std::vector<std::vector<int>> lists(3);
    const int density[3] = {1, 4, 16};
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < n_rows_buf; ++r)
            if (density[k] == 1 || (int)(gen() % density[k]) == 0) lists[k].push_back(r);
    int *d_rows[3];
    for (int k = 0; k < 3; ++k) {
        hipMalloc(&d_rows[k], lists[k].size() * sizeof(int));
        hipMemcpy(d_rows[k], lists[k].data(), lists[k].size() * sizeof(int), hipMemcpyHostToDevice);
    }
    printf("EXPECT %-34s requested_MB=%.3f sectors64_MB=%.3f lines128_MB=%.3f\n", "wide_kernel", bytes / 1e6, bytes / 1e6, bytes / 1e6);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(wide_kernel, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const float4 *>(buf), bytes / 16, sink);
    hipDeviceSynchronize();
    char name[96];
    for (int k = 0; k < 3; ++k) {
        const int n = (int)lists[k].size();
        auto launch = [&](auto kern, int lanes, int g_first, int n_groups, const char *tag) {
            int rpb = (256 / lanes) * 16, nrb = (n + rpb - 1) / rpb, nrb8 = (nrb + 7) / 8 * 8;
            snprintf(name, sizeof name, "%s 1/%d rows g=%d..%d", tag, density[k], g_first, g_first + n_groups - 1);
            expect(name, lists[k], lanes * 16, g_first, n_groups);
            for (int rep = 0; rep < 3; ++rep)
                hipLaunchKernelGGL(kern, dim3(nrb8 * n_groups), dim3(256), 0, 0, buf, d_rows[k], n, g_first, n_groups, ROW_BYTES, sink);
            hipDeviceSynchronize();
        };
        // order of the launches = order of the EXPECT lines (three dispatches each)
        launch(piece_kernel<25>, 25, 0, 1, "piece400");       // whole rows (km_rownorm's pattern)
        launch(piece_kernel<7>, 7, 0, 1, "piece112");         // one 112-byte feature group
        launch(piece_kernel<7>, 7, 1, 1, "piece112");
        launch(piece_kernel<7>, 7, 0, 4, "piece112");         // the four groups (112 + 112 + 112 + 64 bytes) side by side, ids 8 apart
        launch(piece_kernel<5>, 5, 0, 1, "piece80");          // one 80-byte feature group (the fold)
        launch(piece_kernel<5>, 5, 0, 5, "piece80");          // the five groups side by side
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
