// Microbenchmark: the strip-form flush's memory walk, without its multiply-adds, over two layouts of the same bytes.
//   row-major   : Sigma[b][row][ld]              -- a strip workgroup reads 2 KB of every row, 16 KB apart (the shipped layout)
//   strip-major : Sigma[b][strip][row][256 cols] -- a strip workgroup reads one contiguous block of N x 2 KB
// One workgroup of 16 waves per (filter, strip), 8-row groups per wave, non-temporal loads, +1, non-temporal stores: what
// k_flush_strip does to memory (ekf_delayed.hip).  160 KB of dynamic LDS keeps it at one workgroup per CU like the flush.
// build: hipcc --offload-arch=gfx950 -O3 -o strip_walk strip_walk.hip      run: ./strip_walk [B=4096] [N=2003]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int kWaves = 16;

// map: 0 = a filter's workgroups on one XCD (the flush's decode); 1 = plain order (a filter's strips dealt round the XCDs);
// rows_per_block / row_blocks: the strip cut into row blocks (workgroup = one row block of one strip); order: 0 = strips
// fastest (row-major dispatch), 1 = row blocks fastest
template <bool STRIP_MAJOR>
__global__ __launch_bounds__(64 * kWaves, 1) void k_walk(double* __restrict__ sigma, int N, int ld, int strips, int B, int map,
                                                         int row_blocks, int rows_per_block, int order) {
    extern __shared__ double2_t lds[];
    const int P = strips * row_blocks;
    int b, pp;
    const int full = (B / 8) * 8 * P;
    if (map == 1) {
        b = blockIdx.x / P;
        pp = blockIdx.x % P;
    } else if ((int)blockIdx.x < full) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / P) * 8 + xcd;
        pp = slot % P;
    } else {
        const int rest = blockIdx.x - full;
        b = (B / 8) * 8 + rest / P;
        pp = rest % P;
    }
    const int p = order ? pp / row_blocks : pp % strips;
    const int rb = order ? pp % row_blocks : pp / strips;
    const int row_begin = rb * rows_per_block, row_end = row_begin + rows_per_block < N ? row_begin + rows_per_block : N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) lds[0] = double2_t{0.0, 0.0};
    const size_t filter = (size_t)b * N * ld;
    // double2 units
    const size_t rs = STRIP_MAJOR ? 128 : (size_t)(ld >> 1);
    double2_t* col = reinterpret_cast<double2_t*>(sigma + filter) + (STRIP_MAJOR ? (size_t)p * N * 128 : (size_t)p * 128) + lane;
    const int ngroups = (row_end - row_begin + 7) >> 3;
    for (int g = wave; g < ngroups; g += kWaves) {
        const int r = row_begin + 8 * g;
        double2_t a[8][2];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t row = r + u < row_end ? r + u : row_end - 1;
            a[u][0] = __builtin_nontemporal_load(col + row * rs);
            a[u][1] = __builtin_nontemporal_load(col + row * rs + 64);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { a[u][0] += 1.0; a[u][1] += 1.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (r + u >= row_end) break;
            __builtin_nontemporal_store(a[u][0], col + (size_t)(r + u) * rs);
            __builtin_nontemporal_store(a[u][1], col + (size_t)(r + u) * rs + 64);
        }
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 2003;
    const int ld = (N + 255) / 256 * 256, strips = ld / 256;
    const size_t bytes = (size_t)B * N * ld * sizeof(double);
    double* s = nullptr;
    if (hipMalloc(&s, bytes) != hipSuccess) { printf("hipMalloc of %.1f GB failed\n", bytes / 1e9); return 1; }
    hipMemset(s, 0, bytes);
    const int lds = 160 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_walk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_walk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Cfg { int map, row_blocks, order; const char* name; } cfgs[] = {
        {0, 1, 0, "whole strips, filter on one XCD"}, {1, 1, 0, "whole strips, strips round the XCDs"},
        {0, 16, 0, "16 row blocks, strips fastest "}, {0, 16, 1, "16 row blocks, blocks fastest "},
        {1, 16, 0, "16 row blocks, strips fastest, plain order"}, {0, 4, 0, "4 row blocks, strips fastest  "}};
    for (const Cfg& c : cfgs) {
        const int rows = ((N + c.row_blocks - 1) / c.row_blocks + 7) & ~7, rbl = (N + rows - 1) / rows;
        const dim3 grid((unsigned)((long long)B * strips * rbl));
        for (int rep = 0; rep < 3; rep++)
            for (int mode = 0; mode < 2; mode++) {
                hipEventRecord(e0);
                if (mode) hipLaunchKernelGGL(k_walk<true>, grid, dim3(64 * kWaves), lds, 0, s, N, ld, strips, B, c.map, rbl, rows, c.order);
                else hipLaunchKernelGGL(k_walk<false>, grid, dim3(64 * kWaves), lds, 0, s, N, ld, strips, B, c.map, rbl, rows, c.order);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) printf("%-44s %s: %.2f ms, %.3f TB/s\n", c.name, mode ? "strip-major" : "row-major  ", ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
            }
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    hipFree(s);
    return 0;
}
