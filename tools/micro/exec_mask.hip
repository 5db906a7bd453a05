// Microbenchmark: does a wavefront's fp64 VALU instruction cost less when only lane 0 is active?
// (single-lane "serial" sections of the LDS-resident kernels: ekf_small.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(double* out, int iters, int active_lanes) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001, c = 0.5;
    if ((int)threadIdx.x < active_lanes) {
        for (int it = 0; it < iters; it++) {  // a dependent chain, like the trigonometry of one lane
#pragma unroll
            for (int i = 0; i < 16; i++) a = __builtin_fma(a, b, c);
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
int main() {
    double* out; (void)hipMalloc(&out, 8 * 64 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lanes : {64, 32, 16, 1}) {
        const int iters = 200000;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, out, iters, lanes);  // one wave per CU
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("active lanes %2d: %.2f ns per dependent v_fma_f64 (one wave per CU)\n", lanes, ms * 1e6 / (iters * 16.0));
    }
    return 0;
}
