// Microbenchmark: sustained v_fma_f64 rate on the device (vector operands vs one scalar operand).
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool SCALAR>
__global__ __launch_bounds__(256) void k(double* out, const double* __restrict__ kin, int iters) {
    double a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-3 + i;
    const double x = 1.0000001 + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
        const double s0 = SCALAR ? kin[it & 15] : x, s1 = SCALAR ? kin[(it + 1) & 15] : x * 1.0000001;
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __builtin_fma(s1, x, __builtin_fma(s0, a[i], 0.5));
    }
    double s = 0; for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double *out, *kin; hipMalloc(&out, 8 * 256 * 4096); hipMalloc(&kin, 8 * 16); hipMemset(kin, 0, 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int sc = 0; sc < 2; sc++) for (int wpb = 1; wpb <= 4; wpb *= 2) {
        const int blocks = 256 * wpb * 2, iters = 20000;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (sc) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, out, kin, iters);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, out, kin, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("scalar_operand=%d blocks/CU=%d: %.1f TFLOP/s fp64 (%.2f ms)\n", sc, wpb * 2,
                            2.0 * 2 * 16 * iters * 256.0 * blocks / (ms * 1e-3) / 1e12, ms);
        }
    }
    return 0;
}
