// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate (ceiling of an MFMA-tiled rank-2k flush).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        a += 1e-12;
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(double* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb = 1; wpb <= 4; wpb *= 2) {
        const int blocks = 256 * wpb, iters = 20000;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("independent accumulators=%d waves/SIMD=%d: %.1f TFLOP/s fp64 MFMA (%.2f ms)\n", NACC, wpb,
                            2.0 * 16 * 16 * 4 * NACC * (double)iters * 4.0 * blocks / (ms * 1e-3) / 1e12, ms);
        }
    }
}
int main() {
    double* out; hipMalloc(&out, 8 * 256 * 1024);
    run<1>(out); run<2>(out); run<4>(out);
    return 0;
}
