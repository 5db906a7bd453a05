// Microbenchmark: do fp64 MFMA and fp64 VALU FMA co-execute on a CU?  Waves alternate roles; the aggregate rate is
// compared with each unit alone (mfma_f64_peak: 46 TF, fp64_peak: 72 TF).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters, int mode) {  // mode 0: all VALU, 1: all MFMA, 2: mixed by wave
    const int wave = threadIdx.x >> 6;
    const bool mfma = mode == 1 || (mode == 2 && (wave & 1));
    double s = 0;
    if (mfma) {
        double4_t acc[4];
        for (int i = 0; i < 4; i++) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
        double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        double a[16];
        for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 1e-3 + i;
        const double x = 1.0000001 + threadIdx.x * 1e-9, y = x * 1.0000001;
        for (int it = 0; it < iters; it++) {  // 32 FMAs per trip = 4096 flop per wave; one MFMA trip = 4 x 2048 = 8192
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = __builtin_fma(y, x, __builtin_fma(x, a[i], 0.5));
        }
        for (int i = 0; i < 16; i++) s += a[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double* out; (void)hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[3] = {"all VALU", "all MFMA", "mixed by wave"};
    for (int mode = 0; mode < 3; mode++)
        for (int wpb = 2; wpb <= 8; wpb *= 2) {
            const int blocks = 256 * wpb, iters = 20000;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            // flop per wave-trip: VALU 2 * 16 * 2 FMA * 2 flop * 64 lanes = 8192; MFMA 4 * 2048 = 8192
            printf("%-14s blocks/CU=%d: %.1f TFLOP/s fp64 (%.2f ms)\n", names[mode], wpb,
                   8192.0 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12, ms);
        }
    return 0;
}
