// Probe of the v_mfma_f64_16x16x4_f64 operand / result layout: A(i,k) = 100 i + k, B(k,j) = identity-like selectors.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double* out) {
    const int l = threadIdx.x, lj = l & 15, lk = l >> 4;
    // assume A(i,k): lane i + 16k ; B(k,j): lane j + 16k.  A = 1000 + 10 i + k ; B(k,j) = (j == 3 && k == 2) ? 1 : 0
    const double a = 1000.0 + 10.0 * lj + lk;
    const double b = (lj == 3 && lk == 2) ? 1.0 : 0.0;
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; v++) out[l * 4 + v] = c[v];
}
int main() {
    double* d; hipMalloc(&d, 8 * 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D(i, 3) = A(i, 2) = 1000 + 10 i + 2, everything else 0
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) if (h[l * 4 + v] != 0.0) printf("lane %2d reg %d = %.0f\n", l, v, h[l * 4 + v]);
    return 0;
}
