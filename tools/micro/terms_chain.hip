// Microbenchmark: cycles of the per-correction serial chain of a single filter on ONE wavefront (one wave per CU):
//   (a) scalar measurement_terms + innovation_cov + inv2 on lane 0 (what k_gain / k_correct_fused do)
//   (b) the lane-parallel wave_terms
//   (c) atan2, sqrt, a division, a dependent fma on their own
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../ekf_slam_ml_amd/csrc terms_chain.hip -o terms_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ekf_kernels.hpp"
using namespace ekf;

__global__ __launch_bounds__(64) void k_chain(double* out, long long* cyc, int iters, int mode) {
    __shared__ double S55s[5][6];
    __shared__ double oH[10], oSi[4], oNu[2];
    const int lane = threadIdx.x;
    if (lane < 25) S55s[lane / 5][lane % 5] = (lane / 5 == lane % 5) ? 0.5 + 0.01 * lane : 0.001 * lane;
    __syncthreads();
    double tx = 1.3, ty = -0.7, acc = 0.0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (mode == 0) {
            if (lane == 0) {
                MeasTerms m;
                measurement_terms(tx, ty, 1.1 + acc * 1e-9, -0.8, 0.1, 0.05, 0.02, m);
                double S55[5][5], S[2][2], Si[2][2];
                for (int k = 0; k < 5; k++) for (int l = 0; l < 5; l++) S55[k][l] = S55s[k][l];
                innovation_cov(S55, m.H, 0.01, S);
                inv2(S, Si);
                acc += Si[0][0] + m.H[1][4] + (m.z0 - m.zh0) + normalize_angle(m.z1 - m.zh1);
                tx += acc * 1e-12;
            }
        } else if (mode == 1) {
            auto s55 = [&](int k, int l) { return S55s[k][l]; };
            wave_terms(lane, tx, ty, 1.1 + acc * 1e-9, -0.8, 0.1, 0.05, 0.02, 0.01, s55, true, oH, oSi, oNu);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            acc += oSi[0] + oH[9] + oNu[0] + oNu[1];
            tx += acc * 1e-12;
        } else if (mode == 2) { acc += atan2(ty + acc * 1e-9, tx);
        } else if (mode == 3) { acc += sqrt(tx + acc * 1e-9);
        } else if (mode == 4) { acc = (ty + acc) / tx;
        } else if (mode == 5) { acc = __builtin_fma(acc, 1.0000001, 0.5);
        } else if (mode == 6) { acc += normalize_angle(acc * 1e-9 + 0.3);
        } else if (mode == 7) { acc += oH[lane & 7] * 1e-9; oH[lane & 7] = acc; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = acc + tx;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* out; long long* cyc; (void)hipMalloc(&out, 8 * 64 * 256); (void)hipMalloc(&cyc, 8 * 256);
    const char* names[] = {"scalar terms on lane 0", "lane-parallel wave_terms", "atan2", "sqrt", "division", "dependent v_fma_f64", "normalize_angle", "LDS write->read round trip"};
    for (int mode = 0; mode < 8; mode++) {
        const int iters = 2000;
        long long h[256];
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, cyc, iters, mode);
            (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%-28s %8.1f cycles per iteration (s_memtime)\n", names[mode], (double)h[0] / iters);
    }
    return 0;
}
