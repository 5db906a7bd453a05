// Microbenchmark: what a symmetric (mirrored) rewrite of Sigma would stream at.  Per filter the tiles on and above the
// diagonal are read once, scaled, and written twice (in place and mirrored); the traffic is 12 N^2 bytes instead of the
// 16 N^2 of a full read-modify-write.  Tile = TR rows x TC columns, the column groups of a row tile start at its diagonal
// square.  Wave = 8 x 8 lanes of 16 B, so a load / store instruction covers 8 segments of 128 B both in place and mirrored.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int QR, int QC, bool MIRROR, bool NT>   // tile = 32 QR x 32 QC, thread holds QR x 2 rows by QC double2
__global__ __launch_bounds__(256) void k(double* __restrict__ sigma, int N, int ld, size_t stride, int P, int B) {
    constexpr int TR = 32 * QR, TC = 32 * QC;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / P) * 8 + xcd;
    int p = slot % P;
    if (b >= B) return;
    int ti = 0, g;
    for (;; ti++) {
        const int first = MIRROR ? ti * TR : 0;
        const int ng = (ld - first + TC - 1) / TC;
        if (p < ng) { g = p; break; }
        p -= ng;
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int tx = (l & 7) + 8 * (w & 1), ty = (l >> 3) + 8 * (w >> 1);
    double* S = sigma + (size_t)b * stride;
    const int r0 = ti * TR + 2 * ty, c0 = (MIRROR ? ti * TR : 0) + g * TC + 2 * tx;
    double2_t c[QR][2][QC];
#pragma unroll
    for (int i = 0; i < QR; i++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < QC; j++) {
                const int r = r0 + 32 * i + a, cc = c0 + 32 * j;
                c[i][a][j] = double2_t{0.0, 0.0};
                if (r < N && cc < ld) {
                    const double2_t* ptr = reinterpret_cast<const double2_t*>(S + (size_t)r * ld + cc);
                    c[i][a][j] = NT ? __builtin_nontemporal_load(ptr) : *ptr;
                }
            }
#pragma unroll
    for (int i = 0; i < QR; i++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < QC; j++) {
                c[i][a][j] *= 0.9999;
                const int r = r0 + 32 * i + a, cc = c0 + 32 * j;
                if (r < N && cc < ld) {
                    double2_t* ptr = reinterpret_cast<double2_t*>(S + (size_t)r * ld + cc);
                    if (NT) __builtin_nontemporal_store(c[i][a][j], ptr); else *ptr = c[i][a][j];
                }
            }
    if (MIRROR) {
#pragma unroll
        for (int j = 0; j < QC; j++) {
            if (g == 0 && j < QR) continue;   // the diagonal square is written in place only
#pragma unroll
            for (int bb = 0; bb < 2; bb++)
#pragma unroll
                for (int i = 0; i < QR; i++) {
                    const int r = c0 + 32 * j + bb, cc = r0 + 32 * i;   // mirrored position
                    const double2_t v = {bb ? c[i][0][j].y : c[i][0][j].x, bb ? c[i][1][j].y : c[i][1][j].x};
                    if (r < N && cc < N) {
                        double2_t* ptr = reinterpret_cast<double2_t*>(S + (size_t)r * ld + cc);
                        if (NT) __builtin_nontemporal_store(v, ptr); else *ptr = v;
                    }
                }
        }
    }
}

template <int QR, int QC, bool MIRROR, bool NT>
static void run(double* sigma, int N, int ld, int B) {
    constexpr int TR = 32 * QR, TC = 32 * QC;
    int P = 0;
    for (int ti = 0; ti * TR < N; ti++) P += (ld - (MIRROR ? ti * TR : 0) + TC - 1) / TC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<QR, QC, MIRROR, NT>), dim3((B / 8) * 8 * P), dim3(256), 0, 0, sigma, N, ld, (size_t)N * ld, P, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (MIRROR ? 12.0 : 16.0) * N * (double)N * B;
    printf("%-16s %-12s tile %3d x %3d: %7.2f ms, %5.2f TB/s on %2d N^2 bytes\n", MIRROR ? "upper + mirror" : "full rmw",
           NT ? "nontemporal" : "cached", TR, TC, best, bytes / (best * 1e-3) / 1e12, MIRROR ? 12 : 16);
}

int main(int argc, char** argv) {
    const int N = 2003, ld = 2016, B = argc > 1 ? atoi(argv[1]) : 2048;
    double* sigma;
    if (hipMalloc(&sigma, (size_t)B * N * ld * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(sigma, 0, (size_t)B * N * ld * 8);
#define BOTH(QR, QC) run<QR, QC, false, true>(sigma, N, ld, B); run<QR, QC, true, true>(sigma, N, ld, B); run<QR, QC, true, false>(sigma, N, ld, B);
    BOTH(2, 2) BOTH(2, 4) BOTH(2, 8) BOTH(1, 8) BOTH(1, 16) BOTH(4, 4) BOTH(1, 4)
    hipFree(sigma);
    return 0;
}
