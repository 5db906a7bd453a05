// ekf_dense.hip -- dense general-F covariance propagation Sigma <- F * Sigma * F^T + Q in fp32 on the
// gfx950 matrix cores (BASELINE.json configs[3]; SURVEY.md section 8(d) "Dense config 4").
//
// This is the reference's expression `sigma = At*sigma*At.t() + Q` (rigid2d/src/ekf_slam.cpp:101-102)
// executed the way Armadillo executes it -- two dense N x N x N products -- for an ARBITRARY dense At.
// (The reference's own At = I + A has two off-diagonal non-zeros and is served by the O(N) k_predict
// kernel; the dense path exists for motion models whose Jacobian is a genuine dense matrix, and is the
// only place on this path where MFMA applies: 4 N^3 flop over 3*4*N^2 bytes, AI ~ 3.3 k flop/B at
// N = 10003.)
//
//   T      = F * Sigma          "NN": B operand row-major [K][N]
//   Sigma' = T * F^T + Q        "NT": B operand supplied as F[N][K] (k contiguous)
//
// Kernel: 128 x 128 block tile, BK = 32, 4 waves each owning a 64 x 64 sub-tile = 2 x 2
// v_mfma_f32_32x32x2_f32 accumulators (exact f32 FMA chains, 64 FLOP/clk/SIMD = the f32 peak).
// Operands go global -> registers -> LDS (the next K tile's global loads fly under the MFMAs); LDS images are
// [k][i] with an odd row stride so that both the transposing b32 stores and the fragment reads
// (lanes 0-31 = 32 consecutive i at one k, lanes 32-63 the next k) are bank-conflict-free.
// Matrices are ld x ld with ld a multiple of 128 and zero padding, so no tile is ragged.
#include <hip/hip_runtime.h>

#include "ekf_dense.hpp"

namespace ekf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int SA = BM + 1;   // odd stride: conflict-free transposing stores and fragment reads
constexpr int SBT = BN + 1;  // same for a B operand that arrives k-contiguous
constexpr int SBN = BN;      // row-major B is copied row by row with 16-B stores

// Tile id remap: consecutive ids go to different XCDs (round-robin dispatch), so give every XCD a
// contiguous chunk of the tile list, and walk the tiles in groups of GROUP_M block rows so the A panel
// and the B panel of neighbouring tiles are re-used out of that XCD's L2 (speed only).
__device__ __forceinline__ void tile_of(int bid, int tiles, int& tm, int& tn) {
    const int total = tiles * tiles;
    const int nx = 8;
    int id = bid;
    if (total % nx == 0) id = (bid % nx) * (total / nx) + bid / nx;
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles;
    const int g = id / per_group;
    const int first_m = g * GROUP_M;
    const int gm = min(GROUP_M, tiles - first_m);
    const int in_g = id % per_group;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}

// NBUF = 2: double-buffered LDS (66 KB, 2 workgroups per CU, one barrier per K tile);
// NBUF = 1: single buffer (33 KB, 4 workgroups per CU, two barriers per K tile).
template <bool BT, int NBUF>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 4) void k_gemm_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, const float* __restrict__ Qadd, int ld,
                                                     int tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_ELEMS = BK * SA;
    constexpr int B_ELEMS = BT ? BK * SBT : BK * SBN;
    constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;  // buffer b: A image at b*BUF_ELEMS, B image right behind it

    int tm, tn;
    tile_of(blockIdx.x, tiles, tm, tn);
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;

    const float* Ag = A + (size_t)tm * BM * ld;
    const float* Bg = BT ? B + (size_t)tn * BN * ld : B + (size_t)tn * BN;

    f32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; p++) {  // A tile: 128 rows x 32 k, 8 lanes cover one 128-B row segment
            const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
            ra[p] = *reinterpret_cast<const f32x4*>(Ag + (size_t)row * ld + k0 + k4);
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
                rb[p] = *reinterpret_cast<const f32x4*>(Bg + (size_t)row * ld + k0 + k4);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) {  // B tile: 32 k x 128 j, 32 lanes cover one 512-B row segment
                const int k = p * 8 + (t >> 5), j4 = (t & 31) * 4;
                rb[p] = *reinterpret_cast<const f32x4*>(Bg + (size_t)(k0 + k) * ld + j4);
            }
        }
    };
    auto lstore = [&](int buf) {
        float* as = smem + buf * BUF_ELEMS;
        float* bs = as + A_ELEMS;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) as[(k4 + j) * SA + row] = ra[p][j];
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) bs[(k4 + j) * SBT + row] = rb[p][j];
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int k = p * 8 + (t >> 5), j4 = (t & 31) * 4;
                *reinterpret_cast<f32x4*>(bs + k * SBN + j4) = rb[p];
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    constexpr int SB = BT ? SBT : SBN;
    const int nk = ld / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = NBUF == 2 ? (kt & 1) : 0;
        if (kt + 1 < nk) gload((kt + 1) * BK);  // next tile's global loads fly under this tile's MFMAs
        const float* as = smem + cur * BUF_ELEMS + wm * 64 + li;
        const float* bs = smem + cur * BUF_ELEMS + A_ELEMS + wn * 64 + li;
        // fragments of k-step kk+2 are read from LDS before the four MFMAs of k-step kk are issued, so the
        // ds_read latency hides under 4 x 64 cycles of matrix work instead of stalling in front of them
        float a0 = as[lk * SA], a1 = as[lk * SA + 32];
        float b0 = bs[lk * SB], b1 = bs[lk * SB + 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a0n = 0.f, a1n = 0.f, b0n = 0.f, b1n = 0.f;
            if (kk + 2 < BK) {
                a0n = as[(kk + 2 + lk) * SA]; a1n = as[(kk + 2 + lk) * SA + 32];
                b0n = bs[(kk + 2 + lk) * SB]; b1n = bs[(kk + 2 + lk) * SB + 32];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep hipcc from sinking the reads back below the MFMAs
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
        }
        if (kt + 1 < nk) {
            if constexpr (NBUF == 2) {
                lstore(cur ^ 1);  // the other buffer was last read one barrier ago
                __syncthreads();
            } else {
                __syncthreads();  // every wave is done reading the only buffer
                lstore(0);
                __syncthreads();
            }
        }
    }

    // C/D map of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* Cg = C + (size_t)(tm * BM + wm * 64) * ld + tn * BN + wn * 64;
    const float* Qg = Qadd ? Qadd + (size_t)(tm * BM + wm * 64) * ld + tn * BN + wn * 64 : nullptr;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int col = j * 32 + li;
                float v = acc[i][j][r];
                if (Qg) v += Qg[(size_t)row * ld + col];
                Cg[(size_t)row * ld + col] = v;
            }
}

// Measured at N = 10003 (tools/dense_bench.py): single buffer 123.6 TFLOP/s, double buffer 122.0 -- the
// extra resident waves hide the second barrier, and 4 workgroups per CU need only 33 KB of LDS each.
static int g_dense_nbuf = 1;
void dense_gemm_set_buffers(int nbuf) { g_dense_nbuf = nbuf == 1 ? 1 : 2; }

size_t dense_gemm_lds_bytes(bool bt) {
    const int a = BK * SA, b = bt ? BK * SBT : BK * SBN;
    return (size_t)g_dense_nbuf * (a + b) * sizeof(float);
}

hipError_t dense_gemm_prepare() {
    const int a = BK * SA;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32<true, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (a + BK * SBT) * sizeof(float)));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32<false, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (a + BK * SBN) * sizeof(float)));
}

void launch_dense_gemm(const float* A, const float* B, float* C, const float* Qadd, int ld, bool b_transposed,
                       hipStream_t s) {
    const int tiles = ld / BM;
    dim3 grid(tiles * tiles);
    if (g_dense_nbuf == 2) {
        if (b_transposed)
            hipLaunchKernelGGL((k_gemm_f32<true, 2>), grid, dim3(256), dense_gemm_lds_bytes(true), s, A, B, C, Qadd, ld, tiles);
        else
            hipLaunchKernelGGL((k_gemm_f32<false, 2>), grid, dim3(256), dense_gemm_lds_bytes(false), s, A, B, C, Qadd, ld, tiles);
    } else {
        if (b_transposed)
            hipLaunchKernelGGL((k_gemm_f32<true, 1>), grid, dim3(256), dense_gemm_lds_bytes(true), s, A, B, C, Qadd, ld, tiles);
        else
            hipLaunchKernelGGL((k_gemm_f32<false, 1>), grid, dim3(256), dense_gemm_lds_bytes(false), s, A, B, C, Qadd, ld, tiles);
    }
}

}  // namespace ekf
