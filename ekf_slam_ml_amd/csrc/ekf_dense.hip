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

constexpr int BK = 32;

// Tile id -> (tm, tn) in units of 128: walk the tiles in groups of GROUP_M block rows so the A panel and
// the B panel of neighbouring tiles are re-used out of L2 (speed only); when the tile count divides the
// XCD count, consecutive ids (which the dispatcher deals round-robin to the 8 XCDs) are remapped so that
// every XCD owns a contiguous chunk of the list.
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

// One output tile of (64*WT) x (64*WT): 4 waves as 2 x 2, each owning WT x WT accumulators of 32 x 32.
// row0 / col0 = origin of the tile in C.  NBUF = 2: double-buffered LDS, one barrier per K tile;
// NBUF = 1: one buffer, two barriers per K tile, half the LDS (more workgroups per CU).
template <bool BT, int NBUF, int WT>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, const float* __restrict__ B,
                                          float* __restrict__ C, const float* __restrict__ Qadd, int ld, int row0,
                                          int col0, float* smem) {
    constexpr int TM = 64 * WT, TN = 64 * WT;
    constexpr int SA = TM + 1;             // odd stride: conflict-free transposing stores and fragment reads
    constexpr int SB = BT ? TN + 1 : TN;   // a k-contiguous B operand is transposed like A; a row-major one is copied
    constexpr int A_ELEMS = BK * SA;
    constexpr int B_ELEMS = BK * SB;
    constexpr int BUF_ELEMS = (A_ELEMS + B_ELEMS + 3) / 4 * 4;  // keeps the second buffer 16-B aligned
    constexpr int PA = TM / 32;            // A (and transposed-B) staging passes: 32 rows x 32 k per pass
    constexpr int RB = 1024 / TN;          // row-major B: k rows per pass (256 lanes x float4 = 1024 floats)
    constexpr int PB = BK / RB;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const float* Ag = A + (size_t)row0 * ld;
    const float* Bg = BT ? B + (size_t)col0 * ld : B + col0;

    f32x4 ra[PA], rb[BT ? PA : PB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; p++) {  // 8 lanes cover one 128-B row segment
            const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
            ra[p] = *reinterpret_cast<const f32x4*>(Ag + (size_t)row * ld + k0 + k4);
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < PA; p++) {
                const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
                rb[p] = *reinterpret_cast<const f32x4*>(Bg + (size_t)row * ld + k0 + k4);
            }
        } else {
#pragma unroll
            for (int p = 0; p < PB; p++) {  // TN/4 lanes cover one row segment of the tile
                const int k = p * RB + t / (TN / 4), j4 = (t % (TN / 4)) * 4;
                rb[p] = *reinterpret_cast<const f32x4*>(Bg + (size_t)(k0 + k) * ld + j4);
            }
        }
    };
    auto lstore = [&](int buf) {
        float* as = smem + buf * BUF_ELEMS;
        float* bs = as + A_ELEMS;
#pragma unroll
        for (int p = 0; p < PA; p++) {
            const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) as[(k4 + j) * SA + row] = ra[p][j];
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < PA; p++) {
                const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) bs[(k4 + j) * SB + row] = rb[p][j];
            }
        } else {
#pragma unroll
            for (int p = 0; p < PB; p++) {
                const int k = p * RB + t / (TN / 4), j4 = (t % (TN / 4)) * 4;
                *reinterpret_cast<f32x4*>(bs + k * SB + j4) = rb[p];
            }
        }
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; i++)
#pragma unroll
        for (int j = 0; j < WT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int nk = ld / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = NBUF == 2 ? (kt & 1) : 0;
        if (kt + 1 < nk) gload((kt + 1) * BK);  // next tile's global loads fly under this tile's MFMAs
        const float* as = smem + cur * BUF_ELEMS + wm * 32 * WT + li;
        const float* bs = smem + cur * BUF_ELEMS + A_ELEMS + wn * 32 * WT + li;
        // fragments of k-step kk+2 are read from LDS before the MFMAs of k-step kk are issued, so the
        // ds_read latency hides under the matrix work instead of stalling in front of it
        float a[WT], b[WT];
#pragma unroll
        for (int i = 0; i < WT; i++) { a[i] = as[lk * SA + 32 * i]; b[i] = bs[lk * SB + 32 * i]; }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float an[WT], bn[WT];
#pragma unroll
            for (int i = 0; i < WT; i++) { an[i] = 0.f; bn[i] = 0.f; }
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < WT; i++) { an[i] = as[(kk + 2 + lk) * SA + 32 * i]; bn[i] = bs[(kk + 2 + lk) * SB + 32 * i]; }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep hipcc from sinking the reads back below the MFMAs
#pragma unroll
            for (int i = 0; i < WT; i++)
#pragma unroll
                for (int j = 0; j < WT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WT; i++) { a[i] = an[i]; b[i] = bn[i]; }
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
    float* Cg = C + (size_t)(row0 + wm * 32 * WT) * ld + col0 + wn * 32 * WT;
    const float* Qg = Qadd ? Qadd + (size_t)(row0 + wm * 32 * WT) * ld + col0 + wn * 32 * WT : nullptr;
#pragma unroll
    for (int i = 0; i < WT; i++)
#pragma unroll
        for (int j = 0; j < WT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int col = j * 32 + li;
                float v = acc[i][j][r];
                if (Qg) v += Qg[(size_t)row * ld + col];
                Cg[(size_t)row * ld + col] = v;
            }
}

// 79 x 79 = 6241 tiles do not divide the 768 resident workgroup slots (3 per CU: 8.13 rounds), so only the 8 full
// rounds run as 128 x 128 tiles (k_gemm_f32); the last 97 tiles are cut into 64 x 64 quarters
// (k_gemm_f32_tail, launched on a second stream so that its workgroups fill the slots the big kernel's
// last round leaves idle) and end the product in a quarter of a tile time (speed only).
template <bool BT, int NBUF>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 3) void k_gemm_f32(const float* __restrict__ A,
                                                                      const float* __restrict__ B,
                                                                      float* __restrict__ C,
                                                                      const float* __restrict__ Qadd, int ld,
                                                                      int tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int tm, tn;
    tile_of(blockIdx.x, tiles, tm, tn);
    gemm_tile<BT, NBUF, 2>(A, B, C, Qadd, ld, tm * 128, tn * 128, smem);
}

template <bool BT, int NBUF>
__global__ __launch_bounds__(256, 4) void k_gemm_f32_tail(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, const float* __restrict__ Qadd, int ld,
                                                          int tiles, int n_big) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s = blockIdx.x;
    int tm, tn;
    tile_of(n_big + (s >> 2), tiles, tm, tn);
    gemm_tile<BT, NBUF, 1>(A, B, C, Qadd, ld, tm * 128 + ((s >> 1) & 1) * 64, tn * 128 + (s & 1) * 64, smem);
}

// Measured at N = 10003 (tools/dense_bench.py): single buffer 123.6 TFLOP/s, double buffer 122.0 -- the
// extra resident waves hide the second barrier, and 3 workgroups per CU need only 33 KB of LDS each.
static int g_dense_nbuf = 1;
void dense_gemm_set_buffers(int nbuf) { g_dense_nbuf = nbuf == 1 ? 1 : 2; }

size_t dense_gemm_lds_bytes(bool bt) {
    const int a = BK * 129, b = bt ? BK * 129 : BK * 128;
    return (size_t)g_dense_nbuf * ((a + b + 3) / 4 * 4) * sizeof(float);
}

hipError_t dense_gemm_prepare() {
    const int a = BK * 129;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32<true, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (a + BK * 129 + 3) * sizeof(float)));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32<false, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (a + BK * 128 + 3) * sizeof(float)));
}

void dense_gemm_split(int ld, bool has_tail_stream, int* tiles_out, int* n_big_out, int* n_rem_out) {
    const int tiles = ld / kDenseTile;
    const int total = tiles * tiles;
    // full rounds of resident workgroups run 128 x 128 tiles; the remainder is cut into 64 x 64 quarters
    const int slots = 256 * (g_dense_nbuf == 2 ? 2 : 3);  // resident workgroups: __launch_bounds__ of k_gemm_f32
    int n_big = total / slots * slots;
    if (total % 8 == 0 || !has_tail_stream || n_big == 0) n_big = total;  // (the XCD remap needs the list in one piece)
    if (tiles_out) *tiles_out = tiles;
    if (n_big_out) *n_big_out = n_big;
    if (n_rem_out) *n_rem_out = total - n_big;
}

void launch_dense_gemm(const float* A, const float* B, float* C, const float* Qadd, int ld, bool b_transposed,
                       hipStream_t s, hipStream_t s_tail) {
    int tiles, n_big, n_rem;
    dense_gemm_split(ld, s_tail != nullptr, &tiles, &n_big, &n_rem);
    const size_t lds = dense_gemm_lds_bytes(b_transposed);
    if (g_dense_nbuf == 2) {
        if (b_transposed) hipLaunchKernelGGL((k_gemm_f32<true, 2>), dim3(n_big), dim3(256), lds, s, A, B, C, Qadd, ld, tiles);
        else hipLaunchKernelGGL((k_gemm_f32<false, 2>), dim3(n_big), dim3(256), lds, s, A, B, C, Qadd, ld, tiles);
    } else {
        if (b_transposed) hipLaunchKernelGGL((k_gemm_f32<true, 1>), dim3(n_big), dim3(256), lds, s, A, B, C, Qadd, ld, tiles);
        else hipLaunchKernelGGL((k_gemm_f32<false, 1>), dim3(n_big), dim3(256), lds, s, A, B, C, Qadd, ld, tiles);
    }
    if (n_rem > 0) {
        if (b_transposed)
            hipLaunchKernelGGL((k_gemm_f32_tail<true, 1>), dim3(4 * n_rem), dim3(256), lds, s_tail, A, B, C, Qadd, ld, tiles, n_big);
        else
            hipLaunchKernelGGL((k_gemm_f32_tail<false, 1>), dim3(4 * n_rem), dim3(256), lds, s_tail, A, B, C, Qadd, ld, tiles, n_big);
    }
}

}  // namespace ekf
