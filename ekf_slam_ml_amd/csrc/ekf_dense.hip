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
// Kernel: 256 x 128 block tile, BK = 32, 4 waves each owning a 128 x 64 sub-tile = 4 x 2
// v_mfma_f32_32x32x2_f32 accumulators (exact f32 FMA chains, 64 FLOP/clk/SIMD = the f32 peak); 128 accumulator
// registers per lane, two workgroups per CU.
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

// Big-tile id -> (tm, tn), tm in units of 256 rows, tn in units of 128 columns: the list is walked in groups of GROUP_M
// tile rows so that the A panel and the B panel of neighbouring tiles are re-used out of L2 (speed only).
__host__ __device__ inline void big_tile_of(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int g = id / per_group;
    const int first_m = g * GROUP_M;
    const int gm = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_g = id % per_group;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}

// How one product is cut (dense_gemm_split): the ld/256 x ld/128 list of 256 x 128 tiles runs as whole rounds of resident
// workgroups on the main kernel (ids [0, n_big)); what is left of the list (rem_big tiles = 2 rem_big tiles of 128 x 128)
// and, when ld is an odd multiple of 128, the bottom strip of ld/128 tiles of 128 x 128 make the small-tile list of the
// tail kernel, which cuts each of them into four 64 x 64 quarters.
struct DenseSplit {
    int ld, tiles_n, tiles_m, n_big, rem_big, bottom, n_small;
    int n_rows;   // rows of C that are not padding (N): a quarter tile that lies wholly below them is not computed -- its
                  // rows of C are products of A's zero padding and stay the zeros they were allocated as
};
// origin of small tile q (units: elements)
__host__ __device__ inline void small_tile_origin(const DenseSplit& sp, int q, int& row0, int& col0) {
    if (q < 2 * sp.rem_big) {
        int tm, tn;
        big_tile_of(sp.n_big + (q >> 1), sp.tiles_m, sp.tiles_n, tm, tn);
        row0 = tm * 256 + (q & 1) * 128;
        col0 = tn * 128;
    } else {
        row0 = sp.tiles_m * 256;
        col0 = (q - 2 * sp.rem_big) * 128;
    }
}

// One output tile of (64*WTM) x (64*WTN): 4 waves as 2 x 2, each owning WTM x WTN accumulators of 32 x 32.
// row0 / col0 = origin of the tile in C.  NBUF = 2: double-buffered LDS, one barrier per K tile;
// NBUF = 1: one buffer, two barriers per K tile, half the LDS (more workgroups per CU).
// Per k-step (two k values) a wave reads WTM + WTN fragment values from LDS for WTM x WTN MFMAs: 4 reads for 4 MFMAs
// at 2 x 2 (a 128 x 128 tile), 6 for 8 at 4 x 2 (256 x 128) -- and a K tile's staging stores and barriers are shared by
// twice the matrix work.
template <bool BT, int NBUF, int WTM, int WTN>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, const float* __restrict__ B,
                                          float* __restrict__ C, const float* __restrict__ Qadd, int ld, int row0,
                                          int col0, float* smem, int kdim) {
    constexpr int TM = 64 * WTM, TN = 64 * WTN;
    constexpr int SA = TM + 1;             // odd stride: conflict-free transposing stores and fragment reads
    constexpr int SB = BT ? TN + 1 : TN;   // a k-contiguous B operand is transposed like A; a row-major one is copied
    constexpr int A_ELEMS = BK * SA;
    constexpr int B_ELEMS = BK * SB;
    constexpr int BUF_ELEMS = (A_ELEMS + B_ELEMS + 3) / 4 * 4;  // keeps the second buffer 16-B aligned
    constexpr int PA = TM / 32;            // A staging passes: 32 rows x 32 k per pass
    constexpr int PBT = TN / 32;           // transposed-B staging passes
    constexpr int RB = 1024 / TN;          // row-major B: k rows per pass (256 lanes x float4 = 1024 floats)
    constexpr int PB = BK / RB;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const float* Ag = A + (size_t)row0 * ld;
    const float* Bg = BT ? B + (size_t)col0 * ld : B + col0;

    f32x4 ra[PA], rb[BT ? PBT : PB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; p++) {  // 8 lanes cover one 128-B row segment
            const int row = p * 32 + (t >> 3), k4 = (t & 7) * 4;
            ra[p] = *reinterpret_cast<const f32x4*>(Ag + (size_t)row * ld + k0 + k4);
        }
        if constexpr (BT) {
#pragma unroll
            for (int p = 0; p < PBT; p++) {
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
            for (int p = 0; p < PBT; p++) {
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

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; i++)
#pragma unroll
        for (int j = 0; j < WTN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int nk = (kdim + BK - 1) / BK;   // (the K range behind N is zero padding in both operands: 316 -> 313 K tiles at N = 10003)
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = NBUF == 2 ? (kt & 1) : 0;
        if (kt + 1 < nk) gload((kt + 1) * BK);  // next tile's global loads fly under this tile's MFMAs
        const float* as = smem + cur * BUF_ELEMS + wm * 32 * WTM + li;
        const float* bs = smem + cur * BUF_ELEMS + A_ELEMS + wn * 32 * WTN + li;
        // fragments of k-step kk+2 are read from LDS before the MFMAs of k-step kk are issued, so the
        // ds_read latency hides under the matrix work instead of stalling in front of it
        float a[WTM], b[WTN];
#pragma unroll
        for (int i = 0; i < WTM; i++) a[i] = as[lk * SA + 32 * i];
#pragma unroll
        for (int j = 0; j < WTN; j++) b[j] = bs[lk * SB + 32 * j];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float an[WTM], bn[WTN];
#pragma unroll
            for (int i = 0; i < WTM; i++) an[i] = 0.f;
#pragma unroll
            for (int j = 0; j < WTN; j++) bn[j] = 0.f;
            if (kk + 2 < BK) {
#pragma unroll
                for (int i = 0; i < WTM; i++) an[i] = as[(kk + 2 + lk) * SA + 32 * i];
#pragma unroll
                for (int j = 0; j < WTN; j++) bn[j] = bs[(kk + 2 + lk) * SB + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep hipcc from sinking the reads back below the MFMAs
#pragma unroll
            for (int i = 0; i < WTM; i++)
#pragma unroll
                for (int j = 0; j < WTN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WTM; i++) a[i] = an[i];
#pragma unroll
            for (int j = 0; j < WTN; j++) b[j] = bn[j];
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
    float* Cg = C + (size_t)(row0 + wm * 32 * WTM) * ld + col0 + wn * 32 * WTN;
    const float* Qg = Qadd ? Qadd + (size_t)(row0 + wm * 32 * WTM) * ld + col0 + wn * 32 * WTN : nullptr;
#pragma unroll
    for (int i = 0; i < WTM; i++)
#pragma unroll
        for (int j = 0; j < WTN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int col = j * 32 + li;
                float v = acc[i][j][r];
                if (Qg) v += Qg[(size_t)row * ld + col];
                Cg[(size_t)row * ld + col] = v;
            }
}

// N = 10003 (ld = 10112): 39 x 79 = 3081 tiles of 256 x 128 over 512 resident workgroup slots (2 per CU) = 6 whole rounds
// (3072 tiles) on the main kernel; the other 9 big tiles (18 tiles of 128 x 128) and the bottom strip (79 tiles of
// 128 x 128: ld is 39.5 x 256) are cut into 64 x 64 quarters on the tail kernel, which follows on the same stream
// (0.55-0.62 ms; quarters that hold padding rows only are skipped).
// Measured, tools/dense_bench.py at N = 10003: rounds 1-3 ran 128 x 128 tiles, three workgroups per CU, with the tail on a
// second stream: 32.6 ms per propagation (0.76-0.80 of the f32 matrix peak in the steady state of the main kernel itself:
// 4 LDS fragment reads per 4 MFMAs, 32 staging stores and two barriers per 64 MFMAs of a wave).  256 x 128 tiles: 30.7 ms;
// the tail on a lowest-priority second stream: 30.3 ms (its quarter tiles slow the main kernel of the NN product by more
// than their own 0.6 ms when they share the chip with it); the tail behind the main kernel: 30.15 ms = 132.8 TFLOP/s.
template <bool BT>
__global__ __launch_bounds__(256, 2) void k_gemm_f32_big(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, const float* __restrict__ Qadd,
                                                         DenseSplit sp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // consecutive workgroup ids are dealt round-robin to the 8 XCDs: remapped so that every XCD owns a contiguous chunk of
    // the (grouped) tile list and re-uses its panels out of its own L2
    int id = blockIdx.x;
    if (sp.n_big % 8 == 0) id = (id % 8) * (sp.n_big / 8) + id / 8;
    int tm, tn;
    big_tile_of(id, sp.tiles_m, sp.tiles_n, tm, tn);
    gemm_tile<BT, 1, 4, 2>(A, B, C, Qadd, sp.ld, tm * 256, tn * 128, smem, sp.n_rows);
}

template <bool BT>
__global__ __launch_bounds__(256, 4) void k_gemm_f32_tail(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, const float* __restrict__ Qadd,
                                                          DenseSplit sp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int s = blockIdx.x;
    int row0, col0;
    small_tile_origin(sp, s >> 2, row0, col0);
    row0 += ((s >> 1) & 1) * 64;
    if (row0 >= sp.n_rows) return;   // (uniform) padding rows only: at N = 10003 half of the bottom strip's quarters
    gemm_tile<BT, 1, 1, 1>(A, B, C, Qadd, sp.ld, row0, col0 + (s & 1) * 64, smem, sp.n_rows);
}

static size_t lds_bytes(int tm, int tn, bool bt) {
    const int a = BK * (tm + 1), b = bt ? BK * (tn + 1) : BK * tn;
    return (size_t)((a + b + 3) / 4 * 4) * sizeof(float);
}

hipError_t dense_gemm_prepare() {
    // 256 x 128 tiles stage 49.4 KB per workgroup: above the 48 KB a kernel may take without asking
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_big<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(256, 128, true));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32_big<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(256, 128, false));
}

static DenseSplit make_split(int ld, int n_rows = 0) {
    DenseSplit sp{};
    sp.ld = ld;
    sp.n_rows = n_rows > 0 ? n_rows : ld;
    sp.tiles_n = ld / kDenseTile;
    sp.tiles_m = ld / (2 * kDenseTile);
    sp.bottom = (ld % (2 * kDenseTile)) ? 1 : 0;
    const int total_big = sp.tiles_m * sp.tiles_n;
    const int slots = 256 * 2;   // resident workgroups: __launch_bounds__ of k_gemm_f32_big
    int n_big = total_big / slots * slots;
    if (n_big == 0) n_big = total_big;   // (less than one round: all of it on the main kernel)
    sp.n_big = n_big;
    sp.rem_big = total_big - n_big;
    sp.n_small = 2 * sp.rem_big + sp.bottom * sp.tiles_n;
    return sp;
}

void dense_gemm_split(int ld, int* tiles_out, int* n_big_out, int* n_rem_out) {
    const DenseSplit sp = make_split(ld);
    if (tiles_out) *tiles_out = sp.tiles_n;
    if (n_big_out) *n_big_out = sp.n_big;
    if (n_rem_out) *n_rem_out = sp.n_small;
}

void dense_gemm_tile_map(int ld, unsigned char* map) {
    const DenseSplit sp = make_split(ld);
    const int t = sp.tiles_n;
    for (int i = 0; i < t * t; i++) map[i] = 255;
    for (int id = 0; id < sp.n_big; id++) {   // (the XCD remap permutes ids inside [0, n_big): the set of tiles is the same)
        int tm, tn;
        big_tile_of(id, sp.tiles_m, sp.tiles_n, tm, tn);
        map[(2 * tm) * t + tn] = 0;
        map[(2 * tm + 1) * t + tn] = 0;
    }
    for (int q = 0; q < sp.n_small; q++) {
        int r0, c0;
        small_tile_origin(sp, q, r0, c0);
        map[(r0 / 128) * t + c0 / 128] = 1;
    }
}

void launch_dense_gemm(const float* A, const float* B, float* C, const float* Qadd, int ld, bool b_transposed,
                       hipStream_t s, int n_rows) {
    const DenseSplit sp = make_split(ld, n_rows);
    if (sp.n_big > 0) {
        const size_t lds = lds_bytes(256, 128, b_transposed);
        if (b_transposed) hipLaunchKernelGGL((k_gemm_f32_big<true>), dim3(sp.n_big), dim3(256), lds, s, A, B, C, Qadd, sp);
        else hipLaunchKernelGGL((k_gemm_f32_big<false>), dim3(sp.n_big), dim3(256), lds, s, A, B, C, Qadd, sp);
    }
    if (sp.n_small > 0) {   // behind the main kernel on the same stream (see k_gemm_f32_big)
        const size_t lds = lds_bytes(64, 64, b_transposed);
        if (b_transposed) hipLaunchKernelGGL((k_gemm_f32_tail<true>), dim3(4 * sp.n_small), dim3(256), lds, s, A, B, C, Qadd, sp);
        else hipLaunchKernelGGL((k_gemm_f32_tail<false>), dim3(4 * sp.n_small), dim3(256), lds, s, A, B, C, Qadd, sp);
    }
}

}  // namespace ekf
