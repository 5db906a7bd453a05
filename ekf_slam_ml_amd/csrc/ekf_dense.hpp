// ekf_dense.hpp -- launcher of the fp32 MFMA GEMM used by the dense covariance propagation.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ekf {
constexpr int kDenseTile = 128;  // ld must be a multiple of this
// C[ld x ld] = A * B (+ Qadd), all row-major fp32 with zero padding up to ld.
// b_transposed: B is supplied as Bt[j][k] (i.e. C = A * Bt^T).
// n_rows (0 = ld): rows of C that are not padding; A's rows from there on are zero and C's were allocated zero.
void launch_dense_gemm(const float* A, const float* B, float* C, const float* Qadd, int ld, bool b_transposed,
                       hipStream_t s, int n_rows = 0);
// how launch_dense_gemm cuts a product: *tiles = ld / 128; n_big tiles of 256 x 128 on the main kernel (k_gemm_f32_big,
// whole rounds of resident workgroups); n_rem tiles of 128 x 128 -- the rest of the big-tile list and the bottom strip of
// an ld that is an odd multiple of 128 -- done as 4 * n_rem quarter tiles (k_gemm_f32_tail)
void dense_gemm_split(int ld, int* tiles, int* n_big, int* n_rem);
// map [tiles][tiles] over the 128 x 128 blocks of C: 0 = computed by the main kernel, 1 = by the tail kernel (255 never)
void dense_gemm_tile_map(int ld, unsigned char* map);
hipError_t dense_gemm_prepare();  // raises the dynamic-LDS limit of the main kernel (49.4 KB per workgroup)
}  // namespace ekf
