// ekf_dense.hpp -- launcher of the fp32 MFMA GEMM used by the dense covariance propagation.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ekf {
constexpr int kDenseTile = 128;  // ld must be a multiple of this
// C[ld x ld] = A * B (+ Qadd), all row-major fp32 with zero padding up to ld.
// b_transposed: B is supplied as Bt[j][k] (i.e. C = A * Bt^T).
// s_tail (nullable): second stream for the quarter-tile remainder kernel; the CALLER orders the two streams
// around the call (both must be after the producers of A and B; consumers of C must wait for both).
void launch_dense_gemm(const float* A, const float* B, float* C, const float* Qadd, int ld, bool b_transposed,
                       hipStream_t s, hipStream_t s_tail);
// how launch_dense_gemm cuts the ld/128 x ld/128 tile list: n_big full 128 x 128 tiles (k_gemm_f32) and n_rem tiles
// done as 4 * n_rem quarter tiles (k_gemm_f32_tail, only with a tail stream)
void dense_gemm_split(int ld, bool has_tail_stream, int* tiles, int* n_big, int* n_rem);
size_t dense_gemm_lds_bytes(bool b_transposed);
void dense_gemm_set_buffers(int nbuf);  // 1 (default) or 2 LDS buffers per workgroup
hipError_t dense_gemm_prepare();  // raises the dynamic-LDS limit of both instantiations (66 KB > 64 KB default)
}  // namespace ekf
