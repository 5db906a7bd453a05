// ekf_fused.hip -- one landmark correction in ONE launch (single filter, mid-size maps).
//
// The eager correction of ekf_kernels.hip is two dependent launches: k_gain (gathers 5 rows + 5 columns of
// Sigma, builds K and H*Sigma) and k_rank2 (streams Sigma).  For one filter at n = 200 ... 1000 both are a few
// microseconds of work, so the launch boundary and the host's launch cost dominate (DESIGN.md section 5:
// 11 us per correction at n = 200).  They cannot simply be merged in place: a workgroup that streams its tile
// would overwrite rows/columns another workgroup still has to gather.  Here the update is OUT OF PLACE
// (Sigma_next = Sigma - K (H Sigma), state_next = state + K nu; the host swaps the two buffers after every
// correction), so every workgroup gathers what it needs from the old buffer by itself:
//   * the 5x5 block, pose and landmark -> H, S^-1, nu           (every workgroup, redundantly; O(1))
//   * (H Sigma)(:, its 512 columns) from the 5 rows of the old Sigma                     (ekf_slam.cpp:178)
//   * K(its rows, :) from the 5 columns of its own rows                                  (ekf_slam.cpp:178)
//   * then streams its tile  Sigma_next = Sigma - K (H Sigma)                             (ekf_slam.cpp:191-192)
// Same operations in the same order as k_gain + k_rank2 -> bit-identical results.  Traffic is unchanged (read
// N^2 + write N^2) plus a redundant 5-row gather per tile, which at these sizes comes from L2 / Infinity Cache.
// Memory doubles, which is why this path serves single filters, not the 131-GB Monte-Carlo pool.
#include "ekf_kernels.hpp"

#include <climits>

namespace ekf {

__global__ __launch_bounds__(256) void k_correct_fused(PoolView pv, CmdSrc src, double* __restrict__ sig_next,
                                                       double* __restrict__ st_next, int rows_per_block) {
    const int b = blockIdx.z, tid = threadIdx.x;
    const int N = pv.N, ld = pv.ld;  // N may be a discovered prefix (data_association)
    const double* __restrict__ cur = pv.sigma + (size_t)b * pv.sigma_stride;
    double* __restrict__ nxt = sig_next + (size_t)b * pv.sigma_stride;
    const double* __restrict__ st = pv.state + (size_t)b * ld;
    double* __restrict__ stn = st_next + (size_t)b * ld;
    const int ld2n = ld >> 1, ld2a = (N + 1) >> 1;
    const int c2 = blockIdx.x * 256 + tid;
    const int row_begin = blockIdx.y * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);
    const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
    const double2_t* __restrict__ cur2 = reinterpret_cast<const double2_t*>(cur);
    double2_t* __restrict__ nxt2 = reinterpret_cast<double2_t*>(nxt);

    __shared__ double sh_S55[25];
    __shared__ double sh_H[10];
    __shared__ double sh_Si[4];
    __shared__ double sh_nu[2];
    __shared__ double2_t sh_K[256];

    int lm = -1;
    double sx = 0.0, sy = 0.0;
    if (src.mode == SRC_SENSOR_VECTOR) {
        lm = src.lm_imm;
        sx = src.sensor[(size_t)b * 2 * pv.n + 2 * lm];
        sy = src.sensor[(size_t)b * 2 * pv.n + 2 * lm + 1];
    } else if (src.mode == SRC_COMPACT_LOG) {
        const size_t slot = (size_t)b * src.vmax + src.v;
        lm = src.lm_idx[slot];
        if (lm >= 0) {
            sx = src.z_xy[slot * 2];
            sy = src.z_xy[slot * 2 + 1];
        }
    } else {
        const AssocRec a = src.assoc[b];
        lm = a.active ? a.lm : -1;
        sx = src.meas[(size_t)b * src.meas_stride];
        sy = src.meas[(size_t)b * src.meas_stride + 1];
    }
    // the state beyond the active dimension is carried over unchanged
    if (lead)
        for (int r = N + tid; r < ld; r += 256) stn[r] = st[r];

    if (lm < 0 || lm >= pv.n) {  // measurement dropped (ekf_slam.cpp:330): the buffers still swap, so copy
        if (c2 < ld2a)
            for (int r = row_begin; r < row_end; r++) nxt2[(size_t)r * ld2n + c2] = cur2[(size_t)r * ld2n + c2];
        if (blockIdx.x == 0)
            for (int r = row_begin + tid; r < row_end; r += 256) stn[r] = st[r];
        if (lead && tid == 0) pv.rec[b].active = 0;
        return;
    }

    // all gathers are issued up front and fly together
    double2_t gk[5];
    if (c2 < ld2a) {
#pragma unroll
        for (int k = 0; k < 5; k++) gk[k] = cur2[(size_t)idx5(k, lm) * ld2n + c2];   // rows of H*Sigma
    }
    double p[5];
    const int kr = row_begin + tid;
    if (kr < row_end) {
        gather_row5(cur + (size_t)kr * ld, lm, p);                                   // columns of Sigma*H^T: three loads
    }
    if (tid < 25) sh_S55[tid] = cur[(size_t)idx5(tid / 5, lm) * ld + idx5(tid % 5, lm)];
    double theta = 0.0, x = 0.0, y = 0.0, tx = 0.0, ty = 0.0;
    if (tid == 0) {
        if (src.fresh_pose) {
            theta = st[0]; x = st[1]; y = st[2];
        } else {
            const double* sn = pv.snap + (size_t)b * 4;
            theta = sn[0]; x = sn[1]; y = sn[2];
        }
        tx = st[2 * lm + 3];
        ty = st[2 * lm + 4];
    }
    __syncthreads();
    if (tid == 0) {
        MeasTerms m;
        measurement_terms(tx, ty, sx, sy, theta, x, y, m);
        double S55[5][5], S[2][2], Si[2][2];
        for (int k = 0; k < 5; k++)
            for (int l = 0; l < 5; l++) S55[k][l] = sh_S55[k * 5 + l];
        innovation_cov(S55, m.H, pv.p.r_meas, S);
        inv2(S, Si);
        for (int a = 0; a < 2; a++)
            for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = m.H[a][k];
        sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
        sh_nu[0] = m.z0 - m.zh0;                    // :182
        sh_nu[1] = normalize_angle(m.z1 - m.zh1);   // :183
        if (lead) {
            if (src.write_snap) {  // written only now: every workgroup has read what it needs from state, not snap
                double* sn = pv.snap + (size_t)b * 4;
                sn[0] = theta; sn[1] = x; sn[2] = y;
            }
            CorrRec rc;
            rc.nu0 = sh_nu[0]; rc.nu1 = sh_nu[1];
            rc.active = 1; rc.lm = lm; rc.n_active = 0; rc.pad = 0;
            pv.rec[b] = rc;
            touch_landmark(pv, b, lm);
        }
    }
    __syncthreads();

    if (kr < row_end) {
        double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sht0 += p[k] * sh_H[k];
            sht1 += p[k] * sh_H[5 + k];
        }
        sh_K[tid] = double2_t{sht0 * sh_Si[0] + sht1 * sh_Si[2], sht0 * sh_Si[1] + sht1 * sh_Si[3]};  // :178
    }
    __syncthreads();

    if (c2 < ld2a) {
        double2_t g0{0.0, 0.0}, g1{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            g0.x += sh_H[k] * gk[k].x; g0.y += sh_H[k] * gk[k].y;
            g1.x += sh_H[5 + k] * gk[k].x; g1.y += sh_H[5 + k] * gk[k].y;
        }
        if (2 * c2 + 1 >= N) { g0.y = 0.0; g1.y = 0.0; }  // column N: pad, or the first column beyond the prefix
        for (int r = row_begin; r < row_end; r++) {
            const double2_t k = sh_K[r - row_begin];
            double2_t v = cur2[(size_t)r * ld2n + c2];
            v.x = v.x - (k.x * g0.x + k.y * g1.x);  // :191-192
            v.y = v.y - (k.x * g0.y + k.y * g1.y);
            nxt2[(size_t)r * ld2n + c2] = v;
        }
    }
    if (blockIdx.x == 0 && kr < row_end) {  // state = state + Ki*z_diff (:186); theta wrapped (:187)
        const double2_t k = sh_K[tid];
        double s = st[kr] + (k.x * sh_nu[0] + k.y * sh_nu[1]);
        if (kr == 0) s = normalize_angle(s);
        stn[kr] = s;
    }
}

void launch_correct_fused(const PoolView& pv, const CmdSrc& src, double* sigma_next, double* state_next,
                          hipStream_t s) {
    const int ld2a = (pv.N + 1) / 2;
    const int strips = (ld2a + 255) / 256;
    // enough workgroups to cover the chip a couple of times; each re-gathers 5 rows of its strip
    int rows = 16;
    while (rows > 2 && (long long)strips * ((pv.N + rows - 1) / rows) * pv.B < 512) rows >>= 1;
    dim3 grid(strips, (pv.N + rows - 1) / rows, pv.B);
    hipLaunchKernelGGL(k_correct_fused, grid, dim3(256), 0, s, pv, src, sigma_next, state_next, rows);
}

}  // namespace ekf

namespace ekf {

// ---------------------------------------------------------------------------------------------
// data_association() of one measurement in TWO launches instead of four: k_maha (scores, and the correction
// terms of every scored landmark) and this kernel, which is k_assoc_decide + k_correct_fused in one.  Every
// workgroup repeats the decision (the lexicographic (d, i) min over the M scores, the two gates, the position of
// a new landmark) -- it is a few hundred bytes and one reduction -- and then gathers, builds K and streams its
// tile exactly as k_correct_fused does.  For a matched landmark H, S^-1 and nu come from k_maha's record (same
// pose, same Sigma, same functions: bit-identical to rebuilding them); a new landmark has none and builds them.
// Nothing is updated in place: the association record, the state and Sigma are written to their "next" buffers
// and the host swaps all three.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_associate_fused(PoolView pv, MeasSrc ms, const double* __restrict__ scores,
                                                         AssocRec* __restrict__ assoc_next, int* __restrict__ assoc_out_j,
                                                         double* __restrict__ sig_next, double* __restrict__ st_next,
                                                         int rows_per_block) {
    const int b = blockIdx.z, tid = threadIdx.x;
    const int N = pv.N, ld = pv.ld, n = pv.n;
    const double* __restrict__ cur = pv.sigma + (size_t)b * pv.sigma_stride;
    double* __restrict__ nxt = sig_next + (size_t)b * pv.sigma_stride;
    const double* __restrict__ st = pv.state + (size_t)b * ld;
    double* __restrict__ stn = st_next + (size_t)b * ld;
    const int ld2n = ld >> 1, ld2a = (N + 1) >> 1;
    const int c2 = blockIdx.x * 256 + tid;
    const int row_begin = blockIdx.y * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);
    const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
    const double2_t* __restrict__ cur2 = reinterpret_cast<const double2_t*>(cur);
    double2_t* __restrict__ nxt2 = reinterpret_cast<double2_t*>(nxt);
    const double meas[2] = {ms.by_value ? ms.vx : ms.xy[(size_t)b * ms.stride], ms.by_value ? ms.vy : ms.xy[(size_t)b * ms.stride + 1]};

    __shared__ double sh_d[4];
    __shared__ int sh_i[4];
    __shared__ int sh_lm, sh_new;
    __shared__ double sh_t[2];  // position of a landmark initialised by this measurement
    __shared__ double sh_S55[25];
    __shared__ double sh_H[10];
    __shared__ double sh_Si[4];
    __shared__ double sh_nu[2];
    __shared__ double2_t sh_K[256];

    // ---- decision, ekf_slam.cpp:293-330 (k_assoc_decide) ----
    const int M = pv.assoc[b].known_count;
    double best = pv.p.gate_new;  // :293
    int bi = INT_MAX;
    for (int i = tid; i < M; i += 256) {
        const double d = scores[(size_t)b * n + i];
        if (d < best) { best = d; bi = i; }  // :305-309
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(best, off, kWave);
        const int oi = __shfl_down(bi, off, kWave);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    if ((tid & 63) == 0) { sh_d[tid >> 6] = best; sh_i[tid >> 6] = bi; }
    if (lead)  // the state beyond the active dimension is carried over unchanged
        for (int r = N + tid; r < ld; r += 256) stn[r] = st[r];
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++)
            if (sh_d[w] < best || (sh_d[w] == best && sh_i[w] < bi)) { best = sh_d[w]; bi = sh_i[w]; }
        const int idx = (bi == INT_MAX) ? M : bi;  // :294 min_maha_idx = known_count
        int known_count = M, is_new = 0;
        if (idx == M && idx < n) {  // :318-327 new landmark
            const double theta = st[0], x = st[1], y = st[2];
            const double sx = meas[0], sy = meas[1];
            const double ri = sqrt(sx * sx + sy * sy);
            const double phii = atan2(sy, sx);
            sh_t[0] = x + ri * cos(phii + theta);
            sh_t[1] = y + ri * sin(phii + theta);
            known_count = M + 1;
            best = 0.0;
            is_new = 1;
        }
        const int active = (best < pv.p.gate_update) && idx < n;  // :330
        sh_lm = active ? idx : -1;
        sh_new = is_new;
        if (lead) {
            AssocRec a;
            a.known_count = known_count; a.lm = active ? idx : -1; a.active = active; a.pad = 0; a.best = best;
            assoc_next[b] = a;
            if (assoc_out_j) assoc_out_j[b] = a.lm;
        }
    }
    __syncthreads();
    const int lm = sh_lm;
    const int is_new = sh_new;

    if (lm < 0) {  // measurement dropped: the buffers still swap, so copy
        if (c2 < ld2a)
            for (int r = row_begin; r < row_end; r++) nxt2[(size_t)r * ld2n + c2] = cur2[(size_t)r * ld2n + c2];
        if (blockIdx.x == 0)
            for (int r = row_begin + tid; r < row_end; r += 256) stn[r] = st[r];
        if (lead && tid == 0) pv.rec[b].active = 0;
        return;
    }

    // ---- correction, :331-390 (k_correct_fused) ----
    double2_t gk[5];
    if (c2 < ld2a) {
#pragma unroll
        for (int k = 0; k < 5; k++) gk[k] = cur2[(size_t)idx5(k, lm) * ld2n + c2];
    }
    double p[5];
    const int kr = row_begin + tid;
    if (kr < row_end) {
        gather_row5(cur + (size_t)kr * ld, lm, p);
    }
    if (is_new || !ms.terms) {  // no record from k_maha: build H, S^-1, nu from the 5x5 block (uniform branch)
        if (tid < 25) sh_S55[tid] = cur[(size_t)idx5(tid / 5, lm) * ld + idx5(tid % 5, lm)];
        __syncthreads();
        if (tid == 0) {
            const double tx = is_new ? sh_t[0] : st[2 * lm + 3], ty = is_new ? sh_t[1] : st[2 * lm + 4];
            MeasTerms m;
            measurement_terms(tx, ty, meas[0], meas[1], st[0], st[1], st[2], m);  // fresh pose, :331-333
            double S55[5][5], S[2][2], Si[2][2];
            for (int k = 0; k < 5; k++)
                for (int l = 0; l < 5; l++) S55[k][l] = sh_S55[k * 5 + l];
            innovation_cov(S55, m.H, pv.p.r_meas, S);
            inv2(S, Si);
            for (int a = 0; a < 2; a++)
                for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = m.H[a][k];
            sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
            sh_nu[0] = m.z0 - m.zh0;
            sh_nu[1] = normalize_angle(m.z1 - m.zh1);
        }
    } else {
        const double* tr = ms.terms + ((size_t)b * n + lm) * 16;
        if (tid < 10) sh_H[tid] = tr[tid];
        else if (tid < 14) sh_Si[tid - 10] = tr[tid];
        else if (tid == 14) sh_nu[0] = tr[14];                      // :182
        else if (tid == 15) sh_nu[1] = normalize_angle(tr[15]);     // :183 (the score used it unwrapped, :269)
    }
    if (lead && tid == 0) {
        CorrRec rc;
        rc.nu0 = 0.0; rc.nu1 = 0.0; rc.active = 1; rc.lm = lm; rc.n_active = 0; rc.pad = 0;
        pv.rec[b] = rc;
        touch_landmark(pv, b, lm);
    }
    __syncthreads();

    if (kr < row_end) {
        double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sht0 += p[k] * sh_H[k];
            sht1 += p[k] * sh_H[5 + k];
        }
        sh_K[tid] = double2_t{sht0 * sh_Si[0] + sht1 * sh_Si[2], sht0 * sh_Si[1] + sht1 * sh_Si[3]};  // :178
    }
    __syncthreads();

    if (c2 < ld2a) {
        double2_t g0{0.0, 0.0}, g1{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            g0.x += sh_H[k] * gk[k].x; g0.y += sh_H[k] * gk[k].y;
            g1.x += sh_H[5 + k] * gk[k].x; g1.y += sh_H[5 + k] * gk[k].y;
        }
        if (2 * c2 + 1 >= N) { g0.y = 0.0; g1.y = 0.0; }
        for (int r = row_begin; r < row_end; r++) {
            const double2_t k = sh_K[r - row_begin];
            double2_t v = cur2[(size_t)r * ld2n + c2];
            v.x = v.x - (k.x * g0.x + k.y * g1.x);  // :389-390
            v.y = v.y - (k.x * g0.y + k.y * g1.y);
            nxt2[(size_t)r * ld2n + c2] = v;
        }
    }
    if (blockIdx.x == 0 && kr < row_end) {  // state = state + Ki*z_diff (:384); theta wrapped (:385)
        const double2_t k = sh_K[tid];
        double base = st[kr];
        if (is_new && kr == 2 * lm + 3) base = sh_t[0];  // the landmark this measurement initialised (:321-322)
        if (is_new && kr == 2 * lm + 4) base = sh_t[1];
        double s = base + (k.x * sh_nu[0] + k.y * sh_nu[1]);
        if (kr == 0) s = normalize_angle(s);
        stn[kr] = s;
    }
}

void launch_associate_fused(const PoolView& pv, const MeasSrc& ms, const double* scores, AssocRec* assoc_next,
                            int* assoc_out_j, double* sigma_next, double* state_next, hipStream_t s) {
    const int ld2a = (pv.N + 1) / 2;
    const int strips = (ld2a + 255) / 256;
    int rows = 16;
    while (rows > 2 && (long long)strips * ((pv.N + rows - 1) / rows) * pv.B < 512) rows >>= 1;
    dim3 grid(strips, (pv.N + rows - 1) / rows, pv.B);
    hipLaunchKernelGGL(k_associate_fused, grid, dim3(256), 0, s, pv, ms, scores, assoc_next, assoc_out_j, sigma_next,
                       state_next, rows);
}

}  // namespace ekf
