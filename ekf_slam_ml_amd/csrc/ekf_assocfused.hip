// ekf_assocfused.hip -- data_association() of a SINGLE filter beyond the LDS-resident path (ekf_slam.cpp:278-402) with
// the covariance streamed ONCE per call: one launch per reading that scores, decides and builds the gain against the
// STORED covariance minus the call's pending rank-2 pairs, then one k_rank2v pass (ekf_callfused.hip) at the end of the
// call.  The round-1 form was two launches per reading (k_maha, k_associate_fused), the second streaming all of Sigma.
//
// k_assoc_score, one landmark per thread: the 25 entries of Sigma[c5(i), c5(i)] are the stored entries minus the pending
//              pairs, in order, with the rank-2 kernel's own expression (the values the per-reading path would have
//              read, bit for bit); summation order of innovation_cov = k_maha's shuffle folds.  (Scoring all M landmarks
//              redundantly inside every workgroup of the next kernel was tried first: 19 us per reading at M = 1000.)
// k_assoc_meas, grid = slices of 256 state indices (8 workgroups at n = 1000), 512 threads each:
//   decision   lexicographic (d, i) minimum, the gates 10.0 / 1.0, landmark initialisation (:293-330): identical in
//              every workgroup; workgroup 0 records it
//   gain       threads < 256: K(i, :) and G(:, i) of their state index from the same reconstruction of five rows /
//              columns (k_gain's arithmetic), appended as pair `pc`; state(i) += K nu, out of place (:376-385)
// Same operations in the same order as k_maha + k_assoc_decide + k_gain + k_rank2 -> decisions, state and covariance
// bit-identical (tests/test_gpu_fused.py).
#include "ekf_kernels.hpp"

#include <climits>

namespace ekf {

constexpr int kAssocThreads = 512;
constexpr int kAssocSlice = 256;

// Scores of one reading against the M known landmarks: ONE LANDMARK PER THREAD, 64-thread workgroups spread over the
// chip (the per-landmark chain -- two atan2, the divisions of H, S, S^-1 -- is ~800 instructions; M = 1000 takes two
// rounds on 8 redundant workgroups but one round on 16 workgroups of 64).  Leaves score and correction terms per landmark.
__global__ __launch_bounds__(64) void k_assoc_score(PoolView pv, double mx, double my,
                                                    const AssocRec* __restrict__ assoc_in, double* __restrict__ scores,
                                                    double* __restrict__ terms, const double* __restrict__ Ub,
                                                    const double* __restrict__ Vb, int pc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ld = pv.ld;
    const double* __restrict__ Sg = pv.sigma;
    const double* __restrict__ st = pv.state;
    const int M = assoc_in[0].known_count;
    if (i >= M || i >= pv.n) return;
    MeasTerms m;
    measurement_terms(st[2 * i + 3], st[2 * i + 4], mx, my, st[0], st[1], st[2], m);   // fresh pose, :219-221
    double S55[5][5], S[2][2], Si[2][2];
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
        for (int l = 0; l < 5; l++) S55[k][l] = Sg[(size_t)idx5(k, i) * ld + idx5(l, i)];
    for (int v = 0; v < pc; v++) {   // ... as they stand NOW: minus the pending pairs of the call, in order
        double kr[5][2], gc[5][2];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int c = idx5(k, i);
            kr[k][0] = Ub[(size_t)(2 * v) * ld + c]; kr[k][1] = Ub[(size_t)(2 * v + 1) * ld + c];
            gc[k][0] = Vb[(size_t)(2 * v) * ld + c]; gc[k][1] = Vb[(size_t)(2 * v + 1) * ld + c];
        }
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
    }
    innovation_cov(S55, m.H, pv.p.r_meas, S);   // sums in the order of k_maha's shuffle folds
    inv2(S, Si);
    const double v0 = m.z0 - m.zh0, v1 = m.z1 - m.zh1;   // bearing NOT wrapped, :269
    const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
    const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
    scores[i] = t0 * v0 + t1 * v1;
    double* tr = terms + (size_t)i * 16;
#pragma unroll
    for (int k = 0; k < 5; k++) { tr[k] = m.H[0][k]; tr[5 + k] = m.H[1][k]; }
    tr[10] = Si[0][0]; tr[11] = Si[0][1]; tr[12] = Si[1][0]; tr[13] = Si[1][1];
    tr[14] = v0; tr[15] = v1;
}

__global__ __launch_bounds__(kAssocThreads) void k_assoc_meas(PoolView pv, double mx, double my,
                                                              const AssocRec* __restrict__ assoc_in,
                                                              AssocRec* __restrict__ assoc_next, int* __restrict__ assoc_out_j,
                                                              double* __restrict__ state_out, double* __restrict__ Uall,
                                                              double* __restrict__ Vall, int* __restrict__ cnt_out, int pc,
                                                              int Nb, int zero_upto, const double* __restrict__ scores,
                                                              const double* __restrict__ terms) {
    const int tid = threadIdx.x;
    const int n = pv.n, ld = pv.ld;
    const double* __restrict__ Sg = pv.sigma;
    const double* __restrict__ st = pv.state;
    // (the factor rows < 2 pc are read-only here; rows 2 pc, 2 pc + 1 are written: distinct rows of the same buffers)
    const double* Ub = Uall;
    const double* Vb = Vall;
    const bool lead = blockIdx.x == 0;
    // last reading of a pass: the pair rows up to the streaming kernel's (rounded) correction count become exact no-ops
    {
        const int i0 = blockIdx.x * kAssocSlice + tid;
        if (tid < kAssocSlice && i0 < ld)
            for (int v = pc + 1; v < zero_upto; v++) {
                Uall[(size_t)(2 * v) * ld + i0] = 0.0; Uall[(size_t)(2 * v + 1) * ld + i0] = 0.0;
                Vall[(size_t)(2 * v) * ld + i0] = 0.0; Vall[(size_t)(2 * v + 1) * ld + i0] = 0.0;
            }
    }

    __shared__ double sh_d[kAssocThreads / 64];
    __shared__ int sh_i[kAssocThreads / 64];
    __shared__ int sh_lm, sh_new;
    __shared__ double sh_t[2];
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ double sh_K5[kCallV][5][2], sh_G5[kCallV][5][2];   // pending pairs at {0,1,2} and at the winner's two indices

    const int M = assoc_in[0].known_count;
    const double theta = st[0], x = st[1], y = st[2];   // fresh pose, :219-221 / :331-333
    if (tid < 6 * pc) {   // pose part of the pending pairs
        const int v = tid / 6, r = (tid % 6) >> 1, h = tid & 1;
        sh_K5[v][r][h] = Ub[(size_t)(2 * v + h) * ld + r];
        sh_G5[v][r][h] = Vb[(size_t)(2 * v + h) * ld + r];
    }
    __syncthreads();

    // ---- scores, :300-309: left by k_assoc_score ----
    double best = pv.p.gate_new;  // :293
    int bi = INT_MAX;
    for (int i = tid; i < M; i += kAssocThreads) {
        const double d = scores[i];
        if (d < best) { best = d; bi = i; }  // :305-309 (NaN never wins)
    }
    double rd = best;
    int ri = bi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(rd, off, kWave);
        const int oi = __shfl_down(ri, off, kWave);
        if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
    }
    if ((tid & 63) == 0) { sh_d[tid >> 6] = rd; sh_i[tid >> 6] = ri; }
    __syncthreads();
    if (tid == 0) {   // :293-330
        for (int w = 1; w < kAssocThreads / 64; w++)
            if (sh_d[w] < rd || (sh_d[w] == rd && sh_i[w] < ri)) { rd = sh_d[w]; ri = sh_i[w]; }
        const int idx = (ri == INT_MAX) ? M : ri;   // :294 min_maha_idx = known_count
        int known_count = M, is_new = 0;
        if (idx == M && idx < n) {                  // :318-327 new landmark
            const double rr = sqrt(mx * mx + my * my);
            const double phi = atan2(my, mx);
            sh_t[0] = x + rr * cos(phi + theta);
            sh_t[1] = y + rr * sin(phi + theta);
            known_count = M + 1;
            rd = 0.0;
            is_new = 1;
        }
        const int active = (rd < pv.p.gate_update) && idx < n;   // :330
        sh_lm = active ? idx : -1;
        sh_new = is_new;
        if (lead) {
            AssocRec a;
            a.known_count = known_count; a.lm = active ? idx : -1; a.active = active; a.pad = 0; a.best = rd;
            assoc_next[0] = a;
            if (assoc_out_j) assoc_out_j[0] = a.lm;
            cnt_out[0] = pc + 1;
            CorrRec rc;
            rc.nu0 = 0.0; rc.nu1 = 0.0; rc.active = active; rc.lm = a.lm; rc.n_active = 0; rc.pad = 0;
            pv.rec[0] = rc;
            if (active) touch_landmark(pv, 0, idx);
        }
    }
    __syncthreads();
    const int lm = sh_lm;
    const int is_new = sh_new;
    const int i = blockIdx.x * kAssocSlice + tid;
    double* Uw = Uall + (size_t)(2 * pc) * ld;
    double* Vw = Vall + (size_t)(2 * pc) * ld;
    if (lm < 0) {   // dropped (uniform): a zero pair keeps the call's pair index = reading index; the state is carried over
        if (tid < kAssocSlice && i < ld) {
            Uw[i] = 0.0; Uw[ld + i] = 0.0; Vw[i] = 0.0; Vw[ld + i] = 0.0;
            state_out[i] = st[i];
            // (a landmark initialised by a reading that is then dropped cannot occur: a new landmark gets best = 0)
        }
        return;
    }
    // pending pairs at the winner's two indices; the winner's terms (from the thread that scored it, or built for a new one)
    if (tid < 4 * pc) {
        const int v = tid >> 2, q = (tid >> 1) & 1, h = tid & 1;
        sh_K5[v][3 + q][h] = Ub[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
        sh_G5[v][3 + q][h] = Vb[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
    }
    if (!is_new && tid < 16) {   // the winner's terms as the scoring kernel left them
        const double tv = terms[(size_t)lm * 16 + tid];
        if (tid < 10) sh_H[tid] = tv;
        else if (tid < 14) sh_Si[tid - 10] = tv;
        else if (tid == 14) sh_nu[0] = tv;
        else sh_nu[1] = normalize_angle(tv);   // :183 (the score used it unwrapped)
    }
    __syncthreads();
    if (is_new && tid == 0) {   // :331-381 with the fresh pose: a new landmark has no score record
        MeasTerms m;
        measurement_terms(sh_t[0], sh_t[1], mx, my, theta, x, y, m);
        double S55[5][5], S[2][2], Si[2][2];
        for (int k = 0; k < 5; k++)
            for (int l = 0; l < 5; l++) {
                double xe = Sg[(size_t)idx5(k, lm) * ld + idx5(l, lm)];
                for (int v = 0; v < pc; v++) xe = xe - (sh_K5[v][k][0] * sh_G5[v][l][0] + sh_K5[v][k][1] * sh_G5[v][l][1]);
                S55[k][l] = xe;
            }
        innovation_cov(S55, m.H, pv.p.r_meas, S);
        inv2(S, Si);
        for (int k = 0; k < 5; k++) { sh_H[k] = m.H[0][k]; sh_H[5 + k] = m.H[1][k]; }
        sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
        sh_nu[0] = m.z0 - m.zh0;
        sh_nu[1] = normalize_angle(m.z1 - m.zh1);
    }
    if (is_new) __syncthreads();   // uniform
    // ---- K = Sigma H^T S^-1 (:376), G = H Sigma over the prefix; pair pc; state ----
    if (tid < kAssocSlice && i < ld) {
        double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0, so = i < pv.N ? st[i] : 0.0;
        if (i < Nb) {
            double p[5], g[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int c = idx5(k, lm);
                p[k] = Sg[(size_t)i * ld + c];   // column gather (Sigma * H^T reads columns)
                g[k] = Sg[(size_t)c * ld + i];   // row gather    (H * Sigma reads rows)
            }
            for (int v = 0; v < pc; v++) {
                const double kr0 = Ub[(size_t)(2 * v) * ld + i], kr1 = Ub[(size_t)(2 * v + 1) * ld + i];
                const double gr0 = Vb[(size_t)(2 * v) * ld + i], gr1 = Vb[(size_t)(2 * v + 1) * ld + i];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    p[k] = p[k] - (kr0 * sh_G5[v][k][0] + kr1 * sh_G5[v][k][1]);
                    g[k] = g[k] - (sh_K5[v][k][0] * gr0 + sh_K5[v][k][1] * gr1);
                }
            }
            double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                sht0 += p[k] * sh_H[k];
                sht1 += p[k] * sh_H[5 + k];
                g0 += sh_H[k] * g[k];
                g1 += sh_H[5 + k] * g[k];
            }
            k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];
            k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
            double base = st[i];
            if (is_new && i == 2 * lm + 3) base = sh_t[0];   // the landmark this reading initialised (:321-322)
            if (is_new && i == 2 * lm + 4) base = sh_t[1];
            so = base + (k0 * sh_nu[0] + k1 * sh_nu[1]);   // :384
            if (i == 0) so = normalize_angle(so);            // :385
        }
        Uw[i] = k0; Uw[ld + i] = k1; Vw[i] = g0; Vw[ld + i] = g1;
        state_out[i] = so;
    }
}

void launch_assoc_meas(const PoolView& pv, double mx, double my, const AssocRec* assoc_in, AssocRec* assoc_next, int* assoc_out_j,
                       double* state_out, double* U, double* V, int* cnt_out, int pc, int Nb, int zero_upto, int m_bound,
                       double* scores, double* terms, hipStream_t s) {
    if (m_bound > 0)
        hipLaunchKernelGGL(k_assoc_score, dim3((m_bound + 63) / 64), dim3(64), 0, s, pv, mx, my, assoc_in, scores, terms, U, V, pc);
    hipLaunchKernelGGL(k_assoc_meas, dim3((pv.ld + kAssocSlice - 1) / kAssocSlice), dim3(kAssocThreads), 0, s, pv, mx, my, assoc_in,
                       assoc_next, assoc_out_j, state_out, U, V, cnt_out, pc, Nb, zero_upto, scores, terms);
}

}  // namespace ekf
