// ekf_assocfused.hip -- data_association() of a SINGLE filter beyond the LDS-resident path (ekf_slam.cpp:278-402) with
// the covariance streamed ONCE per call and ONE launch per reading: reading j's launch decides and builds the gain against
// the STORED covariance minus the call's pending rank-2 pairs, and already scores reading j + 1; one k_rank2v pass
// (ekf_callfused.hip) ends the call.  (Round 1: k_maha + k_associate_fused per reading, the second streaming all of Sigma;
// round 2: k_assoc_score + k_assoc_meas per reading.)
//
// k_assoc_score   the scores of the FIRST reading of a pass, one landmark per thread: the 25 entries of
//                 Sigma[c5(i), c5(i)] are the stored entries minus the pending pairs, in order, with the rank-2 kernel's
//                 own expression (the values the per-reading path would have read, bit for bit); summation order of
//                 innovation_cov = k_maha's shuffle folds.
// k_assoc_reading every workgroup repeats the decision (a reduction over M scores), then either builds the gain for a slice
//                 of 256 state indices or scores the NEXT reading for 256 landmarks -- see the kernel.
// Same operations in the same order as k_maha + k_assoc_decide + k_gain + k_rank2 -> decisions, state and covariance
// bit-identical (tests/test_gpu_fused.py, tests/test_gpu_callfused.py).
#include "ekf_kernels.hpp"

#include <climits>

namespace ekf {

constexpr int kAssocThreads = 256;
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

// K(r, :) and G(:, r) of the correction the workgroup has decided on, at state index r: k_gain's arithmetic on the stored
// covariance minus the call's pc pending pairs (sh_K5 / sh_G5: the pairs at the five indices c5(lm))
__device__ __forceinline__ void assoc_gain_at(const double* __restrict__ Sg, const double* __restrict__ Ub,
                                              const double* __restrict__ Vb, int ld, int r, int lm, int pc,
                                              const double (*sh_K5)[5][2], const double (*sh_G5)[5][2],
                                              const double* sh_H, const double* sh_Si, double& k0, double& k1, double& g0,
                                              double& g1) {
    double p[5], g[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int c = idx5(k, lm);
        p[k] = Sg[(size_t)r * ld + c];   // column gather (Sigma * H^T reads columns)
        g[k] = Sg[(size_t)c * ld + r];   // row gather    (H * Sigma reads rows)
    }
    // the pending factor rows of index r: pair v + 1 in flight while pair v is folded in
    double kr0 = 0.0, kr1 = 0.0, gr0 = 0.0, gr1 = 0.0;
    if (pc > 0) { kr0 = Ub[r]; kr1 = Ub[(size_t)ld + r]; gr0 = Vb[r]; gr1 = Vb[(size_t)ld + r]; }
    for (int v = 0; v < pc; v++) {
        const int vn = v + 1 < pc ? v + 1 : v;
        const double nk0 = Ub[(size_t)(2 * vn) * ld + r], nk1 = Ub[(size_t)(2 * vn + 1) * ld + r];
        const double ng0 = Vb[(size_t)(2 * vn) * ld + r], ng1 = Vb[(size_t)(2 * vn + 1) * ld + r];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            p[k] = p[k] - (kr0 * sh_G5[v][k][0] + kr1 * sh_G5[v][k][1]);
            g[k] = g[k] - (sh_K5[v][k][0] * gr0 + sh_K5[v][k][1] * gr1);
        }
        kr0 = nk0; kr1 = nk1; gr0 = ng0; gr1 = ng1;
    }
    double sht0 = 0.0, sht1 = 0.0;
    g0 = 0.0; g1 = 0.0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        sht0 += p[k] * sh_H[k];
        sht1 += p[k] * sh_H[5 + k];
        g0 += sh_H[k] * g[k];
        g1 += sh_H[5 + k] * g[k];
    }
    k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];   // :376
    k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
}

// ---------------------------------------------------------------------------------------------
// ONE launch per reading j of the call.  Every workgroup first takes the decision for reading j from the scores the
// previous launch of the call left (a reduction over M values: cheap enough to repeat in every workgroup, unlike the
// scoring itself), then plays one of two roles -- no workgroup waits for another:
//   gain   (blocks < gain_blocks, a slice of 256 state indices each): K(i, :), G(:, i) of reading j's correction,
//          appended as pair pc; state(i) += K nu, out of place (:376-385)
//   score  (the other blocks, one landmark per thread; only when a reading j + 1 follows in this pass): the Mahalanobis
//          scores of reading j + 1 (:300-309) against the covariance and state AS THEY WILL STAND after reading j's
//          correction -- the thread rebuilds what it needs of pair pc itself: K and G at its landmark's two indices (the
//          pose's three come once per workgroup), by the gain role's own function, so the values are the gain role's bit for
//          bit -- and leaves scores and correction terms for the next launch.
// scores / terms / association record / state ping-pong between launches (a fast workgroup of launch j never writes what
// a slow one still reads).  Same operations in the same order as k_maha + k_assoc_decide + k_gain + k_rank2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAssocThreads) void k_assoc_reading(PoolView pv, double mx, double my, int has_next, double mxn,
                                                                 double myn, const AssocRec* __restrict__ assoc_in,
                                                                 AssocRec* __restrict__ assoc_next, int* __restrict__ assoc_out_j,
                                                                 double* __restrict__ state_out, double* __restrict__ Uall,
                                                                 double* __restrict__ Vall, int* __restrict__ cnt_out, int pc,
                                                                 int Nb, int zero_upto, int gain_blocks,
                                                                 const double* __restrict__ scores, const double* __restrict__ terms,
                                                                 double* __restrict__ scores_out, double* __restrict__ terms_out) {
    const int tid = threadIdx.x;
    const int n = pv.n, ld = pv.ld;
    const double* __restrict__ Sg = pv.sigma;
    const double* __restrict__ st = pv.state;
    // (the factor rows < 2 pc are read-only here; rows 2 pc, 2 pc + 1 are written by the gain role: distinct rows)
    const double* Ub = Uall;
    const double* Vb = Vall;
    const bool gain_role = (int)blockIdx.x < gain_blocks;
    const bool lead = blockIdx.x == 0;
    const int i = blockIdx.x * kAssocSlice + tid;   // gain role: state index
    // last reading of a pass: the pair rows up to the streaming kernel's (rounded) correction count become exact no-ops
    if (gain_role && i < ld)
        for (int v = pc + 1; v < zero_upto; v++) {
            Uall[(size_t)(2 * v) * ld + i] = 0.0; Uall[(size_t)(2 * v + 1) * ld + i] = 0.0;
            Vall[(size_t)(2 * v) * ld + i] = 0.0; Vall[(size_t)(2 * v + 1) * ld + i] = 0.0;
        }

    __shared__ double sh_d[kAssocThreads / 64];
    __shared__ int sh_i[kAssocThreads / 64];
    __shared__ int sh_lm, sh_new, sh_Mn;
    __shared__ double sh_t[2];
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ double sh_K5[kCallV][5][2], sh_G5[kCallV][5][2];   // pending pairs at {0,1,2} and at the winner's two indices
    __shared__ double sh_Kp[3][2], sh_Gp[3][2], sh_pose[3];       // score role: pair pc and the corrected state at the pose indices

    const int M = assoc_in[0].known_count;
    const double theta = st[0], x = st[1], y = st[2];   // fresh pose, :219-221 / :331-333
    if (tid < 6 * pc) {   // pose part of the pending pairs
        const int v = tid / 6, r = (tid % 6) >> 1, h = tid & 1;
        sh_K5[v][r][h] = Ub[(size_t)(2 * v + h) * ld + r];
        sh_G5[v][r][h] = Vb[(size_t)(2 * v + h) * ld + r];
    }

    // ---- decision, :293-330, from the scores left by the previous launch (k_assoc_score for the first reading of a pass) ----
    double best = pv.p.gate_new;  // :293
    int bi = INT_MAX;
    for (int q = tid; q < M; q += kAssocThreads) {
        const double d = scores[q];
        if (d < best) { best = d; bi = q; }  // :305-309 (NaN never wins)
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
    if (tid == 0) {
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
        sh_Mn = known_count;
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
    const int Mn = sh_Mn;
    double* Uw = Uall + (size_t)(2 * pc) * ld;
    double* Vw = Vall + (size_t)(2 * pc) * ld;
    if (lm >= 0) {   // (uniform)
        // pending pairs at the winner's two indices; the winner's terms (from the thread that scored it, or built for a new one)
        if (tid < 4 * pc) {
            const int v = tid >> 2, q = (tid >> 1) & 1, h = tid & 1;
            sh_K5[v][3 + q][h] = Ub[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
            sh_G5[v][3 + q][h] = Vb[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
        }
        if (!is_new && tid >= 64 && tid < 80) {   // the winner's terms as the scoring launch left them
            const int q = tid - 64;
            const double tv = terms[(size_t)lm * 16 + q];
            if (q < 10) sh_H[q] = tv;
            else if (q < 14) sh_Si[q - 10] = tv;
            else if (q == 14) sh_nu[0] = tv;
            else sh_nu[1] = normalize_angle(tv);   // :183 (the score used it unwrapped)
        }
        __syncthreads();
        if (is_new) {   // (uniform)
            if (tid == 0) {   // :331-381 with the fresh pose: a new landmark has no score record
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
            __syncthreads();
        }
    }
    // corrected state at index r (:384-385); base: the stored state, or the position this reading has just initialised (:321-322)
    auto state_after = [&](int r, double k0, double k1) {
        double base = st[r];
        if (is_new && r == 2 * lm + 3) base = sh_t[0];
        if (is_new && r == 2 * lm + 4) base = sh_t[1];
        double so = base + (k0 * sh_nu[0] + k1 * sh_nu[1]);
        if (r == 0) so = normalize_angle(so);
        return so;
    };

    if (gain_role) {
        if (i >= ld) return;
        if (lm < 0) {   // dropped (uniform): a zero pair keeps the call's pair index = reading index; the state is carried over
            Uw[i] = 0.0; Uw[ld + i] = 0.0; Vw[i] = 0.0; Vw[ld + i] = 0.0;
            state_out[i] = st[i];
            // (a landmark initialised by a reading that is then dropped cannot occur: a new landmark gets best = 0)
            return;
        }
        // ---- K = Sigma H^T S^-1 (:376), G = H Sigma over the prefix; pair pc; state ----
        double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0, so = i < pv.N ? st[i] : 0.0;
        if (i < Nb) {
            assoc_gain_at(Sg, Ub, Vb, ld, i, lm, pc, sh_K5, sh_G5, sh_H, sh_Si, k0, k1, g0, g1);
            so = state_after(i, k0, k1);
        }
        Uw[i] = k0; Uw[ld + i] = k1; Vw[i] = g0; Vw[ld + i] = g1;
        state_out[i] = so;
        return;
    }

    // ---- score role: reading j + 1 against the filter as it stands after reading j ----
    if (!has_next) return;
    const bool corr = lm >= 0;   // (uniform) reading j appends a non-zero pair
    if (tid < 3) {               // pair pc and the corrected state at the pose indices: once per workgroup
        double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0, so = st[tid];
        if (corr) {
            assoc_gain_at(Sg, Ub, Vb, ld, tid, lm, pc, sh_K5, sh_G5, sh_H, sh_Si, k0, k1, g0, g1);
            so = state_after(tid, k0, k1);
        }
        sh_Kp[tid][0] = k0; sh_Kp[tid][1] = k1; sh_Gp[tid][0] = g0; sh_Gp[tid][1] = g1;
        sh_pose[tid] = so;
    }
    __syncthreads();
    const int li = (blockIdx.x - gain_blocks) * kAssocThreads + tid;   // landmark scored by this thread
    if (li >= Mn || li >= n) return;
    // pair pc and the corrected position at the landmark's two indices
    double kq[5][2], gq[5][2], pos[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { kq[k][0] = sh_Kp[k][0]; kq[k][1] = sh_Kp[k][1]; gq[k][0] = sh_Gp[k][0]; gq[k][1] = sh_Gp[k][1]; }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int r = 3 + 2 * li + q;
        double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0, so = st[r];
        if (corr) {
            assoc_gain_at(Sg, Ub, Vb, ld, r, lm, pc, sh_K5, sh_G5, sh_H, sh_Si, k0, k1, g0, g1);
            so = state_after(r, k0, k1);
        }
        kq[3 + q][0] = k0; kq[3 + q][1] = k1; gq[3 + q][0] = g0; gq[3 + q][1] = g1;
        pos[q] = so;
    }
    MeasTerms m;
    measurement_terms(pos[0], pos[1], mxn, myn, sh_pose[0], sh_pose[1], sh_pose[2], m);   // fresh pose, :219-221
    double S55[5][5], S[2][2], Si[2][2];
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
        for (int l = 0; l < 5; l++) S55[k][l] = Sg[(size_t)idx5(k, li) * ld + idx5(l, li)];
    for (int v = 0; v < pc; v++) {   // ... as they stand NOW: minus the pending pairs of the call, in order
        double kr[5][2], gc[5][2];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int c = idx5(k, li);
            kr[k][0] = Ub[(size_t)(2 * v) * ld + c]; kr[k][1] = Ub[(size_t)(2 * v + 1) * ld + c];
            gc[k][0] = Vb[(size_t)(2 * v) * ld + c]; gc[k][1] = Vb[(size_t)(2 * v + 1) * ld + c];
        }
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
    }
    if (corr) {   // ... and minus reading j's own pair
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kq[k][0] * gq[l][0] + kq[k][1] * gq[l][1]);
    }
    innovation_cov(S55, m.H, pv.p.r_meas, S);   // sums in the order of k_maha's shuffle folds
    inv2(S, Si);
    const double v0 = m.z0 - m.zh0, v1 = m.z1 - m.zh1;   // bearing NOT wrapped, :269
    const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
    const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
    scores_out[li] = t0 * v0 + t1 * v1;
    double* tr = terms_out + (size_t)li * 16;
#pragma unroll
    for (int k = 0; k < 5; k++) { tr[k] = m.H[0][k]; tr[5 + k] = m.H[1][k]; }
    tr[10] = Si[0][0]; tr[11] = Si[0][1]; tr[12] = Si[1][0]; tr[13] = Si[1][1];
    tr[14] = v0; tr[15] = v1;
}

void launch_assoc_score(const PoolView& pv, double mx, double my, const AssocRec* assoc_in, const double* U, const double* V,
                        int pc, int m_bound, double* scores, double* terms, hipStream_t s) {
    if (m_bound > 0)
        hipLaunchKernelGGL(k_assoc_score, dim3((m_bound + 63) / 64), dim3(64), 0, s, pv, mx, my, assoc_in, scores, terms, U, V, pc);
}

void launch_assoc_reading(const PoolView& pv, double mx, double my, int has_next, double mxn, double myn,
                          const AssocRec* assoc_in, AssocRec* assoc_next, int* assoc_out_j, double* state_out, double* U,
                          double* V, int* cnt_out, int pc, int Nb, int zero_upto, int m_bound_next, const double* scores,
                          const double* terms, double* scores_out, double* terms_out, hipStream_t s) {
    const int gain_blocks = (pv.ld + kAssocSlice - 1) / kAssocSlice;
    const int score_blocks = has_next ? (m_bound_next + kAssocThreads - 1) / kAssocThreads : 0;
    hipLaunchKernelGGL(k_assoc_reading, dim3(gain_blocks + score_blocks), dim3(kAssocThreads), 0, s, pv, mx, my, has_next, mxn,
                       myn, assoc_in, assoc_next, assoc_out_j, state_out, U, V, cnt_out, pc, Nb, zero_upto, gain_blocks, scores,
                       terms, scores_out, terms_out);
}

}  // namespace ekf
