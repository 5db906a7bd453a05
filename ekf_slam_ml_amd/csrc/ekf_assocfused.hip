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

constexpr int kAssocThreads = 256;     // k_assoc_reading: four wavefronts with different roles ...
constexpr int kAssocLandmarks = 64;    // ... around 64 landmarks (small workgroups spread the gathers over many CUs)

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
    // correction terms of every scored landmark, [16][n]: consecutive landmarks are consecutive addresses
    double* tr = terms + i;
    const size_t ts = (size_t)pv.n;
#pragma unroll
    for (int k = 0; k < 5; k++) { tr[k * ts] = m.H[0][k]; tr[(5 + k) * ts] = m.H[1][k]; }
    tr[10 * ts] = Si[0][0]; tr[11 * ts] = Si[0][1]; tr[12 * ts] = Si[1][0]; tr[13 * ts] = Si[1][1];
    tr[14 * ts] = v0; tr[15 * ts] = v1;
}

// K(r, :) and G(:, r) of the correction the workgroup has decided on, at state index r: k_gain's arithmetic on the stored
// covariance minus the call's pc pending pairs.  sh_K5 / sh_G5: the pairs at the five indices c5(lm); uv[v] = the pairs'
// own values at index r, {U[2v](r), U[2v+1](r), V[2v](r), V[2v+1](r)} (loaded by the caller long before: no memory
// round trip inside).  p / g: Sigma(r, c5(lm)) and Sigma(c5(lm), r) as stored.
__device__ __forceinline__ void assoc_gain_at(double (&p)[5], double (&g)[5], const double (*uv)[4], int pc,
                                              const double (*sh_K5)[5][2], const double (*sh_G5)[5][2],
                                              const double* sh_H, const double* sh_Si, double& k0, double& k1, double& g0,
                                              double& g1) {
#pragma unroll
    for (int v = 0; v < kCallV - 1; v++)
        if (v < pc) {   // (uniform)
#pragma unroll
            for (int k = 0; k < 5; k++) {
                p[k] = p[k] - (uv[v][0] * sh_G5[v][k][0] + uv[v][1] * sh_G5[v][k][1]);
                g[k] = g[k] - (sh_K5[v][k][0] * uv[v][2] + sh_K5[v][k][1] * uv[v][3]);
            }
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
// ONE launch per reading j of the call; no workgroup waits for another.  A workgroup of four wavefronts owns 64 landmarks
// and their 128 state indices (workgroup 0 also the pose); lane l of every wave belongs to landmark li = 64 b + l:
//   wave A   state index 3 + 2 li: gain of reading j's correction there
//   wave B   state index 4 + 2 li: the gain there, and the landmark's score for reading j + 1
//   wave P   lanes 0..2: the pose indices 0..2; lane 3: the next reading in polar form, once per workgroup
//   wave S   scans the scores of reading j for the decision (with the others' help)
// 1. everything that does not depend on the decision is requested at once: the pending pairs' values at the owned
//    indices, the state entries, the scores; wave B also the landmark's 5 x 5 block of Sigma
// 2. the decision for reading j, :293-330, from the scores the previous launch left (a reduction over M values: cheap
//    enough to repeat in every workgroup, unlike the scoring itself)
// 3. gain: K(r, :), G(:, r) of reading j's correction at the owned index -> pair pc of the call, state(r) += K nu out of
//    place (:376-385).  While the gathers of Sigma(r, c5) / Sigma(c5, r) are in flight, wave B brings its block up to date
//    with the call's pending pairs.  Waves A and P publish their gains to the workgroup.
// 4. when a reading j + 1 follows in this pass: wave B folds pair pc into its block (its own values + wave A's + the pose's)
//    and scores the landmark (:300-309) against the covariance and state AS THEY STAND after reading j's correction; score
//    and correction terms are left for the next launch.
// scores / terms / association record / state ping-pong between launches (a fast workgroup of launch j never writes what
// a slow one still reads).  Same operations in the same order as k_maha + k_assoc_decide + k_gain + k_rank2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAssocThreads) void k_assoc_reading(
    PoolView pv, double mx, double my, int has_next, double mxn, double myn, const AssocRec* __restrict__ assoc_in,
    AssocRec* __restrict__ assoc_next, int* __restrict__ assoc_out_j, double* __restrict__ state_out, double* __restrict__ Uall,
    double* __restrict__ Vall, int* __restrict__ cnt_out, int pc, int Nb, int zero_upto, int m_bound,
    const double* __restrict__ scores, const double* __restrict__ terms, double* __restrict__ scores_out,
    double* __restrict__ terms_out, long long* __restrict__ trace) {
    const int tid = threadIdx.x;
    const int n = pv.n, N = pv.N, ld = pv.ld;
    // diagnostics (ekf_phase_trace): lane 0 of wave B of workgroup 0 stamps the 100 MHz wall clock, 16 slots per reading
#define AR_TR(k) do { if (trace && blockIdx.x == 0 && tid == 64) trace[pc * 16 + (k)] = wall_clock64(); } while (0)
    AR_TR(0);
    const double* __restrict__ Sg = pv.sigma;
    const double* __restrict__ st = pv.state;
    // (the factor rows < 2 pc are read-only here; rows 2 pc .. are written: distinct rows of the same buffers)
    const double* Ub = Uall;
    const double* Vb = Vall;
    const bool lead = blockIdx.x == 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool wA = wave == 0, wB = wave == 1, wP = wave == 2;
    const int li = blockIdx.x * kAssocLandmarks + lane;   // waves A, B: landmark
    // the state index this thread owns (none: -1)
    const int r = wA ? 3 + 2 * li : wB ? 4 + 2 * li : (wP && lane < 3) ? lane : -1;
    const bool own = r >= 0 && r < ld && (!wP || lead);   // ... and writes (the pose is written by workgroup 0 only)
    const bool gains = r >= 0 && r < Nb;                  // ... and has a gain to build (inside the active prefix)
    const int rl = r >= 0 && r < N ? r : 0;               // clamped for loads
    const int ra = wB ? (3 + 2 * li < N ? 3 + 2 * li : 0) : 0;   // wave B: its partner's index (clamped)

    __shared__ double sh_d[kAssocThreads / 64];
    __shared__ int sh_i[kAssocThreads / 64];
    __shared__ int sh_lm, sh_new, sh_Mn;
    __shared__ double sh_t[2];
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ double sh_K5[kCallV][5][2], sh_G5[kCallV][5][2];   // pending pairs at {0,1,2} and at the winner's two indices
    __shared__ double sh_Kp[3][2], sh_Gp[3][2], sh_pose[3];       // pair pc and the corrected state at the pose indices
    __shared__ double sh_KA[kAssocLandmarks][2], sh_GA[kAssocLandmarks][2], sh_soA[kAssocLandmarks];   // ... at wave A's indices
    __shared__ double sh_z[2];                                    // (r, phi) of the next reading, :142-146

    // ---- 1. requests that do not depend on the decision ----
    const int M = assoc_in[0].known_count;
    const double theta = st[0], x = st[1], y = st[2];   // fresh pose, :219-221 / :331-333
    double uv[kCallV - 1][4];                           // pending pairs at the owned index
    double ua[kCallV - 1][4];                           // wave B: ... and at its partner's (for the block update)
#pragma unroll
    for (int v = 0; v < kCallV - 1; v++) {
        const int vc = v < pc ? v : 0;
        uv[v][0] = Ub[(size_t)(2 * vc) * ld + rl]; uv[v][1] = Ub[(size_t)(2 * vc + 1) * ld + rl];
        uv[v][2] = Vb[(size_t)(2 * vc) * ld + rl]; uv[v][3] = Vb[(size_t)(2 * vc + 1) * ld + rl];
    }
    const double st_r = st[rl];
    const bool scorer = wB && has_next && li < n;   // (whether li < known count after the decision is seen later)
    double S55[5][5];
    if (scorer) {
#pragma unroll
        for (int v = 0; v < kCallV - 1; v++) {
            const int vc = v < pc ? v : 0;
            ua[v][0] = Ub[(size_t)(2 * vc) * ld + ra]; ua[v][1] = Ub[(size_t)(2 * vc + 1) * ld + ra];
            ua[v][2] = Vb[(size_t)(2 * vc) * ld + ra]; ua[v][3] = Vb[(size_t)(2 * vc + 1) * ld + ra];
        }
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int l = 0; l < 5; l++) S55[k][l] = Sg[(size_t)idx5(k, li) * ld + idx5(l, li)];
    }
    if (tid < 6 * pc) {   // pose part of the pending pairs
        const int v = tid / 6, q = (tid % 6) >> 1, h = tid & 1;
        sh_K5[v][q][h] = Ub[(size_t)(2 * v + h) * ld + q];
        sh_G5[v][q][h] = Vb[(size_t)(2 * v + h) * ld + q];
    }
    double best = pv.p.gate_new;  // :293
    int bi = INT_MAX;
    // the scan of the scores, every thread of the workgroup, four loads in flight per thread (m_bound >= M is the host's
    // bound: the loads wait neither for M nor for each other)
    for (int q0 = tid; q0 < m_bound; q0 += 4 * kAssocThreads) {
        double d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) d[u] = scores[min(q0 + u * kAssocThreads, m_bound - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u * kAssocThreads;
            if (q < M && d[u] < best) { best = d[u]; bi = q; }  // :305-309 (NaN never wins)
        }
    }
    // last reading of a pass: the pair rows up to the streaming kernel's (rounded) correction count become exact no-ops
    if (own)
        for (int v = pc + 1; v < zero_upto; v++) {
            Uall[(size_t)(2 * v) * ld + r] = 0.0; Uall[(size_t)(2 * v + 1) * ld + r] = 0.0;
            Vall[(size_t)(2 * v) * ld + r] = 0.0; Vall[(size_t)(2 * v + 1) * ld + r] = 0.0;
        }
    if (wP && lane == 3 && has_next) {   // the next reading in polar form, once per workgroup
        sh_z[0] = sqrt(mxn * mxn + myn * myn);
        sh_z[1] = atan2(myn, mxn);
    }
    AR_TR(1);

    // ---- 2. decision, :293-330 ----
    double rd = best;
    int ri = bi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(rd, off, kWave);
        const int oi = __shfl_down(ri, off, kWave);
        if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
    }
    if (lane == 0) { sh_d[wave] = rd; sh_i[wave] = ri; }
    __syncthreads();
    AR_TR(2);
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
    AR_TR(3);
    const int lm = sh_lm;
    const int is_new = sh_new;
    const int Mn = sh_Mn;
    const bool corr = lm >= 0;   // (uniform) reading j appends a non-zero pair
    double* Uw = Uall + (size_t)(2 * pc) * ld;
    double* Vw = Vall + (size_t)(2 * pc) * ld;

    // the landmark's block as it stands NOW, for the next reading's score: minus the pending pairs of the call, in order
    // (wave B; the pose values of the pairs are read straight from the factor rows: wave-uniform addresses).  Placed
    // behind the requests of step 3 so that it runs while they are in flight.
    auto pending_block = [&]() {
        if (scorer) {
#pragma unroll
            for (int v = 0; v < kCallV - 1; v++)
                if (v < pc) {   // (uniform)
                    double kr[5][2], gc[5][2];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        kr[k][0] = Ub[(size_t)(2 * v) * ld + k]; kr[k][1] = Ub[(size_t)(2 * v + 1) * ld + k];
                        gc[k][0] = Vb[(size_t)(2 * v) * ld + k]; gc[k][1] = Vb[(size_t)(2 * v + 1) * ld + k];
                    }
                    kr[3][0] = ua[v][0]; kr[3][1] = ua[v][1]; gc[3][0] = ua[v][2]; gc[3][1] = ua[v][3];
                    kr[4][0] = uv[v][0]; kr[4][1] = uv[v][1]; gc[4][0] = uv[v][2]; gc[4][1] = uv[v][3];
#pragma unroll
                    for (int k = 0; k < 5; k++)
#pragma unroll
                        for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
                }
        }
    };

    // ---- 3. gain: pair pc and the state at the owned index ----
    double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0;
    double so = r >= 0 && r < N ? st_r : 0.0;
    if (corr) {   // (uniform)
        // Sigma(r, c5(lm)) and Sigma(c5(lm), r): requested before anything else waits
        double p[5], g[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int c = idx5(k, lm);
            p[k] = Sg[(size_t)rl * ld + c];   // column gather (Sigma * H^T reads columns)
            g[k] = Sg[(size_t)c * ld + rl];   // row gather    (H * Sigma reads rows)
        }
        // pending pairs at the winner's two indices; the winner's terms (from the thread that scored it, or built for a new
        // one): requested into registers, stored to LDS behind the block update, which runs while they are in flight
        double wk = 0.0, wg = 0.0, tv = 0.0;
        if (tid < 4 * pc) {
            const int v = tid >> 2, q = (tid >> 1) & 1, h = tid & 1;
            wk = Ub[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
            wg = Vb[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
        }
        if (!is_new && wP && lane >= 16 && lane < 32) tv = terms[(size_t)(lane - 16) * n + lm];   // as the scoring launch left them
        AR_TR(4);
        pending_block();
        AR_TR(5);
        if (tid < 4 * pc) {
            const int v = tid >> 2, q = (tid >> 1) & 1, h = tid & 1;
            sh_K5[v][3 + q][h] = wk;
            sh_G5[v][3 + q][h] = wg;
        }
        if (!is_new && wP && lane >= 16 && lane < 32) {
            const int q = lane - 16;
            if (q < 10) sh_H[q] = tv;
            else if (q < 14) sh_Si[q - 10] = tv;
            else if (q == 14) sh_nu[0] = tv;
            else sh_nu[1] = normalize_angle(tv);   // :183 (the score used it unwrapped)
        }
        __syncthreads();
        AR_TR(6);
        if (is_new) {   // (uniform)
            if (tid == 0) {   // :331-381 with the fresh pose: a new landmark has no score record
                MeasTerms m;
                measurement_terms(sh_t[0], sh_t[1], mx, my, theta, x, y, m);
                double B55[5][5], S[2][2], Si[2][2];
                for (int k = 0; k < 5; k++)
                    for (int l = 0; l < 5; l++) {
                        double xe = Sg[(size_t)idx5(k, lm) * ld + idx5(l, lm)];
                        for (int v = 0; v < pc; v++) xe = xe - (sh_K5[v][k][0] * sh_G5[v][l][0] + sh_K5[v][k][1] * sh_G5[v][l][1]);
                        B55[k][l] = xe;
                    }
                innovation_cov(B55, m.H, pv.p.r_meas, S);
                inv2(S, Si);
                for (int k = 0; k < 5; k++) { sh_H[k] = m.H[0][k]; sh_H[5 + k] = m.H[1][k]; }
                sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
                sh_nu[0] = m.z0 - m.zh0;
                sh_nu[1] = normalize_angle(m.z1 - m.zh1);
            }
            __syncthreads();
        }
        if (gains) {
            assoc_gain_at(p, g, uv, pc, sh_K5, sh_G5, sh_H, sh_Si, k0, k1, g0, g1);
            // corrected state (:384-385); base: the stored state, or the position this reading has just initialised (:321-322)
            double base = st_r;
            if (is_new && r == 2 * lm + 3) base = sh_t[0];
            if (is_new && r == 2 * lm + 4) base = sh_t[1];
            so = base + (k0 * sh_nu[0] + k1 * sh_nu[1]);
            if (r == 0) so = normalize_angle(so);
        }
    } else {
        pending_block();
    }
    AR_TR(7);
    // (dropped reading: a zero pair keeps the call's pair index = reading index; the state is carried over.  A landmark
    // initialised by a reading that is then dropped cannot occur: a new landmark gets best = 0)
    if (own) {
        Uw[r] = k0; Uw[ld + r] = k1; Vw[r] = g0; Vw[ld + r] = g1;
        state_out[r] = so;
    }
    if (!has_next) return;   // (uniform)

    // ---- 4. score of reading j + 1 against the filter as it stands after reading j ----
    if (wA) { sh_KA[lane][0] = k0; sh_KA[lane][1] = k1; sh_GA[lane][0] = g0; sh_GA[lane][1] = g1; sh_soA[lane] = so; }
    if (wP && lane < 3) { sh_Kp[lane][0] = k0; sh_Kp[lane][1] = k1; sh_Gp[lane][0] = g0; sh_Gp[lane][1] = g1; sh_pose[lane] = so; }
    __syncthreads();
    AR_TR(8);
    if (!scorer || li >= Mn) return;
    if (corr) {   // ... minus reading j's own pair
        double kr[5][2], gc[5][2];
#pragma unroll
        for (int k = 0; k < 3; k++) { kr[k][0] = sh_Kp[k][0]; kr[k][1] = sh_Kp[k][1]; gc[k][0] = sh_Gp[k][0]; gc[k][1] = sh_Gp[k][1]; }
        kr[3][0] = sh_KA[lane][0]; kr[3][1] = sh_KA[lane][1]; gc[3][0] = sh_GA[lane][0]; gc[3][1] = sh_GA[lane][1];
        kr[4][0] = k0; kr[4][1] = k1; gc[4][0] = g0; gc[4][1] = g1;
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
    }
    AR_TR(9);
    MeasTerms m;
    m.z0 = sh_z[0]; m.z1 = sh_z[1];
    predicted_terms(sh_soA[lane], so, sh_pose[0], sh_pose[1], sh_pose[2], m);   // fresh pose, :219-221
    AR_TR(10);
    double S[2][2], Si[2][2];
    innovation_cov(S55, m.H, pv.p.r_meas, S);   // sums in the order of k_maha's shuffle folds
    inv2(S, Si);
    const double v0 = m.z0 - m.zh0, v1 = m.z1 - m.zh1;   // bearing NOT wrapped, :269
    const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
    const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
    scores_out[li] = t0 * v0 + t1 * v1;
    double* tr = terms_out + li;   // [16][n]
    const size_t ts = (size_t)n;
#pragma unroll
    for (int k = 0; k < 5; k++) { tr[k * ts] = m.H[0][k]; tr[(5 + k) * ts] = m.H[1][k]; }
    tr[10 * ts] = Si[0][0]; tr[11 * ts] = Si[0][1]; tr[12 * ts] = Si[1][0]; tr[13 * ts] = Si[1][1];
    tr[14 * ts] = v0; tr[15 * ts] = v1;
    AR_TR(11);
#undef AR_TR
}

void launch_assoc_score(const PoolView& pv, double mx, double my, const AssocRec* assoc_in, const double* U, const double* V,
                        int pc, int m_bound, double* scores, double* terms, hipStream_t s) {
    if (m_bound > 0)
        hipLaunchKernelGGL(k_assoc_score, dim3((m_bound + 63) / 64), dim3(64), 0, s, pv, mx, my, assoc_in, scores, terms, U, V, pc);
}

void launch_assoc_reading(const PoolView& pv, double mx, double my, int has_next, double mxn, double myn,
                          const AssocRec* assoc_in, AssocRec* assoc_next, int* assoc_out_j, double* state_out, double* U,
                          double* V, int* cnt_out, int pc, int Nb, int zero_upto, int m_bound, const double* scores,
                          const double* terms, double* scores_out, double* terms_out, hipStream_t s, long long* trace) {
    const int landmarks = (pv.ld - 3 + 1) / 2;   // owners of every index of the padded row beyond the pose
    hipLaunchKernelGGL(k_assoc_reading, dim3((landmarks + kAssocLandmarks - 1) / kAssocLandmarks), dim3(kAssocThreads),
                       0, s, pv, mx, my, has_next, mxn, myn, assoc_in, assoc_next, assoc_out_j, state_out, U, V, cnt_out, pc, Nb,
                       zero_upto, m_bound, scores, terms, scores_out, terms_out, trace);
}

}  // namespace ekf
