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
// bit-identical (tests/test_gpu_forms.py, tests/test_gpu_callfused.py).
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
                                                    const double* __restrict__ Vb, int pc, double* __restrict__ blk) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ld = pv.ld;
    const double* __restrict__ Sg = pv.sigma;
    const double* __restrict__ st = pv.state;
    const int M = assoc_in[0].known_count;
    if (i >= pv.n || (i >= M && !blk)) return;
    double S55[5][5], S[2][2], Si[2][2];
#pragma unroll
    for (int k = 0; k < 5; k++)
#pragma unroll
        for (int l = 0; l < 5; l++) S55[k][l] = Sg[(size_t)idx5(k, i) * ld + idx5(l, i)];
    // blk ([25][n], the reading launches' block cache): every landmark's block as stored -- the launches of the pass keep
    // it current by folding each new pair in (undiscovered landmarks too: their K = G = 0 leaves their part untouched)
    if (blk && pc == 0) {
#pragma unroll
        for (int e = 0; e < 25; e++) blk[(size_t)e * pv.n + i] = S55[e / 5][e % 5];
    }
    if (i >= M) return;
    MeasTerms m;
    measurement_terms(st[2 * i + 3], st[2 * i + 4], mx, my, st[0], st[1], st[2], m);   // fresh pose, :219-221
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
    double* __restrict__ terms_out, long long* __restrict__ trace, double* __restrict__ blk, char* pub_host, int pub_j,
    unsigned pub_seq) {
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
    if (scorer && blk) {   // the block as the previous launch (or the scoring launch) left it: current, coalesced
#pragma unroll
        for (int e = 0; e < 25; e++) S55[e / 5][e % 5] = blk[(size_t)e * n + li];
    } else if (scorer) {
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
            if (pub_host) {   // the host's copy (see k_assoc_call): behind the call's last decision the record and the number
                reinterpret_cast<int*>(pub_host + kAssocDecOff)[pub_j] = a.lm;
                if (pub_seq) {
                    *reinterpret_cast<AssocRec*>(pub_host) = a;
                    __threadfence_system();
                    *reinterpret_cast<volatile unsigned*>(pub_host + kAssocSeqOff) = pub_seq;
                }
            }
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
        if (scorer && !blk) {   // (with the block cache there is nothing to bring up to date)
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
        gather_row5(Sg + (size_t)rl * ld, lm, p);   // column gather (Sigma * H^T reads columns): three loads
#pragma unroll
        for (int k = 0; k < 5; k++) g[k] = Sg[(size_t)idx5(k, lm) * ld + rl];   // row gather (H * Sigma reads rows)
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
    if (!scorer || (li >= Mn && !blk)) return;
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
        if (blk) {   // (thread-private entries: the next launch's loads follow this launch in stream order)
#pragma unroll
            for (int e = 0; e < 25; e++) blk[(size_t)e * n + li] = S55[e / 5][e % 5];
        }
    }
    if (li >= Mn) return;
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

// ---------------------------------------------------------------------------------------------
// A WHOLE data_association() call (<= kCallV readings) of a single filter in ONE launch, while the discovered part of
// the map fits one workgroup: thread t carries landmark t -- its 5 x 5 block of Sigma, its position and its two state
// indices -- in registers from the first reading of the call to the last; a helper wave carries the pose.  Per reading:
// score (no memory access: the block is kept CURRENT by folding each new pair into it, instead of re-gathering it and
// re-applying every pending pair), reduction + decision, one round trip for Sigma(r, c5(lm)) / Sigma(c5(lm), r), gain,
// pair appended, block updated.  The pass over Sigma (k_rank2v) follows as a second launch: two launches per CALL where the
// per-reading forms take two per READING.  Same operations in the same order as the other forms (a landmark that is not
// discovered yet has K = G = 0 exactly, so folding the pairs into its block from the start leaves it bit-identical).
// ---------------------------------------------------------------------------------------------
constexpr int kCallLandmarks = 448;   // landmarks one workgroup carries: 7 wavefronts + the helper wave = 512 threads

// THREADS: 256 (up to 192 landmarks: one wave per SIMD, the whole register file -- four pending pairs' values requested with
// the gathers) or 512 (up to 448: requested behind them, four at a time)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_assoc_call(PoolView pv, AssocCallArgs a, int* __restrict__ assoc_out,
                                                                    double* __restrict__ Uall, double* __restrict__ Vall,
                                                                    int* __restrict__ cnt_out, int zero_upto,
                                                                    long long* __restrict__ trace, char* pub_host,
                                                                    int pub_j0, unsigned pub_seq) {
    const int tid = threadIdx.x, lane = tid & 63;
    // diagnostics (ekf_phase_trace): thread 0 stamps the 100 MHz wall clock, 16 slots per reading (slot 15 of reading 0: start)
#define AC_TR(j, k) do { if (trace && tid == 0) trace[(j) * 16 + (k)] = wall_clock64(); } while (0)
    AC_TR(0, 15);
    const int nmain = (int)blockDim.x - 64;
    const bool helper = tid >= nmain;
    const int hl = tid - nmain;
    const int n = pv.n, N = pv.N, ld = pv.ld;
    const double* __restrict__ Sg = pv.sigma;
    double* st = pv.state;   // (updated in place: one workgroup, ordered by its barriers)
    double* Ub = Uall;
    double* Vb = Vall;
    const int J = a.J;
    // the state indices this thread owns: main thread t -> 3 + 2t, 4 + 2t; helper lanes 0..2 -> the pose
    const int t = tid;
    const bool has_lm = !helper && t < n;
    const int nown = helper ? (hl < 3 ? 1 : 0) : (has_lm ? 2 : 0);
    const int r0 = helper ? (hl < 3 ? hl : 0) : (has_lm ? 3 + 2 * t : 0);

    constexpr int kPre = THREADS <= 256 ? kCallV - 1 : 0;
    constexpr int kBatch = THREADS <= 256 ? 4 : 2;
    __shared__ double sh_d[THREADS / 64];
    __shared__ int sh_i[THREADS / 64];
    __shared__ int sh_M, sh_lm, sh_new;
    __shared__ double sh_t[2];
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ double sh_K5[kCallV][5][2], sh_G5[kCallV][5][2];
    __shared__ double sh_pose[3], sh_z[kCallV][2];
    __shared__ double sh_terms[THREADS - 64][17];   // H, S^-1, nu of every scored landmark (the winner's are needed)

    // ---- the call's inputs: block, position, pose, the readings in polar form (one lane each) ----
    double S55[5][5], pos[2] = {0.0, 0.0};
    if (has_lm) {
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int l = 0; l < 5; l++) S55[k][l] = Sg[(size_t)idx5(k, t) * ld + idx5(l, t)];
        pos[0] = st[r0]; pos[1] = st[r0 + 1];
    }
    // Sigma(r, 0..2) and Sigma(0..2, r) of the owned indices: the stored covariance does not change during the call (the pairs
    // stay pending), and every correction gathers the pose's three rows and columns -- fetched once, kept in registers
    const int rb = nown > 0 ? r0 : 0;            // (threads without an index gather from row / column 0: never used)
    const int rb1 = nown > 1 ? r0 + 1 : rb;
    // (the 512-thread form has no registers to spare for it and gathers them per reading)
    constexpr bool kPosePanel = THREADS <= 256;
    double pp[2][3], gp[2][3];
    auto pose_panel = [&]() {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double* row = Sg + (size_t)(q ? rb1 : rb) * ld;
            const D2u c01 = *reinterpret_cast<const D2u*>(row);
            pp[q][0] = c01.x; pp[q][1] = c01.y; pp[q][2] = row[2];
        }
#pragma unroll
        for (int k3 = 0; k3 < 3; k3++) {
            const D2u gg = *reinterpret_cast<const D2u*>(Sg + (size_t)k3 * ld + rb);
            gp[0][k3] = gg.x; gp[1][k3] = gg.y;
        }
    };
    if constexpr (kPosePanel) pose_panel();
    if (helper && hl < 3) { pos[0] = st[hl]; sh_pose[hl] = pos[0]; }
    if (helper && hl >= 8 && hl < 8 + J) {
        const double mx = a.xy[hl - 8][0], my = a.xy[hl - 8][1];
        sh_z[hl - 8][0] = sqrt(mx * mx + my * my);   // :142-146
        sh_z[hl - 8][1] = atan2(my, mx);
    }
    if (tid == 0) { sh_M = pv.assoc[0].known_count; sh_lm = -1; }
    __syncthreads();

    int pc = 0;   // pairs appended so far (uniform)
    for (int j = 0; j < J; j++) {   // :291 sequential, state-carrying
        const double mx = a.xy[j][0], my = a.xy[j][1];
        const int M = sh_M;
        const double theta = sh_pose[0], x = sh_pose[1], y = sh_pose[2];   // fresh pose per reading, :219-221 / :331-333
        AC_TR(j, 0);
        // ---- score, :300-309: this thread's landmark, everything in registers ----
        double best = pv.p.gate_new;  // :293
        int bi = INT_MAX;
        if (has_lm && t < M) {
            MeasTerms m;
            m.z0 = sh_z[j][0]; m.z1 = sh_z[j][1];
            predicted_terms(pos[0], pos[1], theta, x, y, m);
            double S[2][2], Si[2][2];
            innovation_cov(S55, m.H, pv.p.r_meas, S);   // sums in the order of k_maha's shuffle folds
            inv2(S, Si);
            const double v0 = m.z0 - m.zh0, v1 = m.z1 - m.zh1;   // bearing NOT wrapped, :269
            const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
            const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
            const double sc = t0 * v0 + t1 * v1;
            if (sc < best) { best = sc; bi = t; }   // :305-309 (NaN never wins)
            double* tr = sh_terms[t];
#pragma unroll
            for (int k = 0; k < 5; k++) { tr[k] = m.H[0][k]; tr[5 + k] = m.H[1][k]; }
            tr[10] = Si[0][0]; tr[11] = Si[0][1]; tr[12] = Si[1][0]; tr[13] = Si[1][1];
            tr[14] = v0; tr[15] = v1;
        }
        AC_TR(j, 1);
        // lexicographic (d, i) minimum = the sequential scan's first strict minimum
        double rd = best;
        int ri = bi;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_down(rd, off, kWave);
            const int oi = __shfl_down(ri, off, kWave);
            if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
        }
        if (lane == 0) { sh_d[tid >> 6] = rd; sh_i[tid >> 6] = ri; }
        __syncthreads();
        AC_TR(j, 2);
        if (tid == 0) {   // :293-330
            for (int w = 1; w < (int)blockDim.x / 64; w++)
                if (sh_d[w] < rd || (sh_d[w] == rd && sh_i[w] < ri)) { rd = sh_d[w]; ri = sh_i[w]; }
            const int idx = (ri == INT_MAX) ? M : ri;   // :294 min_maha_idx = known_count
            int Mn = M, is_new = 0;
            if (idx == M && idx < n) {                  // :318-327 new landmark
                const double rr = sqrt(mx * mx + my * my);
                const double phi = atan2(my, mx);
                sh_t[0] = x + rr * cos(phi + theta);
                sh_t[1] = y + rr * sin(phi + theta);
                Mn = M + 1;
                rd = 0.0;
                is_new = 1;
            }
            const int active = (rd < pv.p.gate_update) && idx < n;   // :330
            sh_M = Mn;
            sh_lm = active ? idx : -1;
            sh_new = is_new;
            assoc_out[j] = active ? idx : -1;
            // The host's copy (mapped memory, see k_publish_assoc): the decision now, and behind the call's LAST decision the
            // record and the sequence number -- the host goes on while this launch builds the last gain.
            if (pub_host) {
                reinterpret_cast<int*>(pub_host + kAssocDecOff)[pub_j0 + j] = active ? idx : -1;
                if (pub_seq && j == a.J - 1) {
                    AssocRec rec;
                    rec.known_count = Mn; rec.lm = active ? idx : -1; rec.active = active; rec.pad = 0; rec.best = 0.0;
                    *reinterpret_cast<AssocRec*>(pub_host) = rec;
                    __threadfence_system();
                    *reinterpret_cast<volatile unsigned*>(pub_host + kAssocSeqOff) = pub_seq;
                }
            }
        }
        __syncthreads();
        AC_TR(j, 3);
        const int lm = sh_lm;
        const int is_new = sh_new;
        if (lm < 0) {   // dropped (uniform); the barrier keeps sh_lm stable for slow waves
            // a landmark that was initialised and then dropped (gate_update <= 0: 0.0 < gate is false) keeps the position
            // of :321-322 all the same -- its thread takes it now, the correction below never runs
            if (is_new && has_lm && t == sh_M - 1) { pos[0] = sh_t[0]; pos[1] = sh_t[1]; }
            __syncthreads();
            continue;
        }
        // ---- requests: Sigma(r, c5(lm)) and Sigma(c5(lm), r) for the owned indices; the pairs' values at the winner ----
        // the active dimension of this reading (landmarks are appended in discovery order, :318-327): K, G are exact zeros beyond
        int mact = a.known_count + j + 1 < n ? a.known_count + j + 1 : n;
        if (a.touched_hwm > mact) mact = a.touched_hwm;
        const int Nb = a.active_prefix ? 3 + 2 * mact : N;
        // Wide accesses: a thread's two indices are neighbours, and so are the columns {3 + 2 lm, 4 + 2 lm} of a row --
        // 16-byte loads (8-byte aligned) fetch two entries each.  A column gather costs the memory pipe one request per
        // lane and instruction whatever its width: 4 instructions per reading here instead of 20.
        double p[2][5], g[2][5];
        if constexpr (!kPosePanel) pose_panel();
        {
            const size_t cl = 3 + 2 * (size_t)lm;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const double* row = Sg + (size_t)(q ? rb1 : rb) * ld;   // Sigma(r, c5(lm)): column gather (Sigma * H^T)
                const D2u cll = *reinterpret_cast<const D2u*>(row + cl);
                p[q][0] = pp[q][0]; p[q][1] = pp[q][1]; p[q][2] = pp[q][2]; p[q][3] = cll.x; p[q][4] = cll.y;
            }
#pragma unroll
            for (int k = 3; k < 5; k++) {                                // Sigma(c5(lm), r): row gather (H * Sigma)
                const D2u gg = *reinterpret_cast<const D2u*>(Sg + (size_t)idx5(k, lm) * ld + rb);
                g[0][k] = gg.x; g[1][k] = gg.y;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { g[0][k] = gp[0][k]; g[1][k] = gp[1][k]; }
        }
        // ... and the pending pairs' values at the owned indices (the first kPre pairs: one round trip with the gathers;
        // later pairs take one more per four)
        D2u f0[kPre > 0 ? kPre : 1][4];
#pragma unroll
        for (int v = 0; v < kPre; v++) {
            const int vc = v < pc ? v : 0;
            f0[v][0] = *reinterpret_cast<const D2u*>(Ub + (size_t)(2 * vc) * ld + rb);
            f0[v][1] = *reinterpret_cast<const D2u*>(Ub + (size_t)(2 * vc + 1) * ld + rb);
            f0[v][2] = *reinterpret_cast<const D2u*>(Vb + (size_t)(2 * vc) * ld + rb);
            f0[v][3] = *reinterpret_cast<const D2u*>(Vb + (size_t)(2 * vc + 1) * ld + rb);
        }
        if (tid < 4 * pc) {
            const int v = tid >> 2, q = (tid >> 1) & 1, h = tid & 1;
            sh_K5[v][3 + q][h] = Ub[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
            sh_G5[v][3 + q][h] = Vb[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
        }
        // ---- the winner's H, S^-1, nu: from the thread that carries the landmark ----
        if (has_lm && t == lm) {
            if (is_new) {   // :331-381 with the fresh pose: the landmark this reading has just initialised (:321-322)
                pos[0] = sh_t[0]; pos[1] = sh_t[1];
                MeasTerms m;
                m.z0 = sh_z[j][0]; m.z1 = sh_z[j][1];
                predicted_terms(pos[0], pos[1], theta, x, y, m);
                double S[2][2], Si[2][2];
                innovation_cov(S55, m.H, pv.p.r_meas, S);
                inv2(S, Si);
                for (int k = 0; k < 5; k++) { sh_H[k] = m.H[0][k]; sh_H[5 + k] = m.H[1][k]; }
                sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
                sh_nu[0] = m.z0 - m.zh0;
                sh_nu[1] = normalize_angle(m.z1 - m.zh1);
            }
        }
        if (!is_new && tid < 16) {   // as the landmark's thread left them when it scored
            const double tv = sh_terms[lm][tid];
            if (tid < 10) sh_H[tid] = tv;
            else if (tid < 14) sh_Si[tid - 10] = tv;
            else if (tid == 14) sh_nu[0] = tv;
            else sh_nu[1] = normalize_angle(tv);   // :183 (the score used it unwrapped)
        }
        AC_TR(j, 4);
        __syncthreads();
        AC_TR(j, 5);
        // ---- K = Sigma H^T S^-1 (:376), G = H Sigma at the owned indices; pair pc; state ----
        double kq[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, gq[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        // ... as they stand now: minus the pending pairs, in order
        auto fold_rc = [&](int v, const D2u (&f)[4]) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                p[0][k] = p[0][k] - (f[0].x * sh_G5[v][k][0] + f[1].x * sh_G5[v][k][1]);
                g[0][k] = g[0][k] - (sh_K5[v][k][0] * f[2].x + sh_K5[v][k][1] * f[3].x);
                p[1][k] = p[1][k] - (f[0].y * sh_G5[v][k][0] + f[1].y * sh_G5[v][k][1]);
                g[1][k] = g[1][k] - (sh_K5[v][k][0] * f[2].y + sh_K5[v][k][1] * f[3].y);
            }
        };
        auto load_rc = [&](int v, D2u (&f)[4]) {
            f[0] = *reinterpret_cast<const D2u*>(Ub + (size_t)(2 * v) * ld + rb);
            f[1] = *reinterpret_cast<const D2u*>(Ub + (size_t)(2 * v + 1) * ld + rb);
            f[2] = *reinterpret_cast<const D2u*>(Vb + (size_t)(2 * v) * ld + rb);
            f[3] = *reinterpret_cast<const D2u*>(Vb + (size_t)(2 * v + 1) * ld + rb);
        };
        if (nown > 0) {
#pragma unroll
            for (int v = 0; v < kPre; v++)
                if (v < pc) fold_rc(v, f0[v]);   // (uniform)
            int v = kPre;
            for (; v + kBatch <= pc; v += kBatch) {   // (kBatch pairs' values requested together)
                D2u f[kBatch][4];
#pragma unroll
                for (int w = 0; w < kBatch; w++) load_rc(v + w, f[w]);
#pragma unroll
                for (int w = 0; w < kBatch; w++) fold_rc(v + w, f[w]);
            }
            for (; v < pc; v++) {
                D2u f[4];
                load_rc(v, f);
                fold_rc(v, f);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int r = r0 + q;
            if (q < nown && r < Nb) {
                double sht0 = 0.0, sht1 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    sht0 += p[q][k] * sh_H[k];
                    sht1 += p[q][k] * sh_H[5 + k];
                    g0 += sh_H[k] * g[q][k];
                    g1 += sh_H[5 + k] * g[q][k];
                }
                kq[q][0] = sht0 * sh_Si[0] + sht1 * sh_Si[2];   // :376
                kq[q][1] = sht0 * sh_Si[1] + sht1 * sh_Si[3];
                gq[q][0] = g0; gq[q][1] = g1;
                double sv = pos[q] + (kq[q][0] * sh_nu[0] + kq[q][1] * sh_nu[1]);   // :384
                if (r == 0) sv = normalize_angle(sv);                                  // :385
                pos[q] = sv;
            }
        }
        if (nown == 2) {   // the new pair at the thread's two (neighbouring) indices: 16-byte stores
            *reinterpret_cast<D2u*>(Ub + (size_t)(2 * pc) * ld + r0) = D2u{kq[0][0], kq[1][0]};
            *reinterpret_cast<D2u*>(Ub + (size_t)(2 * pc + 1) * ld + r0) = D2u{kq[0][1], kq[1][1]};
            *reinterpret_cast<D2u*>(Vb + (size_t)(2 * pc) * ld + r0) = D2u{gq[0][0], gq[1][0]};
            *reinterpret_cast<D2u*>(Vb + (size_t)(2 * pc + 1) * ld + r0) = D2u{gq[0][1], gq[1][1]};
        } else if (nown == 1) {
            Ub[(size_t)(2 * pc) * ld + r0] = kq[0][0]; Ub[(size_t)(2 * pc + 1) * ld + r0] = kq[0][1];
            Vb[(size_t)(2 * pc) * ld + r0] = gq[0][0]; Vb[(size_t)(2 * pc + 1) * ld + r0] = gq[0][1];
        }
        if (helper && hl < 3) {   // pose part of the new pair, and the pose the next reading sees
            sh_K5[pc][hl][0] = kq[0][0]; sh_K5[pc][hl][1] = kq[0][1]; sh_G5[pc][hl][0] = gq[0][0]; sh_G5[pc][hl][1] = gq[0][1];
            sh_pose[hl] = pos[0];
        }
        AC_TR(j, 6);
        __syncthreads();
        AC_TR(j, 7);
        // ---- the landmark's block takes the new pair (k_rank2's expression): it stays current for the next reading ----
        if (has_lm) {
            double kr[5][2], gc[5][2];
#pragma unroll
            for (int k = 0; k < 3; k++) { kr[k][0] = sh_K5[pc][k][0]; kr[k][1] = sh_K5[pc][k][1]; gc[k][0] = sh_G5[pc][k][0]; gc[k][1] = sh_G5[pc][k][1]; }
#pragma unroll
            for (int q = 0; q < 2; q++) { kr[3 + q][0] = kq[q][0]; kr[3 + q][1] = kq[q][1]; gc[3 + q][0] = gq[q][0]; gc[3 + q][1] = gq[q][1]; }
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
                for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
        }
        pc++;
        AC_TR(j, 8);
    }
#undef AC_TR
    // ---- the call's results: state, pair rows the pass does not use, records ----
    for (int q = 0; q < nown; q++) st[r0 + q] = pos[q];
    for (int r = tid; r < ld; r += (int)blockDim.x)
        for (int v = pc; v < zero_upto; v++) {
            Ub[(size_t)(2 * v) * ld + r] = 0.0; Ub[(size_t)(2 * v + 1) * ld + r] = 0.0;
            Vb[(size_t)(2 * v) * ld + r] = 0.0; Vb[(size_t)(2 * v + 1) * ld + r] = 0.0;
        }
    // (pair rows of indices no thread owns -- beyond the carried landmarks -- are exact zeros)
    for (int r = 3 + 2 * min(nmain, n) + tid; r < ld; r += (int)blockDim.x)
        for (int v = 0; v < pc; v++) {
            Ub[(size_t)(2 * v) * ld + r] = 0.0; Ub[(size_t)(2 * v + 1) * ld + r] = 0.0;
            Vb[(size_t)(2 * v) * ld + r] = 0.0; Vb[(size_t)(2 * v + 1) * ld + r] = 0.0;
        }
    const int Mend = sh_M;
    if (has_lm && t < Mend) {   // touched-set bookkeeping: every landmark below the known count counts as touched
        unsigned char* tf = pv.touch_flag;
        if (!tf[t]) {
            tf[t] = 1;
            const int slot = atomicAdd(&pv.touch_count[0], 1);
            pv.touch_list[slot] = t;
        }
    }
    if (tid == 0) {
        cnt_out[0] = pc;
        AssocRec rec;
        rec.known_count = Mend; rec.lm = sh_lm; rec.active = sh_lm >= 0; rec.pad = 0; rec.best = 0.0;
        pv.assoc[0] = rec;
        CorrRec rc;
        rc.nu0 = 0.0; rc.nu1 = 0.0; rc.active = sh_lm >= 0; rc.lm = sh_lm; rc.n_active = 0; rc.pad = 0;
        pv.rec[0] = rc;
    }
}

int assoc_call_capacity() { return kCallLandmarks; }

void launch_assoc_call(const PoolView& pv, const AssocCallArgs& a, int carried, int* assoc_out, double* U, double* V,
                       int* cnt_out, int zero_upto, hipStream_t s, long long* trace, char* pub_host, int pub_j0,
                       unsigned pub_seq) {
    const int waves = (carried + 63) / 64 > 0 ? (carried + 63) / 64 : 1;
    if (waves <= 3)
        hipLaunchKernelGGL(k_assoc_call<256>, dim3(1), dim3(64 * waves + 64), 0, s, pv, a, assoc_out, U, V, cnt_out, zero_upto, trace,
                           pub_host, pub_j0, pub_seq);
    else
        hipLaunchKernelGGL(k_assoc_call<512>, dim3(1), dim3(64 * waves + 64), 0, s, pv, a, assoc_out, U, V, cnt_out, zero_upto, trace,
                           pub_host, pub_j0, pub_seq);
}

void launch_assoc_score(const PoolView& pv, double mx, double my, const AssocRec* assoc_in, const double* U, const double* V,
                        int pc, int m_bound, double* scores, double* terms, hipStream_t s, double* blk) {
    const int cover = blk ? pv.n : m_bound;   // with the block cache: every landmark's block is taken, the known ones scored
    if (cover > 0)
        hipLaunchKernelGGL(k_assoc_score, dim3((cover + 63) / 64), dim3(64), 0, s, pv, mx, my, assoc_in, scores, terms, U, V, pc,
                           blk);
}

void launch_assoc_reading(const PoolView& pv, double mx, double my, int has_next, double mxn, double myn,
                          const AssocRec* assoc_in, AssocRec* assoc_next, int* assoc_out_j, double* state_out, double* U,
                          double* V, int* cnt_out, int pc, int Nb, int zero_upto, int m_bound, const double* scores,
                          const double* terms, double* scores_out, double* terms_out, hipStream_t s, long long* trace,
                          double* blk, char* pub_host, int pub_j, unsigned pub_seq) {
    const int landmarks = (pv.ld - 3 + 1) / 2;   // owners of every index of the padded row beyond the pose
    hipLaunchKernelGGL(k_assoc_reading, dim3((landmarks + kAssocLandmarks - 1) / kAssocLandmarks), dim3(kAssocThreads),
                       0, s, pv, mx, my, has_next, mxn, myn, assoc_in, assoc_next, assoc_out_j, state_out, U, V, cnt_out, pc, Nb,
                       zero_upto, m_bound, scores, terms, scores_out, terms_out, trace, blk, pub_host, pub_j, pub_seq);
}

}  // namespace ekf
