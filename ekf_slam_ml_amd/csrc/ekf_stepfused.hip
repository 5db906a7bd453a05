// ekf_stepfused.hip -- one STEP of the unknown-association node loop (data_association() of <= kCallV readings,
// ekf_slam.cpp:278-402) for every filter of a pool in ONE launch, for discovered prefixes of ANY size: one workgroup per
// filter, and the filter's covariance is streamed ONCE per step instead of once per reading.
//
// The LDS-resident step kernel (k_pool_associate, ekf_small.hip) stops at N_b = 104; beyond it a step used to be four
// launches per measurement slot (k_maha, k_assoc_decide, k_gain, k_rank2), each streaming or gathering from every
// filter's 32-MB slab.  Here the corrections of a step are kept as factor pairs (K_v, G_v) until the step ends --
// the idea of ekf_callfused.hip applied to data_association, whose decisions are sequential:
//   for every reading j of the filter, in order:
//     scores      one landmark per thread: the 25 entries of Sigma[c5(i), c5(i)] are the stored (step-begin) entries
//                 minus the pending pairs of this step, applied in order with the rank-2 kernel's own expression --
//                 the values the per-reading path would have read; the fresh pose and landmark come from the state,
//                 which IS updated after every reading (:219-221, :331-333)
//     decision    lexicographic (d, i) minimum, the two gates, landmark initialisation (:293-330)
//     gain        K = Sigma H^T S^-1 and G = H Sigma over the filter's prefix from the same reconstruction of the five
//                 rows / columns; appended as pair `pc`; state += K nu (:376-385)
//   then ONE pass over the prefix applies the pairs to every element in order: x <- (x - K_0 G_0) - K_1 G_1 ...
// Everything is the arithmetic of k_maha / k_assoc_decide / k_gain / k_rank2 in the same order -> decisions, state and
// covariance are bit-identical to the four-launch path (tests/test_gpu_batch_unknown.py).
#include "ekf_kernels.hpp"

#include <climits>

namespace ekf {

constexpr int kStepThreads = 512;
constexpr int kSpecRows = 4 * kCallV + 6;   // k_pool_step_spec: per guessed landmark 2 column + 2 row vectors, then 3 + 3 for the pose
constexpr int kStepPendingPairs = 64;   // delayed mode: pairs a filter can carry between flushes (= max_pending() / 2)

// DELAYED (ekf_batch_set_update_mode(k > 0), SURVEY.md section 8(f) f2 applied to data_association): the pairs of a step are
// not applied at its end but stay pending ACROSS steps in the pool's factor store (prediction() maps them, k_predict);
// this step's readings see the covariance as "stored minus ALL pending pairs": p0 pairs from earlier steps + its own.
// The store is flushed every few steps (k_flush: one pass over Sigma per flush instead of per step).  Every filter appends
// exactly `zero_upto` pairs per step (zero pairs beyond its own readings), so the pending count stays uniform over the pool.
template <bool DELAYED>
__global__ __launch_bounds__(kStepThreads) void k_pool_step_unknown(PoolView pv, const double* __restrict__ meas_all,
                                                                    const int* __restrict__ count, int jmax, int min_active,
                                                                    int* __restrict__ assoc_out, double* __restrict__ Uall,
                                                                    double* __restrict__ Vall,
                                                                    unsigned long long* __restrict__ corr_counter,
                                                                    int* __restrict__ cnt_out, int zero_upto, int p0,
                                                                    int pair_rows, double* __restrict__ blocks_all,
                                                                    const double* __restrict__ pred, int sym,
                                                                    const double* __restrict__ spec_all,
                                                                    const int* __restrict__ specw_all) {
    constexpr int kPairs = DELAYED ? kStepPendingPairs : kCallV;
    // cnt_out == nullptr: the final pass runs here (one workgroup streams its filter's covariance).  Otherwise the step
    // ends with the pairs in Uall / Vall (rows beyond the filter's pair count zero-filled up to `zero_upto` pairs) and
    // cnt_out[b] pairs -- the caller streams all covariances with k_rank2v, which spreads every filter over the chip.
    const int b = blockIdx.x, tid = threadIdx.x;
    const int J = count ? count[b] : jmax;
    const int n = pv.n, ld = pv.ld;
    if (J <= 0 && !DELAYED) {  // uniform: nothing to do for this filter in this step
        if (cnt_out && tid == 0) cnt_out[b] = 0;
        return;
    }
    // (no __restrict__: the pairs and the state are written and re-read by this workgroup, ordered by its barriers)
    double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    double* st = pv.state + (size_t)b * ld;
    double* Ub = Uall + (size_t)b * pair_rows * ld;
    double* Vb = Vall + (size_t)b * pair_rows * ld;
    const double* meas = meas_all + (size_t)b * jmax * 2;
    int* out = assoc_out + (size_t)b * jmax;

    __shared__ double sh_d[kStepThreads / 64];
    __shared__ int sh_i[kStepThreads / 64];
    __shared__ int sh_M, sh_lm, sh_new, sh_applied, sh_slot;
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ double2_t sh_K[32][kCallV];   // K_v of a row block of the final pass
    // pending pairs at the five indices c5 of the landmark in hand: [v][0..2] = pose rows / columns (kept up to date as
    // pairs are appended), [v][3..4] = the winner's two rows / columns (fetched after the decision)
    __shared__ double sh_K5[kPairs][5][2], sh_G5[kPairs][5][2];

    const int kc0 = pv.assoc[b].known_count;
    // the filter's active dimension for this step: rows / columns beyond it still hold constructor values, where K and G
    // are exact zeros (landmarks are appended in discovery order, :318-327)
    int Nb = 3 + 2 * min(n, kc0 + J);
    if (min_active > Nb) Nb = min_active;
    if (Nb > pv.N) Nb = pv.N;
    if (tid == 0) { sh_M = kc0; sh_applied = 0; }
    if constexpr (DELAYED) {   // pose part of the pairs pending from earlier steps
        for (int e = tid; e < 12 * p0; e += kStepThreads) {
            const int v = e / 12, q = (e % 12) >> 2, h = (e >> 1) & 1, uvsel = e & 1;
            if (uvsel == 0) sh_K5[v][q][h] = Ub[(size_t)(2 * v + h) * ld + q];
            else sh_G5[v][q][h] = Vb[(size_t)(2 * v + h) * ld + q];
        }
    }
    // ---- block cache (delayed mode): blk[e][i] = entry e of Sigma[c5(i), c5(i)] AS IT STANDS NOW (stored minus every pending
    // pair), kept current by folding each new pair in.  A score then reads 25 consecutive-in-i values instead of 25 scattered
    // entries of Sigma plus 8 values per pending pair (up to 64 pairs are pending).  (Re)built from Sigma when nothing is
    // pending; carried across steps, where this step's prediction() is applied to it first (the 5 x 5 block is closed under At . At^T + Q: rows /
    // columns 0, 1, 2 are inside it; k_predict's arithmetic).  Landmark i is always handled by thread i mod 512.
    double* blk = DELAYED ? blocks_all + (size_t)b * 25 * n : nullptr;
    if (!DELAYED) {
    } else if (p0 == 0) {
        for (int i = tid; i < n; i += kStepThreads) {
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
                for (int l = 0; l < 5; l++) blk[(size_t)(5 * k + l) * n + i] = Sg[(size_t)idx5(k, i) * ld + idx5(l, i)];
        }
    } else {
        const double a10 = pred[(size_t)b * 2], a20 = pred[(size_t)b * 2 + 1];
        for (int i = tid; i < n; i += kStepThreads) {
            double c[5][5];
#pragma unroll
            for (int e = 0; e < 25; e++) c[e / 5][e % 5] = blk[(size_t)e * n + i];
            double o[5][5];
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
                for (int l = 0; l < 5; l++) o[k][l] = c[k][l];
#pragma unroll
            for (int l = 3; l < 5; l++) {   // rows 1, 2 at the landmark's columns; columns 1, 2 at its rows
                o[1][l] = a10 * c[0][l] + c[1][l];
                o[2][l] = a20 * c[0][l] + c[2][l];
                o[l][1] = c[l][0] * a10 + c[l][1];
                o[l][2] = c[l][0] * a20 + c[l][2];
            }
            double T[3][3];
#pragma unroll
            for (int l = 0; l < 3; l++) {
                T[0][l] = c[0][l];
                T[1][l] = a10 * c[0][l] + c[1][l];
                T[2][l] = a20 * c[0][l] + c[2][l];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                o[k][0] = T[k][0];
                o[k][1] = T[k][0] * a10 + T[k][1];
                o[k][2] = T[k][0] * a20 + T[k][2];
            }
            o[0][0] += pv.p.q_pose; o[1][1] += pv.p.q_pose; o[2][2] += pv.p.q_pose;   // Q = diag(q,q,q,0...) :40-43
#pragma unroll
            for (int e = 0; e < 25; e++) blk[(size_t)e * n + i] = o[e / 5][e % 5];
        }
    }
    __threadfence_block();
    __syncthreads();
    if (J <= 0) {   // (delayed mode, uniform) a filter that sits the step out: its cache is up to date, its pairs are zero pairs
        if (cnt_out && tid == 0) cnt_out[b] = 0;
        for (int r = tid; r < ld; r += kStepThreads)
            for (int v = p0; v < p0 + zero_upto; v++) {
                Ub[(size_t)(2 * v) * ld + r] = 0.0; Ub[(size_t)(2 * v + 1) * ld + r] = 0.0;
                Vb[(size_t)(2 * v) * ld + r] = 0.0; Vb[(size_t)(2 * v + 1) * ld + r] = 0.0;
            }
        return;
    }

    int pc = p0;  // pending pairs the next reading sees: earlier steps' (delayed mode) + this step's (uniform)
    for (int j = 0; j < J; j++) {  // :291 sequential, state-carrying
        const double mx = meas[2 * j], my = meas[2 * j + 1];
        const int M = sh_M;
        const double theta = st[0], x = st[1], y = st[2];   // fresh pose per reading, :219-221 / :331-333
        // ---- scores, :300-309: one landmark per thread; the thread keeps H, S^-1, nu of its best landmark ----
        double best = pv.p.gate_new;  // :293
        int bi = INT_MAX;
        double bH[10], bSi[4], bnu[2];
        for (int i = tid; i < M; i += kStepThreads) {
            MeasTerms m;
            measurement_terms(st[2 * i + 3], st[2 * i + 4], mx, my, theta, x, y, m);
            double S55[5][5], S[2][2], Si[2][2];
            if constexpr (DELAYED) {
#pragma unroll
                for (int e = 0; e < 25; e++) S55[e / 5][e % 5] = blk[(size_t)e * n + i];   // current: every pending pair folded in
            } else {
                const int ia = 3 + 2 * i;
    #pragma unroll
                for (int k = 0; k < 5; k++)
    #pragma unroll
                    for (int l = 0; l < 5; l++) S55[k][l] = Sg[(size_t)idx5(k, i) * ld + idx5(l, i)];
                // the entries as they stand NOW: minus the pending pairs, in order (k_rank2's expression).  The pairs' values at
                // the landmark's two indices come from memory: two pairs are requested together (in delayed mode up to 64
                // pairs are pending, and one round trip per pair was most of the step)
                auto fold_pair = [&](int v, const double (&ku)[2][2], const double (&gu)[2][2]) {
                    double kr[5][2], gc[5][2];
    #pragma unroll
                    for (int k = 0; k < 3; k++) { kr[k][0] = sh_K5[v][k][0]; kr[k][1] = sh_K5[v][k][1]; gc[k][0] = sh_G5[v][k][0]; gc[k][1] = sh_G5[v][k][1]; }
    #pragma unroll
                    for (int q = 0; q < 2; q++) { kr[3 + q][0] = ku[q][0]; kr[3 + q][1] = ku[q][1]; gc[3 + q][0] = gu[q][0]; gc[3 + q][1] = gu[q][1]; }
    #pragma unroll
                    for (int k = 0; k < 5; k++)
    #pragma unroll
                        for (int l = 0; l < 5; l++) S55[k][l] = S55[k][l] - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
                };
                auto load_pair = [&](int v, double (&ku)[2][2], double (&gu)[2][2]) {
    #pragma unroll
                    for (int q = 0; q < 2; q++) {
                        ku[q][0] = Ub[(size_t)(2 * v) * ld + ia + q]; ku[q][1] = Ub[(size_t)(2 * v + 1) * ld + ia + q];
                        gu[q][0] = Vb[(size_t)(2 * v) * ld + ia + q]; gu[q][1] = Vb[(size_t)(2 * v + 1) * ld + ia + q];
                    }
                };
                int v = 0;
                for (; v + 2 <= pc; v += 2) {
                    double ku[2][2][2], gu[2][2][2];
    #pragma unroll
                    for (int w = 0; w < 2; w++) load_pair(v + w, ku[w], gu[w]);
    #pragma unroll
                    for (int w = 0; w < 2; w++) fold_pair(v + w, ku[w], gu[w]);
                }
                for (; v < pc; v++) {
                    double ku[2][2], gu[2][2];
                    load_pair(v, ku, gu);
                    fold_pair(v, ku, gu);
                }
            }
            innovation_cov(S55, m.H, pv.p.r_meas, S);   // sums in the order of k_maha's shuffle folds
            inv2(S, Si);
            const double v0 = m.z0 - m.zh0, v1 = m.z1 - m.zh1;   // bearing NOT wrapped, :269
            const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
            const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
            const double sc = t0 * v0 + t1 * v1;
            if (sc < best) {   // :305-309 (ascending i within the thread: the first minimum is kept; NaN never wins)
                best = sc; bi = i;
#pragma unroll
                for (int k = 0; k < 5; k++) { bH[k] = m.H[0][k]; bH[5 + k] = m.H[1][k]; }
                bSi[0] = Si[0][0]; bSi[1] = Si[0][1]; bSi[2] = Si[1][0]; bSi[3] = Si[1][1];
                bnu[0] = v0; bnu[1] = v1;
            }
        }
        // lexicographic (d, i) minimum = the sequential scan's first strict minimum
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
        if (tid == 0) {   // :293-330 (k_assoc_decide)
            for (int w = 1; w < kStepThreads / 64; w++)
                if (sh_d[w] < rd || (sh_d[w] == rd && sh_i[w] < ri)) { rd = sh_d[w]; ri = sh_i[w]; }
            const int idx = (ri == INT_MAX) ? M : ri;   // :294 min_maha_idx = known_count
            int Mn = M, is_new = 0;
            if (idx == M && idx < n) {                  // :318-327 new landmark
                const double rr = sqrt(mx * mx + my * my);
                const double phi = atan2(my, mx);
                st[2 * idx + 3] = x + rr * cos(phi + theta);
                st[2 * idx + 3 + 1] = y + rr * sin(phi + theta);
                Mn = M + 1;
                rd = 0.0;
                is_new = 1;
            }
            const int active = (rd < pv.p.gate_update) && idx < n;   // :330
            sh_M = Mn;
            sh_lm = active ? idx : -1;
            sh_new = is_new;
            out[j] = sh_lm;
            if (active) sh_applied++;
            int slot = -1;   // the launch in front guessed this winner (k_pool_step_spec): its old part is ready
            if (DELAYED && spec_all && active && !is_new)
                for (int q = 0; q < kCallV; q++) if (specw_all[(size_t)b * kCallV + q] == idx) { slot = q; break; }
            sh_slot = slot;
        }
        __syncthreads();
        const int lm = sh_lm;
        if (lm < 0) { __syncthreads(); continue; }   // dropped (uniform); the barrier keeps sh_lm stable for slow waves
        if (tid < 4 * pc) {   // pending pairs at the winner's two rows / columns
            const int v = tid >> 2, q = (tid >> 1) & 1, h = tid & 1;
            sh_K5[v][3 + q][h] = Ub[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
            sh_G5[v][3 + q][h] = Vb[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
        }
        __syncthreads();
        // ---- H, S^-1, nu of the winner: from the thread that scored it, or built for a new landmark (:331-381) ----
        if (sh_new) {
            if (tid == 0) {
                MeasTerms m;
                measurement_terms(st[2 * lm + 3], st[2 * lm + 4], mx, my, theta, x, y, m);
                double S55[5][5], S[2][2], Si[2][2];
                if constexpr (DELAYED) {
                    for (int e = 0; e < 25; e++) S55[e / 5][e % 5] = blk[(size_t)e * n + lm];   // (current, see above)
                } else {
                    for (int k = 0; k < 5; k++)
                        for (int l = 0; l < 5; l++) {
                            double xe = Sg[(size_t)idx5(k, lm) * ld + idx5(l, lm)];
                            for (int v = 0; v < pc; v++) xe = xe - (sh_K5[v][k][0] * sh_G5[v][l][0] + sh_K5[v][k][1] * sh_G5[v][l][1]);
                            S55[k][l] = xe;
                        }
                }
                innovation_cov(S55, m.H, pv.p.r_meas, S);
                inv2(S, Si);
                for (int k = 0; k < 5; k++) { sh_H[k] = m.H[0][k]; sh_H[5 + k] = m.H[1][k]; }
                sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
                sh_nu[0] = m.z0 - m.zh0;
                sh_nu[1] = normalize_angle(m.z1 - m.zh1);
            }
        } else if (bi == lm) {   // exactly one thread scored landmark lm
#pragma unroll
            for (int k = 0; k < 10; k++) sh_H[k] = bH[k];
#pragma unroll
            for (int k = 0; k < 4; k++) sh_Si[k] = bSi[k];
            sh_nu[0] = bnu[0];
            sh_nu[1] = normalize_angle(bnu[1]);   // :183 (the score used it unwrapped)
        }
        __syncthreads();
        // ---- K = Sigma H^T S^-1, G = H Sigma over the prefix (k_gain), appended as pair pc; state += K nu ----
        for (int r = tid; r < ld; r += kStepThreads) {
            double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0;
            if (r < Nb) {
                // sym (the symmetric option of the delayed mode, uniform): Sigma H^T is taken as (H Sigma)^T -- no column
                // gather, no read of the U half of the pending pairs
                double p[5], g[5];
                const int slot = DELAYED ? sh_slot : -1;
                int v = 0;
                if (slot >= 0) {
                    // guessed right: "stored minus the pairs of earlier steps" for this landmark and the pose indices was
                    // rebuilt once for the whole step by k_pool_step_spec, in this loop's own order -- ten coalesced loads
                    // instead of the column sectors, the rows and p0 x 4 factor rows; only this step's pairs are left to fold
                    const double* sp = spec_all + (size_t)b * kSpecRows * ld;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        if (!sym) p[k] = sp[(size_t)(4 * kCallV + k) * ld + r];
                        g[k] = sp[(size_t)(4 * kCallV + 3 + k) * ld + r];
                    }
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        if (!sym) p[3 + q] = sp[(size_t)(4 * slot + q) * ld + r];
                        g[3 + q] = sp[(size_t)(4 * slot + 2 + q) * ld + r];
                    }
                    v = p0;
                } else {
                    if (!sym) gather_row5(Sg + (size_t)r * ld, lm, p);   // column gather (Sigma * H^T reads columns): three loads
#pragma unroll
                    for (int k = 0; k < 5; k++) g[k] = Sg[(size_t)idx5(k, lm) * ld + r];   // row gather (H * Sigma reads rows)
                }
                // ... as they stand now: minus the pending pairs, in order (four pairs' values requested together)
                auto fold_rc = [&](int v, const double (&f)[4]) {
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        if (!sym) p[k] = p[k] - (f[0] * sh_G5[v][k][0] + f[1] * sh_G5[v][k][1]);
                        g[k] = g[k] - (sh_K5[v][k][0] * f[2] + sh_K5[v][k][1] * f[3]);
                    }
                };
                auto load_rc = [&](int v, double (&f)[4]) {
                    if (!sym) { f[0] = Ub[(size_t)(2 * v) * ld + r]; f[1] = Ub[(size_t)(2 * v + 1) * ld + r]; }
                    f[2] = Vb[(size_t)(2 * v) * ld + r]; f[3] = Vb[(size_t)(2 * v + 1) * ld + r];
                };
                for (; v + 4 <= pc; v += 4) {
                    double f[4][4];
#pragma unroll
                    for (int w = 0; w < 4; w++) load_rc(v + w, f[w]);
#pragma unroll
                    for (int w = 0; w < 4; w++) fold_rc(v + w, f[w]);
                }
                for (; v < pc; v++) {
                    double f[4];
                    load_rc(v, f);
                    fold_rc(v, f);
                }
                if (sym) {
#pragma unroll
                    for (int k = 0; k < 5; k++) p[k] = g[k];
                }
                double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    sht0 += p[k] * sh_H[k];
                    sht1 += p[k] * sh_H[5 + k];
                    g0 += sh_H[k] * g[k];
                    g1 += sh_H[5 + k] * g[k];
                }
                k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];   // :178 / :376
                k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
                double s = st[r] + (k0 * sh_nu[0] + k1 * sh_nu[1]);   // :384
                if (r == 0) s = normalize_angle(s);                    // :385
                st[r] = s;
            }
            // (the pair is read back by cur_entry only after the barrier; entries beyond the prefix are exact zeros)
            Ub[(size_t)(2 * pc) * ld + r] = k0;
            Ub[(size_t)(2 * pc + 1) * ld + r] = k1;
            Vb[(size_t)(2 * pc) * ld + r] = g0;
            Vb[(size_t)(2 * pc + 1) * ld + r] = g1;
            if (r < 3) { sh_K5[pc][r][0] = k0; sh_K5[pc][r][1] = k1; sh_G5[pc][r][0] = g0; sh_G5[pc][r][1] = g1; }   // pose part of the new pair
        }
        __threadfence_block();
        __syncthreads();
        if constexpr (DELAYED) {   // the cached blocks take the new pair (k_rank2's expression); beyond the active dimension K = G = 0
            const int nfold = min(n, (Nb - 3) >> 1);
            for (int i = tid; i < nfold; i += kStepThreads) {
                const int ia = 3 + 2 * i;
                double kr[5][2], gc[5][2];
#pragma unroll
                for (int k = 0; k < 3; k++) { kr[k][0] = sh_K5[pc][k][0]; kr[k][1] = sh_K5[pc][k][1]; gc[k][0] = sh_G5[pc][k][0]; gc[k][1] = sh_G5[pc][k][1]; }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    kr[3 + q][0] = Ub[(size_t)(2 * pc) * ld + ia + q]; kr[3 + q][1] = Ub[(size_t)(2 * pc + 1) * ld + ia + q];
                    gc[3 + q][0] = Vb[(size_t)(2 * pc) * ld + ia + q]; gc[3 + q][1] = Vb[(size_t)(2 * pc + 1) * ld + ia + q];
                }
#pragma unroll
                for (int e = 0; e < 25; e++) {
                    const int k = e / 5, l = e % 5;
                    const double xv = blk[(size_t)e * n + i];
                    blk[(size_t)e * n + i] = xv - (kr[k][0] * gc[l][0] + kr[k][1] * gc[l][1]);
                }
            }
        }
        pc++;
        __threadfence_block();
        __syncthreads();
    }

    if (cnt_out) {   // the pass is the caller's: unused pair rows become exact no-ops
        for (int r = tid; r < ld; r += kStepThreads)
            for (int v = pc; v < p0 + zero_upto; v++) {
                Ub[(size_t)(2 * v) * ld + r] = 0.0; Ub[(size_t)(2 * v + 1) * ld + r] = 0.0;
                Vb[(size_t)(2 * v) * ld + r] = 0.0; Vb[(size_t)(2 * v + 1) * ld + r] = 0.0;
            }
        if (tid == 0) cnt_out[b] = pc - p0;
    }
    // ---- ONE pass over the prefix: every element takes the step's corrections in order (k_rank2's expression) ----
    if (!DELAYED && pc > 0 && !cnt_out) {
        const int ld2n = ld >> 1, ld2a = (Nb + 1) >> 1;
        double2_t* S2 = reinterpret_cast<double2_t*>(Sg);
        const double2_t* V2 = reinterpret_cast<const double2_t*>(Vb);
        for (int c0 = 0; c0 < ld2a; c0 += kStepThreads) {
            const int c2 = c0 + tid;
            const bool col_live = c2 < ld2a;
            double2_t g0[kCallV], g1[kCallV];
#pragma unroll
            for (int v = 0; v < kCallV; v++) {
                g0[v] = double2_t{0.0, 0.0}; g1[v] = double2_t{0.0, 0.0};
                if (v < pc && col_live) { g0[v] = V2[(size_t)(2 * v) * ld2n + c2]; g1[v] = V2[(size_t)(2 * v + 1) * ld2n + c2]; }
            }
            for (int rb = 0; rb < Nb; rb += 32) {
                __syncthreads();   // the previous block's sh_K has been consumed
                for (int e = tid; e < 32 * kCallV; e += kStepThreads) {
                    const int rr = e / kCallV, v = e - rr * kCallV;
                    const int row = rb + rr;
                    double2_t kk{0.0, 0.0};
                    if (v < pc && row < Nb) kk = double2_t{Ub[(size_t)(2 * v) * ld + row], Ub[(size_t)(2 * v + 1) * ld + row]};
                    sh_K[rr][v] = kk;
                }
                __syncthreads();
                if (col_live) {
                    const int nrow = min(32, Nb - rb);
                    for (int u0 = 0; u0 < nrow; u0 += 8) {
                        double2_t xv[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) xv[u] = S2[(size_t)min(rb + u0 + u, Nb - 1) * ld2n + c2];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
#pragma unroll
                            for (int v = 0; v < kCallV; v++) {
                                if (v < pc) {   // uniform
                                    const double2_t kk = sh_K[u0 + u][v];
                                    xv[u].x = xv[u].x - (kk.x * g0[v].x + kk.y * g1[v].x);
                                    xv[u].y = xv[u].y - (kk.x * g0[v].y + kk.y * g1[v].y);
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++)
                            if (u0 + u < nrow) S2[(size_t)(rb + u0 + u) * ld2n + c2] = xv[u];
                    }
                }
            }
        }
    }
    if (tid == 0) {
        AssocRec a;
        a.known_count = sh_M; a.lm = sh_lm; a.active = sh_lm >= 0; a.pad = 0; a.best = 0.0;
        pv.assoc[b] = a;
        if (corr_counter && sh_applied) atomicAdd(corr_counter, (unsigned long long)sh_applied);
        // touched-set bookkeeping (active-set mode): every landmark below the new known_count counts as touched
        unsigned char* tf = pv.touch_flag + (size_t)b * pv.n;
        for (int i = 0; i < sh_M && i < n; i++)
            if (!tf[i]) { tf[i] = 1; pv.touch_list[(size_t)b * pv.n + pv.touch_count[b]] = i; pv.touch_count[b]++; }
    }
}

// ---------------------------------------------------------------------------------------------
// The old part of a delayed step's gains, once per step (see launch_pool_step_spec in ekf_kernels.hpp).
// grid (ceil(ld / 512), B), 256 threads, a lane owns the indices r, r + 1 (k_gain_delayed_pair's layout).  Every workgroup
// repeats the guess (a reduction over the filter's landmarks: cheap) and gathers the old pairs' entries at the guessed
// indices; the lanes then stream the old pairs ONCE for all guesses.  Arithmetic: the step kernel's fold_rc, pair by pair
// in the same order, so that what it continues from is what it would have had.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool_step_spec(PoolView pv, const double* __restrict__ meas_all,
                                                        const int* __restrict__ count, int jmax, const double* __restrict__ Uall,
                                                        const double* __restrict__ Vall, int p0, int pair_rows, int sym,
                                                        double* __restrict__ spec_all, int* __restrict__ specw_all) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int n = pv.n, ld = pv.ld, N = pv.N;
    const int J = count ? count[b] : jmax;
    int* wout = specw_all + (size_t)b * kCallV;
    if (J <= 0) {   // (uniform) nothing to prepare (p0 = 0 right behind a flush: the old part is the stored entries themselves --
                    // still one gather of the column sectors per step instead of one per reading)
        if (blockIdx.x == 0 && tid < kCallV) wout[tid] = -1;
        return;
    }
    const double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    const double* st = pv.state + (size_t)b * ld;
    const double* Ub = Uall + (size_t)b * pair_rows * ld;
    const double* Vb = Vall + (size_t)b * pair_rows * ld;
    const double* meas = meas_all + (size_t)b * jmax * 2;
    const int M = min(pv.assoc[b].known_count, n);
    __shared__ double sh_d[4][kCallV];
    __shared__ int sh_i[4][kCallV];
    __shared__ int sh_w[kCallV];
    __shared__ double sh_Kp[kStepPendingPairs][3][2], sh_Gp[kStepPendingPairs][3][2];            // old pairs at the pose indices
    __shared__ double sh_Kw[kStepPendingPairs][kCallV][2][2], sh_Gw[kStepPendingPairs][kCallV][2][2];   // ... at the guessed landmarks
    // ---- the guess: reading j lands at pose (+) reading; its landmark is the nearest one (the decision itself is the step
    // kernel's: a wrong guess costs it the rebuild it would have done anyway) ----
    const double theta = st[0], x = st[1], y = st[2];
    const double c = cos(theta), sn = sin(theta);
    double bd[kCallV];
    int bi[kCallV];
#pragma unroll
    for (int j = 0; j < kCallV; j++) { bd[j] = __builtin_huge_val(); bi[j] = -1; }
    for (int i = tid; i < M; i += 256) {
        const double lx = st[2 * i + 3], ly = st[2 * i + 4];
#pragma unroll
        for (int j = 0; j < kCallV; j++) {
            if (j < J) {
                const double wx = x + (meas[2 * j] * c - meas[2 * j + 1] * sn), wy = y + (meas[2 * j] * sn + meas[2 * j + 1] * c);
                const double d = (lx - wx) * (lx - wx) + (ly - wy) * (ly - wy);
                if (d < bd[j]) { bd[j] = d; bi[j] = i; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kCallV; j++) {
        double rd = bd[j];
        int ri = bi[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_down(rd, off, kWave);
            const int oi = __shfl_down(ri, off, kWave);
            if (od < rd || (od == rd && oi >= 0 && (ri < 0 || oi < ri))) { rd = od; ri = oi; }
        }
        if (lane == 0) { sh_d[tid >> 6][j] = rd; sh_i[tid >> 6][j] = ri; }
    }
    __syncthreads();
    if (tid < kCallV) {
        double rd = sh_d[0][tid];
        int ri = sh_i[0][tid];
        for (int w = 1; w < 4; w++)
            if (sh_d[w][tid] < rd || (sh_d[w][tid] == rd && sh_i[w][tid] >= 0 && (ri < 0 || sh_i[w][tid] < ri))) { rd = sh_d[w][tid]; ri = sh_i[w][tid]; }
        sh_w[tid] = tid < J ? ri : -1;
    }
    __syncthreads();
    if (tid == 0) {   // one slot per distinct landmark
        for (int j = 1; j < kCallV; j++)
            for (int q = 0; q < j; q++)
                if (sh_w[j] >= 0 && sh_w[j] == sh_w[q]) { sh_w[j] = -1; break; }
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < kCallV) wout[tid] = sh_w[tid];
    // ---- the old pairs' entries at the pose indices and at the guessed landmarks' indices ----
    for (int e = tid; e < p0 * 12; e += 256) {
        const int v = e / 12, q = (e % 12) >> 2, h = (e >> 1) & 1, uvsel = e & 1;
        if (uvsel == 0) sh_Kp[v][q][h] = Ub[(size_t)(2 * v + h) * ld + q];
        else sh_Gp[v][q][h] = Vb[(size_t)(2 * v + h) * ld + q];
    }
    for (int e = tid; e < p0 * kCallV * 8; e += 256) {
        const int v = e / (kCallV * 8), rest = e - v * (kCallV * 8);
        const int s = rest >> 3, q = (rest >> 2) & 1, h = (rest >> 1) & 1, uvsel = rest & 1;
        const int lm = sh_w[s];
        double val = 0.0;
        if (lm >= 0) val = (uvsel == 0 ? Ub : Vb)[(size_t)(2 * v + h) * ld + 3 + 2 * lm + q];
        if (uvsel == 0) sh_Kw[v][s][q][h] = val; else sh_Gw[v][s][q][h] = val;
    }
    __syncthreads();
    // ---- stream the old pairs once: r, r + 1 per lane ----
    const int r = 2 * (blockIdx.x * 256 + tid);
    if (r >= ld) return;
    double* sp = spec_all + (size_t)b * kSpecRows * ld;
    const double2_t zero2 = {0.0, 0.0};
    const bool live = r < N, two = r + 1 < N;
    double2_t pp[3], gp[3], pw[kCallV][2], gw[kCallV][2];
#pragma unroll
    for (int k = 0; k < 3; k++) { pp[k] = zero2; gp[k] = zero2; }
#pragma unroll
    for (int s = 0; s < kCallV; s++) { pw[s][0] = pw[s][1] = gw[s][0] = gw[s][1] = zero2; }
    if (live) {
        const double* rw0 = Sg + (size_t)r * ld;
        const double* rw1 = Sg + (size_t)(two ? r + 1 : r) * ld;
        if (!sym) {
            const D2u a0 = *reinterpret_cast<const D2u*>(rw0), a1 = *reinterpret_cast<const D2u*>(rw1);
            pp[0] = double2_t{a0.x, a1.x}; pp[1] = double2_t{a0.y, a1.y}; pp[2] = double2_t{rw0[2], rw1[2]};
        }
#pragma unroll
        for (int k = 0; k < 3; k++) gp[k] = *reinterpret_cast<const double2_t*>(Sg + (size_t)k * ld + r);
#pragma unroll
        for (int s = 0; s < kCallV; s++) {
            const int lm = sh_w[s];
            if (lm < 0) continue;   // (uniform)
            const size_t cl = 3 + 2 * (size_t)lm;
            if (!sym) {
                const D2u b0 = *reinterpret_cast<const D2u*>(rw0 + cl), b1 = *reinterpret_cast<const D2u*>(rw1 + cl);
                pw[s][0] = double2_t{b0.x, b1.x}; pw[s][1] = double2_t{b0.y, b1.y};
            }
            gw[s][0] = *reinterpret_cast<const double2_t*>(Sg + cl * ld + r);
            gw[s][1] = *reinterpret_cast<const double2_t*>(Sg + (cl + 1) * ld + r);
        }
        for (int v = 0; v < p0; v++) {
            double2_t f0 = zero2, f1 = zero2;
            if (!sym) {
                f0 = *reinterpret_cast<const double2_t*>(Ub + (size_t)(2 * v) * ld + r);
                f1 = *reinterpret_cast<const double2_t*>(Ub + (size_t)(2 * v + 1) * ld + r);
            }
            const double2_t f2 = *reinterpret_cast<const double2_t*>(Vb + (size_t)(2 * v) * ld + r);
            const double2_t f3 = *reinterpret_cast<const double2_t*>(Vb + (size_t)(2 * v + 1) * ld + r);
            // (the step kernel's fold_rc, component by component)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (!sym) {
                    pp[k].x = pp[k].x - (f0.x * sh_Gp[v][k][0] + f1.x * sh_Gp[v][k][1]);
                    pp[k].y = pp[k].y - (f0.y * sh_Gp[v][k][0] + f1.y * sh_Gp[v][k][1]);
                }
                gp[k].x = gp[k].x - (sh_Kp[v][k][0] * f2.x + sh_Kp[v][k][1] * f3.x);
                gp[k].y = gp[k].y - (sh_Kp[v][k][0] * f2.y + sh_Kp[v][k][1] * f3.y);
            }
#pragma unroll
            for (int s = 0; s < kCallV; s++) {
                if (sh_w[s] < 0) continue;   // (uniform)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if (!sym) {
                        pw[s][q].x = pw[s][q].x - (f0.x * sh_Gw[v][s][q][0] + f1.x * sh_Gw[v][s][q][1]);
                        pw[s][q].y = pw[s][q].y - (f0.y * sh_Gw[v][s][q][0] + f1.y * sh_Gw[v][s][q][1]);
                    }
                    gw[s][q].x = gw[s][q].x - (sh_Kw[v][s][q][0] * f2.x + sh_Kw[v][s][q][1] * f3.x);
                    gw[s][q].y = gw[s][q].y - (sh_Kw[v][s][q][0] * f2.y + sh_Kw[v][s][q][1] * f3.y);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        *reinterpret_cast<double2_t*>(sp + (size_t)(4 * kCallV + k) * ld + r) = pp[k];
        *reinterpret_cast<double2_t*>(sp + (size_t)(4 * kCallV + 3 + k) * ld + r) = gp[k];
    }
#pragma unroll
    for (int s = 0; s < kCallV; s++) {
        if (sh_w[s] < 0) continue;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            *reinterpret_cast<double2_t*>(sp + (size_t)(4 * s + q) * ld + r) = pw[s][q];
            *reinterpret_cast<double2_t*>(sp + (size_t)(4 * s + 2 + q) * ld + r) = gw[s][q];
        }
    }
}

int spec_rows() { return kSpecRows; }

void launch_pool_step_spec(const PoolView& pv, const double* meas, const int* count, int jmax, const Pending& pend,
                           double* spec, int* specw, hipStream_t s) {
    hipLaunchKernelGGL(k_pool_step_spec, dim3((pv.ld / 2 + 255) / 256, pv.B), dim3(256), 0, s, pv, meas, count, jmax, pend.U,
                       pend.V, pend.count / 2, pend.cap, pend.symmetric != 0, spec, specw);
}

void launch_pool_step_unknown(const PoolView& pv, const double* meas, const int* count, int jmax, int min_active,
                              int* assoc_out, double* U, double* V, unsigned long long* corr_counter, hipStream_t s,
                              int* cnt_out, int zero_upto, double* blocks) {
    hipLaunchKernelGGL(k_pool_step_unknown<false>, dim3(pv.B), dim3(kStepThreads), 0, s, pv, meas, count, jmax, min_active,
                       assoc_out, U, V, corr_counter, cnt_out, zero_upto, 0, 2 * kCallV, blocks, nullptr, 0, nullptr, nullptr);
}

void launch_pool_step_unknown_delayed(const PoolView& pv, const double* meas, const int* count, int jmax, int min_active,
                                      int* assoc_out, const Pending& pend, unsigned long long* corr_counter, int* cnt_scratch,
                                      double* blocks, const double* pred, hipStream_t s, const double* spec, const int* specw) {
    hipLaunchKernelGGL(k_pool_step_unknown<true>, dim3(pv.B), dim3(kStepThreads), 0, s, pv, meas, count, jmax, min_active,
                       assoc_out, pend.U, pend.V, corr_counter, cnt_scratch, jmax, pend.count / 2, pend.cap, blocks, pred,
                       pend.symmetric != 0, spec, specw);
}

int step_pending_pairs_max() { return kStepPendingPairs; }

}  // namespace ekf
