// ekf_callfused.hip -- a whole measurement() call (ekf_slam.cpp:108-197) as TWO launches, whatever the number V of
// visible landmarks: Sigma is streamed ONCE per call instead of once per landmark, and the result is still
// bit-identical to the per-landmark path (same operations on every element, in the same order).
//
// Observation.  Correction v of a call updates Sigma <- Sigma - K_v G_v with K_v = Sigma H_v^T S_v^-1 (N x 2) and
// G_v = H_v Sigma (2 x N), all taken from the covariance AFTER corrections 0..v-1.  H_v has the five non-zero columns
// c5(v) = {0, 1, 2, 3+2i_v, 4+2i_v}.  Let C = {0,1,2} + {3+2i_v, 4+2i_v : v < V} be the "core" index set of the call
// (Nc = 3 + 2V indices).  Then
//   * H_v, S_v, nu_v depend only on the core block Sigma[C, C] and the core state -- a (3+2V)-dimensional filter that
//     is closed under the corrections of the call (its update needs K_v[C] and G_v[C], which come from the block itself);
//   * G_v[c] for ANY column c needs only column c of the row panel Sigma[C, :], and that column's update
//     Sigma[C, c] -= K_v[C] G_v[c] needs nothing else: the panel recursion is independent per column;
//   * K_v[r] for ANY row r needs only row r of the column panel Sigma[:, C], whose update
//     Sigma[r, C] -= K_v[r] G_v[C] is independent per row.
// So all V factor pairs (K_v, G_v) of a call follow from O(V^2 N) work on two thin panels, without touching the bulk
// of Sigma, and every other element then takes its V rank-2 updates in ONE read-modify-write:
//       x <- ((x - (K_0[r].G_0[c])) - (K_1[r].G_1[c])) - ...      -- the very sequence the per-landmark path applies.
//
// k_call_factors  grid (column/row slices of 256, filters).  Waves 0-2 of every workgroup run the core filter (the
//                 transcendental chain, lane-parallel: wave_terms) one correction AHEAD of the other waves (wave 1 updates the
//                 core block, the slice waves carry one panel column and one panel row per thread in registers and emit the
//                 factors U = K, Vf = G); the chain's loop is rolled so that its code stays in the instruction cache.
//                 The state update state += K_v nu_v (:186-187) is row-local and goes out of place (state_out).
// k_rank2v        the streaming pass: Sigma[r][c] -= sum_v (sequentially) K_v[r] G_v[c]; K through the scalar cache,
//                 G in registers, 16 N^2 bytes per CALL (chunks of kCallV corrections per pass).
#include "ekf_kernels.hpp"

namespace ekf {

constexpr int kNcMax = 3 + 2 * kCallV;
int rank2v_round_count(int vcount);

__device__ __forceinline__ void wave_sync_lds() {
    // one wavefront: DS instructions of a wave execute in order, so only the COMPILER must not reorder across this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// position of global index-5 entry k of correction v inside the core: {0, 1, 2, 3 + 2v, 4 + 2v}
__device__ __forceinline__ constexpr int core5(int k, int v) { return k < 3 ? k : 3 + 2 * v + (k - 3); }

// wave 0 = the core filter; SLICE slice threads (256 for pools; 64 for a single filter, whose row gathers -- one or two
// 64-B sectors per row and landmark, uncoalesced -- are then spread over 4x as many CUs)

// diagnostics: stamp slot k of row `who` (0: lane 0 of the control wave, 1: lane 0 of the first slice wave)
#define CF_TR(who, k)                                                                                              \
    do {                                                                                                           \
        if (src.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == kCtl * (who) && (k) < kTraceSlots)        \
            src.trace[(who) * kTraceSlots + (k)] = wall_clock64();                                                \
    } while (0)

constexpr int kCtl = 192;   // control threads: wave 0 geometry / S / gains, wave 1 angles, wave 2 the core block's update

template <int SLICE>
__global__ __launch_bounds__(kCtl + SLICE) void k_call_factors(PoolView pv, CallSrc src, double* __restrict__ Uall,
                                                                 double* __restrict__ Vall, int* __restrict__ cnt_out,
                                                                 double* __restrict__ state_out, int zero_upto) {
    constexpr int kFactorThreads = kCtl + SLICE;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int N = pv.N, ld = pv.ld;
    const double* __restrict__ Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    const double* __restrict__ st = pv.state + (size_t)b * ld;
    double* __restrict__ so = state_out + (size_t)b * ld;
    double* __restrict__ Ub = Uall + (size_t)b * 2 * kCallV * ld;
    double* __restrict__ Vb = Vall + (size_t)b * 2 * kCallV * ld;

    __shared__ int sh_lm[kCallV];
    __shared__ int sh_cidx[kNcMax + 1];   // global index of core position j (positions beyond the call's core: 0)
    __shared__ double sh_zs[kCallV][2];
    __shared__ double sh_pose[4];
    __shared__ double sh_Cb[kNcMax][kNcMax + 1];
    __shared__ double sh_sc[kNcMax + 1];
    __shared__ double sh_tv[kCallV][16];                 // H[10], S^-1[4], nu[2] per correction
    __shared__ double sh_Kc[kCallV][kNcMax][2];          // K_v on the core rows
    __shared__ double sh_Gc[kCallV][2][kNcMax + 1];      // G_v on the core columns
    __shared__ double sh_pr[8];                          // folded prediction: a10, a20, u0, u1, u2
    __shared__ int sh_done[4];                           // in-step hand-offs between the control waves (see below)

    CF_TR(0, 0); CF_TR(1, 0);
    if (src.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) src.trace[61] = __builtin_amdgcn_s_memtime();   // shader clock
    // ---- which corrections: [v0, v0 + vcount) of this filter's call ----
    if (tid < kCallV) {
        int lm = -1;
        double sx = 0.0, sy = 0.0;
        const int v = src.v0 + tid;
        if (tid < src.vcount) {
            if (src.mode == SRC_INLINE) {
                lm = src.inl_lm[tid];
                sx = src.inl_xy[tid][0];
                sy = src.inl_xy[tid][1];
            } else if (src.mode == SRC_SENSOR_VECTOR) {
                if (v < src.vlist[0]) {
                    lm = src.vlist[1 + v];
                    sx = src.sensor[(size_t)b * 2 * pv.n + 2 * lm];
                    sy = src.sensor[(size_t)b * 2 * pv.n + 2 * lm + 1];
                }
            } else if (v < src.vmax) {
                const size_t slot = (size_t)b * src.vmax + v;
                lm = src.lm_idx[slot];
                if (lm >= 0) { sx = src.z_xy[slot * 2]; sy = src.z_xy[slot * 2 + 1]; }
            }
        }
        if (lm >= pv.n) lm = -1;
        sh_lm[tid] = lm;
        sh_zs[tid][0] = sx; sh_zs[tid][1] = sy;
        sh_cidx[3 + 2 * tid] = lm >= 0 ? 3 + 2 * lm : 0;
        sh_cidx[4 + 2 * tid] = lm >= 0 ? 4 + 2 * lm : 0;
        if (tid < 3) sh_cidx[tid] = tid;
    }
    // The landmark list.  SRC_INLINE: it sits in the kernel arguments, so every wave has it (and the addresses of all its
    // gathers) at once, without a barrier in front of the loads; the other modes read it from memory through LDS.
    // Landmarks are listed in ascending order and -1 padded: the count is the leading run.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lmr[kCallV];
    int cnt = 0;
    if (src.mode == SRC_INLINE) {   // (uniform)
        bool run = true;
#pragma unroll
        for (int v = 0; v < kCallV; v++) {
            int lm = v < src.vcount ? src.inl_lm[v] : -1;
            if (lm >= pv.n) lm = -1;
            run = run && lm >= 0;
            if (run) cnt++;
            lmr[v] = run ? lm : 0;
        }
    } else {
        __syncthreads();
        const unsigned long long miss = __ballot(lane >= kCallV || sh_lm[lane < kCallV ? lane : 0] < 0);
        cnt = __builtin_ctzll(miss | (1ull << kCallV));
#pragma unroll
        for (int v = 0; v < kCallV; v++) lmr[v] = v < cnt ? sh_lm[v] : 0;
    }
    const int Nc = 3 + 2 * cnt;
    auto cidx = [&](int j) {   // global index of core position j (clamped to a valid one beyond Nc); j compile-time below
        return j < 3 ? j : 3 + 2 * lmr[(j - 3) >> 1] + ((j - 3) & 1);
    };

    // ---- requests: every wave issues its loads now; nothing below waits for another wave until barrier P1 ----
    const int s = tid - kCtl;
    const int i = blockIdx.x * SLICE + s;
    const bool slice = tid >= kCtl;
    const bool w0 = tid < 64;
    const bool live = slice && i < N;
    double Rcol[kNcMax], Crow[kNcMax], st_i = 0.0;
    if (slice) {
        // panels: slice thread s carries column i of Sigma[C, :] and row i of Sigma[:, C], i = slice base + s
        const int ic = live ? i : 0;
#pragma unroll
        for (int j = 0; j < kNcMax; j++)   // unconditional, clamped: all loads in flight together, one round trip
            Rcol[j] = Sg[(size_t)cidx(j) * ld + ic];   // coalesced across the slice
        {   // row i at the core columns: {0, 1}, {2}, and one 16-byte access per landmark (its two columns are neighbours)
            const double* rowi = Sg + (size_t)ic * ld;
            const D2u c01 = *reinterpret_cast<const D2u*>(rowi);
            Crow[0] = c01.x; Crow[1] = c01.y; Crow[2] = rowi[2];
#pragma unroll
            for (int v = 0; v < kCallV; v++) {
                const D2u cl = *reinterpret_cast<const D2u*>(rowi + cidx(3 + 2 * v));
                Crow[3 + 2 * v] = cl.x; Crow[4 + 2 * v] = cl.y;
            }
        }
        st_i = st[ic];
#pragma unroll
        for (int j = 0; j < kNcMax; j++)
            if (!live || j >= Nc) { Rcol[j] = 0.0; Crow[j] = 0.0; }
        if (!live) st_i = 0.0;
    } else {
#pragma unroll
        for (int j = 0; j < kNcMax; j++) { Rcol[j] = 0.0; Crow[j] = 0.0; }
    }
    if (wave == 0) {
        // ---- core block and core state: lane = (row group, column), all gathers issued before the first store.  (sh_cidx
        // was written by lanes of this very wave: DS instructions of a wave execute in order) ----
        wave_sync_lds();
        constexpr int kRounds = (kNcMax * kNcMax + 63) / 64;
        double cbv[kRounds];
#pragma unroll
        for (int q = 0; q < kRounds; q++) {
            const int e = lane + 64 * q;
            const int j = min(e / kNcMax, kNcMax - 1), c = e - (e / kNcMax) * kNcMax;   // compile-time divisor
            cbv[q] = Sg[(size_t)sh_cidx[j] * ld + sh_cidx[c]];   // positions beyond Nc are clamped copies, never used
        }
        const double scv = st[sh_cidx[lane < kNcMax ? lane : 0]];
#pragma unroll
        for (int q = 0; q < kRounds; q++) {
            const int e = lane + 64 * q;
            if (e < kNcMax * kNcMax) sh_Cb[e / kNcMax][e % kNcMax] = cbv[q];
        }
        if (lane < kNcMax) sh_sc[lane] = scv;
    } else if (wave == 1) {
        // the pose every correction of the call uses, captured ONCE (:109-111) -- with a folded prediction() the PREDICTED
        // pose (theta is not wrapped, :99).  Its sines and cosines only need theta: lanes 0 / 1 evaluate sin and cos of
        // theta and theta + dtheta (two calls deep instead of four) while the other waves' gathers are in flight.
        double pose_k = 0.0;
        if (lane < 3) pose_k = src.fresh_pose ? st[lane] : pv.snap[(size_t)b * 4 + lane];
        if (src.has_twist) {   // (uniform)
            const double th = st[0];
            const double arg = (lane & 1) ? th + src.dtheta : th;
            const double sv = sin(arg), cv = cos(arg);
            const double sin_t = lane_bcast(sv, 0), sin_td = lane_bcast(sv, 1), cos_t = lane_bcast(cv, 0), cos_td = lane_bcast(cv, 1);
            const double dtheta = src.dtheta, dx = src.dx;
            double u0, u1, u2, a10, a20;
            if (fabs(dtheta) < pv.p.straight_eps) {  // :79-86
                u0 = 0;
                u1 = dx * cos_t;
                u2 = dx * sin_t;
                a10 = -dx * sin_t;
                a20 = dx * cos_t;
            } else {  // :88-94
                u0 = dtheta;
                u1 = -(dx / dtheta) * sin_t + (dx / dtheta) * sin_td;
                u2 = (dx / dtheta) * cos_t - (dx / dtheta) * cos_td;
                a10 = -(dx / dtheta) * cos_t + (dx / dtheta) * cos_td;
                a20 = -(dx / dtheta) * sin_t + (dx / dtheta) * sin_td;
            }
            if (lane == 0) {
                sh_pr[0] = a10; sh_pr[1] = a20; sh_pr[2] = u0; sh_pr[3] = u1; sh_pr[4] = u2;
                if (blockIdx.x == 0) { src.pred_out[(size_t)b * 2] = a10; src.pred_out[(size_t)b * 2 + 1] = a20; }
            }
            pose_k = pose_k + (lane == 0 ? u0 : lane == 1 ? u1 : u2);
        }
        if (lane < 3) {
            sh_pose[lane] = pose_k;
            if (blockIdx.x == 0 && src.fresh_pose) pv.snap[(size_t)b * 4 + lane] = pose_k;
        }
    } else if (wave == 2) {
        // factors of unused core rows are exact zeros (the slices run their loops to kNcMax)
        for (int e = lane; e < kCallV * kNcMax; e += 64) { sh_Kc[e / kNcMax][e % kNcMax][0] = 0.0; sh_Kc[e / kNcMax][e % kNcMax][1] = 0.0; }
        for (int e = lane; e < kCallV * (kNcMax + 1); e += 64) { sh_Gc[e / (kNcMax + 1)][0][e % (kNcMax + 1)] = 0.0; sh_Gc[e / (kNcMax + 1)][1][e % (kNcMax + 1)] = 0.0; }
    }
    CF_TR(0, 1); CF_TR(1, 1);
    __syncthreads();   // P1: block, core state, pose, prediction terms and the landmark list are in LDS
    // bookkeeping (count for the streaming pass, correction record, touched set): the lanes of the LAST slice wave, in
    // parallel, off the control waves' critical path
    if (blockIdx.x == 0 && tid >= kFactorThreads - 64) {
        if (lane == 8) cnt_out[b] = cnt;
        if (lane == 10) {
            CorrRec rc;
            rc.nu0 = 0.0; rc.nu1 = 0.0; rc.active = cnt > 0; rc.lm = cnt > 0 ? sh_lm[cnt - 1] : -1; rc.n_active = 0; rc.pad = 0;
            pv.rec[b] = rc;
        }
        if (lane < cnt) {   // first touches: one landmark per lane (the order of the touch list is irrelevant)
            const int lm = sh_lm[lane];
            unsigned char* tf = pv.touch_flag + (size_t)b * pv.n;
            if (!tf[lm]) {
                tf[lm] = 1;
                const int slot = atomicAdd(&pv.touch_count[b], 1);
                pv.touch_list[(size_t)b * pv.n + slot] = lm;
            }
        }
    }
    const double theta = sh_pose[0], x = sh_pose[1], y = sh_pose[2];
    auto terms_geo = [&](int v) {
        wave_terms_geo(lane, sh_sc[3 + 2 * v], sh_sc[4 + 2 * v], sh_zs[v][0], sh_zs[v][1], x, y, &sh_tv[v][0], &sh_tv[v][14]);
    };
    auto terms_ang = [&](int v) {
        wave_terms_ang(lane, sh_sc[3 + 2 * v], sh_sc[4 + 2 * v], sh_zs[v][0], sh_zs[v][1], theta, x, y, true, &sh_tv[v][15]);
    };
    // Between P1 and P2: the state-only terms of correction 0 (waves 1 and 2: they need the pose and the landmark, not the
    // block) beside the prediction folded into what this workgroup holds (wave 0: the block; the slices: their panels).
    if (wave == 1 && cnt > 0) terms_ang(0);
    if (wave == 2 && cnt > 0) terms_geo(0);
    if (tid < 4) sh_done[tid] = tid == 2 ? 0 : 1;   // correction 0's terms are done at P2; no block update yet
    if (src.has_twist) {
        // ---- prediction(), :101-102 (the structured arithmetic of k_predict) ----
        const double a10 = sh_pr[0], a20 = sh_pr[1];
        if (w0) {
            // core block: rows / columns 1, 2 beyond the pose block, then the 3 x 3 block from its OLD values
            double c33[3][3];
            if (lane == 0)
                for (int r = 0; r < 3; r++)
                    for (int k = 0; k < 3; k++) c33[r][k] = sh_Cb[r][k];
            if (lane >= 3 && lane < Nc) {
                const int k = lane;
                const double q0 = sh_Cb[0][k], q1 = sh_Cb[1][k], q2 = sh_Cb[2][k];
                const double r0c = sh_Cb[k][0], r1c = sh_Cb[k][1], r2c = sh_Cb[k][2];
                sh_Cb[1][k] = a10 * q0 + q1;
                sh_Cb[2][k] = a20 * q0 + q2;
                sh_Cb[k][1] = r0c * a10 + r1c;
                sh_Cb[k][2] = r0c * a20 + r2c;
            }
            if (lane == 0) {
                double Tm[3][3];
                for (int k = 0; k < 3; k++) {
                    Tm[0][k] = c33[0][k];
                    Tm[1][k] = a10 * c33[0][k] + c33[1][k];
                    Tm[2][k] = a20 * c33[0][k] + c33[2][k];
                }
                for (int r = 0; r < 3; r++) {
                    sh_Cb[r][0] = Tm[r][0];
                    sh_Cb[r][1] = Tm[r][0] * a10 + Tm[r][1];
                    sh_Cb[r][2] = Tm[r][0] * a20 + Tm[r][2];
                }
                sh_Cb[0][0] += pv.p.q_pose;  // Q = diag(q,q,q,0...) :40-43
                sh_Cb[1][1] += pv.p.q_pose;
                sh_Cb[2][2] += pv.p.q_pose;
            }
            if (lane < 3) sh_sc[lane] = sh_sc[lane] + sh_pr[2 + lane];   // :99 (theta not wrapped)
        } else if (slice && live && i >= 3) {
            Rcol[1] = a10 * Rcol[0] + Rcol[1];   // rows 1, 2 of column i
            Rcol[2] = a20 * Rcol[0] + Rcol[2];
            Crow[1] = Crow[0] * a10 + Crow[1];   // columns 1, 2 of row i
            Crow[2] = Crow[0] * a20 + Crow[2];
        }
    }
    __syncthreads();   // P2: the block is predicted, correction 0's state-only terms are in LDS
    if (src.has_twist && live && i < 3) {   // pose rows / columns: they ARE rows / columns of the (now predicted) core block
#pragma unroll
        for (int j = 0; j < kNcMax; j++)
            if (j < Nc) { Rcol[j] = sh_Cb[j][i]; Crow[j] = sh_Cb[i][j]; }
        st_i = st_i + sh_pr[2 + i];
    }
    CF_TR(0, 2); CF_TR(1, 2);

    // ---- the core filter, pipelined over three wavefronts, and the slices ----
    // phase 2 of a correction on wave 0: S, S^-1 (lane-parallel, the arithmetic of wave_terms_s) from the block after
    // correction v - 1, then K_v on the core and the core STATE update.  Every LDS operand is requested up front; S^-1
    // goes to LDS for the slices and stays in registers for the gains (no write -> read round trip).
    // core_SK(v): S, S^-1, K_v (needs H_v, nu0 and the block after correction v - 1; NOT nu1).  Leaves K_v(lane, :) and the
    // lane's core state entry in ck0 / ck1 / cst for core_state(v), which needs the angle wave's nu1.
    double ck0 = 0.0, ck1 = 0.0, cst = 0.0;
    auto core_SK = [&](int v) {
        wave_sync_lds();
        const int ha = (lane / 5) & 1, hl = lane % 5;
        const int lc = lane < kNcMax ? lane : 0;
        double s5[5], H0[5], H1[5], p[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            s5[k] = sh_Cb[core5(k, v)][core5(hl, v)];
            H0[k] = sh_tv[v][k]; H1[k] = sh_tv[v][5 + k];
            p[k] = sh_Cb[lc][core5(k, v)];
        }
        cst = sh_sc[lc];
        double hs = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) hs += (ha ? H1[k] : H0[k]) * s5[k];
        const int sa = (lane >> 1) & 1, sb = lane & 1;
        double sv = 0.0;
#pragma unroll
        for (int l = 0; l < 5; l++) {
            const double h0l = lane_bcast(hs, l), h1l = lane_bcast(hs, 5 + l);
            sv += (sa ? h1l : h0l) * (sb ? H1[l] : H0[l]);
        }
        if (sa == sb) sv += pv.p.r_meas;
        const double S00 = lane_bcast(sv, 0), S01 = lane_bcast(sv, 1), S10 = lane_bcast(sv, 2), S11 = lane_bcast(sv, 3);
        const double det = S00 * S11 - S01 * S10;
        const int sq4 = lane & 3;
        const double si = (sq4 == 0 ? S11 : sq4 == 1 ? -S01 : sq4 == 2 ? -S10 : S00) / det;
        if (lane < 4) sh_tv[v][10 + lane] = si;
        const double Si0 = lane_bcast(si, 0), Si1 = lane_bcast(si, 1), Si2 = lane_bcast(si, 2), Si3 = lane_bcast(si, 3);
        double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sht0 += p[k] * H0[k];
            sht1 += p[k] * H1[k];
        }
        ck0 = sht0 * Si0 + sht1 * Si2;   // :178
        ck1 = sht0 * Si1 + sht1 * Si3;
        if (lane < Nc) {   // lane = core row
            sh_Kc[v][lane][0] = ck0;
            sh_Kc[v][lane][1] = ck1;
        }
    };
    auto core_state = [&](int v) {   // core state (:186-187)
        wave_sync_lds();
        const double nu0 = sh_tv[v][14], nu1 = sh_tv[v][15];
        if (lane < Nc) {
            double st_new = cst + (ck0 * nu0 + ck1 * nu1);
            if (lane == 0) st_new = normalize_angle(st_new);
            sh_sc[lane] = st_new;
        }
    };
    // Hand-offs between the three control waves inside a step go through counters in LDS (sh_done[0] angle terms,
    // [1] geometry terms, [2] block updates completed): a wave publishes after its LDS writes (DS instructions of a
    // wave execute in order) and a waiting wave polls; every spin is bounded.  Only barrier A, which the slices need too,
    // stays a workgroup barrier -- the slices pass ONE barrier per correction instead of two, and wave 0 no longer
    // waits for the angle wave before S, S^-1 and K (it needs nu1 only for the last ten instructions of a correction).
    // Memory order: the counter is published with a workgroup-scope RELEASE store and read with a workgroup-scope ACQUIRE
    // load -- what the memory model asks for, not the in-order execution of DS instructions (the compiler waits for the
    // wave's LDS writes in front of the store; that s_waitcnt was there already).
    auto publish = [&](int which, int value) {
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&sh_done[which], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // All waves of a workgroup are resident together, so the counter arrives unless a wave has died.  The spin is bounded
    // all the same; when the bound is hit the wave does NOT go on as if its operands were there: it raises the pool's
    // device error word (the host fails every later call of the handle with EKF_ERR_HIP) and only then leaves the loop, so
    // that the workgroup still reaches its barriers and the grid drains.
    auto await = [&](int which, int value) {
        bool arrived = false;
        for (int spin = 0; spin < (1 << 22); spin++) {
            if (__hip_atomic_load(&sh_done[which], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= value) { arrived = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!arrived && lane == 0) report_device_error(pv, kErrHandoffTimeout);
        __builtin_amdgcn_wave_barrier();
    };
    // G_v = H_v Sigma[c5(v), :] on the core columns: needs H and the updated block, not S^-1 -- wave 2, beside wave 0
    auto core_G = [&](int v) {
        wave_sync_lds();
        if (lane < Nc) {   // lane = core column
            double g0 = 0.0, g1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const double g = sh_Cb[core5(k, v)][lane];
                g0 += sh_tv[v][k] * g;
                g1 += sh_tv[v][5 + k] * g;
            }
            sh_Gc[v][0][lane] = g0;
            sh_Gc[v][1][lane] = g1;
        }
    };
    // Four roles, each with its own loop over the corrections.  Every wave passes the same workgroup barriers -- P1, P2
    // above, then A(t) in front of step t; inside a step the control waves hand over through the LDS counters.  Step t
    // works on correction t + 1 (its landmark is up to date: the core state was advanced with the gains of t) beside
    // correction t's updates:
    //   wave 0   geometry of t + 1 (ranges, the quotients of H, nu0: one sqrt and one division deep) -> [block of t ready]
    //            -> S, S^-1, K of t + 1 -> [nu1 ready] -> core state of t + 1
    //   wave 1   angles of t + 1 (the two atan2, the wraps, nu1) -- beside wave 0, not in front of it
    //   wave 2   the core block's own rank-2 update for correction t -> [H of t + 1 ready] -> G of t + 1 on the core columns
    //   slices   factors of correction t for their index, panels and state updated
    // (correction 0's state-only terms ran between P1 and P2, its geometry on wave 2.)
    // The loops of the control waves are ROLLED: their code is fetched once and then runs from the instruction cache.
    // (Unrolled eight times -- as the slices' loop must be, their panels live in statically indexed registers -- the
    // kernel was 17 k instructions of straight-line code that every workgroup fetched cold: 2.05 us per correction; rolled:
    // 1.70 us, the chain of one wavefront at ~9 cycles per dependent fp64 instruction; angles on their own wave: 1.52;
    // phase 2 trimmed: 1.44; one workgroup barrier per correction: 1.34.)
    if (wave == 0) {
        if (cnt > 0) { core_SK(0); core_state(0); }
#pragma nounroll
        for (int t = 0; t < cnt; t++) {
            __syncthreads();   // A(t)
            CF_TR(0, 3 + 5 * t);
            if (t + 1 < cnt) {
                terms_geo(t + 1);
                publish(1, t + 2);
                CF_TR(0, 4 + 5 * t);
                await(2, t + 1);          // the block after correction t
                CF_TR(0, 5 + 5 * t);
                core_SK(t + 1);
                await(0, t + 2);          // nu1 of correction t + 1
                core_state(t + 1);
                CF_TR(0, 6 + 5 * t);
            }
        }
    } else if (wave == 1) {
#pragma nounroll
        for (int t = 0; t < cnt; t++) {
            __syncthreads();   // A(t)
            if (t + 1 < cnt) {
                terms_ang(t + 1);
                publish(0, t + 2);
            }
        }
    } else if (wave == 2) {
        if (cnt > 0) core_G(0);
#pragma nounroll
        for (int t = 0; t < cnt; t++) {
            __syncthreads();   // A(t)
            // the core block's own rank-2 update (:191-192); compile-time divisor, positions beyond Nc hold zero factors
#pragma unroll
            for (int q = 0; q < (kNcMax * kNcMax + 63) / 64; q++) {
                const int e = lane + 64 * q;
                const int j = e / kNcMax, c = e - j * kNcMax;
                if (j < Nc && c < Nc)
                    sh_Cb[j][c] = sh_Cb[j][c] - (sh_Kc[t][j][0] * sh_Gc[t][0][c] + sh_Kc[t][j][1] * sh_Gc[t][1][c]);
            }
            publish(2, t + 1);
            if (t + 1 < cnt) { await(1, t + 2); core_G(t + 1); }   // (H of correction t + 1 from wave 0)
        }
    } else {
#pragma unroll
        for (int t = 0; t < kCallV; t++) {
          if (t < cnt) {   // uniform (no break: the loop must unroll so that the panel registers are indexed statically)
            __syncthreads();
            CF_TR(1, 3 + 5 * t);
            const int v = t;
            double H0[5], H1[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { H0[k] = sh_tv[v][k]; H1[k] = sh_tv[v][5 + k]; }
            // G_v[i] = H_v Sigma[c5(v), i]   (rows of H*Sigma, ekf_slam.cpp:178)
            double g0 = 0.0, g1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const double g = Rcol[core5(k, v)];
                g0 += H0[k] * g;
                g1 += H1[k] * g;
            }
            // K_v[i] = Sigma[i, c5(v)] H_v^T S_v^-1
            double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const double p = Crow[core5(k, v)];
                sht0 += p * H0[k];
                sht1 += p * H1[k];
            }
            double k0 = sht0 * sh_tv[v][10] + sht1 * sh_tv[v][12];
            double k1 = sht0 * sh_tv[v][11] + sht1 * sh_tv[v][13];
            if (!live) { g0 = 0.0; g1 = 0.0; k0 = 0.0; k1 = 0.0; }   // pad entries of the factors stay exact zeros
            if (i < ld) {
                Vb[(size_t)(2 * v) * ld + i] = g0;
                Vb[(size_t)(2 * v + 1) * ld + i] = g1;
                Ub[(size_t)(2 * v) * ld + i] = k0;
                Ub[(size_t)(2 * v + 1) * ld + i] = k1;
            }
            // the panels take the correction (same expression as every other element of Sigma)
#pragma unroll
            for (int j = 0; j < kNcMax; j++) {
                Rcol[j] = Rcol[j] - (sh_Kc[v][j][0] * g0 + sh_Kc[v][j][1] * g1);
                Crow[j] = Crow[j] - (k0 * sh_Gc[v][0][j] + k1 * sh_Gc[v][1][j]);
            }
            st_i = st_i + (k0 * sh_tv[v][14] + k1 * sh_tv[v][15]);   // :186
            if (i == 0) st_i = normalize_angle(st_i);                 // :187
            CF_TR(1, 4 + 5 * t);
          }
        }
    }
    CF_TR(0, 60); CF_TR(1, 60);
    if (src.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) src.trace[62] = __builtin_amdgcn_s_memtime();
    if (slice && i < ld) {
        so[i] = live ? st_i : 0.0;
        for (int v = cnt; v < zero_upto; v++) {   // rows of the pass this filter does not use: exact no-ops for k_rank2v
            Vb[(size_t)(2 * v) * ld + i] = 0.0; Vb[(size_t)(2 * v + 1) * ld + i] = 0.0;
            Ub[(size_t)(2 * v) * ld + i] = 0.0; Ub[(size_t)(2 * v + 1) * ld + i] = 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming pass: every element takes the corrections of the call in order.  Same tiling and pipeline as k_rank2
// (strip of 256 double2 columns; rows in groups of UR through a ring of three register buffers, the loads of group g+1
// issued before the stores of group g, no branch in the main loop); a lane keeps G_v of its two columns for the CNT
// corrections of the pass in registers, K_v(r, :) is wave-uniform (scalar loads).  CNT is a template parameter (the
// pass's correction count rounded up; k_call_factors zero-fills the factor rows a filter does not use, and a zero
// factor is an exact no-op).  grid (strips, row blocks, B).
// ---------------------------------------------------------------------------------------------
// PRED: a prediction() is folded in front of the corrections (single filter): element (r, c) first becomes
// (At Sigma At^T + Q)(r, c) -- rows 1, 2 take a_r * row 0, columns 1, 2 take column 0 * a_c, the 3 x 3 pose block its own
// formula (the structured arithmetic of k_predict, operation for operation) -- all from values this lane or its wave
// neighbours hold: row 0 sits in the same row group as rows 1, 2, column 0 in lane 0 of strip 0.
// KL: the K values of the workgroup's row block (<= 128 rows x CNT pairs, 16 KB) are staged once in LDS and read from there
// (one uniform-address ds_read_b128 per row and pair) instead of through the scalar cache.  Pools with four or more
// corrections per pass: a row needs 2 CNT scalars from 2 CNT different cache lines, scalar loads return out of order (one
// batch in flight, ~1600 cycles under a saturated L2) and the pass was bound by that latency -- 14.3 ms per pass of 8
// corrections at B = 512, n = 1000 against 5.2 ms of stream time.
template <int UR, bool NT, int CNT, bool PRED, bool KL = false>
__global__ __launch_bounds__(256) void k_rank2v(double* __restrict__ sigma, const double* __restrict__ Uall,
                                                const double* __restrict__ Vall, const int* __restrict__ cnt_all, int N,
                                                int ld, size_t sigma_stride, int rows_per_block,
                                                const double* __restrict__ pred, double q_pose) {
    const int b = blockIdx.z;
    if (cnt_all[b] <= 0) return;   // this filter has nothing to correct in this pass
    double a10 = 0.0, a20 = 0.0;
    if constexpr (PRED) { a10 = pred[(size_t)b * 2]; a20 = pred[(size_t)b * 2 + 1]; }
    const int ld2n = ld >> 1, ld2a = (N + 1) >> 1;
    const int c2 = blockIdx.x * 256 + threadIdx.x;
    const int row_begin = blockIdx.y * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);
    if (row_begin >= N) return;   // (uniform)
    const double* __restrict__ Ub = Uall + (size_t)b * 2 * kCallV * ld;
    __shared__ double2_t sh_Kl[KL ? 128 : 1][CNT];
    const double2_t* __restrict__ Vb2 = reinterpret_cast<const double2_t*>(Vall + (size_t)b * 2 * kCallV * ld);
    double2_t* __restrict__ col = reinterpret_cast<double2_t*>(sigma + (size_t)b * sigma_stride) + c2;
    auto load_group = [&](double2_t (&buf)[UR], int row) {
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const double2_t* p = col + (size_t)(row + u) * ld2n;
            if constexpr (NT) buf[u] = __builtin_nontemporal_load(p);
            else buf[u] = *p;
        }
    };
    const int nfull = (row_end - row_begin) / UR;  // uniform
    const bool col_live = c2 < ld2a;
    double2_t A[UR];
    double2_t g0[CNT], g1[CNT];
    if constexpr (KL) {
        // A workgroup lives for ~20 us; its K values (global -> LDS -> barrier), its G values and its first row group are
        // three round trips if taken one after the other -- a fifth of its life.  All three are requested at once.
        constexpr int kPer = (128 * CNT + 255) / 256;   // K entries staged per thread (rows fastest: consecutive
        const int nrows = row_end - row_begin;          // threads read consecutive rows of one K vector); <= 128 rows
        double2_t kst[kPer];
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int e = threadIdx.x + 256 * q;
            const int v = min(e / nrows, CNT - 1), rr = e - (e / nrows) * nrows;
            kst[q] = double2_t{Ub[(size_t)(2 * v) * ld + row_begin + rr], Ub[(size_t)(2 * v + 1) * ld + row_begin + rr]};
        }
        if (col_live) {
            if (nfull > 0) load_group(A, row_begin);
#pragma unroll
            for (int v = 0; v < CNT; v++) { g0[v] = Vb2[(size_t)(2 * v) * ld2n + c2]; g1[v] = Vb2[(size_t)(2 * v + 1) * ld2n + c2]; }
        }
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int e = threadIdx.x + 256 * q;
            if (e < nrows * CNT) sh_Kl[e - (e / nrows) * nrows][e / nrows] = kst[q];
        }
        __syncthreads();
        if (!col_live) return;
    } else {
        if (!col_live) return;
#pragma unroll
        for (int v = 0; v < CNT; v++) { g0[v] = Vb2[(size_t)(2 * v) * ld2n + c2]; g1[v] = Vb2[(size_t)(2 * v + 1) * ld2n + c2]; }
    }

    auto apply = [&](double2_t& x, int row) {
#pragma unroll
        for (int v = 0; v < CNT; v++) {
            double k0, k1;
            if constexpr (KL) { const double2_t kk = sh_Kl[row - row_begin][v]; k0 = kk.x; k1 = kk.y; }
            else { k0 = Ub[(size_t)(2 * v) * ld + row]; k1 = Ub[(size_t)(2 * v + 1) * ld + row]; }   // uniform -> s_load
            x.x = x.x - (k0 * g0[v].x + k1 * g1[v].x);
            x.y = x.y - (k0 * g0[v].y + k1 * g1[v].y);
        }
    };
    // prediction of one row group held in registers (PRED only).  Strip 0: lane 0 holds columns (0, 1), lane 1 (2, 3).
    auto predict_group = [&](double2_t (&buf)[UR], int row) {
        if constexpr (PRED) {
            const bool strip0 = blockIdx.x == 0 && threadIdx.x < 64;   // the wavefront that holds columns 0..3 (uniform)
            if (row == 0) {   // (uniform) the group with rows 0, 1, 2: UR >= 4
                const double2_t o0 = buf[0], o1 = buf[1], o2 = buf[2];
                // rows 1, 2 beyond the pose block: a_r * row 0 + row r   (k_predict: a10 * s0 + s1)
                buf[1].x = a10 * o0.x + o1.x; buf[1].y = a10 * o0.y + o1.y;
                buf[2].x = a20 * o0.x + o2.x; buf[2].y = a20 * o0.y + o2.y;
                if (strip0) {
                    // the 3 x 3 block from its OLD values: c[r][k], columns 0, 1 in lane 0, column 2 in lane 1
                    double c[3][3];
                    c[0][0] = __shfl(o0.x, 0, kWave); c[0][1] = __shfl(o0.y, 0, kWave); c[0][2] = __shfl(o0.x, 1, kWave);
                    c[1][0] = __shfl(o1.x, 0, kWave); c[1][1] = __shfl(o1.y, 0, kWave); c[1][2] = __shfl(o1.x, 1, kWave);
                    c[2][0] = __shfl(o2.x, 0, kWave); c[2][1] = __shfl(o2.y, 0, kWave); c[2][2] = __shfl(o2.x, 1, kWave);
                    double T[3][3], Sn[3][3];
                    for (int k = 0; k < 3; k++) {
                        T[0][k] = c[0][k];
                        T[1][k] = a10 * c[0][k] + c[1][k];
                        T[2][k] = a20 * c[0][k] + c[2][k];
                    }
                    for (int r = 0; r < 3; r++) {
                        Sn[r][0] = T[r][0];
                        Sn[r][1] = T[r][0] * a10 + T[r][1];
                        Sn[r][2] = T[r][0] * a20 + T[r][2];
                    }
                    Sn[0][0] += q_pose; Sn[1][1] += q_pose; Sn[2][2] += q_pose;
                    if (c2 == 0) {
                        buf[0] = double2_t{Sn[0][0], Sn[0][1]}; buf[1] = double2_t{Sn[1][0], Sn[1][1]}; buf[2] = double2_t{Sn[2][0], Sn[2][1]};
                    } else if (c2 == 1) {   // column 2 from the block; column 3 of rows 1, 2 was set above, row 0 is unchanged
                        buf[0].x = Sn[0][2]; buf[1].x = Sn[1][2]; buf[2].x = Sn[2][2];
                    }
                }
            }
            if (strip0) {   // columns 1, 2 of the rows beyond the pose block: column 0 * a_c + column c   (k_predict: r0 * a10 + r1)
#pragma unroll
                for (int u = 0; u < UR; u++) {
                    const double col0 = __shfl(buf[u].x, 0, kWave);   // the OLD column-0 entry (the prediction leaves it alone)
                    if (row + u >= 3) {
                        if (c2 == 0) buf[u].y = buf[u].x * a10 + buf[u].y;
                        else if (c2 == 1) buf[u].x = col0 * a20 + buf[u].x;
                    }
                }
            }
        }
    };
    // corrections outermost, rows innermost: for one correction the UR rows x 2 columns of a lane are 2 UR independent
    // short chains, so the fp64 pipe always has an instruction that does not wait for the previous one (rows outermost,
    // one element runs its CNT dependent updates back to back: PMC showed the waves of an 8-correction pass stalled on
    // instruction issue for half of their cycles).  Per element the corrections are still applied in order.
    auto finish_group = [&](double2_t (&buf)[UR], int row) {
        predict_group(buf, row);
#pragma unroll
        for (int v = 0; v < CNT; v++) {
#pragma unroll
            for (int u = 0; u < UR; u++) {
                double k0, k1;
                if constexpr (KL) { const double2_t kk = sh_Kl[row + u - row_begin][v]; k0 = kk.x; k1 = kk.y; }
                else { k0 = Ub[(size_t)(2 * v) * ld + row + u]; k1 = Ub[(size_t)(2 * v + 1) * ld + row + u]; }   // uniform -> s_load
                buf[u].x = buf[u].x - (k0 * g0[v].x + k1 * g1[v].x);
                buf[u].y = buf[u].y - (k0 * g0[v].y + k1 * g1[v].y);
            }
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            double2_t* p = col + (size_t)(row + u) * ld2n;
            if constexpr (NT) __builtin_nontemporal_store(buf[u], p);
            else *p = buf[u];
        }
    };

    int r = row_begin;
    if (nfull > 0) {
        double2_t Bf[UR], Cf[UR];
        if constexpr (!KL) load_group(A, r);
        int g = 0;
        for (; g + 3 < nfull; g += 3) {  // invariant: A holds group g; no branch inside
            load_group(Bf, r + UR);
            finish_group(A, r);
            load_group(Cf, r + 2 * UR);
            finish_group(Bf, r + UR);
            load_group(A, r + 3 * UR);
            finish_group(Cf, r + 2 * UR);
            r += 3 * UR;
        }
        const int rem = nfull - g;  // 1, 2 or 3 groups left, A already loaded
        if (rem == 1) {
            finish_group(A, r);
        } else if (rem == 2) {
            load_group(Bf, r + UR);
            finish_group(A, r);
            finish_group(Bf, r + UR);
        } else {
            load_group(Bf, r + UR);
            finish_group(A, r);
            load_group(Cf, r + 2 * UR);
            finish_group(Bf, r + UR);
            finish_group(Cf, r + 2 * UR);
        }
        r += rem * UR;
    }
    for (; r < row_end; r++) {   // < UR leftover rows of the last row block (never rows 0..2: N >= UR)
        double2_t x = col[(size_t)r * ld2n];
        if constexpr (PRED) {
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                const double col0 = __shfl(x.x, 0, kWave);
                if (c2 == 0) x.y = x.x * a10 + x.y;
                else if (c2 == 1) x.x = col0 * a20 + x.x;
            }
        }
        apply(x, r);
        col[(size_t)r * ld2n] = x;
    }
}

// test hook: what await() does when its bound is hit
__global__ void k_raise_device_error(PoolView pv) { report_device_error(pv, kErrHandoffTimeout); }
void launch_raise_device_error(const PoolView& pv, hipStream_t s) {
    hipLaunchKernelGGL(k_raise_device_error, dim3(1), dim3(1), 0, s, pv);
}

void launch_call_factors(const PoolView& pv, const CallSrc& src, double* U, double* V, int* cnt, double* state_out,
                         hipStream_t s) {
    if ((long long)pv.B * ((pv.ld + 255) / 256) >= 128)
        hipLaunchKernelGGL(k_call_factors<256>, dim3((pv.ld + 255) / 256, pv.B), dim3(kCtl + 256), 0, s, pv, src, U, V, cnt,
                           state_out, rank2v_round_count(src.vcount));
    else
        hipLaunchKernelGGL(k_call_factors<64>, dim3((pv.ld + 63) / 64, pv.B), dim3(kCtl + 64), 0, s, pv, src, U, V, cnt,
                           state_out, rank2v_round_count(src.vcount));
}

template <int UR, int CNT>
static void launch_rank2v_c(const PoolView& pv, const double* U, const double* V, const int* cnt, bool nt, int rows, hipStream_t s,
                            const double* pred) {
    const int ld2a = (pv.N + 1) / 2;
    dim3 grid((ld2a + 255) / 256, (pv.N + rows - 1) / rows, pv.B);
#define EKF_R2V_ARGS pv.sigma, U, V, cnt, pv.N, pv.ld, pv.sigma_stride, rows, pred, pv.p.q_pose
    if (pred) {   // single filter with a folded prediction (small pool: temporal accesses)
        hipLaunchKernelGGL((k_rank2v<UR, false, CNT, true>), grid, dim3(256), 0, s, EKF_R2V_ARGS);
    } else if (CNT >= 4 && rows >= 32 && rows <= 128) {   // (big pools: rows >= 32) K through LDS
        if (nt) hipLaunchKernelGGL((k_rank2v<UR, true, CNT, false, true>), grid, dim3(256), 0, s, EKF_R2V_ARGS);
        else hipLaunchKernelGGL((k_rank2v<UR, false, CNT, false, true>), grid, dim3(256), 0, s, EKF_R2V_ARGS);
    } else if (nt) {
        hipLaunchKernelGGL((k_rank2v<UR, true, CNT, false>), grid, dim3(256), 0, s, EKF_R2V_ARGS);
    } else {
        hipLaunchKernelGGL((k_rank2v<UR, false, CNT, false>), grid, dim3(256), 0, s, EKF_R2V_ARGS);
    }
#undef EKF_R2V_ARGS
}

// corrections per pass are rounded up to an instantiated count; k_call_factors zero-fills up to it
int rank2v_round_count(int vcount) { return vcount <= 1 ? 1 : vcount <= 2 ? 2 : vcount <= 4 ? 4 : vcount <= 6 ? 6 : kCallV; }

void launch_rank2v(const PoolView& pv, const double* U, const double* V, const int* cnt, int vcount, const Rank2Tuning& t,
                   hipStream_t s, const double* pred) {
    const size_t pool_bytes = (size_t)pv.B * pv.N * ((size_t)pv.N * sizeof(double));
    const bool nt = t.nontemporal >= 0 ? t.nontemporal != 0 : pool_bytes > ((size_t)192 << 20);
    const int ld2a = (pv.N + 1) / 2;
    const long long strips = (long long)pv.B * ((ld2a + 255) / 256);
    // measured (tools/callfused_sweep.py, B = 4096, n = 1000, V = 2): 64 rows per workgroup in 16-row groups streams at
    // 6.47 TB/s (32 rows: 6.19) -- a workgroup first fetches its lanes' G values, which more rows amortise
    int rows = t.rows_per_block > 0 ? t.rows_per_block
                                    : (strips * pv.N >= 256LL * 8 * 64 ? 64 : (strips * pv.N >= 256LL * 4 * 8 ? 8 : 4));
    if (pred && rows < 4) rows = 4;   // a folded prediction needs rows 0..2 inside one full row group (UR >= 4)
    const int c = rank2v_round_count(vcount);
    // (with >= 4 corrections per pass a workgroup holds 2 waves/SIMD: 32 rows per workgroup stream at 5.76 TB/s, 64 at
    // 5.52, 128 at 5.28 -- bench.py unknown_association_large_prefix --rows R)
    if (c >= 4 && rows == 64 && t.rows_per_block <= 0) rows = 32;
    const bool big = rows >= 32;
    const bool u16 = big && t.group_rows != 8;
    const bool u8 = big && t.group_rows != 4;   // (passes of >= 4 corrections: 8-row groups unless the tuning asks for 4)
    switch (c) {
        case 1: u16 ? launch_rank2v_c<16, 1>(pv, U, V, cnt, nt, rows, s, pred) : big ? launch_rank2v_c<8, 1>(pv, U, V, cnt, nt, rows, s, pred) : launch_rank2v_c<4, 1>(pv, U, V, cnt, nt, rows, s, pred); break;
        case 2: u16 ? launch_rank2v_c<16, 2>(pv, U, V, cnt, nt, rows, s, pred) : big ? launch_rank2v_c<8, 2>(pv, U, V, cnt, nt, rows, s, pred) : launch_rank2v_c<4, 2>(pv, U, V, cnt, nt, rows, s, pred); break;
        case 4: u8 ? launch_rank2v_c<8, 4>(pv, U, V, cnt, nt, rows, s, pred) : launch_rank2v_c<4, 4>(pv, U, V, cnt, nt, rows, s, pred); break;
        case 6: u8 ? launch_rank2v_c<8, 6>(pv, U, V, cnt, nt, rows, s, pred) : launch_rank2v_c<4, 6>(pv, U, V, cnt, nt, rows, s, pred); break;
        default: u8 ? launch_rank2v_c<8, kCallV>(pv, U, V, cnt, nt, rows, s, pred) : launch_rank2v_c<4, kCallV>(pv, U, V, cnt, nt, rows, s, pred); break;
    }
}

}  // namespace ekf
