// ekf_small.hip -- LDS-resident kernels for small maps and small discovered prefixes:
//   k_small_measure_inline  one prediction() + measurement() tick of one filter, inputs by value (single-filter API)
//   k_small_associate   one data_association() call of one filter                  (single-filter API)
//   k_pool_associate    one STEP of data_association() for every filter of a pool  (ekf_batch_run_unknown)
//   k_pool_run_known    a whole RUN of prediction() + measurement() per filter     (ekf_batch_run_known)
//
// At the reference's own operating point (n = 20 landmarks, N = 43, nuslam/src/slam.cpp:250) the whole
// covariance is 15 KB: streaming it through HBM with two launches per visible landmark is all launch
// latency.  Here ONE workgroup loads Sigma and the state into LDS, runs the entire measurement() call --
// pose capture, first-call landmark initialisation, and every visible landmark's correction in ascending
// order (ekf_slam.cpp:108-197) -- with workgroup barriers between the phases, and writes both back:
// one launch per API call instead of 2V + 1.  The arithmetic (gather order, summation order, update
// expression) is that of k_gain / k_rank2, so the result is bit-identical to the multi-kernel path.
// LDS rows use an odd stride so that the column gather Sigma(r, c5) is bank-conflict-free.
#include "ekf_kernels.hpp"

#include <climits>

namespace ekf {


// First touches are recorded in LDS while the corrections run and written out once at the end, one landmark per
// lane: touch_landmark()'s global read-modify-write would sit on the single lane's critical path of every correction.
// (The order of a filter's touch list is irrelevant: k_rank2_active treats its rows independently.)
__device__ __forceinline__ void flush_touches(const PoolView& pv, int b, const unsigned char* sh_touch, int n, int nthreads) {
    unsigned char* tf = pv.touch_flag + (size_t)b * pv.n;
    for (int i = threadIdx.x; i < n; i += nthreads)
        if (sh_touch[i] && !tf[i]) {
            tf[i] = 1;
            const int slot = atomicAdd(&pv.touch_count[b], 1);
            pv.touch_list[(size_t)b * pv.n + slot] = i;
        }
}

// prediction() on the LDS image (ekf_slam.cpp:55-106): the structured arithmetic of k_predict, operation for
// operation.  Every thread of the workgroup calls it; the image must be complete on entry (barrier before), and is
// consistent on return (barrier inside).
template <int THREADS>
__device__ __forceinline__ void small_predict(double* S, double* st, int ldS, int N, double dtheta, double dx,
                                              const Params& prm) {
    const int tid = threadIdx.x;
    const double theta = st[0];
    double u0, u1, u2, a10, a20;
    if (fabs(dtheta) < prm.straight_eps) {  // :79-86
        u0 = 0;
        u1 = dx * cos(theta);
        u2 = dx * sin(theta);
        a10 = -dx * sin(theta);
        a20 = dx * cos(theta);
    } else {  // :88-94
        u0 = dtheta;
        u1 = -(dx / dtheta) * sin(theta) + (dx / dtheta) * sin(theta + dtheta);
        u2 = (dx / dtheta) * cos(theta) - (dx / dtheta) * cos(theta + dtheta);
        a10 = -(dx / dtheta) * cos(theta) + (dx / dtheta) * cos(theta + dtheta);
        a20 = -(dx / dtheta) * sin(theta) + (dx / dtheta) * sin(theta + dtheta);
    }
    double c[3][3], px = 0.0, py = 0.0;
    if (tid == 0) {
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) c[r][k] = S[r * ldS + k];
        px = st[1]; py = st[2];
    }
    __syncthreads();  // all threads hold the old theta; thread 0 may now move the pose
    for (int k = 3 + tid; k < N; k += THREADS) {
        const double q0 = S[0 * ldS + k], q1 = S[1 * ldS + k], q2 = S[2 * ldS + k];
        double* rowk = S + k * ldS;
        const double r0 = rowk[0], r1 = rowk[1], r2 = rowk[2];
        S[1 * ldS + k] = a10 * q0 + q1;
        S[2 * ldS + k] = a20 * q0 + q2;
        rowk[1] = r0 * a10 + r1;
        rowk[2] = r0 * a20 + r2;
    }
    if (tid == 0) {
        st[0] = theta + u0;  // :99 -- theta is NOT wrapped after the prediction
        st[1] = px + u1;
        st[2] = py + u2;
        double T[3][3];
        for (int k = 0; k < 3; k++) {
            T[0][k] = c[0][k];
            T[1][k] = a10 * c[0][k] + c[1][k];
            T[2][k] = a20 * c[0][k] + c[2][k];
        }
        for (int r = 0; r < 3; r++) {
            S[r * ldS + 0] = T[r][0];
            S[r * ldS + 1] = T[r][0] * a10 + T[r][1];
            S[r * ldS + 2] = T[r][0] * a20 + T[r][2];
        }
        S[0] += prm.q_pose;  // Q = diag(q,q,q,0...) :40-43
        S[1 * ldS + 1] += prm.q_pose;
        S[2 * ldS + 2] += prm.q_pose;
    }
    __syncthreads();
}


// has_twist: a prediction(dtheta, dx) the host deferred into this launch runs first, on the LDS image.
template <int THREADS>
__device__ __forceinline__ void small_measure_body(const PoolView& pv, int b, const double* __restrict__ sens,
                                                   const unsigned char* __restrict__ vis, int do_init, int has_twist,
                                                   double dtheta, double dx, double* sm) {
    const int tid = threadIdx.x;
    const int N = pv.N, ld = pv.ld, n = pv.n;
    const int ldS = N | 1;
    double* S = sm;            // [N][ldS]
    double* st = S + (size_t)N * ldS;  // [N]
    double* Gg = st + N;       // [2][N]
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];

    double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    double* stg = pv.state + (size_t)b * ld;
    for (int r = tid >> 6; r < N; r += THREADS / 64)
        for (int c = tid & 63; c < N; c += 64) S[r * ldS + c] = Sg[(size_t)r * ld + c];
    for (int r = tid; r < N; r += THREADS) st[r] = stg[r];
    __syncthreads();
    if (has_twist) small_predict<THREADS>(S, st, ldS, N, dtheta, dx, pv.p);  // uniform

    const double theta = st[0], x = st[1], y = st[2];  // captured ONCE, ekf_slam.cpp:109-111
    if (do_init) {                                     // :113-128, all n landmarks regardless of visibility
        __syncthreads();                               // everybody holds the pose before the map is rewritten
        for (int i = tid; i < n; i += THREADS) {
            const double sx = sens[2 * i], sy = sens[2 * i + 1];
            const double ri = sqrt(sx * sx + sy * sy);
            const double phii = atan2(sy, sx);
            st[2 * i + 3] = x + ri * cos(phii + theta);
            st[2 * i + 3 + 1] = y + ri * sin(phii + theta);
        }
        __syncthreads();
    }

    // (r, phi) of every visible reading at once, one landmark per lane (:142-146); n <= 50 on this path
    __shared__ double sh_zr[64], sh_zp[64];
    __shared__ unsigned char sh_touch[64];
    if (tid < 64) sh_touch[tid] = 0;
    for (int i = tid; i < n; i += THREADS)
        if (vis[i]) {
            const double sx = sens[2 * i], sy = sens[2 * i + 1];
            sh_zr[i] = sqrt(sx * sx + sy * sy);
            sh_zp[i] = atan2(sy, sx);
        }
    __syncthreads();
    for (int lm = 0; lm < n; lm++) {  // :132-194 (uniform loop: every lane sees the same visible[] byte)
        if (!vis[lm]) continue;
        if (tid == 0) {
            MeasTerms m;
            m.z0 = sh_zr[lm]; m.z1 = sh_zp[lm];
            predicted_terms(st[2 * lm + 3], st[2 * lm + 4], theta, x, y, m);
            double S55[5][5], Sm[2][2], Si[2][2];
            for (int k = 0; k < 5; k++)
                for (int l = 0; l < 5; l++) S55[k][l] = S[idx5(k, lm) * ldS + idx5(l, lm)];
            innovation_cov(S55, m.H, pv.p.r_meas, Sm);
            inv2(Sm, Si);
            for (int a = 0; a < 2; a++)
                for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = m.H[a][k];
            sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
            sh_nu[0] = m.z0 - m.zh0;                   // :182
            sh_nu[1] = normalize_angle(m.z1 - m.zh1);  // :183
            sh_touch[lm] = 1;
        }
        __syncthreads();
        double k0 = 0.0, k1 = 0.0;
        const int r = tid;
        if (r < N) {
            double sht0 = 0.0, sht1 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int c = idx5(k, lm);
                const double p = S[r * ldS + c];
                const double g = S[c * ldS + r];
                sht0 += p * sh_H[k];
                sht1 += p * sh_H[5 + k];
                g0 += sh_H[k] * g;
                g1 += sh_H[5 + k] * g;
            }
            k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];  // :178
            k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
            Gg[r] = g0;
            Gg[N + r] = g1;
        }
        __syncthreads();
        // Sigma <- (I - K H) Sigma (:191-192): lane tid owns row tid's K, so rows are walked by their owner
        if (r < N) {
            // (row and the two G rows never overlap: lets the LDS reads of later columns start before earlier writes)
            double* __restrict__ row = S + r * ldS;
            const double* __restrict__ G0 = Gg;
            const double* __restrict__ G1 = Gg + N;
#pragma unroll 4
            for (int c = 0; c < N; c++) row[c] = row[c] - (k0 * G0[c] + k1 * G1[c]);
            double s = st[r] + (k0 * sh_nu[0] + k1 * sh_nu[1]);  // :186
            if (r == 0) s = normalize_angle(s);                  // :187
            st[r] = s;
        }
        __syncthreads();
    }

    for (int r = tid >> 6; r < N; r += THREADS / 64)
        for (int c = tid & 63; c < N; c += 64) Sg[(size_t)r * ld + c] = S[r * ldS + c];
    for (int r = tid; r < N; r += THREADS) stg[r] = st[r];
    flush_touches(pv, b, sh_touch, n, THREADS);
}

// One filter, inputs by value: `in` lives in the kernel-argument segment the launch packet already carries.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_small_measure_inline(PoolView pv, SmallInline in, int do_init,
                                                                  int has_twist, double dtheta, double dx) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    small_measure_body<THREADS>(pv, 0, in.sensor, in.visible, do_init, has_twist, dtheta, dx, sm);
}

// ---------------------------------------------------------------------------------------------
// data_association() of a small map in ONE launch (ekf_slam.cpp:278-402): for every measurement, in order,
// the Mahalanobis scores of the known landmarks (one landmark per LANE here -- everything a score needs is in
// LDS, and innovation_cov() sums in the order of k_maha's shuffle folds), the sequential-scan decision with its two gates and the
// new-landmark initialisation (k_assoc_decide), and the correction with the fresh pose (k_gain + k_rank2).
// Same arithmetic as the multi-kernel chain -> bit-identical results and decisions.
// ---------------------------------------------------------------------------------------------
// N: the filter's active dimension (the leading N x N block of Sigma is all these J measurements can touch);
// returns (on every lane) the number of corrections applied.
template <int THREADS>
__device__ int small_associate_body(const PoolView& pv, int b, int N, const double* __restrict__ meas, int J,
                                    int known_count_in, int* __restrict__ assoc_out, double* sm, int has_twist = 0,
                                    double dtheta = 0.0, double dx = 0.0) {
    const int tid = threadIdx.x;
    const int ld = pv.ld, n = pv.n;
    const int ldS = N | 1;
    double* S = sm;
    double* st = S + (size_t)N * ldS;
    double* Gg = st + N;
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ int sh_M, sh_lm, sh_new, sh_applied;
    __shared__ unsigned char sh_touch[64];
    if (tid < 64) sh_touch[tid] = 0;

    double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    double* stg = pv.state + (size_t)b * ld;
    for (int r = tid >> 6; r < N; r += THREADS / 64)
        for (int c = tid & 63; c < N; c += 64) S[r * ldS + c] = Sg[(size_t)r * ld + c];
    for (int r = tid; r < N; r += THREADS) st[r] = stg[r];
    if (tid == 0) { sh_M = known_count_in; sh_applied = 0; }
    __syncthreads();
    if (has_twist) small_predict<THREADS>(S, st, ldS, N, dtheta, dx, pv.p);  // a deferred prediction() (uniform)

    for (int j = 0; j < J; j++) {                       // :291 sequential, state-carrying
        const double mx = meas[2 * j], my = meas[2 * j + 1];
        const int M = sh_M;
        // :300-309, one landmark per LANE (M <= 50 < 64: wave 0 holds them all), everything from LDS.  The lane
        // keeps H, S^-1 and nu of its landmark: if it wins, they are exactly what the correction needs.
        MeasTerms m;
        double Si[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
        double v0 = 0.0, v1 = 0.0;
        double d = pv.p.gate_new;                        // :293
        int di = INT_MAX;
        if (tid < M) {
            const int i = tid;
            measurement_terms(st[2 * i + 3], st[2 * i + 4], mx, my, st[0], st[1], st[2], m);  // fresh pose, :219-221
            double S55[5][5], Sm[2][2];
#pragma unroll
            for (int k = 0; k < 5; k++)
#pragma unroll
                for (int l = 0; l < 5; l++) S55[k][l] = S[idx5(k, i) * ldS + idx5(l, i)];
            innovation_cov(S55, m.H, pv.p.r_meas, Sm);   // same summation order as k_maha's shuffle folds
            inv2(Sm, Si);
            v0 = m.z0 - m.zh0; v1 = m.z1 - m.zh1;        // bearing NOT wrapped, :269
            const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
            const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
            const double sc = t0 * v0 + t1 * v1;
            if (sc < d) { d = sc; di = i; }              // :305-309 (NaN never wins)
        }
        if (tid < kWave) {  // sequential-scan semantics = lexicographic (d, i) minimum, as in k_assoc_decide
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double od = __shfl_down(d, off, kWave);
                const int oi = __shfl_down(di, off, kWave);
                if (od < d || (od == d && oi < di)) { d = od; di = oi; }
            }
        }
        if (tid == 0) {                                  // :293-330
            double best = d;
            const int idx = (di == INT_MAX) ? M : di;
            int Mn = M, is_new = 0;
            if (idx == M && idx < n) {                   // :318-327 new landmark
                const double theta = st[0], x = st[1], y = st[2];
                const double ri = sqrt(mx * mx + my * my);
                const double phii = atan2(my, mx);
                st[2 * idx + 3] = x + ri * cos(phii + theta);
                st[2 * idx + 3 + 1] = y + ri * sin(phii + theta);
                Mn = M + 1;
                best = 0.0;
                is_new = 1;
            }
            const int active = (best < pv.p.gate_update) && idx < n;
            sh_M = Mn;
            sh_lm = active ? idx : -1;
            sh_new = is_new;
            assoc_out[j] = sh_lm;
            if (active) { sh_applied++; sh_touch[idx] = 1; }
        }
        __syncthreads();
        const int lm = sh_lm;
        const int is_new = sh_new;
        if (lm < 0) {                                    // dropped (uniform)
            // every wave has read sh_lm / sh_new before lane 0 may overwrite them in the next measurement's decision
            // (128-thread workgroups: wave 0 alone scores and decides, and could run ahead of a slower wave 1)
            __syncthreads();
            continue;
        }
        if (is_new) {                                    // :331-381 with the FRESH pose: a new landmark has no record
            if (tid == 0) {
                MeasTerms mn;
                measurement_terms(st[2 * lm + 3], st[2 * lm + 4], mx, my, st[0], st[1], st[2], mn);
                double S55[5][5], Sm[2][2], Sn[2][2];
                for (int k = 0; k < 5; k++)
                    for (int l = 0; l < 5; l++) S55[k][l] = S[idx5(k, lm) * ldS + idx5(l, lm)];
                innovation_cov(S55, mn.H, pv.p.r_meas, Sm);
                inv2(Sm, Sn);
                for (int a = 0; a < 2; a++)
                    for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = mn.H[a][k];
                sh_Si[0] = Sn[0][0]; sh_Si[1] = Sn[0][1]; sh_Si[2] = Sn[1][0]; sh_Si[3] = Sn[1][1];
                sh_nu[0] = mn.z0 - mn.zh0;
                sh_nu[1] = normalize_angle(mn.z1 - mn.zh1);
            }
        } else if (tid == lm) {                          // the winning lane hands over what it scored with
            for (int a = 0; a < 2; a++)
                for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = m.H[a][k];
            sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
            sh_nu[0] = v0;
            sh_nu[1] = normalize_angle(v1);              // :183 (the score used it unwrapped)
        }
        __syncthreads();
        double k0 = 0.0, k1 = 0.0;
        const int r = tid;
        if (r < N) {
            double sht0 = 0.0, sht1 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int c = idx5(k, lm);
                const double p = S[r * ldS + c];
                const double g = S[c * ldS + r];
                sht0 += p * sh_H[k];
                sht1 += p * sh_H[5 + k];
                g0 += sh_H[k] * g;
                g1 += sh_H[5 + k] * g;
            }
            k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];
            k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
            Gg[r] = g0;
            Gg[N + r] = g1;
        }
        __syncthreads();
        if (r < N) {
            // (row and the two G rows never overlap: lets the LDS reads of later columns start before earlier writes)
            double* __restrict__ row = S + r * ldS;
            const double* __restrict__ G0 = Gg;
            const double* __restrict__ G1 = Gg + N;
#pragma unroll 4
            for (int c = 0; c < N; c++) row[c] = row[c] - (k0 * G0[c] + k1 * G1[c]);
            double s = st[r] + (k0 * sh_nu[0] + k1 * sh_nu[1]);
            if (r == 0) s = normalize_angle(s);
            st[r] = s;
        }
        __syncthreads();
    }

    for (int r = tid >> 6; r < N; r += THREADS / 64)
        for (int c = tid & 63; c < N; c += 64) Sg[(size_t)r * ld + c] = S[r * ldS + c];
    for (int r = tid; r < N; r += THREADS) stg[r] = st[r];
    if (tid == 0) {
        AssocRec a;
        a.known_count = sh_M; a.lm = sh_lm; a.active = sh_lm >= 0; a.pad = 0; a.best = 0.0;
        pv.assoc[b] = a;
    }
    flush_touches(pv, b, sh_touch, n < 64 ? n : 64, THREADS);
    __syncthreads();
    return sh_applied;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_small_associate(PoolView pv, const double* __restrict__ meas, int J,
                                                             int known_count_in, int* __restrict__ assoc_out,
                                                             int has_twist, double dtheta, double dx) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    small_associate_body<THREADS>(pv, blockIdx.x, pv.N, meas, J, known_count_in, assoc_out, sm, has_twist, dtheta, dx);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_small_associate_inline(PoolView pv, SmallInlineMeas in, int J,
                                                                    int known_count_in, int* __restrict__ assoc_out,
                                                                    int has_twist, double dtheta, double dx) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    small_associate_body<THREADS>(pv, 0, pv.N, in.xy, J, known_count_in, assoc_out, sm, has_twist, dtheta, dx);
}

// The same for a whole pool and one step of an unknown-association log: filter b takes its count[b] readings,
// continues from its own device-resident known_count, and works inside its own discovered prefix
// N_b = max(min_active, 3 + 2*min(n, known_count + count)) -- everything beyond still holds constructor values,
// where K and H*Sigma are exact zeros (see ekf_associate).  The host guarantees N_b <= small_max_dim() for
// every filter of the launch (its bound is refreshed from the device), so one launch per STEP replaces
// 4 launches per measurement slot.  assoc_out[b][jmax]: decisions; slots >= count keep -2.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_pool_associate(PoolView pv, const double* __restrict__ meas,
                                                            const int* __restrict__ count, int jmax, int min_active,
                                                            int* __restrict__ assoc_out,
                                                            unsigned long long* __restrict__ corr_counter) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x;
    const int J = count ? count[b] : jmax;  // count == nullptr: every filter has exactly jmax readings
    if (J <= 0) return;  // uniform
    const int kc = pv.assoc[b].known_count;
    int m = kc + J < pv.n ? kc + J : pv.n;
    int N = 3 + 2 * m;
    if (min_active > N) N = min_active;
    if (N > pv.N) N = pv.N;  // pv.N = the launch's bound (sizes the LDS)
    const int applied = small_associate_body<THREADS>(pv, b, N, meas + (size_t)b * jmax * 2, J, kc, assoc_out + (size_t)b * jmax, sm);
    if (threadIdx.x == 0 && corr_counter && applied) atomicAdd(corr_counter, (unsigned long long)applied);
}

// ---------------------------------------------------------------------------------------------
// A whole RANGE OF STEPS of a known-association log for a pool of small maps in ONE launch: workgroup b keeps
// filter b's Sigma and state in LDS from the first to the last step -- prediction() (ekf_slam.cpp:55-106, the
// structured arithmetic of k_predict), the top of measurement() (:109-128) and every logged correction
// (:132-194, the arithmetic of k_small_measure_inline) -- so HBM sees the log once and Sigma twice per run instead of
// (2 + 2V) launches per step.  This is the reference's own operating point (n = 20) at Monte-Carlo scale.
// Bit-identical to the multi-kernel replay.  The next step's log slots are fetched while the current step runs.
// ---------------------------------------------------------------------------------------------
// THREADS: one lane per row of Sigma, so 64 (a single wavefront, N <= 64: the n = 20 case) or 128.  With 256-thread
// workgroups only wave 0 of each would carry rows, and the wave 0s of all resident workgroups share a SIMD.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_pool_run_known(PoolView pv, const double* __restrict__ twist,
                                                                  const int* __restrict__ lm_idx,
                                                                  const double* __restrict__ z_xy,
                                                                  const double* __restrict__ init_xy, int vmax,
                                                                  int t0, int t1, int do_init) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x, tid = threadIdx.x, B = pv.B;
    const int N = pv.N, ld = pv.ld, n = pv.n;
    const int ldS = N | 1;
    double* S = sm;                    // [N][ldS]
    double* st = S + (size_t)N * ldS;  // [N]
    double* Gg = st + N;               // [2][N]
    __shared__ double sh_H[10], sh_Si[4], sh_nu[2];
    __shared__ double sh_tw[2], sh_z[2 * 64];
    __shared__ int sh_lmv[64];
    __shared__ unsigned char sh_touch[64];
    if (tid < 64) sh_touch[tid] = 0;

    double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    double* stg = pv.state + (size_t)b * ld;
    for (int r = tid >> 6; r < N; r += THREADS / 64)
        for (int c = tid & 63; c < N; c += 64) S[r * ldS + c] = Sg[(size_t)r * ld + c];
    for (int r = tid; r < N; r += THREADS) st[r] = stg[r];

    // log slots of step t for this filter, one slot per lane (vmax <= 64)
    int lm_n = -1;
    double zx_n = 0.0, zy_n = 0.0, tw_n = 0.0;
    auto fetch = [&](int t) {
        if (t < t1) {
            const size_t slot = ((size_t)t * B + b) * vmax + tid;
            if (tid < vmax) { lm_n = lm_idx[slot]; zx_n = z_xy[slot * 2]; zy_n = z_xy[slot * 2 + 1]; }
            if (tid < 2) tw_n = twist[((size_t)t * B + b) * 2 + tid];
        }
    };
    fetch(t0);

    for (int t = t0; t < t1; t++) {
        __syncthreads();  // previous step done with sh_lmv / sh_z / sh_tw; first trip: LDS image complete
        // (r, phi) of every reading of the step at once, one reading per lane (ekf_slam.cpp:142-146)
        if (tid < vmax) { sh_lmv[tid] = lm_n; sh_z[2 * tid] = sqrt(zx_n * zx_n + zy_n * zy_n); sh_z[2 * tid + 1] = atan2(zy_n, zx_n); }
        if (tid < 2) sh_tw[tid] = tw_n;
        __syncthreads();
        fetch(t + 1);  // flies under this step's arithmetic

        small_predict<THREADS>(S, st, ldS, N, sh_tw[0], sh_tw[1], pv.p);  // prediction(), ekf_slam.cpp:55-106

        // ---- measurement(), ekf_slam.cpp:108-197 ----
        const double theta = st[0], x = st[1], y = st[2];  // captured ONCE per call, :109-111
        if (do_init) {                                     // first call only, :113-128
            __syncthreads();
            const double* sens = init_xy + (size_t)b * 2 * n;
            for (int i = tid; i < n; i += THREADS) {
                const double sx = sens[2 * i], sy = sens[2 * i + 1];
                const double ri = sqrt(sx * sx + sy * sy);
                const double phii = atan2(sy, sx);
                st[2 * i + 3] = x + ri * cos(phii + theta);
                st[2 * i + 3 + 1] = y + ri * sin(phii + theta);
            }
            do_init = 0;
            __syncthreads();
        }
        for (int v = 0; v < vmax; v++) {  // :132-194, ascending landmark order, -1 padded
            const int lm = sh_lmv[v];
            if (lm < 0) break;            // uniform
            if (tid == 0) {
                MeasTerms m;
                m.z0 = sh_z[2 * v]; m.z1 = sh_z[2 * v + 1];
                predicted_terms(st[2 * lm + 3], st[2 * lm + 4], theta, x, y, m);
                double S55[5][5], Sm[2][2], Si[2][2];
                for (int k = 0; k < 5; k++)
                    for (int l = 0; l < 5; l++) S55[k][l] = S[idx5(k, lm) * ldS + idx5(l, lm)];
                innovation_cov(S55, m.H, pv.p.r_meas, Sm);
                inv2(Sm, Si);
                for (int a = 0; a < 2; a++)
                    for (int k = 0; k < 5; k++) sh_H[a * 5 + k] = m.H[a][k];
                sh_Si[0] = Si[0][0]; sh_Si[1] = Si[0][1]; sh_Si[2] = Si[1][0]; sh_Si[3] = Si[1][1];
                sh_nu[0] = m.z0 - m.zh0;                   // :182
                sh_nu[1] = normalize_angle(m.z1 - m.zh1);  // :183
                sh_touch[lm] = 1;
            }
            __syncthreads();
            double k0 = 0.0, k1 = 0.0;
            const int r = tid;
            if (r < N) {
                double sht0 = 0.0, sht1 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const int c = idx5(k, lm);
                    const double p = S[r * ldS + c];
                    const double g = S[c * ldS + r];
                    sht0 += p * sh_H[k];
                    sht1 += p * sh_H[5 + k];
                    g0 += sh_H[k] * g;
                    g1 += sh_H[5 + k] * g;
                }
                k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];  // :178
                k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
                Gg[r] = g0;
                Gg[N + r] = g1;
            }
            __syncthreads();
            if (r < N) {
                double* __restrict__ row = S + r * ldS;
                const double* __restrict__ G0 = Gg;
                const double* __restrict__ G1 = Gg + N;
#pragma unroll 4
                for (int c = 0; c < N; c++) row[c] = row[c] - (k0 * G0[c] + k1 * G1[c]);  // :191-192
                double s = st[r] + (k0 * sh_nu[0] + k1 * sh_nu[1]);                             // :186
                if (r == 0) s = normalize_angle(s);                                             // :187
                st[r] = s;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int r = tid >> 6; r < N; r += THREADS / 64)
        for (int c = tid & 63; c < N; c += 64) Sg[(size_t)r * ld + c] = S[r * ldS + c];
    for (int r = tid; r < N; r += THREADS) stg[r] = st[r];
    flush_touches(pv, b, sh_touch, n, THREADS);
}

size_t small_lds_bytes(int N) { return sizeof(double) * ((size_t)N * (N | 1) + 3 * (size_t)N); }
int small_max_dim() { return 104; }  // N <= 104: 87 KB of LDS, one lane per row (128-thread workgroups)

hipError_t small_prepare() {
    hipError_t e = hipSuccess;
    for (const void* f : {reinterpret_cast<const void*>(&k_small_measure_inline<64>),
                          reinterpret_cast<const void*>(&k_small_measure_inline<128>),
                          reinterpret_cast<const void*>(&k_small_associate_inline<64>),
                          reinterpret_cast<const void*>(&k_small_associate_inline<128>),
                          reinterpret_cast<const void*>(&k_small_associate<64>),
                          reinterpret_cast<const void*>(&k_small_associate<128>),
                          reinterpret_cast<const void*>(&k_pool_associate<64>),
                          reinterpret_cast<const void*>(&k_pool_associate<128>)}) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes(small_max_dim()));
        if (e != hipSuccess) return e;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pool_run_known<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes(64));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pool_run_known<128>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds_bytes(small_max_dim()));
    if (e != hipSuccess) return e;
    return hipSuccess;
}

void launch_pool_run_known(const PoolView& pv, const double* twist, const int* lm_idx, const double* z_xy,
                           const double* init_xy, int vmax, int t0, int t1, int do_init, hipStream_t s) {
    if (pv.N <= 64)
        hipLaunchKernelGGL(k_pool_run_known<64>, dim3(pv.B), dim3(64), small_lds_bytes(pv.N), s, pv, twist, lm_idx, z_xy,
                           init_xy, vmax, t0, t1, do_init);
    else
        hipLaunchKernelGGL(k_pool_run_known<128>, dim3(pv.B), dim3(128), small_lds_bytes(pv.N), s, pv, twist, lm_idx, z_xy,
                           init_xy, vmax, t0, t1, do_init);
}

void launch_pool_associate(const PoolView& pv, const double* meas, const int* count, int jmax, int min_active,
                           int* assoc_out, unsigned long long* corr_counter, hipStream_t s) {
    // one lane per row of the prefix: a single wavefront while N <= 64 (wave 0 of a 256-thread workgroup would carry
    // all rows, and the wave 0s of the resident workgroups share one SIMD)
    if (pv.N <= 64)
        hipLaunchKernelGGL(k_pool_associate<64>, dim3(pv.B), dim3(64), small_lds_bytes(pv.N), s, pv, meas, count, jmax,
                           min_active, assoc_out, corr_counter);
    else
        hipLaunchKernelGGL(k_pool_associate<128>, dim3(pv.B), dim3(128), small_lds_bytes(pv.N), s, pv, meas, count, jmax,
                           min_active, assoc_out, corr_counter);
}

void launch_small_associate(const PoolView& pv, const double* meas, int J, int known_count, int* assoc_out,
                            int has_twist, double dtheta, double dx, hipStream_t s) {
    if (pv.N <= 64)
        hipLaunchKernelGGL(k_small_associate<64>, dim3(pv.B), dim3(64), small_lds_bytes(pv.N), s, pv, meas, J, known_count,
                           assoc_out, has_twist, dtheta, dx);
    else
        hipLaunchKernelGGL(k_small_associate<128>, dim3(pv.B), dim3(128), small_lds_bytes(pv.N), s, pv, meas, J,
                           known_count, assoc_out, has_twist, dtheta, dx);
}

void launch_small_associate_inline(const PoolView& pv, const SmallInlineMeas& in, int J, int known_count, int* assoc_out,
                                   int has_twist, double dtheta, double dx, hipStream_t s) {
    if (pv.N <= 64)
        hipLaunchKernelGGL(k_small_associate_inline<64>, dim3(1), dim3(64), small_lds_bytes(pv.N), s, pv, in, J, known_count,
                           assoc_out, has_twist, dtheta, dx);
    else
        hipLaunchKernelGGL(k_small_associate_inline<128>, dim3(1), dim3(128), small_lds_bytes(pv.N), s, pv, in, J,
                           known_count, assoc_out, has_twist, dtheta, dx);
}

void launch_small_measure_inline(const PoolView& pv, const SmallInline& in, int do_init, int has_twist, double dtheta,
                                 double dx, hipStream_t s) {
    if (pv.N <= 64)
        hipLaunchKernelGGL(k_small_measure_inline<64>, dim3(1), dim3(64), small_lds_bytes(pv.N), s, pv, in, do_init,
                           has_twist, dtheta, dx);
    else
        hipLaunchKernelGGL(k_small_measure_inline<128>, dim3(1), dim3(128), small_lds_bytes(pv.N), s, pv, in, do_init,
                           has_twist, dtheta, dx);
}

}  // namespace ekf
