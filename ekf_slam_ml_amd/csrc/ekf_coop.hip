// ekf_coop.hip -- a whole prediction() + measurement() tick of ONE mid-size filter (104 < N <= ~1500, e.g. the
// n = 200 of BASELINE.json configs[1]) in ONE launch, the covariance resident in LDS across all its corrections.
//
// The per-correction launch of ekf_fused.hip costs ~8 us at n = 200: a kernel boundary plus a chain of dependent
// memory round trips, for 2.6 MB of traffic that lives in L2 anyway.  Here G workgroups (one per CU) split the
// rows of Sigma between them and keep them in LDS from the first to the last visible landmark of the call
// (ekf_slam.cpp:132-194):
//   * workgroup w owns R consecutive rows starting at 3 + w*R (R even: the two rows of a landmark never split) and
//     the state entries of those rows;
//   * EVERY workgroup also carries its own replica of the three pose rows 0..2 and of the pose: their update needs
//     only K(0..2, :) = Sigma(0..2, c5) H^T S^-1, which the replica itself provides, so all replicas stay bit-identical
//     without ever being exchanged; prediction() (rows/columns 1, 2 only, ekf_slam.cpp:101-102) is local for the same
//     reason;
//   * per correction exactly ONE workgroup-to-all hand-off: the owner of landmark i has everything H, S^-1, nu and
//     G = H Sigma need (the pose-row replica and its own rows 3+2i, 4+2i), computes them, and publishes G (2 x N) and the
//     16 scalar terms; every other workgroup waits for that slot, builds K for its own rows from its own LDS image and
//     applies Sigma(r, :) -= K(r, :) G -- no grid barrier, no second exchange.
// The hand-off is the write-through form of the CDNA4 guide (cdna_hip_programming.md, Guideline 16 R1): payload by
// agent-scope (sc1) stores, every storing wave drains vmcnt, workgroup barrier, ONE lane stores the flag; consumers
// poll that one word relaxed and read the payload with agent-scope (sc1) loads only, which bypass the CU's L1.  Results do
// not depend on placement or timing.  Every spin is bounded: on a time-out the workgroup raises the error word (host-
// mapped) and leaves without writing, so the grid always drains.  All G <= #CUs workgroups must be resident at once
// (each asks for > 80 KB of LDS: one per CU), which holds on a GPU that is not shared with a kernel that never ends.
//
// The result is written OUT OF PLACE (sigma_next / state_next, swapped by the host): a workgroup that starts late
// still reads the call's input, never another workgroup's output.  Same operations in the same order as k_gain +
// k_rank2 / k_correct_fused -> bit-identical (tests/test_gpu_coop.py).
#include "ekf_kernels.hpp"

namespace ekf {

__device__ __forceinline__ void store_wt(double* p, double v) {  // write-through (sc1) 8-byte store
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_l2(const double* p) {      // agent-scope (sc1) load: never served by this CU's L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned kCoopSpinLimit = 1u << 18;  // ~0.2-0.5 s of polling, then give up

// optional phase trace (ekf_cooperative_trace): lane 0 of every workgroup stamps the 100 MHz wall clock
#define COOP_CY(k)                                                                      \
    do {                                                                                \
        if (a.trace && tid == 0 && v == 1) a.trace[w * kCoopTraceSlots + (k)] = clock64(); \
    } while (0)
#define COOP_TR(k)                                                                      \
    do {                                                                                \
        if (a.trace && tid == 0 && (k) < kCoopTraceSlots) a.trace[w * kCoopTraceSlots + (k)] = wall_clock64(); \
    } while (0)

// LDS image: rows of NS = N + 1 doubles (N is odd, so NS is even and every row starts 16-B aligned); column N is the
// zero pad column of Sigma and stays zero (its G entry is an exact zero).  All row traffic is double2.
// R (rows per workgroup) is a template parameter: every row loop is branch-free and fully unrolled.  The last workgroup
// may own fewer rows; its missing rows are zero-filled phantoms (their K is 0, they stay 0 and are never written back).
template <int THREADS, int R>
__global__ __launch_bounds__(THREADS) void k_coop_measure(PoolView pv, CoopArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, w = blockIdx.x, lane = tid & 63;
    const int N = pv.N, ld = pv.ld;
    const int NS = N + 1, NS2 = NS >> 1, ld2 = ld >> 1;
    const int r0 = 3 + w * R;                       // first owned row
    const int nr = min(R, N - r0);                  // owned rows (the last workgroup may own fewer); > 0 by the grid size
    double* __restrict__ P = sm;                            // [3][NS]  replica of the pose rows
    double* __restrict__ T = P + 3 * NS;            // [R][NS]  owned rows
    double* __restrict__ Gb = T + R * NS;           // [2][NS]  G = H Sigma of the current correction
    double* __restrict__ stl = Gb + 2 * NS;         // [R]      owned state entries
    double* __restrict__ Kt = stl + R;                      // [R + 3][2]  K rows: 0..2 pose, 3.. owned
    // (all scratch lives in the dynamic region: static __shared__ variables would shift its base off 16-B alignment)
    double* __restrict__ sh_pose = Kt + 2 * (R + 3) + ((R + 3) & 1) * 0;   // [4]
    double* __restrict__ sh_H = sh_pose + 4;                // [10]
    double* __restrict__ sh_Si = sh_H + 10;                 // [4]
    double* __restrict__ sh_nu = sh_Si + 4;                 // [2]
    double* __restrict__ sh_tr = sh_nu + 2;                 // [4] sin/cos of the prediction
    double* __restrict__ zs = sh_tr + 4;                    // [R] sensor readings (x, y) of the owned landmarks
    int* sh_okp = reinterpret_cast<int*>(zs + R);
    double2_t* __restrict__ P2 = reinterpret_cast<double2_t*>(P);
    double2_t* __restrict__ T2 = reinterpret_cast<double2_t*>(T);
    double2_t* __restrict__ G2 = reinterpret_cast<double2_t*>(Gb);

    const double2_t* __restrict__ cur2 = reinterpret_cast<const double2_t*>(pv.sigma);
    const double* __restrict__ st = pv.state;
    double2_t* __restrict__ nxt2 = reinterpret_cast<double2_t*>(a.sigma_next);
    double* __restrict__ stn = a.state_next;

    COOP_TR(0);
    if (a.trace && tid == 0) a.trace[w * kCoopTraceSlots + 60] = clock64();   // shader clock, for the frequency
    // ---- load: pose rows (everybody), owned rows, owned state, pose.  All loads of a pass are issued before the first
    // ---- LDS store; the prediction's sines and cosines (which only need theta) are evaluated while they fly. ----
    const double th_in = st[0];
    const double2_t zero2{0.0, 0.0};
    for (int c2 = tid; c2 < NS2; c2 += THREADS) {
        double2_t pr[3], tr_[R];
#pragma unroll
        for (int r = 0; r < 3; r++) pr[r] = cur2[r * ld2 + c2];
#pragma unroll
        for (int u = 0; u < R; u++) tr_[u] = cur2[min(r0 + u, N - 1) * ld2 + c2];
#pragma unroll
        for (int u = 0; u < R; u++) T2[u * NS2 + c2] = u < nr ? tr_[u] : zero2;
#pragma unroll
        for (int r = 0; r < 3; r++) P2[r * NS2 + c2] = pr[r];
    }
    if (tid < R) {
        stl[tid] = tid < nr ? st[r0 + tid] : 0.0;
        if (a.inl_count < 0) zs[tid] = a.sensor[min(r0 - 3 + tid, N - 4)];
    }
    if (tid >= 64 && tid < 67) sh_pose[tid - 64] = st[tid - 64];
    if (a.has_twist && tid < 64) {
        // lanes 0/1: sin and cos of theta / theta + dtheta -- two calls instead of four on the critical path
        const double arg = (lane & 1) ? th_in + a.dtheta : th_in;
        const double sv = sin(arg), cv = cos(arg);
        if (lane < 2) { sh_tr[lane] = sv; sh_tr[2 + lane] = cv; }
    }
    __syncthreads();
    COOP_TR(1);

    // ---- prediction(), ekf_slam.cpp:55-106: the structured arithmetic of k_predict on the split image ----
    if (a.has_twist) {
        const double theta = sh_pose[0], dtheta = a.dtheta, dx = a.dx;
        const double sin_t = sh_tr[0], sin_td = sh_tr[1], cos_t = sh_tr[2], cos_td = sh_tr[3];
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
        double c33[3][3], px = 0.0, py = 0.0;
        if (tid == 0) {
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) c33[r][k] = P[r * NS + k];
            px = sh_pose[1]; py = sh_pose[2];
        }
        __syncthreads();  // everybody holds the old theta and thread 0 the old 3x3 block
        for (int k = 3 + tid; k < N; k += THREADS) {       // rows 1, 2 beyond the pose block
            const double q0 = P[k], q1 = P[NS + k], q2 = P[2 * NS + k];
            P[NS + k] = a10 * q0 + q1;
            P[2 * NS + k] = a20 * q0 + q2;
        }
        if (tid >= 64 && tid < 64 + R) {                    // columns 1, 2 of the owned rows
            double* rowk = T + (tid - 64) * NS;
            const double q0 = rowk[0], q1 = rowk[1], q2 = rowk[2];
            rowk[1] = q0 * a10 + q1;
            rowk[2] = q0 * a20 + q2;
        }
        if (tid == 0) {
            sh_pose[0] = theta + u0;  // :99 -- theta is NOT wrapped after the prediction
            sh_pose[1] = px + u1;
            sh_pose[2] = py + u2;
            double Tm[3][3];
            for (int k = 0; k < 3; k++) {
                Tm[0][k] = c33[0][k];
                Tm[1][k] = a10 * c33[0][k] + c33[1][k];
                Tm[2][k] = a20 * c33[0][k] + c33[2][k];
            }
            for (int r = 0; r < 3; r++) {
                P[r * NS + 0] = Tm[r][0];
                P[r * NS + 1] = Tm[r][0] * a10 + Tm[r][1];
                P[r * NS + 2] = Tm[r][0] * a20 + Tm[r][2];
            }
            P[0] += pv.p.q_pose;  // Q = diag(q,q,q,0...) :40-43
            P[NS + 1] += pv.p.q_pose;
            P[2 * NS + 2] += pv.p.q_pose;
        }
        __syncthreads();
    }

    COOP_TR(2);
    // ---- top of measurement(), :109-128: the pose is captured ONCE; first call: every landmark from the sensor vector ----
    const double theta = sh_pose[0], x = sh_pose[1], y = sh_pose[2];
    const int lm0 = (r0 - 3) >> 1;          // first owned landmark
    if (a.do_init) {
        if (tid < (nr >> 1)) {
            const double sx = zs[2 * tid], sy = zs[2 * tid + 1];
            const double ri = sqrt(sx * sx + sy * sy);
            const double phii = atan2(sy, sx);
            stl[2 * tid] = x + ri * cos(phii + theta);
            stl[2 * tid + 1] = y + ri * sin(phii + theta);
        }
        __syncthreads();
    }

    COOP_TR(3);
    // ---- the visible landmarks in ascending order, :132-194 ----
    const bool inl = a.inl_count >= 0;
    const int V = inl ? a.inl_count : a.vlist[0];
    int lm_next = V > 0 ? (inl ? a.inl_lm[0] : a.vlist[1]) : 0;
    for (int v = 0; v < V; v++) {
        const int lm = lm_next;
        if (v + 1 < V) lm_next = inl ? a.inl_lm[v + 1] : a.vlist[2 + v];   // (scalar load: arrives while this correction runs)
        COOP_TR(4 + 6 * v);
        const int owner = (2 * lm) / R;
        double* slot = a.xchg + v * a.xstride;
        if (w == owner) {
            const int la = 2 * (lm - lm0);   // local row of 3 + 2 lm
            COOP_CY(50);
            if (tid < 64) {                  // wave 0: H, S^-1, nu with the STALE pose (:137-183), lane-parallel
                const double sx = inl ? a.inl_xy[v][0] : zs[la], sy = inl ? a.inl_xy[v][1] : zs[la + 1];
                auto s55 = [&](int k, int l) {
                    const double* rowk = k < 3 ? P + k * NS : T + (la + k - 3) * NS;
                    return rowk[idx5(l, lm)];
                };
                wave_terms(lane, stl[la], stl[la + 1], sx, sy, theta, x, y, pv.p.r_meas, s55, true, sh_H, sh_Si, sh_nu);
            }
            COOP_CY(51);
            __syncthreads();
            COOP_CY(52);
            COOP_TR(4 + 6 * v + 1);
            const double2_t* ra = T2 + la * NS2;
            const double2_t* rb = ra + NS2;
            for (int c2 = tid; c2 < NS2; c2 += THREADS) {   // G = H Sigma: rows 0,1,2 from the replica, rows of lm from the tile
                const double2_t gk[5] = {P2[c2], P2[NS2 + c2], P2[2 * NS2 + c2], ra[c2], rb[c2]};
                double2_t g0{0.0, 0.0}, g1{0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    g0.x += sh_H[k] * gk[k].x; g0.y += sh_H[k] * gk[k].y;
                    g1.x += sh_H[5 + k] * gk[k].x; g1.y += sh_H[5 + k] * gk[k].y;
                }
                G2[c2] = g0; G2[NS2 + c2] = g1;
                store_wt(slot + 2 * c2, g0.x); store_wt(slot + 2 * c2 + 1, g0.y);
                store_wt(slot + ld + 2 * c2, g1.x); store_wt(slot + ld + 2 * c2 + 1, g1.y);
            }
            if (tid < 10) store_wt(slot + 2 * ld + tid, sh_H[tid]);
            else if (tid < 14) store_wt(slot + 2 * ld + tid, sh_Si[tid - 10]);
            else if (tid < 16) store_wt(slot + 2 * ld + tid, sh_nu[tid - 14]);
            COOP_CY(53);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
            COOP_CY(54);
            __syncthreads();
            COOP_CY(55);
            if (tid == 0) __hip_atomic_store(a.flags + v, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            COOP_TR(4 + 6 * v + 2);
        } else {
            if (tid == 64) {                                   // ONE lane (of wave 1) polls ONE word, relaxed
                int ok = 0;
                for (unsigned spin = 0; spin < kCoopSpinLimit; spin++) {
                    if (__hip_atomic_load(a.flags + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok) __hip_atomic_store(a.err, 1u + (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                *sh_okp = ok;
            }
            __syncthreads();
            if (!*sh_okp) return;   // uniform: gave up -- nothing of this workgroup is written, the host reports the error
            COOP_TR(4 + 6 * v + 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the loads below the poll
            // every load of the block is agent-scope; all of a pass are in flight before the first LDS store
            double tv = 0.0;
            if (tid >= THREADS - 16) tv = load_l2(slot + 2 * ld + (tid - (THREADS - 16)));
            for (int c2 = tid; c2 < NS2; c2 += THREADS) {
                const double a0 = load_l2(slot + 2 * c2), a1 = load_l2(slot + 2 * c2 + 1);
                const double b0 = load_l2(slot + ld + 2 * c2), b1 = load_l2(slot + ld + 2 * c2 + 1);
                G2[c2] = double2_t{a0, a1};
                G2[NS2 + c2] = double2_t{b0, b1};
            }
            if (tid >= THREADS - 16) {
                const int k = tid - (THREADS - 16);
                if (k < 10) sh_H[k] = tv;
                else if (k < 14) sh_Si[k - 10] = tv;
                else sh_nu[k - 14] = tv;
            }
        }
        __syncthreads();
        if (w != owner) COOP_TR(4 + 6 * v + 2);
        // K = Sigma H^T S^-1 (:178) for the pose rows (replica) and the owned rows
        if (tid < R + 3) {
            const double* src = tid < 3 ? P + tid * NS : T + (tid - 3) * NS;
            double p[5];
#pragma unroll
            for (int k = 0; k < 5; k++) p[k] = src[idx5(k, lm)];
            double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                sht0 += p[k] * sh_H[k];
                sht1 += p[k] * sh_H[5 + k];
            }
            Kt[2 * tid] = sht0 * sh_Si[0] + sht1 * sh_Si[2];
            Kt[2 * tid + 1] = sht0 * sh_Si[1] + sht1 * sh_Si[3];
        }
        COOP_CY(56);
        __syncthreads();
        COOP_TR(4 + 6 * v + 3);
        COOP_CY(57);
        // Sigma <- (I - K H) Sigma (:191-192) on the replica rows and the owned rows; a thread keeps its columns' G and
        // takes the rows in groups: all LDS reads of a group are issued before its first write
        for (int c2 = tid; c2 < NS2; c2 += THREADS) {
            const double2_t g0 = G2[c2], g1 = G2[NS2 + c2];
            {
                double2_t pv3[3], kk[3];
#pragma unroll
                for (int r = 0; r < 3; r++) { pv3[r] = P2[r * NS2 + c2]; kk[r] = reinterpret_cast<const double2_t*>(Kt)[r]; }
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    pv3[r].x = pv3[r].x - (kk[r].x * g0.x + kk[r].y * g1.x);
                    pv3[r].y = pv3[r].y - (kk[r].x * g0.y + kk[r].y * g1.y);
                }
#pragma unroll
                for (int r = 0; r < 3; r++) P2[r * NS2 + c2] = pv3[r];
            }
            double2_t tv2[R], kk[R];
#pragma unroll
            for (int u = 0; u < R; u++) { tv2[u] = T2[u * NS2 + c2]; kk[u] = reinterpret_cast<const double2_t*>(Kt)[u + 3]; }
#pragma unroll
            for (int u = 0; u < R; u++) {
                tv2[u].x = tv2[u].x - (kk[u].x * g0.x + kk[u].y * g1.x);
                tv2[u].y = tv2[u].y - (kk[u].x * g0.y + kk[u].y * g1.y);
            }
#pragma unroll
            for (int u = 0; u < R; u++) T2[u * NS2 + c2] = tv2[u];
        }
        COOP_CY(58);
        // state = state + Ki*z_diff (:186), theta wrapped (:187): pose replica + owned entries (waves 2, 3: off wave 0)
        if (tid >= 128 && tid < 128 + R + 3) {
            const int q = tid - 128;
            if (q < 3) {
                double s = sh_pose[q] + (Kt[2 * q] * sh_nu[0] + Kt[2 * q + 1] * sh_nu[1]);
                if (q == 0) s = normalize_angle(s);
                sh_pose[q] = s;
            } else {
                stl[q - 3] = stl[q - 3] + (Kt[2 * q] * sh_nu[0] + Kt[2 * q + 1] * sh_nu[1]);
            }
        }
        __syncthreads();
        COOP_CY(59);
        COOP_TR(4 + 6 * v + 4);
    }
    COOP_TR(kCoopTraceSlots - 2);

    // ---- write back, out of place (double2 rows; the pad columns beyond N + 1 are written as zeros) ----
    for (int c2 = tid; c2 < ld2; c2 += THREADS) {
        for (int rr = 0; rr < nr; rr++) nxt2[(r0 + rr) * ld2 + c2] = c2 < NS2 ? T2[rr * NS2 + c2] : zero2;
        if (w == 0)
#pragma unroll
            for (int r = 0; r < 3; r++) nxt2[r * ld2 + c2] = c2 < NS2 ? P2[r * NS2 + c2] : zero2;
    }
    if (tid < nr) stn[r0 + tid] = stl[tid];
    if (w == 0) {
        if (tid < 3) stn[tid] = sh_pose[tid];
        for (int c = N + tid; c < ld; c += THREADS) stn[c] = 0.0;
        if (tid == 0) {
            double* sn = pv.snap;
            sn[0] = theta; sn[1] = x; sn[2] = y;
            CorrRec rc;
            rc.nu0 = sh_nu[0]; rc.nu1 = sh_nu[1]; rc.active = V > 0; rc.n_active = 0; rc.pad = 0;
            rc.lm = V > 0 ? (inl ? a.inl_lm[V - 1] : a.vlist[V]) : -1;
            pv.rec[0] = rc;
            for (int v = 0; v < V; v++) touch_landmark(pv, 0, inl ? a.inl_lm[v] : a.vlist[1 + v]);
        }
    }
    COOP_TR(kCoopTraceSlots - 1);
    if (a.trace && tid == 0) a.trace[w * kCoopTraceSlots + 61] = clock64();
}

size_t coop_lds_bytes(int N, int R) {
    const size_t need = sizeof(double) * ((3 + R + 2) * (N + 1) + 2 * R + 2 * (R + 3) + 32);
    const size_t one_per_cu = 82 * 1024;   // > half of the 160 KB: at most one workgroup per CU
    return need > one_per_cu ? need : one_per_cu;
}

// rows per workgroup for a map of dimension N on a device with `cus` compute units; 0 = does not fit.  One of the
// instantiated values: the smallest that needs at most target_wgs (default 64) workgroups, else the one with the most
// workgroups that still fit one per CU.
static const int kCoopRows[] = {2, 4, 6, 8, 12, 16};
int coop_rows_per_wg(int N, int cus, int target_wgs) {
    if (N <= 3 || cus <= 0) return 0;
    int gmax = target_wgs > 0 ? target_wgs : 64;
    if (gmax > cus) gmax = cus;
    const size_t cap = (size_t)160 * 1024 - 2048;
    int fallback = 0;
    for (int R : kCoopRows) {
        const int G = (N - 3 + R - 1) / R;
        if (coop_lds_bytes(N, R) > cap) break;
        if (G <= gmax) return R;
        if (G <= cus && !fallback) fallback = R;
    }
    return fallback;
}

hipError_t coop_prepare() {
    const void* fns[] = {reinterpret_cast<const void*>(&k_coop_measure<256, 2>), reinterpret_cast<const void*>(&k_coop_measure<256, 4>),
                         reinterpret_cast<const void*>(&k_coop_measure<256, 6>), reinterpret_cast<const void*>(&k_coop_measure<256, 8>),
                         reinterpret_cast<const void*>(&k_coop_measure<256, 12>), reinterpret_cast<const void*>(&k_coop_measure<256, 16>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void launch_coop_measure(const PoolView& pv, const CoopArgs& a, hipStream_t s) {
    const int R = a.rows_per_wg;
    const int G = (pv.N - 3 + R - 1) / R;
    const size_t lds = coop_lds_bytes(pv.N, R);
    switch (R) {
        case 2: hipLaunchKernelGGL((k_coop_measure<256, 2>), dim3(G), dim3(256), lds, s, pv, a); break;
        case 4: hipLaunchKernelGGL((k_coop_measure<256, 4>), dim3(G), dim3(256), lds, s, pv, a); break;
        case 6: hipLaunchKernelGGL((k_coop_measure<256, 6>), dim3(G), dim3(256), lds, s, pv, a); break;
        case 8: hipLaunchKernelGGL((k_coop_measure<256, 8>), dim3(G), dim3(256), lds, s, pv, a); break;
        case 12: hipLaunchKernelGGL((k_coop_measure<256, 12>), dim3(G), dim3(256), lds, s, pv, a); break;
        default: hipLaunchKernelGGL((k_coop_measure<256, 16>), dim3(G), dim3(256), lds, s, pv, a); break;
    }
}

}  // namespace ekf
