// ekf_kernels.hip -- hand-written gfx950 kernels for the rigid2d::EKF_SLAM hot path.
// See ekf_kernels.hpp for the HBM layout.  Reference line numbers are rigid2d/src/ekf_slam.cpp.
#include "ekf_kernels.hpp"

#include <mutex>

#include <limits.h>

namespace ekf {


// ---------------------------------------------------------------------------------------------
// Constructor state, ekf_slam.cpp:27-53: Sigma0 = blockdiag(0_3x3, sigma0 * I_2n), state = 0.
// grid (row blocks, B); 256 threads; each block writes 8 rows with 16-B stores.
// ---------------------------------------------------------------------------------------------
constexpr int kInitRows = 8;

__global__ __launch_bounds__(256) void k_init(PoolView pv) {
    const int b = blockIdx.y;
    const int ld2 = pv.ld >> 1;
    double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    const int r0 = blockIdx.x * kInitRows;
    for (int rr = 0; rr < kInitRows; rr++) {
        const int r = r0 + rr;
        if (r >= pv.N) break;
        double2_t* row = reinterpret_cast<double2_t*>(Sg + (size_t)r * pv.ld);
        for (int c2 = threadIdx.x; c2 < ld2; c2 += 256) {
            double2_t v = {0.0, 0.0};
            if (r >= 3) {
                if (2 * c2 == r) v.x = 1.0 * pv.p.sigma0_landmark;
                if (2 * c2 + 1 == r) v.y = 1.0 * pv.p.sigma0_landmark;
            }
            row[c2] = v;
        }
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < pv.ld; i += 256) {
            pv.state[(size_t)b * pv.ld + i] = 0.0;
            pv.Kg[(size_t)b * 2 * pv.ld + i] = 0.0;
            pv.Kg[(size_t)b * 2 * pv.ld + pv.ld + i] = 0.0;
            pv.Gh[(size_t)b * 2 * pv.ld + i] = 0.0;
            pv.Gh[(size_t)b * 2 * pv.ld + pv.ld + i] = 0.0;
        }
        if (threadIdx.x < 4) pv.snap[(size_t)b * 4 + threadIdx.x] = 0.0;
        for (int i = threadIdx.x; i < pv.n; i += 256) pv.touch_flag[(size_t)b * pv.n + i] = 0;
        if (threadIdx.x == 0) {
            pv.rec[b] = CorrRec{0.0, 0.0, 0, -1, 0, 0};
            pv.assoc[b] = AssocRec{0, -1, 0, 0, 0.0};
            pv.touch_count[b] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// prediction(), ekf_slam.cpp:55-106.  At = I + A has two off-diagonal non-zeros (A(1,0), A(2,0),
// :85-86/:93-94), so At*Sigma*At^T + Q only changes rows 1,2 and columns 1,2 of Sigma: O(N) work,
// about 10*8*N bytes instead of two N^3 DGEMMs.  One workgroup per filter: every thread needs the
// OLD theta before thread 0 overwrites the pose, and __syncthreads() is the only fence needed.
// ---------------------------------------------------------------------------------------------
constexpr int kPredictThreads = 1024;

__global__ __launch_bounds__(kPredictThreads) void k_predict(PoolView pv, const double* twist_dev, double dth_imm,
                                                            double dx_imm, Pending pend, double* pred_out) {
    const int b = blockIdx.x;
    const int N = pv.N, ld = pv.ld;
    double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    double* st = pv.state + (size_t)b * ld;
    const double dtheta = twist_dev ? twist_dev[(size_t)b * 2 + 0] : dth_imm;  // twist.angular()  :67
    const double dx = twist_dev ? twist_dev[(size_t)b * 2 + 1] : dx_imm;       // twist.linearX()  :69
    const double theta = st[0];

    // Latency matters here (one workgroup per filter, a 10 Hz node calls this on a single filter): the
    // covariance entries do not depend on theta, so the first two trips' loads are issued before the
    // trigonometry and fly under it.
    constexpr int PRE = 2;
    double s0[PRE], s1[PRE], s2[PRE], r0[PRE], r1[PRE], r2[PRE];
    // pend.symmetric == 2 (a delayed known-association run with the mirrored flush): the tiles on and above the diagonal
    // are the covariance until the next flush mirrors them, so columns 1 and 2 are only kept up inside the first
    // diagonal square -- the 16-KB-strided sector per row (most of this kernel's traffic) is not touched
    // pend.colp (column panel of a delayed known-association run, see Pending): columns 0..2 live as the panel's rows
    // 0..2 (coalesced) and are kept current THERE; the matrix's own columns 1, 2 are left to the next flush
    double* __restrict__ cp = pend.colp ? pend.colp + (size_t)b * pend.colp_rows * ld : nullptr;
    const short* __restrict__ lms = pend.colp ? pend.lmslot + (size_t)b * pv.n : nullptr;
    const int col_rows = pend.symmetric == 2 ? kSymSquare : N;
    // active-set mode: row/column k of a never-corrected landmark is exactly zero against the pose block,
    // so its update a*0 + 0 = 0 is skipped (bit-identical)
    const unsigned char* tf = pv.touch_flag + (size_t)b * pv.n;
    bool live[PRE];
#pragma unroll
    for (int q = 0; q < PRE; q++) {
        const int k = 3 + (int)threadIdx.x + q * kPredictThreads;
        live[q] = k < N && (!pv.active_set || tf[(k - 3) >> 1]);
        if (live[q]) {
            s0[q] = Sg[k];                      // row 0 (coalesced)
            s1[q] = Sg[(size_t)1 * ld + k];
            s2[q] = Sg[(size_t)2 * ld + k];
            if (cp) {
                r0[q] = cp[k]; r1[q] = cp[(size_t)ld + k]; r2[q] = cp[(size_t)2 * ld + k];   // columns 0..2, as rows
            } else if (k < col_rows) {
                const double* rowk = Sg + (size_t)k * ld;  // columns 0..2 of row k (one 32-B sector)
                r0[q] = rowk[0]; r1[q] = rowk[1]; r2[q] = rowk[2];
            }
        }
    }
    double c[3][3], px = 0.0, py = 0.0;
    if (threadIdx.x == 0) {
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++) c[r][k] = Sg[(size_t)r * ld + k];
        px = st[1]; py = st[2];
    }

    // sin and cos of theta and of theta + dtheta: even lanes take the first angle, odd lanes the second (two calls deep
    // instead of four; same functions on the same arguments, bit for bit), broadcast through the scalar file
    const int lane = threadIdx.x & 63;
    const double arg = (lane & 1) ? theta + dtheta : theta;
    const double sv = sin(arg), cv = cos(arg);
    const double sin_t = lane_bcast(sv, 0), sin_td = lane_bcast(sv, 1), cos_t = lane_bcast(cv, 0), cos_td = lane_bcast(cv, 1);
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
    __syncthreads();  // all threads hold the old theta; thread 0 may now move the pose
    if (pred_out && threadIdx.x == 0) { pred_out[(size_t)b * 2] = a10; pred_out[(size_t)b * 2 + 1] = a20; }

#pragma unroll
    for (int q = 0; q < PRE; q++) {
        const int k = 3 + (int)threadIdx.x + q * kPredictThreads;
        if (live[q]) {
            double* rowk = Sg + (size_t)k * ld;
            const double n1 = a10 * s0[q] + s1[q], n2 = a20 * s0[q] + s2[q];
            Sg[(size_t)1 * ld + k] = n1;
            Sg[(size_t)2 * ld + k] = n2;
            if (cp) {
                cp[(size_t)ld + k] = r0[q] * a10 + r1[q];
                cp[(size_t)2 * ld + k] = r0[q] * a20 + r2[q];
                const int sl = lms[(k - 3) >> 1];
                if (sl >= 0) {   // a planned landmark column: its entries at the rows 1, 2
                    double* pr = cp + (size_t)(3 + 2 * sl + ((k - 3) & 1)) * ld;
                    pr[1] = n1; pr[2] = n2;
                }
            } else if (k < col_rows) {
                rowk[1] = r0[q] * a10 + r1[q];
                rowk[2] = r0[q] * a20 + r2[q];
            }
        }
    }
    for (int k = 3 + threadIdx.x + PRE * kPredictThreads; k < N; k += kPredictThreads) {
        if (pv.active_set && !tf[(k - 3) >> 1]) continue;
        const double t0 = Sg[k];
        const double t1 = Sg[(size_t)1 * ld + k];
        const double t2 = Sg[(size_t)2 * ld + k];
        const double n1 = a10 * t0 + t1, n2 = a20 * t0 + t2;
        Sg[(size_t)1 * ld + k] = n1;
        Sg[(size_t)2 * ld + k] = n2;
        if (cp) {
            const double q0 = cp[k], q1 = cp[(size_t)ld + k], q2 = cp[(size_t)2 * ld + k];
            cp[(size_t)ld + k] = q0 * a10 + q1;
            cp[(size_t)2 * ld + k] = q0 * a20 + q2;
            const int sl = lms[(k - 3) >> 1];
            if (sl >= 0) {
                double* pr = cp + (size_t)(3 + 2 * sl + ((k - 3) & 1)) * ld;
                pr[1] = n1; pr[2] = n2;
            }
        } else if (k < col_rows) {
            double* rowk = Sg + (size_t)k * ld;
            const double q0 = rowk[0], q1 = rowk[1], q2 = rowk[2];
            rowk[1] = q0 * a10 + q1;
            rowk[2] = q0 * a20 + q2;
        }
    }
    // Delayed-update mode: Sigma = Sigma_base - sum_j u_j v_j^T with pending factors; At (.) At^T maps the
    // base (above) and every factor: u <- At u, v <- At v (only entries 1 and 2 change).
    for (int j = threadIdx.x; j < pend.count; j += kPredictThreads) {
        double* u = pend.U + ((size_t)b * pend.cap + j) * ld;
        double* v = pend.V + ((size_t)b * pend.cap + j) * ld;
        const double u0_ = u[0], v0_ = v[0];
        const double u1 = a10 * u0_ + u[1], u2 = a20 * u0_ + u[2], v1 = a10 * v0_ + v[1], v2 = a20 * v0_ + v[2];
        u[1] = u1; u[2] = u2;
        v[1] = v1; v[2] = v2;
        if (pend.uvc) {   // the transposed copy of the factors at the panel's indices (see Pending::uvc): indices 1, 2
            double2_t* uc = pend.uvc + (size_t)b * pend.colp_rows * pend.cap;
            uc[(size_t)1 * pend.cap + j] = double2_t{u1, v1};
            uc[(size_t)2 * pend.cap + j] = double2_t{u2, v2};
        }
    }
    // The CURRENT rows / columns a delayed run keeps for the pose indices and its planned landmarks (Pending::cur) are parts
    // of Sigma as it stands now: At (.) At^T maps them like everything else -- the pose block here, the pose vectors' long
    // part deferred to the next gain launch, a landmark's column / row at its entries 1, 2 here.
    if (pend.cur) {
        const int slots = (pend.colp_rows - 3) / 2;
        double* cu = pend.cur + (size_t)b * (6 + 4 * slots) * ld;
        const int* cv = pend.curv_in + (size_t)b * (1 + slots);
        if (cv[0] >= 0) {
            // (the pose vectors over the indices >= 3 -- Sigma(k, 1) = Sigma(k, 0) a10 + Sigma(k, 1), Sigma(1, k) = a10 Sigma(0, k)
            // + Sigma(1, k), ... -- are left to the next gain launch, which applies the accumulated map to the values it
            // loads: see Pending::apred_in)
            if (threadIdx.x == 65) { pend.apred_in[(size_t)b * 2] += a10; pend.apred_in[(size_t)b * 2 + 1] += a20; }
            if (threadIdx.x == 64) {   // the pose block of the kept vectors, from ITS values (rows of cur: Sigma(r, k))
                double cc[3][3], T[3][3], Sn[3][3];
                for (int r = 0; r < 3; r++)
                    for (int k = 0; k < 3; k++) cc[r][k] = cu[(size_t)(3 + r) * ld + k];
                for (int k = 0; k < 3; k++) {
                    T[0][k] = cc[0][k];
                    T[1][k] = a10 * cc[0][k] + cc[1][k];
                    T[2][k] = a20 * cc[0][k] + cc[2][k];
                }
                for (int r = 0; r < 3; r++) {
                    Sn[r][0] = T[r][0];
                    Sn[r][1] = T[r][0] * a10 + T[r][1];
                    Sn[r][2] = T[r][0] * a20 + T[r][2];
                }
                Sn[0][0] += pv.p.q_pose; Sn[1][1] += pv.p.q_pose; Sn[2][2] += pv.p.q_pose;
                for (int r = 0; r < 3; r++)
                    for (int k = 0; k < 3; k++) { cu[(size_t)(3 + r) * ld + k] = Sn[r][k]; cu[(size_t)k * ld + r] = Sn[r][k]; }
            }
        }
        if ((int)threadIdx.x < 4 * slots) {
            const int sl = threadIdx.x >> 2, q = threadIdx.x & 3;
            if (cv[1 + sl] >= 0) {
                double* vec = cu + (size_t)(6 + 4 * sl + q) * ld;
                const double e0 = vec[0], e1 = vec[1], e2 = vec[2];
                if (q < 2) { vec[1] = a10 * e0 + e1; vec[2] = a20 * e0 + e2; }   // a column: Sigma(1, c), Sigma(2, c)
                else { vec[1] = e0 * a10 + e1; vec[2] = e0 * a20 + e2; }          // a row: Sigma(c, 1), Sigma(c, 2)
            }
        }
    }
    if (threadIdx.x == 0) {
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
            Sg[(size_t)r * ld + 0] = T[r][0];
            Sg[(size_t)r * ld + 1] = T[r][0] * a10 + T[r][1];
            Sg[(size_t)r * ld + 2] = T[r][0] * a20 + T[r][2];
        }
        Sg[0] += pv.p.q_pose;  // Q = diag(q,q,q,0...) :40-43
        Sg[(size_t)1 * ld + 1] += pv.p.q_pose;
        Sg[(size_t)2 * ld + 2] += pv.p.q_pose;
        if (cp)   // the pose block, mirrored into the panel (row c of the panel = column c of Sigma)
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) cp[(size_t)k * ld + r] = Sg[(size_t)r * ld + k];
    }
}

// ---------------------------------------------------------------------------------------------
// Top of measurement(), ekf_slam.cpp:109-128: capture (theta,x,y) once -- the per-landmark loop
// keeps using this STALE pose -- and, on the first call only, initialise ALL n landmarks from the
// sensor vector regardless of visibility.  grid (ceil(n/256), B).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_measure_begin(PoolView pv, const double* init_xy, int do_init) {
    const int b = blockIdx.y;
    double* st = pv.state + (size_t)b * pv.ld;
    const double theta = st[0], x = st[1], y = st[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double* sn = pv.snap + (size_t)b * 4;
        sn[0] = theta; sn[1] = x; sn[2] = y;
    }
    if (do_init) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < pv.n) {
            const double sx = init_xy[(size_t)b * 2 * pv.n + 2 * i];
            const double sy = init_xy[(size_t)b * 2 * pv.n + 2 * i + 1];
            const double ri = sqrt(sx * sx + sy * sy);
            const double phii = atan2(sy, sx);
            st[2 * i + 3] = x + ri * cos(phii + theta);
            st[2 * i + 3 + 1] = y + ri * sin(phii + theta);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Gain kernel: everything of one landmark correction (ekf_slam.cpp:137-187) except the O(N^2)
// covariance stream.  H has 5 non-zero columns c5 = {0,1,2,3+2i,4+2i}, so
//   S  = H5 * Sigma[c5,c5] * H5^T + R           (25 covariance entries)
//   K  = Sigma[:,c5] * H5^T * S^-1              (5 strided column reads per row)
//   G  = H5 * Sigma[c5,:]                       (5 coalesced row reads per column)
// Reads Sigma/state only, writes scratch (Kg, Gh, rec) only -> race-free across workgroups; the
// state update state += K*nu (:186-187) happens in the rank-2 kernel, after the kernel boundary.
// grid (ceil((N + 1)/256), B), 256 threads: rows up to the active dimension N (a discovered prefix in
// data_association(); each filter of a pool may narrow it further, CorrRec.n_active).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gain(PoolView pv, CmdSrc src) {
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    int N = pv.N;
    const int ld = pv.ld;
    __shared__ double sh_S55[25];
    __shared__ double sh_H[10];
    __shared__ double sh_Si[4];

    int lm = -1;
    int n_active = 0;
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
        if (src.min_active > 0) {  // this filter's own discovered prefix
            n_active = max(src.min_active, 3 + 2 * a.known_count);
            if (n_active < N) N = n_active; else n_active = 0;
        }
        sx = src.meas[(size_t)b * src.meas_stride];
        sy = src.meas[(size_t)b * src.meas_stride + 1];
    }
    if (lm < 0 || lm >= pv.n) {  // nothing to correct for this filter in this slot
        if (blockIdx.x == 0 && tid == 0) pv.rec[b].active = 0;
        return;
    }
    // K and G beyond this filter's active dimension are never read by its rank-2 pass
    if (n_active > 0 && blockIdx.x > 0 && blockIdx.x * 256 >= N) return;

    const double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    const double* st = pv.state + (size_t)b * ld;
    // Everything this workgroup reads depends only on lm, so all loads are issued up front and fly
    // together (latency matters for a single filter): the lane's 5 column + 5 row entries, the 5x5
    // block, and -- on lane 0 -- the pose and the landmark.
    const int r = blockIdx.x * 256 + tid;
    double p[5], g[5];
    // active-set mode: a row of a never-corrected landmark (other than lm itself) has Sigma(r, c5) = 0 exactly,
    // hence K(r,:) = 0: the 16-KB-strided column gather is skipped for it (bit-identical downstream)
    const bool krow = r < N && (!pv.active_set || r < 3 || ((r - 3) >> 1) == lm ||
                                pv.touch_flag[(size_t)b * pv.n + ((r - 3) >> 1)]);
    if (r < N) {
        if (krow) gather_row5(Sg + (size_t)r * ld, lm, p);   // column gather (Sigma * H^T reads columns): three loads
        else {
#pragma unroll
            for (int k = 0; k < 5; k++) p[k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) g[k] = Sg[(size_t)idx5(k, lm) * ld + r];   // row gather (H * Sigma reads rows)
    }
    if (tid < 25) sh_S55[tid] = Sg[(size_t)idx5(tid / 5, lm) * ld + idx5(tid % 5, lm)];
    double theta = 0.0, x = 0.0, y = 0.0, tx = 0.0, ty = 0.0;
    if (tid == 0) {
        if (src.fresh_pose) {  // data_association re-reads the pose per measurement, :331-333
            theta = st[0]; x = st[1]; y = st[2];
        } else {               // measurement() keeps the pose captured at :109-111
            const double* sn = pv.snap + (size_t)b * 4;
            theta = sn[0]; x = sn[1]; y = sn[2];
        }
        tx = st[2 * lm + 3];
        ty = st[2 * lm + 4];
        if (src.write_snap && blockIdx.x == 0) {
            double* sn = pv.snap + (size_t)b * 4;
            sn[0] = theta; sn[1] = x; sn[2] = y;
        }
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
        if (blockIdx.x == 0) {
            CorrRec rc;
            rc.nu0 = m.z0 - m.zh0;                    // :182
            rc.nu1 = normalize_angle(m.z1 - m.zh1);   // :183
            rc.active = 1;
            rc.lm = lm;
            rc.n_active = n_active;
            rc.pad = 0;
            pv.rec[b] = rc;
            touch_landmark(pv, b, lm);
        }
    }
    __syncthreads();

    if (r >= ld) return;
    double k0 = 0.0, k1 = 0.0, g0 = 0.0, g1 = 0.0;
    if (r < N) {
        double sht0 = 0.0, sht1 = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            sht0 += p[k] * sh_H[k];
            sht1 += p[k] * sh_H[5 + k];
            g0 += sh_H[k] * g[k];
            g1 += sh_H[5 + k] * g[k];
        }
        k0 = sht0 * sh_Si[0] + sht1 * sh_Si[2];  // K = (Sigma H^T) S^-1   :178
        k1 = sht0 * sh_Si[1] + sht1 * sh_Si[3];
    }
    double2_t* Kg = reinterpret_cast<double2_t*>(pv.Kg + (size_t)b * 2 * ld);
    double* Gh = pv.Gh + (size_t)b * 2 * ld;
    Kg[r] = double2_t{k0, k1};
    Gh[r] = g0;
    Gh[ld + r] = g1;
}

// ---------------------------------------------------------------------------------------------
// Rank-2 covariance correction, the bandwidth-bound core: Sigma <- (I - K H) Sigma
// (ekf_slam.cpp:191-192) evaluated as Sigma[r][c] -= K[r][0]*G[0][c] + K[r][1]*G[1][c].
// Algorithmic traffic 2*8*N^2 bytes (read + write Sigma once), 4 N^2 flop -> HBM-bound.
//
// A workgroup owns a strip of 256 double2 columns (4 KiB per row; every wave instruction moves
// 1 KiB contiguous) and rows_per_block rows.  A lane keeps its two G values in registers and streams
// its column down the rows in groups of U rows through a ring of three register buffers: the U loads
// of group g+1 are issued BEFORE the U stores of group g, so in the in-order vmcnt queue a load is
// older than the stores around it and waiting for it does not drain them.  The main loop has no branch
// (the lane-validity test wraps it), which keeps hipcc's s_waitcnt counts exact.  K(r,:) is
// wave-uniform and arrives through the scalar cache (lgkmcnt), off the vector-memory queue.
// Sigma, K, G, rec and state never alias (__restrict__), so those scalar loads stay legal after the
// first Sigma store.  The same workgroups apply state += K*nu (:186) and the theta wrap (:187).
// grid (ceil(ceil(N/2)/256), row blocks, B).
// ---------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ double2_t ld2(const double2_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st2(double2_t* p, double2_t v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// TPB: 256 lanes per strip, or 64 / 128 / 192 when the active row is shorter than that (a narrow map, or a
// discovered prefix): the waves of a workgroup share nothing, so a narrower workgroup only sheds idle wavefronts.
template <int U, bool NT, int TPB>
__device__ __forceinline__ void rank2_tile(double* __restrict__ sigma, const double* __restrict__ Kg_all,
                                           const double* __restrict__ Gh_all, const CorrRec* __restrict__ rec,
                                           double* __restrict__ state, int N_launch, int ld, size_t sigma_stride,
                                           int rows_per_block, int bx, int by, int b) {
    // N is the ACTIVE dimension: rows and columns >= N are exactly untouched by this correction (their K
    // and G entries are exact zeros), which data_association() exploits -- landmarks are appended in
    // discovery order, so everything beyond 3 + 2*known_count still holds its constructor value.
    if (!rec[b].active) return;
    const int na = rec[b].n_active;  // per-filter discovered prefix (batched data association), 0 = none
    const int N = na > 0 ? min(na, N_launch) : N_launch;
    const int ld2n = ld >> 1;
    const int ld2a = (N + 1) >> 1;  // double2 columns that hold an active column
    const int c2 = bx * TPB + threadIdx.x;
    const double2_t* __restrict__ Kg = reinterpret_cast<const double2_t*>(Kg_all + (size_t)b * 2 * ld);
    const int row_begin = by * rows_per_block;
    const int row_end = min(N, row_begin + rows_per_block);
    if (row_begin >= N) return;

    if (c2 < ld2a) {
        const double2_t g0 = reinterpret_cast<const double2_t*>(Gh_all + (size_t)b * 2 * ld)[c2];
        const double2_t g1 = reinterpret_cast<const double2_t*>(Gh_all + (size_t)b * 2 * ld + ld)[c2];
        double2_t* __restrict__ col = reinterpret_cast<double2_t*>(sigma + (size_t)b * sigma_stride) + c2;

        auto load_group = [&](double2_t (&buf)[U], int row) {
#pragma unroll
            for (int u = 0; u < U; u++) buf[u] = ld2<NT>(col + (size_t)(row + u) * ld2n);
        };
        auto finish_group = [&](double2_t (&buf)[U], int row) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double2_t k = Kg[row + u];  // uniform address -> s_load
                double2_t v = buf[u];
                v.x = v.x - (k.x * g0.x + k.y * g1.x);
                v.y = v.y - (k.x * g0.y + k.y * g1.y);
                st2<NT>(col + (size_t)(row + u) * ld2n, v);
            }
        };

        const int nfull = (row_end - row_begin) / U;  // uniform
        int r = row_begin;
        if (nfull > 0) {
            // Three register buffers: hipcc orders a reload of a buffer behind the completion of the
            // stores that read it (vmcnt), so the buffer reloaded in a sub-step must be the one stored
            // TWO sub-steps ago -- then U loads + U stores stay in flight across every wait.
            double2_t A[U], Bf[U], Cf[U];
            load_group(A, r);
            int g = 0;
            for (; g + 3 < nfull; g += 3) {  // invariant: A holds group g; no branch inside
                load_group(Bf, r + U);
                finish_group(A, r);
                load_group(Cf, r + 2 * U);
                finish_group(Bf, r + U);
                load_group(A, r + 3 * U);
                finish_group(Cf, r + 2 * U);
                r += 3 * U;
            }
            const int rem = nfull - g;  // 1, 2 or 3 groups left, A already loaded
            if (rem == 1) {
                finish_group(A, r);
            } else if (rem == 2) {
                load_group(Bf, r + U);
                finish_group(A, r);
                finish_group(Bf, r + U);
            } else {
                load_group(Bf, r + U);
                finish_group(A, r);
                load_group(Cf, r + 2 * U);
                finish_group(Bf, r + U);
                finish_group(Cf, r + 2 * U);
            }
            r += rem * U;
        }
        if (r < row_end) {  // < U leftover rows of the last row block: all their loads fly together
            const int rem = row_end - r;  // uniform
            double2_t T[U];
#pragma unroll
            for (int u = 0; u < U - 1; u++)
                if (u < rem) T[u] = ld2<NT>(col + (size_t)(r + u) * ld2n);
#pragma unroll
            for (int u = 0; u < U - 1; u++)
                if (u < rem) {
                    const double2_t k = Kg[r + u];
                    double2_t v = T[u];
                    v.x = v.x - (k.x * g0.x + k.y * g1.x);
                    v.y = v.y - (k.x * g0.y + k.y * g1.y);
                    st2<NT>(col + (size_t)(r + u) * ld2n, v);
                }
        }
    }

    // state = state + Ki*z_diff (:186); state(0) = normalize_angle(state(0)) (:187)
    if (bx == 0) {
        const CorrRec rc = rec[b];
        double* st = state + (size_t)b * ld;
        for (int rr = row_begin + (int)threadIdx.x; rr < row_end; rr += TPB) {
            const double2_t k = Kg[rr];
            double s = st[rr] + (k.x * rc.nu0 + k.y * rc.nu1);
            if (rr == 0) s = normalize_angle(s);
            st[rr] = s;
        }
    }
}

template <int U, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void k_rank2(double* __restrict__ sigma, const double* __restrict__ Kg_all,
                                               const double* __restrict__ Gh_all, const CorrRec* __restrict__ rec,
                                               double* __restrict__ state, int N_launch, int ld,
                                               size_t sigma_stride, int rows_per_block) {
    rank2_tile<U, NT, TPB>(sigma, Kg_all, Gh_all, rec, state, N_launch, ld, sigma_stride, rows_per_block, blockIdx.x, blockIdx.y,
                           blockIdx.z);
}

// The same tiles taken from ONE queue by resident workgroups (one per CU's worth of registers), for pools with many
// more tiles than CUs.  The dispatcher deals workgroup i to XCD i % 8 -- a fixed eighth of the grid each -- and XCDs (and
// CUs) do not stream at the same rate: tools/micro/strip_walk.hip moves the same bytes in 40.9 ms with short-lived
// workgroups, 40.7 ms with a queue per XCD and 38.7 ms (6.94 TB/s) with one queue for the chip.  A workgroup asks for its
// next tile (one atomicAdd, issued a tile ahead) when it is done with the last; tiles run column chunks fastest, then row
// blocks, then filters, so the chip sweeps one filter at a time.  Same arithmetic per element: bit-identical.
template <int U, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void k_rank2_queue(double* __restrict__ sigma, const double* __restrict__ Kg_all,
                                                     const double* __restrict__ Gh_all, const CorrRec* __restrict__ rec,
                                                     double* __restrict__ state, int N_launch, int ld, size_t sigma_stride,
                                                     int rows_per_block, int chunks, int row_blocks, unsigned total,
                                                     unsigned* __restrict__ queue) {
    __shared__ unsigned sh_next[2];
    unsigned ahead = 0;
    if (threadIdx.x == 0) ahead = atomicAdd(queue, 1u);
    for (int par = 0;; par ^= 1) {
        if (threadIdx.x == 0) sh_next[par] = ahead;
        __syncthreads();   // (one barrier per tile: the slot written now is read by everybody before it is written again two tiles on)
        const unsigned s = __builtin_amdgcn_readfirstlane(sh_next[par]);   // (uniform: K and rec stay scalar loads in the tile body)
        if (s >= total) break;
        if (threadIdx.x == 0) ahead = atomicAdd(queue, 1u);   // the tile after this one: back long before it is needed
        const int bx = s % chunks, by = (s / chunks) % row_blocks, b = s / ((unsigned)chunks * row_blocks);
        rank2_tile<U, NT, TPB>(sigma, Kg_all, Gh_all, rec, state, N_launch, ld, sigma_stride, rows_per_block, bx, by, b);
    }
}

// ---------------------------------------------------------------------------------------------
// Row-PACKED form of the same update for pools whose rows do not fill the 256-lane strips (n = 200: 208 of 256 lanes
// carry a column, 0.68 of peak): rows are contiguous in memory (ld doubles each, no gap), so P consecutive rows are ONE
// contiguous virtual row of P * ld/2 double2 columns, and P is chosen so that the strips of 256 lanes tile it almost
// exactly (n = 200: P = 6 -> 1248 of 1280 lanes).  A lane's slot s fixes its physical sub-row p = s / (ld/2) and its
// column pair c2 = s % (ld/2), hence its two G values; going down the virtual rows it visits rows vr * P + p.  K(row, :)
// is uniform except in a wavefront that straddles two sub-rows (at most two: ld/2 >= 64), where each lane selects one
// of two scalar-loaded values.  Same expression per element as k_rank2 -> bit-identical.  Pad columns are streamed too
// (their G is 0).  Full-width launches only: every K and G entry up to ld must be valid (known-association paths).
// grid (ceil(P * ld/2 / 256), virtual row blocks, B).
// ---------------------------------------------------------------------------------------------
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_rank2_packed(double* __restrict__ sigma, const double* __restrict__ Kg_all,
                                                      const double* __restrict__ Gh_all, const CorrRec* __restrict__ rec,
                                                      double* __restrict__ state, int N, int ld, size_t sigma_stride, int P,
                                                      int vrows_per_block) {
    const int b = blockIdx.z;
    if (!rec[b].active) return;
    const int ld2n = ld >> 1;
    const int width = P * ld2n;                    // double2 columns of a virtual row
    const int s = blockIdx.x * 256 + threadIdx.x;  // slot in the virtual row
    const int nvfull = N / P;                      // virtual rows whose P sub-rows all exist
    const int vr_begin = blockIdx.y * vrows_per_block;
    if (vr_begin * P >= N) return;
    const int vr_end = min(nvfull, vr_begin + vrows_per_block);
    const double2_t* __restrict__ Kg = reinterpret_cast<const double2_t*>(Kg_all + (size_t)b * 2 * ld);

    if (s < width) {
        const int p = s / ld2n, c2 = s - p * ld2n;
        // sub-rows of this wavefront: p_lo for its first lane, p_lo + 1 beyond the row boundary
        const int wave_base = blockIdx.x * 256 + (threadIdx.x & ~63);
        const int p_lo = __builtin_amdgcn_readfirstlane(wave_base / ld2n);
        const bool hi = p != p_lo;
        const int p_hi = min(p_lo + 1, P - 1);
        const double2_t g0 = reinterpret_cast<const double2_t*>(Gh_all + (size_t)b * 2 * ld)[c2];
        const double2_t g1 = reinterpret_cast<const double2_t*>(Gh_all + (size_t)b * 2 * ld + ld)[c2];
        double2_t* __restrict__ col = reinterpret_cast<double2_t*>(sigma + (size_t)b * sigma_stride) + s;

        auto load_group = [&](double2_t (&buf)[U], int vr) {
#pragma unroll
            for (int u = 0; u < U; u++) buf[u] = ld2<NT>(col + (size_t)(vr + u) * width);
        };
        auto finish_group = [&](double2_t (&buf)[U], int vr) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const double2_t klo = Kg[(vr + u) * P + p_lo], khi = Kg[(vr + u) * P + p_hi];  // uniform -> s_load
                const double2_t k = hi ? khi : klo;
                double2_t v = buf[u];
                v.x = v.x - (k.x * g0.x + k.y * g1.x);
                v.y = v.y - (k.x * g0.y + k.y * g1.y);
                st2<NT>(col + (size_t)(vr + u) * width, v);
            }
        };
        const int nfull = vr_end > vr_begin ? (vr_end - vr_begin) / U : 0;  // uniform
        int r = vr_begin;
        if (nfull > 0) {
            double2_t A[U], Bf[U], Cf[U];
            load_group(A, r);
            int g = 0;
            for (; g + 3 < nfull; g += 3) {  // invariant: A holds group g; no branch inside
                load_group(Bf, r + U);
                finish_group(A, r);
                load_group(Cf, r + 2 * U);
                finish_group(Bf, r + U);
                load_group(A, r + 3 * U);
                finish_group(Cf, r + 2 * U);
                r += 3 * U;
            }
            const int rem = nfull - g;
            if (rem == 1) {
                finish_group(A, r);
            } else if (rem == 2) {
                load_group(Bf, r + U);
                finish_group(A, r);
                finish_group(Bf, r + U);
            } else {
                load_group(Bf, r + U);
                finish_group(A, r);
                load_group(Cf, r + 2 * U);
                finish_group(Bf, r + U);
                finish_group(Cf, r + 2 * U);
            }
            r += rem * U;
        }
        // leftover virtual rows of the block, and the ragged last virtual row of the matrix (sub-rows beyond N - 1 absent)
        const int vr_last = min((N + P - 1) / P, vr_begin + vrows_per_block);
        for (; r < vr_last; r++) {
            const int row = r * P + p;
            if (row < N) {
                const double2_t k = Kg[row];
                double2_t v = col[(size_t)r * width];
                v.x = v.x - (k.x * g0.x + k.y * g1.x);
                v.y = v.y - (k.x * g0.y + k.y * g1.y);
                col[(size_t)r * width] = v;
            }
        }
    }

    // state = state + Ki*z_diff (:186); state(0) = normalize_angle(state(0)) (:187)
    if (blockIdx.x == 0) {
        const CorrRec rc = rec[b];
        double* st = state + (size_t)b * ld;
        const int row_begin = vr_begin * P, row_end = min(N, (vr_begin + vrows_per_block) * P);
        for (int rr = row_begin + (int)threadIdx.x; rr < row_end; rr += 256) {
            const double2_t k = Kg[rr];
            double sv = st[rr] + (k.x * rc.nu0 + k.y * rc.nu1);
            if (rr == 0) sv = normalize_angle(sv);
            st[rr] = sv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Active-set form of the same update (opt-in, reported separately): only the rows of the touched set
// -- pose rows and the rows of landmarks that have ever been corrected -- can have K != 0, so only they
// are streamed: 2*8*N*(3 + 2*touched) bytes instead of 2*8*N^2.  Bit-identical to k_rank2 for finite
// states.  A workgroup owns a strip of 256 double2 columns and rows_per_block SLOTS of the row list.
// grid (strips, ceil((3 + 2*max_touched)/rows_per_block), B).
// ---------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void k_rank2_active(PoolView pv, int rows_per_block) {
    const int b = blockIdx.z;
    if (!pv.rec[b].active) return;
    const int N = pv.N, ld = pv.ld, ld2n = ld >> 1, ld2a = (N + 1) >> 1;
    const int nslots = 3 + 2 * pv.touch_count[b];
    const int s0 = blockIdx.y * rows_per_block;
    if (s0 >= nslots) return;
    const int s1 = min(nslots, s0 + rows_per_block);
    const int* __restrict__ list = pv.touch_list + (size_t)b * pv.n;
    auto slot_row = [&](int s) { return s < 3 ? s : 3 + 2 * list[(s - 3) >> 1] + ((s - 3) & 1); };
    const double2_t* __restrict__ Kg = reinterpret_cast<const double2_t*>(pv.Kg + (size_t)b * 2 * ld);
    const int c2 = blockIdx.x * 256 + threadIdx.x;
    if (c2 < ld2a) {
        const double2_t g0 = reinterpret_cast<const double2_t*>(pv.Gh + (size_t)b * 2 * ld)[c2];
        const double2_t g1 = reinterpret_cast<const double2_t*>(pv.Gh + (size_t)b * 2 * ld + ld)[c2];
        double2_t* col = reinterpret_cast<double2_t*>(pv.sigma + (size_t)b * pv.sigma_stride) + c2;
        constexpr int U = 8;
        for (int s = s0; s < s1; s += U) {
            int rows[U];
            double2_t v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                rows[u] = s + u < s1 ? slot_row(s + u) : -1;
                if (rows[u] >= 0) v[u] = ld2<NT>(col + (size_t)rows[u] * ld2n);
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (rows[u] >= 0) {
                    const double2_t k = Kg[rows[u]];
                    v[u].x = v[u].x - (k.x * g0.x + k.y * g1.x);
                    v[u].y = v[u].y - (k.x * g0.y + k.y * g1.y);
                    st2<NT>(col + (size_t)rows[u] * ld2n, v[u]);
                }
        }
    }
    if (blockIdx.x == 0) {  // state = state + Ki*z_diff (:186), theta wrap (:187), touched rows only
        const CorrRec rc = pv.rec[b];
        double* st = pv.state + (size_t)b * ld;
        for (int s = s0 + (int)threadIdx.x; s < s1; s += 256) {
            const int rr = slot_row(s);
            const double2_t k = Kg[rr];
            double sv = st[rr] + (k.x * rc.nu0 + k.y * rc.nu1);
            if (rr == 0) sv = normalize_angle(sv);
            st[rr] = sv;
        }
    }
}

__global__ void k_touch_all(PoolView pv) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < pv.n; i += blockDim.x) {
        pv.touch_flag[(size_t)b * pv.n + i] = 1;
        pv.touch_list[(size_t)b * pv.n + i] = i;
    }
    if (threadIdx.x == 0) pv.touch_count[b] = pv.n;
}

// ---------------------------------------------------------------------------------------------
// Mahalanobis scores, calculate_maha_dis() ekf_slam.cpp:217-276: ONE LANDMARK PER WAVEFRONT.
// 25 lanes fetch the 5x5 sub-block Sigma[c5,c5] in one go; H*Sigma*H^T is folded with wave
// shuffles in the CPU restatement's summation order; the innovation bearing is NOT wrapped (:269).
// grid (ceil(M/4), B), 4 waves per workgroup; M = host bound of the known count (n without one).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maha(PoolView pv, MeasSrc ms, double* scores, int m_override, Pending pend) {
    const int b = blockIdx.y;
    if (ms.count && ms.j >= ms.count[b]) return;  // this filter has no measurement in this slot
    const double meas[2] = {ms.xy[(size_t)b * ms.stride], ms.xy[(size_t)b * ms.stride + 1]};
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int i = blockIdx.x * 4 + wave;
    const int M = m_override >= 0 ? m_override : pv.assoc[b].known_count;
    if (i >= M || i >= pv.n) return;  // wave-uniform
    const int ld = pv.ld;
    const double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    const double* st = pv.state + (size_t)b * ld;

    double v = 0.0;
    if (lane < 25) {
        const int rr = idx5(lane / 5, i), cc = idx5(lane % 5, i);
        v = Sg[(size_t)rr * ld + cc];
        // delayed mode: the entry of the CURRENT covariance = base entry minus the pending pairs, in correction order
        // (the arithmetic of k_gain_delayed's 5x5 block)
        const double* Ub = pend.U + (size_t)b * pend.cap * ld;
        const double* Vb = pend.V + (size_t)b * pend.cap * ld;
        for (int j = 0; j < pend.count; j += 2)
            v = __builtin_fma(-Ub[(size_t)(j + 1) * ld + rr], Vb[(size_t)(j + 1) * ld + cc],
                              __builtin_fma(-Ub[(size_t)j * ld + rr], Vb[(size_t)j * ld + cc], v));
    }

    MeasTerms m;  // fresh pose per score, :219-221
    measurement_terms(st[2 * i + 3], st[2 * i + 4], meas[0], meas[1], st[0], st[1], st[2], m);

    // lanes l = 0..4 build column l of H*Sigma[c5,c5]
    const int l5 = lane < 5 ? lane : 4;
    double hs0 = 0.0, hs1 = 0.0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const double vk = __shfl(v, k * 5 + l5, kWave);
        hs0 += m.H[0][k] * vk;
        hs1 += m.H[1][k] * vk;
    }
    double S[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
    for (int l = 0; l < 5; l++) {
        const double h0 = __shfl(hs0, l, kWave), h1 = __shfl(hs1, l, kWave);
        S[0][0] += h0 * m.H[0][l];
        S[0][1] += h0 * m.H[1][l];
        S[1][0] += h1 * m.H[0][l];
        S[1][1] += h1 * m.H[1][l];
    }
    S[0][0] += pv.p.r_meas;
    S[1][1] += pv.p.r_meas;
    double Si[2][2];
    inv2(S, Si);
    const double v0 = m.z0 - m.zh0, v1 = m.z1 - m.zh1;
    const double t0 = v0 * Si[0][0] + v1 * Si[1][0];
    const double t1 = v0 * Si[0][1] + v1 * Si[1][1];
    if (lane == 0) {
        scores[(size_t)b * pv.n + i] = t0 * v0 + t1 * v1;
    }
}

__global__ void k_assoc_begin(PoolView pv, const int* known_count_dev, int known_count_imm) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= pv.B) return;
    AssocRec a;
    a.known_count = known_count_dev ? known_count_dev[b] : known_count_imm;
    a.lm = -1; a.active = 0; a.pad = 0; a.best = 0.0;
    pv.assoc[b] = a;
}

// ---------------------------------------------------------------------------------------------
// Decision step of data_association() for one measurement, ekf_slam.cpp:293-330: sequential-scan
// semantics (first index attaining the strict minimum below gate_new) as a lexicographic (d, i)
// min-reduction; NaN scores never win (`d < min` is false).  New landmark initialisation
// (:200-214, :318-327) and the gate_update test (:330) run on one lane.  grid (B), 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assoc_decide(PoolView pv, MeasSrc ms, const double* scores,
                                                      int* assoc_out, int out_stride, int j,
                                                      unsigned long long* corr_counter) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    if (ms.count && ms.j >= ms.count[b]) {  // no measurement in this slot: nothing is decided, nothing corrected
        if (tid == 0) {
            pv.assoc[b].lm = -1;
            pv.assoc[b].active = 0;
            if (assoc_out) assoc_out[(size_t)b * out_stride + j] = -2;
        }
        return;
    }
    const double* meas = ms.xy + (size_t)b * ms.stride;
    __shared__ double sh_d[4];
    __shared__ int sh_i[4];
    const int M = pv.assoc[b].known_count;
    double best = pv.p.gate_new;  // :293
    int bi = INT_MAX;
    for (int i = tid; i < M; i += 256) {
        const double d = scores[(size_t)b * pv.n + i];
        if (d < best) { best = d; bi = i; }  // :305-309
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_down(best, off, kWave);
        const int oi = __shfl_down(bi, off, kWave);
        if (od < best || (od == best && oi < bi)) { best = od; bi = oi; }
    }
    if ((tid & 63) == 0) { sh_d[tid >> 6] = best; sh_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++)
            if (sh_d[w] < best || (sh_d[w] == best && sh_i[w] < bi)) { best = sh_d[w]; bi = sh_i[w]; }
        int idx = (bi == INT_MAX) ? M : bi;  // :294 min_maha_idx = known_count
        int known_count = M;
        double* st = pv.state + (size_t)b * pv.ld;
        if (idx == M && idx < pv.n) {  // :318-327 new landmark
            const double theta = st[0], x = st[1], y = st[2];
            const double sx = meas[0], sy = meas[1];
            const double ri = sqrt(sx * sx + sy * sy);
            const double phii = atan2(sy, sx);
            st[2 * idx + 3] = x + ri * cos(phii + theta);
            st[2 * idx + 3 + 1] = y + ri * sin(phii + theta);
            known_count = M + 1;
            best = 0.0;
        }
        // :330.  idx == n (map full, no match) can only pass the gate with non-reference
        // parameters (gate_new < gate_update); the reference would index out of bounds there.
        const int active = (best < pv.p.gate_update) && idx < pv.n;
        AssocRec a;
        a.known_count = known_count;
        a.lm = active ? idx : -1;
        a.active = active;
        a.pad = 0;
        a.best = best;
        pv.assoc[b] = a;
        if (assoc_out) assoc_out[(size_t)b * out_stride + j] = a.lm;
        if (corr_counter && active) atomicAdd(corr_counter, 1ull);
    }
}

// per-filter digest {sum state, sum |state|, sum sigma, sum |sigma|} -> out[b][4] (atomics)
__global__ __launch_bounds__(256) void k_checksum(PoolView pv, double* out) {
    const int b = blockIdx.y;
    const int N = pv.N, ld = pv.ld;
    const double* Sg = pv.sigma + (size_t)b * pv.sigma_stride;
    double s = 0.0, sa = 0.0, t = 0.0, ta = 0.0;
    for (int r = blockIdx.x; r < N; r += gridDim.x) {
        for (int c = threadIdx.x; c < N; c += 256) {
            const double v = Sg[(size_t)r * ld + c];
            t += v; ta += fabs(v);
        }
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < N; c += 256) {
            const double v = pv.state[(size_t)b * ld + c];
            s += v; sa += fabs(v);
        }
    __shared__ double sh[4][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, kWave); sa += __shfl_down(sa, off, kWave);
        t += __shfl_down(t, off, kWave); ta += __shfl_down(ta, off, kWave);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = s; sh[threadIdx.x >> 6][1] = sa;
        sh[threadIdx.x >> 6][2] = t; sh[threadIdx.x >> 6][3] = ta;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        atomicAdd(out + (size_t)b * 4 + threadIdx.x, v);
    }
}

__global__ void k_gather_poses(PoolView pv, double* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= pv.B) return;
    for (int k = 0; k < 3; k++) out[(size_t)b * 3 + k] = pv.state[(size_t)b * pv.ld + k];
}

// ---- launchers -------------------------------------------------------------------------------

// The result of a data_association() call handed to the host without a copy engine and without a stream synchronisation
// (measured, tools/micro/d2h_latency.hip: kernel + hipMemcpyAsync + hipStreamSynchronize 15.4 us, kernel that writes mapped
// host memory + a host spin on a flag word 6.8 us): record and decisions first, system-scope fence, then the sequence number.
__global__ __launch_bounds__(64) void k_publish_assoc(const AssocRec* __restrict__ rec, const int* __restrict__ decisions, int J,
                                                      char* host, unsigned seq) {
    int* hd = reinterpret_cast<int*>(host + kAssocDecOff);
    for (int j = threadIdx.x; j < J; j += 64) hd[j] = decisions[j];
    if (threadIdx.x == 0) *reinterpret_cast<AssocRec*>(host) = rec[0];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned*>(host + kAssocSeqOff) = seq;
}

void launch_publish_assoc(const AssocRec* rec, const int* decisions, int J, char* host, unsigned seq, hipStream_t s) {
    hipLaunchKernelGGL(k_publish_assoc, dim3(1), dim3(64), 0, s, rec, decisions, J, host, seq);
}

void launch_init(const PoolView& pv, hipStream_t s) {
    dim3 grid((pv.N + kInitRows - 1) / kInitRows, pv.B);
    hipLaunchKernelGGL(k_init, grid, dim3(256), 0, s, pv);
}

void launch_predict(const PoolView& pv, const double* twist_dev, double dtheta, double dx, const Pending& pend,
                    hipStream_t s, double* pred_out) {
    hipLaunchKernelGGL(k_predict, dim3(pv.B), dim3(kPredictThreads), 0, s, pv, twist_dev, dtheta, dx, pend, pred_out);
}

void launch_measure_begin(const PoolView& pv, const double* init_xy, int do_init, hipStream_t s) {
    const int gx = do_init ? (pv.n + 255) / 256 : 1;
    hipLaunchKernelGGL(k_measure_begin, dim3(gx > 0 ? gx : 1, pv.B), dim3(256), 0, s, pv, init_xy, do_init);
}

void launch_gain(const PoolView& pv, const CmdSrc& src, hipStream_t s) {
    // rows [0, N] only: K and G beyond the active dimension (pv.N may be a discovered prefix) are never read
    const int cover = pv.N + 1 < pv.ld ? pv.N + 1 : pv.ld;
    hipLaunchKernelGGL(k_gain, dim3((cover + 255) / 256, pv.B), dim3(256), 0, s, pv, src);
}

static int device_cus() {   // CUs of the current device (grid of the resident kernels)
    constexpr int kMaxDev = 64;
    static std::once_flag once[kMaxDev];
    static int cus[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 0;
    std::call_once(once[dev], [dev]() {
        int v = 0;
        cus[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess ? v : 0;
    });
    return cus[dev];
}

template <int U, int TPB>
static void launch_rank2_ut(const PoolView& pv, int rows, bool nt, hipStream_t s, bool resident) {
    const int ld2a = (pv.N + 1) / 2;
    dim3 grid((ld2a + TPB - 1) / TPB, (pv.N + rows - 1) / rows, pv.B);
    // resident workgroups on one tile queue (rank2_resident: big pools only), the queue word zeroed in front
    const long long total = (long long)grid.x * grid.y * grid.z;
    const int cus = device_cus();
    if (resident && hipMemsetAsync(pv.queue, 0, sizeof(unsigned), s) == hipSuccess) {
        if (nt)
            hipLaunchKernelGGL((k_rank2_queue<U, true, TPB>), dim3(cus), dim3(TPB), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                               pv.N, pv.ld, pv.sigma_stride, rows, (int)grid.x, (int)grid.y, (unsigned)total, pv.queue);
        else
            hipLaunchKernelGGL((k_rank2_queue<U, false, TPB>), dim3(cus), dim3(TPB), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                               pv.N, pv.ld, pv.sigma_stride, rows, (int)grid.x, (int)grid.y, (unsigned)total, pv.queue);
        return;
    }
    if (nt)
        hipLaunchKernelGGL((k_rank2<U, true, TPB>), grid, dim3(TPB), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                           pv.N, pv.ld, pv.sigma_stride, rows);
    else
        hipLaunchKernelGGL((k_rank2<U, false, TPB>), grid, dim3(TPB), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                           pv.N, pv.ld, pv.sigma_stride, rows);
}

template <int U>
static void launch_rank2_u(const PoolView& pv, int rows, bool nt, hipStream_t s, bool resident) {
    const int ld2a = (pv.N + 1) / 2;  // double2 columns of an active row
    if (ld2a <= 64) launch_rank2_ut<U, 64>(pv, rows, nt, s, resident);
    else if (ld2a <= 128) launch_rank2_ut<U, 128>(pv, rows, nt, s, resident);
    else if (ld2a <= 192) launch_rank2_ut<U, 192>(pv, rows, nt, s, resident);
    else launch_rank2_ut<U, 256>(pv, rows, nt, s, resident);
}

// which instantiation of k_rank2 (and how many rows per workgroup) a launch over this view takes
void rank2_variant(const PoolView& pv, const Rank2Tuning& t, int* u_out, int* nt_out, int* tpb_out, int* rows_out) {
    // Non-temporal only when the pool cannot stay resident in the 256 MiB Infinity Cache between
    // two corrections; a single filter's covariance (32 MB at n = 1000) should stay cached.
    // (the bytes this launch touches: pv.N may be a discovered prefix of a much larger pool)
    const size_t pool_bytes = (size_t)pv.B * pv.N * ((size_t)pv.N * sizeof(double));
    const bool nt = t.nontemporal >= 0 ? t.nontemporal != 0 : pool_bytes > ((size_t)192 << 20);
    int u = t.group_rows;
    int rows = t.rows_per_block;
    if (rows <= 0) {
        // Measured on MI355X (profiles/): a big pool streams fastest with 32 rows per workgroup taken as
        // two 16-row groups (32 x 16 B in flight per lane, 2 waves/SIMD): 6.17 TB/s algorithmic.  A small
        // pool needs enough workgroups to cover 256 CUs a few times over.
        const long long strips = (long long)pv.B * (((pv.N + 1) / 2 + 255) / 256);
        const long long total = strips * pv.N;
        rows = total >= 256LL * 8 * 32 ? 32 : (total >= 256LL * 4 * 8 ? 8 : 4);
    }
    if (rows > 1024) rows = 1024;
    if (u != 2 && u != 4 && u != 8 && u != 16) u = rows >= 32 ? 16 : (rows >= 8 ? 8 : (rows >= 4 ? 4 : 2));
    const int ld2a = (pv.N + 1) / 2;
    const int tpb = ld2a <= 64 ? 64 : (ld2a <= 128 ? 128 : (ld2a <= 192 ? 192 : 256));
    if (u_out) *u_out = u;
    if (nt_out) *nt_out = nt ? 1 : 0;
    if (tpb_out) *tpb_out = tpb;
    if (rows_out) *rows_out = rows;
}

// P > 1: the row-packed kernel serves this (full-width) view -- the smallest P <= 32 whose virtual rows fill the
// 256-lane strips to >= 97 %; 1: the plain kernel (rows already fill their strips, rows shorter than a wavefront, or a
// small pool that needs more workgroups than packing leaves)
int rank2_packing(const PoolView& pv, const Rank2Tuning& t) {
    const int ld2n = pv.ld / 2;
    if (!t.row_packing) return 1;                             // EKF_FORM_ROW_PACKING off: the plain kernel
    if (ld2n < 64) return 1;                                  // a wavefront may straddle at most two sub-rows
    if (pick_ld(pv.N) != pv.ld) return 1;                     // a prefix view: columns beyond it must stay untouched
    auto util = [&](int P) { const long long w = (long long)P * ld2n; return (double)w / (double)((w + 255) / 256 * 256); };
    if (util(1) >= 0.93) return 1;
    if ((long long)pv.B * pv.N * ld2n < 256LL * 256 * 64) return 1;   // too little work for fat workgroups
    for (int P = 2; P <= 32; P++)
        if (util(P) >= 0.97) return P;
    return 1;
}

// Whether a launch over this view runs as resident workgroups on one tile queue (k_rank2_queue): the 256-lane, 16-row-group
// instantiation on pools with at least 16 tiles per CU
bool rank2_resident(const PoolView& pv, const Rank2Tuning& t) {
    int u, nti, tpb, rows;
    rank2_variant(pv, t, &u, &nti, &tpb, &rows);
    if (!t.tile_queue || !pv.queue || u != 16 || tpb != 256) return false;
    const int cus = device_cus();
    const long long total = (long long)(((pv.N + 1) / 2 + tpb - 1) / tpb) * ((pv.N + rows - 1) / rows) * pv.B;
    return cus > 0 && total >= 16LL * cus && total < 0x7fffffffLL;
}

void launch_rank2(const PoolView& pv, const Rank2Tuning& t, hipStream_t s, bool full_width) {
    int u, nti, tpb, rows;
    rank2_variant(pv, t, &u, &nti, &tpb, &rows);
    const bool nt = nti != 0;
    const int P = full_width ? rank2_packing(pv, t) : 1;
    if (P > 1) {
        const int ld2n = pv.ld / 2;
        // measured (tools/rank2_width_sweep.py, profiles/r02/rank2_width_sweep.txt): ONE 8-row group per workgroup
        // (n = 200, B = 16384: 6.26 TB/s = 0.78 of peak; 16 rows 0.73, 32 rows 0.64; the plain kernel 0.67)
        const int vrows = t.rows_per_block > 0 ? t.rows_per_block : 8;
        dim3 grid((P * ld2n + 255) / 256, ((pv.N + P - 1) / P + vrows - 1) / vrows, pv.B);
        if (t.group_rows == 16) {
            if (nt) hipLaunchKernelGGL((k_rank2_packed<16, true>), grid, dim3(256), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                                       pv.N, pv.ld, pv.sigma_stride, P, vrows);
            else hipLaunchKernelGGL((k_rank2_packed<16, false>), grid, dim3(256), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                                    pv.N, pv.ld, pv.sigma_stride, P, vrows);
        } else {
            if (nt) hipLaunchKernelGGL((k_rank2_packed<8, true>), grid, dim3(256), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                                       pv.N, pv.ld, pv.sigma_stride, P, vrows);
            else hipLaunchKernelGGL((k_rank2_packed<8, false>), grid, dim3(256), 0, s, pv.sigma, pv.Kg, pv.Gh, pv.rec, pv.state,
                                    pv.N, pv.ld, pv.sigma_stride, P, vrows);
        }
        return;
    }
    const bool resident = rank2_resident(pv, t);
    switch (u) {
        case 2: launch_rank2_u<2>(pv, rows, nt, s, resident); break;
        case 4: launch_rank2_u<4>(pv, rows, nt, s, resident); break;
        case 16: launch_rank2_u<16>(pv, rows, nt, s, resident); break;
        default: launch_rank2_u<8>(pv, rows, nt, s, resident); break;
    }
}

void launch_rank2_active(const PoolView& pv, const Rank2Tuning& t, int max_touched, hipStream_t s) {
    const size_t pool_bytes = (size_t)pv.B * pv.sigma_stride * sizeof(double);
    const bool nt = t.nontemporal >= 0 ? t.nontemporal != 0 : pool_bytes > ((size_t)192 << 20);
    const int rows = 8;
    const int nslots = 3 + 2 * (max_touched < pv.n ? max_touched : pv.n);
    dim3 grid(((pv.N + 1) / 2 + 255) / 256, (nslots + rows - 1) / rows, pv.B);
    if (nt) hipLaunchKernelGGL((k_rank2_active<true>), grid, dim3(256), 0, s, pv, rows);
    else hipLaunchKernelGGL((k_rank2_active<false>), grid, dim3(256), 0, s, pv, rows);
}

void launch_touch_all(const PoolView& pv, hipStream_t s) {
    hipLaunchKernelGGL(k_touch_all, dim3(pv.B), dim3(256), 0, s, pv);
}

void launch_maha(const PoolView& pv, const MeasSrc& ms, double* scores, int m_override, int m_bound,
                 hipStream_t s, const Pending* pend) {
    if (pv.n <= 0) return;
    int m = m_override >= 0 ? m_override : pv.n;  // landmarks that can be scored in this launch
    if (m_bound >= 0 && m_bound < m) m = m_bound;
    if (m <= 0) return;
    const Pending none{nullptr, nullptr, 0, 0, 0};
    hipLaunchKernelGGL(k_maha, dim3((m + 3) / 4, pv.B), dim3(256), 0, s, pv, ms, scores, m_override, pend ? *pend : none);
}

void launch_assoc_begin(const PoolView& pv, const int* known_count_dev, int known_count_imm, hipStream_t s) {
    hipLaunchKernelGGL(k_assoc_begin, dim3((pv.B + 255) / 256), dim3(256), 0, s, pv, known_count_dev, known_count_imm);
}

void launch_assoc_decide(const PoolView& pv, const MeasSrc& ms, const double* scores, int* assoc_out,
                         int out_stride, int j, unsigned long long* corr_counter, hipStream_t s) {
    hipLaunchKernelGGL(k_assoc_decide, dim3(pv.B), dim3(256), 0, s, pv, ms, scores, assoc_out, out_stride, j,
                       corr_counter);
}

__global__ void k_normalize_angles(const double* __restrict__ in, int count, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = normalize_angle(in[i]);
}

void launch_normalize_angles(const double* in, int count, double* out, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(k_normalize_angles, dim3((count + 255) / 256), dim3(256), 0, s, in, count, out);
}

void launch_checksum(const PoolView& pv, double* out, hipStream_t s) {
    const int gx = pv.N < 64 ? pv.N : 64;
    hipLaunchKernelGGL(k_checksum, dim3(gx, pv.B), dim3(256), 0, s, pv, out);
}

void launch_gather_poses(const PoolView& pv, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_poses, dim3((pv.B + 255) / 256), dim3(256), 0, s, pv, out);
}

}  // namespace ekf
