// ekf_sim.hip -- on-device Monte-Carlo input generator and consistency statistics
// (SURVEY.md section 8(f) row f4).  Reproduces, per filter, the value distributions of the reference's
// simulator nurtlesim/src/tube_world.cpp (noisy commanded twist :191-208, wheel slip :214-227, robot-frame
// landmark readings + Gaussian noise with a visibility radius :369-414) and the caller's odometry
// marshalling (Odometer::getCurrentTwist, nuslam/src/slam.cpp:173-176), writing the compact
// known-association log directly into HBM -- no 100-GB host logs for configs[4].
//
// Every random number is a pure function of (seed, global filter id, step, kind, k) through
// splitmix64 + Box-Muller, the same addressing as the host generator ekf_slam_ml_amd/synth.py: integer
// parts are bit-identical, transcendental parts (log/cos/sqrt) agree to rounding.
#include "ekf_kernels.hpp"
#include "ekf_sim.hpp"

namespace ekf {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    unsigned long long z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long sim_key(unsigned long long seed, unsigned long long fid,
                                                      unsigned long long step, unsigned long long kind,
                                                      unsigned long long k) {
    const unsigned long long a = splitmix64(seed ^ (fid * 0xD1B54A32D192ED03ull));
    const unsigned long long b = splitmix64(a + step * 0x8CB92BA72F3D8DD7ull);
    const unsigned long long c = splitmix64(b + kind * 0xABC98388FB8FAC03ull);
    return splitmix64(c + k);
}
__device__ __forceinline__ double uniform01(unsigned long long seed, unsigned long long fid, unsigned long long step,
                                            unsigned long long kind, unsigned long long k) {
    return (double)(sim_key(seed, fid, step, kind, k) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal01(unsigned long long seed, unsigned long long fid, unsigned long long step,
                                           unsigned long long kind, unsigned long long k) {
    double u1 = uniform01(seed, fid, step, kind, 2 * k);
    const double u2 = uniform01(seed, fid, step, kind, 2 * k + 1);
    u1 = fmax(u1, 1.1102230246251565e-16);  // 2^-53
    return sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2);
}
constexpr unsigned long long KIND_CMD = 1, KIND_SLIP = 2, KIND_SENSOR = 3, KIND_SHUFFLE = 5, KIND_SCAN = 6;

// DiffDrive::getBodyTwistForUpdate, rigid2d/src/diff_drive.cpp:38-47
__device__ __forceinline__ void body_twist(const SimParams& p, double left, double right, double& ang, double& lin) {
    const double D = p.wheel_base * 0.5, r = p.wheel_radius;
    ang = (r / (2.0 * D)) * (right - left);
    lin = (r / 2.0) * (right + left);
}

// One thread per filter: the true trajectory and the odometry twists of all T steps.
__global__ void k_sim_trajectory(SimParams p, int B, int T, double* __restrict__ twist, double* __restrict__ truth) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long fid = (unsigned long long)p.first_filter_id + b;
    double theta = 0.0, x = 0.0, y = 0.0;
    for (int t = 0; t < T; t++) {
        // callback_vel: noise only on non-zero commands (tube_world.cpp:191-208)
        const double xv = p.v_cmd + (fabs(p.v_cmd) >= 1e-4 ? p.vx_std * normal01(p.seed, fid, t, KIND_CMD, 0) : 0.0);
        const double av = p.w_cmd + (fabs(p.w_cmd) >= 1e-4 ? p.the_std * normal01(p.seed, fid, t, KIND_CMD, 1) : 0.0);
        // DiffDrive::calculateWheelVelocity, diff_drive.cpp:24-36
        const double D = p.wheel_base * 0.5, r = p.wheel_radius;
        const double wl = -(D / r) * av + (1.0 / r) * xv, wr = (D / r) * av + (1.0 / r) * xv;
        double dl = 0.0, dr = 0.0;
        for (int k = 0; k < p.ticks_per_step; k++) {  // publishJointState, tube_world.cpp:214-227
            const double sl = p.slip_min + (p.slip_max - p.slip_min) * uniform01(p.seed, fid, t, KIND_SLIP, 2 * k);
            const double sr = p.slip_min + (p.slip_max - p.slip_min) * uniform01(p.seed, fid, t, KIND_SLIP, 2 * k + 1);
            dl = (wl / 100.0) * sl;
            dr = (wr / 100.0) * sr;
            double dth, ddx;
            body_twist(p, dl, dr, dth, ddx);
            if (fabs(dth) < 1e-9) {
                x = x + ddx * cos(theta);
                y = y + ddx * sin(theta);
            } else {
                const double rad = ddx / dth;
                const double nx = x - rad * sin(theta) + rad * sin(theta + dth);
                const double ny = y + rad * cos(theta) - rad * cos(theta + dth);
                x = nx; y = ny;
            }
            theta = theta + dth;
        }
        double ta, tx;
        body_twist(p, dl * 10.0, dr * 10.0, ta, tx);  // Odometer::getCurrentTwist, slam.cpp:173-176
        twist[((size_t)t * B + b) * 2] = ta;
        twist[((size_t)t * B + b) * 2 + 1] = tx;
        truth[((size_t)t * B + b) * 3] = theta;
        truth[((size_t)t * B + b) * 3 + 1] = x;
        truth[((size_t)t * B + b) * 3 + 2] = y;
    }
}

// One workgroup per (filter, step): readings of the vmax nearest landmarks within the visibility radius,
// in ascending landmark order (publishFakeSensor, tube_world.cpp:369-414).  Step 0 instead fills init_xy
// with ALL n readings and leaves the slots empty (state_update_flag is still false, slam.cpp:315-327).
__global__ __launch_bounds__(256) void k_sim_readings(SimParams p, int B, int n, int vmax,
                                                      const double* __restrict__ world,
                                                      const double* __restrict__ truth, int* __restrict__ lm_idx,
                                                      double* __restrict__ z_xy, double* __restrict__ init_xy,
                                                      int* __restrict__ slot_active) {
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const unsigned long long fid = (unsigned long long)p.first_filter_id + b;
    const double th = truth[((size_t)t * B + b) * 3], px = truth[((size_t)t * B + b) * 3 + 1],
                 py = truth[((size_t)t * B + b) * 3 + 2];
    const double c = cos(th), s = sin(th);
    int* slots = lm_idx + ((size_t)t * B + b) * vmax;
    double* zz = z_xy + ((size_t)t * B + b) * vmax * 2;
    if (t == 0) {
        for (int i = tid; i < n; i += 256) {
            const double dx = world[2 * i] - px, dy = world[2 * i + 1] - py;
            init_xy[(size_t)b * 2 * n + 2 * i] = (c * dx + s * dy) + p.sensor_std * normal01(p.seed, fid, 0, KIND_SENSOR, 2ull * i);
            init_xy[(size_t)b * 2 * n + 2 * i + 1] = (-s * dx + c * dy) + p.sensor_std * normal01(p.seed, fid, 0, KIND_SENSOR, 2ull * i + 1);
        }
        for (int v = tid; v < vmax; v += 256) { slots[v] = -1; zz[2 * v] = 0.0; zz[2 * v + 1] = 0.0; }
        return;
    }
    __shared__ double sh_d[4];
    __shared__ int sh_i[4];
    __shared__ int chosen[64];
    __shared__ int n_chosen;
    __shared__ int exhausted;
    const int kmax = vmax < 64 ? vmax : 64;
    if (tid == 0) { n_chosen = 0; exhausted = 0; }
    __syncthreads();
    const double lim2 = p.max_visible_dis * p.max_visible_dis;
    for (int pass = 0; pass < kmax; pass++) {
        // nearest not-yet-chosen landmark (lexicographic (d2, i) minimum)
        double best = 1.0e300;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            bool taken = false;
            for (int q = 0; q < n_chosen; q++) taken |= (chosen[q] == i);
            if (taken) continue;
            const double dx = world[2 * i] - px, dy = world[2 * i + 1] - py;
            const double rx = c * dx + s * dy, ry = -s * dx + c * dy;
            const double d2 = rx * rx + ry * ry;
            if (d2 < best || (d2 == best && i < bi)) { best = d2; bi = i; }
        }
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
            if (bi != 0x7fffffff && best <= lim2) chosen[n_chosen++] = bi;
            else exhausted = 1;  // nothing left within the radius
        }
        __syncthreads();
        if (exhausted) break;  // uniform
    }
    if (tid == 0) {
        const int m = n_chosen;
        for (int a = 1; a < m; a++) {  // ascending landmark index = loop order of ekf_slam.cpp:132
            const int v = chosen[a];
            int q = a - 1;
            while (q >= 0 && chosen[q] > v) { chosen[q + 1] = chosen[q]; q--; }
            chosen[q + 1] = v;
        }
        for (int v = 0; v < vmax; v++) {
            if (v < m) {
                const int i = chosen[v];
                const double dx = world[2 * i] - px, dy = world[2 * i + 1] - py;
                slots[v] = i;
                zz[2 * v] = (c * dx + s * dy) + p.sensor_std * normal01(p.seed, fid, t, KIND_SENSOR, 2ull * i);
                zz[2 * v + 1] = (-s * dx + c * dy) + p.sensor_std * normal01(p.seed, fid, t, KIND_SENSOR, 2ull * i + 1);
                atomicAdd(&slot_active[(size_t)t * vmax + v], 1);
            } else {
                slots[v] = -1; zz[2 * v] = 0.0; zz[2 * v + 1] = 0.0;
            }
        }
    }
}

// The kmax nearest landmarks within the visibility radius of pose (c, s, px, py), by repeated lexicographic
// (d2, index) minimum over the workgroup: chosen[0..n_chosen) in ascending distance.  256 threads.
__device__ void select_nearest(int n, int kmax, double lim2, const double* __restrict__ world, double c, double s,
                               double px, double py, int* chosen, int* n_chosen, int* exhausted, double* sh_d,
                               int* sh_i) {
    const int tid = threadIdx.x;
    if (tid == 0) { *n_chosen = 0; *exhausted = 0; }
    __syncthreads();
    for (int pass = 0; pass < kmax; pass++) {
        double best = 1.0e300;
        int bi = 0x7fffffff;
        const int nch = *n_chosen;
        for (int i = tid; i < n; i += 256) {
            bool taken = false;
            for (int q = 0; q < nch; q++) taken |= (chosen[q] == i);
            if (taken) continue;
            const double dx = world[2 * i] - px, dy = world[2 * i + 1] - py;
            const double rx = c * dx + s * dy, ry = -s * dx + c * dy;
            const double d2 = rx * rx + ry * ry;
            if (d2 < best || (d2 == best && i < bi)) { best = d2; bi = i; }
        }
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
            if (bi != 0x7fffffff && best <= lim2) chosen[(*n_chosen)++] = bi;
            else *exhausted = 1;  // nothing left within the radius
        }
        __syncthreads();
        if (*exhausted) break;  // uniform
    }
}

// One workgroup per (filter, step): the scan_measures vector of nuslam/src/unknown_data_assoc.cpp:309-320 as the
// fake sensor would fill it -- noisy robot-frame (x, y) of the (at most jmax nearest) landmarks within the
// visibility radius, in a per-step shuffled order (ascending KIND_SHUFFLE key), EVERY step including step 0.
// Host twin: synth.make_unknown_log.
__global__ __launch_bounds__(256) void k_sim_unknown_readings(SimParams p, int B, int n, int jmax,
                                                              const double* __restrict__ world,
                                                              const double* __restrict__ truth,
                                                              int* __restrict__ count, double* __restrict__ meas) {
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const unsigned long long fid = (unsigned long long)p.first_filter_id + b;
    const double th = truth[((size_t)t * B + b) * 3], px = truth[((size_t)t * B + b) * 3 + 1],
                 py = truth[((size_t)t * B + b) * 3 + 2];
    const double c = cos(th), s = sin(th);
    __shared__ double sh_d[4];
    __shared__ int sh_i[4];
    __shared__ int chosen[64];
    __shared__ double key[64];
    __shared__ int n_chosen, exhausted;
    select_nearest(n, jmax < 64 ? jmax : 64, p.max_visible_dis * p.max_visible_dis, world, c, s, px, py, chosen,
                   &n_chosen, &exhausted, sh_d, sh_i);
    if (tid == 0) {
        const int m = n_chosen;
        for (int a = 0; a < m; a++) key[a] = uniform01(p.seed, fid, t, KIND_SHUFFLE, (unsigned long long)chosen[a]);
        for (int a = 1; a < m; a++) {  // ascending (key, landmark)
            const int v = chosen[a];
            const double kv = key[a];
            int q = a - 1;
            while (q >= 0 && (key[q] > kv || (key[q] == kv && chosen[q] > v))) {
                chosen[q + 1] = chosen[q]; key[q + 1] = key[q]; q--;
            }
            chosen[q + 1] = v; key[q + 1] = kv;
        }
        double* zz = meas + ((size_t)t * B + b) * jmax * 2;
        for (int v = 0; v < jmax; v++) {
            if (v < m) {
                const int i = chosen[v];
                const double dx = world[2 * i] - px, dy = world[2 * i + 1] - py;
                zz[2 * v] = (c * dx + s * dy) + p.sensor_std * normal01(p.seed, fid, t, KIND_SENSOR, 2ull * i);
                zz[2 * v + 1] = (-s * dx + c * dy) + p.sensor_std * normal01(p.seed, fid, t, KIND_SENSOR, 2ull * i + 1);
            } else {
                zz[2 * v] = 0.0; zz[2 * v + 1] = 0.0;
            }
        }
        count[(size_t)t * B + b] = m;
    }
}

// One workgroup per scan: n_beams ranges of a 2-D lidar at poses[s] in a square walled world with n tubes of
// one radius (publishScan, nurtlesim/src/tube_world.cpp:451-577), plus N(0, range_std) per beam (:571).
//   lp.model 0  clean ray geometry: nearest of the wall hit, range_max and the first intersection with every tube
//   lp.model 1  publishScan's own procedure, step for step (scan_reference_beam below)
// Host twin: synth.make_scans.  Candidate tubes (centre within range_max + radius) are compacted into LDS first.
constexpr int kCandMax = 512;

// what publishScan does for one tube and one beam (:518-565): the tube is looked at only inside a bearing window around it
// (:526-552), and the hit is the nearer intersection of the LINE through the robot and the beam's end point with the tube's
// circle, worked out in the tube's frame (getLineCircleIntersection, :420-450).  (tx, ty): the tube in the robot frame;
// (start, end): its window; curr: the beam's angle; (x2, y2): the beam's end point at range_max.
__device__ __forceinline__ double scan_reference_beam(double tx, double ty, double start, double end, double curr, double x2,
                                                      double y2, double radius, double min_r) {
    bool flag;
    if (start > 0 && end < 0) flag = curr > start || curr < end;
    else flag = curr > start && curr < end;
    if (!flag) return min_r;
    const double x1 = -tx, y1 = -ty;
    const double xb = x2 - tx, yb = y2 - ty;
    const double dx = xb - x1, dy = yb - y1;
    const double dr = sqrt(dx * dx + dy * dy);
    const double D = x1 * yb - xb * y1;
    const double delta = radius * radius * (dr * dr) - D * D;
    if (!(delta > 0)) return min_r;
    const double sgn = dy < 0 ? -1.0 : 1.0, sq = sqrt(delta), dr2 = dr * dr;
    const double ix1 = (D * dy + sgn * dx * sq) / dr2, iy1 = (-D * dx + fabs(dy) * sq) / dr2;
    const double ix2 = (D * dy - sgn * dx * sq) / dr2, iy2 = (-D * dx - fabs(dy) * sq) / dr2;
    const double d1 = sqrt((x1 - ix1) * (x1 - ix1) + (y1 - iy1) * (y1 - iy1));
    const double d2 = sqrt((x1 - ix2) * (x1 - ix2) + (y1 - iy2) * (y1 - iy2));
    return fmin(fmin(d1, d2), min_r);
}
__global__ __launch_bounds__(256) void k_sim_scans(SimParams p, LidarParams lp, int B, int n, int t0,
                                                   const double* __restrict__ world,
                                                   const double* __restrict__ poses, double* __restrict__ ranges) {
    const int sidx = blockIdx.x, tid = threadIdx.x;
    const unsigned long long fid = (unsigned long long)p.first_filter_id + (unsigned long long)(sidx % B);
    const unsigned long long step = (unsigned long long)(t0 + sidx / B);
    const double th0 = poses[(size_t)sidx * 3], ox = poses[(size_t)sidx * 3 + 1], oy = poses[(size_t)sidx * 3 + 2];
    __shared__ double cand[2 * kCandMax];
    __shared__ int n_cand;
    if (tid == 0) n_cand = 0;
    __syncthreads();
    const double reach = lp.range_max + lp.tube_radius;
    for (int i = tid; i < n; i += 256) {
        const double fx = ox - world[2 * i], fy = oy - world[2 * i + 1];
        if (fx * fx + fy * fy <= reach * reach) {
            const int k = atomicAdd(&n_cand, 1);
            if (k < kCandMax) { cand[2 * k] = fx; cand[2 * k + 1] = fy; }
        }
    }
    __syncthreads();
    const int nc = n_cand;
    const double half = lp.border_width / 2.0, r2 = lp.tube_radius * lp.tube_radius;
    if (lp.model == 1) {   // (uniform) publishScan's own procedure
        __shared__ double cref[4 * kCandMax];   // per candidate: the tube in the robot frame and its bearing window
        const double c0 = cos(th0), s0 = sin(th0);
        const double window = 2.0 * atan2(lp.tube_radius, lp.range_min);   // :480
        auto tube_terms = [&](double ex, double ey, double& tx, double& ty, double& st, double& en) {
            tx = c0 * ex + s0 * ey; ty = -s0 * ex + c0 * ey;               // Ttw(world_tube), :520-521
            const double tb = atan2(ty, tx);                                // :524
            st = normalize_angle(tb - window / 2.0); en = normalize_angle(tb + window / 2.0);
        };
        if (nc <= kCandMax)
            for (int k = tid; k < nc; k += 256)
                tube_terms(-cand[2 * k], -cand[2 * k + 1], cref[4 * k], cref[4 * k + 1], cref[4 * k + 2], cref[4 * k + 3]);
        __syncthreads();
        const double res = 2.0 * 3.141592653589793 / (double)lp.n_beams;
        for (int i = tid; i < lp.n_beams; i += 256) {
            const double curr = normalize_angle(res * (double)i);           // :496
            const double x2 = lp.range_max * cos(curr), y2 = lp.range_max * sin(curr);
            const double x_dis = half - ox, y_dis = half - oy;
            const double box = normalize_angle(res * (double)i + th0);      // :502
            const double y_t = box < 0 ? -(lp.border_width - y_dis) : y_dis;
            const double x_t = (box > 3.141592653589793 / 2.0 || box < -3.141592653589793 / 2.0) ? -(lp.border_width - x_dis) : x_dis;
            double min_r = fmin(fmin(x_t / cos(box), y_t / sin(box)), lp.range_max);   // :512-516
            if (nc <= kCandMax) {
                for (int k = 0; k < nc; k++)
                    min_r = scan_reference_beam(cref[4 * k], cref[4 * k + 1], cref[4 * k + 2], cref[4 * k + 3], curr, x2, y2,
                                                lp.tube_radius, min_r);
            } else {
                for (int k = 0; k < n; k++) {
                    double tx, ty, st, en;
                    tube_terms(world[2 * k] - ox, world[2 * k + 1] - oy, tx, ty, st, en);
                    min_r = scan_reference_beam(tx, ty, st, en, curr, x2, y2, lp.tube_radius, min_r);
                }
            }
            ranges[(size_t)sidx * lp.n_beams + i] = min_r + lp.range_std * normal01(p.seed, fid, step, KIND_SCAN, (unsigned long long)i);
        }
        return;
    }
    for (int i = tid; i < lp.n_beams; i += 256) {
        const double ang = 2.0 * 3.141592653589793 * (double)i / (double)lp.n_beams;
        const double th = th0 + ang;
        const double dx = cos(th), dy = sin(th);
        const double inf = __builtin_huge_val();
        const double tx = dx > 0 ? (half - ox) / dx : (dx < 0 ? (-half - ox) / dx : inf);
        const double ty = dy > 0 ? (half - oy) / dy : (dy < 0 ? (-half - oy) / dy : inf);
        double r = fmin(fmin(tx, ty), lp.range_max);
        if (nc <= kCandMax) {
            for (int k = 0; k < nc; k++) {
                const double fx = cand[2 * k], fy = cand[2 * k + 1];
                const double bq = fx * dx + fy * dy;
                const double cq = fx * fx + fy * fy - r2;
                const double disc = bq * bq - cq;
                if (disc > 0) {
                    const double tt = -bq - sqrt(disc);
                    if (tt > 0 && tt < r) r = tt;
                }
            }
        } else {  // more tubes in reach than the LDS list holds: walk the whole map
            for (int k = 0; k < n; k++) {
                const double fx = ox - world[2 * k], fy = oy - world[2 * k + 1];
                const double bq = fx * dx + fy * dy;
                const double cq = fx * fx + fy * fy - r2;
                const double disc = bq * bq - cq;
                if (disc > 0) {
                    const double tt = -bq - sqrt(disc);
                    if (tt > 0 && tt < r) r = tt;
                }
            }
        }
        ranges[(size_t)sidx * lp.n_beams + i] = r + lp.range_std * normal01(p.seed, fid, step, KIND_SCAN, (unsigned long long)i);
    }
}

// Monte-Carlo consistency of the batch against the simulated truth of step t: per filter the pose error
// e = (wrap(theta - theta*), x - x*, y - y*), NEES = e^T P^-1 e with P = Sigma[0:3,0:3].
// out[b][4] = {NEES, ex^2 + ey^2, etheta^2, trace P}
__global__ void k_mc_stats(PoolView pv, const double* __restrict__ truth_t, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= pv.B) return;
    const double* st = pv.state + (size_t)b * pv.ld;
    const double* S = pv.sigma + (size_t)b * pv.sigma_stride;
    const double e0 = normalize_angle(st[0] - truth_t[(size_t)b * 3]);
    const double e1 = st[1] - truth_t[(size_t)b * 3 + 1], e2 = st[2] - truth_t[(size_t)b * 3 + 2];
    const double a = S[0], bq = S[1], c = S[2], d = S[pv.ld], e = S[pv.ld + 1], f = S[pv.ld + 2], g = S[2 * pv.ld],
                 h = S[2 * pv.ld + 1], i = S[2 * pv.ld + 2];
    const double A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
    const double det = a * A + bq * Bc + c * Cc;
    // x = P^-1 e by the adjugate
    const double x0 = (A * e0 + (c * h - bq * i) * e1 + (bq * f - c * e) * e2) / det;
    const double x1 = (Bc * e0 + (a * i - c * g) * e1 + (c * d - a * f) * e2) / det;
    const double x2 = (Cc * e0 + (bq * g - a * h) * e1 + (a * e - bq * d) * e2) / det;
    out[(size_t)b * 4] = e0 * x0 + e1 * x1 + e2 * x2;
    out[(size_t)b * 4 + 1] = e1 * e1 + e2 * e2;
    out[(size_t)b * 4 + 2] = e0 * e0;
    out[(size_t)b * 4 + 3] = a + e + i;
}

void launch_sim(const SimParams& p, int B, int n, int T, int vmax, const double* world, double* twist, double* truth,
                int* lm_idx, double* z_xy, double* init_xy, int* slot_active, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_trajectory, dim3((B + 127) / 128), dim3(128), 0, s, p, B, T, twist, truth);
    hipLaunchKernelGGL(k_sim_readings, dim3(B, T), dim3(256), 0, s, p, B, n, vmax, world, truth, lm_idx, z_xy, init_xy,
                       slot_active);
}

void launch_sim_unknown(const SimParams& p, int B, int n, int T, int jmax, const double* world, double* twist,
                        double* truth, int* count, double* meas, bool trajectory, hipStream_t s) {
    if (trajectory) hipLaunchKernelGGL(k_sim_trajectory, dim3((B + 127) / 128), dim3(128), 0, s, p, B, T, twist, truth);
    if (count)
        hipLaunchKernelGGL(k_sim_unknown_readings, dim3(B, T), dim3(256), 0, s, p, B, n, jmax, world, truth, count, meas);
}

void launch_sim_scans(const SimParams& p, const LidarParams& lp, int B, int n, int S, int t0, const double* world,
                      const double* poses, double* ranges, hipStream_t s) {
    if (S <= 0) return;
    hipLaunchKernelGGL(k_sim_scans, dim3(S), dim3(256), 0, s, p, lp, B, n, t0, world, poses, ranges);
}

void launch_mc_stats(const PoolView& pv, const double* truth_t, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_mc_stats, dim3((pv.B + 127) / 128), dim3(128), 0, s, pv, truth_t, out);
}

}  // namespace ekf
