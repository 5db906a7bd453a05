// ekf_capi_sim.hip -- C ABI of include/ekfslam.h, on-device inputs and statistics: the simulator's logs and laser
// scans, the batched circle fitting, Monte-Carlo consistency.
#include "ekf_runtime.hpp"

using namespace ekfrt;

extern "C" {

void ekf_default_sim_params(ekf_sim_params* out) {
    if (!out) return;
    out->seed = 5000000ull;
    out->first_filter_id = 0;
    out->v_cmd = 0.5; out->w_cmd = 0.06;
    out->vx_std = 0.01; out->the_std = 0.01;      // noise_param.yaml:3,5
    out->slip_min = 0.90; out->slip_max = 1.10;   // noise_param.yaml:6-7
    out->sensor_std = 0.005;                      // noise_param.yaml:8-9
    out->max_visible_dis = 0.7;                   // noise_param.yaml:10
    out->wheel_base = 0.16; out->wheel_radius = 0.033;  // fake_turtle_param.yaml:6-7
    out->ticks_per_step = 10;
}

ekf_status ekf_batch_simulate_known_log(ekf_batch_handle hb, const ekf_sim_params* sp, const double* world_xy, int T,
                                        int vmax) {
    if (!hb || !sp || !world_xy || T <= 0 || vmax < 0 || vmax > 64 || sp->ticks_per_step < 1)
        return fail(EKF_ERR_INVALID, "ekf_batch_simulate_known_log: bad argument (vmax <= 64)");
    Pool& P = hb->pool;
    EKFC(P.use());
    EKFC(free_log(P));
    const int B = P.pv.B, n = P.pv.n;
    const size_t n_tw = (size_t)T * B * 2, n_lm = (size_t)T * B * vmax, n_z = n_lm * 2, n_in = (size_t)B * 2 * n,
                 n_tr = (size_t)T * B * 3;
    HIPC(hipMalloc((void**)&P.log_twist, sizeof(double) * n_tw));
    HIPC(hipMalloc((void**)&P.log_lm, sizeof(int) * (n_lm ? n_lm : 1)));
    HIPC(hipMalloc((void**)&P.log_z, sizeof(double) * (n_z ? n_z : 1)));
    HIPC(hipMalloc((void**)&P.log_init, sizeof(double) * (n_in ? n_in : 1)));
    HIPC(hipMalloc((void**)&P.log_truth, sizeof(double) * n_tr));
    P.log_bytes = sizeof(double) * (n_tw + n_z + n_in + n_tr) + sizeof(int) * n_lm;
    double* d_world = nullptr;
    int* d_active = nullptr;
    const size_t n_act = (size_t)T * (vmax > 0 ? vmax : 1);
    HIPC(hipMalloc((void**)&d_world, sizeof(double) * 2 * (n > 0 ? n : 1)));
    HIPC(hipMalloc((void**)&d_active, sizeof(int) * n_act));
    ekf_status st = EKF_OK;
    auto body = [&]() -> ekf_status {
        if (n > 0) HIPC(hipMemcpyAsync(d_world, world_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, P.stream));
        HIPC(hipMemsetAsync(d_active, 0, sizeof(int) * n_act, P.stream));
        ekf::SimParams p{sp->seed, sp->first_filter_id, sp->v_cmd, sp->w_cmd, sp->vx_std, sp->the_std, sp->slip_min,
                         sp->slip_max, sp->sensor_std, sp->max_visible_dis, sp->wheel_base, sp->wheel_radius,
                         sp->ticks_per_step};
        ekf::launch_sim(p, B, n, T, vmax, d_world, P.log_twist, P.log_truth, P.log_lm, P.log_z, P.log_init, d_active,
                        P.stream);
        HIPC(hipGetLastError());
        std::vector<int> active(n_act, 0);
        HIPC(hipMemcpyAsync(active.data(), d_active, sizeof(int) * n_act, hipMemcpyDeviceToHost, P.stream));
        HIPC(hipStreamSynchronize(P.stream));
        P.slot_active.swap(active);
        std::vector<int> lm_host(n_lm ? n_lm : 1, -1);
        if (n_lm) HIPC(hipMemcpy(lm_host.data(), P.log_lm, sizeof(int) * n_lm, hipMemcpyDeviceToHost));
        P.compute_log_touch_bound(lm_host.data(), T, vmax);
        return EKF_OK;
    };
    st = body();
    (void)hipFree(d_world);
    (void)hipFree(d_active);
    if (st != EKF_OK) return st;
    P.T = T;
    P.truth_is_unknown_log = 0;
    P.vmax = vmax;
    return EKF_OK;
}

ekf_status ekf_batch_download_log(ekf_batch_handle hb, double* twist, int* lm_idx, double* z_xy, double* init_xy,
                                  double* true_pose) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.T <= 0) return fail(EKF_ERR_STATE, "no log on the device");
    if (true_pose && !P.log_truth) return fail(EKF_ERR_STATE, "the uploaded log carries no simulated truth");
    EKFC(P.use());
    HIPC(hipStreamSynchronize(P.stream));
    const size_t B = P.pv.B, T = P.T, vmax = P.vmax, n = P.pv.n;
    if (twist) HIPC(hipMemcpy(twist, P.log_twist, sizeof(double) * T * B * 2, hipMemcpyDeviceToHost));
    if (lm_idx && vmax) HIPC(hipMemcpy(lm_idx, P.log_lm, sizeof(int) * T * B * vmax, hipMemcpyDeviceToHost));
    if (z_xy && vmax) HIPC(hipMemcpy(z_xy, P.log_z, sizeof(double) * T * B * vmax * 2, hipMemcpyDeviceToHost));
    if (init_xy && n) HIPC(hipMemcpy(init_xy, P.log_init, sizeof(double) * B * 2 * n, hipMemcpyDeviceToHost));
    if (true_pose) HIPC(hipMemcpy(true_pose, P.log_truth, sizeof(double) * T * B * 3, hipMemcpyDeviceToHost));
    return EKF_OK;
}

ekf_status ekf_batch_mc_stats(ekf_batch_handle hb, int t, double out[6]) {
    if (!hb || !out) return fail(EKF_ERR_INVALID, "null argument");
    Pool& P = hb->pool;
    const double* truth = P.truth_is_unknown_log ? P.ulog_truth : P.log_truth;
    const int Tl = P.truth_is_unknown_log ? P.uT : P.T;
    if (!truth) return fail(EKF_ERR_STATE, "ekf_batch_mc_stats needs a simulated log (ground truth)");
    if (t < 0 || t >= Tl) return fail(EKF_ERR_INVALID, "step outside the log");
    EKFC(P.use());
    EKFC(P.flush());
    ekf::launch_mc_stats(P.pv, truth + (size_t)t * P.pv.B * 3, P.digest_dev, P.stream);
    EKFC(checked_launch());
    std::vector<double> h((size_t)4 * P.pv.B);
    EKFC(P.download(h.data(), P.digest_dev, sizeof(double) * h.size()));
    double nees = 0, nmax = 0, p2 = 0, a2 = 0, tr = 0, inside = 0;
    for (int b = 0; b < P.pv.B; b++) {
        const double v = h[(size_t)b * 4];
        nees += v; if (v > nmax) nmax = v;
        p2 += h[(size_t)b * 4 + 1]; a2 += h[(size_t)b * 4 + 2]; tr += h[(size_t)b * 4 + 3];
        if (v < 7.815) inside += 1.0;
    }
    const double Bn = (double)P.pv.B;
    out[0] = nees / Bn; out[1] = nmax; out[2] = std::sqrt(p2 / Bn); out[3] = std::sqrt(a2 / Bn); out[4] = tr / Bn;
    out[5] = inside / Bn;
    return EKF_OK;
}

void ekf_default_lidar_params(ekf_lidar_params* out) {
    if (!out) return;
    out->n_beams = 360;          // tube_world.cpp:452
    out->range_std = 0.005;      // noise_param.yaml
    out->range_max = 3.5;        // tube_world.cpp:476
    out->border_width = 2.0;     // tube_param.yaml
    out->tube_radius = 0.0762;   // tube_param.yaml
    out->model = 0;              // clean ray geometry (1: publishScan's own procedure)
    out->range_min = 0.12;       // tube_world.cpp:475
}

static ekf::SimParams to_sim(const ekf_sim_params* sp) {
    return ekf::SimParams{sp->seed, sp->first_filter_id, sp->v_cmd, sp->w_cmd, sp->vx_std, sp->the_std, sp->slip_min,
                          sp->slip_max, sp->sensor_std, sp->max_visible_dis, sp->wheel_base, sp->wheel_radius,
                          sp->ticks_per_step};
}

static bool lidar_ok(const ekf_lidar_params* lp) {
    return lp->n_beams >= 8 && lp->n_beams <= ekf::circles_max_beams() && lp->range_max > 0 && lp->border_width > 0 &&
           lp->tube_radius > 0 && lp->range_std >= 0 && (lp->model == 0 || (lp->model == 1 && lp->range_min > 0));
}

ekf_status ekf_batch_simulate_unknown_log(ekf_batch_handle hb, const ekf_sim_params* sp, const ekf_lidar_params* lidar,
                                          const double* world_xy, int T, int jmax) {
    if (!hb || !sp || !world_xy || T <= 0 || jmax < 1 || jmax > 64 || sp->ticks_per_step < 1 || (lidar && !lidar_ok(lidar)))
        return fail(EKF_ERR_INVALID, "ekf_batch_simulate_unknown_log: bad argument (1 <= jmax <= 64, 8 <= n_beams <= 1024)");
    Pool& P = hb->pool;
    EKFC(P.use());
    EKFC(free_ulog(P));
    const int B = P.pv.B, n = P.pv.n;
    const size_t n_tw = (size_t)T * B * 2, n_ct = (size_t)T * B, n_me = n_ct * jmax * 2, n_as = n_ct * jmax, n_tr = n_ct * 3;
    HIPC(hipMalloc((void**)&P.ulog_twist, sizeof(double) * n_tw));
    HIPC(hipMalloc((void**)&P.ulog_count, sizeof(int) * n_ct));
    HIPC(hipMalloc((void**)&P.ulog_meas, sizeof(double) * n_me));
    HIPC(hipMalloc((void**)&P.ulog_assoc, sizeof(int) * n_as));
    HIPC(hipMalloc((void**)&P.ulog_truth, sizeof(double) * n_tr));
    if (!P.corr_counter) HIPC(hipMalloc((void**)&P.corr_counter, sizeof(unsigned long long)));
    P.ulog_bytes = sizeof(double) * (n_tw + n_me + n_tr) + sizeof(int) * (n_ct + n_as);
    double *d_world = nullptr, *d_ranges = nullptr, *d_radii = nullptr;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_world, sizeof(double) * 2 * (n > 0 ? n : 1)));
        if (n > 0) HIPC(hipMemcpyAsync(d_world, world_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice, P.stream));
        HIPC(hipMemsetAsync(P.ulog_meas, 0, sizeof(double) * n_me, P.stream));
        HIPC(hipMemsetAsync(P.ulog_assoc, 0xFF, sizeof(int) * n_as, P.stream));  // -1; run_unknown overwrites
        const ekf::SimParams p = to_sim(sp);
        if (!lidar) {
            ekf::launch_sim_unknown(p, B, n, T, jmax, d_world, P.ulog_twist, P.ulog_truth, P.ulog_count, P.ulog_meas,
                                    true, P.stream);
        } else {
            ekf::launch_sim_unknown(p, B, n, T, jmax, d_world, P.ulog_twist, P.ulog_truth, nullptr, nullptr, true, P.stream);
            const ekf::LidarParams lp{lidar->n_beams, lidar->range_std, lidar->range_max, lidar->border_width,
                                      lidar->tube_radius, lidar->model, lidar->range_min};
            // scans are produced and consumed in chunks of whole steps (<= 256 MiB of ranges at a time)
            size_t steps_per_chunk = ((size_t)256 << 20) / (sizeof(double) * lp.n_beams * B);
            if (steps_per_chunk < 1) steps_per_chunk = 1;
            if (steps_per_chunk > (size_t)T) steps_per_chunk = T;
            HIPC(hipMalloc((void**)&d_ranges, sizeof(double) * steps_per_chunk * B * lp.n_beams));
            HIPC(hipMalloc((void**)&d_radii, sizeof(double) * steps_per_chunk * B * jmax));
            for (int t0 = 0; t0 < T; t0 += (int)steps_per_chunk) {
                const int tc = T - t0 < (int)steps_per_chunk ? T - t0 : (int)steps_per_chunk;
                const int S = tc * B;
                ekf::launch_sim_scans(p, lp, B, n, S, t0, d_world, P.ulog_truth + (size_t)t0 * B * 3, d_ranges, P.stream);
                ekf::launch_circles(d_ranges, S, lp.n_beams, jmax, P.ulog_meas + (size_t)t0 * B * jmax * 2, d_radii,
                                    P.ulog_count + (size_t)t0 * B, nullptr, nullptr, P.stream);
            }
        }
        HIPC(hipGetLastError());
        P.ucount_host.assign(n_ct, 0);
        HIPC(hipMemcpyAsync(P.ucount_host.data(), P.ulog_count, sizeof(int) * n_ct, hipMemcpyDeviceToHost, P.stream));
        HIPC(hipStreamSynchronize(P.stream));
        return EKF_OK;
    };
    const ekf_status st = body();
    for (void* q : {(void*)d_world, (void*)d_ranges, (void*)d_radii})
        if (q) (void)hipFree(q);
    if (st != EKF_OK) return st;
    // decisions start as "not run" (-2)
    {
        std::vector<int> fill(n_as, -2);
        HIPC(hipMemcpy(P.ulog_assoc, fill.data(), sizeof(int) * n_as, hipMemcpyHostToDevice));
    }
    P.uT = T;
    P.ujmax = jmax;
    P.truth_is_unknown_log = 1;
    return EKF_OK;
}

ekf_status ekf_batch_download_unknown_log(ekf_batch_handle hb, double* twist, int* count, double* meas_xy,
                                          double* true_pose) {
    if (!hb) return fail(EKF_ERR_INVALID, "null handle");
    Pool& P = hb->pool;
    if (P.uT <= 0) return fail(EKF_ERR_STATE, "no unknown-association log on the device");
    if (true_pose && !P.ulog_truth) return fail(EKF_ERR_STATE, "the uploaded log carries no simulated truth");
    EKFC(P.use());
    HIPC(hipStreamSynchronize(P.stream));
    const size_t B = P.pv.B, T = P.uT, J = P.ujmax;
    if (twist) HIPC(hipMemcpy(twist, P.ulog_twist, sizeof(double) * T * B * 2, hipMemcpyDeviceToHost));
    if (count) HIPC(hipMemcpy(count, P.ulog_count, sizeof(int) * T * B, hipMemcpyDeviceToHost));
    if (meas_xy && J) HIPC(hipMemcpy(meas_xy, P.ulog_meas, sizeof(double) * T * B * J * 2, hipMemcpyDeviceToHost));
    if (true_pose) HIPC(hipMemcpy(true_pose, P.ulog_truth, sizeof(double) * T * B * 3, hipMemcpyDeviceToHost));
    return EKF_OK;
}

ekf_status ekf_simulate_scans(int device, const ekf_sim_params* sp, const ekf_lidar_params* lidar, const double* world_xy,
                              int n, const double* poses, int S, int step, double* ranges_out) {
    if (!sp || !lidar || !lidar_ok(lidar) || n < 0 || (n > 0 && !world_xy) || S < 0 || (S > 0 && (!poses || !ranges_out)) ||
        step < 0)
        return fail(EKF_ERR_INVALID, "ekf_simulate_scans: bad argument (8 <= n_beams <= 1024)");
    if (S == 0) return EKF_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= count) return fail(EKF_ERR_INVALID, "device index out of range");
    HIPC(hipSetDevice(device));
    double *d_world = nullptr, *d_poses = nullptr, *d_ranges = nullptr;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_world, sizeof(double) * 2 * (n > 0 ? n : 1)));
        HIPC(hipMalloc((void**)&d_poses, sizeof(double) * 3 * S));
        HIPC(hipMalloc((void**)&d_ranges, sizeof(double) * (size_t)S * lidar->n_beams));
        if (n > 0) HIPC(hipMemcpy(d_world, world_xy, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(d_poses, poses, sizeof(double) * 3 * S, hipMemcpyHostToDevice));
        const ekf::LidarParams lp{lidar->n_beams, lidar->range_std, lidar->range_max, lidar->border_width,
                                  lidar->tube_radius, lidar->model, lidar->range_min};
        // B = S, t0 = step: scan s draws the noise stream of filter first_filter_id + s at that step
        ekf::launch_sim_scans(to_sim(sp), lp, S, n, S, step, d_world, d_poses, d_ranges, nullptr);
        HIPC(hipGetLastError());
        HIPC(hipMemcpy(ranges_out, d_ranges, sizeof(double) * (size_t)S * lidar->n_beams, hipMemcpyDeviceToHost));
        return EKF_OK;
    };
    const ekf_status st = body();
    for (void* q : {(void*)d_world, (void*)d_poses, (void*)d_ranges})
        if (q) (void)hipFree(q);
    return st;
}

ekf_status ekf_normalize_angles(int device, const double* in, int count, double* out) {
    if (count < 0 || (count > 0 && (!in || !out))) return fail(EKF_ERR_INVALID, "ekf_normalize_angles: bad argument");
    if (count == 0) return EKF_OK;
    int devs = 0;
    if (hipGetDeviceCount(&devs) != hipSuccess || devs <= 0)
        return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= devs) return fail(EKF_ERR_INVALID, "device index out of range");
    HIPC(hipSetDevice(device));
    double *d_in = nullptr, *d_out = nullptr;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_in, sizeof(double) * count));
        HIPC(hipMalloc((void**)&d_out, sizeof(double) * count));
        HIPC(hipMemcpy(d_in, in, sizeof(double) * count, hipMemcpyHostToDevice));
        ekf::launch_normalize_angles(d_in, count, d_out, nullptr);
        HIPC(hipGetLastError());
        HIPC(hipMemcpy(out, d_out, sizeof(double) * count, hipMemcpyDeviceToHost));
        return EKF_OK;
    };
    const ekf_status st = body();
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return st;
}

ekf_status ekf_circle_fit_scans(int device, const double* ranges, int S, int n_beams, int max_out, double* centres,
                                double* radii, int* counts, double* all_clusters, int* n_clusters) {
    if (!ranges || !centres || !radii || !counts || S < 0 || n_beams < 1 || max_out < 1 ||
        n_beams > ekf::circles_max_beams())
        return fail(EKF_ERR_INVALID, "ekf_circle_fit_scans: bad argument (n_beams must be 1..1024)");
    if (S == 0) return EKF_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(EKF_ERR_NO_DEVICE, "no HIP device visible: libekfslam_hip has no CPU path");
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= count) return fail(EKF_ERR_INVALID, "device index out of range");
    HIPC(hipSetDevice(device));
    const int mc = ekf::circles_max_clusters();
    double *d_r = nullptr, *d_c = nullptr, *d_rad = nullptr, *d_all = nullptr;
    int *d_cnt = nullptr, *d_nc = nullptr;
    ekf_status st = EKF_OK;
    auto body = [&]() -> ekf_status {
        HIPC(hipMalloc((void**)&d_r, sizeof(double) * (size_t)S * n_beams));
        HIPC(hipMalloc((void**)&d_c, sizeof(double) * (size_t)S * max_out * 2));
        HIPC(hipMalloc((void**)&d_rad, sizeof(double) * (size_t)S * max_out));
        HIPC(hipMalloc((void**)&d_cnt, sizeof(int) * (size_t)S));
        HIPC(hipMalloc((void**)&d_nc, sizeof(int) * (size_t)S));
        if (all_clusters) HIPC(hipMalloc((void**)&d_all, sizeof(double) * (size_t)S * mc * 4));
        HIPC(hipMemcpy(d_r, ranges, sizeof(double) * (size_t)S * n_beams, hipMemcpyHostToDevice));
        HIPC(hipMemset(d_c, 0, sizeof(double) * (size_t)S * max_out * 2));
        HIPC(hipMemset(d_rad, 0, sizeof(double) * (size_t)S * max_out));
        if (d_all) HIPC(hipMemset(d_all, 0, sizeof(double) * (size_t)S * mc * 4));
        ekf::launch_circles(d_r, S, n_beams, max_out, d_c, d_rad, d_cnt, d_all, d_nc, nullptr);
        HIPC(hipGetLastError());
        HIPC(hipMemcpy(centres, d_c, sizeof(double) * (size_t)S * max_out * 2, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(radii, d_rad, sizeof(double) * (size_t)S * max_out, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(counts, d_cnt, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost));
        if (n_clusters) HIPC(hipMemcpy(n_clusters, d_nc, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost));
        if (all_clusters) HIPC(hipMemcpy(all_clusters, d_all, sizeof(double) * (size_t)S * mc * 4, hipMemcpyDeviceToHost));
        return EKF_OK;
    };
    st = body();
    for (void* p : {(void*)d_r, (void*)d_c, (void*)d_rad, (void*)d_cnt, (void*)d_nc, (void*)d_all})
        if (p) (void)hipFree(p);
    return st;
}

// ---- dense fp32 propagation (configs[3]) -----------------------------------------------------

}  // extern "C"
