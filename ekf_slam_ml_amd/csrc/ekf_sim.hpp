// ekf_sim.hpp -- on-device Monte-Carlo log generator + consistency statistics (ekf_sim.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ekf_kernels.hpp"

namespace ekf {

struct SimParams {  // mirrors ekf_sim_params of include/ekfslam.h
    unsigned long long seed;
    long long first_filter_id;
    double v_cmd, w_cmd, vx_std, the_std, slip_min, slip_max, sensor_std, max_visible_dis;
    double wheel_base, wheel_radius;
    int ticks_per_step;
};

struct LidarParams {  // mirrors ekf_lidar_params of include/ekfslam.h
    int n_beams;
    double range_std, range_max, border_width, tube_radius;
    int model;          // 0 clean ray geometry, 1 publishScan's bearing window + line-circle procedure
    double range_min;
};

// unknown-association inputs: twist/truth as launch_sim, then per (step, filter) up to jmax shuffled readings
// of the landmarks within the visibility radius -> count [T][B], meas [T][B][jmax][2]
void launch_sim_unknown(const SimParams& p, int B, int n, int T, int jmax, const double* world, double* twist,
                        double* truth, int* count, double* meas, bool trajectory, hipStream_t s);
// simulated laser scans: scan s = (step t0 + s / B, filter s % B) seen from poses[s] -> ranges [S][n_beams]
void launch_sim_scans(const SimParams& p, const LidarParams& lp, int B, int n, int S, int t0, const double* world,
                      const double* poses, double* ranges, hipStream_t s);

void launch_sim(const SimParams& p, int B, int n, int T, int vmax, const double* world, double* twist, double* truth,
                int* lm_idx, double* z_xy, double* init_xy, int* slot_active, hipStream_t s);
void launch_mc_stats(const PoolView& pv, const double* truth_t, double* out, hipStream_t s);

}  // namespace ekf
