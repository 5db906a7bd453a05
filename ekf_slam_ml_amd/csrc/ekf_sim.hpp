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

void launch_sim(const SimParams& p, int B, int n, int T, int vmax, const double* world, double* twist, double* truth,
                int* lm_idx, double* z_xy, double* init_xy, int* slot_active, hipStream_t s);
void launch_mc_stats(const PoolView& pv, const double* truth_t, double* out, hipStream_t s);

}  // namespace ekf
