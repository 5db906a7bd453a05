// Drop-in replacement for rigid2d/include/rigid2d/ekf_slam.hpp of tonylitianyu/EKF-SLAM-ML: the eight public
// signatures nuslam compiles against (reference header :19-57) over the MI355X filter core of this repository.
// Copy this file and ../../src/ekf_slam.cpp over the reference's two files, add this repository's root to the
// include path and link ekf_slam_ml_amd/libekfslam_hip.so.  tests/test_shim.py compiles
// both files against the reference's real rigid2d.hpp.
#ifndef EKF_SLAM_INCLUDE_GUARD_HPP
#define EKF_SLAM_INCLUDE_GUARD_HPP

#include <vector>
#include "rigid2d.hpp"      // rigid2d::Twist2D, rigid2d::Vector2D -- unchanged reference header
#include <armadillo>        // the node passes arma::mat in and expects arma::mat back
#include "ekf_slam_ml_amd/host/ekf_slam.hpp"

using namespace arma;       // nuslam relies on it: slam.cpp:217,380 and unknown_data_assoc.cpp spell bare `mat`

namespace rigid2d
{
    class EKF_SLAM {
    public:
        EKF_SLAM();
        EKF_SLAM(int n_measurements);
        void prediction(const rigid2d::Twist2D & twist);
        void measurement(mat sensor_reading, std::vector<bool> visible_list, std::vector<bool> known_list);
        void data_association(std::vector<rigid2d::Vector2D> measures, std::vector<bool> &known_list);
        double getStateX();
        double getStateY();
        double getStateTheta();
        mat getStateLandmark();

    private:
        // state, Q, sigma, n and landmark_init_flag of the reference (:61-65) live in HBM behind this handle owner.
        // Its rule of five gives this class the copy/move behaviour `slam_agent = rigid2d::EKF_SLAM(n)` needs
        // (nuslam/src/slam.cpp:213,428): copy = deep device copy, move = handle swap.
        ekfslam::EKF_SLAM core;
    };
}

#endif
