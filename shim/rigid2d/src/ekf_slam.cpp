// Drop-in replacement for rigid2d/src/ekf_slam.cpp: every method forwards to the header-only mirror
// (ekf_slam_ml_amd/host/ekf_slam.hpp), which marshals the arguments into the C ABI of include/ekfslam.h.
// The reference line ranges name what each C entry point replaces.
#include "rigid2d/ekf_slam.hpp"

namespace rigid2d
{
    EKF_SLAM::EKF_SLAM() {}                                            // :24-25, empty object

    EKF_SLAM::EKF_SLAM(int n_measurements) : core(n_measurements) {}   // :27-53  -> ekf_create

    void EKF_SLAM::prediction(const rigid2d::Twist2D & twist)          // :55-106 -> ekf_predict
    {
        core.prediction(twist);   // angular() and linearX(); linearY() is ignored like :70
    }

    void EKF_SLAM::measurement(mat sensor_reading, std::vector<bool> visible_list, std::vector<bool> known_list)
    {
        core.measurement(sensor_reading, visible_list, known_list);   // :108-197 -> ekf_measure_known (memptr, n_elem)
    }

    void EKF_SLAM::data_association(std::vector<rigid2d::Vector2D> measures, std::vector<bool> &known_list)
    {
        core.data_association(measures, known_list);   // :278-402 -> ekf_associate; known_list is in/out (:323)
    }

    double EKF_SLAM::getStateX() { return core.getStateX(); }          // :404-406 -> ekf_get_pose
    double EKF_SLAM::getStateY() { return core.getStateY(); }          // :408-410
    double EKF_SLAM::getStateTheta() { return core.getStateTheta(); }  // :412-414

    mat EKF_SLAM::getStateLandmark()                                   // :416-418 -> ekf_get_landmarks
    {
        return core.getStateLandmarkAs<mat>();   // 2n x 1 column, copied out of the staging vector
    }
}
