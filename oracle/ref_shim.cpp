// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" doors into the reference's OWN rigid2d.cpp / diff_drive.cpp, which
// oracle/Makefile compiles from where they lie under /root/reference (they need
// libm only).  Used to pin oracle/ekf_oracle.c's normalize_angle and body-twist
// restatements against the real reference code.  ekf_slam.cpp itself cannot be
// built here (needs Armadillo), see oracle/ekf_oracle.c header.
#include "rigid2d/rigid2d.hpp"
#include "rigid2d/diff_drive.hpp"

extern "C" {

// rigid2d/src/rigid2d.cpp:336-345
double ref_normalize_angle(double rad) { return rigid2d::normalize_angle(rad); }

// rigid2d/src/diff_drive.cpp:38-47 via the accessors rigid2d.cpp:116-131
void ref_body_twist(double wheel_base, double wheel_radius, double left, double right, double* out) {
    rigid2d::DiffDrive dd(wheel_base, wheel_radius);
    rigid2d::Twist2D t = dd.getBodyTwistForUpdate(left, right);
    out[0] = t.angular();
    out[1] = t.linearX();
    out[2] = t.linearY();
}

// nuslam/src/slam.cpp:173-176: Odometer::getCurrentTwist scales the 100 Hz wheel deltas x10
void ref_current_twist(double wheel_base, double wheel_radius, double dleft, double dright, double* out) {
    rigid2d::DiffDrive dd(wheel_base, wheel_radius);
    rigid2d::Twist2D t = dd.getBodyTwistForUpdate(dleft * 10.0, dright * 10.0);
    out[0] = t.angular();
    out[1] = t.linearX();
    out[2] = t.linearY();
}

// DiffDrive::updatePose KATs live in rigid2d/tests/tests.cpp:334-383
void ref_update_pose(double wheel_base, double wheel_radius, double left, double right, double* out) {
    rigid2d::DiffDrive dd(wheel_base, wheel_radius);
    dd.updatePose(left, right);
    out[0] = dd.getTheta();
    out[1] = dd.getPosition().x;
    out[2] = dd.getPosition().y;
}

}
