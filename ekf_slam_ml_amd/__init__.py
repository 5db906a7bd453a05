"""MI355X-native EKF-SLAM filter core (hot path of tonylitianyu/EKF-SLAM-ML's rigid2d::EKF_SLAM)."""
