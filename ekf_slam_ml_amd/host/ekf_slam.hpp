// ekf_slam.hpp -- header-only C++ mirror of rigid2d::EKF_SLAM over the C ABI (include/ekfslam.h).
//
// Same public surface as the reference class (rigid2d/include/rigid2d/ekf_slam.hpp:19-57): same
// method names, argument order and meaning, by-value / by-reference passing and in/out behaviour
// (known_list of data_association is updated in place, ekf_slam.cpp:323).  The reference's argument
// types come from Armadillo and rigid2d; this mirror is generic over them so that it compiles both
//   * on a ROS box with the real types (arma::mat, rigid2d::Twist2D, rigid2d::Vector2D) -- see
//     INTEGRATION.md for the two-file shim that makes it a drop-in for rigid2d/src/ekf_slam.cpp, and
//   * stand-alone (std::vector<double>, ekfslam::Twist2D, ekfslam::Vector2D below) for the tests.
//
// Ownership: the object owns one ekf_handle (device state).  It is default-constructible (empty, like
// EKF_SLAM::EKF_SLAM(), ekf_slam.cpp:24-25), copyable (deep device copy -- nuslam copy-assigns a
// temporary into its member, nuslam/src/slam.cpp:213,428) and movable.  Not thread-safe, exactly like
// the reference (single-threaded ROS spinner, slam.cpp:525).
//
// Errors: the reference declares none; Armadillo would throw std::logic_error / std::runtime_error.
// Here every failing C-ABI status becomes std::runtime_error(ekf_last_error()); calling a method on an
// empty object throws std::logic_error.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/ekfslam.h"

namespace ekfslam {

// Layout twin of rigid2d::Vector2D (rigid2d.hpp:68-72: two public doubles x, y).
struct Vector2D {
    double x = 0.0;
    double y = 0.0;
};

// Accessor twin of rigid2d::Twist2D (rigid2d.hpp:162-190, rigid2d.cpp:100-131).
class Twist2D {
public:
    Twist2D() = default;
    Twist2D(double angular, const Vector2D& linear) : ang(angular), lin(linear) {}
    double linearX() const { return lin.x; }
    double linearY() const { return lin.y; }
    double angular() const { return ang; }

private:
    double ang = 0.0;
    Vector2D lin;
};

namespace detail {
// arma::mat exposes memptr()/n_elem AND size() (and, in recent versions, data()-like members); std::vector
// exposes data()/size() only.  The overloads are RANKED (rank<1> is tried first, rank<0> is its base class)
// so a type offering both spellings takes the Armadillo one instead of being ambiguous.
template <int I> struct rank : rank<I - 1> {};
template <> struct rank<0> {};
template <class M>
auto data_of(const M& m, rank<1>) -> decltype(static_cast<const double*>(m.memptr())) { return m.memptr(); }
template <class M>
auto data_of(const M& m, rank<0>) -> decltype(static_cast<const double*>(m.data())) { return m.data(); }
template <class M>
const double* data_of(const M& m) { return data_of(m, rank<1>{}); }
template <class M>
auto size_of(const M& m, rank<1>) -> decltype(static_cast<size_t>(m.n_elem)) { return static_cast<size_t>(m.n_elem); }
template <class M>
auto size_of(const M& m, rank<0>) -> decltype(static_cast<size_t>(m.size())) { return static_cast<size_t>(m.size()); }
template <class M>
size_t size_of(const M& m) { return size_of(m, rank<1>{}); }

inline void check(ekf_status st, const char* where) {
    if (st != EKF_OK) throw std::runtime_error(std::string(where) + ": " + ekf_last_error());
}
}  // namespace detail

class EKF_SLAM {
public:
    /// create an empty EKF_SLAM object, no practical usage (ekf_slam.hpp:22-23)
    EKF_SLAM() = default;

    /// create an EKF_SLAM object for n_measurements tubes (ekf_slam.hpp:25-27, ekf_slam.cpp:27-53)
    explicit EKF_SLAM(int n_measurements, const ekf_params* params = nullptr, int device = -1) : n(n_measurements) {
        detail::check(ekf_create(n_measurements, params, device, &h), "EKF_SLAM");
    }

    EKF_SLAM(const EKF_SLAM& o) : n(o.n) {
        if (o.h) detail::check(ekf_clone(o.h, &h), "EKF_SLAM(copy)");
    }
    EKF_SLAM(EKF_SLAM&& o) noexcept : h(o.h), n(o.n) { o.h = nullptr; }
    EKF_SLAM& operator=(EKF_SLAM o) noexcept {  // copy-and-swap covers copy and move assignment
        std::swap(h, o.h);
        std::swap(n, o.n);
        return *this;
    }
    ~EKF_SLAM() { if (h) ekf_destroy(h); }

    /// prediction update stage based on odometry twist (ekf_slam.hpp:29-31, ekf_slam.cpp:55-106)
    template <class TwistT>
    void prediction(const TwistT& twist) {
        detail::check(ekf_predict(handle(), twist.angular(), twist.linearX()), "prediction");
    }

    /// correction update stage based on measurement (ekf_slam.hpp:33-36, ekf_slam.cpp:108-197).
    /// sensor_reading: 2n x 1 column (x, y per tube); known_list is unused by the reference.
    template <class MatT>
    void measurement(MatT sensor_reading, std::vector<bool> visible_list, std::vector<bool> known_list) {
        (void)known_list;
        if (detail::size_of(sensor_reading) != static_cast<size_t>(2 * n) || visible_list.size() != static_cast<size_t>(n))
            throw std::logic_error("measurement: sensor_reading must hold 2n values and visible_list n");
        std::vector<uint8_t> vis(visible_list.begin(), visible_list.end());  // vector<bool> is bit-packed
        detail::check(ekf_measure_known(handle(), detail::data_of(sensor_reading), vis.data()), "measurement");
    }

    /// unknown data association and correction update (ekf_slam.hpp:38-41, ekf_slam.cpp:278-402).
    /// known_list is IN/OUT: entries are set as landmarks are initialised (:323).
    template <class Vec2T>
    void data_association(std::vector<Vec2T> measures, std::vector<bool>& known_list) {
        static_assert(std::is_standard_layout<Vec2T>::value && sizeof(Vec2T) == 2 * sizeof(double),
                      "Vector2D must be two packed doubles {x, y} (rigid2d.hpp:68-72)");
        if (known_list.size() != static_cast<size_t>(n)) throw std::logic_error("data_association: known_list must hold n entries");
        std::vector<uint8_t> known(known_list.begin(), known_list.end());
        const double* xy = measures.empty() ? nullptr : &measures[0].x;
        detail::check(ekf_associate(handle(), xy, static_cast<int>(measures.size()), known.data(), nullptr),
                      "data_association");
        for (size_t i = 0; i < known.size(); i++) known_list[i] = known[i] != 0;
    }

    /// estimated x / y / orientation (ekf_slam.hpp:43-53, ekf_slam.cpp:404-414)
    double getStateX() { return pose()[1]; }
    double getStateY() { return pose()[2]; }
    double getStateTheta() { return pose()[0]; }

    /// estimated landmark positions, rows 3..N-1 of the state (ekf_slam.hpp:55-57, ekf_slam.cpp:416-418)
    std::vector<double> getStateLandmark() {
        std::vector<double> out(static_cast<size_t>(2 * n));
        detail::check(ekf_get_landmarks(handle(), out.data()), "getStateLandmark");
        return out;
    }
    /// the same as a 2n x 1 column of a matrix type constructible as MatT(const double* ptr, rows, cols)
    /// with copy semantics -- arma::mat's auxiliary-memory constructor (what the shim returns to nuslam)
    template <class MatT>
    MatT getStateLandmarkAs() {
        const std::vector<double> v = getStateLandmark();
        return MatT(v.data(), v.size(), 1);
    }

    // ---- beyond the reference surface: snapshot / restore (the reference has no checkpointing) ----
    int landmarks() const { return n; }
    int dim() const { return 3 + 2 * n; }
    bool empty() const { return h == nullptr; }
    ekf_handle native_handle() { return handle(); }
    std::vector<double> state() {
        std::vector<double> s(static_cast<size_t>(dim()));
        detail::check(ekf_get_state(handle(), s.data()), "state");
        return s;
    }
    std::vector<double> covariance() {  // row-major N x N
        std::vector<double> c(static_cast<size_t>(dim()) * dim());
        detail::check(ekf_get_cov(handle(), c.data()), "covariance");
        return c;
    }

private:
    ekf_handle h = nullptr;
    int n = 0;

    ekf_handle handle() const {
        if (!h) throw std::logic_error("EKF_SLAM: default-constructed (empty) object");
        return h;
    }
    struct Pose { double v[3]; double operator[](int i) const { return v[i]; } };
    Pose pose() {
        Pose p;
        detail::check(ekf_get_pose(handle(), p.v), "getState");
        return p;
    }
};

// Mirror of rigid2d::CircleFitting's production entry point (rigid2d/include/rigid2d/circle_fitting.hpp:27,
// rigid2d/src/circle_fitting.cpp:298-304): laser ranges in, centres of the clusters classified as circles out,
// in cluster order -- what nuslam/src/landmarks.cpp:141 publishes as the scan_sensor markers.
class CircleFitting {
public:
    explicit CircleFitting(int device = -1, int max_circles = 64) : dev(device), cap(max_circles) {}

    template <class Vec2T = Vector2D>
    std::vector<Vec2T> approxCirclePositions(std::vector<double> ranges) {
        static_assert(std::is_standard_layout<Vec2T>::value && sizeof(Vec2T) == 2 * sizeof(double),
                      "Vector2D must be two packed doubles {x, y} (rigid2d.hpp:68-72)");
        std::vector<double> centres(static_cast<size_t>(cap) * 2), radii(static_cast<size_t>(cap));
        int count = 0;
        detail::check(ekf_circle_fit_scans(dev, ranges.data(), 1, static_cast<int>(ranges.size()), cap, centres.data(),
                                           radii.data(), &count, nullptr, nullptr),
                      "approxCirclePositions");
        std::vector<Vec2T> out(static_cast<size_t>(count));
        for (int i = 0; i < count; i++) { out[i].x = centres[2 * i]; out[i].y = centres[2 * i + 1]; }
        r_cluster.assign(radii.begin(), radii.begin() + count);
        return out;
    }
    /// radii of the circles returned by the last call (the reference keeps r_cluster for ALL clusters)
    const std::vector<double>& get_r_circles() const { return r_cluster; }

private:
    int dev, cap;
    std::vector<double> r_cluster;
};

}  // namespace ekfslam
