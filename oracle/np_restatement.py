"""Second, independent CPU restatement of rigid2d::EKF_SLAM in NumPy (TEST INFRASTRUCTURE ONLY).

Written literally from rigid2d/src/ekf_slam.cpp with dense ``@`` products and
``np.linalg.inv`` -- a second opinion on oracle/ekf_oracle.c (the two must agree to
<= 1e-12 per block, tests/test_oracle.py).  PARITY UNPINNED for the same reason as the C
oracle: the reference needs Armadillo and holds no EKF_SLAM fixtures."""
from __future__ import annotations

import math

import numpy as np

PI = 3.14159265358979323846  # rigid2d.hpp:13


def normalize_angle(rad):
    """rigid2d/src/rigid2d.cpp:336-345."""
    reduced = math.fmod(rad, 2 * PI)
    ang = math.fmod(reduced + 2 * PI, 2 * PI)
    if ang > PI:
        ang -= 2 * PI
    return ang


class NumpyEKF:
    def __init__(self, n):  # ekf_slam.cpp:27-53
        self.n = n
        N = 3 + 2 * n
        self.sigma = np.zeros((N, N))
        self.sigma[3:, 3:] = np.eye(2 * n) * 100
        self.Q = np.zeros((N, N))
        self.Q[0, 0] = self.Q[1, 1] = self.Q[2, 2] = 0.0001
        self.state = np.zeros(N)
        self.landmark_init_flag = False

    def prediction(self, dtheta, dx):  # ekf_slam.cpp:55-106
        N = 3 + 2 * self.n
        theta = self.state[0]
        update = np.zeros(N)
        A = np.zeros((N, N))
        if abs(dtheta) < 0.000001:
            update[1] = dx * math.cos(theta)
            update[2] = dx * math.sin(theta)
            A[1, 0] = -dx * math.sin(theta)
            A[2, 0] = dx * math.cos(theta)
        else:
            update[0] = dtheta
            update[1] = -(dx / dtheta) * math.sin(theta) + (dx / dtheta) * math.sin(theta + dtheta)
            update[2] = (dx / dtheta) * math.cos(theta) - (dx / dtheta) * math.cos(theta + dtheta)
            A[1, 0] = -(dx / dtheta) * math.cos(theta) + (dx / dtheta) * math.cos(theta + dtheta)
            A[2, 0] = -(dx / dtheta) * math.sin(theta) + (dx / dtheta) * math.sin(theta + dtheta)
        self.state = self.state + update
        At = np.eye(N) + A
        self.sigma = At @ self.sigma @ At.T + self.Q

    def _terms(self, i, sx, sy, theta, x, y):
        tx, ty = self.state[2 * i + 3], self.state[2 * i + 4]
        z = np.array([math.sqrt(sx ** 2 + sy ** 2), math.atan2(sy, sx)])
        zhat = np.array([math.sqrt((tx - x) ** 2 + (ty - y) ** 2),
                         normalize_angle(math.atan2(ty - y, tx - x) - theta)])
        dx_, dy_ = tx - x, ty - y
        d = dx_ ** 2 + dy_ ** 2
        H = np.zeros((2, 3 + 2 * self.n))
        H[:, 0:3] = [[0, -dx_ / math.sqrt(d), -dy_ / math.sqrt(d)], [-1, dy_ / d, -dx_ / d]]
        H[:, 3 + 2 * i:5 + 2 * i] = [[dx_ / math.sqrt(d), dy_ / math.sqrt(d)], [-dy_ / d, dx_ / d]]
        return z, zhat, H

    def _correct(self, i, sx, sy, theta, x, y):  # ekf_slam.cpp:137-192 / :331-390
        z, zhat, H = self._terms(i, sx, sy, theta, x, y)
        R = np.diag([0.01, 0.01])
        K = self.sigma @ H.T @ np.linalg.inv(H @ self.sigma @ H.T + R)
        zd = z - zhat
        zd[1] = normalize_angle(zd[1])
        self.state = self.state + K @ zd
        self.state[0] = normalize_angle(self.state[0])
        kh = K @ H
        self.sigma = (np.eye(kh.shape[0]) - kh) @ self.sigma

    def measurement(self, sensor_xy, visible):  # ekf_slam.cpp:108-197
        theta, x, y = self.state[0], self.state[1], self.state[2]
        if not self.landmark_init_flag:
            for i in range(self.n):
                sx, sy = sensor_xy[2 * i], sensor_xy[2 * i + 1]
                ri = math.sqrt(sx ** 2 + sy ** 2)
                phii = math.atan2(sy, sx)
                self.state[2 * i + 3] = x + ri * math.cos(phii + theta)
                self.state[2 * i + 4] = y + ri * math.sin(phii + theta)
            self.landmark_init_flag = True
        for i in range(self.n):
            if not visible[i]:
                continue
            self._correct(i, sensor_xy[2 * i], sensor_xy[2 * i + 1], theta, x, y)

    def maha(self, sx, sy, i):  # ekf_slam.cpp:217-276
        theta, x, y = self.state[0], self.state[1], self.state[2]
        z, zhat, H = self._terms(i, sx, sy, theta, x, y)
        psi = H @ self.sigma @ H.T + np.diag([0.01, 0.01])
        v = z - zhat
        return float(v @ np.linalg.inv(psi) @ v)

    def data_association(self, meas_xy, known):  # ekf_slam.cpp:278-402
        known_count = 0
        for k in known:
            if k:
                known_count += 1
            else:
                break
        assoc = []
        for (mx, my) in np.asarray(meas_xy, dtype=float).reshape(-1, 2):
            best, idx = 10.0, known_count
            for i in range(known_count):
                d = self.maha(mx, my, i)
                if d < best:
                    best, idx = d, i
            if idx == known_count and idx < self.n:
                theta, x, y = self.state[0], self.state[1], self.state[2]
                ri = math.sqrt(mx ** 2 + my ** 2)
                phii = math.atan2(my, mx)
                self.state[2 * idx + 3] = x + ri * math.cos(phii + theta)
                self.state[2 * idx + 4] = y + ri * math.sin(phii + theta)
                known[known_count] = 1
                known_count += 1
                best = 0.0
            if best < 1.0:
                self._correct(idx, mx, my, self.state[0], self.state[1], self.state[2])
                assoc.append(idx)
            else:
                assoc.append(-1)
        return np.array(assoc, dtype=np.int32)


# ---- rigid2d::CircleFitting, literal LAPACK-backed transcription (second opinion) ---------------------

def np_cluster(ranges):
    """circle_fitting.cpp:11-90 -> list of clusters, each a list of beam indices."""
    n = len(ranges)
    thres = 0.2
    clusters, cur = [], [0]
    for i in range(1, n):
        if abs(ranges[i] - ranges[i - 1]) < thres and i != n - 1:
            pass
        else:
            if len(cur) > 6:
                clusters.append(cur)
            cur = []
        cur.append(i)
    if not clusters:
        return []                      # the reference indexes an empty vector here (UB)
    if abs(ranges[clusters[0][0]] - ranges[clusters[-1][-1]]) < thres:
        last = clusters[-1]
        clusters[0] = list(last) + clusters[0]   # :63-66 (a single cluster is prepended to itself ...)
        clusters.pop()                          # :68-69 (... and popped)
    return clusters


def np_beam_xy(ranges, i):
    n = len(ranges)
    res = 2 * PI / n
    if i == 0:
        return ranges[0] * math.cos(0.0), ranges[0] * math.sin(0.0)
    a = normalize_angle(i * res)
    return ranges[i] * math.cos(a), ranges[i] * math.sin(a)


def np_circle_regress(xy):
    """circle_fitting.cpp:104-232 with numpy.linalg.svd / eig / solve -> (cx, cy, r)."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    m = len(xy)
    x_mean, y_mean = xy[:, 0].sum() / m, xy[:, 1].sum() / m
    x, y = xy[:, 0] - x_mean, xy[:, 1] - y_mean
    z = x ** 2 + y ** 2
    z_mean = z.sum() / m
    Z = np.stack([z, x, y, np.ones(m)], axis=1)
    H_inv = np.zeros((4, 4))
    H_inv[0, 3] = 0.5; H_inv[1, 1] = 1.0; H_inv[2, 2] = 1.0; H_inv[3, 0] = 0.5; H_inv[3, 3] = -2.0 * z_mean
    _, s, Vt = np.linalg.svd(Z, full_matrices=True)
    V = Vt.T
    if s[3] < 1e-12:
        A = V[:, 3]
    else:
        Y = V @ np.diag(s) @ V.T
        Q = Y @ H_inv @ Y
        w, E = np.linalg.eig(Q)
        idx, best = 0, 1000.0
        for e in range(4):
            if w[e].real > 0 and w[e].real < best:
                best, idx = w[e].real, e
        A = np.linalg.solve(Y, E[:, idx].real)
    a = -A[1] / (2 * A[0])
    b = -A[2] / (2 * A[0])
    R_sqr = (A[1] ** 2 + A[2] ** 2 - 4 * A[0] * A[3]) / (4 * A[0] ** 2)
    return np.array([a + x_mean, b + y_mean, math.sqrt(R_sqr)])


def np_is_circle(xy, radius):
    """circle_fitting.cpp:234-296."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    p1, p2 = xy[0], xy[-1]
    s = 0.0
    for k in range(1, len(xy) - 1):
        a, b = p1 - xy[k], p2 - xy[k]
        s += math.acos((a[0] * b[0] + a[1] * b[1]) / (math.hypot(a[0], a[1]) * math.hypot(b[0], b[1])))
    mean = s / (len(xy) - 2)
    return mean > 1.5708 and mean < 2.3562 and radius < 0.2


def np_approx_circle_positions(ranges):
    """circle_fitting.cpp:298-304 -> (clean centres, all clusters [x, y, r, is_circle])."""
    out, clean = [], []
    for cl in np_cluster(list(ranges)):
        xy = np.array([np_beam_xy(ranges, i) for i in cl])
        cx, cy, r = np_circle_regress(xy)
        ok = np_is_circle(xy, r)
        out.append([cx, cy, r, float(ok)])
        if ok:
            clean.append([cx, cy])
    return np.array(clean).reshape(-1, 2), np.array(out).reshape(-1, 4)
