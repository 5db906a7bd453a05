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
