"""ctypes door to oracle/libekf_oracle*.so (TEST INFRASTRUCTURE ONLY, see ekf_oracle.c)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_ubyte)


def build(force=False):
    """Compile the checker (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libekf_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ekf_oracle.c", "circle_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)


def _cpu_has(*flags):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":")[1].split())
                    return all(x in have for x in flags)
    except OSError:
        pass
    return False


def _load(fast=False):
    build()
    name = "libekf_oracle_fast.so" if (fast and _cpu_has("avx2", "fma")) else "libekf_oracle.so"
    # EKF_ORACLE_SANITIZED=1 (tests/test_sanitizers.py, `make -C oracle asan`): the same sources built with
    # -fsanitize=address,undefined; the process must have been started with libasan preloaded
    if os.environ.get("EKF_ORACLE_SANITIZED") == "1":
        name = "libekf_oracle_asan.so"
    lib = C.CDLL(os.path.join(_HERE, name))
    lib.ekfo_create.restype = C.c_void_p
    lib.ekfo_create.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.ekfo_destroy.argtypes = [C.c_void_p]
    lib.ekfo_dim.argtypes = [C.c_void_p]
    lib.ekfo_normalize_angle.restype = C.c_double
    lib.ekfo_normalize_angle.argtypes = [C.c_double]
    lib.ekfo_body_twist.argtypes = [C.c_double] * 4 + [_dp]
    lib.ekfo_prediction.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.ekfo_measurement.argtypes = [C.c_void_p, _dp, _bp]
    lib.ekfo_measurement_compact.restype = C.c_int
    lib.ekfo_measurement_compact.argtypes = [C.c_void_p, _dp, _ip, _dp, C.c_int]
    lib.ekfo_maha.restype = C.c_double
    lib.ekfo_maha.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
    lib.ekfo_data_association.restype = C.c_int
    lib.ekfo_data_association.argtypes = [C.c_void_p, _dp, C.c_int, _bp, _ip]
    lib.ekfo_data_association_m.restype = C.c_int
    lib.ekfo_data_association_m.argtypes = [C.c_void_p, _dp, C.c_int, _bp, _ip, _dp]
    for f in ("ekfo_get_state", "ekfo_set_state", "ekfo_get_cov", "ekfo_set_cov"):
        getattr(lib, f).argtypes = [C.c_void_p, _dp]
    lib.ekfo_set_init_flag.argtypes = [C.c_void_p, C.c_int]
    lib.ekfo_get_init_flag.argtypes = [C.c_void_p]
    lib.ekfo_batch_run_known.restype = C.c_int
    lib.ekfo_batch_run_known.argtypes = [C.c_int] * 6 + [_dp, _ip, _dp, _dp, _dp, _dp, C.c_int, _dp]
    lib.cf_regress.argtypes = [_dp, _dp, C.c_int, _dp]
    lib.cf_is_circle.restype = C.c_int
    lib.cf_is_circle.argtypes = [_dp, _dp, C.c_int, C.c_double]
    lib.cf_approx_circle_positions.restype = C.c_int
    lib.cf_approx_circle_positions.argtypes = [_dp, C.c_int, C.c_int, _dp, _dp, _dp, _ip]
    lib.cf_cluster_summary.restype = C.c_int
    lib.cf_cluster_summary.argtypes = [_dp, C.c_int, C.c_int, _ip, _dp]
    return lib


_libs = {}


def lib(fast=False):
    if fast not in _libs:
        _libs[fast] = _load(fast)
    return _libs[fast]


def _d(a):
    return a.ctypes.data_as(_dp)


DENSE, STRUCTURED = 0, 1
MARGIN_KEYS = ("to_gate_new", "to_gate_update", "winner_to_runner_up", "smallest_score", "decision_relevant")


def new_margins():
    return np.full(5, np.inf)


class OracleEKF:
    """Mirror of rigid2d::EKF_SLAM (ekf_slam.hpp:19-57) over the C restatement."""

    def __init__(self, n, mode=DENSE, fast=False, params=None):
        """params: optional 6-tuple (sigma0_landmark, q_pose, r_meas, gate_new, gate_update, straight_eps)."""
        self._lib = lib(fast)
        self.n, self.N, self.mode = n, 3 + 2 * n, mode
        pp = (C.c_double * 6)(*[float(x) for x in params]) if params is not None else None
        self._h = self._lib.ekfo_create(n, mode, pp)
        if not self._h:
            raise MemoryError("ekfo_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ekfo_destroy(self._h)
            self._h = None

    def prediction(self, dtheta, dx):
        self._lib.ekfo_prediction(self._h, float(dtheta), float(dx))

    def measurement(self, sensor_xy, visible):
        s = np.ascontiguousarray(sensor_xy, dtype=np.float64)
        v = np.ascontiguousarray(visible, dtype=np.uint8)
        assert s.size == 2 * self.n and v.size == self.n
        self._lib.ekfo_measurement(self._h, _d(s), v.ctypes.data_as(_bp))

    def measurement_compact(self, init_xy, lm_idx, z_xy):
        i = np.ascontiguousarray(init_xy, dtype=np.float64)
        l = np.ascontiguousarray(lm_idx, dtype=np.int32)
        z = np.ascontiguousarray(z_xy, dtype=np.float64)
        return self._lib.ekfo_measurement_compact(self._h, _d(i), l.ctypes.data_as(_ip), _d(z), l.size)

    def maha(self, mx, my, i):
        return self._lib.ekfo_maha(self._h, float(mx), float(my), int(i))

    def data_association(self, meas_xy, known, margins=None):
        """known: uint8[n] numpy array, modified in place; returns per-measurement landmark (-1 dropped).
        margins: optional float64[5] (new_margins()), MIN-accumulated over calls: relative distance of every score to
        gate_new / gate_update, winner-to-runner-up gap, smallest score, decision-relevant minimum (ekf_oracle.c, ekfo_data_association_m)."""
        m = np.ascontiguousarray(meas_xy, dtype=np.float64).reshape(-1, 2)
        assert known.dtype == np.uint8 and known.size == self.n and known.flags.c_contiguous
        assoc = np.full(len(m), -1, dtype=np.int32)
        if margins is not None:
            assert margins.dtype == np.float64 and margins.size == 5 and margins.flags.c_contiguous
            self._lib.ekfo_data_association_m(self._h, _d(m), len(m), known.ctypes.data_as(_bp),
                                              assoc.ctypes.data_as(_ip), _d(margins))
            return assoc
        self._lib.ekfo_data_association(self._h, _d(m), len(m), known.ctypes.data_as(_bp),
                                        assoc.ctypes.data_as(_ip))
        return assoc

    @property
    def state(self):
        out = np.empty(self.N)
        self._lib.ekfo_get_state(self._h, _d(out))
        return out

    @state.setter
    def state(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert v.size == self.N
        self._lib.ekfo_set_state(self._h, _d(v))

    @property
    def cov(self):
        out = np.empty((self.N, self.N))
        self._lib.ekfo_get_cov(self._h, _d(out))
        return out

    @cov.setter
    def cov(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert v.shape == (self.N, self.N)
        self._lib.ekfo_set_cov(self._h, _d(v))

    def set_init_flag(self, f):
        self._lib.ekfo_set_init_flag(self._h, int(f))


def normalize_angle(rad):
    return lib().ekfo_normalize_angle(float(rad))


def body_twist(wheel_base, wheel_radius, left, right):
    out = np.zeros(2)
    lib().ekfo_body_twist(wheel_base, wheel_radius, left, right, _d(out))
    return out


def batch_run_known(log, mode=STRUCTURED, t_warm=0, nthreads=0, want_cov=False, fast=True):
    """Replay a synth.KnownLog on the CPU checker; returns (state[B,N], cov or None, stats)."""
    cfg = log.cfg
    B, n, T, vmax = cfg.filters, cfg.n, cfg.steps, log.lm_idx.shape[2]
    N = 3 + 2 * n
    tw = np.ascontiguousarray(log.twist, dtype=np.float64)
    li = np.ascontiguousarray(log.lm_idx, dtype=np.int32)
    zz = np.ascontiguousarray(log.z_xy, dtype=np.float64)
    ii = np.ascontiguousarray(log.init_xy, dtype=np.float64)
    st = np.empty((B, N))
    cv = np.empty((B, N, N)) if want_cov else None
    stats = np.zeros(3)
    rc = lib(fast).ekfo_batch_run_known(B, n, mode, T, t_warm, vmax, _d(tw), li.ctypes.data_as(_ip), _d(zz),
                                       _d(ii), _d(st), _d(cv) if want_cov else None, nthreads, _d(stats))
    if rc != 0:
        raise MemoryError("ekfo_batch_run_known failed")
    return st, cv, {"seconds": stats[0], "corrections": int(stats[1]), "threads": int(stats[2])}


class RefRigid2D:
    """oracle/_ref: the reference's own rigid2d.cpp/diff_drive.cpp (built by oracle/Makefile)."""

    def __init__(self):
        p = os.path.join(_HERE, "_ref", "librigid2d_ref.so")
        if not os.path.exists(p):
            raise FileNotFoundError(p)
        self._lib = C.CDLL(p)
        self._lib.ref_normalize_angle.restype = C.c_double
        self._lib.ref_normalize_angle.argtypes = [C.c_double]
        for f in ("ref_body_twist", "ref_current_twist", "ref_update_pose"):
            getattr(self._lib, f).argtypes = [C.c_double] * 4 + [_dp]

    def normalize_angle(self, r):
        return self._lib.ref_normalize_angle(float(r))

    def _call3(self, f, *a):
        out = np.zeros(3)
        getattr(self._lib, f)(*[float(x) for x in a], _d(out))
        return out

    def body_twist(self, wb, wr, left, right):
        return self._call3("ref_body_twist", wb, wr, left, right)

    def current_twist(self, wb, wr, dl, dr):
        return self._call3("ref_current_twist", wb, wr, dl, dr)

    def update_pose(self, wb, wr, left, right):
        return self._call3("ref_update_pose", wb, wr, left, right)


# ---- rigid2d::CircleFitting (oracle/circle_oracle.c) -------------------------------------------------

def circle_regress(xy):
    """circleRegression() for one cluster of (x, y) points -> (cx, cy, r); circle_fitting.cpp:104-232."""
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    xs, ys = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    out = np.zeros(3)
    lib().cf_regress(_d(xs), _d(ys), len(xs), _d(out))
    return out


def circle_clusters(ranges):
    """clusteringRanges(): (sizes, first range of each cluster); circle_fitting.cpp:11-90."""
    r = np.ascontiguousarray(ranges, dtype=np.float64)
    sizes = np.zeros(300, dtype=np.int32)
    first = np.zeros(300)
    nc = lib().cf_cluster_summary(_d(r), len(r), 300, sizes.ctypes.data_as(_ip), _d(first))
    return sizes[:nc].copy(), first[:nc].copy()


def approx_circle_positions(ranges, max_out=64):
    """approxCirclePositions(): (clean centres [k,2], clean radii [k], all clusters [c,4] = x,y,r,is_circle)."""
    r = np.ascontiguousarray(ranges, dtype=np.float64)
    xy, rad, allc = np.zeros((max_out, 2)), np.zeros(max_out), np.zeros((300, 4))
    nc = C.c_int()
    k = lib().cf_approx_circle_positions(_d(r), len(r), max_out, _d(xy), _d(rad), _d(allc), C.byref(nc))
    return xy[:k].copy(), rad[:k].copy(), allc[:nc.value].copy()
