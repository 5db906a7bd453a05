"""ctypes binding of libekfslam_hip.so (include/ekfslam.h) -- the only way Python reaches the filter.

There is no fallback: if the shared library is missing or no gfx950 device is visible, every
constructor raises.  The classes mirror the reference's call surface:

  EKF_SLAM      rigid2d::EKF_SLAM                      rigid2d/include/rigid2d/ekf_slam.hpp:19-57
  BatchEKF      B independent EKF_SLAM objects driven by a device-resident log (configs[4])
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EKF_LIB_PATH", os.path.join(_HERE, "libekfslam_hip.so"))  # (override: A/B of two builds)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_uint8)
_fp = C.POINTER(C.c_float)

STATUS = {0: "EKF_OK", 1: "EKF_ERR_INVALID", 2: "EKF_ERR_NO_DEVICE", 3: "EKF_ERR_HIP", 4: "EKF_ERR_NOMEM",
          5: "EKF_ERR_STATE"}

# every symbol include/ekfslam.h declares (tests/test_host.py checks the .so exports them all)
SYMBOLS = [
    "ekf_last_error", "ekf_default_params", "ekf_leading_dimension", "ekf_device_count",
    "ekf_create", "ekf_destroy", "ekf_clone", "ekf_predict", "ekf_measure_known", "ekf_associate",
    "ekf_maha_scores", "ekf_get_pose", "ekf_get_landmarks", "ekf_dim", "ekf_get_state", "ekf_set_state",
    "ekf_get_cov", "ekf_set_cov", "ekf_get_init_flag", "ekf_set_init_flag", "ekf_sync", "ekf_set_tuning", "ekf_set_active_set", "ekf_batch_set_active_set", "ekf_batch_get_touched",
    "ekf_batch_create", "ekf_batch_destroy", "ekf_batch_reset", "ekf_batch_device_bytes",
    "ekf_batch_upload_known_log", "ekf_batch_run_known", "ekf_batch_upload_unknown_log", "ekf_batch_run_unknown",
    "ekf_batch_get_known_counts", "ekf_batch_get_decisions", "ekf_batch_get_state", "ekf_batch_get_cov",
    "ekf_batch_get_poses", "ekf_batch_checksum", "ekf_batch_set_tuning",
    "ekf_set_update_mode", "ekf_batch_set_update_mode",
    "ekf_default_sim_params", "ekf_batch_simulate_known_log", "ekf_batch_download_log", "ekf_batch_mc_stats",
    "ekf_circle_fit_scans", "ekf_normalize_angles",
    "ekf_default_lidar_params", "ekf_batch_simulate_unknown_log", "ekf_batch_download_unknown_log", "ekf_simulate_scans",
    "ekf_dense_create", "ekf_dense_destroy", "ekf_dense_set", "ekf_dense_propagate", "ekf_dense_get_sigma",
    "ekf_dense_launch_info", "ekf_dense_tile_map", "ekf_batch_rank2_variant", "ekf_batch_rank2_resident",
    "ekf_set_profiling", "ekf_get_profile", "ekf_batch_set_known_counts",
    "ekf_set_forms", "ekf_get_forms", "ekf_batch_set_forms", "ekf_batch_get_forms", "ekf_batch_form_counts",
    "ekf_phase_trace", "ekf_test_raise_device_error",
]

# ekf_form (include/ekfslam.h): launch structures the library may take where they apply; all exact forms are bit-identical
FORM_SMALL_MAP, FORM_CALL_FUSED, FORM_ACTIVE_PREFIX = 1 << 0, 1 << 2, 1 << 3   # (1 << 1: retired, ignored)
FORM_STEP_FUSED, FORM_STEP_SPLIT_PASS, FORM_DELAYED_PAIR, FORM_ROW_PACKING = 1 << 4, 1 << 5, 1 << 6, 1 << 7
FORM_STRIP_FLUSH, FORM_STRIP_FLUSH_ALWAYS = 1 << 8, 1 << 9
FORM_COLUMN_PANEL, FORM_COLUMN_PANEL_ONE_SLOT, FORM_STEP_SPECULATE, FORM_CURRENT_COLUMNS = 1 << 10, 1 << 11, 1 << 12, 1 << 13
FORM_TILE_QUEUE = 1 << 14
FORMS_DEFAULT = ((1 << 9) - 1) | FORM_COLUMN_PANEL | FORM_STEP_SPECULATE | FORM_CURRENT_COLUMNS | FORM_TILE_QUEUE


class EkfError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"{STATUS.get(status, status)}: {text}")
        self.status = status


class Params(C.Structure):
    """ekf_params (include/ekfslam.h); defaults are the reference's hard-coded constants."""
    _fields_ = [("sigma0_landmark", C.c_double), ("q_pose", C.c_double), ("r_meas", C.c_double),
                ("gate_new", C.c_double), ("gate_update", C.c_double), ("straight_eps", C.c_double)]


class KnownLogC(C.Structure):
    _fields_ = [("T", C.c_int), ("vmax", C.c_int), ("twist", _dp), ("lm_idx", _ip), ("z_xy", _dp), ("init_xy", _dp)]


class UnknownLogC(C.Structure):
    _fields_ = [("T", C.c_int), ("jmax", C.c_int), ("twist", _dp), ("count", _ip), ("meas_xy", _dp)]


class SimParams(C.Structure):
    """ekf_sim_params (include/ekfslam.h): noise model of the reference's simulator."""
    _fields_ = [("seed", C.c_ulonglong), ("first_filter_id", C.c_longlong), ("v_cmd", C.c_double), ("w_cmd", C.c_double),
                ("vx_std", C.c_double), ("the_std", C.c_double), ("slip_min", C.c_double), ("slip_max", C.c_double),
                ("sensor_std", C.c_double), ("max_visible_dis", C.c_double), ("wheel_base", C.c_double),
                ("wheel_radius", C.c_double), ("ticks_per_step", C.c_int)]


class LidarParams(C.Structure):
    """ekf_lidar_params (include/ekfslam.h): the simulator's 2-D lidar."""
    _fields_ = [("n_beams", C.c_int), ("range_std", C.c_double), ("range_max", C.c_double),
                ("border_width", C.c_double), ("tube_radius", C.c_double), ("model", C.c_int), ("range_min", C.c_double)]


def default_lidar(**kw):
    lp = LidarParams()
    load().ekf_default_lidar_params(C.byref(lp))
    for k, v in kw.items():
        setattr(lp, k, v)
    return lp


class RunStats(C.Structure):
    _fields_ = [("elapsed_ms", C.c_double), ("rank2_ms", C.c_double), ("rank2_launches", C.c_longlong),
                ("corrections", C.c_longlong), ("filter_steps", C.c_longlong),
                ("rank2_bytes_per_launch", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """Load libekfslam_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; if torch will be used in this process it must be loaded
    # first so that both share ONE HIP runtime (the dynamic loader dedups by SONAME).
    lib = C.CDLL(LIB_PATH)
    h = C.c_void_p
    lib.ekf_last_error.restype = C.c_char_p
    lib.ekf_default_params.argtypes = [C.POINTER(Params)]
    lib.ekf_leading_dimension.argtypes = [C.c_int]
    lib.ekf_leading_dimension.restype = C.c_int
    lib.ekf_device_count.restype = C.c_int
    lib.ekf_default_sim_params.argtypes = [C.POINTER(SimParams)]
    lib.ekf_default_lidar_params.argtypes = [C.POINTER(LidarParams)]
    lib.ekf_default_lidar_params.restype = None
    sig = {
        "ekf_create": [C.c_int, C.POINTER(Params), C.c_int, C.POINTER(h)],
        "ekf_destroy": [h],
        "ekf_clone": [h, C.POINTER(h)],
        "ekf_predict": [h, C.c_double, C.c_double],
        "ekf_measure_known": [h, _dp, _bp],
        "ekf_associate": [h, _dp, C.c_int, _bp, _ip],
        "ekf_maha_scores": [h, C.c_double, C.c_double, C.c_int, _dp],
        "ekf_get_pose": [h, _dp],
        "ekf_get_landmarks": [h, _dp],
        "ekf_dim": [h, _ip, _ip],
        "ekf_get_state": [h, _dp],
        "ekf_set_state": [h, _dp],
        "ekf_get_cov": [h, _dp],
        "ekf_set_cov": [h, _dp],
        "ekf_get_init_flag": [h, _ip],
        "ekf_set_init_flag": [h, C.c_int],
        "ekf_sync": [h],
        "ekf_set_active_set": [h, C.c_int],
        "ekf_batch_set_active_set": [h, C.c_int],
        "ekf_batch_get_touched": [h, _ip],
        "ekf_batch_upload_unknown_log": [h, C.POINTER(UnknownLogC)],
        "ekf_batch_run_unknown": [h, C.c_int, C.c_int, C.c_int, C.POINTER(RunStats)],
        "ekf_batch_get_known_counts": [h, _ip],
        "ekf_batch_get_decisions": [h, _ip],
        "ekf_set_tuning": [h, C.c_int, C.c_int, C.c_int],
        "ekf_batch_create": [C.c_int, C.c_int, C.POINTER(Params), C.c_int, C.POINTER(h)],
        "ekf_batch_destroy": [h],
        "ekf_batch_reset": [h],
        "ekf_batch_device_bytes": [h, C.POINTER(C.c_size_t)],
        "ekf_batch_upload_known_log": [h, C.POINTER(KnownLogC)],
        "ekf_batch_run_known": [h, C.c_int, C.c_int, C.c_int, C.POINTER(RunStats)],
        "ekf_batch_get_state": [h, C.c_int, _dp],
        "ekf_batch_get_cov": [h, C.c_int, _dp],
        "ekf_batch_get_poses": [h, _dp],
        "ekf_batch_checksum": [h, _dp],
        "ekf_batch_set_tuning": [h, C.c_int, C.c_int, C.c_int],
        "ekf_set_update_mode": [h, C.c_int, C.c_int],
        "ekf_batch_set_update_mode": [h, C.c_int, C.c_int],
        "ekf_batch_simulate_known_log": [h, C.POINTER(SimParams), _dp, C.c_int, C.c_int],
        "ekf_batch_download_log": [h, _dp, _ip, _dp, _dp, _dp],
        "ekf_batch_mc_stats": [h, C.c_int, _dp],
        "ekf_batch_simulate_unknown_log": [h, C.POINTER(SimParams), C.POINTER(LidarParams), _dp, C.c_int, C.c_int],
        "ekf_batch_download_unknown_log": [h, _dp, _ip, _dp, _dp],
        "ekf_simulate_scans": [C.c_int, C.POINTER(SimParams), C.POINTER(LidarParams), _dp, C.c_int, _dp, C.c_int,
                               C.c_int, _dp],
        "ekf_normalize_angles": [C.c_int, _dp, C.c_int, _dp],
        "ekf_circle_fit_scans": [C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _ip, _dp, _ip],
        "ekf_dense_create": [C.c_int, C.c_int, C.POINTER(h)],
        "ekf_dense_destroy": [h],
        "ekf_dense_set": [h, _fp, _fp, _fp],
        "ekf_dense_propagate": [h, C.c_int, _dp],
        "ekf_dense_get_sigma": [h, _fp],
        "ekf_dense_launch_info": [h, _ip, _ip, _ip, _ip],
        "ekf_dense_tile_map": [h, _bp],
        "ekf_batch_rank2_variant": [h, _ip, _ip, _ip, _ip],
        "ekf_batch_rank2_resident": [h, _ip],
        "ekf_batch_set_known_counts": [h, _ip],
        "ekf_set_profiling": [h, C.c_int],
        "ekf_set_forms": [h, C.c_uint],
        "ekf_get_forms": [h, C.POINTER(C.c_uint)],
        "ekf_batch_set_forms": [h, C.c_uint],
        "ekf_batch_get_forms": [h, C.POINTER(C.c_uint)],
        "ekf_batch_form_counts": [h, C.POINTER(C.c_longlong)],
        "ekf_phase_trace": [h, C.c_int, C.POINTER(C.c_longlong)],
        "ekf_test_raise_device_error": [h],
        "ekf_get_profile": [h, _dp, C.POINTER(C.c_longlong)],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib


def _check(st):
    if st != 0:
        raise EkfError(st, load().ekf_last_error().decode(errors="replace"))


def device_count():
    return load().ekf_device_count()


def leading_dimension(n):
    """doubles between two rows of a filter's covariance for a map of n landmarks (ekf_leading_dimension; needs no device)"""
    return load().ekf_leading_dimension(int(n))


def default_params():
    p = Params()
    load().ekf_default_params(C.byref(p))
    return p


def _d(a):
    return a.ctypes.data_as(_dp)


class EKF_SLAM:
    """rigid2d::EKF_SLAM over the C ABI: same method names, argument meaning and in/out behaviour
    as rigid2d/include/rigid2d/ekf_slam.hpp:19-57 (arma::mat -> float64 array,
    std::vector<bool> -> uint8 array, Twist2D -> (angular, linearX), Vector2D -> (x, y) rows)."""

    def __init__(self, n_measurements, params=None, device=-1, _handle=None):
        self._lib = load()
        self.n = int(n_measurements)
        self.N = 3 + 2 * self.n
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        _check(self._lib.ekf_create(self.n, C.byref(params) if params is not None else None, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ekf_destroy(self._h)
            self._h = None

    __del__ = close

    def clone(self):
        """Copy construction / assignment (slam_agent = EKF_SLAM(n), nuslam/src/slam.cpp:428)."""
        h = C.c_void_p()
        _check(self._lib.ekf_clone(self._h, C.byref(h)))
        return EKF_SLAM(self.n, _handle=h)

    def prediction(self, twist):
        """twist = (angular, linearX[, linearY]); linearY is ignored like ekf_slam.cpp:70."""
        _check(self._lib.ekf_predict(self._h, float(twist[0]), float(twist[1])))

    def measurement(self, sensor_reading, visible_list, known_list=None):
        s = np.ascontiguousarray(sensor_reading, dtype=np.float64).reshape(-1)
        v = np.ascontiguousarray(visible_list, dtype=np.uint8).reshape(-1)
        if s.size != 2 * self.n or v.size != self.n:
            raise ValueError("sensor_reading must hold 2n values and visible_list n")
        _check(self._lib.ekf_measure_known(self._h, _d(s), v.ctypes.data_as(_bp)))

    def data_association(self, measures, known_list):
        """known_list: uint8 ndarray of n entries, updated IN PLACE (ekf_slam.cpp:323).
        Returns the landmark each measurement corrected (-1 = dropped)."""
        m = np.ascontiguousarray(measures, dtype=np.float64).reshape(-1, 2)
        if not (isinstance(known_list, np.ndarray) and known_list.dtype == np.uint8
                and known_list.size == self.n and known_list.flags.c_contiguous):
            raise ValueError("known_list must be a contiguous uint8 array of n entries (it is in/out)")
        assoc = np.full(len(m), -1, dtype=np.int32)
        _check(self._lib.ekf_associate(self._h, _d(m), len(m), known_list.ctypes.data_as(_bp),
                                       assoc.ctypes.data_as(_ip)))
        return assoc

    def maha_scores(self, measure, M):
        out = np.empty(int(M))
        _check(self._lib.ekf_maha_scores(self._h, float(measure[0]), float(measure[1]), int(M), _d(out)))
        return out

    def _pose(self):
        out = np.empty(3)
        _check(self._lib.ekf_get_pose(self._h, _d(out)))
        return out

    def getStateX(self):
        return float(self._pose()[1])

    def getStateY(self):
        return float(self._pose()[2])

    def getStateTheta(self):
        return float(self._pose()[0])

    def getStateLandmark(self):
        out = np.empty(2 * self.n)
        _check(self._lib.ekf_get_landmarks(self._h, _d(out)))
        return out

    @property
    def state(self):
        out = np.empty(self.N)
        _check(self._lib.ekf_get_state(self._h, _d(out)))
        return out

    @state.setter
    def state(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
        if v.size != self.N:
            raise ValueError("state must hold N = 3 + 2n values")
        _check(self._lib.ekf_set_state(self._h, _d(v)))

    @property
    def cov(self):
        out = np.empty((self.N, self.N))
        _check(self._lib.ekf_get_cov(self._h, _d(out)))
        return out

    @cov.setter
    def cov(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        if v.shape != (self.N, self.N):
            raise ValueError("cov must be N x N")
        _check(self._lib.ekf_set_cov(self._h, _d(v)))

    @property
    def landmark_init_flag(self):
        f = C.c_int()
        _check(self._lib.ekf_get_init_flag(self._h, C.byref(f)))
        return bool(f.value)

    @landmark_init_flag.setter
    def landmark_init_flag(self, f):
        _check(self._lib.ekf_set_init_flag(self._h, int(bool(f))))

    def sync(self):
        _check(self._lib.ekf_sync(self._h))

    def set_active_set(self, enable=True):
        """Stream only the rows of the touched set in the eager correction (exact; opt-in)."""
        _check(self._lib.ekf_set_active_set(self._h, int(bool(enable))))

    @property
    def forms(self):
        f = C.c_uint()
        _check(self._lib.ekf_get_forms(self._h, C.byref(f)))
        return f.value

    def set_forms(self, forms=FORMS_DEFAULT):
        """which launch structures may be taken (FORM_* bits; test / measurement hook, results do not depend on it)"""
        _check(self._lib.ekf_set_forms(self._h, int(forms)))

    def _form(self, bit, enable):
        self.set_forms(self.forms | bit if enable else self.forms & ~bit)

    def set_small_map_path(self, enable=True):
        self._form(FORM_SMALL_MAP, enable)

    def set_active_prefix(self, enable=True):
        self._form(FORM_ACTIVE_PREFIX, enable)

    def set_call_fused(self, enable=True):
        """measurement() as two launches per call (factor panels + one pass over Sigma); default on, bit-identical"""
        self._form(FORM_CALL_FUSED, enable)

    def set_tuning(self, rows_per_block=0, nontemporal=-1, group_rows=0):
        _check(self._lib.ekf_set_tuning(self._h, rows_per_block, nontemporal, group_rows))

    def phase_trace(self, enable=True, fetch=False):
        """shader-clock stamps of the last two-launch measurement() call: array [2, 64] (row 0 the control wave, row 1
        the first slice wave of workgroup 0), or None"""
        out = np.zeros((2, 64), dtype=np.int64)
        _check(self._lib.ekf_phase_trace(self._h, int(bool(enable)),
                                         out.ctypes.data_as(C.POINTER(C.c_longlong)) if fetch else None))
        return out if fetch else None

    def set_profiling(self, enable=True):
        """HIP-event timing of every covariance-streaming (class 0) and scoring (class 1) launch; resets the sums"""
        _check(self._lib.ekf_set_profiling(self._h, int(bool(enable))))

    def profile(self):
        """{stream_ms, stream_launches, score_ms, score_launches} since set_profiling(True)"""
        ms = (C.c_double * 2)()
        ln = (C.c_longlong * 2)()
        _check(self._lib.ekf_get_profile(self._h, ms, ln))
        return {"stream_ms": ms[0], "stream_launches": ln[0], "score_ms": ms[1], "score_launches": ln[1]}

    def set_update_mode(self, max_pending_corrections=0, symmetric_gather=False):
        """0 = eager covariance stream per correction; k > 0 = delayed rank-2k update (flush every k); symmetric_gather:
        the opt-in symmetric option (Sigma H^T taken as (H Sigma)^T, mirrored flush) -- see ekf_set_update_mode"""
        _check(self._lib.ekf_set_update_mode(self._h, int(max_pending_corrections), int(symmetric_gather)))


class BatchEKF:
    """B independent filters on one GPU, replaying a device-resident known-association log."""

    def __init__(self, B, n, params=None, device=-1):
        self._lib = load()
        self.B, self.n, self.N = int(B), int(n), 3 + 2 * int(n)
        h = C.c_void_p()
        _check(self._lib.ekf_batch_create(self.B, self.n, C.byref(params) if params is not None else None,
                                          device, C.byref(h)))
        self._h = h
        self.T = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ekf_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _check(self._lib.ekf_batch_reset(self._h))

    def device_bytes(self):
        b = C.c_size_t()
        _check(self._lib.ekf_batch_device_bytes(self._h, C.byref(b)))
        return b.value

    def upload_known_log(self, twist, lm_idx, z_xy, init_xy):
        tw = np.ascontiguousarray(twist, dtype=np.float64)
        li = np.ascontiguousarray(lm_idx, dtype=np.int32)
        zz = np.ascontiguousarray(z_xy, dtype=np.float64)
        ii = np.ascontiguousarray(init_xy, dtype=np.float64)
        T, B = tw.shape[0], tw.shape[1]
        if B != self.B or tw.shape != (T, B, 2) or li.shape[:2] != (T, B) or zz.shape != li.shape + (2,) \
                or ii.shape != (B, 2 * self.n):
            raise ValueError("log arrays must be twist[T,B,2], lm_idx[T,B,vmax], z_xy[T,B,vmax,2], init_xy[B,2n]")
        log = KnownLogC(T, li.shape[2], _d(tw), li.ctypes.data_as(_ip), _d(zz), _d(ii))
        _check(self._lib.ekf_batch_upload_known_log(self._h, C.byref(log)))
        self.T, self._vmax = T, li.shape[2]

    def _sim_params(self, cfg):
        sp = SimParams()
        self._lib.ekf_default_sim_params(C.byref(sp))
        sp.seed, sp.first_filter_id = int(cfg.seed), int(cfg.first_filter_id)
        sp.v_cmd, sp.w_cmd, sp.vx_std, sp.the_std = cfg.v_cmd, cfg.w_cmd, cfg.vx_std, cfg.the_std
        sp.slip_min, sp.slip_max, sp.sensor_std, sp.max_visible_dis = cfg.slip_min, cfg.slip_max, cfg.sensor_std, cfg.max_visible_dis
        sp.ticks_per_step = cfg.ticks_per_step
        return sp

    def simulate_unknown_log(self, cfg, world, steps=None, jmax=None, lidar=None):
        """Unknown-association inputs generated ON THE DEVICE: the fake sensor's shuffled readings (lidar=None;
        host twin synth.make_unknown_log) or simulated laser scans pushed through the batched circle fitting
        (lidar = LidarParams; host twins synth.make_scans + the circle checker)."""
        w = np.ascontiguousarray(world, dtype=np.float64)
        if w.shape != (self.n, 2):
            raise ValueError("world must be [n, 2]")
        T = int(cfg.steps if steps is None else steps)
        J = int(cfg.vmax if jmax is None else jmax)
        sp = self._sim_params(cfg)
        _check(self._lib.ekf_batch_simulate_unknown_log(self._h, C.byref(sp), C.byref(lidar) if lidar is not None else None,
                                                        _d(w), T, J))
        self.uT, self._jmax = T, J

    def download_unknown_log(self, want_truth=True):
        T, B, J = self.uT, self.B, self._jmax
        tw, ct, me = np.empty((T, B, 2)), np.empty((T, B), dtype=np.int32), np.empty((T, B, J, 2))
        tp = np.empty((T, B, 3)) if want_truth else None
        _check(self._lib.ekf_batch_download_unknown_log(self._h, _d(tw), ct.ctypes.data_as(_ip), _d(me),
                                                        _d(tp) if want_truth else None))
        return tw, ct, me, tp

    def simulate_known_log(self, cfg, world, steps=None, vmax=None):
        """Generate the log ON THE DEVICE from a synth.SimConfig (same noise model and random-number
        addressing as synth.make_known_log, which stays the host-side twin for the tests)."""
        sp = SimParams()
        self._lib.ekf_default_sim_params(C.byref(sp))
        sp.seed, sp.first_filter_id = int(cfg.seed), int(cfg.first_filter_id)
        sp.v_cmd, sp.w_cmd, sp.vx_std, sp.the_std = cfg.v_cmd, cfg.w_cmd, cfg.vx_std, cfg.the_std
        sp.slip_min, sp.slip_max, sp.sensor_std, sp.max_visible_dis = cfg.slip_min, cfg.slip_max, cfg.sensor_std, cfg.max_visible_dis
        sp.ticks_per_step = cfg.ticks_per_step
        w = np.ascontiguousarray(world, dtype=np.float64)
        if w.shape != (self.n, 2):
            raise ValueError("world must be [n, 2]")
        T = int(cfg.steps if steps is None else steps)
        V = int(cfg.vmax if vmax is None else vmax)
        _check(self._lib.ekf_batch_simulate_known_log(self._h, C.byref(sp), _d(w), T, V))
        self.T, self._vmax = T, V

    def download_log(self, want_truth=True):
        T, B, V, n = self.T, self.B, self._vmax, self.n
        tw, li, zz, ii = np.empty((T, B, 2)), np.empty((T, B, V), dtype=np.int32), np.empty((T, B, V, 2)), np.empty((B, 2 * n))
        tp = np.empty((T, B, 3)) if want_truth else None
        _check(self._lib.ekf_batch_download_log(self._h, _d(tw), li.ctypes.data_as(_ip), _d(zz), _d(ii),
                                                _d(tp) if want_truth else None))
        return tw, li, zz, ii, tp

    def mc_stats(self, t):
        out = np.empty(6)
        _check(self._lib.ekf_batch_mc_stats(self._h, int(t), _d(out)))
        return dict(zip(("nees_mean", "nees_max", "rmse_xy", "rmse_theta", "mean_trace_pose_cov", "frac_nees_below_95pct"), out))

    def run_known(self, t_begin=0, t_end=None, time_kernels=False):
        st = RunStats()
        _check(self._lib.ekf_batch_run_known(self._h, t_begin, self.T if t_end is None else t_end,
                                             int(time_kernels), C.byref(st)))
        return st.as_dict()

    def upload_unknown_log(self, twist, count, meas_xy):
        """twist[T,B,2], count[T,B], meas_xy[T,B,jmax,2]: prediction + data_association per step."""
        tw = np.ascontiguousarray(twist, dtype=np.float64)
        ct = np.ascontiguousarray(count, dtype=np.int32)
        me = np.ascontiguousarray(meas_xy, dtype=np.float64)
        T, B = tw.shape[0], tw.shape[1]
        if B != self.B or tw.shape != (T, B, 2) or ct.shape != (T, B) or me.ndim != 4 or me.shape[:2] != (T, B) \
                or me.shape[3] != 2:
            raise ValueError("log arrays must be twist[T,B,2], count[T,B], meas_xy[T,B,jmax,2]")
        log = UnknownLogC(T, me.shape[2], _d(tw), ct.ctypes.data_as(_ip), _d(me))
        _check(self._lib.ekf_batch_upload_unknown_log(self._h, C.byref(log)))
        self.uT, self._jmax = T, me.shape[2]

    def run_unknown(self, t_begin=0, t_end=None, time_kernels=False):
        st = RunStats()
        _check(self._lib.ekf_batch_run_unknown(self._h, t_begin, self.uT if t_end is None else t_end,
                                               int(time_kernels), C.byref(st)))
        return st.as_dict()

    @property
    def forms(self):
        f = C.c_uint()
        _check(self._lib.ekf_batch_get_forms(self._h, C.byref(f)))
        return f.value

    def set_forms(self, forms=FORMS_DEFAULT):
        """which launch structures may be taken (FORM_* bits; test / measurement hook, results do not depend on it)"""
        _check(self._lib.ekf_batch_set_forms(self._h, int(forms)))

    def _form(self, bit, enable):
        self.set_forms(self.forms | bit if enable else self.forms & ~bit)

    def form_counts(self):
        """covariance passes per form since creation: plain / strip flushes, paired delayed gain launches, call-fused
        passes, per-landmark rank-2 streams, step-fused launches with a separate pass, mirrored flushes, delayed gain
        launches that read the column panel"""
        c = (C.c_longlong * 8)()
        _check(self._lib.ekf_batch_form_counts(self._h, c))
        return dict(zip(("flush_plain", "flush_strip", "gain_pairs", "call_fused_passes", "rank2_streams", "step_split_passes",
                         "flush_mirrored", "gain_from_panel"), (int(x) for x in c)))

    def set_active_prefix(self, enable=True):
        self._form(FORM_ACTIVE_PREFIX, enable)

    def set_small_map_path(self, enable=True):
        self._form(FORM_SMALL_MAP, enable)

    def set_row_packing(self, enable=True):
        """narrow maps: the row-packed rank-2 kernel (default) / the plain kernel"""
        self._form(FORM_ROW_PACKING, enable)

    def set_strip_flush(self, mode="auto"):
        """delayed mode: "auto" (default: beyond 40 pending vectors on pools that fill the chip), "never", "always" """
        f = self.forms & ~(FORM_STRIP_FLUSH | FORM_STRIP_FLUSH_ALWAYS)
        self.set_forms(f | {"auto": FORM_STRIP_FLUSH, "never": 0, "always": FORM_STRIP_FLUSH | FORM_STRIP_FLUSH_ALWAYS}[mode])

    def set_known_counts(self, counts):
        """every filter's known_count (leading run of its known_list) -- the batch twin of the known_list argument"""
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(counts, dtype=np.int32), (self.B,)))
        _check(self._lib.ekf_batch_set_known_counts(self._h, c.ctypes.data_as(_ip)))

    def known_counts(self):
        out = np.empty(self.B, dtype=np.int32)
        _check(self._lib.ekf_batch_get_known_counts(self._h, out.ctypes.data_as(_ip)))
        return out

    def decisions(self):
        out = np.empty((self.uT, self.B, self._jmax), dtype=np.int32)
        _check(self._lib.ekf_batch_get_decisions(self._h, out.ctypes.data_as(_ip)))
        return out

    def state(self, b):
        out = np.empty(self.N)
        _check(self._lib.ekf_batch_get_state(self._h, int(b), _d(out)))
        return out

    def cov(self, b):
        out = np.empty((self.N, self.N))
        _check(self._lib.ekf_batch_get_cov(self._h, int(b), _d(out)))
        return out

    def poses(self):
        out = np.empty((self.B, 3))
        _check(self._lib.ekf_batch_get_poses(self._h, _d(out)))
        return out

    def checksum(self):
        out = np.empty(4)
        _check(self._lib.ekf_batch_checksum(self._h, _d(out)))
        return out

    def set_tuning(self, rows_per_block=0, nontemporal=-1, group_rows=0):
        _check(self._lib.ekf_batch_set_tuning(self._h, rows_per_block, nontemporal, group_rows))

    def set_call_fused(self, enable=True):
        """every measurement() call of the pool as factor panels + ONE pass over Sigma (exact; default on); off = the
        eager per-landmark stream bench.py quotes `value` / `roofline` on"""
        self._form(FORM_CALL_FUSED, enable)

    def set_delayed_pairing(self, enable=True):
        """delayed mode: two consecutive log slots of a step per launch (default) / one launch per landmark"""
        self._form(FORM_DELAYED_PAIR, enable)

    def set_step_fused(self, enable=True):
        """unknown association beyond the LDS-resident path: True / 1 = one launch per step, two for big prefixes
        (default); 2 = always one launch; False / 0 = four launches per measurement slot"""
        e = int(enable)
        f = self.forms & ~(FORM_STEP_FUSED | FORM_STEP_SPLIT_PASS)
        self.set_forms(f | (0 if e == 0 else FORM_STEP_FUSED | (FORM_STEP_SPLIT_PASS if e == 1 else 0)))

    def rank2_kernel(self):
        """name of the k_rank2 instantiation a full-width eager correction of this pool launches, + rows per workgroup"""
        v = [C.c_int() for _ in range(4)]
        _check(self._lib.ekf_batch_rank2_variant(self._h, *[C.byref(x) for x in v]))
        u, nt, tpb, rows = (x.value for x in v)
        res = C.c_int()
        _check(self._lib.ekf_batch_rank2_resident(self._h, C.byref(res)))
        return f"ekf::k_rank2{'_queue' if res.value else ''}<{u},{'true' if nt else 'false'},{tpb}>", rows

    def set_active_set(self, enable=True):
        """Stream only the rows of the touched set in the eager correction (exact; opt-in)."""
        _check(self._lib.ekf_batch_set_active_set(self._h, int(bool(enable))))

    def touched(self):
        """Per filter: how many landmarks have been corrected at least once."""
        out = np.zeros(self.B, dtype=np.int32)
        _check(self._lib.ekf_batch_get_touched(self._h, out.ctypes.data_as(_ip)))
        return out

    def set_update_mode(self, max_pending_corrections=0, symmetric_gather=False):
        """0 = eager covariance stream per correction; k > 0 = delayed rank-2k update (flush every k)."""
        _check(self._lib.ekf_batch_set_update_mode(self._h, int(max_pending_corrections), int(symmetric_gather)))


class DensePropagator:
    """Sigma <- F Sigma F^T + Q for an arbitrary dense F, fp32 on the matrix cores (configs[3]):
    the reference's `sigma = At*sigma*At.t() + Q` (ekf_slam.cpp:101-102) as two dense products."""

    def __init__(self, N, device=-1):
        self._lib = load()
        self.N = int(N)
        h = C.c_void_p()
        _check(self._lib.ekf_dense_create(self.N, device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ekf_dense_destroy(self._h)
            self._h = None

    __del__ = close

    def _f(self, a):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != (self.N, self.N):
            raise ValueError("matrices must be N x N")
        return a, a.ctypes.data_as(_fp)

    def set(self, F=None, Sigma=None, Q=None):
        keep = [self._f(x) for x in (F, Sigma, Q)]
        _check(self._lib.ekf_dense_set(self._h, keep[0][1], keep[1][1], keep[2][1]))

    def propagate(self, iterations=1):
        ms = C.c_double()
        _check(self._lib.ekf_dense_propagate(self._h, int(iterations), C.byref(ms)))
        return ms.value

    @property
    def sigma(self):
        out = np.empty((self.N, self.N), dtype=np.float32)
        _check(self._lib.ekf_dense_get_sigma(self._h, out.ctypes.data_as(_fp)))
        return out

    def launch_info(self):
        """{ld, tiles, n_big, n_tail}: how one product is cut into 256 x 128 tiles and a quarter-tile tail (test hook)"""
        v = [C.c_int() for _ in range(4)]
        _check(self._lib.ekf_dense_launch_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("ld", "tiles", "n_big", "n_tail"), (x.value for x in v)))

    def tile_map(self):
        """bool [tiles][tiles] over the 128 x 128 blocks of the result: True = computed by the tail kernel"""
        t = self.launch_info()["tiles"]
        m = np.zeros(t * t, dtype=np.uint8)
        _check(self._lib.ekf_dense_tile_map(self._h, m.ctypes.data_as(_bp)))
        assert set(np.unique(m)) <= {0, 1}, "a block of the result is computed by no kernel"
        return m.reshape(t, t).astype(bool)


MAX_CLUSTERS = 128


def simulate_scans(poses, world, seed=7, first_filter_id=0, step=0, lidar=None, device=-1):
    """poses [S, 3] = (theta, x, y) -> ranges [S, n_beams] from the on-device lidar simulator (host twin:
    synth.make_scans with fid = first_filter_id + s and the same step)."""
    lib = load()
    sp = SimParams()
    lib.ekf_default_sim_params(C.byref(sp))
    sp.seed, sp.first_filter_id = int(seed), int(first_filter_id)
    lp = lidar if lidar is not None else default_lidar()
    ps = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 3)
    w = np.ascontiguousarray(world, dtype=np.float64).reshape(-1, 2)
    out = np.empty((len(ps), lp.n_beams))
    _check(lib.ekf_simulate_scans(device, C.byref(sp), C.byref(lp), _d(w), len(w), _d(ps), len(ps), int(step), _d(out)))
    return out


def normalize_angles(values, device=-1):
    """rigid2d::normalize_angle (rigid2d.cpp:336-345) evaluated by the device helper the kernels use."""
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    out = np.empty_like(v)
    _check(load().ekf_normalize_angles(device, _d(v), len(v), _d(out)))
    return out


def circle_fit_scans(ranges, max_out=32, device=-1, want_all=False):
    """Batched rigid2d::CircleFitting::approxCirclePositions (circle_fitting.cpp:298-304) on the GPU.
    ranges [S, n_beams] -> list of per-scan centre arrays [k_s, 2] and radii [k_s]
    (+ per-scan [c_s, 4] = x, y, r, is_circle of every cluster when want_all)."""
    r = np.ascontiguousarray(ranges, dtype=np.float64)
    if r.ndim == 1:
        r = r[None, :]
    S, nb = r.shape
    cen = np.zeros((S, max_out, 2))
    rad = np.zeros((S, max_out))
    cnt = np.zeros(S, dtype=np.int32)
    ncl = np.zeros(S, dtype=np.int32)
    allc = np.zeros((S, MAX_CLUSTERS, 4)) if want_all else None
    _check(load().ekf_circle_fit_scans(device, _d(r), S, nb, max_out, _d(cen), _d(rad), cnt.ctypes.data_as(_ip),
                                       _d(allc) if want_all else None, ncl.ctypes.data_as(_ip)))
    centres = [cen[s, :cnt[s]].copy() for s in range(S)]
    radii = [rad[s, :cnt[s]].copy() for s in range(S)]
    if want_all:
        return centres, radii, [allc[s, :ncl[s]].copy() for s in range(S)]
    return centres, radii
