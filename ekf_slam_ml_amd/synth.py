"""Deterministic synthetic odometry + landmark logs for the EKF-SLAM hot path.

Host-side input generator (not part of the filter).  It reproduces the value
distributions of the reference's simulator so that the filter sees inputs of
the shape the nuslam nodes feed it:

* wheel slip ~U(slip_min, slip_max) on each 100 Hz wheel delta and Gaussian
  noise on the commanded twist (nurtlesim/src/tube_world.cpp:191-227,
  nurtlesim/config/noise_param.yaml:2-7);
* the odometry twist handed to ``prediction()`` is the LAST 100 Hz wheel delta
  scaled x10 through DiffDrive::getBodyTwistForUpdate (nuslam/src/slam.cpp:173-176,
  rigid2d/src/diff_drive.cpp:38-47);
* landmark readings are robot-frame (x, y) + N(0, covar_sensor) per axis, visible
  iff the true range <= max_visible_dis (tube_world.cpp:369-414,
  noise_param.yaml:8-10);
* the first ``measurement()`` call receives all n readings with ``visible`` all
  false (nuslam/src/slam.cpp:305-333: state_update_flag is still false).

The reference seeds std::mt19937 from std::random_device (tube_world.cpp:184-189),
so its runs are not reproducible; here every random number is a pure function of
(seed, filter id, step, kind, k) through splitmix64 + Box-Muller, bit-identical
on every box.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# rigid2d/config/fake_turtle_param.yaml:6-7
WHEEL_BASE = 0.16
WHEEL_RADIUS = 0.033
# nurtlesim/config/tube_param.yaml:2-3
TUBE_X = [0.5, 0.7, 0.7, 0.2, 0.6, -0.3, -0.7, -0.3, -0.7, 0.0]
TUBE_Y = [0.1, 0.7, 0.4, -0.3, -0.8, -0.6, -0.2, 0.5, 0.7, 1.1]

_U64 = np.uint64
_GOLDEN = _U64(0x9E3779B97F4A7C15)
_M1 = _U64(0xBF58476D1CE4E5B9)
_M2 = _U64(0x94D049BB133111EB)

KIND_CMD, KIND_SLIP, KIND_SENSOR, KIND_WORLD, KIND_SHUFFLE = 1, 2, 3, 4, 5


def splitmix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    x = np.asarray(x, dtype=_U64)
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> _U64(30))) * _M1
        z = (z ^ (z >> _U64(27))) * _M2
        return z ^ (z >> _U64(31))


def _key(seed, fid, step, kind, k):
    with np.errstate(over="ignore"):
        a = splitmix64(_U64(seed) ^ (np.asarray(fid, dtype=_U64) * _U64(0xD1B54A32D192ED03)))
        b = splitmix64(a + np.asarray(step, dtype=_U64) * _U64(0x8CB92BA72F3D8DD7))
        c = splitmix64(b + _U64(kind) * _U64(0xABC98388FB8FAC03))
        return splitmix64(c + np.asarray(k, dtype=_U64))


def uniform01(seed, fid, step, kind, k):
    """U[0,1) with 53 random bits."""
    return (_key(seed, fid, step, kind, k) >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal01(seed, fid, step, kind, k):
    """Standard normal by Box-Muller on two counter-addressed uniforms."""
    k = np.asarray(k, dtype=_U64)
    u1 = uniform01(seed, fid, step, kind, k * _U64(2))
    u2 = uniform01(seed, fid, step, kind, k * _U64(2) + _U64(1))
    u1 = np.maximum(u1, 2.0 ** -53)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_world(n, half_extent, min_spacing, seed, use_reference_tubes=True):
    """n landmark positions in [-half_extent, half_extent]^2 with a minimum spacing.

    The first 10 are the reference's tubes (tube_param.yaml:2-3) when they fit."""
    pts = []
    if use_reference_tubes and half_extent >= 1.2:
        for x, y in zip(TUBE_X, TUBE_Y):
            if len(pts) < n:
                pts.append((x, y))
    k = 0
    while len(pts) < n:
        cx = (uniform01(seed, 0, 0, KIND_WORLD, 2 * k) * 2.0 - 1.0) * half_extent
        cy = (uniform01(seed, 0, 0, KIND_WORLD, 2 * k + 1) * 2.0 - 1.0) * half_extent
        k += 1
        if k > 200 * n + 10000:
            raise ValueError("world too dense for the requested spacing")
        p = np.array(pts) if pts else np.zeros((0, 2))
        if len(pts) == 0 or np.min((p[:, 0] - cx) ** 2 + (p[:, 1] - cy) ** 2) >= min_spacing ** 2:
            pts.append((float(cx), float(cy)))
    return np.array(pts, dtype=np.float64)


def body_twist(left, right, wheel_base=WHEEL_BASE, wheel_radius=WHEEL_RADIUS):
    """DiffDrive::getBodyTwistForUpdate, rigid2d/src/diff_drive.cpp:38-47."""
    D = wheel_base * 0.5
    r = wheel_radius
    return (r / (2.0 * D)) * (right - left), (r / 2.0) * (right + left)


def wheel_velocity(ang, vx, wheel_base=WHEEL_BASE, wheel_radius=WHEEL_RADIUS):
    """DiffDrive::calculateWheelVelocity, rigid2d/src/diff_drive.cpp:24-36."""
    D = wheel_base * 0.5
    r = wheel_radius
    return -(D / r) * ang + (1.0 / r) * vx, (D / r) * ang + (1.0 / r) * vx


def _integrate(theta, x, y, dth, dx):
    """True-pose arc integration of a body twist (dth, dx) over one tick."""
    small = np.abs(dth) < 1e-9
    dth_s = np.where(small, 1.0, dth)
    rad = dx / dth_s
    nx = np.where(small, x + dx * np.cos(theta), x - rad * np.sin(theta) + rad * np.sin(theta + dth))
    ny = np.where(small, y + dx * np.sin(theta), y + rad * np.cos(theta) - rad * np.cos(theta + dth))
    return theta + dth, nx, ny


@dataclass
class SimConfig:
    n: int = 20
    steps: int = 1000                 # T filter steps (10 Hz)
    filters: int = 1                  # B independent Monte-Carlo runs
    seed: int = 20211023
    first_filter_id: int = 0          # global id of filter 0 (multi-GPU sharding)
    half_extent: float = 1.5
    min_spacing: float = 0.25
    v_cmd: float = 0.1                # commanded circle
    w_cmd: float = 0.4
    vx_std: float = 0.01              # noise_param.yaml:3,5
    the_std: float = 0.01
    slip_min: float = 0.90            # noise_param.yaml:6-7
    slip_max: float = 1.10
    sensor_std: float = 0.005         # noise_param.yaml:8-9
    max_visible_dis: float = 0.7      # noise_param.yaml:10
    vmax: int = 8                     # cap on readings per step (nearest first)
    ticks_per_step: int = 10          # 100 Hz sim under a 10 Hz filter
    world_seed: int | None = None     # defaults to seed (shared world across filters)


@dataclass
class KnownLog:
    """Compact known-association log: visible readings only, ascending landmark index.

    twist[T,B,2] = (dtheta, dx) fed to prediction(); lm_idx[T,B,vmax] (int32, -1 pads);
    z_xy[T,B,vmax,2] robot-frame readings; init_xy[B,2n] = the sensor_reading vector of
    the first measurement() call (all n tubes, ekf_slam.cpp:113-128)."""
    cfg: SimConfig
    world: np.ndarray
    twist: np.ndarray
    lm_idx: np.ndarray
    z_xy: np.ndarray
    init_xy: np.ndarray
    true_pose: np.ndarray = field(repr=False, default=None)
    wheel: np.ndarray = field(repr=False, default=None)  # [T,B,2] last 100 Hz wheel deltas (joint_states velocity)

    @property
    def corrections(self):
        return int((self.lm_idx >= 0).sum())

    def expand_step(self, t, b=0):
        """(sensor_reading[2n], visible[n]) as SLAM::callback_fake_sensor builds them
        (nuslam/src/slam.cpp:305-333)."""
        n = self.cfg.n
        if t == 0:
            return self.init_xy[b].copy(), np.zeros(n, dtype=np.uint8)
        sensor = np.zeros(2 * n)
        vis = np.zeros(n, dtype=np.uint8)
        for v in range(self.lm_idx.shape[2]):
            i = int(self.lm_idx[t, b, v])
            if i < 0:
                break
            sensor[2 * i:2 * i + 2] = self.z_xy[t, b, v]
            vis[i] = 1
        return sensor, vis


@dataclass
class UnknownLog:
    """Unknown-association log: per step J <= jmax robot-frame readings in shuffled order
    (the scan_measures vector of nuslam/src/unknown_data_assoc.cpp:309-320)."""
    cfg: SimConfig
    world: np.ndarray
    twist: np.ndarray      # [T,B,2]
    count: np.ndarray      # [T,B] int32
    meas_xy: np.ndarray    # [T,B,jmax,2]
    truth_idx: np.ndarray  # [T,B,jmax] int32 generating landmark (diagnostics only)
    true_pose: np.ndarray = field(repr=False, default=None)
    wheel: np.ndarray = field(repr=False, default=None)  # [T,B,2] last 100 Hz wheel deltas


def _simulate(cfg: SimConfig):
    B, T, n = cfg.filters, cfg.steps, cfg.n
    wseed = cfg.seed if cfg.world_seed is None else cfg.world_seed
    world = make_world(n, cfg.half_extent, cfg.min_spacing, wseed)
    fid = np.arange(B, dtype=np.uint64) + np.uint64(cfg.first_filter_id)
    theta = np.zeros(B)
    x = np.zeros(B)
    y = np.zeros(B)
    twist = np.zeros((T, B, 2))
    wheel = np.zeros((T, B, 2))
    true_pose = np.zeros((T, B, 3))
    for t in range(T):
        # callback_vel: noise only on non-zero commands (tube_world.cpp:191-208)
        xv = cfg.v_cmd + (cfg.vx_std * normal01(cfg.seed, fid, t, KIND_CMD, 0) if abs(cfg.v_cmd) >= 1e-4 else 0.0)
        av = cfg.w_cmd + (cfg.the_std * normal01(cfg.seed, fid, t, KIND_CMD, 1) if abs(cfg.w_cmd) >= 1e-4 else 0.0)
        wl, wr = wheel_velocity(av, xv)
        dl = dr = None
        for k in range(cfg.ticks_per_step):
            sl = cfg.slip_min + (cfg.slip_max - cfg.slip_min) * uniform01(cfg.seed, fid, t, KIND_SLIP, 2 * k)
            sr = cfg.slip_min + (cfg.slip_max - cfg.slip_min) * uniform01(cfg.seed, fid, t, KIND_SLIP, 2 * k + 1)
            dl = (wl / 100.0) * sl
            dr = (wr / 100.0) * sr
            dth, ddx = body_twist(dl, dr)
            theta, x, y = _integrate(theta, x, y, dth, ddx)
        # Odometer::getCurrentTwist: last 100 Hz delta x10 (slam.cpp:173-176)
        tw_a, tw_x = body_twist(dl * 10.0, dr * 10.0)
        twist[t, :, 0] = tw_a
        twist[t, :, 1] = tw_x
        wheel[t, :, 0] = dl
        wheel[t, :, 1] = dr
        true_pose[t, :, 0], true_pose[t, :, 1], true_pose[t, :, 2] = theta, x, y
    return world, twist, true_pose, fid, wheel


def _robot_frame(world, pose):
    """True robot-frame coordinates of every landmark; pose [B,3] -> [B,n,2]."""
    th, px, py = pose[:, 0:1], pose[:, 1:2], pose[:, 2:3]
    dx = world[None, :, 0] - px
    dy = world[None, :, 1] - py
    c, s = np.cos(th), np.sin(th)
    return np.stack([c * dx + s * dy, -s * dx + c * dy], axis=-1)


def make_known_log(cfg: SimConfig) -> KnownLog:
    B, T, n, vmax = cfg.filters, cfg.steps, cfg.n, cfg.vmax
    world, twist, true_pose, fid, wheel = _simulate(cfg)
    lm_idx = np.full((T, B, vmax), -1, dtype=np.int32)
    z_xy = np.zeros((T, B, vmax, 2))
    init_xy = np.zeros((B, 2 * n))
    lm = np.arange(n, dtype=np.uint64)
    for t in range(T):
        rf = _robot_frame(world, true_pose[t])                       # [B,n,2]
        if t == 0:
            nx = cfg.sensor_std * normal01(cfg.seed, fid[:, None], t, KIND_SENSOR, lm[None, :] * np.uint64(2))
            ny = cfg.sensor_std * normal01(cfg.seed, fid[:, None], t, KIND_SENSOR, lm[None, :] * np.uint64(2) + np.uint64(1))
            init_xy[:, 0::2] = rf[:, :, 0] + nx
            init_xy[:, 1::2] = rf[:, :, 1] + ny
            continue
        rng2 = rf[:, :, 0] ** 2 + rf[:, :, 1] ** 2
        k = min(vmax, n)
        near = np.argpartition(rng2, k - 1, axis=1)[:, :k] if k < n else np.tile(np.arange(n), (B, 1))
        near.sort(axis=1)                                            # ascending landmark index
        d2 = np.take_along_axis(rng2, near, axis=1)
        ok = d2 <= cfg.max_visible_dis ** 2
        nu = near.astype(np.uint64)
        nx = cfg.sensor_std * normal01(cfg.seed, fid[:, None], t, KIND_SENSOR, nu * np.uint64(2))
        ny = cfg.sensor_std * normal01(cfg.seed, fid[:, None], t, KIND_SENSOR, nu * np.uint64(2) + np.uint64(1))
        zx = np.take_along_axis(rf[:, :, 0], near, axis=1) + nx
        zy = np.take_along_axis(rf[:, :, 1], near, axis=1) + ny
        for b in range(B) if B <= 64 else ():
            sel = np.nonzero(ok[b])[0]
            lm_idx[t, b, :len(sel)] = near[b, sel]
            z_xy[t, b, :len(sel), 0] = zx[b, sel]
            z_xy[t, b, :len(sel), 1] = zy[b, sel]
        if B > 64:
            # vectorised compaction: stable-sort visible entries to the front
            order = np.argsort(~ok, axis=1, kind="stable")
            cnt = ok.sum(axis=1)
            keep = np.arange(k)[None, :] < cnt[:, None]
            lm_idx[t, :, :k] = np.where(keep, np.take_along_axis(near, order, axis=1), -1)
            z_xy[t, :, :k, 0] = np.where(keep, np.take_along_axis(zx, order, axis=1), 0.0)
            z_xy[t, :, :k, 1] = np.where(keep, np.take_along_axis(zy, order, axis=1), 0.0)
    return KnownLog(cfg, world, twist, lm_idx, z_xy, init_xy, true_pose, wheel)


def make_unknown_log(cfg: SimConfig) -> UnknownLog:
    """Every step (including step 0) carries up to vmax shuffled readings of the
    landmarks within max_visible_dis; the filter discovers landmarks in that order."""
    B, T, n, jmax = cfg.filters, cfg.steps, cfg.n, cfg.vmax
    world, twist, true_pose, fid, wheel = _simulate(cfg)
    count = np.zeros((T, B), dtype=np.int32)
    meas = np.zeros((T, B, jmax, 2))
    truth = np.full((T, B, jmax), -1, dtype=np.int32)
    for t in range(T):
        rf = _robot_frame(world, true_pose[t])
        rng2 = rf[:, :, 0] ** 2 + rf[:, :, 1] ** 2
        for b in range(B):
            vis = np.nonzero(rng2[b] <= cfg.max_visible_dis ** 2)[0]
            if len(vis) > jmax:
                vis = vis[np.argsort(rng2[b, vis], kind="stable")[:jmax]]
            key = uniform01(cfg.seed, fid[b], t, KIND_SHUFFLE, vis.astype(np.uint64))
            vis = vis[np.argsort(key, kind="stable")]
            vu = vis.astype(np.uint64)
            nx = cfg.sensor_std * normal01(cfg.seed, fid[b], t, KIND_SENSOR, vu * np.uint64(2))
            ny = cfg.sensor_std * normal01(cfg.seed, fid[b], t, KIND_SENSOR, vu * np.uint64(2) + np.uint64(1))
            J = len(vis)
            count[t, b] = J
            meas[t, b, :J, 0] = rf[b, vis, 0] + nx
            meas[t, b, :J, 1] = rf[b, vis, 1] + ny
            truth[t, b, :J] = vis
    return UnknownLog(cfg, world, twist, count, meas, truth, true_pose, wheel)


# ---- the BASELINE.json configurations -------------------------------

def config1(steps=1000):
    """configs[0]: n = 20 known association, 1000 steps, seed 20211023."""
    return SimConfig(n=20, steps=steps, filters=1, seed=20211023, half_extent=1.5, min_spacing=0.25,
                     max_visible_dis=0.7, vmax=20)


def config2(steps=2000):
    """configs[1]: n = 200 known association, world [-5,5]^2, V ~ 8 per step, seed 2."""
    return SimConfig(n=200, steps=steps, filters=1, seed=2, half_extent=5.0, min_spacing=0.3,
                     v_cmd=0.3, w_cmd=0.1, max_visible_dis=1.15, vmax=8)


def config3(steps=2000):
    """configs[2]: n = 1000 unknown association, world [-12,12]^2, spacing >= 0.6, J ~ 8, seed 3."""
    return SimConfig(n=1000, steps=steps, filters=1, seed=3, half_extent=12.0, min_spacing=0.6,
                     v_cmd=0.5, w_cmd=0.06, max_visible_dis=1.3, vmax=8)


def config5(filters=4096, steps=20, first_filter_id=0, n=1000):
    """configs[4], one GPU's share: B filters, n = 1000 known association, one shared world,
    per-filter noise streams 5e6 + global filter id, exactly V = 2 readings per step."""
    return SimConfig(n=n, steps=steps, filters=filters, seed=5_000_000, first_filter_id=first_filter_id,
                     half_extent=12.0, min_spacing=0.6, v_cmd=0.5, w_cmd=0.06,
                     max_visible_dis=1.0e9, vmax=2, world_seed=5)


# ---- simulated 360-beam laser scans (input of the circle-fitting front end) --------
# tube_world.cpp:454-577 (publishScan): beam i at angle 2*pi*i/360 in the robot frame, nearest hit of the
# square world border (world_border_width 2.0, tube_param.yaml:5) or of a tube (radius 0.0762,
# tube_param.yaml:4), capped at range_max 3.5, plus N(0, range_std = 0.005) (noise_param.yaml:11).
TUBE_RADIUS = 0.0762
WORLD_BORDER_WIDTH = 2.0
KIND_SCAN = 6


def _normalize_angle(a):
    """rigid2d::normalize_angle (rigid2d.cpp:336-345): double fmod, range (-pi, pi]"""
    two_pi = 2.0 * np.pi
    a = np.fmod(np.fmod(a, two_pi) + two_pi, two_pi)
    return np.where(a > np.pi, a - two_pi, a)


def _scans_reference_model(poses, world, n_beams, range_max, border, tube_radius, range_min):
    """publishScan's own procedure, step for step (nurtlesim/src/tube_world.cpp:496-570), noise-free ranges [S, n_beams]."""
    S = len(poses)
    res = 2.0 * np.pi / n_beams                                    # :462
    i = np.arange(n_beams, dtype=np.float64)
    theta, px, py = poses[:, 0:1], poses[:, 1:2], poses[:, 2:3]
    window = 2.0 * np.arctan2(tube_radius, range_min)              # :480 largest_tube_scan_theta
    curr = np.broadcast_to(_normalize_angle(res * i)[None, :], (S, n_beams))   # :496
    x2, y2 = range_max * np.cos(curr), range_max * np.sin(curr)    # :497-498 (min_r = 3.5 there)
    x_dis, y_dis = border / 2.0 - px, border / 2.0 - py            # :460-461
    box = _normalize_angle(res * i[None, :] + theta)               # :502
    y_t = np.where(box < 0, -(border - y_dis), y_dis)              # :503-505
    x_t = np.where((box > np.pi / 2.0) | (box < -np.pi / 2.0), -(border - x_dis), x_dis)   # :507-509
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.minimum(x_t / np.cos(box), y_t / np.sin(box))      # :512
    min_r = np.minimum(r, range_max)                               # :516
    c, s_ = np.cos(theta), np.sin(theta)
    for (wx, wy) in world:                                         # :518-566
        ex, ey = wx - px, wy - py
        tx, ty = c * ex + s_ * ey, -s_ * ex + c * ey               # Ttw(world_tube), :520-521
        tb = np.arctan2(ty, tx)                                    # :524
        start, end = _normalize_angle(tb - window / 2.0), _normalize_angle(tb + window / 2.0)   # :526-527
        inside = (curr > start) & (curr < end)
        flag = np.where((start > 0) & (end < 0), (curr > start) | (curr < end), inside)        # :530-552
        # getLineCircleIntersection (:420-450) in the tube's frame: turtle at (-tx, -ty), beam end at (x2 - tx, y2 - ty)
        x1, y1 = -tx, -ty
        xb, yb = x2 - tx, y2 - ty
        dx, dy = xb - x1, yb - y1
        dr = np.sqrt(dx ** 2 + dy ** 2)
        D = x1 * yb - xb * y1
        delta = tube_radius ** 2 * dr ** 2 - D ** 2
        hit = flag & (delta > 0)
        sq = np.sqrt(np.where(hit, delta, 0.0))
        sgn = np.where(dy < 0, -1.0, 1.0)
        ix1, iy1 = (D * dy + sgn * dx * sq) / dr ** 2, (-D * dx + np.abs(dy) * sq) / dr ** 2
        ix2, iy2 = (D * dy - sgn * dx * sq) / dr ** 2, (-D * dx - np.abs(dy) * sq) / dr ** 2
        d1 = np.sqrt((x1 - ix1) ** 2 + (y1 - iy1) ** 2)
        d2 = np.sqrt((x1 - ix2) ** 2 + (y1 - iy2) ** 2)
        min_r = np.where(hit, np.minimum(np.minimum(d1, d2), min_r), min_r)   # :563-564
    return min_r


def make_scans(poses, world=None, n_beams=360, seed=7, range_std=0.005, range_max=3.5,
               border=WORLD_BORDER_WIDTH, tube_radius=TUBE_RADIUS, fid=None, step=0, model=0, range_min=0.12):
    """poses [S, 3] = (theta, x, y) -> ranges [S, n_beams] (float64), deterministic in (seed, fid, step, beam);
    fid defaults to the scan id.  Device twin: k_sim_scans (ekf_sim.hip).  model 0: clean ray geometry; model 1:
    publishScan's own bearing-window + line-circle procedure (ekf_lidar_params.model, include/ekfslam.h)."""
    poses = np.asarray(poses, dtype=np.float64).reshape(-1, 3)
    if world is None:
        world = np.stack([TUBE_X, TUBE_Y], axis=1)
    S = len(poses)
    if model == 1:
        r = _scans_reference_model(poses, np.asarray(world, dtype=np.float64), n_beams, range_max, border, tube_radius, range_min)
        sid = (np.arange(S, dtype=np.uint64) if fid is None else np.asarray(fid, dtype=np.uint64).reshape(S))[:, None]
        stp = np.broadcast_to(np.asarray(step, dtype=np.uint64).reshape(-1, 1), (S, 1))
        return r + range_std * normal01(seed, sid, stp, KIND_SCAN, np.arange(n_beams, dtype=np.uint64)[None, :])
    ang = 2.0 * np.pi * np.arange(n_beams) / n_beams
    th = poses[:, 0:1] + ang[None, :]                       # world-frame beam direction [S, nb]
    dx, dy = np.cos(th), np.sin(th)
    ox, oy = poses[:, 1:2], poses[:, 2:3]
    half = border / 2.0
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(dx > 0, (half - ox) / dx, np.where(dx < 0, (-half - ox) / dx, np.inf))
        ty = np.where(dy > 0, (half - oy) / dy, np.where(dy < 0, (-half - oy) / dy, np.inf))
    r = np.minimum(np.minimum(tx, ty), range_max)
    for (cx, cy) in world:                                  # ray-circle intersection, nearest root
        fx, fy = ox - cx, oy - cy
        bq = fx * dx + fy * dy
        cq = fx * fx + fy * fy - tube_radius ** 2
        disc = bq * bq - cq
        hit = disc > 0
        t = -bq - np.sqrt(np.where(hit, disc, 0.0))
        r = np.where(hit & (t > 0) & (t < r), t, r)
    sid = (np.arange(S, dtype=np.uint64) if fid is None else np.asarray(fid, dtype=np.uint64).reshape(S))[:, None]
    stp = np.broadcast_to(np.asarray(step, dtype=np.uint64).reshape(-1, 1), (S, 1))
    noise = range_std * normal01(seed, sid, stp, KIND_SCAN, np.arange(n_beams, dtype=np.uint64)[None, :])
    return r + noise
