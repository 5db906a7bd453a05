"""-m gpu: delayed rank-2k covariance update -- Sigma kept as
Sigma_base - sum K_j (H Sigma)_j, rewritten once per k corrections.  Must give the eager path's results
to rounding: checked against the CPU checker at the north_star tolerance and against the eager HIP path."""
import numpy as np
import pytest

from ekf_slam_ml_amd import synth
from parity import FP64_TOL, assert_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 3, 16, 64])
def test_single_filter_delayed_vs_oracle(hip, oracle, k):
    steps = 50
    log = synth.make_known_log(synth.config2(steps=steps))
    f, o = hip.EKF_SLAM(200), oracle.OracleEKF(200, oracle.STRUCTURED)
    f.set_update_mode(k)
    for t in range(steps):
        sensor, vis = log.expand_step(t)
        f.prediction(log.twist[t, 0]); o.prediction(*log.twist[t, 0])
        f.measurement(sensor, vis);    o.measurement(sensor, vis)
        if t in (7, 23):  # reads in the middle of a pending window force a flush
            assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, f"k={k} step {t}")
    assert log.corrections > 250
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, f"k={k}")
    f.close()


def test_delayed_then_association_and_back(hip, oracle):
    """Pending factors must be folded in before data_association() scores against Sigma, and the filter
    must keep working when the mode is switched on a live object."""
    n = 20
    log = synth.make_known_log(synth.config1(steps=40))
    f, o = hip.EKF_SLAM(n), oracle.OracleEKF(n, oracle.DENSE)
    f.set_update_mode(8)
    kf, ko = np.ones(n, dtype=np.uint8), np.ones(n, dtype=np.uint8)
    for t in range(40):
        sensor, vis = log.expand_step(t)
        f.prediction(log.twist[t, 0]); o.prediction(*log.twist[t, 0])
        f.measurement(sensor, vis);    o.measurement(sensor, vis)
        if t % 10 == 9:
            m = log.z_xy[t, 0, :2]
            assert np.array_equal(f.data_association(m, kf), o.data_association(m, ko))
            sc = f.maha_scores(m[0], n)
            want = np.array([o.maha(m[0][0], m[0][1], i) for i in range(n)])
            assert np.abs(sc - want).max() / np.abs(want).max() < FP64_TOL
        if t == 20:
            f.set_update_mode(0)   # back to eager in mid-run
        if t == 30:
            f.set_update_mode(5)
    c = f.clone()
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, "mode switches")
    assert np.array_equal(c.cov, f.cov) and np.array_equal(c.state, f.state)
    c.close(); f.close()


def test_batch_delayed_equals_eager(hip, oracle):
    cfg = synth.SimConfig(n=60, steps=24, filters=5, seed=321, half_extent=2.0, min_spacing=0.2,
                          max_visible_dis=0.9, vmax=4)
    log = synth.make_known_log(cfg)
    assert (log.lm_idx >= 0).sum(axis=2).min() < (log.lm_idx >= 0).sum(axis=2).max()  # ragged slots -> zero pairs
    outs = {}
    for k in (0, 2, 7, 64):
        bt = hip.BatchEKF(5, 60)
        bt.set_update_mode(k)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        st = bt.run_known(0, 10)
        st2 = bt.run_known(10, 24, time_kernels=True)
        assert st["corrections"] + st2["corrections"] == log.corrections
        outs[k] = ([bt.state(b) for b in range(5)], [bt.cov(b) for b in range(5)], bt.checksum())
        if k:
            assert 1 <= st2["rank2_launches"] <= 14 * 4 // k + 2  # flushes, not one pass per correction
        bt.close()
    ref_s, ref_c, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
    for k, (ss, cc, cs) in outs.items():
        for b in range(5):
            assert_parity(ss[b], cc[b], ref_s[b], ref_c[b], FP64_TOL, f"k={k} filter {b} vs checker")
            assert_parity(ss[b], cc[b], outs[0][0][b], outs[0][1][b], 1e-11, f"k={k} filter {b} vs eager")


def test_batch_delayed_n1000(hip, oracle):
    log = synth.make_known_log(synth.config5(filters=6, steps=9, n=1000))
    bt = hip.BatchEKF(6, 1000)
    bt.set_update_mode(8)
    bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    st = bt.run_known(0, 9, time_kernels=True)
    assert st["corrections"] == 6 * 8 * 2 and st["rank2_launches"] == 2  # 16 corrections per filter / 8 per flush
    for b in (0, 5):
        o = oracle.OracleEKF(1000, oracle.STRUCTURED)
        for t in range(9):
            o.prediction(*log.twist[t, b])
            o.measurement_compact(log.init_xy[b], log.lm_idx[t, b], log.z_xy[t, b])
        assert_parity(bt.state(b), bt.cov(b), o.state, o.cov, FP64_TOL, f"filter {b}")
    bt.close()


def test_symmetric_gather_option(hip, oracle):
    """Delayed mode with Sigma(c, r) read for Sigma(r, c): still within the north_star tolerance."""
    cfg = synth.SimConfig(n=60, steps=40, filters=4, seed=77, half_extent=2.0, min_spacing=0.2,
                          max_visible_dis=0.9, vmax=4)
    log = synth.make_known_log(cfg)
    bt = hip.BatchEKF(4, 60)
    bt.set_update_mode(16, symmetric_gather=True)
    bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    bt.run_known()
    ref_s, ref_c, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
    for b in range(4):
        assert_parity(bt.state(b), bt.cov(b), ref_s[b], ref_c[b], FP64_TOL, f"filter {b}")
    bt.close()
    f, o = hip.EKF_SLAM(200), oracle.OracleEKF(200, oracle.STRUCTURED)
    f.set_update_mode(8, symmetric_gather=True)
    log1 = synth.make_known_log(synth.config2(steps=40))
    for t in range(40):
        sensor, vis = log1.expand_step(t)
        f.prediction(log1.twist[t, 0]); o.prediction(*log1.twist[t, 0])
        f.measurement(sensor, vis);    o.measurement(sensor, vis)
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, "single filter, symmetric gather")
    f.close()


@pytest.mark.parametrize("B", [8, 11, 17])
def test_flush_grid_decode_for_any_batch_size(hip, B):
    """The flush kernel's XCD-aware 1-D grid (filters dealt to XCDs in groups of 8 + a plain remainder)
    must cover every filter exactly once: delayed == eager for batch sizes around the group size."""
    cfg = synth.SimConfig(n=150, steps=10, filters=B, seed=900 + B, half_extent=3.0, min_spacing=0.2,
                          max_visible_dis=1e9, vmax=3)
    log = synth.make_known_log(cfg)
    res = []
    for k in (0, 6):
        bt = hip.BatchEKF(B, 150)
        bt.set_update_mode(k)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        bt.run_known()
        res.append([(bt.state(b), bt.cov(b)) for b in range(B)])
        bt.close()
    for b in range(B):
        assert_parity(res[1][b][0], res[1][b][1], res[0][b][0], res[0][b][1], 1e-11, f"B={B} filter {b}")


@pytest.mark.parametrize("k", [4, 32])
def test_delayed_data_association_without_flush(hip, oracle, k):
    """data_association() in delayed mode (single filter): Mahalanobis scores against Sigma_base minus the pending pairs
    (ekf_slam.cpp:217-276), decision and landmark initialisation as always, the winner's correction appended to the
    factor store -- no flush in front of the call.  Decisions identical to the checker, state / covariance within 1e-9;
    known-association calls are mixed in so that pending pairs of both kinds coexist."""
    n, T = 120, 40
    cfg = synth.SimConfig(n=n, steps=T, filters=1, seed=2024, half_extent=3.0, min_spacing=0.35, max_visible_dis=1.1, vmax=6)
    ulog = synth.make_unknown_log(cfg)
    f, o = hip.EKF_SLAM(n), oracle.OracleEKF(n, oracle.STRUCTURED)
    f.set_update_mode(k)
    kf, ko = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    total = 0
    for t in range(T):
        J = int(ulog.count[t, 0])
        m = ulog.meas_xy[t, 0, :J]
        f.prediction(ulog.twist[t, 0]); o.prediction(*ulog.twist[t, 0])
        a, b = f.data_association(m, kf), o.data_association(m, ko)
        assert np.array_equal(a, b), f"step {t}: decisions differ"
        assert np.array_equal(kf, ko)
        total += int((a >= 0).sum())
        if t % 7 == 3 and ko.sum() >= 2:   # a known-association call on two discovered landmarks in between
            sensor = np.zeros(2 * n); vis = np.zeros(n, dtype=np.uint8)
            for j in range(min(J, 2)):
                if a[j] >= 0:
                    sensor[2 * a[j]:2 * a[j] + 2] = m[j]; vis[a[j]] = 1
            f.set_init_flag(1) if hasattr(f, "set_init_flag") else None
            o.set_init_flag(1)
            f.landmark_init_flag = True
            f.measurement(sensor, vis); o.measurement(sensor, vis)
    assert total > 60 and ko.sum() >= 6
    assert_parity(f.state, f.cov, o.state, o.cov, FP64_TOL, f"delayed association k={k}")
    f.close()


@pytest.mark.parametrize("B,n,k,vmax", [(3, 130, 5, 3), (9, 500, 16, 2), (4, 1000, 8, 4), (2, 333, 3, 1), (5, 61, 32, 1), (3, 300, 40, 2), (2, 200, 37, 1)])
def test_strip_form_flush_is_bit_identical_to_the_plain_flush(hip, B, n, k, vmax):
    """k_flush_strip (pools that fill the chip: V strip in LDS, 8 rows x 4 columns per lane, 4-vector scalar batches)
    applies the pending pairs to every element in the same order with the same fused multiply-adds as k_flush: forced
    here on small pools (set_strip_flush("always"): EKF_FORM_STRIP_FLUSH_ALWAYS) and compared bit for bit -- strips that end inside the matrix
    (ld/2 not a multiple of 128), N not a multiple of 8 (partial last group), pending counts that are not a
    multiple of 4 (k odd with one correction per step) and ragged counts across filters (zero pairs)."""
    cfg = synth.SimConfig(n=n, steps=2 * k + 3, filters=B, seed=4000 + n, half_extent=4.0, min_spacing=0.15,
                          max_visible_dis=1e9 if vmax == 1 else 2.0, vmax=vmax)
    log = synth.make_known_log(cfg)
    res = []
    for strip, rows in (("always", 0), ("never", 0), ("never", 16)):
        bt = hip.BatchEKF(B, n)
        bt.set_update_mode(k)
        bt.set_strip_flush(strip)
        bt.set_tuning(rows_per_block=rows)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        st = bt.run_known(0, cfg.steps, time_kernels=True)
        assert st["rank2_launches"] >= 2
        fc = bt.form_counts()
        assert (fc["flush_strip"], fc["flush_plain"]) == ((st["rank2_launches"], 0) if strip == "always" else (0, st["rank2_launches"]))
        res.append(([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
        bt.close()
    for other in (1, 2):
        for b in range(B):
            assert np.array_equal(res[0][0][b], res[other][0][b]), f"filter {b} state"
            assert np.array_equal(res[0][1][b], res[other][1][b]), f"filter {b} covariance"
    assert np.all(np.isfinite(res[0][1][0]))


@pytest.mark.parametrize("B,n,k,vmax", [(3, 130, 6, 3), (9, 500, 16, 2), (4, 1000, 32, 2), (2, 333, 3, 1), (3, 300, 40, 2), (2, 200, 37, 1), (11, 160, 64, 2)])
def test_mirrored_flush_of_the_symmetric_option(hip, oracle, B, n, k, vmax):
    """k_flush_sym (symmetric option of set_update_mode, N >= 256): the tiles on and above the diagonal carry the plain
    flush's multiply-adds in the same order -- bit-identical there after ONE flush from the same base -- and every tile
    above the diagonal is written a second time, mirrored: below the diagonal squares the result is the exact mirror
    image.  Ragged shapes: N not a multiple of 32 (partial last row tile), column groups that end inside the matrix,
    pending counts that are not a multiple of 8 (partial last V chunk), pools that are not a multiple of 8 filters.
    The whole run stays within 1e-9 of the CPU checker."""
    T = k // vmax if vmax > 1 else k          # at most k corrections: the run's only flush is the one at its end
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=5000 + n, half_extent=4.0, min_spacing=0.15,
                          max_visible_dis=1e9 if vmax == 1 else 2.0, vmax=vmax)
    log = synth.make_known_log(cfg)
    res = []
    for rows in (0, 16):                      # rows_per_block != 0 pins the plain row-block flush
        bt = hip.BatchEKF(B, n)
        bt.set_update_mode(k, symmetric_gather=True)
        bt.set_tuning(rows_per_block=rows)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        st = bt.run_known(0, T, time_kernels=True)
        fc = bt.form_counts()
        assert st["rank2_launches"] == 1
        assert (fc["flush_mirrored"], fc["flush_plain"] + fc["flush_strip"]) == ((1, 0) if rows == 0 else (0, 1))
        res.append(([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
        bt.close()
    N = 3 + 2 * n
    tile = np.arange(N) // 32
    above = tile[:, None] < tile[None, :]                     # strictly above the diagonal squares (32 x 32)
    on_or_above = tile[:, None] <= tile[None, :]
    ref_s, ref_c, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
    for b in range(B):
        m, p = res[0][1][b], res[1][1][b]
        assert np.array_equal(res[0][0][b], res[1][0][b]), f"filter {b} state"
        assert np.array_equal(m[on_or_above], p[on_or_above]), f"filter {b}: tiles on and above the diagonal"
        assert np.array_equal(m.T[above], m[above]), f"filter {b}: mirror image"
        assert np.all(np.isfinite(m))
        assert_parity(res[0][0][b], m, ref_s[b], ref_c[b], FP64_TOL, f"filter {b} vs the checker")


@pytest.mark.parametrize("blind_tail", [0, 3])
def test_symmetric_option_over_many_flushes(hip, oracle, blind_tail):
    """The symmetric option end to end at n = 1000: paired gain steps that rebuild only the rows Sigma(c, .), predictions
    that keep up rows 1, 2 only (the tiles on and above the diagonal are the covariance between flushes), mirrored
    flushes every 8 steps, 40 steps in two runs: 1e-9 against the CPU checker, and the covariance handed back is symmetric
    to the bit outside the 32 x 32 diagonal squares.  blind_tail: the last steps see no landmark -- predictions behind the
    last flush, whose column entries the run has to restore before it returns."""
    B, n, T = 3, 1000, 40
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=321, half_extent=8.0, min_spacing=0.2, max_visible_dis=2.0, vmax=2)
    log = synth.make_known_log(cfg)
    if blind_tail:
        log.lm_idx[T - blind_tail:] = -1
        log.lm_idx[17:19] = -1           # (and the steps in front of a run boundary)
    bt = hip.BatchEKF(B, n)
    bt.set_update_mode(16, symmetric_gather=True)
    bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    bt.run_known(0, 19)
    c = bt.cov(1)
    assert np.array_equal(c[32:, 1], c[1, 32:]) and np.array_equal(c[32:, 2], c[2, 32:])
    bt.run_known(19, T)
    assert bt.form_counts()["flush_mirrored"] >= 4
    ref_s, ref_c, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
    tile = np.arange(3 + 2 * n) // 32
    above = tile[:, None] < tile[None, :]
    for b in range(B):
        c = bt.cov(b)
        assert_parity(bt.state(b), c, ref_s[b], ref_c[b], FP64_TOL, f"filter {b}")
        assert np.array_equal(c.T[above], c[above])
    bt.close()


@pytest.mark.parametrize("k,vmax", [(32, 2), (7, 5), (3, 4), (2, 3), (1, 2), (16, 1)])
def test_paired_delayed_gain_steps(hip, oracle, k, vmax):
    """Delayed mode, two log slots per launch (k_gain_delayed_pair: the pending factor rows are read once for both
    corrections; the second correction sees the first through the 7 x 7 core block every workgroup carries): against the
    one-launch-per-landmark form at 1e-11, against the CPU checker at 1e-9 -- ragged visible counts (filters whose second
    slot is empty, filters without any), odd slot counts (a trailing single launch), flush boundaries inside a step
    (k = 3, 7), k too small for a pair (k = 1), and one landmark per step (never paired)."""
    B, n, T = 6, 90, 14
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=600 + k + vmax, half_extent=2.5, min_spacing=0.2,
                          max_visible_dis=1.0 if vmax > 1 else 1e9, vmax=vmax)
    log = synth.make_known_log(cfg)
    counts = (log.lm_idx >= 0).sum(axis=2)
    if vmax > 1:
        assert counts.min() < counts.max()
    outs = []
    for pairing in (True, False):
        bt = hip.BatchEKF(B, n)
        bt.set_update_mode(k)
        bt.set_delayed_pairing(pairing)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        bt.run_known(0, 5)
        st = bt.run_known(5, T, time_kernels=True)
        assert st["corrections"] == int(counts[5:].sum())
        outs.append(([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
        bt.close()
    ref_s, ref_c, _ = oracle.batch_run_known(log, oracle.STRUCTURED, want_cov=True, fast=False)
    for b in range(B):
        assert_parity(outs[0][0][b], outs[0][1][b], ref_s[b], ref_c[b], FP64_TOL, f"paired, filter {b} vs checker")
        assert_parity(outs[0][0][b], outs[0][1][b], outs[1][0][b], outs[1][1][b], 1e-11, f"paired vs per landmark, filter {b}")


def test_paired_delayed_gain_steps_random_shapes(hip, oracle):
    """Seeded sweep over pool shapes for the paired gain step: map sizes around the 512-index workgroup slices (n = 254,
    255, 256 -> N = 511, 513, 515), tiny maps (the 7 core indices overlap the whole state), visible counts 0 ... 6, every
    k from 1 to 9 -- paired vs per landmark at 1e-11, and one filter per shape against the CPU checker at 1e-9."""
    rng = np.random.default_rng(2718)
    shapes = [(3, 2), (4, 3), (3, 254), (2, 255), (2, 256)] + [(int(rng.integers(1, 6)), int(rng.integers(4, 140))) for _ in range(7)]
    for idx, (B, n) in enumerate(shapes):
        k = 1 + idx % 9
        vmax = int(rng.integers(1, 7))
        T = 8
        cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=9000 + idx, half_extent=2.5, min_spacing=0.05 if n > 100 else 0.2,
                              max_visible_dis=float(rng.choice([0.8, 1.5, 1e9])), vmax=min(vmax, n))
        log = synth.make_known_log(cfg)
        outs = []
        for pairing in (True, False):
            bt = hip.BatchEKF(B, n)
            bt.set_update_mode(k, symmetric_gather=bool(idx % 5 == 4))
            bt.set_delayed_pairing(pairing)
            bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
            bt.run_known(0, 3); bt.run_known(3, T)
            outs.append(([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)]))
            bt.close()
        for b in range(B):
            assert_parity(outs[0][0][b], outs[0][1][b], outs[1][0][b], outs[1][1][b], 1e-11, f"shape {idx} (B={B}, n={n}, k={k}) filter {b}")
        o = oracle.OracleEKF(n, oracle.STRUCTURED)
        for t in range(T):
            o.prediction(*log.twist[t, 0]); o.measurement_compact(log.init_xy[0], log.lm_idx[t, 0], log.z_xy[t, 0])
        assert_parity(outs[0][0][0], outs[0][1][0], o.state, o.cov, FP64_TOL, f"shape {idx} vs checker")


@pytest.mark.parametrize("symmetric", [False, True])
def test_delayed_at_the_million_steps_configuration(hip, oracle, symmetric):
    """(symmetric: the same run with the opt-in symmetric option -- both flushes mirrored, row-only gain steps.)
    The exact configuration of bench.py's >= 1e6 update steps/s leg (ekf_slam.cpp:178-192 at n = 1000): k = 32
    corrections per flush, the two corrections of a step in one gain launch (k_gain_delayed_pair), and a pool big enough
    (B = 128: 1024 strip workgroups) that launch_flush takes the strip form BY ITSELF at 64 pending vectors -- asserted
    through the form counters, not forced.  17 steps = 34 corrections per filter: one automatic flush at 32 (strip form)
    and the run's closing flush of 2 (plain form); three filters against the structured checker at 1e-9."""
    B, n, k, T = 128, 1000, 32, 18
    log = synth.make_known_log(synth.config5(filters=B, steps=T, n=n))
    assert ((log.lm_idx[1:] >= 0).sum(axis=2) == 2).all()   # V = 2 everywhere: every step is one paired launch
    bt = hip.BatchEKF(B, n)
    bt.set_update_mode(k, symmetric_gather=symmetric)
    assert bt.forms == hip.FORMS_DEFAULT                    # nothing forced
    bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    st = bt.run_known(0, T, time_kernels=True)
    assert st["corrections"] == B * 34 and st["rank2_launches"] == 2
    fc = bt.form_counts()
    if symmetric:
        assert fc["flush_mirrored"] == 2 and fc["flush_strip"] + fc["flush_plain"] == 0 and fc["gain_pairs"] == T - 1, fc
    else:
        assert fc["flush_strip"] == 1 and fc["flush_plain"] == 1 and fc["gain_pairs"] == T - 1, fc
    for b in (0, 61, B - 1):
        o = oracle.OracleEKF(n, oracle.STRUCTURED)
        for t in range(T):
            o.prediction(*log.twist[t, b])
            o.measurement_compact(log.init_xy[b], log.lm_idx[t, b], log.z_xy[t, b])
        assert_parity(bt.state(b), bt.cov(b), o.state, o.cov, FP64_TOL, f"filter {b}")
    bt.close()


@pytest.mark.parametrize("B,n,k,vmax,strip,blind_tail", [
    (5, 130, 8, 2, False, 0),     # pairs, plain flush, several flush periods
    (4, 130, 8, 2, True, 0),      # ... strip-form flush writes / reads the panel
    (3, 200, 5, 3, False, 2),     # pair + trailing single launch per step, flushes inside steps, run ends on predictions
    (3, 140, 16, 1, True, 3),     # single launches only (k_gain_delayed), strip flush, blind tail
    (2, 333, 3, 5, False, 0),     # more landmarks per period than k: some planned, flush boundaries everywhere
    (6, 1000, 32, 2, False, 0),   # the bench shape (n = 1000, k = 32, V = 2)
    (2, 127, 4, 4, True, 1),      # N = 257: just above the panel's minimum dimension
])
def test_column_panel_is_bit_identical(hip, oracle, B, n, k, vmax, strip, blind_tail):
    """EKF_FORM_COLUMN_PANEL (delayed known-association runs): the flush writes columns 0..2 and the columns of the
    landmarks of the next corrections as contiguous panel rows, the gain kernels read Sigma H^T's operands (ekf_slam.cpp:178)
    from there, prediction() keeps columns 0..2 in the panel and leaves the matrix's to the next flush.  Same values from
    another address: state and covariance must be BIT-identical to the run without the panel -- with the plan holding every
    landmark of a period, with the one-slot plan (panel rows and matrix gathers mixed in one launch), across run boundaries
    that hand the panel over, across boundaries that drop it (a getter in between), and when a run ends on predictions with
    nothing pending (the repair).  One filter against the CPU checker at 1e-9."""
    T = 19
    cfg = synth.SimConfig(n=n, steps=T, filters=B, seed=4100 + n + k, half_extent=3.0, min_spacing=0.1,
                          max_visible_dis=1.1 if vmax > 2 else 1e9, vmax=vmax, v_cmd=0.8, w_cmd=0.5)
    log = synth.make_known_log(cfg)
    lm = log.lm_idx.copy()
    if blind_tail:
        lm[-blind_tail:] = -1          # the run ends on predictions: nothing pending, matrix columns 1, 2 behind
    lm[7, 0] = -1                      # one filter sits a step out
    base = hip.FORMS_DEFAULT | (hip.FORM_STRIP_FLUSH_ALWAYS if strip else 0)
    nocur = base & ~hip.FORM_CURRENT_COLUMNS
    # "current" / "current_one_slot": the panel WITH the kept current rows / columns (EKF_FORM_CURRENT_COLUMNS, the default):
    # the same sums in another association -- compared at 1e-10
    variants = {"panel": nocur, "one_slot": nocur | hip.FORM_COLUMN_PANEL_ONE_SLOT, "off": base & ~hip.FORM_COLUMN_PANEL,
                "current": base, "current_one_slot": base | hip.FORM_COLUMN_PANEL_ONE_SLOT}
    outs, counts = {}, {}
    for name, forms in variants.items():
        for cuts in ((T,), (4, 5, 11, T)):          # one run; four consecutive runs (the panel is handed over)
            bt = hip.BatchEKF(B, n)
            bt.set_forms(forms)
            bt.set_update_mode(k)
            bt.upload_known_log(log.twist, lm, log.z_xy, log.init_xy)
            t0 = 0
            for t1 in cuts:
                bt.run_known(t0, t1)
                t0 = t1
            counts[name, cuts] = bt.form_counts()
            outs[name, cuts] = ([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)])
            bt.close()
    # a getter between two runs drops the panel: the next run starts without one (still the same numbers)
    for name in ("panel", "current"):
        bt = hip.BatchEKF(B, n)
        bt.set_forms(variants[name])
        bt.set_update_mode(k)
        bt.upload_known_log(log.twist, lm, log.z_xy, log.init_xy)
        bt.run_known(0, 9)
        mid = bt.state(0)
        bt.run_known(9, T)
        outs[name, "getter"] = ([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)])
        bt.close()
    # (where the flushes fall changes the delayed mode's rounding -- its reconstruction and its flush contract their
    # multiply-adds -- so every variant is compared with the panel-less run over the SAME run boundaries)
    bt = hip.BatchEKF(B, n)
    bt.set_forms(variants["off"])
    bt.set_update_mode(k)
    bt.upload_known_log(log.twist, lm, log.z_xy, log.init_xy)
    bt.run_known(0, 9); bt.run_known(9, T)
    outs["off", "getter"] = ([bt.state(b) for b in range(B)], [bt.cov(b) for b in range(B)])
    bt.close()
    for (name, cuts), (st, cv) in outs.items():
        ref = outs["off", cuts]
        for b in range(B):
            if name.startswith("current"):
                # (1e-10: on the n = 1000 shape the rebuilt form itself sits 1.3e-10 from the eager run, this form 1.2e-10)
                assert np.abs(st[b] - ref[0][b]).max() < 1e-10, f"{name} {cuts}, filter {b}"
                assert np.abs(cv[b] - ref[1][b]).max() / np.abs(ref[1][b]).max() < 1e-10, f"{name} {cuts}, filter {b}"
            else:
                assert np.array_equal(st[b], ref[0][b]) and np.array_equal(cv[b], ref[1][b]), f"{name} {cuts}, filter {b}"
    assert counts["off", (T,)]["gain_from_panel"] == 0
    for name in ("panel", "one_slot", "current"):
        c1, c4 = counts[name, (T,)], counts[name, (4, 5, 11, T)]
        assert c4["gain_from_panel"] > 0, c4          # a closing flush hands the panel to the next run
        if c1["flush_plain"] + c1["flush_strip"] >= 2:  # (a single run has a panel from its first flush on)
            assert c1["gain_from_panel"] > 0, c1
        assert c4["gain_from_panel"] >= c1["gain_from_panel"]
    o = oracle.OracleEKF(n, oracle.STRUCTURED)
    b = B - 1
    for t in range(T):
        o.prediction(*log.twist[t, b]); o.measurement_compact(log.init_xy[b], lm[t, b], log.z_xy[t, b])
    assert_parity(outs["panel", (T,)][0][b], outs["panel", (T,)][1][b], o.state, o.cov, FP64_TOL, "panel vs checker")
    assert_parity(outs["current", (T,)][0][b], outs["current", (T,)][1][b], o.state, o.cov, FP64_TOL, "current columns vs checker")
    assert mid.shape == (3 + 2 * n,)


def test_column_panel_at_the_million_steps_configuration(hip):
    """bench.py's delayed leg as the driver runs it: a warm-up run, then the timed run of whole flush periods on the same
    handle with nothing in between -- the warm-up's closing flush writes the panel for the timed run's first corrections,
    so EVERY gain launch of the timed run reads the panel (asserted), and the result is bit-identical to the run with the
    panel off (B = 128 so that the strip-form flush is taken by itself, as at B = 4096)."""
    B, n, k, W, K = 128, 1000, 32, 5, 32
    T = 1 + W + K
    log = synth.make_known_log(synth.config5(filters=B, steps=T, n=n))
    res = []
    for forms in (hip.FORMS_DEFAULT & ~hip.FORM_CURRENT_COLUMNS, hip.FORMS_DEFAULT & ~hip.FORM_COLUMN_PANEL, hip.FORMS_DEFAULT):
        bt = hip.BatchEKF(B, n)
        bt.set_forms(forms)
        bt.set_update_mode(k)
        bt.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
        bt.run_known(0, 1 + W)
        before = bt.form_counts()
        bt.run_known(1 + W, T)
        after = bt.form_counts()
        res.append((after["gain_from_panel"] - before["gain_from_panel"], after["gain_pairs"] - before["gain_pairs"],
                    after["flush_strip"] - before["flush_strip"], [bt.state(b) for b in (0, 77, B - 1)], bt.cov(B - 1)))
        bt.close()
    assert res[0][0] == res[0][1] == K and res[0][2] == 2, res[0][:3]     # every timed gain launch read the panel
    assert res[1][0] == 0 and res[1][1] == K
    for a, b in zip(res[0][3], res[1][3]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][4], res[1][4])
    # the default adds the kept current rows / columns: the same launches, results equal to rounding
    assert res[2][0] == res[2][1] == K and res[2][2] == 2
    for a, b in zip(res[2][3], res[1][3]):
        assert np.abs(a - b).max() < 1e-10
    assert np.abs(res[2][4] - res[1][4]).max() / np.abs(res[1][4]).max() < 1e-10
