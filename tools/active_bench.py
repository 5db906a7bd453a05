import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekf_slam_ml_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K, W = 32, 4
cfg = synth.config5(filters=B, steps=1 + W + K, n=1000)
bt = capi.BatchEKF(B, 1000)
bt.simulate_known_log(cfg, synth.make_world(1000, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
for on in (0, 1):
    bt.reset(); bt.set_active_set(on)
    bt.run_known(0, 1 + W)
    st = bt.run_known(1 + W, 1 + W + K, time_kernels=True)
    print(f"B={B} active_set={on}: {st['corrections'] / (st['elapsed_ms'] * 1e-3):10.0f} corrections/s, elapsed {st['elapsed_ms']:8.2f} ms, rank2 {st['rank2_ms']:.2f} ms over {st['rank2_launches']} launches, other {st['elapsed_ms'] - st['rank2_ms']:.2f} ms", flush=True)
print(bt.checksum())
