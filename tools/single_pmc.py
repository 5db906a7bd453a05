"""One n = 1000 filter (Sigma = 32 MB, smaller than the 256 MB Infinity Cache), known association: run under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to compare HBM traffic per correction with the algorithmic
2*8*N^2 bytes (the counters must under-report when Sigma stays cached)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_ml_amd import capi, synth
fused = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
cfg = synth.config3(steps=60)
log = synth.make_known_log(cfg)
f = capi.EKF_SLAM(1000)
f.set_call_fused(fused)
t0 = time.perf_counter(); corr = 0
for t in range(cfg.steps):
    s, v = log.expand_step(t)
    f.prediction(log.twist[t, 0]); f.measurement(s, v); corr += int(v.sum())
f.sync()
print(f"fused={fused}: {corr} corrections, {(time.perf_counter() - t0) / max(corr, 1) * 1e6:.1f} us each; algorithmic bytes per correction "
      f"{16 * 2003 ** 2 / 1e6:.2f} MB")
