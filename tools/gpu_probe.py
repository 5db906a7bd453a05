"""Scratch probe run on the GPU box: torch + libekfslam_hip in one process, first batch timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if "--torch" in sys.argv:
    import torch
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0))
    x = torch.ones(4, device="cuda") * 2
    print("torch sum", float(x.sum()))
from ekf_slam_ml_amd import capi, synth
import __graft_entry__ as g
g.smoke()
for B, n in ((8, 1000), (64, 1000), (512, 1000)):
    t0 = time.time()
    log = synth.make_known_log(synth.config5(filters=B, steps=8, n=n))
    t1 = time.time()
    bf = capi.BatchEKF(B, n)
    bf.upload_known_log(log.twist, log.lm_idx, log.z_xy, log.init_xy)
    bf.run_known(0, 2)
    for nt in (0, 1):
        for rows in (4, 8, 16, 32):
            bf.set_tuning(rows, nt)
            st = bf.run_known(2, 8, time_kernels=True)
            gbs = st["rank2_bytes_per_launch"] / (st["rank2_ms"] / st["rank2_launches"] * 1e-3) / 1e9
            print(f"B={B} n={n} nt={nt} rows={rows}: {st['elapsed_ms']:.2f} ms, rank2 {st['rank2_ms']:.2f} ms / {st['rank2_launches']} launches, "
                  f"{st['corrections'] / (st['elapsed_ms'] * 1e-3):.0f} corr/s, rank2 alg {gbs:.0f} GB/s", flush=True)
            bf.reset(); bf.run_known(0, 2)
    print("gen", t1 - t0, "bytes", bf.device_bytes() / 1e9, "checksum", bf.checksum())
    bf.close()
