"""Does the delayed update gain from overlapping one half-pool's flush with the other half's gain steps?  Two BatchEKF
handles of B/2 filters each (own HIP streams), driven by two host threads, the second started half a flush window later;
against ONE handle of B filters.  usage: python tools/delayed_overlap.py [B=4096] [k=32]"""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime)
from ekf_slam_ml_amd import capi, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W, K = 2, 64          # timed: 64 steps = 128 corrections per filter = 4 whole flushes at k = 32


def make(Bh, first):
    cfg = synth.config5(filters=Bh, steps=1 + W + K + k, first_filter_id=first, n=1000)
    bt = capi.BatchEKF(Bh, 1000)
    bt.simulate_known_log(cfg, synth.make_world(1000, cfg.half_extent, cfg.min_spacing, cfg.world_seed))
    bt.set_update_mode(k)
    bt.run_known(0, 1 + W)
    return bt


one = make(B, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
st = one.run_known(1 + W, 1 + W + K)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"one handle of {B}: {st['corrections'] / (t1 - t0):.0f} update steps/s", flush=True)
one.close()

halves = [make(B // 2, 0), make(B // 2, B // 2)]
halves[1].run_known(1 + W, 1 + W + k // 4)   # half a window ahead: its flushes fall between the other half's
torch.cuda.synchronize()
corr = [0, 0]


def work(i, t_from):
    corr[i] = halves[i].run_known(t_from, t_from + K)["corrections"]


th = [threading.Thread(target=work, args=(0, 1 + W)), threading.Thread(target=work, args=(1, 1 + W + k // 4))]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"two handles of {B // 2}, half a window apart: {sum(corr) / (t1 - t0):.0f} update steps/s", flush=True)
for h in halves: h.close()
