import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401  (one HIP runtime)
from ekf_slam_ml_amd import capi as hip
from oracle import binding as oracle
import test_gpu_fuzz as fz
bad = 0
START = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
COUNT = int(sys.argv[2]) if len(sys.argv) > 2 else 600
for seed in range(START, START + COUNT):
    try:
        fz._scenario(hip, oracle, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:200], flush=True)
    if seed % 250 == 0: print("seed", seed, "failures so far", bad, flush=True)
print("done, failures:", bad)
