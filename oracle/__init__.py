"""TEST INFRASTRUCTURE ONLY -- CPU checker for the EKF-SLAM hot path.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (ekf_slam_ml_amd) never imports anything from here."""
