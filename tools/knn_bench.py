"""Throughput of the exhaustive kNN (scan_rs::nn::knn) on synthetic PCA scores: n x d standard normal, k neighbours."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
v = np.random.default_rng(0).standard_normal((n, d))
sa.knn(v[:1000], k)
t0 = time.perf_counter()
out = sa.knn(v, k)
dt = time.perf_counter() - t0
flops = 3.0 * n * n * d  # subtract + fused multiply-add per coordinate pair
print(f"knn n={n} d={d} k={k}: {dt*1e3:.1f} ms incl. PCIe, {n/dt:.0f} cells/s, {flops/dt/1e12:.2f} TFLOP/s f64 "
      f"(vector peak 78.6), first row {out[0][:5]}")
