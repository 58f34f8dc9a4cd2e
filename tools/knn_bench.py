"""Throughput of the exact kNN (scan_rs::nn::knn) on synthetic PCA scores: n x d, k neighbours; the matrix-core filter path
(bf16 MFMA filter + exact f64 rerank) against the exhaustive f64 kernel (global option "knn_exhaustive"), results compared."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
also_exhaustive = (len(sys.argv) > 4 and sys.argv[4] == "both") or n <= 300_000
rng = np.random.default_rng(0)
centres = rng.standard_normal((20, d)) * 2.0
v = (centres[rng.integers(0, 20, size=n)] + rng.standard_normal((n, d))) * np.linspace(1.0, 0.4, d)
sa.knn(v[:1000], k)
res = {}
for mode in (["filter", "exhaustive"] if also_exhaustive else ["filter"]):
    sa.set_global_option("knn_exhaustive", 1 if mode == "exhaustive" else 0)
    t0 = time.perf_counter()
    out = sa.knn(v, k)
    dt = time.perf_counter() - t0
    res[mode] = out
    flops = 3.0 * n * n * d  # subtract + fused multiply-add per coordinate pair
    print(f"knn[{mode}] n={n} d={d} k={k}: {dt*1e3:.1f} ms incl. PCIe of the points, {n/dt:.0f} cells/s, "
          f"{flops/dt/1e12:.2f} TFLOP/s f64-equivalent, first row {out[0][:5]}", flush=True)
if len(res) == 2:
    print("identical:", bool(np.array_equal(res["filter"], res["exhaustive"])))
