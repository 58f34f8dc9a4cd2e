"""Kernel time of the device-side Cholesky + inverse step (chol_rinv_kernel) for a few sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
rng = np.random.default_rng(0)
d = (rng.random((20, 30)) < 0.3).astype(np.uint32)
g0 = sa.AdaptiveMat.from_dense(d)
for n in (16, 50, 100, 128):
    x = rng.standard_normal((4 * n, n))
    g = x.T @ x
    g0.profile_enable(True)
    g0.profile_reset()
    for _ in range(5):
        g0.chol_rinv(g, 4 * n)
    st = g0.profile_get()["chol_rinv"]
    print(n, st["total_ms"] / st["launches"] * 1e3, "us")
