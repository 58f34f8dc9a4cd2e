import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
rng = np.random.default_rng(0)
n, k = 500, 50
# spectrum like a PCA projection Gram: few large, decaying bulk
q = np.linalg.qr(rng.standard_normal((n, n)))[0]
lam = np.concatenate([np.linspace(1e8, 2e7, 50), 1e7 * np.exp(-np.arange(n - 50) / 60.0)])
g = (q * lam) @ q.T; g = (g + g.T) / 2
for _ in range(2):
    t0 = time.perf_counter(); w, z = sa.host_sym_eig_topk(g, k); t1 = time.perf_counter()
print("topk ms", (t1 - t0) * 1e3)
wr = np.linalg.eigvalsh(g)[::-1][:k]
print("w rel err", np.max(np.abs(w - wr) / wr), "orth", np.max(np.abs(z.T @ z - np.eye(k))), "resid", np.max(np.abs(g @ z - z * w)) / wr[0])
t0 = time.perf_counter(); np.linalg.eigh(g); print("numpy eigh ms", (time.perf_counter() - t0) * 1e3)
