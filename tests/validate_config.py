"""One-off parity run at a BASELINE.json configuration size: the GPU path (C ABI) against the CPU oracle on the SAME
synthetic matrix and the SAME start panel.  Default = configs[1]: 100k cells x 33k genes @ 3 %, top-50 PCA.
The oracle needs minutes of one host core at this size, so this is a script (result quoted in DESIGN.md), not a test.

    python tests/validate_config.py [cells] [genes] [density] [k]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import scanrs_amd as sa
import scanrs_oracle as so
from scanrs_amd.synth import synth_counts

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 33_000
density = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
k = int(sys.argv[4]) if len(sys.argv) > 4 else 50

t0 = time.time()
m = synth_counts(cells, genes, density, 0)  # cells x genes CSR == genes x cells CSC
print(f"matrix {genes} x {cells}, nnz {m.nnz} ({time.time()-t0:.1f} s)", flush=True)
omega = so.omega_panel((2 * k, genes), 0)

g = sa.AdaptiveMat.from_csmat(genes, cells, sa.CSC, m.indptr, m.indices, m.data)
t0 = time.time()
sa.normalize(g, sa.Normalization.CellRanger)
u, s, v = sa.BkSvd().run_pca(g, k, omega=omega)
print(f"gpu: {time.time()-t0:.2f} s (first call, includes the transposed-copy build)", flush=True)

so.build()
o = so.AdaptiveMat(genes, cells, so.CSC, m.indptr, m.indices, m.data)
t0 = time.time()
uo, s_o, vo = so.BkSvd().run_pca(so.normalize(o, "cellranger"), k, omega=omega)
print(f"oracle: {time.time()-t0:.1f} s on one core", flush=True)


def sign_fix(a, ref):
    return a * np.sign(np.sum(a * ref, axis=0))


print(f"sigma: max rel err {np.max(np.abs(s - s_o) / s_o):.3e}   (north-star tolerance 1e-4)")
print(f"U (genes x k): max abs err {np.max(np.abs(sign_fix(u, uo) - uo)):.3e}")
print(f"V (cells x k): max abs err {np.max(np.abs(sign_fix(v, vo) - vo)):.3e}")
gap = np.min(np.abs(np.diff(s_o)) / s_o[:-1])
print(f"smallest relative gap between consecutive singular values: {gap:.3e}")
