"""Generates tests/golden/config2_100k.npz: the CPU oracle's result on BASELINE.json configs[1]
(100k cells x 33k genes @ 3 % nnz, CellRanger normalisation, BkSvd{2.0, 5} top-50) so that the driver-run
`-m gpu` suite can hold the HIP path to it without the oracle's minutes of CPU time on the GPU box.

    python tests/golden/make_config_fixtures.py            # ~3 min of one host core, ~6 GB of host memory
    python tests/golden/make_config_fixtures.py config3    # configs[2], 1M cells: ~1 h of one host core, ~30 GB of host memory

Inputs are reproducible on the GPU box: `scanrs_amd.synth.synth_counts_fast(100_000, 33_000, 0.03, 0)` (numpy Generator,
PCG64) and the start panel `scanrs_oracle.omega_panel((100, 33_000), 0)` restated inside the product as
`scanrs_omega_fill`. The fixture holds the matrix checksums (a generator mismatch fails loudly, not as a parity
error), all 50 singular values, and the sign-normalised loadings on a fixed subsample of rows of U (genes) and V (cells)
plus every column's sum and sum of absolute values over ALL rows (so that rows outside the subsample are covered too)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import scanrs_oracle as so  # noqa: E402
from scanrs_amd.synth import synth_counts_fast  # noqa: E402

CELLS, GENES, DENSITY, K, SEED = 100_000, 33_000, 0.03, 50, 0
N_SUB = 500


def sign_normalise(a):
    """every column's entry of largest magnitude made positive (a rule both sides can apply on their own)"""
    idx = np.argmax(np.abs(a), axis=0)
    return a * np.sign(a[idx, np.arange(a.shape[1])])


def main():
    t0 = time.time()
    m = synth_counts_fast(CELLS, GENES, DENSITY, SEED)  # cells x genes CSR == genes x cells CSC
    print(f"matrix: nnz {m.nnz} ({time.time() - t0:.1f} s)", flush=True)
    omega = so.omega_panel((2 * K, GENES), 0)
    so.build()
    o = so.AdaptiveMat(GENES, CELLS, so.CSC, m.indptr, m.indices, m.data)
    t0 = time.time()
    u, s, v = so.BkSvd().run_pca(so.normalize(o, "cellranger"), K, omega=omega)
    print(f"oracle: {time.time() - t0:.1f} s", flush=True)
    u, v = sign_normalise(u), v * np.sign(u[np.argmax(np.abs(u), axis=0), np.arange(K)])
    gi = np.linspace(0, GENES - 1, N_SUB).astype(np.int64)
    ci = np.linspace(0, CELLS - 1, N_SUB).astype(np.int64)
    out = os.path.join(ROOT, "tests", "golden", "config2_100k.npz")
    np.savez_compressed(
        out,
        cells=CELLS, genes=GENES, density=DENSITY, k=K, seed=SEED,
        nnz=np.int64(m.nnz), sum_indices=np.int64(m.indices.astype(np.int64).sum()), sum_values=np.int64(m.data.astype(np.int64).sum()),
        indptr_probe=m.indptr[:: CELLS // 100].astype(np.int64),
        sigma=s, gene_rows=gi, cell_rows=ci, u_sub=u[gi], v_sub=v[ci],
        u_colsum=u.sum(axis=0), v_colsum=v.sum(axis=0), u_colabs=np.abs(u).sum(axis=0), v_colabs=np.abs(v).sum(axis=0),
    )
    print(f"wrote {out} ({os.path.getsize(out) / 1e3:.0f} kB)")


def main_config3():
    """configs[2]: 1 M cells x 33 k genes @ 3 % -> tests/golden/config3_1M.npz. Input from `synth_counts_par` (one numpy
    generator per chunk of 8192 cells: the GPU box regenerates the 10^9 nonzeros with a process pool in well under a minute).
    The oracle runs its serial loops (the reference's own accumulation order); only LAPACK is threaded."""
    from scanrs_amd.synth import synth_counts_par

    cells = 1_000_000
    t0 = time.time()
    ip, ix, vv = synth_counts_par(cells, GENES, DENSITY, SEED)
    print(f"matrix: nnz {int(ip[-1])} ({time.time() - t0:.1f} s)", flush=True)
    omega = so.omega_panel((2 * K, GENES), 0)
    so.build()
    o = so.AdaptiveMat(GENES, cells, so.CSC, ip, ix, vv)
    t0 = time.time()
    u, s, v = so.BkSvd().run_pca(so.normalize(o, "cellranger"), K, omega=omega)
    print(f"oracle: {time.time() - t0:.1f} s", flush=True)
    u, v = sign_normalise(u), v * np.sign(u[np.argmax(np.abs(u), axis=0), np.arange(K)])
    gi = np.linspace(0, GENES - 1, N_SUB).astype(np.int64)
    ci = np.linspace(0, cells - 1, N_SUB).astype(np.int64)
    out = os.path.join(ROOT, "tests", "golden", "config3_1M.npz")
    np.savez_compressed(
        out,
        cells=cells, genes=GENES, density=DENSITY, k=K, seed=SEED,
        nnz=np.int64(ip[-1]), sum_indices=np.int64(ix.astype(np.int64).sum()), sum_values=np.int64(vv.astype(np.int64).sum()),
        indptr_probe=ip[:: cells // 100].astype(np.int64),
        sigma=s, gene_rows=gi, cell_rows=ci, u_sub=u[gi], v_sub=v[ci],
        u_colsum=u.sum(axis=0), v_colsum=v.sum(axis=0), u_colabs=np.abs(u).sum(axis=0), v_colabs=np.abs(v).sum(axis=0),
    )
    print(f"wrote {out} ({os.path.getsize(out) / 1e3:.0f} kB)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "config3":
        main_config3()
    else:
        main()
