"""scan-rs-cmd mirror (tools/scan_rs_cmd.cpp over include/scanrs_amd.hpp): MTX in, svd_{u,d,v}.csv.gz out,
same flags and outputs as the reference CLI (tools/src/bin/cmd.rs:16-104)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "scan-rs_amd", "lib", "scan-rs-cmd")


def _write_mtx(path, m):
    coo = m.tocoo()
    with gzip.open(path, "wt") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n% comment\n")
        f.write(f"{m.shape[0]} {m.shape[1]} {coo.nnz}\n")
        order = np.random.default_rng(0).permutation(coo.nnz)  # unsorted triplets
        for i in order:
            f.write(f"{coo.row[i] + 1} {coo.col[i] + 1} {coo.data[i]}\n")


def _read_csv(path):
    with gzip.open(path, "rt") as f:
        return np.array([[float(x) for x in line.strip().split(",")] for line in f if line.strip()])


@pytest.mark.parametrize("norm", ["cellranger", "seuratlog", "binomialpearson"])
def test_cli_matches_api(tmp_path, norm):
    import scanrs_amd as sa
    import scanrs_oracle as so
    from scanrs_amd.synth import synth_counts

    m = synth_counts(900, 250, 0.1, 2).T.tocsr()  # genes x cells, as Cell Ranger writes matrix.mtx
    m.sort_indices()
    mtx = os.path.join(tmp_path, "matrix.mtx.gz")
    _write_mtx(mtx, m)
    out = os.path.join(tmp_path, "out")
    r = subprocess.run([CLI, mtx, "-o", out, "-n", norm, "-d", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    u, d, v = (_read_csv(os.path.join(out, f"svd_{x}.csv.gz")) for x in "udv")
    assert u.shape == (250, 7) and d.shape == (7, 1) and v.shape == (900, 7)
    g = sa.AdaptiveMat.from_csmat(250, 900, sa.CSR, m.indptr, m.indices, m.data)
    u2, s2, v2 = sa.BkSvd().run_pca(sa.normalize(g, sa.Normalization.from_str(norm)), 7)
    assert np.array_equal(d[:, 0], s2) and np.array_equal(u, u2) and np.array_equal(v, v2)  # CSV round-trips f64 exactly
    o = so.AdaptiveMat(250, 900, so.CSR, m.indptr, m.indices, m.data)
    o = so.normalize(o, norm) if norm != "binomialpearson" else so.binom_pearson_resid(o)
    _, s_o, _ = so.BkSvd().run_pca(o, 7)
    assert np.max(np.abs(d[:, 0] - s_o) / s_o) < 1e-8


def test_cli_errors(tmp_path):
    r = subprocess.run([CLI], capture_output=True, text=True)
    assert r.returncode == 2 and "INPUT" in r.stderr
    mtx = os.path.join(tmp_path, "m.mtx.gz")
    import scipy.sparse as sp

    _write_mtx(mtx, sp.csr_matrix(np.array([[1, 2, 0], [0, 3, 4]], dtype=np.uint32)))
    r = subprocess.run([CLI, mtx, "-n", "nope"], capture_output=True, text=True)
    assert r.returncode == 1 and "Normalization not recognized: nope" in r.stderr
    r = subprocess.run([CLI, mtx, "-o", str(tmp_path), "-d", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "invalid k" in r.stderr
