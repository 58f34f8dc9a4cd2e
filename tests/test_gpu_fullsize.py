"""Full-size checks at BASELINE.json's headline shape (1 M cells x 33 k genes, 3 % nnz) through
size-independent properties: integer linearity and checksums (bit-exact), agreement of the two
orientations, agreement of the two product kernels, and the defining relations of the returned SVD."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CELLS, N_GENES, DENSITY, K = 1_000_000, 33_000, 0.03, 50


@pytest.fixture(scope="module")
def big():
    import torch

    import scanrs_amd as sa
    from scanrs_amd.synth import synth_counts_torch

    if not sa.device_available():
        pytest.fail("gfx950 device required")
    dev = torch.device("cuda", 0)
    indptr, indices, values = synth_counts_torch(N_CELLS, N_GENES, DENSITY, 0, dev)
    stats = {
        "nnz": int(indptr[-1].item()),
        "total": int(values.to(torch.int64).sum().item()),
        "lib": torch.segment_reduce(values.to(torch.float64), "sum", offsets=indptr).to(torch.int64).cpu().numpy(),
        "gene": torch.zeros(N_GENES, dtype=torch.int64, device=dev).index_add_(0, indices.to(torch.int64), values.to(torch.int64)).cpu().numpy(),
    }
    mat = sa.AdaptiveMat.from_device(N_GENES, N_CELLS, sa.CSC, indptr.data_ptr(), indices.data_ptr(), values.data_ptr())
    del indptr, indices, values
    torch.cuda.empty_cache()
    return sa, mat, stats


def test_shape_nnz_and_checksums(big):
    sa, mat, st = big
    assert mat.shape() == [N_GENES, N_CELLS] and mat.nnz() == st["nnz"]
    lib = mat.sum_axis(0, np.uint32)  # per barcode, on the cell-major copy
    gene = mat.sum_axis(1, np.uint32)  # per gene, on the transposed (gene-major) copy
    assert np.array_equal(lib.astype(np.int64), st["lib"])
    assert np.array_equal(gene.astype(np.int64), st["gene"])
    assert int(lib.astype(np.int64).sum()) == st["total"] == int(gene.astype(np.int64).sum())  # checksum of checksums


def test_integer_products_linearity_and_orientations(big):
    sa, mat, st = big
    rng = np.random.default_rng(0)
    x = rng.integers(0, 1000, size=(N_CELLS, 2), dtype=np.uint32)
    y = rng.integers(0, 1000, size=(N_CELLS, 2), dtype=np.uint32)
    ax, ay, axy = mat.dot(x), mat.dot(y), mat.dot(x + y)
    assert np.array_equal(axy, ax + ay)  # exact in wrapping u32 arithmetic
    ones = np.ones((N_CELLS, 1), dtype=np.uint32)
    assert np.array_equal(mat.dot(ones)[:, 0].astype(np.int64), st["gene"])
    # lhs.dot(A) runs the other orientation's kernel: 1^T A = library sizes
    ones_g = np.ones((1, N_GENES), dtype=np.uint32)
    assert np.array_equal(mat.rdot(ones_g)[0].astype(np.int64), st["lib"])
    # <w, A x> == <A^T w, x>  (mod 2^32)
    w = rng.integers(0, 1000, size=(1, N_GENES), dtype=np.uint32)
    lhs = (w.astype(np.uint64) @ ax.astype(np.uint64))[0] & 0xFFFFFFFF
    rhs = (mat.rdot(w).astype(np.uint64) @ x.astype(np.uint64))[0] & 0xFFFFFFFF
    assert np.array_equal(lhs, rhs)


def test_product_kernels_agree_at_full_size(big):
    sa, mat, st = big
    rng = np.random.default_rng(1)
    m = sa.normalize(mat.view(), sa.Normalization.CellRanger)
    q = rng.standard_normal((N_CELLS, 16))
    ql = rng.standard_normal((16, N_GENES))
    m.set_spmm_path(1)
    a1, b1 = m.dot(q), m.rdot(ql)
    m.set_spmm_path(2)
    a2, b2 = m.dot(q), m.rdot(ql)
    m.set_spmm_path(0)
    # same sums in a different association order: 1e-11 of the column scale
    assert np.max(np.abs(a1 - a2)) <= 1e-11 * np.max(np.abs(a1))
    assert np.max(np.abs(b1 - b2)) <= 1e-11 * np.max(np.abs(b1))
    # centred rows: every gene has mean 0 over the barcodes  =>  A 1 = 0
    z = m.dot(np.ones((N_CELLS, 1)))
    assert np.max(np.abs(z)) < 1e-6


def test_pca_defining_relations(big):
    sa, mat, st = big
    m = sa.normalize(mat.view(), sa.Normalization.CellRanger)
    u, s, v = sa.BkSvd().run_pca(m, K)
    assert u.shape == (N_GENES, K) and v.shape == (N_CELLS, K) and s.shape == (K,)
    assert np.all(np.diff(s) <= 0) and s[-1] > 0  # sorted
    assert np.max(np.abs(u.T @ u - np.eye(K))) < 1e-10
    assert np.max(np.abs(v.T @ v - np.eye(K))) < 1e-10
    # Ritz relations of the returned triplets: A^T u_i = s_i v_i exactly (by construction of V),
    # and ||A v_i - s_i u_i|| small for the converged leading components
    atu = m.rdot(u.T.copy()).T
    assert np.max(np.abs(atu - v * s)) < 1e-9 * s[0]
    av = m.dot(v)
    resid = np.linalg.norm(av - u * s, axis=0) / s
    assert np.all(resid[:10] < 1e-6), resid[:10]
    # idempotence / determinism: a second run is bitwise identical
    u2, s2, v2 = sa.BkSvd().run_pca(m, K)
    assert np.array_equal(s, s2) and np.array_equal(u, u2) and np.array_equal(v, v2)
