"""scan_rs::nn (scan-rs/src/nn.rs): exact kNN of the PCA scores on the device against the exhaustive oracle, on the
shapes of the reference's own tests (`test_knn` nn.rs:157-168, `test_find_nn` :170-199, `test_symmetry` :201-229)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import knn_oracle as ko  # noqa: E402

ALL_PTS = np.array([[float(i), float(i)] for i in range(10)])


def test_oracle_reproduces_the_reference_find_nn_table():
    # nn.rs:170-199: tree over points 1, 3, 4, 7 of the diagonal, one neighbour of every diagonal point, self included
    tree = ALL_PTS[[1, 3, 4, 7]]
    got = ko.exhaustive_find_nn(ALL_PTS, tree, 1, True)[:, 0]
    assert got.tolist() == [0, 0, 0, 1, 2, 2, 3, 3, 3, 3]


def test_oracle_symmetry_case_distance_classes():
    # nn.rs:201-229: the expected table depends on the ball tree's order among equidistant points; the distance classes
    # (outlier first / last) are what an exhaustive search must reproduce
    v = np.eye(5)
    v[0, 4] = 3.0
    ref = np.array([[4, 2, 1, 3], [4, 2, 3, 0], [4, 1, 3, 0], [4, 1, 2, 0], [2, 1, 3, 0]])
    got = ko.exhaustive_knn(v, 4)
    for r in range(5):
        assert sorted(got[r]) == sorted(ref[r])
    assert got[0, 0] == ref[0, 0] == 4  # the one closer point of row 0 comes first in both
    assert all(got[r, 3] == ref[r, 3] == 0 for r in (1, 2, 3, 4))  # and the outlier last for everyone else


@pytest.mark.gpu
def test_knn_matches_exhaustive_on_the_reference_shapes():
    import scanrs_amd as sa

    rng = np.random.default_rng(0)
    for ncells in (3, 5, 50, 100):
        for d in (1, 2, 3, 5, 10, 20, 50):
            v = rng.standard_normal((ncells, d))
            full = ko.exhaustive_knn(v, min(ncells - 1, 50))
            for k in (1, 5, 10, 25, 50):
                if k >= ncells:
                    continue
                got = sa.knn(v, k)
                assert got.dtype == np.uint32
                assert np.array_equal(got.astype(np.int64), full[:, :k]), (ncells, d, k)


@pytest.mark.gpu
def test_find_nn_reference_table_and_padding():
    import scanrs_amd as sa

    tree = ALL_PTS[[1, 3, 4, 7]]
    out = sa.find_nn(ALL_PTS, 1, tree, True)
    assert out[:, 0].tolist() == [0, 0, 0, 1, 2, 2, 3, 3, 3, 3]
    # fewer points than k: the tail keeps T::max_value() (nn.rs:66)
    out = sa.find_nn(ALL_PTS[:3], 6, tree, True)
    assert np.all(out[:, 4:] == np.iinfo(np.uint32).max)
    assert np.array_equal(out.astype(np.int64), ko.exhaustive_find_nn(ALL_PTS[:3], tree, 6, True))
    # include_self = false drops the tree point whose index equals the query's row number
    assert np.array_equal(sa.find_nn(tree, 3, tree, False).astype(np.int64), ko.exhaustive_find_nn(tree, tree, 3, False))
    v = np.eye(5)
    v[0, 4] = 3.0
    assert np.array_equal(sa.knn(v, 4).astype(np.int64), ko.exhaustive_knn(v, 4))  # ties in ascending index order


@pytest.mark.gpu
def test_knn_larger_set_all_dimension_kernels_and_duplicates():
    import scanrs_amd as sa

    rng = np.random.default_rng(3)
    for n, d, k in ((3000, 50, 15), (700, 7, 128), (513, 33, 40), (300, 100, 9), (260, 128, 5)):
        v = rng.standard_normal((n, d))
        v[10] = v[3]  # an exact duplicate is a legitimate neighbour at distance 0
        got = sa.knn(v, k).astype(np.int64)
        want = ko.exhaustive_knn(v, k)
        assert np.array_equal(got, want), (n, d, k)
        assert got[3, 0] == 10 and got[10, 0] == 3
    with pytest.raises(sa.ScanrsError):
        sa.knn(rng.standard_normal((10, 129)), 2)
    with pytest.raises(sa.ScanrsError):
        sa.knn(rng.standard_normal((300, 4)), 129)


def _knn_both_ways(sa, fn, monkeypatch):
    """run `fn` with the matrix-core filter allowed and with the exhaustive kernel forced (global option, read per call)"""
    sa.set_global_option("knn_exhaustive", 0)
    a = fn()
    sa.set_global_option("knn_exhaustive", 1)
    try:
        b = fn()
    finally:
        sa.set_global_option("knn_exhaustive", 0)
    return a, b


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k", [(40_000, 50, 15), (36_000, 7, 30), (150_000, 50, 15)])
def test_filtered_knn_equals_exhaustive(monkeypatch, n, d, k):
    """Large point sets go through the bf16-MFMA filter + exact f64 rerank (knn.hip): the result must be IDENTICAL to the
    exhaustive f64 kernel's — same neighbours, same order, ties by index — on PCA-score-like data (decaying column scales,
    clusters), with exact duplicates, for one (n <= 131k) and two (n > 131k) filter rounds."""
    import scanrs_amd as sa

    rng = np.random.default_rng(n + d)
    centres = rng.standard_normal((12, d)) * 3.0
    v = centres[rng.integers(0, 12, size=n)] + rng.standard_normal((n, d))
    v *= np.linspace(1.0, 0.3, d)  # decaying spectrum, as PCA scores have
    v[17] = v[5]
    v[n - 1] = v[5]  # three exactly coincident points
    got, want = _knn_both_ways(sa, lambda: sa.knn(v, k), monkeypatch)
    assert np.array_equal(got, want)
    assert set(got[5, :2].tolist()) == {17, n - 1} and got[5, 0] == 17  # distance-0 ties in ascending index order


@pytest.mark.gpu
def test_filtered_find_nn_and_overflow_fallback(monkeypatch):
    import scanrs_amd as sa

    rng = np.random.default_rng(5)
    pts = rng.standard_normal((50_000, 20))
    qs = rng.standard_normal((3_000, 20))
    for include_self in (True, False):
        got, want = _knn_both_ways(sa, lambda: sa.find_nn(qs, 10, pts, include_self), monkeypatch)
        assert np.array_equal(got, want)
    # 3000 coincident points: every one of them has > 1024 candidates at distance 0 -> the lists overflow and those queries
    # are redone exhaustively; the answer is still exact (ties by index)
    pts2 = pts.copy()
    pts2[:3000] = pts2[0]
    got, want = _knn_both_ways(sa, lambda: sa.knn(pts2, 8), monkeypatch)
    assert np.array_equal(got, want)
    assert got[0].tolist() == [1, 2, 3, 4, 5, 6, 7, 8] and got[2999].tolist() == [0, 1, 2, 3, 4, 5, 6, 7]


@pytest.mark.gpu
def test_knn_device_on_pca_scores(monkeypatch):
    """scanrs_knn_device on the scores scanrs_pca_result_device hands out (leading dimension != d) = knn of the host copy"""
    import scanrs_amd as sa
    from scanrs_amd.synth import synth_counts

    m = synth_counts(40_000, 600, 0.05, 2)
    g = sa.AdaptiveMat.from_csmat(m.shape[1], m.shape[0], sa.CSC, m.indptr, m.indices, m.data)
    sa.normalize(g, sa.Normalization.CellRanger)
    u, s, v = sa.BkSvd().run_pca(g, 9)
    s2, res = sa.BkSvd().run_pca_device(g, 9)
    assert res.ld_v == 10 and res.k == 9
    a = sa.knn_device(res.d_v, m.shape[0], res.ld_v, res.k, 15)
    sa.set_global_option("knn_exhaustive", 1)
    try:
        b = sa.knn(v, 15)
    finally:
        sa.set_global_option("knn_exhaustive", 0)
    assert np.array_equal(a, b)
