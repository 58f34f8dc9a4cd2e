"""Pin the CPU oracle against the reference's own inline known-answer tables
(tests/golden/reference_tables.json; sources cited inside the fixture)."""
import numpy as np
import pytest

import scanrs_oracle as so


def _mat(dense, storage=so.CSR):
    return so.AdaptiveMat.from_dense(np.array(dense, dtype=np.uint32), storage)


def assert_close(a, b, rtol=1e-7, atol=1e-12):
    # the reference's assert_close (sqz/src/matrix_map.rs:339-357)
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.all(np.abs(a - b) <= np.abs(b) * rtol + atol), (a, b)


@pytest.mark.parametrize("storage", [so.CSR, so.CSC])
def test_normalization_tables(golden, storage):
    g = golden["normalization"]
    tol = g["abs_tol"]
    m = lambda: _mat(g["dense"], storage)
    out = so.normalize_with_size_factor(m(), "cellranger").to_dense()
    assert np.allclose(out, g["cellranger"]["expected"], rtol=0, atol=tol)
    out = so.normalize(m(), "cellranger").to_dense()
    assert np.allclose(out, g["cellranger"]["expected"], rtol=0, atol=tol)
    out = so.normalize_with_size_factor(m(), "cellranger8").to_dense()
    assert np.allclose(out, g["cellranger8"]["expected"], rtol=0, atol=tol)
    out = so.normalize_with_size_factor(m(), "logtransform").to_dense()
    assert np.allclose(out, g["logtransform"]["expected"], rtol=0, atol=tol)
    # size factors: 1 + column sums of the picked features (normalization.rs:645-646)
    sf = g["size_factor_lognorm"]
    dense = np.array(g["dense"], dtype=np.uint32)
    size_factors = sf["size_factor_offset"] + dense[sf["features_picked"], :].sum(axis=0)
    out = so.log_normalize_with_size_factor(m(), None, so.LOG_TWO, size_factors.astype(np.uint32)).to_dense()
    assert np.allclose(out, sf["expected"], rtol=0, atol=tol)
    fp = g["fixed_point"]
    out = so.log1p_normalize_fixed_point(_mat(fp["dense"], storage), so.LOG_TWO, fp["base"], fp["exponent"]).to_dense()
    assert np.allclose(out, fp["expected"], rtol=0, atol=tol)


def test_multinomial(golden):
    g = golden["normalization"]["multinomial"]
    n, pi = so.fit_multinomial_model(_mat(g["dense"]))
    assert_close(n, g["expected_n"])
    assert_close(pi, g["expected_pi"])


@pytest.mark.parametrize("storage", [so.CSR, so.CSC])
def test_mat_stats(golden, storage):
    g = golden["mat_stats"]
    a = _mat(g["input_a"], storage)
    assert a.sum_axis(0, np.uint32).tolist() == g["sum0"]
    assert a.sum_axis(1, np.uint32).tolist() == g["sum1"]
    assert np.allclose(a.mean_axis(0), g["mean0"], rtol=0, atol=g["abs_tol"])
    assert np.allclose(a.mean_axis(1), g["mean1"], rtol=0, atol=g["abs_tol"])
    for axis, mk, vk in ((0, "mean0", "var0"), (1, "mean1", "var1")):
        mean, var = a.mean_var_axis(axis)
        assert np.allclose(mean, g[mk], rtol=0, atol=g["abs_tol"])
        assert np.allclose(var, g[vk], rtol=0, atol=g["abs_tol"])


def test_mat_misc_center(golden):
    g = golden["mat_misc"]
    a = _mat(g["dense"])
    dense = np.array(g["dense"], dtype=np.float64)
    for axis in (0, 1):
        assert_close(a.sum_axis(axis), dense.sum(axis=axis))
        assert_close(a.mean_axis(axis), dense.mean(axis=axis))
        assert_close(a.var_axis(axis), dense.var(axis=axis))
    assert_close(a.view().center(0, None).to_dense(), g["centered_cols"])
    assert_close(a.view().center(1, None).to_dense(), g["centered_rows"])


def test_matrix_map_tables(golden):
    g = golden["matrix_map"]
    # the tables use f64 inputs; integer-valued so the u32 storage carries them exactly
    sa = g["scale_axis"]
    f = np.array(sa["scale_factors"])
    orig = _mat(np.array(sa["orig"]))
    for mat, fac_axis, exp in ((orig, 0, sa["expected_rows"]), (orig, 1, sa["expected_cols"])):
        got = mat.compose_map(so.MapOp(so.OP_SCALE_AXIS, axis=fac_axis, a=f)).to_dense()
        assert_close(got, exp)
        # transpose check (matrix_map.rs:375-399): map.t() on orig.t() gives expected.t()
        got_t = mat.compose_map(so.MapOp(so.OP_SCALE_AXIS, axis=fac_axis, a=f)).t().to_dense()
        assert_close(got_t, np.array(exp).T)
    c = g["composed"]
    orig = _mat(np.array(c["orig"]))
    rows = so.MapOp(so.OP_SCALE_AXIS, axis=0, a=np.array(c["row_factors"]))
    assert_close(orig.compose_map(rows).apply(so.OP_SQUARE).to_dense(), c["scale_then_square"])
    assert_close(orig.apply(so.OP_SQUARE).compose_map(rows).to_dense(), c["square_then_scale"])
    assert_close(orig.apply(so.OP_SQUARE).compose_map(rows).t().to_dense(), np.array(c["square_then_scale"]).T)
    ln = np.array(g["scalar_ln1p"]["orig"], dtype=np.float64)
    assert_close(_mat(ln).apply(so.OP_LN_1P).to_dense(), np.log(ln + 1.0))


def test_median(golden):
    for xs, exp in golden["median"]["int_cases"]:
        assert int(so.median_mut(np.array(xs, dtype=np.uint32))) == exp
    for xs, exp in golden["median"]["float_cases"]:
        assert float(so.median_mut(np.array(xs, dtype=np.float64))) == exp
    assert so.median_mut(np.array([], dtype=np.uint32)) is None


def test_one_dim_nan_guard(golden):
    g = golden["one_dim_nan_guard"]
    mat = _mat(np.array(g["values"], dtype=np.uint32).reshape(g["shape"]))
    out = so.normalize(mat.view(), "cellranger").t().to_dense()
    assert not np.isnan(out).any()


def _random_matrices(rng, n, step):
    # shapes as sqz/src/mat.rs:1225-1237 (random_matrices), values in [1, range)
    import scipy.sparse as sp

    for i in range(0, n, step):
        rows = int(rng.integers(0, i + 1) + rng.integers(0, i + 1))
        cols = int(rng.integers(0, i + 1) + rng.integers(0, i + 1))
        rng_max = int(rng.integers(2, 50))
        dense = np.zeros((rows, cols), dtype=np.uint32)
        if rows and cols:
            mask = rng.random((rows, cols)) < rng.random()
            dense[mask] = rng.integers(1, rng_max, size=int(mask.sum()))
        storage = so.CSR if rng.random() < 0.5 else so.CSC
        yield rows, cols, dense, so.AdaptiveMat.from_dense(dense, storage)


def test_sparse_dot_dense_integer_exact():
    # property of sqz/src/mat.rs:1406-1486: sparse product == densified product, integer exact
    rng = np.random.default_rng(0)
    for rows, cols, dense, sparse in _random_matrices(rng, 600, 30):
        q = rng.integers(0, 100, size=(cols, int(rng.integers(0, 64))), dtype=np.uint32)
        assert np.array_equal(sparse.dot(q), (dense.astype(np.uint64) @ q.astype(np.uint64)).astype(np.uint32))
        ql = rng.integers(0, 100, size=(int(rng.integers(0, 64)), rows), dtype=np.uint32)
        assert np.array_equal(sparse.rdot(ql), (ql.astype(np.uint64) @ dense.astype(np.uint64)).astype(np.uint32))
        v = rng.integers(0, 100, size=cols, dtype=np.uint32)
        assert np.array_equal(sparse.dot(v), (dense.astype(np.uint64) @ v.astype(np.uint64)).astype(np.uint32))


def test_low_rank_offset_products():
    # sqz/src/low_rank_offset.rs:145-173
    rng = np.random.default_rng(1)
    for rows, cols, dense, sparse in _random_matrices(rng, 300, 30):
        if rows == 0 or cols == 0:
            continue
        for rank in range(1, 5):
            u = rng.random((rows, rank))
            v = rng.random((rank, cols))
            lr = so.LowRankOffset(sparse.values_into(), u, v)
            d = lr.to_dense()
            q = rng.random((cols, 7))
            assert_close(lr.dot(q), d @ q, rtol=1e-7, atol=1e-9)
            ql = rng.random((5, rows))
            assert_close(lr.rdot(ql), ql @ d, rtol=1e-7, atol=1e-9)
            assert_close(lr.t().to_dense(), d.T)


def _simple_deterministic_ex(m, n):
    x = np.arange(m * n, dtype=np.int64)
    return (x % 7 + x % 4 + x % 50 + x % 47 + x % 12).astype(np.float64).reshape(m, n)


def _test_svd(a, nu, run, thr):
    # TestSvd::test_svd (scan-rs/src/dim_red/test.rs:58-110)
    import scipy.linalg as sl

    dense = a.dot(np.eye(a.shape()[1]))
    _, s_gt, vt_gt = sl.svd(dense, full_matrices=False, lapack_driver="gesvd")
    u, s, v = run(a, nu)
    av = a.dot(v)
    frob = so.frobenius(av - u * s)
    s_err = np.max(np.abs((s - s_gt[:nu]) / s_gt[:nu]))
    av_gt = np.abs(a.dot(vt_gt[:nu, :].T))
    proj_err = np.max(np.abs((np.abs(av) - av_gt) / av_gt))
    assert frob < thr["frob_err_max"]
    assert s_err < thr["s_err_max"]
    assert proj_err < thr["proj_err_max"]


@pytest.mark.parametrize("solver", ["bk", "rand", "irlba"])
def test_svd_drivers_on_deterministic(golden, solver):
    thr = golden["svd_thresholds"]
    for m, n in thr["shapes"]:
        a = so.DenseMat(_simple_deterministic_ex(m, n))
        if solver == "bk":
            run = lambda a, nu: so.BkSvd().run_pca(a, nu)
        elif solver == "rand":
            run = lambda a, nu: so.RandSvd().run_pca(a, nu)
        else:
            run = lambda a, nu: so.irlba(a, nu, 0.00001, 300)[:3]
        _test_svd(a, thr["nu"], run, thr)


def test_svd_bk_on_sparse_lowrankoffset():
    # the normalised LowRankOffset operator through both svd_bk branches
    rng = np.random.default_rng(3)
    dense = rng.poisson(rng.gamma(0.4, 2.0, size=(60, 1)) * np.ones((1, 300))).astype(np.uint32)
    dense[:, 0] += 1
    dense[0, :] += 1
    for storage in (so.CSR, so.CSC):
        a = so.normalize(so.AdaptiveMat.from_dense(dense, storage), "cellranger")
        for op in (a, a.t()):
            u, s, vt = so.svd_bk(op, 5, 10, 5)
            full = np.linalg.svd(op.to_dense(), compute_uv=False)
            assert np.max(np.abs(s - full[:5]) / full[:5]) < 1e-3


def test_smallrng_is_deterministic():
    a = so.omega_panel((3, 5), 0)
    b = so.omega_panel((3, 5), 0)
    assert np.array_equal(a, b) and a.min() >= -1.0 and a.max() < 1.0
    assert not np.array_equal(a, so.omega_panel((3, 5), 1))


def test_config1_plumbing_10k_x_2k():
    # BASELINE.json configs[0]: 10k cells x 2k genes, 5 % nnz, top-10 PCA on the CPU path (no GPU)
    from scanrs_amd.synth import synth_counts

    m = synth_counts(10_000, 2_000, 0.05, 0)
    assert abs(m.nnz / (10_000 * 2_000) - 0.05) < 0.005
    a = so.normalize(so.AdaptiveMat(2_000, 10_000, so.CSC, m.indptr, m.indices, m.data), "cellranger")
    u, s, v = so.BkSvd().run_pca(a, 10)
    assert u.shape == (2_000, 10) and v.shape == (10_000, 10)
    assert np.all(np.diff(s) <= 0)
    assert np.max(np.abs(u.T @ u - np.eye(10))) < 1e-10 and np.max(np.abs(v.T @ v - np.eye(10))) < 1e-10
    # Ritz relation A^T u = s v
    assert np.max(np.abs(a.rdot(u.T).T - v * s)) < 1e-8 * s[0]
