"""sqz::AdaptiveVec (sqz/src/vec.rs): the oracle's restatement of the eight encodings (CPU) and the device decode
behind scanrs_mat_create_adaptive (GPU), modelled on the reference's own property test `test_sparse` /
`exercise_sparse_vec_types` (vec.rs:1379-1514): construct -> iterate returns exactly the nonzero pairs."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import adaptive_vec as av  # noqa: E402


def _random_vec(rng, length, density, vmax, zeros=0.0):
    n = int(round(length * density))
    n = min(length, max(0, n))
    idx = np.sort(rng.choice(length, size=n, replace=False)).astype(np.uint32)
    # mostly small counts with a tail that crosses every overflow threshold (7, 15, 255, 65535)
    val = rng.geometric(0.5, size=n).astype(np.uint64)
    tail = rng.random(n) < 0.08
    val[tail] = rng.integers(1, vmax, size=int(tail.sum()), dtype=np.uint64)
    val = np.minimum(val, vmax).astype(np.uint32)
    if zeros > 0:
        val[rng.random(n) < zeros] = 0  # stored zeros: kept by the sparse encodings, skipped by iteration
    return idx, val


CASES = [(1, 1.0, 5), (20, 0.5, 300), (21, 1.0, 20), (22, 0.3, 70000), (255, 0.1, 16), (256, 0.9, 300), (257, 0.02, 8),
         (1000, 0.0, 5), (5000, 0.03, 100000), (70000, 0.001, 9)]


@pytest.mark.parametrize("kind", av.KINDS)
def test_every_encoding_round_trips(kind):
    rng = np.random.default_rng(av.KIND_CODE[kind])
    for length, density, vmax in CASES:
        for zeros in (0.0, 0.2):
            idx, val = _random_vec(rng, length, density, vmax, zeros)
            v = av.AdaptiveVec.with_kind(kind, length, val, idx)
            pos, out = v.iter()
            keep = val != 0
            assert np.array_equal(pos, idx[keep]) and np.array_equal(out, val[keep]), (kind, length, density)
            assert v.nnz() == len(idx)


def test_choose_storage_follows_the_size_estimates():
    rng = np.random.default_rng(3)
    # nothing stored: V is free (0 bytes)
    assert av.choose_storage(1000, np.zeros(0, dtype=np.uint32)) == "V"
    # every position occupied with small counts: 3 bits per position wins
    assert av.choose_storage(2100, rng.integers(1, 6, size=2100).astype(np.uint32)) == "D3"
    # ... with counts up to 14: nibbles
    assert av.choose_storage(2100, rng.integers(7, 14, size=2100).astype(np.uint32)) == "D4"
    assert av.choose_storage(2100, rng.integers(20, 200, size=2100).astype(np.uint32)) == "D8"
    assert av.choose_storage(2100, rng.integers(300, 60000, size=2100).astype(np.uint32)) == "D16"
    # 3 % occupancy, small counts (the scRNA case, sqz/src/lib.rs:5-8): compressed-index sparse
    assert av.choose_storage(33000, rng.integers(1, 5, size=1000).astype(np.uint32)) == "S3"
    assert av.choose_storage(33000, rng.integers(8, 14, size=1000).astype(np.uint32)) == "S4"
    # S8 only survives when V (8 bytes per entry) is not below the best of the *other* estimates: the S8 branch does not
    # lower min_size (vec.rs:1120-1123), so at 3 % occupancy V wins although S8 would be smaller ...
    assert av.choose_storage(33000, rng.integers(30, 200, size=1000).astype(np.uint32)) == "V"
    # ... and at 30 % occupancy S8 is kept (D8 = 33000 bytes is the bar V has to beat)
    assert av.choose_storage(33000, rng.integers(30, 200, size=10000).astype(np.uint32)) == "S8"
    # a handful of entries in a long vector: the block table alone costs more than 8 bytes per entry
    assert av.choose_storage(1_000_000, np.array([3, 9, 1], dtype=np.uint32)) == "V"


def test_block_starts_layout_matches_the_constructor_walk():
    # vec.rs:335-397 walked by hand: entries in 256-blocks 0, 0, 2 of a 1000-long vector (4 blocks)
    v = av.AdaptiveVec.with_kind("S4", 1000, [1, 2, 3], [5, 200, 600])
    assert v.block_starts.tolist() == [0, 2, 2, 3, 3]
    assert v.index_bytes.tolist() == [5, 200, 600 % 256]
    # first entry in block 3
    v = av.AdaptiveVec.with_kind("S3", 1024, [9], [3 * 256 + 7])
    assert v.block_starts.tolist() == [0, 0, 0, 0, 1]
    assert v.dense.fallback_indexes.tolist() == [0] and v.dense.fallback_values.tolist() == [9]  # keyed by entry number
    # empty vector
    v = av.AdaptiveVec.with_kind("S8", 300, [], [])
    assert v.block_starts.tolist() == [0, 0, 0]


def test_dense_layouts_bit_for_bit():
    # Dense4: low nibble = even position (vec.rs:845-857); 15 marks the fallback
    v = av.AdaptiveVec.with_kind("D4", 5, [3, 20, 7], [0, 1, 4])
    assert v.dense.data.tolist() == [3 | (15 << 4), 0, 7]
    assert v.dense.fallback_indexes.tolist() == [1] and v.dense.fallback_values.tolist() == [20]
    # Dense3: 21 fields per u64, field i at bits 3i (vec.rs:993-1004)
    v = av.AdaptiveVec.with_kind("D3", 23, [5, 7, 1], [0, 20, 22])
    assert v.dense.data.tolist() == [5 | (7 << 60), 1 << 3]
    assert v.dense.fallback_values.tolist() == [7]
    # DenseW<u8>: 255 marks the fallback, values >= 255 go there (vec.rs:737-746)
    v = av.AdaptiveVec.with_kind("D8", 4, [254, 255, 1000], [0, 1, 3])
    assert v.dense.data.tolist() == [254, 255, 0, 255]
    assert v.dense.fallback_values.tolist() == [255, 1000]


# ---- device decode ----------------------------------------------------------------------------------------------------
def _matrix_vectors(rng, n_outer, n_inner, kind):
    rows = []
    dense = np.zeros((n_outer, n_inner), dtype=np.uint32)
    for o in range(n_outer):
        density = [0.0, 0.004, 0.03, 0.3, 1.0][o % 5]
        idx, val = _random_vec(rng, n_inner, density, 100000, zeros=0.1 if o % 3 == 0 else 0.0)
        v = av.AdaptiveVec.new(n_inner, val, idx) if kind is None else av.AdaptiveVec.with_kind(kind, n_inner, val, idx)
        rows.append(v)
        dense[o, idx] = val
    return rows, dense


@pytest.mark.gpu
@pytest.mark.parametrize("kind", list(av.KINDS) + [None])
def test_device_decode_matches_the_encoded_vectors(kind):
    import scanrs_amd as sa

    rng = np.random.default_rng(11 if kind is None else 100 + av.KIND_CODE[kind])
    for n_outer, n_inner in ((1, 1), (7, 21), (23, 300), (40, 5000), (3, 70001)):
        vecs, dense = _matrix_vectors(rng, n_outer, n_inner, kind)
        if kind is None:
            assert len({v.kind for v in vecs}) >= 1
        for storage in (sa.CSR, sa.CSC):
            rows, cols = (n_outer, n_inner) if storage == sa.CSR else (n_inner, n_outer)
            g = sa.AdaptiveMat.from_adaptive_vecs(rows, cols, storage, [v.pieces() for v in vecs])
            want = dense if storage == sa.CSR else dense.T
            assert tuple(g.shape()) == (rows, cols)
            assert g.nnz() == int(np.count_nonzero(dense))
            assert np.array_equal(g.to_dense().astype(np.uint64), want.astype(np.uint64))
            # integer-exact product through the decoded handle (sqz/src/mat.rs:1406-1486)
            q = rng.integers(0, 50, size=(cols, 3)).astype(np.uint32)
            ref = (want.astype(np.uint64) @ q.astype(np.uint64)).astype(np.uint32)
            assert np.array_equal(g.dot(q), ref)


@pytest.mark.gpu
def test_device_decode_of_the_synthetic_matrix_equals_the_triplet_path():
    import scanrs_amd as sa
    from scanrs_amd.synth import synth_counts

    m = synth_counts(3000, 700, 0.05, 5)  # cells x genes CSR
    vecs = av.from_csmat(m.shape[0], m.shape[1], m.indptr, m.indices.astype(np.uint32), m.data.astype(np.uint32))
    kinds = {v.kind for v in vecs}
    assert kinds <= set(av.KINDS) and len(kinds) >= 1
    a = sa.AdaptiveMat.from_adaptive_vecs(m.shape[1], m.shape[0], sa.CSC, [v.pieces() for v in vecs])
    b = sa.AdaptiveMat.from_csmat(m.shape[1], m.shape[0], sa.CSC, m.indptr, m.indices, m.data)
    assert a.nnz() == b.nnz()
    assert np.array_equal(a.sum_axis(0, np.uint32), b.sum_axis(0, np.uint32)) and np.array_equal(a.sum_axis(1, np.uint32), b.sum_axis(1, np.uint32))
    ua, sa_, va = sa.BkSvd().run_pca(sa.normalize(a, sa.Normalization.CellRanger), 6)
    ub, sb, vb = sa.BkSvd().run_pca(sa.normalize(b, sa.Normalization.CellRanger), 6)
    assert np.array_equal(sa_, sb) and np.array_equal(ua, ub) and np.array_equal(va, vb)  # same triplet -> same bits
    compressed = sum(v.mem_size() for v in vecs)
    assert compressed < 0.75 * 8 * m.nnz  # the point of handing the encoded buffers over


@pytest.mark.gpu
def test_device_decode_rejects_malformed_vectors():
    import scanrs_amd as sa

    v = av.AdaptiveVec.with_kind("S4", 300, [1, 2], [3, 280]).pieces()
    with pytest.raises(sa.ScanrsError):
        sa.AdaptiveMat.from_adaptive_vecs(2, 300, sa.CSR, [v])  # one vector for two rows
    with pytest.raises(sa.ScanrsError):
        sa.AdaptiveMat.from_adaptive_vecs(1, 299, sa.CSR, [v])  # length differs from the inner dimension
    bad = dict(v)
    bad["block_starts"] = v["block_starts"][:-1]
    with pytest.raises(sa.ScanrsError):
        sa.AdaptiveMat.from_adaptive_vecs(1, 300, sa.CSR, [bad])
    bad = dict(v)
    bad["kind"] = 9
    with pytest.raises(sa.ScanrsError):
        sa.AdaptiveMat.from_adaptive_vecs(1, 300, sa.CSR, [bad])
    bad = dict(av.AdaptiveVec.with_kind("D8", 300, [1, 2], [3, 280]).pieces())
    bad["data"] = bad["data"][:100]
    with pytest.raises(sa.ScanrsError):
        sa.AdaptiveMat.from_adaptive_vecs(1, 300, sa.CSR, [bad])
