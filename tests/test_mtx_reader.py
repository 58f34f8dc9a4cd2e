"""`load_mtx` (scan-rs/src/mtx.rs:10-51) through the C ABI's scanrs_mtx_read: gzipped MatrixMarket coordinate files ->
CSR arrays. Host-only (no device needed to read); the device hand-over is covered by tests/test_gpu_cli.py."""
import gzip

import numpy as np
import pytest

import scanrs_amd as sa
from scanrs_amd import mtx


def write(path, text, members=1):
    data = text.encode()
    with open(path, "wb") as f:
        step = (len(data) + members - 1) // members
        for i in range(members):  # concatenated gzip members, what MultiGzDecoder reads
            f.write(gzip.compress(data[i * step:(i + 1) * step]))
    return str(path)


def test_load_mtx_triplets_duplicates_and_order(tmp_path):
    rng = np.random.default_rng(2)
    nrow, ncol = 17, 23
    dense = np.zeros((nrow, ncol), dtype=np.int64)
    lines = ["%%MatrixMarket matrix coordinate integer general", "% a comment", f"{nrow} {ncol} 120"]
    for _ in range(120):  # random order, with repeats: TriMat::to_csr sums them
        r, c, v = int(rng.integers(nrow)), int(rng.integers(ncol)), int(rng.integers(1, 50))
        dense[r, c] += v
        lines.append(f"{r + 1}\t{c + 1}   {v}")
    lines.insert(10, "% comment in the middle")
    m = mtx.load_mtx(write(tmp_path / "a.mtx.gz", "\n".join(lines) + "\n", members=3))
    assert (m.rows, m.cols, m.storage) == (nrow, ncol, sa.CSR)
    assert m.indptr.dtype == np.uint64 and m.indices.dtype == np.uint32 and m.values.dtype == np.uint32
    np.testing.assert_array_equal(m.to_dense(), dense)
    assert m.nnz == int((dense != 0).sum())
    for r in range(nrow):  # ascending, no duplicates left
        s, e = int(m.indptr[r]), int(m.indptr[r + 1])
        assert np.all(np.diff(m.indices[s:e].astype(np.int64)) > 0)
    assert m.barcodes == [] and m.feature_ids == []


def test_load_mtx_edge_cases(tmp_path):
    m = mtx.load_mtx(write(tmp_path / "empty.mtx.gz", "%%MatrixMarket\n3 4 0\n"))
    assert (m.rows, m.cols, m.nnz) == (3, 4, 0) and list(m.indptr) == [0, 0, 0, 0]
    m = mtx.load_mtx(write(tmp_path / "big.mtx.gz", "2 2 2\n1 1 4294967295\n2 2 +7\n"))
    assert list(m.values) == [4294967295, 7]
    m = mtx.load_mtx(write(tmp_path / "wrap.mtx.gz", "1 1 2\n1 1 4294967295\n1 1 2\n"))  # u32 sum wraps like a release build
    assert list(m.values) == [1]
    long_line = "1 1 5" + " " * 200_000 + "\n"  # longer than the read buffer
    m = mtx.load_mtx(write(tmp_path / "long.mtx.gz", "1 1 1\n" + long_line))
    assert list(m.values) == [5]


@pytest.mark.parametrize("text,why", [
    ("% only comments\n", "no matrix found"),
    ("3 4\n", "no NNZ"),
    ("3\n", "no NCOL"),
    ("3 4 1\n1 2\n", "missing VAL"),
    ("3 4 1\n1\n", "missing COL"),
    ("3 4 1\n\n", "missing ROW"),                 # a blank line is not skipped by the reference either
    ("3 4 1\n1 2 1.5\n", "invalid digit"),        # parse::<u32>
    ("3 4 1\n1 2 -1\n", "invalid digit"),
    ("3 4 1\n1 2 4294967296\n", "invalid digit"),  # overflows u32
    ("3 4 1\nx 2 1\n", "invalid digit"),
    ("3 4 1\n0 2 1\n", "outside"),                # where the reference panics on the subtraction / add_triplet
    ("3 4 1\n4 2 1\n", "outside"),
    ("3 4 1\n1 5 1\n", "outside"),
])
def test_load_mtx_errors(tmp_path, text, why):
    with pytest.raises(sa.ScanrsError, match=why):
        mtx.load_mtx(write(tmp_path / "bad.mtx.gz", text))


def test_load_mtx_missing_file(tmp_path):
    with pytest.raises(sa.ScanrsError, match="missing.mtx.gz"):
        mtx.load_mtx(str(tmp_path / "missing.mtx.gz"))
