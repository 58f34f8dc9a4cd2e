"""ORACLE — TEST INFRASTRUCTURE ONLY (not a product path, never a fallback).

CPU restatement of the scan-rs (10XGenomics/scan-rs) normalize -> PCA hot path:
the lazy `MatrixMap` chain, `sum_axis`/`mean_axis`, `scale_and_center`,
`LowRankOffset` products and the `svd_bk` / `svd_rand` / `irlba` drivers.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The sparse loops are the plain-C restatement in
`oracle/csrc/oracle_kernels.c` (serial, single thread, same loop order as
`sqz/src/prod.rs`); dense QR / SVD go through numpy/scipy's LAPACK
(`geqrf+orgqr`, `gesdd`, `gesvd`), standing in for the reference's
ndarray-linalg 0.17 -> lax 0.17 -> MKL-sequential calls (`Cargo.lock`; the MKL
build is a git dependency absent from /root/reference).

Parity pin: every function below is checked in `tests/test_oracle_golden.py`
against the reference's own inline known-answer tables (transcribed as data in
`tests/golden/reference_tables.json`).  The seeded-`SmallRng` panel Omega is
"parity unpinned": `rand 0.10.1` is a crates.io dependency whose source is not
under /root/reference; `SmallRng`/`Uniform` below restate the published
xoshiro256++ / SplitMix64 / [1,2)-mantissa algorithms of the rand family and
are used identically by the oracle and the HIP path, and every reference test
on this path compares against an exact SVD, not an Omega-dependent value.

All `file:line` citations are relative to the reference checkout.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_SRC_PATH = os.path.join(_HERE, "csrc", "oracle_kernels.c")

CSR = 0  # same u8 codes as sqz/src/mat.rs:45-65
CSC = 1

OP_INTO, OP_SCALE_AXIS, OP_LN_1P, OP_LOG2_1P, OP_LOG10_1P, OP_SQUARE, OP_BINOM_DEV, OP_BINOM_PEARSON = range(8)


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, -ffp-contract=off so `o + r*lval` is not fused)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(_SRC_PATH):
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", _LIB_PATH, _SRC_PATH, "-lm"]
        )
    return _LIB_PATH


class _COp(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("axis", ctypes.c_int32),
        ("swap", ctypes.c_int32),
        ("_pad", ctypes.c_int32),
        ("a", ctypes.c_void_p),
        ("b", ctypes.c_void_p),
    ]


_lib = None


def _clib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


_THREADS = 1
_OTHER_COPY = {}  # all-core mode only: (indptr address, indices address) -> the other orientation of the same triplet


def set_threads(t: int) -> int:
    """Threads of the sparse loops (bench.py's all-core CPU baseline). 1 = the serial reference order (default, the
    mode every parity pin is taken in); returns the thread count in force. With more than one thread the scatter form
    of the product (CSC storage, sqz/src/prod.rs:190-214) is served by the gather loop on a transposed copy of the
    matrix built once per matrix (what a rayon port would do: output rows dealt to threads, no write conflicts)."""
    global _THREADS
    lib = _clib()
    lib.oracle_set_threads(ctypes.c_int(int(t)))
    _THREADS = int(lib.oracle_get_threads())
    if _THREADS == 1:
        _OTHER_COPY.clear()
    return _THREADS


def max_threads() -> int:
    return int(_clib().oracle_max_threads())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class MapOp:
    """One link of a `ComposedMap` chain (sqz/src/matrix_map.rs:145-197)."""

    kind: int
    axis: int = 0
    a: Optional[np.ndarray] = None
    b: Optional[np.ndarray] = None
    swap: bool = False  # under an odd number of TransposeMap wrappers (matrix_map.rs:42-80)

    def transposed(self) -> "MapOp":
        return MapOp(self.kind, self.axis, self.a, self.b, not self.swap)


def _c_ops(ops: List[MapOp]):
    arr = (_COp * max(1, len(ops)))()
    keep = []
    for i, op in enumerate(ops):
        a = None if op.a is None else np.ascontiguousarray(op.a, dtype=np.float64)
        b = None if op.b is None else np.ascontiguousarray(op.b, dtype=np.float64)
        keep += [a, b]
        arr[i].kind, arr[i].axis, arr[i].swap = op.kind, op.axis, int(op.swap)
        arr[i].a = None if a is None else a.ctypes.data
        arr[i].b = None if b is None else b.ctypes.data
    return arr, len(ops), keep


class AdaptiveMat:
    """`sqz::AdaptiveMat<N, D, M>` (sqz/src/mat.rs:34-42) with the `AdaptiveVec`
    encodings expanded to plain (index, value) pairs (decode semantics of
    `AbsIter`, sqz/src/vec.rs:100-117: ascending index, stored zeros skipped)."""

    def __init__(self, rows, cols, storage, indptr, indices, values, ops: Optional[List[MapOp]] = None):
        self.rows, self.cols, self.storage = int(rows), int(cols), int(storage)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.values = np.ascontiguousarray(values, dtype=np.uint32)
        self.ops: List[MapOp] = [MapOp(OP_INTO)] if ops is None else list(ops)
        n_outer = self.rows if self.storage == CSR else self.cols
        assert self.indptr.shape[0] == n_outer + 1

    # -- constructors ------------------------------------------------------
    @staticmethod
    def from_dense(dense: np.ndarray, storage: int = CSR) -> "AdaptiveMat":
        """`AdaptiveMat::from_dense` (sqz/src/mat.rs:122-150), CSR by default."""
        dense = np.asarray(dense)
        rows, cols = dense.shape
        src = dense if storage == CSR else dense.T
        indptr, indices, values = [0], [], []
        for line in src:
            nz = np.nonzero(line)[0]
            indices.extend(nz.tolist())
            values.extend(line[nz].tolist())
            indptr.append(len(indices))
        return AdaptiveMat(rows, cols, storage, indptr, np.array(indices, dtype=np.uint32), np.array(values, dtype=np.uint32))

    @staticmethod
    def from_scipy(m) -> "AdaptiveMat":
        import scipy.sparse as sp

        if sp.isspmatrix_csc(m):
            st = CSC
        else:
            m = m.tocsr()
            st = CSR
        m.sort_indices()
        return AdaptiveMat(m.shape[0], m.shape[1], st, m.indptr, m.indices, m.data)

    # -- shape / views -----------------------------------------------------
    def shape(self):
        return [self.rows, self.cols]

    @property
    def n_outer(self):
        return self.rows if self.storage == CSR else self.cols

    def view(self) -> "AdaptiveMat":
        return AdaptiveMat(self.rows, self.cols, self.storage, self.indptr, self.indices, self.values, self.ops)

    def t(self) -> "AdaptiveMat":
        """`AdaptiveMat::t` (sqz/src/mat.rs:262-270): flip the storage flag, wrap the map."""
        return AdaptiveMat(
            self.cols, self.rows, 1 - self.storage, self.indptr, self.indices, self.values, [o.transposed() for o in self.ops]
        )

    # -- lazy maps ---------------------------------------------------------
    def set_map(self, ops: List[MapOp]) -> "AdaptiveMat":  # mat.rs:892-898
        return AdaptiveMat(self.rows, self.cols, self.storage, self.indptr, self.indices, self.values, ops)

    def compose_map(self, op: MapOp) -> "AdaptiveMat":  # mat.rs:901-913
        return self.set_map(self.ops + [op])

    def values_into(self) -> "AdaptiveMat":  # mat.rs:916-922
        return self.compose_map(MapOp(OP_INTO))

    def apply(self, kind: int) -> "AdaptiveMat":  # mat.rs:925-933 (ScalarMap, f(0) == 0)
        return self.compose_map(MapOp(kind))

    # -- reductions --------------------------------------------------------
    def _call_args(self):
        arr, n, keep = _c_ops(self.ops)
        return arr, n, keep

    def sum_axis(self, axis: int, dtype=np.float64) -> np.ndarray:
        """`sum_axis` (sqz/src/mat.rs:377-406). axis 0 -> length cols, axis 1 -> length rows."""
        sz = self.cols if axis == 0 else self.rows
        if dtype == np.uint32:
            assert all(o.kind == OP_INTO for o in self.ops), "u32 sums only on the raw count matrix"
            out = np.zeros(sz, dtype=np.uint32)
            _clib().oracle_sum_axis_u32(
                ctypes.c_int(self.storage), ctypes.c_size_t(self.n_outer), _ptr(self.indptr), _ptr(self.indices),
                _ptr(self.values), ctypes.c_int(axis), _ptr(out))
            return out
        out = np.zeros(sz, dtype=np.float64)
        arr, n, _keep = self._call_args()
        _clib().oracle_sum_axis_f64(
            ctypes.c_int(self.storage), ctypes.c_size_t(self.n_outer), _ptr(self.indptr), _ptr(self.indices),
            _ptr(self.values), arr, ctypes.c_int(n), ctypes.c_int(axis), _ptr(out))
        return out

    def mean_axis(self, axis: int) -> np.ndarray:  # mat.rs:273-276
        m = float(self.shape()[axis])
        return self.sum_axis(axis) / m

    def mean_var_axis(self, axis: int):  # mat.rs:285-330
        sz = self.cols if axis == 0 else self.rows
        s = np.zeros(sz)
        s2 = np.zeros(sz)
        arr, n, _keep = self._call_args()
        _clib().oracle_sum_sq_axis_f64(
            ctypes.c_int(self.storage), ctypes.c_size_t(self.n_outer), _ptr(self.indptr), _ptr(self.indices),
            _ptr(self.values), arr, ctypes.c_int(n), ctypes.c_int(axis), _ptr(s), _ptr(s2))
        m = float(self.shape()[axis])
        mean = s / m
        var = s2 / m - mean**2
        return mean, var

    def var_axis(self, axis: int):
        return self.mean_var_axis(axis)[1]

    # -- products (sqz/src/prod.rs, Dot impls sqz/src/mat.rs:1074-1170) ------
    def dot(self, rhs: np.ndarray) -> np.ndarray:
        rhs = np.asarray(rhs)
        one_d = rhs.ndim == 1
        if one_d:  # mat.rs:1092-1112
            rhs = rhs.reshape(-1, 1)
        assert rhs.shape[0] == self.cols, "Dimension mismatch"
        l = rhs.shape[1]
        if rhs.dtype == np.uint32:
            assert all(o.kind == OP_INTO for o in self.ops)
            rhs_c = np.ascontiguousarray(rhs)
            out = np.zeros((self.rows, l), dtype=np.uint32)
            _clib().oracle_spmm_u32(
                ctypes.c_int(self.storage), ctypes.c_size_t(self.n_outer), _ptr(self.indptr), _ptr(self.indices),
                _ptr(self.values), _ptr(rhs_c), ctypes.c_size_t(l), _ptr(out))
        else:
            rhs_c = np.ascontiguousarray(rhs, dtype=np.float64)
            out = np.zeros((self.rows, l), dtype=np.float64)
            arr, n, _keep = self._call_args()
            if _THREADS > 1 and self.storage == CSC:  # all-core baseline: gather on the transposed copy, rows dealt to threads
                ip2, ix2, vv2 = self.other_copy()
                _clib().oracle_spmm_f64(
                    ctypes.c_int(CSR), ctypes.c_size_t(self.rows), _ptr(ip2), _ptr(ix2), _ptr(vv2), arr, ctypes.c_int(n),
                    _ptr(rhs_c), ctypes.c_size_t(l), _ptr(out))
            else:
                _clib().oracle_spmm_f64(
                    ctypes.c_int(self.storage), ctypes.c_size_t(self.n_outer), _ptr(self.indptr), _ptr(self.indices),
                    _ptr(self.values), arr, ctypes.c_int(n), _ptr(rhs_c), ctypes.c_size_t(l), _ptr(out))
        return out[:, 0] if one_d else out

    def other_copy(self):
        """(indptr, indices, values) of the same matrix in the other orientation (all-core baseline only; cached per triplet)."""
        import scipy.sparse as sp

        key = (self.indptr.ctypes.data, self.indices.ctypes.data)
        if key not in _OTHER_COPY:
            cls = sp.csr_matrix if self.storage == CSR else sp.csc_matrix
            m = cls((self.values, self.indices.astype(np.int64), self.indptr.astype(np.int64)), shape=(self.rows, self.cols))
            o = m.tocsc() if self.storage == CSR else m.tocsr()
            o.sort_indices()
            _OTHER_COPY[key] = (np.ascontiguousarray(o.indptr, dtype=np.uint64), np.ascontiguousarray(o.indices, dtype=np.uint32),
                                np.ascontiguousarray(o.data, dtype=np.uint32), self.indptr, self.indices)  # keep the key's arrays alive
        return _OTHER_COPY[key][:3]

    def rdot(self, lhs: np.ndarray) -> np.ndarray:
        """`lhs.dot(&self)` (sqz/src/mat.rs:1124-1132): transpose both, run the other
        storage kernel, `reversed_axes()` the result."""
        lhs = np.asarray(lhs)
        if lhs.ndim == 1:  # mat.rs:1150-1170
            return self.t().dot(lhs)
        return self.t().dot(np.ascontiguousarray(lhs.T)).T

    def to_dense(self) -> np.ndarray:
        """`to_dense` (mat.rs:150-200): map applied to stored nonzeros only."""
        vals = np.zeros(self.values.shape[0], dtype=np.float64)
        arr, n, _keep = self._call_args()
        _clib().oracle_map_values(
            ctypes.c_int(self.storage), ctypes.c_size_t(self.n_outer), _ptr(self.indptr), _ptr(self.indices),
            _ptr(self.values), arr, ctypes.c_int(n), _ptr(vals))
        out = np.zeros((self.rows, self.cols))
        outer = np.repeat(np.arange(self.n_outer), np.diff(self.indptr).astype(np.int64))
        nz = self.values != 0
        if self.storage == CSR:
            out[outer[nz], self.indices[nz]] = vals[nz]
        else:
            out[self.indices[nz], outer[nz]] = vals[nz]
        return out

    # -- centre / scale (sqz/src/mat.rs:937-1001) ----------------------------
    def center(self, axis: int, m: Optional[np.ndarray] = None) -> "LowRankOffset":
        neg_means = -(self.mean_axis(axis) if m is None else np.asarray(m, dtype=np.float64))
        if axis == 0:
            u = np.ones((self.rows, 1))
            v = neg_means.reshape(1, self.cols)
        else:
            u = neg_means.reshape(self.rows, 1)
            v = np.ones((1, self.cols))
        return LowRankOffset(self.values_into(), u, v)

    def scale(self, axis: int, s: Optional[np.ndarray] = None) -> "AdaptiveMat":
        if s is not None:
            factors = 1.0 / np.asarray(s, dtype=np.float64)
        else:
            means_sq = self.mean_axis(axis) ** 2
            sq_means = self.view().apply(OP_SQUARE).mean_axis(axis)
            d = sq_means - means_sq
            with np.errstate(divide="ignore", invalid="ignore"):
                factors = np.where(d == 0.0, 1.0, 1.0 / np.sqrt(d))
        return self.compose_map(MapOp(OP_SCALE_AXIS, axis=1 - axis, a=factors))

    def scale_and_center(self, axis: int, scaling_factors: Optional[np.ndarray] = None) -> "LowRankOffset":
        means = self.mean_axis(axis)
        if scaling_factors is None:
            matsq_means = self.view().apply(OP_SQUARE).mean_axis(axis)
            d = matsq_means - means**2
            with np.errstate(invalid="ignore"):
                scaling_factors = np.where(d <= 0.0, 1.0, np.sqrt(np.where(d <= 0.0, 1.0, d)))
        scaling_factors = np.asarray(scaling_factors, dtype=np.float64)
        means = means / scaling_factors
        return self.scale(axis, scaling_factors).center(axis, means)


class LowRankOffset:
    """`sqz::LowRankOffset` = mat + u*v (sqz/src/low_rank_offset.rs:12-96)."""

    def __init__(self, mat: AdaptiveMat, u: np.ndarray, v: np.ndarray):
        assert mat.rows == u.shape[0] and mat.cols == v.shape[1] and u.shape[1] == v.shape[0]
        self.mat, self.u, self.v = mat, np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64)

    def rows(self):
        return self.mat.rows

    def cols(self):
        return self.mat.cols

    def shape(self):
        return [self.mat.rows, self.mat.cols]

    def inner_sparse(self):
        return self.mat

    def to_dense(self):  # :55-57
        return self.u @ self.v + self.mat.to_dense()

    def t(self):  # :60-65
        return LowRankOffset(self.mat.t(), self.v.T.copy(), self.u.T.copy())

    def dot(self, rhs):  # :76-80
        res = self.mat.dot(rhs)
        res = res + self.u @ (self.v @ rhs)
        return res

    def rdot(self, lhs):  # :91-95
        res = self.mat.rdot(lhs)
        res = res + (lhs @ self.u) @ self.v
        return res


class DenseMat:
    """Dense `Array2<f64>` behind the same operator surface (dim_red/mod.rs:55-65)."""

    def __init__(self, a):
        self.a = np.asarray(a, dtype=np.float64)

    def shape(self):
        return list(self.a.shape)

    def dot(self, rhs):
        return self.a @ rhs

    def rdot(self, lhs):
        return lhs @ self.a


# ---------------------------------------------------------------------------
# scan-rs/src/stats.rs:13-38
def median_mut(xs: np.ndarray):
    """Sort-based median; even length -> `(a + b) / 2` in the element type (integer floor)."""
    if xs.shape[0] == 0:
        return None
    xs.sort()
    n = xs.shape[0]
    if n % 2 == 0:
        if np.issubdtype(xs.dtype, np.integer):
            return xs.dtype.type((int(xs[n // 2]) + int(xs[n // 2 - 1])) % (1 << (8 * xs.dtype.itemsize)) // 2)
        return (xs[n // 2] + xs[n // 2 - 1]) / 2
    return xs[n // 2]


# ---------------------------------------------------------------------------
# scan-rs/src/normalization.rs
LOG_E, LOG_TWO, LOG_TEN = OP_LN_1P, OP_LOG2_1P, OP_LOG10_1P


def log_normalize_with_size_factor(matrix: AdaptiveMat, umi_count_sum, log_base, size_factors=None) -> AdaptiveMat:
    """normalization.rs:138-178."""
    if size_factors is not None:
        normalization_counts = np.asarray(size_factors, dtype=np.uint32)
        assert normalization_counts.shape[0] == matrix.cols
    else:
        normalization_counts = matrix.sum_axis(0, np.uint32)
    umi_counts = matrix.sum_axis(0, np.uint32)
    if umi_count_sum is not None:
        target = float(umi_count_sum)
    else:
        med = median_mut(umi_counts)
        target = 1.0 if med is None else max(float(med), 1.0)
    with np.errstate(divide="ignore"):
        col_scales = target / normalization_counts.astype(np.float64)
    return matrix.compose_map(MapOp(OP_SCALE_AXIS, axis=1, a=col_scales)).apply(log_base)


def log_normalize(matrix, umi_count_sum, log_base):  # :119-129
    return log_normalize_with_size_factor(matrix, umi_count_sum, log_base, None)


def normalize(mat: AdaptiveMat, norm: str) -> LowRankOffset:
    """normalization.rs:46-69."""
    if norm == "cellranger":
        return log_normalize(mat, None, LOG_TWO).scale_and_center(1, None)
    if norm == "cellranger8":
        return log_normalize(mat, None, LOG_TWO).scale_and_center(1, np.ones(mat.rows))
    if norm == "seuratlog":
        return log_normalize(mat, 10_000.0, LOG_E).scale_and_center(1, None)
    raise ValueError("not implemented")


def normalize_with_size_factor(mat: AdaptiveMat, norm: str, size_factors=None) -> LowRankOffset:
    """normalization.rs:72-102."""
    if norm == "cellranger":
        return log_normalize_with_size_factor(mat, None, LOG_TWO, None).scale_and_center(1, None)
    if norm == "cellranger8":
        return log_normalize_with_size_factor(mat, None, LOG_TWO, None).scale_and_center(1, np.ones(mat.rows))
    if norm == "seuratlog":
        return log_normalize_with_size_factor(mat, 10_000.0, LOG_E, None).scale_and_center(1, None)
    if norm == "withsizefactors":
        return log_normalize_with_size_factor(mat, None, LOG_TWO, size_factors).scale_and_center(1, None)
    if norm == "logtransform":
        ones = np.ones(mat.cols, dtype=np.uint32)
        return log_normalize_with_size_factor(mat, 1.0, LOG_TWO, ones).scale_and_center(1, None)
    raise ValueError("not implemented")


def log1p_normalize_fixed_point(matrix: AdaptiveMat, log_base, base: int, exponent: int) -> LowRankOffset:
    """normalization.rs:191-213."""
    factors = np.ones(matrix.cols) / float(base**exponent)
    return matrix.compose_map(MapOp(OP_SCALE_AXIS, axis=1, a=factors)).apply(log_base).scale_and_center(1, None)


def fit_multinomial_model(matrix: AdaptiveMat):
    """normalization.rs:218-227."""
    n = matrix.sum_axis(0)
    total = n.sum()
    pi = matrix.sum_axis(1) / total
    return n, pi


def binom_deviance_resid(matrix: AdaptiveMat) -> LowRankOffset:
    """normalization.rs:232-259."""
    n, pi = fit_multinomial_model(matrix)
    u = np.sqrt(np.log(1.0 / (1.0 - pi))).reshape(matrix.rows, 1)
    v = (-np.sqrt(2.0 * n)).reshape(1, matrix.cols)
    return LowRankOffset(matrix.set_map([MapOp(OP_BINOM_DEV, a=n, b=pi)]), u, v)


def binom_pearson_resid(matrix: AdaptiveMat) -> LowRankOffset:
    """normalization.rs:306-322."""
    n, pi = fit_multinomial_model(matrix)
    u = np.sqrt(pi / (1.0 - pi)).reshape(matrix.rows, 1)
    v = (-np.sqrt(n)).reshape(1, matrix.cols)
    return LowRankOffset(matrix.set_map([MapOp(OP_BINOM_PEARSON, a=n, b=pi)]), u, v)


# ---------------------------------------------------------------------------
# rand-family generator used for the seeded panel Omega ("parity unpinned", see header).
_M64 = (1 << 64) - 1


class SmallRng:
    """xoshiro256++ seeded through SplitMix64 (`SmallRng::seed_from_u64`, call sites
    dim_red/bk_svd.rs:83, rand_svd.rs:77)."""

    def __init__(self, seed: int):
        state = seed & _M64
        s = []
        for _ in range(4):
            state = (state + 0x9E3779B97F4A7C15) & _M64
            z = state
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            s.append(z ^ (z >> 31))
        self.s = s

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & _M64

    def next_u64(self) -> int:
        s = self.s
        result = (self._rotl((s[0] + s[3]) & _M64, 23) + s[0]) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def uniform_m1_1(self, count: int) -> np.ndarray:
        """`Uniform::new(-1.0, 1.0).sample` (bk_svd.rs:84): 52 mantissa bits -> [1,2) - 1, * 2 + (-1)."""
        raw = np.fromiter((self.next_u64() for _ in range(count)), dtype=np.uint64, count=count)
        bits = (raw >> np.uint64(12)) | np.uint64(0x3FF0000000000000)
        v12 = bits.view(np.float64)
        return (v12 - 1.0) * 2.0 + (-1.0)


    def normal(self, count: int) -> np.ndarray:
        """Standard normals by Box-Muller on the same stream, one value per pair of draws (u1 from the top 53 bits + 1/2 ulp,
        cos branch only) — the rule the product's default IRLBA start vector uses (solver.cpp SmallRng::normal). The
        reference draws `Normal` through rand_distr 0.6's ziggurat (irlba.rs:107-112), whose tables are not under
        /root/reference: parity unpinned for the DEFAULT start vector; tests pass v0 explicitly or compare results that
        do not depend on it."""
        out = np.empty(count)
        for i in range(count):
            while True:
                u1 = ((self.next_u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0)
                if u1 > 0.0:
                    break
            u2 = ((self.next_u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0)
            out[i] = np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)
        return out


def omega_panel(shape, seed: int = 0) -> np.ndarray:
    """`Array2::from_shape_simple_fn(shape, || unif.sample(&mut rng))`: row-major fill order."""
    return SmallRng(seed).uniform_m1_1(int(shape[0]) * int(shape[1])).reshape(shape)


# ---------------------------------------------------------------------------
# dense LAPACK stand-ins (ndarray-linalg call sites: bk_svd.rs:94,98,105,123,127,134)
def _qr_q(a: np.ndarray) -> np.ndarray:
    return np.linalg.qr(a, mode="reduced")[0]


def _svddc_some(a: np.ndarray):
    import scipy.linalg as sl

    return sl.svd(a, full_matrices=False, lapack_driver="gesdd")


class CancellationError(Exception):
    """snoop/src/lib.rs:5-18."""


def svd_bk(A, k: int, b: int, n_iter: int, seed: int = 0, omega: Optional[np.ndarray] = None, snoop=None):
    """`svd_bk` (scan-rs/src/dim_red/bk_svd.rs:57-146). Returns (U m*k, sigma k, Vt k*n)."""
    m, n = A.shape()
    if m < 2 or n < 2:
        raise ValueError("The input matrix must be at least 2x2.")
    if k > min(m, n):
        raise ValueError("invalid k")
    b = min(min(m, n), b)

    def progress(p):
        if snoop is not None and snoop(p):
            raise CancellationError()

    if m >= n:
        B = omega_panel((n, b), seed) if omega is None else np.array(omega, dtype=np.float64).reshape(n, b)
        K = np.zeros((n, b * n_iter))
        for i in range(n_iter):
            B = _qr_q(A.rdot(A.dot(B).T).T)
            K[:, i * b:(i + 1) * b] = B
            progress(i / n_iter * 0.8)
        Q = _qr_q(K)
        progress(0.82)
        T = A.dot(Q)
        progress(0.93)
        U0, s0, Vt0 = _svddc_some(T)
        U, sigma, Va = U0[:, :k].copy(), s0[:k].copy(), Vt0[:k, :].copy()
        Va = Va @ Q.T
        progress(1.0)
        return U, sigma, Va
    else:
        B = omega_panel((b, m), seed) if omega is None else np.array(omega, dtype=np.float64).reshape(b, m)
        K = np.zeros((b * n_iter, m))
        for i in range(n_iter):
            T = A.rdot(B).T
            B = _qr_q(A.dot(T)).T
            K[i * b:(i + 1) * b, :] = B
            progress(i / n_iter * 0.8)
        Q = _qr_q(K.T)
        progress(0.82)
        T = A.rdot(Q.T)
        progress(0.93)
        U0, s0, Vt0 = _svddc_some(T)
        U, sigma, Va = U0[:, :k].copy(), s0[:k].copy(), Vt0[:k, :].copy()
        U = Q @ U
        progress(1.0)
        return U, sigma, Va


class BkSvd:
    """`BkSvd` (bk_svd.rs:16-53)."""

    def __init__(self, k_multiplier: float = 2.0, n_iter: int = 5):
        self.k_multiplier, self.n_iter = k_multiplier, n_iter

    def run_pca(self, array, k: int, omega=None, snoop=None):
        bsize = int(np.ceil(k * self.k_multiplier))
        u, s, vt = svd_bk(array, k, bsize, self.n_iter, 0, omega, snoop)
        return u, s, vt.T


def svd_rand(A, k: int, l: int, n_iter: int, seed: int = 0, omega: Optional[np.ndarray] = None):
    """`svd_rand` (scan-rs/src/dim_red/rand_svd.rs:54-129)."""
    m, n = A.shape()
    if m < 2 or n < 2:
        raise ValueError("The input matrix must be at least 2x2.")
    if k > min(m, n):
        raise ValueError("invalid k")
    if m >= n:
        om = omega_panel((n, l), seed) if omega is None else np.array(omega, dtype=np.float64).reshape(n, l)
        Q = _qr_q(A.dot(om))
        for _ in range(n_iter):
            Q = _qr_q(A.rdot(Q.T).T)
            Q = _qr_q(A.dot(Q))
        B = A.rdot(Q.T)
        U0, s0, Vt0 = _svddc_some(B)
        U, sigma, Va = U0[:, :k].copy(), s0[:k].copy(), Vt0[:k, :].copy()
        return Q @ U, sigma, Va
    else:
        om = omega_panel((l, m), seed) if omega is None else np.array(omega, dtype=np.float64).reshape(l, m)
        Q = _qr_q(A.rdot(om).T)
        for _ in range(n_iter):
            Q = _qr_q(A.dot(Q))
            Q = _qr_q(A.rdot(Q.T).T)
        B = A.dot(Q)
        U0, s0, Vt0 = _svddc_some(B)
        U, sigma, Va = U0[:, :k].copy(), s0[:k].copy(), Vt0[:k, :].copy()
        return U, sigma, Va @ Q.T


class RandSvd:
    """`RandSvd` (rand_svd.rs:13-50)."""

    def __init__(self, l_multiplier: float = 10.0, n_iter: int = 2):
        self.l_multiplier, self.n_iter = l_multiplier, n_iter

    def run_pca(self, array, k: int, omega=None):
        l = max(k + 4, int(k * self.l_multiplier))
        u, s, vt = svd_rand(array, k, l, self.n_iter, 0, omega)
        return u, s, vt.T


def _norm(x):  # irlba.rs:13-15
    return np.sqrt(np.sum(x * x))


def _orthog(y, X):  # irlba.rs:19-22
    return y - X @ (X.T @ y)


def _invcheck(x):  # irlba.rs:25-33
    eps2 = 2.0 * np.finfo(np.float64).eps
    return 1.0 / x if x > eps2 else 0.0


def irlba(A, nu: int, tol: float, maxit: int, v0: Optional[np.ndarray] = None, snoop=None):
    """`irlba` (scan-rs/src/dim_red/irlba.rs:71-215). `v0` replaces the Normal(0,1)
    start vector (rand_distr 0.6 ziggurat, not reproducible here); returns (U, sigma, V, mprod)."""
    import scipy.linalg as sl

    m, n = A.shape()
    assert not (m < 2 or n < 2), "The input matrix must be at least 2x2."
    assert nu <= min(m, n), "invalid k"
    m_b = min(nu + 20, min(3 * nu, n))
    mprod, it, j, k = 0, 0, 0, nu
    smax = np.finfo(np.float64).min
    V = np.zeros((n, m_b))
    W = np.zeros((m, m_b))
    F = np.zeros(n)
    B = np.zeros((m_b, m_b))
    if v0 is None:
        v0 = SmallRng(0).normal(n)  # seed 0 as irlba.rs:107; same restatement as the product (see SmallRng.normal)
    v0 = np.asarray(v0, dtype=np.float64)
    V[:, 0] = v0 * (1.0 / _norm(v0))
    u = sigma = vt = None
    while it < maxit:
        if it > 0:
            j = k
        W[:, j] = A.dot(V[:, j])
        mprod += 1
        if it > 0:
            W[:, k] = _orthog(W[:, j], W[:, 0:j])
        s = _norm(W[:, j])
        sinv = _invcheck(s)
        W[:, j] *= sinv
        fnorm = 0.0
        while j < m_b:
            F = A.rdot(W[:, j])
            mprod += 1
            F = F - V[:, j] * s
            F = _orthog(F, V[:, 0:j + 1])
            fnorm = _norm(F)
            F = F * _invcheck(fnorm)
            if j == m_b - 1:
                B[j, j] = s
            else:
                V[:, j + 1] = F
                B[j, j] = s
                B[j, j + 1] = fnorm
                W[:, j + 1] = A.dot(V[:, j + 1])
                mprod += 1
                new_w = A.dot(V[:, j + 1])
                new_w = new_w - W[:, j] * fnorm
                new_w = _orthog(new_w, W[:, 0:j + 1])
                s = _norm(new_w)
                sinv = _invcheck(s)
                W[:, j + 1] = new_w * sinv
            j += 1
        u, sigma, vt = sl.svd(B, full_matrices=True, lapack_driver="gesvd")
        resid = fnorm * u[m_b - 1, :]
        smax = sigma[0] if sigma[0] > smax else smax
        num_converged = sum(1 for i in range(nu) if resid[i] < tol * smax)
        if num_converged < nu:
            k = max(num_converged + nu, k)
            k = min(k, m_b - 3)
        else:
            break
        V[:, 0:k] = V[:, 0:m_b] @ vt.T[:, 0:k]
        V[:, k] = F
        B = np.zeros((m_b, m_b))
        for l in range(k):
            B[l, l] = sigma[l]
        B[0:k, k] = resid[0:k]
        W[:, 0:k] = W[:, 0:m_b] @ u[:, 0:k]
        it += 1
        if snoop is not None and snoop(it / maxit):
            raise CancellationError()
    U = W[:, 0:m_b] @ u[:, 0:nu]
    Vout = V[:, 0:m_b] @ vt.T[:, 0:nu]
    return U, sigma[0:nu].copy(), Vout, mprod


def frobenius(a: np.ndarray) -> float:
    """dim_red/mod.rs:114-122 (sqrt of the sum of squares over the element count)."""
    return float(np.sqrt(np.sum(a * a)) / (a.shape[0] * a.shape[1]))
