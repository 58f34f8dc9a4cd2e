"""scan-rs_amd: MI355X (gfx950) implementation of scan-rs's sparse-count-matrix
normalize -> PCA hot path, behind the C ABI of ``include/scanrs_amd.h``.

This module is the Python host-side mirror used by the tests and ``bench.py``:
same names, argument meaning and error behaviour as the reference's operator
surface (``sqz::AdaptiveMat`` / ``LowRankOffset`` / ``scan_rs::normalization`` /
``scan_rs::dim_red::{BkSvd, RandSvd, Irlba}``), every call going straight
through ctypes into ``lib/libscanrs_amd.so``.  There is no CPU fallback: if the
shared library is missing the import fails, and without a gfx950 device every
compute call raises ``ScanrsError``.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCANRS_AMD_LIB: another build of the same library (A/B timing of two builds on one GPU box; tools/pass_bench.py)
LIB_PATH = os.environ.get("SCANRS_AMD_LIB") or os.path.join(_HERE, "lib", "libscanrs_amd.so")

CSR, CSC = 0, 1
FN_LN_1P, FN_LOG2_1P, FN_LOG10_1P, FN_SQUARE = 2, 3, 4, 5


class Normalization:
    """`enum Normalization` (scan-rs/src/normalization.rs:11-28) and its `FromStr` (:30-43)."""

    CellRanger, CellRanger8, SeuratLog, BinomialDeviance, BinomialPearson, WithSizeFactors, LogTransform = range(7)
    _NAMES = {
        "cellranger": 0,
        "cellranger8": 1,
        "seuratlog": 2,
        "binomialdeviance": 3,
        "binomialpearson": 4,
    }

    @staticmethod
    def from_str(s: str) -> int:
        if s not in Normalization._NAMES:
            raise ValueError(f"Normalization not recognized: {s}")
        return Normalization._NAMES[s]


class ScanrsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[scanrs {code}] {msg}")
        self.code = code


class CancellationError(ScanrsError):
    """snoop::CancellationError (snoop/src/lib.rs:5-18)."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). scan-rs_amd has no CPU fallback."
    )


def _preload_rocm_runtime():
    """PyTorch-ROCm wheels bundle their own HIP / HSA runtime under torch/lib with the same sonames as
    /opt/rocm's. Whichever copy is loaded first serves the whole process, and torch does not find the GPU
    when the system copy came first. So if torch is installed (not imported: find_spec only) its copy is
    loaded up front; without torch the system runtime is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


_preload_rocm_runtime()
_lib = ctypes.CDLL(LIB_PATH)
_lib.scanrs_last_error.restype = ctypes.c_char_p
_lib.scanrs_version.restype = ctypes.c_char_p

_PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_double)
_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int)


class _Snoop(ctypes.Structure):
    _fields_ = [("cancel", ctypes.c_void_p), ("progress", _PROGRESS_FN), ("ctx", ctypes.c_void_p)]


class _KernelStat(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * 48),
        ("launches", ctypes.c_uint64),
        ("total_ms", ctypes.c_double),
        ("algorithmic_bytes", ctypes.c_double),
        ("onchip_gather_bytes", ctypes.c_double),
    ]


def _check(code: int):
    if code != 0:
        msg = _lib.scanrs_last_error().decode("utf-8", "replace")
        if code == 3:
            raise CancellationError(code, msg)
        raise ScanrsError(code, msg)


def device_available() -> bool:
    return bool(_lib.scanrs_device_available())


def version() -> str:
    return _lib.scanrs_version().decode()


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class AtomicSnoop:
    """`snoop::AtomicSnoop` (snoop/src/lib.rs:117-226): cancel flag + progress fraction."""

    def __init__(self):
        self._flag = (ctypes.c_uint8 * 1)(0)
        self.progress = 0.0
        self.history = []
        self._cb = _PROGRESS_FN(self._on_progress)

    def _on_progress(self, _ctx, frac):
        self.progress = frac
        self.history.append(frac)

    def cancel(self):
        self._flag[0] = 1

    def is_cancelled(self):
        return bool(self._flag[0])

    def _struct(self):
        return _Snoop(ctypes.cast(self._flag, ctypes.c_void_p), self._cb, None)


def knn(v, k: int):
    """`nn::knn(v, k)` (scan-rs/src/nn.rs:38-57): (cells x k) u32 indices of the k nearest other rows of `v`, nearest first."""
    v = _f64(np.atleast_2d(v))
    n, d = v.shape
    out = np.zeros((n, k), dtype=np.uint32)
    _check(_lib.scanrs_knn(_p(v), ctypes.c_uint64(n), ctypes.c_uint32(d), ctypes.c_uint32(k), _p(out)))
    return out


def find_nn(v, k: int, tree_points, include_self: bool):
    """`nn::find_nn(v, k, ball_tree, include_self)` (nn.rs:63-83); the "tree" is the point set itself."""
    v, t = _f64(np.atleast_2d(v)), _f64(np.atleast_2d(tree_points))
    if v.shape[1] != t.shape[1]:
        raise ScanrsError(1, "Dimension mismatch")
    out = np.zeros((v.shape[0], k), dtype=np.uint32)
    _check(_lib.scanrs_find_nn(_p(v), ctypes.c_uint64(v.shape[0]), _p(t), ctypes.c_uint64(t.shape[0]), ctypes.c_uint32(v.shape[1]),
                               ctypes.c_uint32(k), ctypes.c_int(1 if include_self else 0), _p(out)))
    return out


class AdaptiveVecDesc(ctypes.Structure):
    """`scanrs_adaptive_vec` (include/scanrs_amd.h): one sqz::AdaptiveVec by its encoded buffers."""
    _fields_ = [("kind", ctypes.c_uint32), ("len", ctypes.c_uint64), ("n_units", ctypes.c_uint64), ("data", ctypes.c_void_p),
                ("data_bytes", ctypes.c_uint64), ("fallback_indexes", ctypes.c_void_p), ("fallback_values", ctypes.c_void_p),
                ("n_fallback", ctypes.c_uint64), ("index_bytes", ctypes.c_void_p), ("block_starts", ctypes.c_void_p),
                ("n_block_starts", ctypes.c_uint64)]


class AdaptiveMat:
    """Device-resident `sqz::AdaptiveMat<N, D, M>` (sqz/src/mat.rs:34-42). Once a low-rank
    offset is installed (`center`, `scale_and_center`, `normalize`) the same object plays
    `sqz::LowRankOffset` (sqz/src/low_rank_offset.rs:12-16)."""

    def __init__(self, handle):
        self._h = ctypes.c_void_p(handle)
        self._keep = []  # callbacks that must outlive the handle

    # -- constructors ----------------------------------------------------------------
    @staticmethod
    def from_csmat(rows: int, cols: int, storage: int, indptr, indices, data, unsorted: bool = False) -> "AdaptiveMat":
        """`AdaptiveMat::from_csmat` (mat.rs:92-124): host indptr(u64) / indices(u32) / data(u32).
        unsorted=True: indices may be in any order inside an outer vector (hdf5-io/src/matrix.rs:66-75 fallback)."""
        indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        data = np.ascontiguousarray(data, dtype=np.uint32)
        n_outer = rows if storage == CSR else cols
        if indptr.shape[0] != n_outer + 1:
            raise ScanrsError(6, "indptr length does not match the outer dimension")
        if indices.shape[0] != data.shape[0] or (indptr.shape[0] and int(indptr[-1]) != indices.shape[0]):
            raise ScanrsError(6, "indices/data length does not match indptr")
        h = ctypes.c_void_p()
        fn = _lib.scanrs_mat_create_unsorted if unsorted else _lib.scanrs_mat_create
        _check(fn(ctypes.c_uint64(rows), ctypes.c_uint64(cols), ctypes.c_int(storage), _p(indptr), _p(indices), _p(data), ctypes.byref(h)))
        return AdaptiveMat(h.value)

    @staticmethod
    def from_adaptive_vecs(rows: int, cols: int, storage: int, vecs) -> "AdaptiveMat":
        """`AdaptiveMat::new(rows, cols, storage, Vec<AdaptiveVec>)` (mat.rs:68-90): `vecs` is one mapping per outer
        vector with the fields of `scanrs_adaptive_vec` (kind, len, n_units, data, fallback_indexes, fallback_values,
        index_bytes, block_starts) holding numpy arrays in the reference's in-memory layout; decoded on the device."""
        n = len(vecs)
        table = (AdaptiveVecDesc * max(n, 1))()
        keep = []  # arrays must outlive the call

        def arr(a, dt):
            if a is None:
                return None, 0
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data_as(ctypes.c_void_p), a.shape[0]

        for i, v in enumerate(vecs):
            d = table[i]
            d.kind, d.len, d.n_units = int(v["kind"]), int(v["len"]), int(v["n_units"])
            data = v.get("data")
            if data is not None:
                data = np.ascontiguousarray(data)
                keep.append(data)
                d.data, d.data_bytes = data.ctypes.data_as(ctypes.c_void_p), data.nbytes
            fi, nfi = arr(v.get("fallback_indexes"), np.uint32)
            fv, nfv = arr(v.get("fallback_values"), np.uint32)
            if nfi != nfv:
                raise ScanrsError(6, "fallback indexes / values differ in length")
            d.fallback_indexes, d.fallback_values, d.n_fallback = fi, fv, nfi
            ib, _ = arr(v.get("index_bytes"), np.uint8)
            bs, nbs = arr(v.get("block_starts"), np.uint32)
            d.index_bytes, d.block_starts, d.n_block_starts = ib, bs, nbs
        h = ctypes.c_void_p()
        _check(_lib.scanrs_mat_create_adaptive(ctypes.c_uint64(rows), ctypes.c_uint64(cols), ctypes.c_int(storage), table,
                                              ctypes.c_uint64(n), ctypes.byref(h)))
        return AdaptiveMat(h.value)

    @staticmethod
    def from_device(rows: int, cols: int, storage: int, indptr_ptr: int, indices_ptr: int, data_ptr: int) -> "AdaptiveMat":
        """Same triplet already in device memory (e.g. `tensor.data_ptr()`); copied, not adopted."""
        h = ctypes.c_void_p()
        _check(
            _lib.scanrs_mat_create_device(
                ctypes.c_uint64(rows), ctypes.c_uint64(cols), ctypes.c_int(storage), ctypes.c_void_p(indptr_ptr),
                ctypes.c_void_p(indices_ptr), ctypes.c_void_p(data_ptr), ctypes.byref(h)))
        return AdaptiveMat(h.value)

    @staticmethod
    def from_scipy(m) -> "AdaptiveMat":
        import scipy.sparse as sp

        if sp.isspmatrix_csc(m):
            st = CSC
        else:
            m = m.tocsr()
            st = CSR
        m.sort_indices()
        return AdaptiveMat.from_csmat(m.shape[0], m.shape[1], st, m.indptr, m.indices, m.data)

    @staticmethod
    def from_dense(dense, storage: int = CSR) -> "AdaptiveMat":
        """`AdaptiveMat::from_dense` (mat.rs:122-150)."""
        import scipy.sparse as sp

        dense = np.asarray(dense)
        m = sp.csr_matrix(dense) if storage == CSR else sp.csc_matrix(dense)
        return AdaptiveMat.from_scipy(m)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None:  # (at interpreter shutdown the module globals may be gone already)
            _lib.scanrs_mat_free(h)
            self._h = ctypes.c_void_p()

    # -- shape / views -----------------------------------------------------------------
    def shape(self):
        r, c = ctypes.c_uint64(), ctypes.c_uint64()
        _check(_lib.scanrs_mat_shape(self._h, ctypes.byref(r), ctypes.byref(c)))
        return [int(r.value), int(c.value)]

    def rows(self):
        return self.shape()[0]

    def cols(self):
        return self.shape()[1]

    def nnz(self):
        n = ctypes.c_uint64()
        _check(_lib.scanrs_mat_nnz(self._h, ctypes.byref(n)))
        return int(n.value)

    def storage(self):
        s = ctypes.c_int()
        _check(_lib.scanrs_mat_storage(self._h, ctypes.byref(s)))
        return int(s.value)

    def view(self) -> "AdaptiveMat":
        h = ctypes.c_void_p()
        _check(_lib.scanrs_mat_view(self._h, ctypes.byref(h)))
        v = AdaptiveMat(h.value)
        v._keep = self._keep
        return v

    def t(self) -> "AdaptiveMat":
        h = ctypes.c_void_p()
        _check(_lib.scanrs_mat_t(self._h, ctypes.byref(h)))
        v = AdaptiveMat(h.value)
        v._keep = self._keep
        return v

    # -- lazy maps --------------------------------------------------------------------------
    def reset_map(self):
        _check(_lib.scanrs_mat_reset_map(self._h))
        return self

    def compose_scale_axis(self, axis: int, factors):
        """`compose_map(ScaleAxis::new(Axis(axis), factors))`."""
        f = _f64(factors)
        want = self.rows() if axis == 0 else self.cols()
        if axis in (0, 1) and f.shape[0] != want:
            raise ScanrsError(1, "Dimension mismatch")
        _check(_lib.scanrs_mat_compose_scale_axis(self._h, ctypes.c_int(axis), _p(f)))
        return self

    def apply(self, scalar_fn: int):
        _check(_lib.scanrs_mat_apply(self._h, ctypes.c_int(scalar_fn)))
        return self

    def set_offset(self, u, v):
        """`LowRankOffset::new(mat, u, v)`: u rows x rank, v rank x cols."""
        u, v = _f64(u), _f64(v)
        r, c = self.shape()
        if u.ndim != 2 or v.ndim != 2 or u.shape[0] != r or v.shape[1] != c or u.shape[1] != v.shape[0]:
            raise ScanrsError(1, "Dimension mismatch")
        _check(_lib.scanrs_mat_set_offset(self._h, ctypes.c_uint32(u.shape[1]), _p(u), _p(v)))
        return self

    def center(self, axis: int, m=None):
        mm = None if m is None else _f64(m)
        _check(_lib.scanrs_mat_center(self._h, ctypes.c_int(axis), _p(mm)))
        return self

    def scale(self, axis: int, s=None):
        ss = None if s is None else _f64(s)
        _check(_lib.scanrs_mat_scale(self._h, ctypes.c_int(axis), _p(ss)))
        return self

    def scale_and_center(self, axis: int, scaling_factors=None):
        ss = None if scaling_factors is None else _f64(scaling_factors)
        _check(_lib.scanrs_mat_scale_and_center(self._h, ctypes.c_int(axis), _p(ss)))
        return self

    # -- reductions ----------------------------------------------------------------------------
    def sum_axis(self, axis: int, dtype=np.float64):
        if axis not in (0, 1):
            raise ScanrsError(6, "axis must be 0 or 1")
        n = self.cols() if axis == 0 else self.rows()
        if dtype == np.uint32:
            out = np.zeros(n, dtype=np.uint32)
            _check(_lib.scanrs_mat_sum_axis_u32(self._h, ctypes.c_int(axis), _p(out)))
        else:
            out = np.zeros(n, dtype=np.float64)
            _check(_lib.scanrs_mat_sum_axis_f64(self._h, ctypes.c_int(axis), _p(out)))
        return out

    def mean_axis(self, axis: int):
        n = self.cols() if axis == 0 else self.rows()
        out = np.zeros(n)
        _check(_lib.scanrs_mat_mean_axis(self._h, ctypes.c_int(axis), _p(out)))
        return out

    def mean_var_axis(self, axis: int):
        n = self.cols() if axis == 0 else self.rows()
        mean, var = np.zeros(n), np.zeros(n)
        _check(_lib.scanrs_mat_mean_var_axis(self._h, ctypes.c_int(axis), _p(mean), _p(var)))
        return mean, var

    def to_dense(self):
        r, c = self.shape()
        out = np.zeros((r, c))
        _check(_lib.scanrs_mat_to_dense(self._h, _p(out)))
        return out

    # -- products ----------------------------------------------------------------------------------
    def dot(self, rhs):
        """`self.dot(&rhs)` (mat.rs:1074-1112, low_rank_offset.rs:68-81)."""
        rhs = np.asarray(rhs)
        one_d = rhs.ndim == 1
        if one_d:
            rhs = rhs.reshape(-1, 1)
        r, c = self.shape()
        if rhs.shape[0] != c:
            raise ScanrsError(1, "Dimension mismatch")
        l = rhs.shape[1]
        if rhs.dtype == np.uint32:
            rhs_c = np.ascontiguousarray(rhs)
            out = np.zeros((r, l), dtype=np.uint32)
            _check(_lib.scanrs_mat_dot_u32(self._h, _p(rhs_c), ctypes.c_uint32(l), _p(out)))
        else:
            rhs_c = _f64(rhs)
            out = np.zeros((r, l))
            _check(_lib.scanrs_mat_dot(self._h, _p(rhs_c), ctypes.c_uint32(l), _p(out)))
        return out[:, 0] if one_d else out

    def rdot(self, lhs):
        """`lhs.dot(&self)` (mat.rs:1114-1170, low_rank_offset.rs:83-96)."""
        lhs = np.asarray(lhs)
        one_d = lhs.ndim == 1
        if one_d:
            lhs = lhs.reshape(1, -1)
        r, c = self.shape()
        if lhs.shape[1] != r:
            raise ScanrsError(1, "Dimension mismatch")
        l = lhs.shape[0]
        if lhs.dtype == np.uint32:
            lhs_c = np.ascontiguousarray(lhs)
            out = np.zeros((l, c), dtype=np.uint32)
            _check(_lib.scanrs_mat_rdot_u32(self._h, _p(lhs_c), ctypes.c_uint32(l), _p(out)))
        else:
            lhs_c = _f64(lhs)
            out = np.zeros((l, c))
            _check(_lib.scanrs_mat_rdot(self._h, _p(lhs_c), ctypes.c_uint32(l), _p(out)))
        return out[0, :] if one_d else out

    def dot_device(self, transpose: bool, rhs_ptr: int, ld_rhs: int, l: int, out_ptr: int, ld_out: int):
        _check(
            _lib.scanrs_mat_dot_device(
                self._h, ctypes.c_int(int(transpose)), ctypes.c_void_p(rhs_ptr), ctypes.c_uint32(ld_rhs), ctypes.c_uint32(l),
                ctypes.c_void_p(out_ptr), ctypes.c_uint32(ld_out)))

    # -- sharding / measurement ------------------------------------------------------------------------
    def set_shard(self, rank: int, world: int, outer_begin: int, outer_global: int, allreduce=None):
        """`allreduce(dev_ptr:int, count:int, dtype:int) -> int` sums in place across ranks (0 f64, 1 u64)."""
        cb = None
        if allreduce is not None:
            def _tramp(_ctx, ptr, count, dtype, _f=allreduce):
                try:
                    return int(_f(int(ptr), int(count), int(dtype)) or 0)
                except Exception:  # never unwind into C
                    import traceback

                    traceback.print_exc()
                    return 1

            cb = _ALLREDUCE_FN(_tramp)
            self._keep.append(cb)
        _check(
            _lib.scanrs_mat_set_shard(
                self._h, ctypes.c_uint32(rank), ctypes.c_uint32(world), ctypes.c_uint64(outer_begin),
                ctypes.c_uint64(outer_global), cb if cb is not None else ctypes.cast(None, _ALLREDUCE_FN), None))
        return self

    def set_shard_comm(self, comm: "Comm", outer_begin: int, outer_global: int):
        """Sharded handle over the library's own transport (`scanrs_mat_set_shard_comm`): RCCL collectives on the handle's
        stream. `comm` must outlive the handle's sharded calls."""
        _check(_lib.scanrs_mat_set_shard_comm(self._h, comm._c, ctypes.c_uint32(comm.rank), ctypes.c_uint32(comm.world),
                                              ctypes.c_uint64(outer_begin), ctypes.c_uint64(outer_global)))
        self._keep.append(comm)
        return self

    def profile_enable(self, on: bool = True):
        _check(_lib.scanrs_profile_enable(self._h, ctypes.c_int(int(on))))

    def profile_reset(self):
        _check(_lib.scanrs_profile_reset(self._h))

    def profile_get(self):
        arr = (_KernelStat * 64)()
        n = ctypes.c_uint32()
        _check(_lib.scanrs_profile_get(self._h, arr, ctypes.c_uint32(64), ctypes.byref(n)))
        return {
            arr[i].name.decode(): {
                "launches": int(arr[i].launches),
                "total_ms": float(arr[i].total_ms),
                "algorithmic_bytes": float(arr[i].algorithmic_bytes),
                "onchip_gather_bytes": float(arr[i].onchip_gather_bytes),
            }
            for i in range(min(int(n.value), 64))
        }

    def sync(self):
        _check(_lib.scanrs_mat_sync(self._h))

    def set_panel_precision(self, precision: int):
        """0 = f64 panels (default, the reference's arithmetic); 1 = f32 gather panels with f64 sums (opt-in fast mode)."""
        _check(_lib.scanrs_mat_set_panel_precision(self._h, ctypes.c_int(precision)))
        return self

    def set_spmm_path(self, path: int):
        """0 auto, 1 plain gather kernel, 2 L2-blocked gather kernel, 3 hybrid LDS tiles + gather (see scanrs_mat_set_spmm_path)."""
        _check(_lib.scanrs_mat_set_spmm_path(self._h, ctypes.c_int(path)))
        return self

    def set_option(self, key: str, value: float):
        """Tuning options of the handle (include/scanrs_amd.h, scanrs_mat_set_option)."""
        _check(_lib.scanrs_mat_set_option(self._h, key.encode(), ctypes.c_double(value)))
        return self

    def counter(self, key: str) -> int:
        v = ctypes.c_uint64()
        _check(_lib.scanrs_mat_get_counter(self._h, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def chol_rinv(self, g, rows: int, pass_no: int = 0):
        """One factor step of the device-side CholeskyQR (scanrs_mat_chol_rinv): (rinv, done, status, err, shift)."""
        g = _f64(g)
        n = g.shape[0]
        rinv = np.zeros((n, n))
        done, status = ctypes.c_int(), ctypes.c_int()
        err, shift = ctypes.c_double(), ctypes.c_double()
        _check(_lib.scanrs_mat_chol_rinv(self._h, _p(g), ctypes.c_uint32(n), ctypes.c_uint64(rows), ctypes.c_int(pass_no), _p(rinv),
                                         ctypes.byref(done), ctypes.byref(status), ctypes.byref(err), ctypes.byref(shift)))
        return rinv, int(done.value), int(status.value), float(err.value), float(shift.value)

    def target_umi(self) -> float:
        t = ctypes.c_double()
        _check(_lib.scanrs_mat_target_umi(self._h, ctypes.byref(t)))
        return float(t.value)


# explicit prototypes of the entry points that take doubles or 64-bit integers by value (ctypes' default conversion is for ints)
_lib.scanrs_mat_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
_lib.scanrs_mat_set_option.restype = ctypes.c_int
_lib.scanrs_set_global_option.argtypes = [ctypes.c_char_p, ctypes.c_double]
_lib.scanrs_set_global_option.restype = ctypes.c_int
_lib.scanrs_mat_get_counter.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
_lib.scanrs_mat_get_counter.restype = ctypes.c_int
_lib.scanrs_debug_wait_never.argtypes = [ctypes.c_double]
_lib.scanrs_debug_wait_never.restype = ctypes.c_int
_lib.scanrs_reserve_device_memory.argtypes = [ctypes.c_uint64]
_lib.scanrs_reserve_device_memory.restype = ctypes.c_int


def set_global_option(key: str, value: float):
    """Process-wide options of the entry points that take no handle (include/scanrs_amd.h, scanrs_set_global_option)."""
    _check(_lib.scanrs_set_global_option(key.encode(), ctypes.c_double(value)))


def init():
    """`scanrs_init`: load the device code and start the one-off table computation now instead of inside the first call."""
    _check(_lib.scanrs_init())


def reserve_device_memory(n_bytes: int):
    """`scanrs_reserve_device_memory`: one allocation now that the library's later large buffers are carved from."""
    _check(_lib.scanrs_reserve_device_memory(ctypes.c_uint64(int(n_bytes))))


def release_cached_memory():
    """Give the device blocks the library keeps for reuse back to the driver (include/scanrs_amd.h, "device_cache_fraction")."""
    _check(_lib.scanrs_release_cached_memory())


def cached_memory_bytes() -> int:
    n = ctypes.c_uint64()
    _check(_lib.scanrs_cached_memory_bytes(ctypes.byref(n)))
    return int(n.value)


def device_memory_in_use() -> int:
    n = ctypes.c_uint64()
    _check(_lib.scanrs_device_memory_in_use(ctypes.byref(n)))
    return int(n.value)


def debug_arena_selftest(rounds: int = 20000, seed: int = 1):
    """The bookkeeping of a device-memory reserve driven on a made-up address range (no device needed): raises when blocks overlap,
    holes fail to merge or the arena is not whole at the end."""
    _lib.scanrs_debug_arena_selftest.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
    _lib.scanrs_debug_arena_selftest.restype = ctypes.c_int
    _check(_lib.scanrs_debug_arena_selftest(ctypes.c_uint32(rounds), ctypes.c_uint64(seed)))


def debug_wait_never(timeout_s: float):
    """The library's bounded wait on an event that is never signalled (no device needed): raises ScanrsError(DEVICE) after
    `timeout_s` seconds, with the report of a real stuck wait."""
    _check(_lib.scanrs_debug_wait_never(ctypes.c_double(timeout_s)))


# ---- scan-rs/src/normalization.rs ---------------------------------------------------------------
def normalize(mat: AdaptiveMat, norm: int) -> AdaptiveMat:
    """`normalize(mat, norm)` (normalization.rs:46-69); consumes `mat` like the reference, returns it."""
    _check(_lib.scanrs_normalize(mat._h, ctypes.c_int(norm), None))
    return mat


def normalize_with_size_factor(mat: AdaptiveMat, norm: int, size_factors=None) -> AdaptiveMat:
    """normalization.rs:72-102."""
    sf = None
    if size_factors is not None:
        sf = np.ascontiguousarray(size_factors, dtype=np.uint32)
        if sf.shape[0] != mat.cols():
            raise ScanrsError(1, "Size of the size factor and matrix columns dont match.")
    _check(_lib.scanrs_normalize(mat._h, ctypes.c_int(norm), _p(sf)))
    return mat


def log_normalize_with_size_factor(mat: AdaptiveMat, umi_count_sum: Optional[float], log_fn: int, size_factors=None):
    """normalization.rs:138-178 (no centre/scale)."""
    sf = None
    if size_factors is not None:
        sf = np.ascontiguousarray(size_factors, dtype=np.uint32)
        if sf.shape[0] != mat.cols():
            raise ScanrsError(1, "Size of the size factor and matrix columns dont match.")
    _check(
        _lib.scanrs_log_normalize(
            mat._h, ctypes.c_double(-1.0 if umi_count_sum is None else float(umi_count_sum)), ctypes.c_int(log_fn), _p(sf)))
    return mat


def log1p_normalize_fixed_point(mat: AdaptiveMat, log_fn: int, base: int, exponent: int) -> AdaptiveMat:
    """normalization.rs:191-213."""
    _check(_lib.scanrs_log1p_normalize_fixed_point(mat._h, ctypes.c_int(log_fn), ctypes.c_uint32(base), ctypes.c_uint32(exponent)))
    return mat


def binom_deviance_resid(mat: AdaptiveMat) -> AdaptiveMat:
    return normalize(mat, Normalization.BinomialDeviance)


def binom_pearson_resid(mat: AdaptiveMat) -> AdaptiveMat:
    return normalize(mat, Normalization.BinomialPearson)


# ---- scan-rs/src/dim_red ---------------------------------------------------------------------------
def omega_fill(seed: int, count: int) -> np.ndarray:
    out = np.zeros(count)
    _check(_lib.scanrs_omega_fill(ctypes.c_uint64(seed), ctypes.c_uint64(count), _p(out)))
    return out


def _snoop_arg(snoop):
    if snoop is None:
        return None, None
    s = snoop._struct()
    return ctypes.byref(s), s


class BkSvd:
    """`BkSvd` (dim_red/bk_svd.rs:16-53)."""

    def __init__(self, k_multiplier: float = 2.0, n_iter: int = 5):
        self.k_multiplier, self.n_iter = k_multiplier, n_iter

    def run_pca(self, matrix: AdaptiveMat, k: int, omega=None, snoop: Optional[AtomicSnoop] = None, seed: int = 0, out=None):
        """Returns (u rows x k, s k, v cols x k) — `PcaResult` (dim_red/mod.rs:47). `out=(u, v)`: the factors are written into
        these C-contiguous float64 arrays of shape (rows, k) / (cols, k) — the C ABI's own form (caller-allocated buffers,
        include/scanrs_amd.h) — instead of fresh ones."""
        r, c = matrix.shape()
        if out is not None:
            u, v = out
            for a_, shp in ((u, (r, k)), (v, (c, k))):
                if not (isinstance(a_, np.ndarray) and a_.dtype == np.float64 and a_.flags.c_contiguous and a_.shape == shp):
                    raise ScanrsError(6, f"out arrays must be C-contiguous float64 of shape {shp}")
            s = np.zeros(max(k, 0))
        else:
            u, s, v = np.zeros((r, max(k, 0))), np.zeros(max(k, 0)), np.zeros((c, max(k, 0)))
        om = None if omega is None else _f64(omega)
        sref, _keep = _snoop_arg(snoop)
        _check(
            _lib.scanrs_pca_bk(
                matrix._h, ctypes.c_uint32(k), ctypes.c_double(self.k_multiplier), ctypes.c_uint32(self.n_iter),
                ctypes.c_uint64(seed), _p(om), sref, _p(u), _p(s), _p(v)))
        return u, s, v

    run_pca_cancellable = run_pca

    def run_pca_device(self, matrix: AdaptiveMat, k: int, omega=None, snoop: Optional[AtomicSnoop] = None, seed: int = 0):
        """The same call with U and V left in device memory: returns (s, PcaResultDevice). The factors are reached through
        `scanrs_pca_result_device` (pointers + leading dimensions) — for a device consumer such as `knn_device`."""
        s = np.zeros(max(k, 0))
        om = None if omega is None else _f64(omega)
        sref, _keep = _snoop_arg(snoop)
        _check(
            _lib.scanrs_pca_bk(
                matrix._h, ctypes.c_uint32(k), ctypes.c_double(self.k_multiplier), ctypes.c_uint32(self.n_iter),
                ctypes.c_uint64(seed), _p(om), sref, None, _p(s), None))
        return s, pca_result_device(matrix)


class Comm:
    """`scanrs_comm`: the library's RCCL communicator of the one-process-per-GPU form. Rank 0 draws the id with
    `Comm.unique_id()` and the host program hands the 128 bytes to the other ranks (bench.py: a torch.distributed
    broadcast on the gloo control plane); every rank then builds `Comm(id, rank, world)` with its device current."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_uint8 * Comm.ID_BYTES)()
        _check(_lib.scanrs_comm_get_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid: bytes, rank: int, world: int):
        if len(uid) != Comm.ID_BYTES:
            raise ScanrsError(6, "unique id must be 128 bytes")
        self.rank, self.world = int(rank), int(world)
        self._c = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * Comm.ID_BYTES).from_buffer_copy(uid)
        _check(_lib.scanrs_comm_create(buf, ctypes.c_uint32(rank), ctypes.c_uint32(world), ctypes.byref(self._c)))

    def info(self) -> dict:
        """`scanrs_comm_info`: ranks / this rank as RCCL itself counts them (ncclCommCount, ncclCommUserRank) and the sum all-reduces
        that went through this communicator so far."""
        n, r = ctypes.c_uint32(), ctypes.c_uint32()
        calls, nbytes = ctypes.c_uint64(), ctypes.c_uint64()
        _check(_lib.scanrs_comm_info(self._c, ctypes.byref(n), ctypes.byref(r), ctypes.byref(calls), ctypes.byref(nbytes)))
        return {"rccl_nranks": int(n.value), "rccl_rank": int(r.value), "allreduce_calls": int(calls.value), "allreduce_bytes": int(nbytes.value)}

    def close(self):
        if getattr(self, "_c", None) is not None and self._c:
            _lib.scanrs_comm_free(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiMat:
    """`scanrs_multi`: the single-process multi-GPU form (one handle for the whole matrix, the library shards it by
    nonzeros over `devices` and drives every shard from its own host thread). `devices` may repeat an id."""

    def __init__(self, rows, cols, storage, indptr, indices, values, n_shards: int, devices=None):
        ip = np.ascontiguousarray(indptr, dtype=np.uint64)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        vv = np.ascontiguousarray(values, dtype=np.uint32)
        self.rows, self.cols, self.storage, self.n_shards = int(rows), int(cols), int(storage), int(n_shards)
        dv = None if devices is None else (ctypes.c_int * n_shards)(*[int(d) for d in devices])
        self._h = ctypes.c_void_p()
        _check(_lib.scanrs_multi_create(ctypes.c_uint64(rows), ctypes.c_uint64(cols), ctypes.c_int(storage), _p(ip), _p(ix), _p(vv),
                                        ctypes.c_uint32(n_shards), dv, ctypes.byref(self._h)))

    def shard_ranges(self):
        out = []
        for i in range(self.n_shards):
            lo, hi, dev = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
            _check(_lib.scanrs_multi_shard(self._h, ctypes.c_uint32(i), None, ctypes.byref(dev), ctypes.byref(lo), ctypes.byref(hi)))
            out.append((dev.value, lo.value, hi.value))
        return out

    def normalize(self, normalization: int, size_factors=None):
        sf = None if size_factors is None else np.ascontiguousarray(size_factors, dtype=np.uint32)
        _check(_lib.scanrs_multi_normalize(self._h, ctypes.c_int(int(normalization)), _p(sf)))
        return self

    def run_pca_bk(self, k: int, k_multiplier: float = 2.0, n_iter: int = 5, omega=None, seed: int = 0, snoop=None):
        u, s, v = np.zeros((self.rows, k)), np.zeros(k), np.zeros((self.cols, k))
        om = None if omega is None else _f64(omega)
        sref, _keep = _snoop_arg(snoop)
        _check(_lib.scanrs_multi_pca_bk(self._h, ctypes.c_uint32(k), ctypes.c_double(k_multiplier), ctypes.c_uint32(n_iter),
                                        ctypes.c_uint64(seed), _p(om), sref, _p(u), _p(s), _p(v)))
        return u, s, v

    def run_pca_rand(self, k: int, l_multiplier: float = 10.0, n_iter: int = 2, omega=None, seed: int = 0):
        u, s, v = np.zeros((self.rows, k)), np.zeros(k), np.zeros((self.cols, k))
        om = None if omega is None else _f64(omega)
        _check(_lib.scanrs_multi_pca_rand(self._h, ctypes.c_uint32(k), ctypes.c_double(l_multiplier), ctypes.c_uint32(n_iter),
                                          ctypes.c_uint64(seed), _p(om), _p(u), _p(s), _p(v)))
        return u, s, v

    def log_normalize(self, umi_count_sum=None, log_fn: int = 3, size_factors=None):
        sf = None if size_factors is None else np.ascontiguousarray(size_factors, dtype=np.uint32)
        _check(_lib.scanrs_multi_log_normalize(self._h, ctypes.c_double(-1.0 if umi_count_sum is None else umi_count_sum),
                                               ctypes.c_int(log_fn), _p(sf)))
        return self

    def run_pca_irlba(self, k: int, tol: float = 1e-4, max_iter: int = 50, v0=None):
        u, s, v = np.zeros((self.rows, k)), np.zeros(k), np.zeros((self.cols, k))
        v0c = None if v0 is None else _f64(v0)
        mp = ctypes.c_uint32()
        _check(_lib.scanrs_multi_pca_irlba(self._h, ctypes.c_uint32(k), ctypes.c_double(tol), ctypes.c_uint32(max_iter), _p(v0c), None,
                                           _p(u), _p(s), _p(v), ctypes.byref(mp)))
        self.mprod = int(mp.value)
        return u, s, v

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.scanrs_multi_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevArray:
    """A run of `n` f64 values of device memory at raw address `ptr` as a __cuda_array_interface__ object
    (torch.as_tensor(DevArray(...), device=...) views it without a copy)."""

    def __init__(self, ptr: int, n: int, typestr: str = "<f8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class PcaResultDevice:
    """Where the last PCA of a handle left its factors in HBM (`scanrs_pca_result_device`): raw device addresses,
    leading dimensions in elements; `u` is rows x k, `v` cols x k. Valid until the handle's next PCA call."""

    def __init__(self, d_u, ld_u, d_v, ld_v, k, rows, cols):
        self.d_u, self.ld_u, self.d_v, self.ld_v, self.k, self.rows, self.cols = d_u, ld_u, d_v, ld_v, k, rows, cols


def pca_result_device(matrix: "AdaptiveMat") -> PcaResultDevice:
    d_u, d_v = ctypes.c_void_p(), ctypes.c_void_p()
    ld_u, ld_v, k = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    _check(_lib.scanrs_pca_result_device(matrix._h, ctypes.byref(d_u), ctypes.byref(ld_u), ctypes.byref(d_v), ctypes.byref(ld_v),
                                         ctypes.byref(k)))
    r, c = matrix.shape()
    return PcaResultDevice(d_u.value, ld_u.value, d_v.value, ld_v.value, k.value, r, c)


def knn_device(d_points: int, n: int, ld: int, d: int, k: int):
    """`nn::knn` on scores that already live in device memory (e.g. PcaResultDevice.d_v / ld_v): (n x k) u32 host array."""
    out = np.zeros((n, k), dtype=np.uint32)
    _check(_lib.scanrs_knn_device(ctypes.c_void_p(d_points), ctypes.c_uint64(n), ctypes.c_uint32(ld), ctypes.c_uint32(d),
                                  ctypes.c_uint32(k), _p(out)))
    return out


class RandSvd:
    """`RandSvd` (dim_red/rand_svd.rs:13-50)."""

    def __init__(self, l_multiplier: float = 10.0, n_iter: int = 2):
        self.l_multiplier, self.n_iter = l_multiplier, n_iter

    def run_pca(self, matrix: AdaptiveMat, k: int, omega=None, seed: int = 0):
        r, c = matrix.shape()
        u, s, v = np.zeros((r, k)), np.zeros(k), np.zeros((c, k))
        om = None if omega is None else _f64(omega)
        _check(
            _lib.scanrs_pca_rand(
                matrix._h, ctypes.c_uint32(k), ctypes.c_double(self.l_multiplier), ctypes.c_uint32(self.n_iter),
                ctypes.c_uint64(seed), _p(om), _p(u), _p(s), _p(v)))
        return u, s, v


class Irlba:
    """`Irlba` (dim_red/irlba.rs:36-67)."""

    def __init__(self, tol: float = 0.0001, max_iter: int = 50):
        self.tol, self.max_iter = tol, max_iter
        self.mprod = 0

    def run_pca(self, matrix: AdaptiveMat, k: int, v0=None, snoop: Optional[AtomicSnoop] = None):
        r, c = matrix.shape()
        u, s, v = np.zeros((r, k)), np.zeros(k), np.zeros((c, k))
        vv = None if v0 is None else _f64(v0)
        sref, _keep = _snoop_arg(snoop)
        mp = ctypes.c_uint32()
        _check(
            _lib.scanrs_pca_irlba(
                matrix._h, ctypes.c_uint32(k), ctypes.c_double(self.tol), ctypes.c_uint32(self.max_iter), _p(vv), sref,
                _p(u), _p(s), _p(v), ctypes.byref(mp)))
        self.mprod = int(mp.value)
        return u, s, v


def plan_shards(indptr, world: int) -> np.ndarray:
    """nnz-balanced contiguous partition of the outer dimension."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
    bounds = np.zeros(world + 1, dtype=np.uint64)
    _check(_lib.scanrs_plan_shards(_p(indptr), ctypes.c_uint64(indptr.shape[0] - 1), ctypes.c_uint32(world), _p(bounds)))
    return bounds


# host-side dense helpers (checked by the CPU test-suite)
def host_chol_upper(g):
    g = _f64(g).copy()
    _check(_lib.scanrs_host_chol_upper(_p(g), ctypes.c_int(g.shape[0])))
    return g


def host_inv_upper(r):
    r = _f64(r).copy()
    _check(_lib.scanrs_host_inv_upper(_p(r), ctypes.c_int(r.shape[0])))
    return r


def host_sym_eig(a):
    a = _f64(a)
    n = a.shape[0]
    w, z = np.zeros(n), np.zeros((n, n))
    _check(_lib.scanrs_host_sym_eig(_p(a), ctypes.c_int(n), _p(w), _p(z)))
    return w, z


def host_sym_eig_topk(a, k):
    a = _f64(a)
    n = a.shape[0]
    w, z = np.zeros(k), np.zeros((n, k))
    _check(_lib.scanrs_host_sym_eig_topk(_p(a), ctypes.c_int(n), ctypes.c_int(k), _p(w), _p(z)))
    return w, z


EXPORTED_SYMBOLS = [
    "scanrs_comm_info", "scanrs_last_error", "scanrs_device_available", "scanrs_version", "scanrs_mat_create", "scanrs_mat_create_unsorted", "scanrs_mat_create_device", "scanrs_mat_create_adaptive", "scanrs_knn", "scanrs_find_nn",
    "scanrs_mat_free", "scanrs_mat_view", "scanrs_mat_t", "scanrs_mat_shape", "scanrs_mat_nnz", "scanrs_mat_storage",
    "scanrs_mat_reset_map", "scanrs_mat_compose_scale_axis", "scanrs_mat_apply", "scanrs_mat_set_offset",
    "scanrs_mat_center", "scanrs_mat_scale", "scanrs_mat_scale_and_center", "scanrs_mat_sum_axis_u32",
    "scanrs_mat_sum_axis_f64", "scanrs_mat_mean_axis", "scanrs_mat_mean_var_axis", "scanrs_mat_to_dense",
    "scanrs_mat_dot", "scanrs_mat_rdot", "scanrs_mat_dot_u32", "scanrs_mat_rdot_u32", "scanrs_mat_dot_device",
    "scanrs_normalize", "scanrs_log_normalize", "scanrs_log1p_normalize_fixed_point", "scanrs_mat_target_umi",
    "scanrs_pca_bk", "scanrs_pca_rand", "scanrs_pca_irlba", "scanrs_pca_result_device", "scanrs_knn_device", "scanrs_omega_fill", "scanrs_mat_set_shard", "scanrs_mat_set_shard_comm", "scanrs_comm_get_unique_id", "scanrs_comm_create", "scanrs_comm_free",
    "scanrs_multi_create", "scanrs_multi_free", "scanrs_multi_n_shards", "scanrs_multi_shard", "scanrs_multi_normalize", "scanrs_multi_pca_bk", "scanrs_multi_pca_rand", "scanrs_multi_pca_irlba", "scanrs_multi_log_normalize",
    "scanrs_plan_shards", "scanrs_profile_enable", "scanrs_profile_reset", "scanrs_profile_get", "scanrs_mat_sync", "scanrs_mat_set_spmm_path", "scanrs_mat_set_option", "scanrs_set_global_option", "scanrs_mat_set_panel_precision",
    "scanrs_mat_chol_rinv", "scanrs_mat_get_counter", "scanrs_host_chol_upper", "scanrs_host_inv_upper", "scanrs_host_sym_eig", "scanrs_host_sym_eig_topk", "scanrs_debug_wait_never", "scanrs_debug_barrier_alone", "scanrs_debug_arena_selftest", "scanrs_init", "scanrs_release_cached_memory", "scanrs_cached_memory_bytes", "scanrs_reserve_device_memory", "scanrs_device_memory_in_use",
    "scanrs_h5_read_csc_matrix", "scanrs_h5_read_adaptive_csr_matrix", "scanrs_h5_read_matrix_metadata", "scanrs_h5_matrix_free",
    "scanrs_h5_matrix_shape", "scanrs_h5_matrix_arrays", "scanrs_h5_matrix_n_strings", "scanrs_h5_matrix_string", "scanrs_h5_matrix_removed",
    "scanrs_h5_read_umi_counts", "scanrs_h5_get_clustering_keys", "scanrs_h5_get_clustering", "scanrs_h5_get_differential_expression",
    "scanrs_h5_read_f64", "scanrs_h5_read_strings", "scanrs_h5_member_names", "scanrs_mtx_read", "scanrs_mat_create_from_file",
]
