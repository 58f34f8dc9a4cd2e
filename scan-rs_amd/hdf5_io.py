"""Host-side mirror of the reference's `hdf5-io` crate (hdf5-io/src/matrix.rs, analysis.rs) over the C ABI's
`scanrs_h5_*` entry points: 10x feature-barcode matrix files -> the arrays `AdaptiveMat` takes, analysis files ->
clusterings and differential-expression tables. The files are parsed by the library's own reader
(csrc/h5lite.cpp); there is no libhdf5 / h5py dependency and no device is needed to read."""
import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Set, Tuple

import numpy as np

from . import CSC, CSR, AdaptiveMat, _check, _lib

FEATURE_TYPE_GENE_EXPRESSION = "Gene Expression"  # matrix.rs:14

_lib.scanrs_h5_matrix_string.restype = ctypes.c_char_p


@dataclass
class FeatureBarcodeMatrix:
    """`GenericFeatureBarcodeMatrix` (scan-types/src/matrix.rs:8-15) / `MatrixMetadata` with the matrix as host
    arrays: `indptr` u64, `indices` u32, `values` u32 in `storage` orientation (features x barcodes)."""
    name: str
    barcodes: List[str]
    feature_ids: List[str]
    feature_names: List[str]
    feature_types: List[str]
    rows: int
    cols: int
    nnz: int
    storage: int = CSC
    indptr: Optional[np.ndarray] = None
    indices: Optional[np.ndarray] = None
    values: Optional[np.ndarray] = None
    removed_features: Set[int] = field(default_factory=set)

    def to_dense(self) -> np.ndarray:
        out = np.zeros((self.rows, self.cols), dtype=np.int64)
        for o in range(len(self.indptr) - 1):
            s, e = int(self.indptr[o]), int(self.indptr[o + 1])
            if self.storage == CSC:
                out[self.indices[s:e], o] = self.values[s:e]
            else:
                out[o, self.indices[s:e]] = self.values[s:e]
        return out

    def to_device(self) -> AdaptiveMat:
        """Upload as the `AdaptiveMat` the solvers take (needs a gfx950 device)."""
        return AdaptiveMat.from_csmat(self.rows, self.cols, self.storage, self.indptr, self.indices, self.values)


def _strings(h, what: int) -> List[str]:
    n = ctypes.c_uint64()
    _check(_lib.scanrs_h5_matrix_n_strings(h, ctypes.c_int(what), ctypes.byref(n)))
    return [_lib.scanrs_h5_matrix_string(h, ctypes.c_int(what), ctypes.c_uint64(i)).decode("utf-8", "replace") for i in range(n.value)]


def _take(h) -> FeatureBarcodeMatrix:
    try:
        rows, cols, nnz, storage = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
        _check(_lib.scanrs_h5_matrix_shape(h, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(nnz), ctypes.byref(storage)))
        ip, ix, vv = ctypes.POINTER(ctypes.c_uint64)(), ctypes.POINTER(ctypes.c_uint32)(), ctypes.POINTER(ctypes.c_uint32)()
        _check(_lib.scanrs_h5_matrix_arrays(h, ctypes.byref(ip), ctypes.byref(ix), ctypes.byref(vv)))
        m = FeatureBarcodeMatrix(
            name=_strings(h, 4)[0], barcodes=_strings(h, 0), feature_ids=_strings(h, 1), feature_names=_strings(h, 2),
            feature_types=_strings(h, 3), rows=rows.value, cols=cols.value, nnz=nnz.value, storage=storage.value)
        if ip:
            n_outer = rows.value if storage.value == CSR else cols.value
            m.indptr = np.ctypeslib.as_array(ip, shape=(n_outer + 1,)).copy()
            m.indices = np.ctypeslib.as_array(ix, shape=(nnz.value,)).copy() if nnz.value else np.zeros(0, np.uint32)
            m.values = np.ctypeslib.as_array(vv, shape=(nnz.value,)).copy() if nnz.value else np.zeros(0, np.uint32)
        rem, n = ctypes.POINTER(ctypes.c_uint64)(), ctypes.c_uint64()
        _check(_lib.scanrs_h5_matrix_removed(h, ctypes.byref(rem), ctypes.byref(n)))
        m.removed_features = {int(rem[i]) for i in range(n.value)}
        return m
    finally:
        _lib.scanrs_h5_matrix_free(h)


def _opt(s: Optional[str]):
    return None if s is None else s.encode()


def read_csc_matrix(filtered_matrix: str) -> FeatureBarcodeMatrix:
    """hdf5-io/src/matrix.rs:56-97."""
    h = ctypes.c_void_p()
    _check(_lib.scanrs_h5_read_csc_matrix(str(filtered_matrix).encode(), ctypes.byref(h)))
    return _take(h)


def read_adaptive_csr_matrix(filtered_matrix: str, retain_feature_like: Optional[str] = None,
                             shrink_row: Optional[int] = None) -> Tuple[FeatureBarcodeMatrix, Set[int]]:
    """hdf5-io/src/matrix.rs:129-199: (matrix with the kept features, indices of the removed features)."""
    h = ctypes.c_void_p()
    _check(_lib.scanrs_h5_read_adaptive_csr_matrix(str(filtered_matrix).encode(), _opt(retain_feature_like),
                                                   ctypes.c_int64(-1 if shrink_row is None else int(shrink_row)), ctypes.byref(h)))
    m = _take(h)
    return m, m.removed_features


def read_matrix_metadata(filtered_matrix: str, retain_feature_like: Optional[str] = None) -> FeatureBarcodeMatrix:
    """hdf5-io/src/matrix.rs:17-54 (`MatrixMetadata`: no matrix arrays, `nnz` from the size of `data`)."""
    h = ctypes.c_void_p()
    _check(_lib.scanrs_h5_read_matrix_metadata(str(filtered_matrix).encode(), _opt(retain_feature_like), ctypes.byref(h)))
    return _take(h)


def read_umi_counts_from_matrix(filtered_matrix: str) -> np.ndarray:
    """hdf5-io/src/matrix.rs:270-299."""
    n = ctypes.c_uint64()
    _check(_lib.scanrs_h5_read_umi_counts(str(filtered_matrix).encode(), None, ctypes.c_uint64(0), ctypes.byref(n)))
    out = np.zeros(n.value, dtype=np.uint32)
    _check(_lib.scanrs_h5_read_umi_counts(str(filtered_matrix).encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n.value), ctypes.byref(n)))
    return out


def _packed(call) -> List[str]:
    n, nbytes = ctypes.c_uint64(), ctypes.c_uint64()
    _check(call(None, ctypes.c_uint64(0), ctypes.byref(n), ctypes.byref(nbytes)))
    buf = ctypes.create_string_buffer(max(1, nbytes.value))
    _check(call(buf, ctypes.c_uint64(nbytes.value), ctypes.byref(n), ctypes.byref(nbytes)))
    raw = buf.raw[: nbytes.value]
    return [s.decode("utf-8", "replace") for s in raw.split(b"\0")[: n.value]]


def get_clustering_keys(analysis_h5: str) -> List[str]:
    """hdf5-io/src/analysis.rs:38-41."""
    p = str(analysis_h5).encode()
    return _packed(lambda buf, cap, n, nb: _lib.scanrs_h5_get_clustering_keys(p, buf, cap, n, nb))


def get_clustering(analysis_h5: str, clustering_key: str) -> Tuple[int, np.ndarray]:
    """hdf5-io/src/analysis.rs:5-20: (num_clusters as u16, clusters as i16)."""
    p, k = str(analysis_h5).encode(), clustering_key.encode()
    nc, n = ctypes.c_uint16(), ctypes.c_uint64()
    _check(_lib.scanrs_h5_get_clustering(p, k, ctypes.byref(nc), None, ctypes.c_uint64(0), ctypes.byref(n)))
    out = np.zeros(n.value, dtype=np.int16)
    _check(_lib.scanrs_h5_get_clustering(p, k, ctypes.byref(nc), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n.value), ctypes.byref(n)))
    return nc.value, out


def get_differential_expression(analysis_h5: str, clustering_key: str) -> np.ndarray:
    """hdf5-io/src/analysis.rs:23-36 (rows of the table)."""
    p, k = str(analysis_h5).encode(), clustering_key.encode()
    r, c = ctypes.c_uint64(), ctypes.c_uint64()
    _check(_lib.scanrs_h5_get_differential_expression(p, k, None, ctypes.c_uint64(0), ctypes.byref(r), ctypes.byref(c)))
    out = np.zeros((r.value, c.value))
    _check(_lib.scanrs_h5_get_differential_expression(p, k, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(out.size), ctypes.byref(r), ctypes.byref(c)))
    return out


# ---- generic access (what the parser tests use) ---------------------------------------------------------------------
def read_dataset(path: str, dataset: str) -> np.ndarray:
    """Any numeric dataset, converted to f64, in its stored shape."""
    p, d = str(path).encode(), dataset.encode()
    dims, rank = (ctypes.c_uint64 * 8)(), ctypes.c_uint32()
    _check(_lib.scanrs_h5_read_f64(p, d, None, ctypes.c_uint64(0), dims, ctypes.byref(rank)))
    shape = tuple(int(dims[i]) for i in range(rank.value))
    out = np.zeros(shape)
    _check(_lib.scanrs_h5_read_f64(p, d, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(max(1, out.size)), dims, ctypes.byref(rank)))
    return out


def read_strings(path: str, dataset: str) -> List[str]:
    p, d = str(path).encode(), dataset.encode()
    return _packed(lambda buf, cap, n, nb: _lib.scanrs_h5_read_strings(p, d, buf, cap, n, nb))


def member_names(path: str, group: str = "/") -> List[str]:
    p, g = str(path).encode(), group.encode()
    return _packed(lambda buf, cap, n, nb: _lib.scanrs_h5_member_names(p, g, buf, cap, n, nb))


def mat_from_file(path: str, retain_feature_like: Optional[str] = None, shrink_row: Optional[int] = None):
    """File -> device handle in one call (`scanrs_mat_create_from_file`): a 10x `.h5` through `read_adaptive_csr_matrix`,
    anything else through `load_mtx`. Returns (AdaptiveMat, FeatureBarcodeMatrix metadata without the arrays' copies)."""
    h, meta = ctypes.c_void_p(), ctypes.c_void_p()
    _check(_lib.scanrs_mat_create_from_file(str(path).encode(), _opt(retain_feature_like),
                                            ctypes.c_int64(-1 if shrink_row is None else int(shrink_row)), ctypes.byref(h), ctypes.byref(meta)))
    return AdaptiveMat(h.value), _take(meta)
