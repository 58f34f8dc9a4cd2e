"""`scan_rs::mtx::load_mtx` (scan-rs/src/mtx.rs:10-51): gzipped MatrixMarket coordinate file -> CSR arrays, read by the
library (csrc/mtx_reader.cpp, host-only). `load_mtx(path).to_device()` is the `AdaptiveMat` the reference returns."""
import ctypes

from . import _check, _lib
from .hdf5_io import FeatureBarcodeMatrix, _take


def load_mtx(path: str) -> FeatureBarcodeMatrix:
    """Rows x cols CSR with u32 values: comments '%', header "NROW NCOL NNZ", 1-based triplets, duplicates summed,
    column indices ascending inside a row (TriMat::to_csr). The string tables of the result are empty."""
    h = ctypes.c_void_p()
    _check(_lib.scanrs_mtx_read(str(path).encode(), ctypes.byref(h)))
    return _take(h)
