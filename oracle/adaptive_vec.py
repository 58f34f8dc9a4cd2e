"""TEST INFRASTRUCTURE ONLY (see oracle/scanrs_oracle.py): CPU restatement of sqz's `AdaptiveVec`
(sqz/src/vec.rs), the eight in-memory encodings a `sqz::AdaptiveMat` is made of.

Parity: pinned by construction against the reference's own property test (`test_sparse`,
vec.rs:1379-1451: construct -> iterate gives back exactly the nonzero (index, value) pairs, for every
encoding) — there are no stored byte-level golden vectors in the reference, and the Rust crate cannot be
built here, so the byte layouts below are restated from the constructors, field by field:

  D3  `Dense3<u32>`   vec.rs:895-1026   u64 words of 21 3-bit fields, 7 = look in the fallback
  D4  `Dense4<u32>`   vec.rs:761-891    bytes of two nibbles (low nibble = even position), 15 = fallback
  D8  `DenseW<u8,_>`  vec.rs:660-757    one byte per position, 255 = fallback
  D16 `DenseW<u16,_>` vec.rs:660-757    one u16 per position, 65535 = fallback
  V   `SimpleSparse`  vec.rs:123-213    u32 indexes + u32 values
  S3/S4/S8 `CompressedIndexSparse<Dense3|Dense4|DenseW<u8>>` vec.rs:222-398: the values as a dense D3/D4/D8
      vector over the stored entries (fallback keyed by entry number), one byte per entry = index mod 256,
      `block_starts[b]` = first entry of the 256-index block b (round_up(len,256)/256 + 1 entries)

`AdaptiveVec.new` follows `choose_storage` (vec.rs:1086-1131) including its quirk that the S8 branch does
not lower `min_size` before the final comparison with V.  `iter()` follows `AbsIter::next` (vec.rs:96-117):
ascending positions, stored zeros skipped.
"""
from __future__ import annotations

import numpy as np

KINDS = ("D3", "D4", "D8", "D16", "V", "S3", "S4", "S8")  # order of `enum AdaptiveVec`, vec.rs:1029-1053
KIND_CODE = {k: i for i, k in enumerate(KINDS)}
_THRESH = {"3": 7, "4": 15, "8": 255, "16": 65535}


def _est_dense(width: str, length: int, values: np.ndarray) -> int:
    over = int(np.count_nonzero(values >= _THRESH[width]))
    if width == "3":  # vec.rs:965-970
        return (length // 21 + 1) * 64 // 8 + over * 8
    if width == "4":  # vec.rs:830-835
        return length // 2 + over * 8
    return length * (1 if width == "8" else 2) + over * 8  # vec.rs:723-728


def _est_sparse(width: str, length: int, values: np.ndarray) -> int:  # vec.rs:327-333
    return _est_dense(width, len(values), values) + len(values) + (length // 256) * 4


def choose_storage(length: int, values: np.ndarray) -> str:
    """vec.rs:1086-1131"""
    opt, min_size = "D3", _est_dense("3", length, values)
    for kind, sz in (("D4", _est_dense("4", length, values)), ("D8", _est_dense("8", length, values)),
                     ("D16", _est_dense("16", length, values)), ("S3", _est_sparse("3", length, values)),
                     ("S4", _est_sparse("4", length, values))):
        if sz < min_size:
            opt, min_size = kind, sz
    if _est_sparse("8", length, values) < min_size:
        opt = "S8"  # min_size deliberately not lowered (vec.rs:1120-1123)
    if len(values) * 8 < min_size:  # vec.rs:1125-1128
        opt = "V"
    return opt


class _Dense:
    """Dense3 / Dense4 / DenseW over `length` positions; fallback = SimpleSparse(indexes, values)."""

    def __init__(self, width: str, length: int, values: np.ndarray, indexes):
        values = np.asarray(values, dtype=np.uint32)
        idx = np.arange(length, dtype=np.uint32) if indexes is None else np.asarray(indexes, dtype=np.uint32)
        if indexes is None and len(values) != length:
            raise AssertionError("must supply a value for each position when indexes == None")
        self.width, self.length, self.nnz = width, length, len(values)
        th = _THRESH[width]
        over = values >= th
        small = np.where(over, th, values).astype(np.uint64)
        self.fallback_indexes = idx[over].astype(np.uint32)
        self.fallback_values = values[over].astype(np.uint32)
        if width == "3":
            data = np.zeros(length // 21 + 1, dtype=np.uint64)
            # later writes replace earlier ones at the same position (set_pos masks the field first, vec.rs:993-1004)
            pos = idx.astype(np.int64)
            last = _last_write(pos)
            np.bitwise_or.at(data, pos[last] // 21, small[last] << (3 * (pos[last] % 21)).astype(np.uint64))
            self.data = data
        elif width == "4":
            data = np.zeros(length // 2 + 1, dtype=np.uint8)
            pos = idx.astype(np.int64)
            last = _last_write(pos)
            np.bitwise_or.at(data, pos[last] >> 1, (small[last] << (4 * (pos[last] & 1)).astype(np.uint64)).astype(np.uint8))
            self.data = data
        else:
            data = np.zeros(length, dtype=np.uint8 if width == "8" else np.uint16)
            data[idx] = small.astype(data.dtype)  # numpy keeps the last assignment for repeated positions
            self.data = data

    def raw(self) -> np.ndarray:
        """the stored small fields of all positions"""
        n = self.length
        if self.width == "3":
            p = np.arange(n, dtype=np.int64)
            return ((self.data[p // 21] >> (3 * (p % 21)).astype(np.uint64)) & np.uint64(7)).astype(np.uint32)
        if self.width == "4":
            p = np.arange(n, dtype=np.int64)
            return ((self.data[p >> 1] >> (4 * (p & 1)).astype(np.uint8)) & 15).astype(np.uint32)
        return self.data[:n].astype(np.uint32)

    def values_at_all(self) -> np.ndarray:
        """`get(i)` for every position (vec.rs:913-926, 779-791, 678-685): the fallback is a binary search."""
        v = self.raw()
        ovf = np.nonzero(v == _THRESH[self.width])[0]
        if len(ovf):
            j = np.searchsorted(self.fallback_indexes, ovf.astype(np.uint32))
            hit = (j < len(self.fallback_indexes))
            hit[hit] &= self.fallback_indexes[j[hit]] == ovf[hit]
            out = np.zeros(len(ovf), dtype=np.uint32)  # SimpleSparse::get -> zero when absent
            out[hit] = self.fallback_values[j[hit]]
            v[ovf] = out
        return v


def _last_write(pos: np.ndarray) -> np.ndarray:
    """indices of the last occurrence of every distinct position, in input order"""
    if len(pos) == 0:
        return np.zeros(0, dtype=np.int64)
    rev = pos[::-1]
    _, first_rev = np.unique(rev, return_index=True)
    return np.sort(len(pos) - 1 - first_rev)


class AdaptiveVec:
    def __init__(self, kind: str, length: int):
        self.kind, self.length = kind, length
        self.dense = None            # D*: the vector itself; S*: the values over the stored entries
        self.indexes = self.values = None  # V
        self.index_bytes = self.block_starts = None  # S*

    # -- constructors (vec.rs:1135-1160) --------------------------------------------------------------
    @staticmethod
    def new(length: int, values, indexes) -> "AdaptiveVec":
        values = np.asarray(values, dtype=np.uint32)
        return AdaptiveVec.with_kind(choose_storage(length, values), length, values, indexes)

    @staticmethod
    def with_kind(kind: str, length: int, values, indexes) -> "AdaptiveVec":
        values = np.asarray(values, dtype=np.uint32)
        indexes = np.asarray(indexes, dtype=np.uint32)
        v = AdaptiveVec(kind, length)
        if kind == "V":
            v.indexes, v.values = indexes.copy(), values.copy()
        elif kind[0] == "D":
            v.dense = _Dense(kind[1:], length, values, indexes)
        else:  # CompressedIndexSparse::construct, vec.rs:335-397
            v.dense = _Dense(kind[1:], len(values), values, None)
            v.index_bytes = (indexes % 256).astype(np.uint8)
            total_blocks = (length + 255) // 256
            block = (indexes // 256).astype(np.int64)
            n_starts = max(total_blocks, int(block.max()) + 1 if len(block) else 0, 1) + 1
            # entry count before each block = start of the block; trailing entry = end of the last block
            v.block_starts = np.searchsorted(block, np.arange(n_starts), side="left").astype(np.uint32)
            v.block_starts[-1] = len(indexes)
        return v

    # -- decode ----------------------------------------------------------------------------------------
    def iter(self):
        """(positions, values) of `AdaptiveVec::iter` as two arrays: ascending, stored zeros skipped."""
        if self.kind == "V":
            keep = self.values != 0
            return self.indexes[keep].astype(np.uint32), self.values[keep].astype(np.uint32)
        if self.kind[0] == "D":
            vals = self.dense.values_at_all()
            pos = np.nonzero(vals)[0]
            return pos.astype(np.uint32), vals[pos].astype(np.uint32)
        vals = self.dense.values_at_all()
        n = len(self.index_bytes)
        entry = np.arange(n, dtype=np.int64)
        block = np.searchsorted(self.block_starts.astype(np.int64), entry, side="right") - 1
        pos = (block << 8) | self.index_bytes.astype(np.int64)
        keep = vals != 0
        return pos[keep].astype(np.uint32), vals[keep].astype(np.uint32)

    def nnz(self) -> int:  # stored entries, zeros included (vec.rs:1163-1165)
        if self.kind == "V":
            return len(self.indexes)
        return self.dense.nnz

    def mem_size(self) -> int:
        if self.kind == "V":
            return 8 * len(self.indexes)
        sz = self.dense.data.nbytes + 8 * len(self.dense.fallback_indexes)
        if self.kind[0] == "S":
            sz += len(self.index_bytes) + 4 * len(self.block_starts)
        return sz

    # -- the pieces a binding hands to scanrs_mat_create_adaptive (include/scanrs_amd.h) ------------
    def pieces(self) -> dict:
        if self.kind == "V":
            return dict(kind=KIND_CODE["V"], len=self.length, n_units=len(self.indexes), data=None,
                        fallback_indexes=self.indexes, fallback_values=self.values, index_bytes=None, block_starts=None)
        d = self.dense
        return dict(kind=KIND_CODE[self.kind], len=self.length, n_units=d.length, data=d.data,
                    fallback_indexes=d.fallback_indexes, fallback_values=d.fallback_values,
                    index_bytes=self.index_bytes, block_starts=self.block_starts)


def from_csmat(n_outer: int, n_inner: int, indptr, indices, data, kind: str | None = None):
    """`AdaptiveMat::from_csmat` (sqz/src/mat.rs:92-124): one AdaptiveVec per outer vector."""
    indptr = np.asarray(indptr, dtype=np.int64)
    out = []
    for o in range(n_outer):
        a, b = indptr[o], indptr[o + 1]
        if kind is None:
            out.append(AdaptiveVec.new(n_inner, data[a:b], indices[a:b]))
        else:
            out.append(AdaptiveVec.with_kind(kind, n_inner, data[a:b], indices[a:b]))
    return out
