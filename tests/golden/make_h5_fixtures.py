"""Writes the HDF5 fixtures the reader tests use, with the real HDF5 library (h5py 3.3 / libhdf5 1.10.6 of the build
container's /opt/conda: `/opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py`), and the expected contents as JSON.

The reference's own fixtures for this reader (hdf5-io/test/ER-0114-T3.h5, empty.h5; hdf5-io/src/matrix.rs:301-360) are
git-LFS stubs in this checkout, so these files stand in for them: the same group layout Cell Ranger writes
(`matrix/{barcodes,data,indices,indptr,shape,features/{id,name,feature_type,genome}}`, chunked + shuffle + gzip), the
"indices not sorted within a column" case of test_cr3_matrix_reverse_sorted, the empty matrix of test_empty_matrix, and
an analysis file with the groups hdf5-io/src/analysis.rs reads. `formats.h5` walks the storage variants a reader meets
(compact / contiguous / chunked, one- and two-level chunk B-trees, edge chunks in 2-D, fletcher32, big-endian, narrow
integers, f32) so that the from-scratch parser is checked against bytes the real library produced.
"""
import json
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(7)

N_FEAT, N_CELL = 30, 40
FEATURE_TYPES = ["Gene Expression"] * 24 + ["Antibody Capture"] * 4 + ["CRISPR Guide Capture"] * 2


def tiny_matrix():
    dense = rng.poisson(0.35, size=(N_FEAT, N_CELL)).astype(np.int64)
    dense[3, :] = 0                      # an all-zero feature (shrink_row filter)
    dense[5, 7] = 71                     # the "MALAT1 / AAACCTGAGCTAGTGG-1 == 71" probe of the reference test
    dense[:, 11] = 0                     # an empty cell
    dense[9, 2] = 70000                  # needs more than 16 bits
    indptr = [0]
    indices, data = [], []
    for c in range(N_CELL):
        r = np.nonzero(dense[:, c])[0]
        indices.extend(r.tolist())
        data.extend(dense[r, c].tolist())
        indptr.append(len(indices))
    barcodes = ["AAACCTGAGCTAGTGG-1" if c == 7 else "".join("ACGT"[(c * 7 + i * 3 + (c >> i)) % 4] for i in range(16)) + "-1" for c in range(N_CELL)]
    ids = [f"ENSG{100000 + i:011d}" if FEATURE_TYPES[i] == "Gene Expression" else f"FB{i:03d}" for i in range(N_FEAT)]
    names = ["MALAT1" if i == 5 else f"GENE{i}" for i in range(N_FEAT)]
    return dense, np.array(indptr), np.array(indices), np.array(data), barcodes, ids, names


def write_matrix_group(f, indptr, indices, data, barcodes, ids, names, *, chunked, barcodes_null=False):
    kw = dict(chunks=True, compression="gzip", compression_opts=4, shuffle=True) if chunked else {}
    g = f.create_group("matrix")
    if barcodes_null:
        g.create_dataset("barcodes", data=h5py.Empty("S18"))
    else:
        g.create_dataset("barcodes", data=np.array(barcodes, dtype="S18"), **kw)
    small = dict(kw)
    if chunked:
        small["chunks"] = (7,)  # many chunks: the chunk B-tree of `data` / `indices` gets two levels
    g.create_dataset("data", data=data.astype(np.int32), maxshape=(None,) if chunked else None, **small)
    g.create_dataset("indices", data=indices.astype(np.int64), maxshape=(None,) if chunked else None, **small)
    g.create_dataset("indptr", data=indptr.astype(np.int64), **kw)
    g.create_dataset("shape", data=np.array([len(ids), len(barcodes)], dtype=np.int32))
    ft = g.create_group("features")
    ft.create_dataset("id", data=np.array(ids, dtype="S16"), **kw)
    ft.create_dataset("name", data=np.array(names, dtype="S10"), **kw)
    ft.create_dataset("feature_type", data=np.array(FEATURE_TYPES[: len(ids)], dtype="S20"), **kw)
    ft.create_dataset("genome", data=np.array(["GRCh38"] * len(ids), dtype="S6"), **kw)
    ft.create_dataset("_all_tag_keys", data=np.array(["genome"], dtype="S6"))
    f.attrs["filetype"] = "matrix"
    f.attrs["version"] = 2
    f.attrs["chemistry_description"] = "Single Cell 3' v3"
    f.attrs["library_ids"] = np.array(["lib0"], dtype="S4")


def main():
    dense, indptr, indices, data, barcodes, ids, names = tiny_matrix()
    expected = {
        "tiny_10x": {
            "n_features": N_FEAT, "n_cells": N_CELL, "indptr": indptr.tolist(), "indices": indices.tolist(), "data": data.tolist(),
            "barcodes": barcodes, "feature_ids": ids, "feature_names": names, "feature_types": FEATURE_TYPES,
            "dense": dense.tolist(), "umi_counts": dense.sum(axis=0).tolist(),
        }
    }
    with h5py.File(os.path.join(HERE, "tiny_10x.h5"), "w") as f:
        write_matrix_group(f, indptr, indices, data, barcodes, ids, names, chunked=True)

    # same matrix, indices descending inside every column (what some Cell Ranger 3 files hold), contiguous storage
    ridx, rdat = indices.copy(), data.copy()
    for c in range(N_CELL):
        s, e = indptr[c], indptr[c + 1]
        ridx[s:e] = indices[s:e][::-1]
        rdat[s:e] = data[s:e][::-1]
    with h5py.File(os.path.join(HERE, "tiny_10x_unsorted.h5"), "w") as f:
        write_matrix_group(f, indptr, ridx, rdat, barcodes, ids, names, chunked=False)

    # no barcodes at all: `barcodes` has a null dataspace (matrix.rs:259-262), no nonzeros
    with h5py.File(os.path.join(HERE, "empty_10x.h5"), "w") as f:
        write_matrix_group(f, np.array([0]), np.array([], dtype=np.int64), np.array([], dtype=np.int64), [], ids, names, chunked=False, barcodes_null=True)

    # analysis file: clustering/<key>/{clusters, num_clusters}, all_differential_expression/<key>/data, matrix/features
    clusters = {"graphclust": rng.integers(1, 6, size=N_CELL), "kmeans_2_clusters": rng.integers(1, 3, size=N_CELL)}
    de = {k: rng.normal(size=(N_FEAT, 3 * int(v.max()))) for k, v in clusters.items()}
    with h5py.File(os.path.join(HERE, "tiny_analysis.h5"), "w") as f:
        for k, v in clusters.items():
            g = f.create_group(f"clustering/_{k}")
            g.create_dataset("clusters", data=v.astype(np.int64), chunks=(16,), compression="gzip", shuffle=True)
            g.create_dataset("num_clusters", data=np.int64(v.max()))
            g.create_dataset("clustering_type", data=np.bytes_(k))
            d = f.create_group(f"all_differential_expression/_{k}")
            d.create_dataset("data", data=de[k], chunks=(8, 4), compression="gzip", shuffle=True)
        ft = f.create_group("matrix/features")
        ft.create_dataset("id", data=np.array(ids, dtype="S16"))
        ft.create_dataset("name", data=np.array(names, dtype="S10"))
    expected["tiny_analysis"] = {
        "keys": sorted(f"_{k}" for k in clusters),
        "clusters": {f"_{k}": v.tolist() for k, v in clusters.items()},
        "num_clusters": {f"_{k}": int(v.max()) for k, v in clusters.items()},
        "de": {f"_{k}": v.tolist() for k, v in de.items()},
    }

    # storage variants
    fm = {}
    with h5py.File(os.path.join(HERE, "formats.h5"), "w") as f:
        a = rng.integers(0, 200, size=13).astype(np.uint8)
        f.create_dataset("u8_contig", data=a)
        fm["u8_contig"] = a.tolist()
        a = rng.integers(-30000, 30000, size=50).astype(">i2")
        f.create_dataset("i16_be", data=a)
        fm["i16_be"] = a.astype(np.int64).tolist()
        a = rng.integers(0, 2**40, size=23).astype(">u8")
        f.create_dataset("u64_be_chunked", data=a, chunks=(5,), fletcher32=True)
        fm["u64_be_chunked"] = a.astype(np.uint64).tolist()
        a = rng.normal(size=17).astype(np.float32)
        f.create_dataset("f32_gzip", data=a, chunks=(4,), compression="gzip", compression_opts=9)
        fm["f32_gzip"] = a.astype(np.float64).tolist()
        a = rng.normal(size=(11, 7))
        f.create_dataset("f64_2d_edge", data=a, chunks=(4, 3), compression="gzip", shuffle=True, fletcher32=True)
        fm["f64_2d_edge"] = a.tolist()
        a = rng.integers(0, 1000, size=(3, 4)).astype(np.int32)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        sid = h5py.h5s.create_simple(a.shape)
        did = h5py.h5d.create(f.id, b"i32_compact", h5py.h5t.NATIVE_INT32, sid, dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, a)
        fm["i32_compact"] = a.tolist()
        a = rng.integers(0, 2**31, size=1000).astype(np.uint32)
        f.create_dataset("u32_many_chunks", data=a, chunks=(3,), maxshape=(None,))  # 334 chunks: 3 B-tree levels with K=32? at least 2
        fm["u32_many_chunks"] = a.tolist()
        f.create_dataset("scalar_i64", data=np.int64(-12345))
        fm["scalar_i64"] = -12345
        a = np.array(["alpha", "be", "", "gamma-delta"], dtype="S11")
        f.create_dataset("strings", data=a)
        fm["strings"] = [s.decode() for s in a]
        g = f.create_group("deep/er/group")
        g.create_dataset("x", data=np.arange(5, dtype=np.int64))
        for i in range(40):  # more links than one symbol-table node holds (2K = 8 by default) -> multi-node group B-tree
            f.create_dataset(f"many/m{i:02d}", data=np.array([i], dtype=np.int16))
        fm["many"] = sorted(f"m{i:02d}" for i in range(40))
    expected["formats"] = fm

    # libver="latest": version-2 object headers, version-3 superblock, link messages, v4 layouts
    with h5py.File(os.path.join(HERE, "tiny_10x_latest.h5"), "w", libver="latest") as f:
        write_matrix_group(f, indptr, indices, data, barcodes, ids, names, chunked=True)

    # version-4 chunk indexes: fixed array (plain, filtered, paged: > 1024 chunks) and implicit (early allocation, no filter)
    fl = {}
    with h5py.File(os.path.join(HERE, "formats_latest.h5"), "w", libver="latest") as f:
        a = rng.integers(0, 2**31, size=50).astype(np.int64)
        f.create_dataset("fixed_array", data=a, chunks=(8,))
        fl["fixed_array"] = a.tolist()
        a = rng.normal(size=(9, 10))
        f.create_dataset("fixed_array_filtered_2d", data=a, chunks=(4, 4), compression="gzip", shuffle=True)
        fl["fixed_array_filtered_2d"] = a.tolist()
        a = rng.integers(0, 60000, size=2500).astype(np.uint16)
        f.create_dataset("fixed_array_paged", data=a, chunks=(2,))
        fl["fixed_array_paged"] = a.tolist()
        a = rng.integers(0, 255, size=(6, 5)).astype(np.uint8)
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_chunk((4, 2))
        dcpl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
        did = h5py.h5d.create(f.id, b"implicit", h5py.h5t.NATIVE_UINT8, h5py.h5s.create_simple(a.shape), dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, a)
        fl["implicit"] = a.tolist()
        # (own group: more than 8 links in one new-style group would switch it to dense storage, which the reader refuses)
        # extensible array (one unlimited dimension): index block only; data blocks reached from the index block, filtered;
        # real super blocks (> 244 chunks); two-dimensional with the unlimited dimension first / last (swizzled linear index)
        a = rng.integers(0, 1000, size=5).astype(np.int32)
        f.create_dataset("ea/ea_small", data=a, chunks=(2,), maxshape=(None,))
        fl["ea/ea_small"] = a.tolist()
        a = rng.normal(size=400)
        f.create_dataset("ea/ea_filtered", data=a, chunks=(2,), maxshape=(None,), compression="gzip", shuffle=True)
        fl["ea/ea_filtered"] = a.tolist()
        a = rng.integers(0, 60000, size=3001).astype(np.uint16)
        f.create_dataset("ea/ea_super", data=a, chunks=(2,), maxshape=(None,))
        fl["ea/ea_super"] = a.tolist()
        a = rng.integers(0, 255, size=(37, 7)).astype(np.uint8)
        f.create_dataset("ea/ea_2d_unlim0", data=a, chunks=(2, 3), maxshape=(None, 7))
        fl["ea/ea_2d_unlim0"] = a.tolist()
        a = rng.integers(0, 255, size=(9, 41)).astype(np.uint8)
        f.create_dataset("ea/ea_2d_unlim1", data=a, chunks=(4, 2), maxshape=(12, None))
        fl["ea/ea_2d_unlim1"] = a.tolist()
        g = f.create_group("grp")
        for i in range(6):
            g.create_dataset(f"d{i}", data=np.array([i * 1.5]))
        fl["grp"] = [f"d{i}" for i in range(6)]
    expected["formats_latest"] = fl

    with open(os.path.join(HERE, "h5_fixtures_expected.json"), "w") as fh:
        json.dump(expected, fh)
    for n in sorted(os.listdir(HERE)):
        if n.endswith(".h5"):
            print(n, os.path.getsize(os.path.join(HERE, n)))


if __name__ == "__main__":
    main()
