"""Synthetic sparse UMI count matrices of the shapes BASELINE.json names.

Model (after scan-rs/src/dim_red/test.rs:227-253, `gene_exp_sim_sprs_ex`): `n_clusters`
cluster profiles of per-gene rates ~ Gamma(0.4, 2.0); a cell belongs to one cluster and has a
log-normal depth factor; gene g is present in cell c with probability
min(1, depth_c * rate[cluster_c, g] * density / mean(rate)); stored counts are
1 + Geometric(0.6) (about 90 % of the values below 4, all well below the 4-bit design point
of sqz/src/vec.rs:758-759).  Cell-major: rows are cells, columns genes, indices ascending.

`synth_counts` (numpy, host) feeds the oracle and the parity tests; `synth_counts_torch`
builds the same model directly in device memory for bench.py (a different random stream).
"""
from __future__ import annotations

import numpy as np


def _profiles(rng, n_clusters, n_genes):
    return rng.gamma(0.4, 2.0, size=(n_clusters, n_genes))


def synth_counts(n_cells: int, n_genes: int, density: float, seed: int = 0, n_clusters: int = 20, chunk: int = 2048):
    """Returns a scipy.sparse.csr_matrix (n_cells x n_genes) with uint32 data, sorted indices."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    rates = _profiles(rng, n_clusters, n_genes)
    scale = density / rates.mean()
    cluster = rng.integers(0, n_clusters, size=n_cells)
    depth = np.exp(rng.normal(0.0, 0.3, size=n_cells))
    indptr = np.zeros(n_cells + 1, dtype=np.uint64)
    idx_parts, val_parts = [], []
    for c0 in range(0, n_cells, chunk):
        c1 = min(n_cells, c0 + chunk)
        p = depth[c0:c1, None] * rates[cluster[c0:c1], :] * scale
        mask = rng.random((c1 - c0, n_genes)) < p
        rows, cols = np.nonzero(mask)
        counts = np.bincount(rows, minlength=c1 - c0)
        indptr[c0 + 1:c1 + 1] = counts
        idx_parts.append(cols.astype(np.uint32))
        val_parts.append(rng.geometric(0.6, size=cols.shape[0]).astype(np.uint32))
    indptr = np.cumsum(indptr, dtype=np.uint64)
    indices = np.concatenate(idx_parts) if idx_parts else np.zeros(0, dtype=np.uint32)
    values = np.concatenate(val_parts) if val_parts else np.zeros(0, dtype=np.uint32)
    m = sp.csr_matrix((values, indices.astype(np.int64), indptr.astype(np.int64)), shape=(n_cells, n_genes))
    m.has_sorted_indices = True
    return m


def synth_counts_fast(n_cells: int, n_genes: int, density: float, seed: int = 0, n_clusters: int = 20, chunk: int = 8192,
                      gene_shape: float = 0.4, shared_profile: float = 0.0):
    """The same model in its Poisson-process form, ~20x faster on the host (used for the BASELINE-size fixtures, which
    both the fixture script and the GPU test have to regenerate): gene g is present in cell c when a Poisson process
    of intensity depth_c * rate[cluster_c, g] * density / mean(rate) has at least one event, i.e. with probability
    1 - exp(-depth * rate * scale) (= depth * rate * scale to first order, the Bernoulli probability of `synth_counts`).
    Per cell: N ~ Poisson(depth * sum of the cluster's scaled rates) events, each assigned to a gene by inverting the
    cluster's cumulative rate table (the positions of a cell drawn already sorted, as normalised partial sums of
    exponential spacings), duplicates merged. Counts 1 + Geometric(0.6) as before. A different random stream
    from `synth_counts`, deterministic for a given numpy Generator implementation."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    rates = _profiles(rng, n_clusters, n_genes)
    if gene_shape != 0.4 or shared_profile > 0.0:
        # heavy-tailed variant (as `synth_counts_torch`): gene rates ~ Gamma(gene_shape), a part of the profile shared by all clusters —
        # a few thousand genes detected in 10-100 % of the cells and most others in far below 1 %, as real 10x matrices have them
        base = rng.gamma(gene_shape, 1.0, size=(1, n_genes))
        own = rng.gamma(gene_shape, 1.0, size=(n_clusters, n_genes))
        rates = shared_profile * base + (1.0 - shared_profile) * own
    scale = density / rates.mean()
    cum = np.cumsum(rates * scale, axis=1)  # n_clusters x n_genes
    total = cum[:, -1].copy()
    big = float(np.ceil(total.max())) + 1.0
    cum_flat = (cum + big * np.arange(n_clusters)[:, None]).ravel()  # one increasing table for all clusters
    cluster = rng.integers(0, n_clusters, size=n_cells)
    depth = np.exp(rng.normal(0.0, 0.3, size=n_cells))
    indptr = np.zeros(n_cells + 1, dtype=np.int64)
    idx_parts, val_parts = [], []
    for c0 in range(0, n_cells, chunk):
        c1 = min(n_cells, c0 + chunk)
        cl = cluster[c0:c1]
        n_ev = rng.poisson(depth[c0:c1] * total[cl])
        # the N event positions of a cell in ascending order = normalised partial sums of N + 1 exponential spacings
        starts = np.concatenate(([0], np.cumsum(n_ev + 1)))
        e = rng.standard_exponential(int(starts[-1]))
        cs = np.cumsum(e)
        seg_base = np.concatenate(([0.0], cs[starts[1:-1] - 1]))  # partial sum before each cell's first spacing
        seg_total = cs[starts[1:] - 1] - seg_base
        keep = np.ones(e.shape[0], dtype=bool)
        keep[starts[1:] - 1] = False  # the (N+1)-th spacing only closes the interval
        cell = np.repeat(np.arange(c1 - c0), n_ev)
        u = (cs[keep] - seg_base[cell]) / seg_total[cell] * total[cl][cell]
        gene = np.searchsorted(cum_flat, u + big * cl[cell], side="right") - cl[cell] * n_genes
        np.clip(gene, 0, n_genes - 1, out=gene)
        first = np.ones(gene.shape[0], dtype=bool)  # ascending per cell: a duplicate equals its predecessor in the same cell
        first[1:] = (gene[1:] != gene[:-1]) | (cell[1:] != cell[:-1])
        indptr[c0 + 1:c1 + 1] = np.bincount(cell[first], minlength=c1 - c0)
        idx_parts.append(gene[first].astype(np.uint32))
        val_parts.append(rng.geometric(0.6, size=int(first.sum())).astype(np.uint32))
    indptr = np.cumsum(indptr)
    indices = np.concatenate(idx_parts) if idx_parts else np.zeros(0, dtype=np.uint32)
    values = np.concatenate(val_parts) if val_parts else np.zeros(0, dtype=np.uint32)
    m = sp.csr_matrix((values, indices.astype(np.int64), indptr), shape=(n_cells, n_genes))
    m.has_sorted_indices = True
    return m


def _par_chunk(args):
    """cells [c0, c1) of `synth_counts_par`: its own generator (seed, chunk id), Poisson-process form of the model"""
    seed, ci, c0, c1, n_genes, cum_flat, total, big, n_clusters = args
    rng = np.random.default_rng([seed, ci])
    cl = rng.integers(0, n_clusters, size=c1 - c0)
    depth = np.exp(rng.normal(0.0, 0.3, size=c1 - c0))
    n_ev = rng.poisson(depth * total[cl])
    starts = np.concatenate(([0], np.cumsum(n_ev + 1)))
    e = rng.standard_exponential(int(starts[-1]))
    cs = np.cumsum(e)
    seg_base = np.concatenate(([0.0], cs[starts[1:-1] - 1]))
    seg_total = cs[starts[1:] - 1] - seg_base
    keep = np.ones(e.shape[0], dtype=bool)
    keep[starts[1:] - 1] = False
    cell = np.repeat(np.arange(c1 - c0), n_ev)
    u = (cs[keep] - seg_base[cell]) / seg_total[cell] * total[cl][cell]
    gene = np.searchsorted(cum_flat, u + big * cl[cell], side="right") - cl[cell] * n_genes
    np.clip(gene, 0, n_genes - 1, out=gene)
    first = np.ones(gene.shape[0], dtype=bool)
    first[1:] = (gene[1:] != gene[:-1]) | (cell[1:] != cell[:-1])
    counts = np.bincount(cell[first], minlength=c1 - c0).astype(np.int64)
    return counts, gene[first].astype(np.uint32), rng.geometric(0.6, size=int(first.sum())).astype(np.uint32)


def synth_counts_par(n_cells: int, n_genes: int, density: float, seed: int = 0, n_clusters: int = 20, chunk: int = 8192, workers: int = 0,
                     _in_helper: bool = False):
    """The model of `synth_counts_fast` with one generator per chunk of `chunk` cells (seed sequence [seed, chunk id]), so the
    chunks can be drawn by a pool of processes: the million-cell fixture input (10^9 nonzeros) in well under a minute on a
    many-core host instead of five. Returns (indptr uint64[n_cells + 1], indices uint32[nnz], values uint32[nnz]), cell-major,
    indices ascending per cell. The result does not depend on `workers`."""
    import multiprocessing as mp
    import os

    rng = np.random.default_rng(seed)
    rates = _profiles(rng, n_clusters, n_genes)
    scale = density / rates.mean()
    cum = np.cumsum(rates * scale, axis=1)
    total = cum[:, -1].copy()
    big = float(np.ceil(total.max())) + 1.0
    cum_flat = (cum + big * np.arange(n_clusters)[:, None]).ravel()
    jobs = [(seed, ci, c0, min(n_cells, c0 + chunk), n_genes, cum_flat, total, big, n_clusters)
            for ci, c0 in enumerate(range(0, n_cells, chunk))]
    workers = workers or min(len(jobs), max(1, (os.cpu_count() or 1) - 1), 64)
    if workers > 1 and len(jobs) > 1:
        if not _in_helper:
            # Never fork a pool from the calling process: in a `-m gpu` test run it has initialised HIP and carries the runtime's
            # threads, and a forked child can inherit a lock whose owner does not exist in it (pool.map then never returns —
            # the intermittent stall of round 3). The pool lives in a FRESH interpreter that has only numpy loaded; the arrays
            # come back through .npy files in a scratch directory.
            return _par_in_subprocess(n_cells, n_genes, density, seed, n_clusters, chunk, workers)
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_par_chunk, jobs, chunksize=1)
    else:
        parts = [_par_chunk(j) for j in jobs]
    indptr = np.zeros(n_cells + 1, dtype=np.uint64)
    np.cumsum(np.concatenate([p[0] for p in parts]), out=indptr[1:])
    indices = np.concatenate([p[1] for p in parts])
    values = np.concatenate([p[2] for p in parts])
    return indptr, indices, values


def _par_in_subprocess(n_cells, n_genes, density, seed, n_clusters, chunk, workers):
    """`synth_counts_par` in a child interpreter started with subprocess (this file run as a script: numpy only, no HIP)."""
    import os
    import shutil
    import subprocess
    import sys
    import tempfile

    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    out = tempfile.mkdtemp(prefix="scanrs_synth_", dir=base)
    try:
        subprocess.run([sys.executable, os.path.abspath(__file__), "par", str(n_cells), str(n_genes), repr(float(density)), str(seed),
                        str(n_clusters), str(chunk), str(workers), out], check=True, timeout=3600)
        return tuple(np.load(os.path.join(out, name + ".npy")) for name in ("indptr", "indices", "values"))
    finally:
        shutil.rmtree(out, ignore_errors=True)


def synth_counts_torch(n_cells: int, n_genes: int, density: float, seed: int, device, cell_begin: int = 0,
                       cell_end: int | None = None, n_clusters: int = 20, chunk: int = 4096, gene_shape: float = 0.4,
                       shared_profile: float = 0.0):
    """Device-side generator: returns (indptr int64[n_local+1], indices int32[nnz], values int32[nnz])
    for cells [cell_begin, cell_end) of the global matrix. Every chunk of `chunk` global cells has its own
    generator seed, so any partition of the cells over ranks yields the same global matrix."""
    import torch

    cell_end = n_cells if cell_end is None else cell_end
    g0 = torch.Generator(device=device)
    g0.manual_seed(seed)
    conc = torch.full((n_clusters, n_genes), 0.4, device=device, dtype=torch.float32)
    # Gamma(0.4, scale 2.0): standard gamma * 2 (drawn once, identically on every rank)
    torch.manual_seed(seed)
    rates = torch._standard_gamma(conc) * 2.0
    if gene_shape != 0.4 or shared_profile > 0.0:
        # experiments only (tools/pass_bench.py): a heavier-tailed gene profile (Gamma shape below 0.4) and / or a part of it
        # shared by all clusters — genes detected in most cells, as real data has them; the default model is untouched
        base = torch._standard_gamma(torch.full((1, n_genes), gene_shape, device=device, dtype=torch.float32))
        own = torch._standard_gamma(torch.full((n_clusters, n_genes), gene_shape, device=device, dtype=torch.float32))
        rates = shared_profile * base + (1.0 - shared_profile) * own
    scale = density / float(rates.mean())
    counts_parts, idx_parts, val_parts = [], [], []
    first_chunk = cell_begin // chunk
    last_chunk = (cell_end + chunk - 1) // chunk
    for ci in range(first_chunk, last_chunk):
        c0, c1 = ci * chunk, min(n_cells, (ci + 1) * chunk)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000003 + 17 * ci + 1)
        n = c1 - c0
        cluster = torch.randint(0, n_clusters, (n,), device=device, generator=g)
        depth = torch.exp(torch.randn(n, device=device, generator=g) * 0.3)
        p = depth[:, None] * rates[cluster, :] * scale
        mask = torch.rand((n, n_genes), device=device, generator=g) < p
        nz = mask.nonzero()
        u = torch.rand(nz.shape[0], device=device, generator=g).clamp_(min=1e-12)
        # Geometric(0.6) on {1, 2, ...}: 1 + floor(log(u) / log(1 - 0.6))
        vals = (1 + torch.floor(torch.log(u) / np.log(0.4))).to(torch.int32)
        # the whole chunk is drawn, then the rows of this rank are kept: the matrix does not depend on the partition
        lo, hi = max(c0, cell_begin) - c0, min(c1, cell_end) - c0
        keep = (nz[:, 0] >= lo) & (nz[:, 0] < hi)
        counts_parts.append(mask[lo:hi].sum(dim=1))
        idx_parts.append(nz[keep, 1].to(torch.int32))
        val_parts.append(vals[keep])
        del p, mask, nz, u
    counts = torch.cat(counts_parts)
    indptr = torch.zeros(counts.shape[0] + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, dim=0)
    return indptr, torch.cat(idx_parts), torch.cat(val_parts)


if __name__ == "__main__":  # helper mode of synth_counts_par: `synth.py par cells genes density seed clusters chunk workers outdir`
    import os
    import sys

    if len(sys.argv) == 10 and sys.argv[1] == "par":
        a = sys.argv
        ip_, ix_, vv_ = synth_counts_par(int(a[2]), int(a[3]), float(a[4]), int(a[5]), int(a[6]), int(a[7]), int(a[8]), _in_helper=True)
        for name_, arr_ in (("indptr", ip_), ("indices", ix_), ("values", vv_)):
            np.save(os.path.join(a[9], name_ + ".npy"), arr_)
    else:
        raise SystemExit("usage: synth.py par cells genes density seed clusters chunk workers outdir")
