"""Simulation of the dense record layout of the tile kernel (round 5, VERDICT r4 item 1, stage ii): positions per nonzero and
what the shared barrier costs when the 8 waves of a workgroup item run the SAME number of record chunks per visit.

Model of the kernel: a wave owns 32 accumulators ("slots"); visit v happens when panel tile v has landed (tiles v-2 .. v in the
ring), so a nonzero of tile t may be worked in visits t .. t+2; the records of a (wave, visit) are a dense list, worked in
chunks of `g` positions; every wave of the item runs N_v chunks in visit v (one number per item and visit: the barrier waits
for nobody), a wave without enough records available pads with null records.

usage: sim_dense_layout.py [cells] [gene_shape shared_profile]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from scanrs_amd.synth import synth_counts_fast  # noqa: E402

T, S, NW = 48, 32, 8


def slots_of(nnz_per_vec, n_inner, x_target, x_min):
    x = nnz_per_vec * (T / n_inner)
    v = np.maximum(1, np.floor(x / x_target + 0.5)).astype(np.int64)
    v[x < x_min] = 0
    return v


def arrivals(indptr, indices, n_inner, x_target, x_min, order="natural"):
    """per (group, tile) record counts; returns (a[groups, tiles], n_overflow_sparse, n_slots)"""
    n_outer = len(indptr) - 1
    nnz_per = np.diff(indptr)
    V = slots_of(nnz_per, n_inner, x_target, x_min)
    slot_first = np.concatenate(([0], np.cumsum(V)))
    n_slots = int(slot_first[-1])
    vec = np.repeat(np.arange(n_outer), nnz_per)
    pos = np.arange(len(indices)) - np.repeat(indptr[:-1], nnz_per)
    keep = V[vec] > 0
    slot = slot_first[vec[keep]] + pos[keep] % V[vec[keep]]
    tile = indices[keep] // T
    nt = (n_inner + T - 1) // T
    if order == "sorted":  # slots ordered by their load, heaviest first
        load = np.bincount(slot, minlength=n_slots)
        rank = np.empty(n_slots, dtype=np.int64)
        rank[np.argsort(-load, kind="stable")] = np.arange(n_slots)
        slot = rank[slot]
    group = slot // S
    ng = (n_slots + S - 1) // S
    a = np.bincount(group * nt + tile, minlength=ng * nt).reshape(ng, nt)
    return a, int((~keep).sum()), n_slots


def simulate(a, g, policy, alpha=1.0, tpp=None, delay=False):
    """a[groups, tiles] -> (positions incl. padding, nonzeros, visits, sum of N_v, dropped)"""
    ng, nt = a.shape
    ni = (ng + NW - 1) // NW
    pad = ni * NW - ng
    if pad:
        a = np.vstack([a, np.zeros((pad, nt), dtype=a.dtype)])
    a = a.reshape(ni, NW, nt)
    tpp = tpp or nt
    mid = np.zeros((ni, NW), dtype=np.int64)  # unworked records of tile v-1 / v-2 when visit v starts
    old = np.zeros_like(mid)
    tot_chunks = 0
    for v in range(nt):
        last = (v + 1) % tpp == 0 or v == nt - 1
        new = a[:, :, v].astype(np.int64)
        avail = (mid + old) if (delay and not last) else new + mid + old  # delay: a tile staged during visit v is first read in visit v + 1
        forced = avail if last else old
        need = -(-forced.max(axis=1) // g)  # chunks forced by the deadlines
        if policy == "eager":
            want = -(-avail.max(axis=1) // g)
        elif policy == "mean":
            want = np.floor(avail.mean(axis=1) * alpha / g + 0.5).astype(np.int64)
        elif policy == "lazy":
            want = need
        elif policy == "q75":  # the wave with the 3rd most records sets the length
            want = -(-np.sort(avail, axis=1)[:, -3] // g)
        else:
            raise ValueError(policy)
        N = np.maximum(need, want)
        cap = (N * g)[:, None] + 0 * old
        d = np.minimum(old, cap); cap = cap - d; old = old - d
        assert not old.any()
        d = np.minimum(mid, cap); cap = cap - d; mid = mid - d
        if delay and not last:
            d = 0 * new
        else:
            d = np.minimum(new, cap)
        new = new - d
        old, mid = mid, new
        if last:
            assert not old.any() and not mid.any()
        tot_chunks += int(N.sum())
    return tot_chunks * g * NW, int(a.sum()), ni * nt, tot_chunks


def main():
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
    kw = {}
    if len(sys.argv) > 3:
        kw = dict(gene_shape=float(sys.argv[2]), shared_profile=float(sys.argv[3]))
    genes = 33000
    t0 = time.time()
    m = synth_counts_fast(cells, genes, 0.03, seed=1, **kw)
    print(f"{cells} cells x {genes} genes, nnz {m.nnz} ({time.time() - t0:.0f} s) {kw}")
    mc = m.tocsc()
    mc.sort_indices()
    for name, ip, ix, n_inner, tpp in (("cell-major", m.indptr, m.indices, genes, None), ("gene-major", mc.indptr, mc.indices, cells, 672)):
        for x_target, x_min, order in ((1.8, 0.5, "natural"), (1.5, 0.5, "natural"), (1.5, 0.5, "sorted"), (1.5, 0.25, "sorted"), (1.2, 0.25, "sorted")):
            a, n_sparse, n_slots = arrivals(ip.astype(np.int64), ix.astype(np.int64), n_inner, x_target, x_min, order)
            nnz = len(ix)
            print(f"== {name}: x_target {x_target} x_min {x_min} order {order}: {n_slots} slots, {a.shape[0]} groups x {a.shape[1]} tiles, "
                  f"sparse-vector overflow {100 * n_sparse / nnz:.2f} %, records per wave-visit mean {a.mean():.1f} sd {a.std():.1f}")
            for g in (16, 8, 4):
                for policy, alpha in (("eager", 1), ("lazy", 1), ("mean", 1.0), ("mean", 1.05), ("q75", 1)):
                    pos, n, visits, chunks = simulate(a, g, policy, alpha, tpp)
                    pos2, _, _, _ = simulate(a, g, policy, alpha, tpp, delay=True)
                    print(f"   g {g:2d} {policy:5s} a {alpha:4.2f}: positions per tile-served nonzero {pos / n:.3f}; per all nonzeros {pos / nnz:.3f}; chunks per visit {chunks / visits:.2f}; tiles read one visit late: {pos2 / n:.3f}")


if __name__ == "__main__":
    main()
