"""debug helper: the tile path on raw counts against scipy, row by row; which part of the tile range does a wrong row miss?"""
import os, sys
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells, genes, l = int(sys.argv[1]), 33000, 100
opts = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:])}
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev)
A = sp.csr_matrix((vv.cpu().numpy().astype(np.float64), ix.cpu().numpy(), ip.cpu().numpy()), shape=(cells, genes))  # cells x genes
rng = np.random.default_rng(0)
xc_h = rng.standard_normal((cells, l))
xc = torch.from_numpy(xc_h).to(dev)
m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
m.set_spmm_path(3)
for k, v in opts.items():
    m.set_option(k, v)
ref = A.T @ xc_h  # genes x l
T = 48
nt = (cells + T - 1) // T
for call in range(3):
    og = torch.zeros(genes, l, device=dev, dtype=torch.float64)
    m.dot_device(False, xc.data_ptr(), l, l, og.data_ptr(), l)
    m.sync()
    a = og.cpu().numpy()
    err = np.abs(a - ref).max(axis=1) / np.abs(ref).max()
    bad = np.nonzero(err > 1e-9)[0]
    print(f"call {call}: bad rows {bad.size}; groups (row // 32): {sorted(set((bad // 32).tolist()))[:20]}", flush=True)
    if bad.size and call == 0:
        AT = A.T.tocsr()
        for r in bad[:3]:
            d = a[r] - ref[r]
            row = AT[r]
            cols, vals = row.indices, row.data
            # contribution of every tile of 48 cells to this row
            tile = cols // T
            contrib = np.zeros((nt, l))
            np.add.at(contrib, tile, vals[:, None] * xc_h[cols])
            # find a contiguous tile range [t0, t1) whose contribution equals -d (missing) by prefix sums
            pre = np.concatenate([np.zeros((1, l)), np.cumsum(contrib, axis=0)])
            best = None
            # coarse: which single tiles / ranges? test ranges that start and end anywhere (O(nt^2) too big): use the projection on column 0
            target = -d[0]
            p0 = pre[:, 0]
            idx = {}
            for t in range(nt + 1):
                idx.setdefault(round(p0[t], 6), t)
            for t1 in range(nt + 1):
                key = round(p0[t1] - target, 6)
                if key in idx and idx[key] < t1:
                    t0 = idx[key]
                    if np.allclose(pre[t1] - pre[t0], -d, rtol=1e-6, atol=1e-6):
                        best = (t0, t1)
                        break
            print(f"   row {r}: nnz {cols.size}, |d|/|ref| {np.abs(d).max() / np.abs(ref[r]).max():.3e}, missing tile range: {best}", flush=True)
