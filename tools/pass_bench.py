"""Time of one sparse pass (b-wide product) in both orientations under the CellRanger map (scale, log2, centre / scale),
for a list of product configurations: `path[:tile_k:tile_s:tile_t:tile_b:overlap]`, e.g. `0 3:2:32:48:4` (0 = default,
2 = L2-blocked gather, 3 = hybrid LDS tiles + gather); a configuration may carry handle options of its own behind slashes,
`0/tile_wtab=0` (A/B in one process, results compared with the first configuration's). usage: pass_bench.py [cells] [l] config..."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
genes, l = 33_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100
kw = {k: float(v) for k, v in (a.split("=") for a in sys.argv[3:] if "=" in a and "/" not in a and not a.startswith("opt."))}  # gene_shape=0.1 shared_profile=1: a heavy-tailed model
opts = {k[4:]: float(v) for k, v in (a.split("=") for a in sys.argv[3:] if a.startswith("opt."))}  # opt.tile_split_x=1.7: handle options
configs = [a for a in sys.argv[3:] if "=" not in a or "/" in a] or ["0"]
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev, **kw)
if kw:
    per_gene = torch.bincount(ix.long(), minlength=genes).float() / cells
    print("gene detection rates: max %.2f, genes above 10 %%: %d, their share of the nonzeros: %.0f %%" % (
        float(per_gene.max()), int((per_gene > 0.1).sum()), 100 * float(per_gene[per_gene > 0.1].sum() / per_gene.sum())), flush=True)
nnz = int(ip[-1].item())
xg = torch.randn(genes, l, device=dev, dtype=torch.float64)
xc = torch.randn(cells, l, device=dev, dtype=torch.float64)
og = torch.zeros(genes, l, device=dev, dtype=torch.float64)
oc = torch.zeros(cells, l, device=dev, dtype=torch.float64)
ref = {}

for cfg in configs:
    own = {k: float(v) for k, v in (a.split("=") for a in cfg.split("/")[1:])}
    parts = [int(x) for x in cfg.split("/")[0].split(":")]
    m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
    m.set_spmm_path(parts[0])
    for key, val in zip(("tile_k", "tile_s", "tile_t", "tile_b", "tile_overlap", "tile_ku", "ov_tile_kb"), parts[1:]):
        m.set_option(key, val)
    for key, val in {**opts, **own}.items():
        m.set_option(key, val)
    sa.normalize(m, sa.Normalization.CellRanger)

    def t(fn, reps=5):
        t0 = time.perf_counter()
        fn()
        m.sync()
        first = (time.perf_counter() - t0) * 1e3
        fn()
        m.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        m.sync()
        return (time.perf_counter() - t0) / reps * 1e3, first

    a, fa = t(lambda: m.dot_device(False, xc.data_ptr(), l, l, og.data_ptr(), l))  # A X: out rows = genes (gene-major copy)
    b, fb = t(lambda: m.dot_device(True, xg.data_ptr(), l, l, oc.data_ptr(), l))   # A^T X: out rows = cells (cell-major copy)
    err = ""
    if not ref:
        ref["g"], ref["c"] = og.clone(), oc.clone()
    else:
        eg = float((og - ref["g"]).abs().max() / ref["g"].abs().max())
        ec = float((oc - ref["c"]).abs().max() / ref["c"].abs().max())
        err = f"  max rel diff vs first config: {eg:.1e} / {ec:.1e}"
    m.profile_enable(True)
    m.profile_reset()
    m.dot_device(False, xc.data_ptr(), l, l, og.data_ptr(), l)
    m.dot_device(True, xg.data_ptr(), l, l, oc.data_ptr(), l)
    m.sync()
    prof = "; ".join(f"{name} x{st['launches']} {st['total_ms']:.2f} ms" for name, st in m.profile_get().items() if st["total_ms"] > 0.3)
    m.profile_enable(False)
    print(f"{cfg:14s} gene-major pass {a:7.2f} ms ({a*1e6*256/nnz:5.2f} ns/nnz/CU, first call {fa:7.1f} ms)   "
          f"cell-major pass {b:7.2f} ms ({b*1e6*256/nnz:5.2f} ns/nnz/CU, first call {fb:7.1f} ms){err}\n           {prof}", flush=True)
    del m
