"""Time of one sparse pass (b-wide product) in both orientations, with the raw map, the CellRanger map, and the
centred / scaled operator — what the map evaluation and the rank-1 offset cost on top of the bare gather."""
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
path = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
nnz = int(ip[-1].item())
if path:
    m.set_spmm_path(path)
del ip, ix, vv
xg = torch.randn(genes, l, device=dev, dtype=torch.float64)
xc = torch.randn(cells, l, device=dev, dtype=torch.float64)
og = torch.zeros(genes, l, device=dev, dtype=torch.float64)
oc = torch.zeros(cells, l, device=dev, dtype=torch.float64)


def t(fn, reps=5):
    fn()
    m.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    m.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def both(tag):
    a = t(lambda: m.dot_device(False, xc.data_ptr(), l, l, og.data_ptr(), l))   # A X: out rows = genes (gene-major copy)
    b = t(lambda: m.dot_device(True, xg.data_ptr(), l, l, oc.data_ptr(), l))    # A^T X: out rows = cells (cell-major copy)
    print(f"{tag:28s} gene-major pass {a:7.2f} ms ({a*1e6*256/nnz:5.2f} ns/nnz/CU)   cell-major pass {b:7.2f} ms ({b*1e6*256/nnz:5.2f} ns/nnz/CU)")


both("raw counts (no map)")
sa.log_normalize_with_size_factor(m, None, sa.FN_LOG2_1P)
both("scale + log2(1+x)")
m.reset_map()
sa.normalize(m, sa.Normalization.CellRanger)
both("CellRanger (scale, centre)")
