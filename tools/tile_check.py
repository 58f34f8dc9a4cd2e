"""Hybrid product against the gather kernels on a large synthetic matrix: both orientations, several panel widths (incl. the
column-chunked wide ones). usage: tile_check.py [cells] [density] [widths...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
density = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
widths = [int(x) for x in sys.argv[3:]] or [100, 200]
genes = 33_000
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, density, 0, dev)
print("nnz", int(ip[-1].item()), flush=True)
outs = {}
for auto in (0, 1):
    m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
    m.set_option("tile_auto", auto)
    if auto:
        m.set_spmm_path(3)
    sa.normalize(m, sa.Normalization.CellRanger)
    for l in widths:
        g = torch.Generator(device=dev)
        g.manual_seed(l)
        xc = torch.randn(cells, l, device=dev, dtype=torch.float64, generator=g)
        xg = torch.randn(genes, l, device=dev, dtype=torch.float64, generator=g)
        og = torch.zeros(genes, l, device=dev, dtype=torch.float64)
        oc = torch.zeros(cells, l, device=dev, dtype=torch.float64)
        m.dot_device(False, xc.data_ptr(), l, l, og.data_ptr(), l)
        m.dot_device(True, xg.data_ptr(), l, l, oc.data_ptr(), l)
        m.sync()
        if auto == 0:
            outs[l] = (og.clone(), oc.clone())
        else:
            rg, rc = outs[l]
            eg = float((og - rg).abs().max() / rg.abs().max())
            ec = float((oc - rc).abs().max() / rc.abs().max())
            bad_g = int(((og - rg).abs().max(dim=1).values > 1e-9 * rg.abs().max()).sum())
            bad_c = int(((oc - rc).abs().max(dim=1).values > 1e-9 * rc.abs().max()).sum())
            print(f"l={l}: gene-major rel diff {eg:.2e} ({bad_g} bad rows), cell-major rel diff {ec:.2e} ({bad_c} bad rows)", flush=True)
            if bad_c:
                rows = ((oc - rc).abs().max(dim=1).values > 1e-9 * rc.abs().max()).nonzero()[:, 0]
                print("   first bad cell rows", rows[:8].tolist(), "last", rows[-3:].tolist())
            if bad_g:
                rows = ((og - rg).abs().max(dim=1).values > 1e-9 * rg.abs().max()).nonzero()[:, 0]
                print("   first bad gene rows", rows[:8].tolist(), "last", rows[-3:].tolist())
        del xc, xg, og, oc
    del m
    torch.cuda.empty_cache()
