import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells, genes, l = 1_000_000, 33_000, 100
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
del ip, ix, vv
sa.normalize(m, sa.Normalization.CellRanger)
og = torch.zeros(genes, l, device=dev, dtype=torch.float64)
oc = torch.zeros(cells, l, device=dev, dtype=torch.float64)
def t(fn, reps=5):
    fn(); m.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    m.sync(); return (time.perf_counter() - t0) / reps * 1e3
for ld in (100, 500, 112, 128):
    xc = torch.randn(cells, ld, device=dev, dtype=torch.float64)
    xg = torch.randn(genes, ld, device=dev, dtype=torch.float64)
    a = t(lambda: m.dot_device(False, xc.data_ptr(), ld, l, og.data_ptr(), l))
    b = t(lambda: m.dot_device(True, xg.data_ptr(), ld, l, oc.data_ptr(), l))
    print(f"ld={ld}: gene-major {a:.2f} ms  cell-major {b:.2f} ms", flush=True)
    del xc, xg
