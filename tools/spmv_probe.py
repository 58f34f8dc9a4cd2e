"""Time of the Ix1 products (IRLBA's A v and A^T w) under the log-normalisation map, with handle options toggled:
usage: spmv_probe.py [cells] [option=v1,v2]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
key, vals = (sys.argv[2].split("=")[0], [float(x) for x in sys.argv[2].split("=")[1].split(",")]) if len(sys.argv) > 2 else (None, [None])
genes = 33_000
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev)
xg = torch.randn(genes, 2, device=dev, dtype=torch.float64)
xc = torch.randn(cells, 2, device=dev, dtype=torch.float64)
og = torch.zeros(genes, 2, device=dev, dtype=torch.float64)
oc = torch.zeros(cells, 2, device=dev, dtype=torch.float64)
ref = None
for v in vals:
    m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
    if key: m.set_option(key, v)
    sa.log_normalize_with_size_factor(m, None, sa.FN_LOG2_1P)
    def t(fn, reps=10):
        for _ in range(3): fn()
        m.sync(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        m.sync(); return (time.perf_counter() - t0) / reps * 1e3
    a = t(lambda: m.dot_device(False, xc.data_ptr(), 2, 1, og.data_ptr(), 2))
    b = t(lambda: m.dot_device(True, xg.data_ptr(), 2, 1, oc.data_ptr(), 2))
    res = (og[:, 0].clone(), oc[:, 0].clone())
    err = ""
    if ref is None: ref = res
    else: err = "  max |diff| vs first: %.3e / %.3e" % (float((res[0] - ref[0]).abs().max()), float((res[1] - ref[1]).abs().max()))
    print(f"{key}={v}: A x (gene-major, short-outer) {a:.3f} ms   A^T y (cell-major, long-outer) {b:.3f} ms{err}", flush=True)
    del m
