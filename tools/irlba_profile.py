"""Per-kernel time of one Irlba run on the synthetic 1M x 33k matrix (HIP events on the library's stream)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
genes, k = 33_000, 50
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
del ip, ix, vv
sa.log_normalize_with_size_factor(m, None, sa.FN_LOG2_1P)
ir = sa.Irlba()
ir.run_pca(m, k)  # builds the transposed copy
m.profile_reset()
m.profile_enable(True)
t0 = time.perf_counter()
u, s, v = ir.run_pca(m, k)
dt = time.perf_counter() - t0
m.profile_enable(False)
prof = m.profile_get()
print(f"irlba: {dt*1e3:.1f} ms, {ir.mprod} products, sigma[:3] = {s[:3]}")
for name, st in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"  {name:44s} launches {st['launches']:6d}  total {st['total_ms']:9.2f} ms  avg {st['total_ms']/max(1,st['launches']):8.4f} ms")
