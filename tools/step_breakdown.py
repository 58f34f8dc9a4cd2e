"""Wall time of the pieces of one bench step as Python sees them (reset + normalize, run_pca host-delivered or
device-resident, dropping the result arrays). usage: step_breakdown.py [cells]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, 33_000, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(33_000, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
bk = sa.BkSvd()
def norm():
    m.reset_map(); sa.normalize(m, sa.Normalization.CellRanger)
for _ in range(2):
    norm(); bk.run_pca(m, 50)
for mode in ("host", "device", "host", "device"):
    tn = tp = tf = 0.0
    for _ in range(5):
        t0 = time.perf_counter(); norm(); m.sync(); t1 = time.perf_counter()
        r = bk.run_pca(m, 50) if mode == "host" else bk.run_pca_device(m, 50)
        t2 = time.perf_counter(); del r; t3 = time.perf_counter()
        tn += t1 - t0; tp += t2 - t1; tf += t3 - t2
    print(f"{mode:6s} normalize {tn/5*1e3:7.2f} ms  run_pca {tp/5*1e3:7.2f} ms  drop results {tf/5*1e3:6.2f} ms")
