"""What the delivery of the factors costs: a step with U / V written into caller arrays against the same step with the result left
in device memory, for several settings of the "d2h_threads" option. usage: delivery_probe.py [cells] [threads,threads,...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
settings = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 8, 16]
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, 33_000, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(33_000, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
bk = sa.BkSvd()
r, c = m.shape()
out_u, out_v = np.zeros((r, 50)), np.zeros((c, 50))
def step(host):
    m.reset_map(); sa.normalize(m, sa.Normalization.CellRanger)
    return bk.run_pca(m, 50, out=(out_u, out_v)) if host else bk.run_pca_device(m, 50)
for _ in range(2): step(True)
def timed(host, n=4):
    best = 1e9
    for _ in range(n):
        m.sync(); t0 = time.perf_counter(); r_ = step(host); m.sync(); best = min(best, time.perf_counter() - t0); del r_
    return best * 1e3
print(f"device-resident: {timed(False):.2f} ms", flush=True)
for t in settings:
    m.set_option("d2h_threads", t)
    print(f"d2h_threads {t:3d}: delivered {timed(True):.2f} ms", flush=True)
print(f"device-resident: {timed(False):.2f} ms", flush=True)
