"""A/B of a handle option by whole steps on ONE box and ONE handle: reset + normalize + BkSvd with host delivery, the settings of the
option taken in turns (ABAB...), best and median of each. usage: step_ab.py cells option=v1,v2[,v3] [rounds]
e.g. step_ab.py 1000000 gemm_direct=1,0"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells = int(sys.argv[1])
key, vals = sys.argv[2].split("=")
vals = [float(v) for v in vals.split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, 33_000, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(33_000, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
bk = sa.BkSvd()
r, c = m.shape()
out_u, out_v = np.zeros((r, 50)), np.zeros((c, 50))
def step():
    m.reset_map(); sa.normalize(m, sa.Normalization.CellRanger)
    return bk.run_pca(m, 50, out=(out_u, out_v))
for v in vals:
    m.set_option(key, v)
    for _ in range(2): step()
times = {v: [] for v in vals}
sig = {}
for _ in range(rounds):
    for v in vals:
        m.set_option(key, v)
        step()  # the first step behind a switch may rebuild what the option governs
        m.sync(); t0 = time.perf_counter(); _, s, _ = step(); m.sync()
        times[v].append((time.perf_counter() - t0) * 1e3)
        sig[v] = s
for v in vals:
    t = sorted(times[v])
    print(f"{key}={v:g}: best {t[0]:.2f} ms, median {t[len(t)//2]:.2f} ms   sigma1 {sig[v][0]:.9f}", flush=True)
