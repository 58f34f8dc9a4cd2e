"""Host-side stage markers of one steady-state step (SCANRS_TRACE=2 prints one line per marker with a steady-clock time stamp and
synchronises nothing): where the calling thread is at which time. usage: stage_times.py [cells] [steps] [option=value ...]"""
import os, subprocess, sys, time
if os.environ.get("STAGE_CHILD") != "1":
    env = dict(os.environ, STAGE_CHILD="1", SCANRS_TRACE="2")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
    lines = [l for l in p.stderr.splitlines() if l.startswith("[scanrs stage]") or l.startswith("[step]")]
    last = max(i for i, l in enumerate(lines) if l.startswith("[step] begin"))
    t0 = None
    prev = None
    for l in lines[last:]:
        f = l.split()
        if l.startswith("[step]"):
            t, what = float(f[2]), " ".join(f[1:2]) + " (python)"
        else:
            t, what = float(f[2]), " ".join(f[3:])
        if t0 is None:
            t0 = prev = t
        print(f"{t - t0:9.3f} ms  (+{t - prev:7.3f})  {what}")
        prev = t
    sys.exit(p.returncode)
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, 33_000, 0.03, 0, dev)
m = sa.AdaptiveMat.from_device(33_000, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
for a in sys.argv[3:]:  # handle options: name=value
    m.set_option(a.split('=')[0], float(a.split('=')[1]))
bk = sa.BkSvd()
r, c = m.shape()
out_u, out_v = np.zeros((r, 50)), np.zeros((c, 50))
def now(): return time.clock_gettime(time.CLOCK_MONOTONIC) * 1e3
for i in range(steps):
    m.sync()
    print(f"[step] begin {now():.3f}", file=sys.stderr, flush=True)
    m.reset_map(); sa.normalize(m, sa.Normalization.CellRanger)
    print(f"[step] normalized {now():.3f}", file=sys.stderr, flush=True)
    bk.run_pca(m, 50, out=(out_u, out_v))
    print(f"[step] end {now():.3f}", file=sys.stderr, flush=True)
