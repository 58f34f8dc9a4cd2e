"""The first call on a fresh handle as Cell Ranger makes it (tools/src/bin/cmd.rs:61-70: load, normalize once, ONE run_pca),
piece by piece as the caller sees it: handle creation from device-resident arrays, normalize, run_pca with host delivery.
With SCANRS_TRACE=1 the library prints its own phases (each forces a synchronisation, so the sum is an upper bound).
usage: first_call.py [cells] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kw = {k: float(v) for k, v in (a.split("=") for a in sys.argv[3:] if "=" in a)}
opts = {k: v for k, v in kw.items() if k in ("side_build", "tile_builder", "tile_build_waves")}
kw = {k: v for k, v in kw.items() if k not in opts}
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
sa.init()
print(f"scanrs_init: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
ip, ix, vv = synth_counts_torch(cells, 33_000, 0.03, 0, dev, **kw)
torch.cuda.synchronize()
bk = sa.BkSvd()
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = sa.AdaptiveMat.from_device(33_000, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
    m.sync()
    for k_, v_ in opts.items():
        m.set_option(k_, v_)
    t1 = time.perf_counter()
    sa.normalize(m, sa.Normalization.CellRanger)
    m.sync()
    t2 = time.perf_counter()
    u, s, v = bk.run_pca(m, 50)
    t3 = time.perf_counter()
    print(f"first call #{r}: create {1e3*(t1-t0):7.1f} ms  normalize {1e3*(t2-t1):7.1f} ms  run_pca {1e3*(t3-t2):7.1f} ms  total {1e3*(t3-t0):7.1f} ms"
          f"   sigma1 {s[0]:.6f}", flush=True)
    print("   counters (ms): " + ", ".join(f"{k_[2:-3]} {m.counter(k_)/1e3:.1f}" for k_ in ("t_layout_us", "t_side_wait_us", "t_start_panel_us", "t_delivery_us", "t_alloc_us"))
          + f", hipMalloc calls {m.counter('alloc_calls')}, cached {sa.cached_memory_bytes()/1e9:.1f} GB", flush=True)
    # a second call on the warm handle for comparison
    t0 = time.perf_counter()
    m.reset_map()
    sa.normalize(m, sa.Normalization.CellRanger)
    u, s, v = bk.run_pca(m, 50)
    t1 = time.perf_counter()
    print(f"   warm call: {1e3*(t1-t0):7.1f} ms", flush=True)
    del m, u, v
    torch.cuda.empty_cache()
