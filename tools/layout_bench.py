"""Tile-layout builders side by side: per-thread walk (tile_builder=0) against the wave-level builder (tile_builder=1) on the same
matrix, both orientations — products must be bit-identical (same layout, same order of additions); run with SCANRS_TRACE=1 for
the phases of each build. usage: layout_bench.py [cells] [waves per CU ...] [gene_shape=.. shared_profile=..]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scanrs_amd as sa
from scanrs_amd.synth import synth_counts_torch

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kw = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:] if "=" in a)}
waves = [int(a) for a in sys.argv[2:] if "=" not in a] or [16]
genes, l = 33_000, 100
dev = torch.device("cuda", 0)
ip, ix, vv = synth_counts_torch(cells, genes, 0.03, 0, dev, **kw)
xg = torch.randn(genes, l, device=dev, dtype=torch.float64)
xc = torch.randn(cells, l, device=dev, dtype=torch.float64)
ref = None
for builder, w in [(0, 0)] + [(1, w) for w in waves]:
    og = torch.zeros(genes, l, device=dev, dtype=torch.float64)
    oc = torch.zeros(cells, l, device=dev, dtype=torch.float64)
    m = sa.AdaptiveMat.from_device(genes, cells, sa.CSC, ip.data_ptr(), ix.data_ptr(), vv.data_ptr())
    m.set_spmm_path(3)
    m.set_option("tile_builder", builder)
    m.set_option("tile_build_waves", w)
    sa.normalize(m, sa.Normalization.CellRanger)
    m.sync()
    t0 = time.perf_counter()
    m.dot_device(True, xg.data_ptr(), l, l, oc.data_ptr(), l)  # cell-major layout
    m.sync()
    t1 = time.perf_counter()
    m.dot_device(False, xc.data_ptr(), l, l, og.data_ptr(), l)  # transposed copy + gene-major layout
    m.sync()
    t2 = time.perf_counter()
    msg = ""
    if ref is None:
        ref = (og.clone(), oc.clone())
    else:
        msg = f"  bit-identical to the per-thread builder: gene-major {bool(torch.equal(og, ref[0]))}, cell-major {bool(torch.equal(oc, ref[1]))}"
    print(f"builder {builder} waves/CU {w:2d}: first cell-major product {1e3*(t1-t0):7.1f} ms, first gene-major product (incl. transposed copy) {1e3*(t2-t1):7.1f} ms{msg}", flush=True)
    del m
    torch.cuda.empty_cache()
    time.sleep(3)  # freed VRAM is scrubbed in the background; an allocation right behind a free waits for it (profiles/microbench/alloc_probe2)
