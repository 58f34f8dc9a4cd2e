#!/usr/bin/env python3
"""bench.py — cells/sec for top-k PCA (CellRanger normalisation + BkSvd) on a synthetic sparse count matrix.

One "step" = one pass of the hot path over the whole matrix already resident in HBM:
reset the lazy map -> normalize(CellRanger) -> BkSvd{2.0, 5}.run_pca(k), through the C ABI
(include/scanrs_amd.h).  Default workload = BASELINE.json configs[2]: 1M cells x 33k genes @ 3 % nnz,
k = 50, on 1 GPU; with --gpus N the same global matrix is range-partitioned over the ranks by cells
(strong scaling; one process per GPU, torch.distributed/RCCL all-reduce supplied to the library as a
callback).  Rank 0 prints ONE JSON line.

Also reported:  roofline  — HBM roofline of the dominant kernel (the gather SpMM), algorithmic bytes per
launch / average launch duration measured with HIP events on the library's stream; cpu_baseline — the
CPU oracle (port of the reference's serial schedule) timed on a bounded sample on rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells", type=int, default=1_000_000)
    ap.add_argument("--genes", type=int, default=33_000)
    ap.add_argument("--density", type=float, default=0.03)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-cells", type=int, default=8000, help="cells of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="also time one step that copies U and V to the host")
    ap.add_argument("--f32-panels", action="store_true",
                    help="opt-in fast mode: gathered panels rounded to f32, f64 sums (NOT the headline configuration)")
    ap.add_argument("--spmm-path", type=int, default=0, help="0 auto, 1 plain gather, 2 L2-blocked gather")
    ap.add_argument("--also-randsvd", action="store_true", help="also time one RandSvd{10, 2} PCA (SURVEY.md §8d: reported alongside)")
    ap.add_argument("--also-irlba", action="store_true",
                    help="also time one Irlba{tol 1e-4, 50} run on the log-normalised (un-centred) matrix, the only input irlba.rs takes")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 only: serve the exchange steps through RCCL (world 1) anyway, to price the hook itself")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="record the per-launch HIP events inside the K timed steps instead of in a second pass of K steps")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # SCANRS_BENCH_SHARED_GPU=1 (tests only): every rank uses GPU 0 and the exchange steps go through gloo on host copies,
    # so the N > 1 code path can be exercised end to end on a 1-GPU box. Never set by the driver.
    shared_gpu = os.environ.get("SCANRS_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if shared_gpu else dev
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist_mod

        dist = dist_mod
        # the image exports NCCL_DEBUG=VERSION, and RCCL prints that banner with printf on stdout in every rank — stdout
        # carries the ONE JSON line, so the banner level (only that one) is switched off; INFO/TRACE etc. are left alone
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = ""
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import scanrs_amd as sa
    from scanrs_amd.synth import synth_counts_torch

    if not sa.device_available():
        raise SystemExit("bench.py needs a gfx950 device (no CPU fallback)")

    # ---- synthetic shard, generated in device memory -------------------------------------------------
    from scanrs_amd.dist import shard_bounds

    lo, hi = shard_bounds(args.cells, world)[rank]
    t0 = time.time()
    indptr, indices, values = synth_counts_torch(args.cells, args.genes, args.density, args.seed, dev, lo, hi)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    nnz_local = int(indptr[-1].item())
    n_local = hi - lo

    # genes x cells (Cell Ranger orientation), stored cell-major = CSC
    t0 = time.time()
    mat = sa.AdaptiveMat.from_device(args.genes, n_local, sa.CSC, indptr.data_ptr(), indices.data_ptr(), values.data_ptr())
    del indptr, indices, values
    torch.cuda.empty_cache()

    if dist is not None:
        from scanrs_amd.dist import make_allreduce

        mat.set_shard(rank, world, lo, args.cells, make_allreduce(dist, dev, stage_through_host=shared_gpu))

    if args.f32_panels:
        mat.set_panel_precision(1)
    if args.spmm_path:
        mat.set_spmm_path(args.spmm_path)
    bk = sa.BkSvd()  # k_multiplier 2.0, n_iter 5: the solver scan-rs-cmd uses (tools/src/bin/cmd.rs:70)
    s_out = np.zeros(args.k)

    def step(download=False):
        import ctypes

        mat.reset_map()
        sa.normalize(mat, sa.Normalization.CellRanger)
        if download:
            return bk.run_pca(mat, args.k)
        sa._check(
            sa._lib.scanrs_pca_bk(
                mat._h, ctypes.c_uint32(args.k), ctypes.c_double(bk.k_multiplier), ctypes.c_uint32(bk.n_iter),
                ctypes.c_uint64(0), None, None, None, s_out.ctypes.data_as(ctypes.c_void_p), None))
        return None, s_out.copy(), None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        mat.sync()

    # first pass also builds the transposed (gene-major) copy: data layout preparation, part of setup
    step()
    barrier()
    t_setup = time.time() - t0
    for _ in range(max(0, args.warmup - 1)):
        step()
    barrier()
    # Timed region: exactly K steps between barriers. The per-launch HIP events that feed the roofline leg put
    # a marker packet before and after each of the ~1700 launches of a step, which costs a few percent of wall
    # time, so by default they are recorded in a second pass of K steps right after the timed one
    # (same inputs, same launches); --events-in-timed-region records them inside the timed steps instead.
    mat.profile_reset()
    mat.profile_enable(bool(args.events_in_timed_region))
    barrier()
    t0 = time.perf_counter()
    sig = None
    for _ in range(args.steps):
        _, sig, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    events_elapsed = elapsed
    if not args.events_in_timed_region:
        mat.profile_enable(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        events_elapsed = time.perf_counter() - t0
    mat.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nn = torch.tensor([nnz_local], device=red_dev, dtype=torch.int64)
        dist.all_reduce(nn)
        nnz_global = int(nn.item())
    else:
        nnz_global = nnz_local
    prof = mat.profile_get()
    ms_per_step = elapsed / args.steps * 1e3
    value = args.cells * args.steps / elapsed

    pcie = None
    if args.pcie:
        barrier()
        t0 = time.perf_counter()
        step(download=True)
        barrier()
        pcie = args.cells / (time.perf_counter() - t0)

    randsvd_ms = None
    if args.also_randsvd and world == 1:
        import ctypes

        mat.reset_map()
        sa.normalize(mat, sa.Normalization.CellRanger)
        barrier()
        t0 = time.perf_counter()
        sa._check(sa._lib.scanrs_pca_rand(mat._h, ctypes.c_uint32(args.k), ctypes.c_double(10.0), ctypes.c_uint32(2), ctypes.c_uint64(0),
                                          None, None, s_out.ctypes.data_as(ctypes.c_void_p), None))
        barrier()
        randsvd_ms = (time.perf_counter() - t0) * 1e3

    irlba_ms = irlba_mprod = None
    if args.also_irlba and world == 1:
        import ctypes

        mat.reset_map()
        sa.log_normalize_with_size_factor(mat, None, sa.FN_LOG2_1P)
        mp = ctypes.c_uint32()
        u_out, v_out = np.zeros((args.genes, args.k)), np.zeros((n_local, args.k))  # irlba returns host arrays (irlba.rs:71-76)
        barrier()
        t0 = time.perf_counter()
        sa._check(sa._lib.scanrs_pca_irlba(mat._h, ctypes.c_uint32(args.k), ctypes.c_double(1e-4), ctypes.c_uint32(50), None, None,
                                           u_out.ctypes.data_as(ctypes.c_void_p), s_out.ctypes.data_as(ctypes.c_void_p),
                                           v_out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(mp)))
        barrier()
        irlba_ms, irlba_mprod = (time.perf_counter() - t0) * 1e3, int(mp.value)

    # ---- roofline of the dominant kernel -------------------------------------------------------------------
    roof = None
    if prof:
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        name, st = dom
        achieved = st["algorithmic_bytes"] / (st["total_ms"] * 1e-3) / 1e9 if st["total_ms"] > 0 else 0.0
        traffic = None
        try:  # HBM bytes per launch from the committed PMC passes (profiles/, separate --pmc runs), headline workload only
            with open(os.path.join(ROOT, "profiles", "r01h_pmc_traffic.json")) as f:
                pm = json.load(f)["kernels"].get(name)
            if pm and args.cells == 1_000_000 and args.genes == 33_000 and world == 1:
                traffic = round(pm["hbm_bytes_per_launch_corrected"])
        except (OSError, ValueError, KeyError):
            traffic = None
        roof = {
            "bound": "hbm",
            "kernel": name,
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "launches_per_step": st["launches"] / args.steps,
            "avg_launch_ms": round(st["total_ms"] / max(1, st["launches"]), 4),
            "algorithmic_bytes_per_launch": round(st["algorithmic_bytes"] / max(1, st["launches"])),
            # the product is bound by the on-chip gather of panel rows (L2 -> CU), not by HBM: DESIGN.md §4.
            # ceiling: 16.8-18.8 TB/s chip-wide for L2-resident indexed rows (MI355X_MICROARCH.md, "Indexed rows")
            "onchip_gather": {
                "achieved_TBps": round(st["onchip_gather_bytes"] / (st["total_ms"] * 1e-3) / 1e12, 2) if st["total_ms"] > 0 else None,
                "ceiling_TBps": 17.8,
                "bytes_per_launch": round(st["onchip_gather_bytes"] / max(1, st["launches"])),
            },
            "kernel_ms_per_step": {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())},
            "events_pass_ms_per_step": round(events_elapsed / args.steps * 1e3, 2),
        }

    # ---- CPU baseline: the oracle (single thread, the reference's serial schedule) on a bounded sample --------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import scanrs_oracle as so
        from threadpoolctl import threadpool_limits

        nc = min(args.cpu_cells, args.cells)
        ip, ix, vv = synth_counts_torch(args.cells, args.genes, args.density, args.seed, dev, 0, nc)
        ip, ix, vv = ip.cpu().numpy().astype(np.uint64), ix.cpu().numpy().astype(np.uint32), vv.cpu().numpy().astype(np.uint32)
        so.build()
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            om = so.AdaptiveMat(args.genes, nc, so.CSC, ip, ix, vv)
            a = so.normalize(om, "cellranger")
            so.BkSvd().run_pca(a, min(args.k, nc))
            t_cpu = time.perf_counter() - t0
        cpu = {
            "value": round(nc / t_cpu, 2),
            "unit": "cells/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {nc} cells of the same synthetic matrix ({args.genes} genes, {args.density:.0%} nnz), "
                      f"normalize(CellRanger) + BkSvd k={min(args.k, nc)}, oracle C loops + LAPACK pinned to 1 thread, {t_cpu:.1f} s",
        }

    if rank == 0:
        out = {
            "metric": "cells/sec for top-50 PCA on 1M x 33k @3% nnz; achieved HBM GB/s vs roofline",
            "value": round(value, 1),
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if not args.f32_panels else "f64 sums over f32-rounded gather panels (opt-in fast mode)",
            "data": "synthetic",
            "config": {
                "workload": f"{args.cells} cells x {args.genes} genes, {args.density:.0%} nnz synthetic sqz CSC, "
                            f"normalize(CellRanger) + BkSvd(k_multiplier=2, n_iter=5) top-{args.k} PCA",
                "nnz": nnz_global,
                "parallelism": f"cells range-partitioned over {world} GPU(s)" + (", all-reduce per product" if world > 1 else ""),
                "setup_s": round(t_setup, 2),
                "datagen_s": round(t_gen, 2),
                "sigma_top3": [round(float(x), 6) for x in (sig[:3] if sig is not None else [])],
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if pcie is not None:
            out["config"]["pcie_inclusive_cells_per_s"] = round(pcie, 1)
        if randsvd_ms is not None:
            out["config"]["randsvd_l10_it2_ms"] = round(randsvd_ms, 1)
        if irlba_ms is not None:
            out["config"]["irlba_tol1e-4_ms"] = round(irlba_ms, 1)
            out["config"]["irlba_matrix_products"] = irlba_mprod
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
