"""Worker for the world_size-2 tests (spawned by torch.distributed.run with the gloo backend).

mode "cpu":  the sharded normalize -> svd_bk schedule the library implements (cells range-partitioned,
             partial products all-reduced) restated with the CPU oracle per shard and the
             SAME collective hook bench.py hands to the library (scanrs_amd.dist.make_allreduce on host
             buffers); result must equal the single-process oracle.
mode "gpu":  the real C ABI path, two processes sharing one GPU, hook staged through the host.
mode "nccl": the real C ABI path with the collective served by RCCL on the device buffers themselves (backend
             "nccl", the hook exactly as bench.py installs it); one GPU per rank, so on a 1-GPU box world = 1.
Writes a JSON verdict per rank into the directory given on the command line."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _shard(m, lo, hi):
    ip = m.indptr[lo:hi + 1].astype(np.uint64)
    a, b = int(ip[0]), int(ip[-1])
    return (ip - ip[0]).astype(np.uint64), m.indices[a:b].astype(np.uint32), m.data[a:b].astype(np.uint32)


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist

    import scanrs_oracle as so
    from scanrs_amd.dist import make_allreduce, shard_bounds
    from scanrs_amd.synth import synth_counts

    if mode == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_cells, n_genes, k = 1200, 300, 6
    m = synth_counts(n_cells, n_genes, 0.08, 3)  # cells x genes CSR, identical on every rank
    lo, hi = shard_bounds(n_cells, world)[rank]
    ip, ix, vv = _shard(m, lo, hi)
    omega = so.omega_panel((2 * k, n_genes), 0)
    # single-process oracle on the whole matrix
    full = so.AdaptiveMat(n_genes, n_cells, so.CSC, m.indptr, m.indices, m.data)
    u_ref, s_ref, v_ref = so.BkSvd().run_pca(so.normalize(full, "cellranger"), k, omega=omega)
    verdict = {"rank": rank}

    if mode == "cpu":
        hook = make_allreduce(dist, None)

        def allreduce(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            assert hook(a.ctypes.data, a.size, 0) == 0
            return a

        # hook on a u64 buffer (histogram path of the distributed median)
        h = np.full(7, rank + 1, dtype=np.uint64)
        assert hook(h.ctypes.data, h.size, 1) == 0
        verdict["hook_u64_ok"] = bool(np.all(h == sum(range(1, world + 1))))
        loc = so.AdaptiveMat(n_genes, hi - lo, so.CSC, ip, ix, vv)
        counts = loc.sum_axis(0, np.uint32)
        gathered = [None] * world
        dist.all_gather_object(gathered, counts)
        target = max(float(so.median_mut(np.concatenate(gathered))), 1.0)
        lm = loc.compose_map(so.MapOp(so.OP_SCALE_AXIS, axis=1, a=target / counts.astype(np.float64))).apply(so.LOG_TWO)
        s1 = allreduce(lm.sum_axis(1))
        s2 = allreduce(lm.view().apply(so.OP_SQUARE).sum_axis(1))
        mean = s1 / n_cells
        d = s2 / n_cells - mean ** 2
        sc = np.where(d <= 0, 1.0, np.sqrt(np.where(d <= 0, 1.0, d)))
        a_loc = lm.scale(1, sc).center(1, mean / sc)  # genes x local cells, u = -mean/sc, v = ones(local)
        b = 2 * k
        P = omega.T.copy()  # genes x b
        K = np.zeros((n_genes, 5 * b))
        for i in range(5):
            T = a_loc.rdot(P.T).T  # local cells x b, no communication
            W = allreduce(a_loc.dot(T))  # partial genes x b (offset term included)
            P = np.linalg.qr(W)[0]
            K[:, i * b:(i + 1) * b] = P
        Q = np.linalg.qr(K)[0]
        T = a_loc.rdot(Q.T).T  # local cells x 5b
        G = allreduce(T.T @ T)
        w, Z = np.linalg.eigh(G)
        order = np.argsort(w)[::-1][:k]
        sig = np.sqrt(w[order])
        U = Q @ Z[:, order]
        V_loc = T @ (Z[:, order] / sig)
        verdict["s_rel"] = float(np.max(np.abs(sig - s_ref) / s_ref))
        sign = np.sign(np.sum(U * u_ref, axis=0))
        verdict["u_abs"] = float(np.max(np.abs(U * sign - u_ref)))
        verdict["v_abs"] = float(np.max(np.abs(V_loc * sign - v_ref[lo:hi])))
    else:
        import scanrs_amd as sa

        if mode == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            hook = make_allreduce(dist, dev)
            for code, dt in ((0, torch.float64), (1, torch.int64)):  # the hook on raw device pointers, both dtypes
                t = torch.full((1000,), rank + 1, device=dev, dtype=dt)
                assert hook(t.data_ptr(), t.numel(), code) == 0
                assert bool(torch.all(t == sum(range(1, world + 1))))
        else:
            torch.cuda.set_device(0)
            dev = torch.device("cuda", 0)
            hook = make_allreduce(dist, dev, stage_through_host=True)
        g = sa.AdaptiveMat.from_csmat(n_genes, hi - lo, sa.CSC, ip, ix, vv)
        g.set_shard(rank, world, lo, n_cells, hook)
        sa.normalize(g, sa.Normalization.CellRanger)
        u, s, v = sa.BkSvd().run_pca(g, k, omega=omega)
        verdict["target_umi"] = g.target_umi()
        verdict["s_rel"] = float(np.max(np.abs(s - s_ref) / s_ref))
        sign = np.sign(np.sum(u * u_ref, axis=0))
        verdict["u_abs"] = float(np.max(np.abs(u * sign - u_ref)))
        verdict["v_abs"] = float(np.max(np.abs(v * sign - v_ref[lo:hi])))
        verdict["v_rows"] = int(v.shape[0])
        # the randomized driver on the same shards (rand_svd.rs:54-129): Omega given explicitly, l = max(k + 4, 10 k)
        l = max(k + 4, 10 * k)
        om = so.omega_panel((l, n_genes), 1)
        ur, sr, vr = sa.RandSvd().run_pca(g, k, omega=om)
        uo, s_o, vo = so.RandSvd().run_pca(so.normalize(full, "cellranger"), k, omega=om)
        verdict["rand_s_rel"] = float(np.max(np.abs(sr - s_o) / s_o))
        sign = np.sign(np.sum(ur * uo, axis=0))
        verdict["rand_u_abs"] = float(np.max(np.abs(ur * sign - uo)))
        verdict["rand_v_abs"] = float(np.max(np.abs(vr * sign - vo[lo:hi])))
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump(verdict, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
