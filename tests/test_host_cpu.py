"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the
host-side dense helpers are correct, compute entry points fail loudly without a device, and the
world_size-2 sharded schedule (gloo) reproduces the single-process result."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sa():
    import scanrs_amd

    return scanrs_amd


def test_library_exports_every_header_symbol(sa):
    hdr = open(os.path.join(ROOT, "include", "scanrs_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(scanrs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"scanrs_progress_fn", "scanrs_allreduce_fn"}
    assert len(declared) >= 40
    lib = ctypes.CDLL(sa.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in scanrs_amd.h but not exported"
    assert declared == set(sa.EXPORTED_SYMBOLS)


def test_headers_are_plain_c99_and_cxx17(tmp_path):
    """The boundary is a C ABI: scanrs_amd.h must go through a C compiler untouched, the mirror through a C++17 one."""
    inc = os.path.join(ROOT, "include")
    c_src = tmp_path / "t.c"
    c_src.write_text('#include "scanrs_amd.h"\nint main(void) { scanrs_adaptive_vec v; v.kind = 0; return (int)v.kind + SCANRS_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(c_src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cc_src = tmp_path / "t.cpp"
    cc_src.write_text('#include "scanrs_amd.hpp"\nint main() { scanrs::BkSvd b; return b.n_iter == 5 ? 0 : 1; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(cc_src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_cpu_fallback_without_device(sa):
    if sa.device_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sa.ScanrsError) as e:
        sa.AdaptiveMat.from_dense(np.eye(3, dtype=np.uint32))
    assert e.value.code == 4 and "no CPU fallback" in str(e.value)


def test_single_process_multi_gpu_form_needs_a_device_too(sa):
    if sa.device_available():
        pytest.skip("checks the behaviour of a host without a GPU")
    ip = np.array([0, 1, 2], dtype=np.uint64)
    with pytest.raises(sa.ScanrsError) as e:
        sa.MultiMat(2, 2, sa.CSR, ip, np.array([0, 1], dtype=np.uint32), np.array([1, 1], dtype=np.uint32), 2, devices=[0, 0])
    assert e.value.code == 4
    with pytest.raises(sa.ScanrsError):  # and bad arguments are refused before anything else
        sa.MultiMat(2, 2, sa.CSR, ip, np.array([0, 1], dtype=np.uint32), np.array([1, 1], dtype=np.uint32), 17)


def test_product_path_never_imports_the_oracle():
    names = ("scanrs_oracle", "liboracle", "import adaptive_vec", "knn_oracle", "oracle/")  # everything under oracle/
    for top in ("scan-rs_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not any(n in txt for n in names), f
    # bench.py may touch the oracle in its cpu_baseline legs only: every import sits under an `if ... no_cpu_baseline` guard
    bench = open(os.path.join(ROOT, "bench.py")).read()
    pos, n_imports = 0, 0
    while True:
        pos = bench.find("import scanrs_oracle", pos)
        if pos < 0:
            break
        head = bench[:pos]
        guard = max(head.rfind("\n    if "), head.rfind("\n        if "))
        assert guard >= 0 and "no_cpu_baseline" in head[guard:head.index("\n", guard + 1)], head[guard:guard + 120]
        pos += 1
        n_imports += 1
    assert n_imports >= 1


def test_host_cholesky_and_inverse(sa):
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 100):
        a = rng.standard_normal((n + 5, n))
        g = a.T @ a + 0.1 * np.eye(n)
        r = sa.host_chol_upper(g)
        assert np.allclose(np.triu(r), r)
        assert np.allclose(r.T @ r, g, rtol=1e-12, atol=1e-12)
        ri = sa.host_inv_upper(r)
        assert np.allclose(ri @ r, np.eye(n), atol=1e-10)
    with pytest.raises(sa.ScanrsError):
        sa.host_chol_upper(np.array([[1.0, 2.0], [2.0, 1.0]]))  # indefinite


def test_host_sym_eig(sa):
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 10, 64, 150):
        a = rng.standard_normal((n, n))
        s = a + a.T
        w, z = sa.host_sym_eig(s)
        assert np.all(np.diff(w) <= 1e-12)
        assert np.allclose(w, np.linalg.eigvalsh(s)[::-1], rtol=1e-11, atol=1e-11)
        assert np.allclose(z @ np.diag(w) @ z.T, s, atol=1e-10)
        assert np.allclose(z.T @ z, np.eye(n), atol=1e-12)
    # clustered / repeated eigenvalues and a Gram matrix with a large dynamic range
    d = np.array([5.0, 5.0, 5.0, 1.0, 1e-8, 0.0])
    q = np.linalg.qr(rng.standard_normal((6, 6)))[0]
    w, z = sa.host_sym_eig(q @ np.diag(d) @ q.T)
    assert np.allclose(w, d, atol=1e-13)


def test_plan_shards_balances_nnz(sa):
    rng = np.random.default_rng(2)
    lens = rng.integers(0, 50, size=1000)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        b = sa.plan_shards(indptr, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b.astype(np.int64)) >= 0)
        per = np.diff(indptr[b.astype(np.int64)].astype(np.int64))
        assert per.sum() == indptr[-1]
        assert per.max() - per.min() <= 2 * 50
    assert sa.plan_shards(np.zeros(1, dtype=np.uint64), 2).tolist() == [0, 0, 0]


def test_omega_fill_matches_oracle_stream(sa):
    import scanrs_oracle as so

    assert np.array_equal(sa.omega_fill(0, 1000), so.omega_panel((10, 100), 0).ravel())
    assert np.array_equal(sa.omega_fill(7, 33), so.omega_panel((33, 1), 7).ravel())


def test_normalization_from_str(sa):
    assert sa.Normalization.from_str("cellranger") == sa.Normalization.CellRanger
    assert sa.Normalization.from_str("binomialpearson") == sa.Normalization.BinomialPearson
    with pytest.raises(ValueError):
        sa.Normalization.from_str("nope")


def _run_world2(mode, tmp_path, world=2):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() % 400)), os.path.join(ROOT, "tests", "dist_worker.py"), mode, str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.load(open(os.path.join(tmp_path, f"rank{i}.json"))) for i in range(world)]


def test_world2_sharded_schedule_gloo(tmp_path):
    for v in _run_world2("cpu", tmp_path):
        assert v["hook_u64_ok"]
        # same algorithm, same panel, different reduction order: far inside the 1e-4 north-star tolerance
        assert v["s_rel"] < 1e-9 and v["u_abs"] < 1e-7 and v["v_abs"] < 1e-7, v


@pytest.mark.gpu
def test_world2_sharded_c_abi_on_one_gpu(tmp_path):
    vs = _run_world2("gpu", tmp_path)
    assert vs[0]["target_umi"] == vs[1]["target_umi"]
    for v in vs:
        assert v["v_rows"] == 600
        assert v["s_rel"] < 1e-8 and v["u_abs"] < 1e-6 and v["v_abs"] < 1e-6, v
        assert v["rand_s_rel"] < 1e-8 and v["rand_u_abs"] < 1e-6 and v["rand_v_abs"] < 1e-6, v


@pytest.mark.gpu
def test_world4_sharded_c_abi_on_one_gpu(tmp_path):
    """Four ranks (four processes sharing the box's one GPU, exchange through the host hook): 300 cells per shard."""
    vs = _run_world2("gpu", tmp_path, world=4)
    assert len({v["target_umi"] for v in vs}) == 1
    for v in vs:
        assert v["v_rows"] == 300
        assert v["s_rel"] < 1e-8 and v["u_abs"] < 1e-6 and v["v_abs"] < 1e-6, v
        assert v["rand_s_rel"] < 1e-8 and v["rand_u_abs"] < 1e-6 and v["rand_v_abs"] < 1e-6, v


def test_world4_sharded_schedule_gloo(tmp_path):
    for v in _run_world2("cpu", tmp_path, world=4):
        assert v["hook_u64_ok"]
        assert v["s_rel"] < 1e-9 and v["u_abs"] < 1e-7 and v["v_abs"] < 1e-7, v


@pytest.mark.gpu
def test_rccl_hook_serves_the_exchange_steps(tmp_path):
    """backend "nccl" (RCCL) on the library's own device buffers, one rank per visible GPU (1 on the test box)."""
    import torch

    world = max(1, min(2, torch.cuda.device_count()))
    for v in _run_world2("nccl", tmp_path, world=world):
        assert v["v_rows"] == 1200 // world
        assert v["s_rel"] < 1e-8 and v["u_abs"] < 1e-6 and v["v_abs"] < 1e-6, v
        assert v["rand_s_rel"] < 1e-8 and v["rand_u_abs"] < 1e-6 and v["rand_v_abs"] < 1e-6, v


def _bench_line(extra_env, launcher, tmp_path, gpus, shape=("--cells", "6000", "--genes", "1500", "--density", "0.04", "--k", "8"), timeout=900):
    env = dict(os.environ)
    env.update(extra_env)
    args = ["--gpus", str(gpus), "--steps", "1", "--warmup", "1", *shape, "--no-cpu-baseline"]
    cmd = [sys.executable] + launcher + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_equal_one_rank(tmp_path):
    """bench.py's N > 1 flow (nnz-balanced shard bounds, per-rank synthetic shard, max-over-ranks timing) end to end through
    the PLAIN command form `python bench.py --gpus 2` (bench.py starts its own torchrun child), two ranks sharing the one
    GPU of the test box (host-hook exchange over gloo): same global matrix, same singular values."""
    one = _bench_line({}, [], tmp_path, 1)
    two = _bench_line({"SCANRS_BENCH_SHARED_GPU": "1"}, [], tmp_path, 2)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["nnz"] == one["config"]["nnz"]
    assert len(two["config"]["per_rank_ms_per_step"]) == 2 and sum(two["config"]["per_rank_nnz"]) == two["config"]["nnz"]
    a, b = np.array(one["config"]["sigma_top3"]), np.array(two["config"]["sigma_top3"])
    assert np.max(np.abs(a - b) / a) < 1e-9
    for d in (one, two):
        for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config", "roofline",
                    "cpu_baseline", "vs_baseline", "steps", "warmup"):
            assert key in d
        assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
        assert "6000x1500" in d["metric"] and "top-8" in d["metric"]  # the label follows the arguments
        assert d["config"]["device_resident_cells_per_s"] > 0 and d["config"]["v_col_norm_err_device_result"] < 1e-9
        assert d["config"]["first_call_s"] > 0 and "transport" in d["config"] and "allreduce_ms_per_step_rank0" in d["config"]


@pytest.mark.gpu
def test_bench_all_ranks_take_the_same_transport_when_one_rank_cannot_load_rccl(tmp_path):
    """Round 3's verdict, item 7: the library's RCCL communicator has never run at world >= 2, and if it cannot be made on ONE rank
    (its `dlopen(RTLD_NOLOAD)` reuse of torch's librccl misbehaving) every rank has to fall to the same other transport or the
    first exchange step never returns. bench.py decides in a pre-flight that involves no collective of the library; here rank 1's
    pre-flight fails by injection, rank 0's passes: both must report the fallback transport, name rank 1 as the reason, and the
    run must finish with the single-rank singular values."""
    one = _bench_line({}, [], tmp_path, 1)
    two = _bench_line({"SCANRS_BENCH_SHARED_GPU": "1", "SCANRS_BENCH_FAIL_COMM_RANK": "1"}, [], tmp_path, 2)
    t = two["config"]["transport"]
    assert "library communicator not used" in t and "rank 1:" in t and "injected" in t and "rank 0:" not in t, t
    a, b = np.array(one["config"]["sigma_top3"]), np.array(two["config"]["sigma_top3"])
    assert np.max(np.abs(a - b) / a) < 1e-9
    # without the injection the same two ranks still may not build the communicator (they share the test box's one GPU): again one decision for all
    two = _bench_line({"SCANRS_BENCH_SHARED_GPU": "1"}, [], tmp_path, 2)
    assert "ranks share one GPU" in two["config"]["transport"]


@pytest.mark.gpu
def test_bench_eight_ranks_equal_one_rank(tmp_path):
    """The driver's widest command form, `python bench.py --gpus 8`, end to end on the one GPU of the test box (eight
    ranks sharing it, host-hook exchange): eight nnz-balanced shards, one JSON line, the same singular values."""
    one = _bench_line({}, [], tmp_path, 1)
    eight = _bench_line({"SCANRS_BENCH_SHARED_GPU": "1"}, [], tmp_path, 8)
    assert eight["n_gpus"] == 8 and len(eight["config"]["per_rank_nnz"]) == 8
    assert sum(eight["config"]["per_rank_nnz"]) == one["config"]["nnz"]
    assert max(eight["config"]["per_rank_nnz"]) - min(eight["config"]["per_rank_nnz"]) < 0.02 * one["config"]["nnz"] / 8  # balanced by nonzeros
    a, b = np.array(one["config"]["sigma_top3"]), np.array(eight["config"]["sigma_top3"])
    assert np.max(np.abs(a - b) / a) < 1e-9


@pytest.mark.gpu
def test_bench_four_ranks_at_200k_cells_within_a_hard_timeout(tmp_path):
    """Round 2's 4-rank shared-GPU runs at 200 k+ cells were killed from outside after minutes without a line. Cause: four
    processes generating their synthetic shards at the same time on ONE device ran ~170x slower than in turn (172 s per
    50 k-cell shard instead of ~1 s) — a property of sharing the test box's GPU, not of the exchange flow; in that test mode
    the ranks now take turns, and a watchdog ends any rank that makes no progress with its stage trail. The same command
    (hybrid product included: 52 M nonzeros per rank) must finish well inside the timeout."""
    d = _bench_line({"SCANRS_BENCH_SHARED_GPU": "1", "SCANRS_BENCH_WATCHDOG_S": "150"}, [], tmp_path, 4,
                    shape=("--cells", "200000", "--k", "50"), timeout=420)
    assert d["n_gpus"] == 4 and len(d["config"]["per_rank_nnz"]) == 4
    assert "transport" in d["config"] and d["config"]["allreduce_ms_per_step_rank0"] > 0
    assert d["config"]["v_col_norm_err_device_result"] < 1e-9


def test_bench_watchdog_names_the_stage(tmp_path):
    """The watchdog itself (no GPU needed): a stage that makes no progress ends the process with code 3 and the trail."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(0, 1.0); w.stage('stuck here'); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "WATCHDOG" in r.stderr and "stuck here" in r.stderr


def test_host_sym_eig_topk(sa):
    rng = np.random.default_rng(5)
    for n, k in ((1, 1), (2, 2), (7, 3), (64, 64), (200, 20)):
        a = rng.standard_normal((n + 2, n))
        g = a.T @ a
        w, z = sa.host_sym_eig_topk(g, k)
        wr = np.linalg.eigvalsh(g)[::-1][:k]
        assert np.allclose(w, wr, rtol=1e-12, atol=1e-12 * wr[0])
        assert np.allclose(z.T @ z, np.eye(k), atol=1e-12)
        assert np.max(np.abs(g @ z - z * w)) <= 1e-12 * wr[0]
    # spectrum shaped like the projected Gram matrix of a PCA: a few large, a flat bulk, a numerically null tail
    n = 300
    q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    lam = np.concatenate([np.linspace(1e8, 5e7, 19), 1e6 * (1 + 1e-3 * rng.random(200)), [3.0, 3.0, 3.0], 1e-9 * rng.random(78)])
    g = q @ np.diag(lam) @ q.T
    g = (g + g.T) / 2
    w, z = sa.host_sym_eig_topk(g, 60)
    ws = np.sort(lam)[::-1][:60]
    assert np.max(np.abs(w - ws) / ws) < 1e-12
    assert np.max(np.abs(z.T @ z - np.eye(60))) < 1e-12
    assert np.max(np.abs(g @ z - z * w)) < 1e-12 * ws[0]


def test_host_sym_eig_topk_vectors_stay_orthonormal_across_the_reorthogonalisation_window(sa):
    """Inverse iteration re-orthogonalises a vector only against the accepted ones within 3e-2 |T| of its eigenvalue (host_linalg.cpp):
    spectra whose gaps sit just above and just below that window, near-double eigenvalues, a decaying and a nearly flat spectrum must
    all come back orthonormal to rounding with small residuals."""
    rng = np.random.default_rng(3)
    n, k = 400, 60
    spectra = [
        1e6 * (1 - 0.31 * np.arange(n) / n),
        1e6 * np.cumprod(np.full(n, 1 / 1.032)),
        np.concatenate([1e6 * (1 - 0.0305 * np.arange(30)), 1e3 * rng.random(n - 30)]),
        np.concatenate([np.repeat(1e6 * (1 - 0.05 * np.arange(15)), 2) * (1 + 1e-9 * rng.random(30)), rng.random(n - 30)]),
        1e4 * rng.random(n) ** 6,
        np.linspace(1.0, 0.97, n) * 1e5,
    ]
    for lam in spectra:
        lam = np.sort(np.abs(lam))[::-1]
        q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        g = (q * lam) @ q.T
        g = (g + g.T) / 2
        w, z = sa.host_sym_eig_topk(g, k)
        assert np.max(np.abs(w - lam[:k])) < 1e-13 * lam[0]
        assert np.max(np.abs(z.T @ z - np.eye(k))) < 1e-13
        assert np.max(np.abs(g @ z - z * w)) < 1e-13 * lam[0]


def test_host_sym_eig_topk_bisection_edge_cases(sa):
    """k <= n / 4 takes the k largest eigenvalues by bisection on Sturm counts (host_linalg.cpp): exactly repeated values,
    indefinite matrices, a diagonal and a zero matrix, the smallest sizes the branch sees."""
    rng = np.random.default_rng(8)
    for n, k in ((4, 1), (8, 2), (40, 10), (101, 25)):
        q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        for lam in (np.concatenate([[5.0] * (k + 1), rng.random(n - k - 1)]),          # the top k + 1 values equal
                    np.concatenate([np.linspace(3.0, 1.0, k), -np.linspace(0.5, 9.0, n - k)]),  # indefinite, |negative| larger
                    np.sort(rng.standard_normal(n))[::-1] * 1e-150,                     # tiny scale
                    np.sort(rng.standard_normal(n))[::-1] * 1e+150):                    # huge scale
            g = (q * lam) @ q.T
            g = (g + g.T) / 2
            w, z = sa.host_sym_eig_topk(g, k)
            ws = np.sort(lam)[::-1][:k]
            scale = np.max(np.abs(lam))
            assert np.max(np.abs(w - ws)) <= 1e-12 * scale, (n, k)
            assert np.max(np.abs(z.T @ z - np.eye(k))) < 1e-10
            assert np.max(np.abs(g @ z - z * w)) <= 1e-11 * scale
        d = np.diag(np.arange(n, 0, -1, dtype=np.float64))
        w, z = sa.host_sym_eig_topk(d, k)
        assert np.array_equal(w, np.arange(n, n - k, -1, dtype=np.float64))
        w, z = sa.host_sym_eig_topk(np.zeros((n, n)), k)
        assert np.all(w == 0.0) and np.max(np.abs(z.T @ z - np.eye(k))) < 1e-12


def test_sym_eig_topk_does_not_depend_on_the_thread_count():
    """The tridiagonalisation sums its partial products in four fixed slots whether one thread or a team walks them
    (host_linalg.cpp): replicated ranks, hosts with fewer cores and a dismissed team all get bit-identical factors."""
    import subprocess
    import sys

    code = (
        "import numpy as np, scanrs_amd as sa, sys\n"
        "sa.set_global_option('eig_threads', int(sys.argv[1]))\n"
        "rng = np.random.default_rng(4)\n"
        "b = rng.standard_normal((800, 800)); a = b @ b.T\n"
        "w, z = sa.host_sym_eig_topk(a, 40)\n"
        "sys.stdout.write(w.tobytes().hex() + z.tobytes().hex())\n"
    )
    outs = []
    for t in ("1", "2", "4"):
        env = dict(os.environ, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", code, t], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 1000


def test_bounded_wait_turns_a_never_signalled_event_into_an_error(sa, capfd):
    """Round 3's verdict: a call that can block for ever is not a drop-in for run_pca. Every host-side wait of the library is a
    poll with a deadline ("sync_timeout_s"); here the library's own wait loop runs on an injected event that is never signalled
    (no device needed) and must come back with SCANRS_ERR_DEVICE within the deadline, naming the wait and the last stage."""
    import time

    t0 = time.perf_counter()
    with pytest.raises(sa.ScanrsError) as ei:
        sa.debug_wait_never(0.3)
    dt = time.perf_counter() - t0
    assert 0.3 <= dt < 1.0, dt
    assert ei.value.code == 4  # SCANRS_ERR_DEVICE
    msg = str(ei.value)
    assert "timed out" in msg and "sync_timeout_s" in msg and "scanrs_debug_wait_never" in msg and "debug: injected wait" in msg, msg
    err = capfd.readouterr().err
    assert "device wait timed out" in err and "stage debug: injected wait" in err
    # the option is validated
    with pytest.raises(sa.ScanrsError):
        sa.set_global_option("sync_timeout_s", -1.0)
    sa.set_global_option("sync_timeout_s", 120.0)


def test_group_barrier_of_the_single_process_form_is_bounded(sa):
    """The barrier between the shard threads of scanrs_multi_* has the same deadline: a shard that never arrives fails the
    others instead of holding them."""
    import ctypes
    import time

    if not hasattr(sa._lib, "scanrs_debug_barrier_alone"):
        pytest.skip("hook not built")
    sa.set_global_option("sync_timeout_s", 0.3)
    try:
        t0 = time.perf_counter()
        rc = sa._lib.scanrs_debug_barrier_alone(ctypes.c_uint32(2))
        dt = time.perf_counter() - t0
        assert rc == 4 and 0.3 <= dt < 1.5, (rc, dt)
        assert b"group barrier timed out" in sa._lib.scanrs_last_error()
    finally:
        sa.set_global_option("sync_timeout_s", 120.0)


def test_every_option_and_counter_is_documented_in_the_header():
    """The option / counter keys the library accepts (capi.cpp) against the ones include/scanrs_amd.h documents: a knob nobody can
    find is as good as an environment variable."""
    import re

    src = open(os.path.join(ROOT, "scan-rs_amd", "csrc", "capi.cpp")).read()
    hdr = open(os.path.join(ROOT, "include", "scanrs_amd.h")).read()
    keys = set(re.findall(r'\bk == "([a-z0-9_]+)"', src))
    assert len(keys) > 30
    missing = sorted(k for k in keys if f'"{k}"' not in hdr)
    assert not missing, f"keys accepted by capi.cpp but not documented in include/scanrs_amd.h: {missing}"


def test_replicated_sharded_split_of_a_step():
    """bench.py reports what part of a step shrinks with the cell range (config.sharded_ms_per_step) and what does not
    (config.replicated_ms_per_step) from the step on the whole matrix and on a second one with about half the nonzeros: the model
    t(f) = replicated + sharded f must be recovered exactly, clamped where noise would make a term negative, and the sharded term of
    rank r of a partition by nonzeros is its nonzero share of it (scanrs_plan_shards cuts by nonzeros)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for rep, sh, f in ((30.0, 190.0, 0.5), (12.5, 100.0, 0.4987), (0.0, 50.0, 0.25), (40.0, 0.0, 0.5)):
        out = bench.split_replicated_sharded(rep + sh, rep + sh * f, f)
        assert abs(out["replicated_ms_per_step"] - rep) < 0.011 and abs(out["sharded_ms_per_step"] - sh) < 0.011
    out = bench.split_replicated_sharded(100.0, 101.0, 0.5)  # the part ran slower than the whole (noise): nothing is sharded
    assert out == {"replicated_ms_per_step": 100.0, "sharded_ms_per_step": 0.0}
    out = bench.split_replicated_sharded(100.0, 10.0, 0.5)  # more than everything cannot be sharded
    assert out == {"replicated_ms_per_step": 0.0, "sharded_ms_per_step": 100.0}
    # ranks of a partition by nonzeros: the shares add up to the sharded term and follow the nonzeros
    import numpy as np

    nnz = np.array([130, 127, 125, 129, 131, 126, 128, 128], dtype=np.int64) * 10**6
    shares = [bench.sharded_share_of_rank(190.0, int(n), int(nnz.sum())) for n in nnz]
    assert abs(sum(shares) - 190.0) < 1e-9 and max(shares) / min(shares) == pytest.approx(131 / 125)
    assert all(name.startswith(bench.SHARDED_KERNEL_PREFIXES) for name in ("spmm_tile_kernel/long-outer", "tile_weights", "row_reduce_u32", "col_moments"))
    assert not any(name.startswith(bench.SHARDED_KERNEL_PREFIXES) for name in ("chol_rinv", "gemm_nn_mfma_f64"))


def test_dense_tile_kernel_leaves_registers_for_a_gather_wave(tmp_path):
    """The dense tile kernel (scan-rs_amd/csrc/tiles_dense.inc + the generated tile_dense_body.inc) pins its registers by hand; the
    compiler adds its own around the asm statement and ignores `amdgpu_num_vgpr`. Two tile waves per SIMD leave room for one wave of
    the overflow gather (72 VGPRs) only while the kernel stays at 216 of the 512 registers: at 220 the gather no longer co-resides
    and a pass regresses by a millisecond (profiles/HISTORY.md, round 5). Read from the code object's metadata."""
    import re
    import shutil
    import subprocess

    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    lib = os.path.join(ROOT, "scan-rs_amd", "lib", "libscanrs_amd.so")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("no llvm binutils in this image")
    copy = tmp_path / "lib.so"
    shutil.copy(lib, copy)
    subprocess.run([objdump, "--offloading", str(copy)], check=True, capture_output=True, cwd=tmp_path)
    found = {}
    for co in sorted(tmp_path.glob("lib.so.*gfx950*")):
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.name:\s+(\S*spmm_tile_dense_kernel\S*)(.*?)\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            found[m.group(1)] = (int(m.group(3)), int(m.group(4)))
    assert len(found) == 3, f"three instances of spmm_tile_dense_kernel (weight stream, tabo, tabi) expected in the library's gfx950 code objects: {sorted(found)}"
    for co in sorted(tmp_path.glob("lib.so.*gfx950*")):  # the flow layout's kernel (tiles_flow.inc): the same budget
        notes = subprocess.run([readelf, "--notes", str(co)], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.name:\s+(\S*spmm_tile_flow_kernel\S*)(.*?)\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            found[m.group(1)] = (int(m.group(3)), int(m.group(4)))
    assert len(found) == 5, sorted(found)
    for name, (vgprs, spills) in found.items():
        assert vgprs <= 216 and spills == 0, (name, vgprs, spills)


def test_generated_tile_kernel_stream_is_the_generators_output(tmp_path):
    """scan-rs_amd/csrc/tile_dense_body.inc is committed next to its generator (tools/gen_tile_dense_asm.py): the file in the tree must
    be what the generator writes with its default settings — which also runs the generator's own check of the DPP wait states."""
    import subprocess

    # round 6: three streams - the weight-stream form and the two forms that evaluate the map inside the kernel (table gathers by the
    # record itself: tabo / tabi); the generator also checks the wait states in front of their v_permlane*_swap instructions
    # ... the same two with the ring handed over by LDS counters (GEN_SYNC=cnt: built, measured slower, not compiled by default) and the
    # FLOW layout's round loop (GEN_FLOW=1, tiles_flow.inc: per-wave streams, ticks, counters; handle option tile_flow)
    for wsrc, name, extra in (("stream", "tile_dense_body.inc", {}), ("tabo", "tile_dense_body_tabo.inc", {}), ("tabi", "tile_dense_body_tabi.inc", {}),
                              ("tabo", "tile_dense_body_tabo_cnt.inc", {"GEN_SYNC": "cnt"}), ("tabi", "tile_dense_body_tabi_cnt.inc", {"GEN_SYNC": "cnt"}),
                              ("tabo", "tile_flow_body_tabo.inc", {"GEN_FLOW": "1"}), ("tabi", "tile_flow_body_tabi.inc", {"GEN_FLOW": "1"})):
        out = tmp_path / name
        env = {k: v for k, v in os.environ.items() if not k.startswith("GEN_")}
        env["GEN_WSRC"] = wsrc
        env.update(extra)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tile_dense_asm.py"), str(out)], check=True, capture_output=True, env=env)
        with open(os.path.join(ROOT, "scan-rs_amd", "csrc", name)) as f:
            assert f.read() == out.read_text(), name


def test_reserve_arena_bookkeeping(sa):
    """scanrs_reserve_device_memory's reserve is an arena since round 5 (best-fit carving, blocks given back, neighbouring holes merged).
    Its bookkeeping driven on a made-up address range — no device needed: thousands of random rounds with all invariants checked after
    every step (include/scanrs_amd.h: scanrs_debug_arena_selftest)."""
    for seed in (1, 2, 3):
        sa.debug_arena_selftest(4000, seed)
