"""Turn the raw rocprofv3 output of tools/profile_round.sh into the small summaries committed under profiles/.
usage: python profiles/summarize.py TAG RAW_DIR      (run on the GPU box; raw traces stay in gpurun_out/)

Writes  profiles/TAG_bench_1M_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (+ the dominant kernel by orientation,
                                                  + every dense-kernel launch class by grid size)
        profiles/TAG_pmc_traffic.json            HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes)
        profiles/TAG_pmc_counters.json           L2 hit rate, L1->L2 requests, SQ issue / wait counters per kernel class
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

import numpy as np


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def classify(name, grid):
    if "spmm_tile_kernel" in name or "spmm_tile_dense_kernel" in name:  # the persistent tile kernel of the hybrid product (both orientations: one grid of 256 workgroups)
        return "spmm_tile_kernel"
    if "spmm_gather_ov_kernel" in name:  # the gather over the overflow part, beside the tile kernel (4 vectors per wave)
        return "spmm_gather2d_ov/long-outer" if grid > 2_500_000 else "spmm_gather2d_ov/short-outer"
    if "spmm_gather2d_kernel" in name:
        return "spmm_gather2d_kernel<1>/long-outer" if grid > 10_000_000 else "spmm_gather2d_kernel<1>/short-outer"
    for k in ("col_moments_kernel", "tile_weights_unit_kernel", "tile_weights_kernel", "tile_ratio_table_kernel", "chol_rinv_kernel", "tile_scale_panel_kernel", "tile_finish_kernel", "tile_assign_wave_kernel", "tile_assign_kernel",
              "tile_slotless_kernel", "validate_stream_kernel", "unpack_key_kernel", "pack_key_kernel", "gram_tiled_kernel", "gemm_tiled_kernel", "slice_walk_kernel<1>", "slice_walk_kernel<0>", "row_reduce2d_kernel<2>", "row_reduce_kernel<2>", "row_reduce_kernel<0>",
              "weighted_colsum_partial_kernel", "spmv2d_kernel", "spmv_lds_kernel", "gram_kernel", "gemm_nn_kernel"):
        if k in name:
            return k
    return None


def counters(d):
    """{kernel class: {counter: [sum over launches, launches]}}"""
    f = one(f"{d}/**/*_counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        key = classify(r["Kernel_Name"], int(r["Grid_Size"]))
        if key is None:
            continue
        c = out[key][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
    return out


def main():
    tag, raw = sys.argv[1:3]
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        commit = os.environ.get("SCANRS_COMMIT", "working tree")
    # the sources the kernels were built from: bench.py quotes `traffic` from this file only while they are unchanged
    import hashlib

    hsrc = hashlib.sha256()
    # (the same list, in the same order, as bench.py's KERNEL_SOURCES)
    for rel in ("scan-rs_amd/csrc/tiles.hip", "scan-rs_amd/csrc/tiles_dense.inc", "scan-rs_amd/csrc/tile_dense_body.inc", "scan-rs_amd/csrc/tile_dense_body_tabo.inc",
                  "scan-rs_amd/csrc/tile_dense_body_tabi.inc", "scan-rs_amd/csrc/kernels.hip",
                "scan-rs_amd/csrc/device_map.hpp"):
        with open(rel, "rb") as f:
            hsrc.update(f.read())
    src_hash = hsrc.hexdigest()[:16]
    if os.environ.get("SCANRS_COMMIT"):
        commit = os.environ["SCANRS_COMMIT"]
    # ---- HBM traffic ---------------------------------------------------------------------------------------------------
    tr = {}
    for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for key, cs in counters(f"{raw}/{sub}").items():
            if cname in cs:
                s, n = cs[cname]
                tr.setdefault(key, {})[cname] = {"launches": n, "sum_KB": s, "avg_KB_per_launch": s / n}
    for k, v in tr.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f = v["FETCH_SIZE"]["avg_KB_per_launch"] * 1024
            w = v["WRITE_SIZE"]["avg_KB_per_launch"] * 1024
            v["hbm_bytes_per_launch_raw"] = f + w
            v["hbm_bytes_per_launch_corrected"] = 2 * f + w
    json.dump({
        "_how": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py "
                "--steps 1 --warmup 1 --no-cpu-baseline --no-host-delivery --no-heavy-tailed ; 1M x 33k, 3% nnz, k=50, MI355X. Counter unit KB; each run holds 3 PCAs "
                "(first pass, timed step, event-recording step). 'corrected' doubles FETCH_SIZE as MI355X_MICROARCH.md section HBM "
                "prescribes for 16-B-per-lane coalesced reads (uncalibrated for the gather pattern: an upper estimate).",
        "commit": commit, "kernel_source_sha256_16": src_hash, "kernels": tr}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    # ---- on-chip counters ----------------------------------------------------------------------------------------------
    cc = {}
    for sub in ("tcc", "sq", "tcp", "lds"):
        for key, cs in counters(f"{raw}/{sub}").items():
            for cn, (s, n) in cs.items():
                cc.setdefault(key, {})[cn] = {"avg_per_launch": s / n, "launches": n}
    for key, v in cc.items():
        g = lambda c: v[c]["avg_per_launch"] if c in v else None  # noqa: E731
        d = {}
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"):
                if g(c) is not None:
                    d[c + "/SQ_WAVE_CYCLES"] = g(c) / g("SQ_WAVE_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
            d["lds_bank_conflict_cycles/lds_active_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        if g("SQ_INSTS_VALU") and g("SQ_INSTS_VMEM_RD"):
            d["valu_insts_per_vmem_read"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_VMEM_RD")
        v["derived"] = d
    class_ms = {}
    # ---- kernel stats -----------------------------------------------------------------------------------------------------
    sfile, tfile = one(f"{raw}/stats/**/*_kernel_stats.csv"), one(f"{raw}/stats/**/*_kernel_trace.csv")
    if sfile and tfile:
        rows = list(csv.DictReader(open(sfile)))
        trc = list(csv.DictReader(open(tfile)))
        with open(f"profiles/{tag}_bench_1M_kernel_stats.csv", "w") as fo:
            fo.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery  (1M x 33k, 3% nnz, "
                     f"k=50; MI355X; 7 PCAs: first pass + 3 timed + 3 event-recording); commit {commit}\n")
            by = collections.defaultdict(list)
            for r in trc:
                key = classify(r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))
                if key:
                    by[(key, int(r["Grid_Size_X"]), int(r.get("Grid_Size_Y", 1) or 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            fo.write("# launch classes (kernel, grid x, grid y): calls, avg ms, total ms\n")
            agg = collections.defaultdict(lambda: [0, 0.0])
            for (key, gx, gy), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                d = np.array(d)
                agg[key][0] += len(d)
                agg[key][1] += d.sum()
                if d.sum() > 1.0 and not key.startswith("spmm"):
                    fo.write(f"#   {key} grid=({gx},{gy}) calls={len(d)} avg_ms={d.mean():.4f} total_ms={d.sum():.2f}\n")
            for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                fo.write(f"# class {key}: calls={n} avg_ms={t / n:.4f} total_ms={t:.2f}\n")
                class_ms[key.split("/")[0]] = round(t / n, 4) if key.split("/")[0] not in class_ms else class_ms[key.split("/")[0]]
            w = csv.writer(fo)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                if "scanrs" in r["Name"] or "rocprim" in r["Name"]:
                    w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    json.dump({
        "_how": "separate rocprofv3 --pmc passes (tools/profile_round.sh): {TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum}, {SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
                "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY}, {TCP_TOTAL_CACHE_ACCESSES_sum "
                "TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum}; same command as the traffic passes; averages per launch of a kernel class. "
                "SQ_* are in quad-cycles summed over waves (MI355X_MICROARCH.md, rocprofv3 PMC slots).",
        "commit": commit, "kernel_source_sha256_16": src_hash,
        # average launch duration per kernel class in the --stats run (no counters): bench.py derives the shader clock the launch held
        # from SQ_BUSY_CYCLES with it
        "avg_launch_ms": class_ms, "kernels": cc}, open(f"profiles/{tag}_pmc_counters.json", "w"), indent=1)
    print(json.dumps({k: round(v.get("hbm_bytes_per_launch_corrected", 0) / 1e6, 1) for k, v in tr.items()}))
    print(json.dumps({k: v.get("derived") for k, v in cc.items()}, indent=1))


if __name__ == "__main__":
    main()
