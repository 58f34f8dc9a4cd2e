"""Turn raw rocprofv3 output directories into the small summaries committed under profiles/.
usage: python profiles/summarize.py TAG STATS_DIR FETCH_DIR WRITE_DIR   (run on the GPU box; raw traces stay in gpurun_out/)"""
import collections
import csv
import glob
import json
import sys

import numpy as np


def one(pattern):
    return glob.glob(pattern, recursive=True)[0]


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    out = {}
    for c, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows = list(csv.DictReader(open(one(f"{d}/**/*_counter_collection.csv"))))
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            n = r["Kernel_Name"]
            if "spmm_gather2d" in n:
                key = "spmm_gather2d_kernel<1>/long-outer" if int(r["Grid_Size"]) > 10_000_000 else "spmm_gather2d_kernel<1>/short-outer"
            elif "gram_tiled_kernel" in n:
                key = "gram_tiled_kernel"
            elif "gemm_tiled_kernel" in n:
                key = "gemm_tiled_kernel"
            elif "row_reduce2d_kernel<2>" in n:
                key = "row_reduce2d_kernel<2>"
            elif "row_reduce_kernel<2>" in n:
                key = "row_reduce_kernel<2>"
            elif "row_reduce_kernel<0>" in n:
                key = "row_reduce_kernel<0>"
            else:
                continue
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            out.setdefault(k, {})[c] = {"launches": n, "sum_KB": v, "avg_KB_per_launch": v / n}
    for k, v in out.items():
        f = v["FETCH_SIZE"]["avg_KB_per_launch"] * 1024
        w = v["WRITE_SIZE"]["avg_KB_per_launch"] * 1024
        v["hbm_bytes_per_launch_raw"] = f + w
        v["hbm_bytes_per_launch_corrected"] = 2 * f + w
    meta = {
        "_how": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py "
                "--steps 1 --warmup 1 --no-cpu-baseline ; 1M x 33k, 3% nnz, k=50, MI355X. Counter unit KB; each run holds 3 PCAs "
                "(warmup, timed step, event-recording step). 'corrected' doubles FETCH_SIZE as MI355X_MICROARCH.md section HBM "
                "prescribes for 16-B-per-lane coalesced reads (uncalibrated for this gather pattern: an upper estimate).",
        "kernels": out,
    }
    json.dump(meta, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    rows = list(csv.DictReader(open(one(f"{stats_dir}/**/*_kernel_stats.csv"))))
    tr = list(csv.DictReader(open(one(f"{stats_dir}/**/*_kernel_trace.csv"))))
    sp = [r for r in tr if "spmm_gather2d" in r["Kernel_Name"]]
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sp])
    g = np.array([int(r["Grid_Size_X"]) for r in sp])
    with open(f"profiles/{tag}_bench_1M_kernel_stats.csv", "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (1M x 33k, 3% nnz, k=50; "
                 "MI355X; 7 PCAs: warmup + 3 timed + 3 event-recording)\n")
        fo.write(f"# spmm_gather2d_kernel<1> by orientation (kernel trace): long-outer = cell-major copy (grid 64M threads) "
                 f"calls={(g > 10_000_000).sum()} avg_ms={d[g > 10_000_000].mean():.4f}; short-outer = gene-major copy "
                 f"calls={(g <= 10_000_000).sum()} avg_ms={d[g <= 10_000_000].mean():.4f}\n")
        w = csv.writer(fo)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            if "scanrs" in r["Name"] or "rocprim" in r["Name"]:
                w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in out.items()}))


if __name__ == "__main__":
    main()
