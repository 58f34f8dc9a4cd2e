#!/bin/bash
# kernel trace of ONE first call (tools/first_call.py, 1 repetition): every kernel of 0.5 ms and more between the handle's validation
# and the end of its first PCA, with stream (queue) and times relative to the validation - which builds run beside which pass.
# usage: tools/first_call_timeline.sh TAG
set -u
TAG=${1:-fc}
OUT=gpurun_out/fctl_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 tools/first_call.py 1000000 1 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
i0 = next(i for i, r in enumerate(rows) if "validate_stream" in r[2])
t0 = rows[i0][0]
tiles = [i for i, r in enumerate(rows) if "spmm_tile_kernel" in r[2] and i > i0]
end = rows[tiles[10]][1] if len(tiles) > 10 else rows[-1][1]
def short(n):
    for p in ("void ", "scanrs::", "(anonymous namespace)::", "rocprim::ROCPRIM_400200_NS::detail::", "rocprim::ROCPRIM_400001_NS::detail::"):
        n = n.replace(p, "")
    return n.split("(")[0][:70]
with open(sys.argv[1] + "/timeline.txt", "w") as fo:
    for s, e, n, q in rows[i0:]:
        if s > end: break
        if e - s >= 500_000:
            fo.write(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f}  q{q:>3}  {(e - s) / 1e6:8.3f} ms  {short(n)}\n")
PY
rm -rf $OUT/raw
cat $OUT/timeline.txt
