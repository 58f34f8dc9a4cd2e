#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/pmc_split
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-host-delivery --no-heavy-tailed --no-randsvd --no-irlba --no-split-probe"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/tcc.json 2> $OUT/tcc.err
python3 - <<'PY'
import csv, glob, collections
for sub in ("fetch", "tcc"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"gpurun_out/pmc_split/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "spmm_tile_dense_kernel" not in n: continue
            key = "<1> tabo cell-major" if "<1>" in n else "<2> tabi gene-major"
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(key, r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(sub, k, {c: (round(x / cnt[(k, c)], 1), cnt[(k, c)]) for c, x in v.items()})
PY
rm -rf $OUT/fetch $OUT/tcc
