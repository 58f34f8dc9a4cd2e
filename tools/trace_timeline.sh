#!/bin/bash
# kernel trace of one bench step + the gap table between the tile kernels (tools/timeline.py); usage: tools/trace_timeline.sh TAG [bench args]
set -u
TAG=${1:-tl}; shift
OUT=gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-heavy-tailed --no-randsvd --no-irlba --no-split-probe "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 tools/timeline.py $OUT/raw 11 > $OUT/timeline.txt 2>&1
rm -rf $OUT/raw
cat $OUT/timeline.txt
