#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + the PMC passes of the headline bench command, raw output under
# gpurun_out/prof_$1/, then profiles/summarize.py turns it into the committed summaries profiles/$1_*.
#   usage: tools/profile_round.sh TAG COMMIT     (e.g. r04a $(git rev-parse --short HEAD))
# Counters go in their own runs (rocprofv3 --pmc with --kernel-trace only), the program directly after `--`.
set -u
TAG=${1:-r04a}
export SCANRS_COMMIT=${2:-"working tree"}   # the build container passes $(git rev-parse --short HEAD): the GPU box has no .git
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-host-delivery --no-heavy-tailed --no-randsvd --no-irlba --no-split-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-delivery --no-heavy-tailed --no-randsvd --no-irlba --no-split-probe > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS > $OUT/tcc.json 2> $OUT/tcc.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq -- python3 bench.py $ARGS > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/tcp -- python3 bench.py $ARGS > $OUT/tcp.json 2> $OUT/tcp.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/lds -- python3 bench.py $ARGS > $OUT/lds.json 2> $OUT/lds.err
python3 profiles/summarize.py $TAG $OUT > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
ls -la $OUT
# the raw traces are large: keep only the summaries for the trip back
mkdir -p gpurun_out/prof_${TAG}_keep && cp profiles/${TAG}_* $OUT/*.json $OUT/summary.log gpurun_out/prof_${TAG}_keep/ 2>/dev/null
rm -rf $OUT
