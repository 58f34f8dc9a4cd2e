#!/bin/bash
# stress: a pytest selection in a loop with stage markers (SCANRS_TRACE=2, capture off); stops at the first run that does not
# finish in $2 seconds and prints where it stood.   usage: stress_test.sh N SECONDS pytest-args...
mkdir -p gpurun_out
n=$1; secs=$2; shift 2
for i in $(seq 1 $n); do
  SCANRS_FAULT_LOG=gpurun_out/stacks.log SCANRS_TRACE=2 timeout -s USR1 -k 20 $secs python -m pytest -m gpu -x -q -s "$@" > gpurun_out/stress_test.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "run $i rc=$rc"; tail -25 gpurun_out/stress_test.log | cut -c1-160; echo "--- python stacks"; head -40 gpurun_out/stacks.log | cut -c1-160; exit 0; fi
done
echo "all $n runs finished"
