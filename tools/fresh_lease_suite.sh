#!/bin/bash
# One run of the whole -m gpu suite as the FIRST process of a freshly leased box (each gpurun call is one), with the stage markers
# of the bounded waits on (SCANRS_TRACE=2) and the Python stacks dumped on a hard timeout. Appends one line to
# gpurun_out/fresh_lease_runs.log; if a wait ever times out, the library's report is in the pytest log kept beside it.
#   usage (from the build container): gpurun --timeout 2700 -- 'bash tools/fresh_lease_suite.sh TAG'
TAG=${1:-run}
mkdir -p gpurun_out
t0=$(date +%s)
SCANRS_FAULT_LOG=gpurun_out/stacks_$TAG.log SCANRS_TRACE=2 timeout -s USR1 -k 30 2400 python -m pytest tests -m gpu -x -q > gpurun_out/suite_$TAG.log 2> gpurun_out/suite_$TAG.err
rc=$?
t1=$(date +%s)
tailline=$(grep -E "passed|failed|error" gpurun_out/suite_$TAG.log | tail -1)
timeouts=$(grep -c "device wait timed out" gpurun_out/suite_$TAG.err)
echo "$TAG rc=$rc wall=$((t1-t0))s bounded-wait-timeouts=$timeouts :: $tailline" | tee -a gpurun_out/fresh_lease_runs.log
if [ $rc -ne 0 ]; then tail -30 gpurun_out/suite_$TAG.log | cut -c1-200; grep -B2 -A12 "device wait timed out" gpurun_out/suite_$TAG.err | head -60; fi
# keep the stderr small for the trip back: only the lines around a timeout (if any) and the tail
tail -c 20000 gpurun_out/suite_$TAG.err > gpurun_out/suite_$TAG.err.tail; rm -f gpurun_out/suite_$TAG.err
