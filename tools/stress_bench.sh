#!/bin/bash
# stress: bench in a loop with stage markers; stop at the first run that does not finish in 60 s
mkdir -p gpurun_out
for i in $(seq 1 $1); do
  SCANRS_TRACE=2 timeout -s ABRT 60 python -X faulthandler bench.py --no-cpu-baseline --no-host-delivery --steps 3 > gpurun_out/stress_out.txt 2> gpurun_out/stress_err.txt
  rc=$?
  if [ $rc -ne 0 ]; then echo "run $i rc=$rc HUNG"; tail -12 gpurun_out/stress_err.txt | cut -c1-150; exit 0; fi
done
echo "all $1 runs finished"
