#!/bin/bash
# stress with host delivery (the download workers of the parked host team) and many steps on one handle + MultiMat shards
for i in $(seq 1 $1); do
  timeout -s ABRT 200 python -X faulthandler bench.py --no-cpu-baseline --no-heavy-tailed --no-randsvd --no-split-probe --steps 30 --warmup 2 > gpurun_out/stress2_out.txt 2> gpurun_out/stress2_err.txt
  rc=$?
  if [ $rc -ne 0 ]; then echo "run $i rc=$rc"; tail -12 gpurun_out/stress2_err.txt | cut -c1-150; exit 0; fi
  python -c "import json; d=json.load(open('gpurun_out/stress2_out.txt')); print('run $i', d['ms_per_step'], d['config']['irlba_ms'])"
done
echo "all $1 runs finished"
