for o in 0 1; do python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-heavy-tailed --opt dense_side_no_lds=$o > gpurun_out/bench_e$o.json 2> gpurun_out/bench_e$o.err; python -c "
import json
d=json.load(open('gpurun_out/bench_e$o.json'))
c=d['config']
print('no_lds=$o', d['value'], d['ms_per_step'], c['device_resident_ms_per_step'], d['roofline']['avg_launch_ms'], {k:v for k,v in d['roofline']['kernel_ms_per_step'].items() if 'gemm' in k or 'gram' in k})
"; done
bash tools/trace_timeline.sh c --opt dense_side_no_lds=1 > /dev/null 2>&1; tail -3 gpurun_out/trace_c/timeline.txt
