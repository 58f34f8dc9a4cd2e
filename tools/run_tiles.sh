python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rank_deficient or dense_outer or heavy_tailed" > gpurun_out/gputest_g.log 2>&1; tail -25 gpurun_out/gputest_g.log | cut -c1-220
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err; python -c "
import json
d=json.load(open('gpurun_out/bench_c.json'))
c=d['config']
print(d['value'], d['ms_per_step'], c['first_call_s'], c['device_resident_ms_per_step'], c['fresh_result_arrays_ms_per_step'], c['resident_bytes_per_nonzero'], c['heavy_tailed_ms_per_step'])
r=d['roofline']; print(r['avg_launch_ms'], r['frac'], r['traffic'], r['traffic_source'], r['onchip'], r['wasted_traffic_ratio'])
"; tail -3 gpurun_out/bench_c.err
