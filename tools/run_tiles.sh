python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lds_staged or wave_level or dense_outer or invalid_sparse or heavy_tailed or device_memory or config" > gpurun_out/gputest_h.log 2>&1; tail -5 gpurun_out/gputest_h.log | cut -c1-220
SCANRS_TRACE=1 python tools/first_call.py 1000000 2 2>&1 | grep -E "create:|first call|counters" | head -20
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err; python -c "
import json
d=json.load(open('gpurun_out/bench_d.json'))
c=d['config']
print(d['value'], d['ms_per_step'], c['first_call_s'], c['device_resident_ms_per_step'], c['fresh_result_arrays_ms_per_step'], c['resident_bytes_per_nonzero'], c['heavy_tailed_ms_per_step'])
print(c['first_call_breakdown_ms'])
r=d['roofline']; print(r['avg_launch_ms'], r['frac'], r['traffic'], r['traffic_source'][:80], r['onchip'], r['wasted_traffic_ratio'])
"; tail -3 gpurun_out/bench_d.err
