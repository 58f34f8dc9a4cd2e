python __graft_entry__.py smoke 2>&1 | tail -1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "helper_thread or deadline or wide_weight or heavy_tailed or lds_staged_product_whole or rank_deficient or dense_outer" > gpurun_out/gputest_i.log 2>&1; tail -5 gpurun_out/gputest_i.log | cut -c1-220
python tools/first_call.py 1000000 3 2>&1 | grep -E "first call|counters"
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; python -c "
import json
d=json.load(open('gpurun_out/bench_f.json'))
c=d['config']
print(d['value'], d['ms_per_step'], c['first_call_s'], c['device_resident_ms_per_step'], c['fresh_result_arrays_ms_per_step'], c['resident_bytes_per_nonzero'], c['heavy_tailed_ms_per_step'])
print(c['first_call_breakdown_ms'])
print({k:v for k,v in d['roofline']['kernel_ms_per_step'].items() if 'weights' in k or 'tile' in k})
"; tail -3 gpurun_out/bench_f.err
