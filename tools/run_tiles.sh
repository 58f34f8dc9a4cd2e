for kb in 0 14336 28672; do echo "== ov_tile_kb $kb"; python tools/pass_bench.py 1000000 100 0 opt.ov_tile_kb=$kb 2>&1 | grep -E "pass|spmm"; done > gpurun_out/pb_ovkb.log 2>&1
cat gpurun_out/pb_ovkb.log | cut -c1-330
