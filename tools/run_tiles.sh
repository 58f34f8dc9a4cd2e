for x in 1.45 1.7 2.0; do for mn in 0.35 0.7; do
echo "== x $x min $mn heavy"; SCANRS_TRACE=1 python tools/pass_bench.py 1000000 100 0 gene_shape=0.1 shared_profile=1 opt.tile_split_x=$x opt.tile_split_min=$mn 2>&1 | grep -E "tile layout:|pass|count|fill" 
done; done > gpurun_out/pb_sweep_heavy.log 2>&1
for x in 1.45 1.7 2.0 2.4; do
echo "== x $x std"; SCANRS_TRACE=1 python tools/pass_bench.py 1000000 100 0 opt.tile_split_x=$x 2>&1 | grep -E "tile layout:|pass|count|fill"
done > gpurun_out/pb_sweep_std.log 2>&1
echo "== split off std"; python tools/pass_bench.py 1000000 100 0 opt.tile_split=0 2>&1 | grep -E "pass" >> gpurun_out/pb_sweep_std.log
cat gpurun_out/pb_sweep_heavy.log gpurun_out/pb_sweep_std.log | cut -c1-250
