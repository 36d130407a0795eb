#!/bin/bash
# Copies what profiles/r3_collect.sh and profiles/r3_collect_prof.sh left in gpurun_out/ into the tracked profiles/r3_* files.
set -e
cp gpurun_out/r3_*_bench.json profiles/
for t in train_c3 train_c3_shard fwd_c3 fwd_c3_shard train_p2 train_c3_shard_single_stream fwd_c3_shard_single_stream train_c3_single_stream roofline_launches roofline_launches_shard; do
    cp gpurun_out/r3_${t}_stats.csv profiles/r3_${t}_kernel_stats.csv
    cp gpurun_out/r3_${t}_by_grid.csv profiles/r3_${t}_kernels_by_grid.csv
done
cp gpurun_out/r3_train_c3_step_traffic.json gpurun_out/r3_train_c3_shard_step_traffic.json profiles/
cp gpurun_out/r3_pointwise_bwd_pair_traffic.json gpurun_out/r3_cell2_traffic.json profiles/
cp gpurun_out/r3_train_c3_kernel_traffic.csv gpurun_out/r3_train_c3_shard_kernel_traffic.csv profiles/
echo copied
