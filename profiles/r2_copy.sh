#!/bin/bash
# Copies what profiles/r2_collect.sh left in gpurun_out/ into the tracked profiles/r2_* files.
set -e
cp gpurun_out/r2_*_bench.json profiles/
cp gpurun_out/train_stats.csv profiles/r2_train_c3_kernel_stats.csv
cp gpurun_out/fwd_stats.csv profiles/r2_fwd_c3_kernel_stats.csv
cp gpurun_out/c5_fwd_stats.csv profiles/r2_fwd_c5_fp8_kernel_stats.csv
cp gpurun_out/p2_train_stats.csv profiles/r2_train_p2_kernel_stats.csv
cp gpurun_out/c0_train_stats.csv profiles/r2_train_c0_kernel_stats.csv
cp gpurun_out/train_roofline_kernels_by_grid.csv profiles/r2_train_c3_roofline_kernels_by_grid.csv
cp gpurun_out/r2_train_c3_kernel_traffic.csv profiles/r2_train_c3_kernel_traffic.csv
cp gpurun_out/r2_pointwise_bwd_traffic.json profiles/r2_pointwise_bwd_traffic.json
cp gpurun_out/r2_train_c3_step_traffic.json profiles/r2_train_c3_step_traffic.json
echo copied
