#!/bin/bash
# rocprofv3 kernel stats of the default bench (train step, c3) -> gpurun_out/train_stats.csv
export TMPDIR=/tmp
rm -rf gpurun_out/prof_train && mkdir -p gpurun_out/prof_train
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o train -- python3 bench.py --steps 10 --warmup 3 > gpurun_out/prof_train.log 2>&1 || exit 1
python profiles/export_stats.py gpurun_out/prof_train/train_results.db gpurun_out/train_stats.csv
python profiles/by_grid.py gpurun_out/prof_train/train_results.db gpurun_out/train_roofline_kernels_by_grid.csv
