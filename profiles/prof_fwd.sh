#!/bin/bash
# rocprofv3 kernel stats of the forward bench (c3) -> gpurun_out/fwd_stats.csv
export TMPDIR=/tmp
rm -rf gpurun_out/prof_fwd && mkdir -p gpurun_out/prof_fwd
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fwd -o fwd -- python3 bench.py --mode fwd --steps 20 --warmup 3 > gpurun_out/prof_fwd.log 2>&1 || exit 1
python profiles/export_stats.py gpurun_out/prof_fwd/fwd_results.db gpurun_out/fwd_stats.csv
