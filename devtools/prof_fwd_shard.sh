#!/bin/bash
# rocprofv3 kernel stats of the forward bench at the c3 shard (B=4) -> gpurun_out/fwd_shard_stats.txt
export TMPDIR=/tmp
rm -rf gpurun_out/prof_fwds && mkdir -p gpurun_out/prof_fwds
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fwds -o fwd -- python3 bench.py --mode fwd --shard --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/prof_fwds.log 2>&1 || exit 1
python profiles/export_stats.py gpurun_out/prof_fwds/fwd_results.db gpurun_out/fwd_shard_stats.csv > gpurun_out/fwd_shard_stats.txt
rm -rf gpurun_out/prof_fwds
