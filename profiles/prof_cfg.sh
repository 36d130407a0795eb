#!/bin/bash
# rocprofv3 kernel stats of one bench configuration:  bash profiles/prof_cfg.sh <config> [fwd|train]
export TMPDIR=/tmp
cfg=${1:-p2}; mode=${2:-train}
rm -rf gpurun_out/prof_$cfg && mkdir -p gpurun_out/prof_$cfg
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$cfg -o run -- python3 bench.py --config $cfg --mode $mode --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$cfg.log 2>&1 || exit 1
python profiles/export_stats.py gpurun_out/prof_$cfg/run_results.db gpurun_out/${cfg}_${mode}_stats.csv
