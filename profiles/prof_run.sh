#!/bin/bash
# rocprofv3 kernel stats of one bench run:  bash profiles/prof_run.sh <tag> <bench.py arguments...>
#   -> gpurun_out/<tag>_stats.csv (per-kernel totals) and gpurun_out/<tag>_by_grid.csv (per kernel and grid size)
export TMPDIR=/tmp
tag=$1; shift
rm -rf gpurun_out/prof_$tag && mkdir -p gpurun_out/prof_$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o run -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || exit 1
python3 profiles/export_stats.py gpurun_out/prof_$tag/run_results.db gpurun_out/${tag}_stats.csv
python3 profiles/by_grid.py gpurun_out/prof_$tag/run_results.db gpurun_out/${tag}_by_grid.csv
