#!/bin/bash
# One --pmc pass (kernel trace only, one counter group within the per-block limits) over a short bench run:
#   bash profiles/pmc_pass.sh <tag> <counter> <bench.py arguments...>     -> gpurun_out/pmc_<tag>/
export TMPDIR=/tmp
tag=$1; counter=$2; shift 2
rm -rf gpurun_out/pmc_$tag && mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 240 rocprofv3 --pmc $counter --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o pmc -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
