#!/bin/bash
# Same-box comparison of two builds of the library: devtools/libplc_head.so (an older build) against the
# in-tree one, alternating, default bench configuration unless CFG is set.
set -e
mkdir -p gpurun_out
for rep in 1 2; do
    for which in head tree; do
        if [ $which = head ]; then export PLC_LIB_PATH=$PWD/devtools/libplc_head.so; else unset PLC_LIB_PATH; fi
        timeout -k 10 200 python3 bench.py --config ${CFG:-c3} --no-cpu-baseline > gpurun_out/ab_lib_$which.json
        python3 -c "import json; d=json.load(open('gpurun_out/ab_lib_$which.json')); print('$which', round(d['ms_per_step'],4), 'ms/step')"
    done
done
