#!/bin/bash
# Attention branches on the side stream (beside the stem / the BPTT loop) against the single stream (PLC_NO_SIDE_STREAM=1).
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_side.txt
PLC_NO_SIDE_STREAM=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_noside.txt
if cmp gpurun_out/digest_side.txt gpurun_out/digest_noside.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_side.txt gpurun_out/digest_noside.txt; exit 1; fi
for cfg in ${CFGS:-c3 c2 p2 c1}; do
    for mode in fwd train; do
        for rep in 1 2; do
            for sw in 1 0; do
                PLC_NO_SIDE_STREAM=$sw timeout -k 10 200 python3 bench.py --config $cfg --mode $mode --no-cpu-baseline > gpurun_out/ab_side.json
                python3 -c "import json; d=json.load(open('gpurun_out/ab_side.json')); print('$cfg $mode PLC_NO_SIDE_STREAM=$sw', round(d['ms_per_step'],4), 'ms/step')"
            done
        done
    done
done
