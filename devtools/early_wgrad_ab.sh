#!/bin/bash
# Gate-conv weight gradients of the later time steps launched from inside the BPTT loop (side stream) against
# the same two launches per conv after the loop (PLC_NO_EARLY_WGRAD=1).  Same launches, so identical bits.
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_early.txt
PLC_NO_EARLY_WGRAD=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_noearly.txt
PLC_NO_SIDE_STREAM=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_noside.txt
if cmp gpurun_out/digest_early.txt gpurun_out/digest_noearly.txt && cmp gpurun_out/digest_early.txt gpurun_out/digest_noside.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_early.txt gpurun_out/digest_noearly.txt gpurun_out/digest_noside.txt; exit 1; fi
REPS=6 timeout -k 10 200 python3 devtools/grad_repeat.py
for cfg in ${CFGS:-c3 c2 c4 c5}; do
    for rep in 1 2 3; do
        for sw in 0 1; do
            PLC_NO_EARLY_WGRAD=$sw timeout -k 10 200 python3 bench.py --config $cfg --mode train --no-cpu-baseline > gpurun_out/ab_early.json
            python3 -c "import json; d=json.load(open('gpurun_out/ab_early.json')); print('$cfg train PLC_NO_EARLY_WGRAD=$sw', round(d['ms_per_step'],4), 'ms/step')"
        done
    done
done
