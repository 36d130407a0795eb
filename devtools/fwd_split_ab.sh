#!/bin/bash
# Two half-batch chains on two streams for the forward recurrence against the single chain (PLC_NO_FWD_SPLIT=1).
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_fsplit.txt
PLC_NO_FWD_SPLIT=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_fone.txt
if cmp gpurun_out/digest_fsplit.txt gpurun_out/digest_fone.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_fsplit.txt gpurun_out/digest_fone.txt; exit 1; fi
for cfg in ${CFGS:-c3 c2 c4}; do
    for mode in fwd train; do
        for rep in 1 2; do
            for sw in 1 0; do
                PLC_NO_FWD_SPLIT=$sw timeout -k 10 200 python3 bench.py --config $cfg --mode $mode --no-cpu-baseline > gpurun_out/ab_fsplit.json
                python3 -c "import json; d=json.load(open('gpurun_out/ab_fsplit.json')); print('$cfg $mode PLC_NO_FWD_SPLIT=$sw', round(d['ms_per_step'],4), 'ms/step')"
            done
        done
    done
done
