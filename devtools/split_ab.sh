#!/bin/bash
# Two half-batch chains on two streams for the backward recurrence against the single chain (PLC_NO_BWD_SPLIT=1).
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_one.txt
PLC_NO_BWD_SPLIT=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_split.txt
if cmp gpurun_out/digest_one.txt gpurun_out/digest_split.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_one.txt gpurun_out/digest_split.txt; exit 1; fi
for cfg in ${CFGS:-c3 c2 c4}; do
    for rep in 1 2; do
    for sw in 0 1; do
        PLC_NO_BWD_SPLIT=$sw timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/ab_split_${cfg}_$sw.json
        python3 -c "import json; d=json.load(open('gpurun_out/ab_split_${cfg}_$sw.json')); print('$cfg PLC_NO_BWD_SPLIT=$sw', round(d['ms_per_step'],4), 'ms/step')"
    done
    done
done
