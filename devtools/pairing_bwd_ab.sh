#!/bin/bash
# Skewed backward loop (paired launches) against the step-by-step one: identical digests, then step times.
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_pair.txt
PLC_NO_PAIRING=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_nopair.txt
if cmp gpurun_out/digest_pair.txt gpurun_out/digest_nopair.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_pair.txt gpurun_out/digest_nopair.txt; exit 1; fi
for cfg in ${CFGS:-c3 c2 c4}; do
    for rep in 1 2; do
    for sw in 0 1; do
        PLC_NO_PAIRING=$sw timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/ab_pair_${cfg}_$sw.json
        python3 -c "import json; d=json.load(open('gpurun_out/ab_pair_${cfg}_$sw.json')); print('$cfg PLC_NO_PAIRING=$sw', round(d['ms_per_step'],4), 'ms/step; fwd', round(1e3*d['config']['global_batch']*d['config']['T']/d['fwd_frames_per_s'],4))"
    done
    done
done
