#!/bin/bash
# Pointwise backward inside the data-gradient conv against the separate pass: identical digests, then step times.
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_fused.txt
PLC_NO_FUSED_PW=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_unfused.txt
if cmp gpurun_out/digest_fused.txt gpurun_out/digest_unfused.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_fused.txt gpurun_out/digest_unfused.txt; exit 1; fi
for cfg in ${CFGS:-c3 c2 c4}; do
    for rep in 1 2; do
    for sw in 0 1; do
        PLC_NO_FUSED_PW=$sw timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/ab_fused_${cfg}_$sw.json
        python3 -c "import json; d=json.load(open('gpurun_out/ab_fused_${cfg}_$sw.json')); print('$cfg PLC_NO_FUSED_PW=$sw', round(d['ms_per_step'],4), 'ms/step')"
    done
    done
done
