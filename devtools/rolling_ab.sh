#!/bin/bash
# Rolling-window convs against the tile-at-a-time kernels: identical digests, then kernel and step times.
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_roll.txt
PLC_NO_ROLLING=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_tile.txt
if cmp gpurun_out/digest_roll.txt gpurun_out/digest_tile.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_roll.txt gpurun_out/digest_tile.txt; exit 1; fi
for cfg in ${CFGS:-c3 c2}; do
    for sw in 0 1; do
        PLC_NO_ROLLING=$sw timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/ab_roll_${cfg}_$sw.json
        python3 -c "import json; d=json.load(open('gpurun_out/ab_roll_${cfg}_$sw.json')); print('$cfg PLC_NO_ROLLING=$sw', round(d['ms_per_step'],4), 'ms/step')"
    done
    bash profiles/prof_cfg.sh $cfg train > /dev/null
    grep -E "convr|16, 0, 2, 3, 0, 1, 4, 2|8, 0, 1, 3, 0, 1, 8" gpurun_out/${cfg}_train_stats.csv | cut -c1-110
done
