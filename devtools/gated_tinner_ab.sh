#!/bin/bash
# Time-innermost tile order of the gated post conv (multiplier tile kept in LDS along a chain of time steps)
# against the image-major order (PLC_NO_GATED_TINNER=1); PLC_GATED_TSEGS = chains per position.
set -e
mkdir -p gpurun_out
timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_tinner.txt
PLC_NO_GATED_TINNER=1 timeout -k 10 200 python3 devtools/grad_digest.py > gpurun_out/digest_imgmajor.txt
if cmp gpurun_out/digest_tinner.txt gpurun_out/digest_imgmajor.txt; then echo "digests identical"; else echo "DIGESTS DIFFER"; paste gpurun_out/digest_tinner.txt gpurun_out/digest_imgmajor.txt; exit 1; fi
for cfg in ${CFGS:-c3}; do
    for mode in fwd train; do
        for rep in 1 2; do
            PLC_NO_GATED_TINNER=1 timeout -k 10 200 python3 bench.py --config $cfg --mode $mode --no-cpu-baseline > gpurun_out/ab_tinner.json
            python3 -c "import json; d=json.load(open('gpurun_out/ab_tinner.json')); print('$cfg $mode image-major', round(d['ms_per_step'],4), 'ms/step')"
            for segs in ${SEGS:-0 1 2 5 10}; do
                PLC_GATED_TSEGS=$segs timeout -k 10 200 python3 bench.py --config $cfg --mode $mode --no-cpu-baseline > gpurun_out/ab_tinner.json
                python3 -c "import json; d=json.load(open('gpurun_out/ab_tinner.json')); print('$cfg $mode time-innermost segs=$segs', round(d['ms_per_step'],4), 'ms/step')"
            done
        done
    done
done
