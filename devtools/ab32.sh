#!/bin/bash
# Same-box A/B of the schedule switches at the default workload (c3, global batch 32 on one GPU).
run() {
  tag=$1; shift
  timeout -k 10 200 env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab32_$tag.json 2>gpurun_out/ab32_$tag.err || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/ab32_$tag.json')); print('$tag: train %.3f ms/step %.0f frames/s | fwd %.0f frames/s' % (d['ms_per_step'], d['value'], d['fwd_frames_per_s']))"
}
run base X=1
run no_bwd_split PLC_NO_BWD_SPLIT=1
run no_fwd_split PLC_NO_FWD_SPLIT=1
run no_side PLC_NO_SIDE_STREAM=1
run no_pairing PLC_NO_PAIRING=1
run base2 X=1
