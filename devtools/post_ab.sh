#!/bin/bash
# Same-box A/B: the two output convs fused into one kernel (default) against the separate launches.
run() {
  tag=$1; shift
  timeout -k 10 200 env "$@" > gpurun_out/postab_$tag.json 2>gpurun_out/postab_$tag.err || { tail -5 gpurun_out/postab_$tag.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/postab_$tag.json')); print('$tag: train %.3f ms/step %.0f frames/s | fwd %.0f frames/s' % (d['ms_per_step'], d['value'], d['fwd_frames_per_s']))"
}
run fused_b32 X=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
run split_b32 PLC_NO_POST_FUSION=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
run fused_b4 X=1 python3 bench.py --shard --no-cpu-baseline
run split_b4 PLC_NO_POST_FUSION=1 python3 bench.py --shard --no-cpu-baseline
run fused_p2 X=1 python3 bench.py --config p2 --shard --no-cpu-baseline
run split_p2 PLC_NO_POST_FUSION=1 python3 bench.py --config p2 --shard --no-cpu-baseline
