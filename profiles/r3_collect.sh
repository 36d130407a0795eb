#!/bin/bash
# The bench lines of this round's record, one after the other, each under its own time limit; results land in
# gpurun_out/r3_*_bench.json (profiles/r3_copy.sh copies them into profiles/).  Run from the repository root.
export TMPDIR=/tmp
set -e
bench() {   # bench <tag> <args...>
    tag=$1; shift
    timeout -k 10 240 python3 bench.py "$@" > gpurun_out/r3_${tag}_bench.json 2> gpurun_out/r3_${tag}_bench.err
    python3 -c "import json,sys; d=json.load(open('gpurun_out/r3_${tag}_bench.json')); print('${tag}:', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], d['dtype'], 'fwd', round(d['fwd_frames_per_s']))"
}
bench train_c3                                   # the default: BASELINE configs[2], global batch 32 on this one GPU
bench train_c3_shard --shard                     # its per-GPU shard (B = 4): what each of 8 GPUs runs
bench fwd_c3 --mode fwd
bench fwd_c3_shard --mode fwd --shard
bench train_c3_shard_graph --shard --graph --no-cpu-baseline
bench train_c2 --config c2 --shard --no-cpu-baseline
bench train_c4 --config c4 --shard --no-cpu-baseline
bench train_c5_fp8 --config c5 --shard --no-cpu-baseline
bench fwd_c5_fp8 --config c5 --shard --mode fwd --no-cpu-baseline
bench train_c5_bf16 --config c5 --shard --gate-dtype bf16 --no-cpu-baseline
bench fwd_c5_bf16 --config c5 --shard --mode fwd --gate-dtype bf16 --no-cpu-baseline
bench train_p2 --config p2 --no-cpu-baseline
bench fwd_p2 --config p2 --mode fwd --no-cpu-baseline
bench train_p2_graph --config p2 --graph --no-cpu-baseline
bench train_p2_shard --config p2 --shard --no-cpu-baseline
bench fwd_p2_shard --config p2 --shard --mode fwd --no-cpu-baseline
bench train_p2_shard_graph --config p2 --shard --graph --no-cpu-baseline
bench train_c1 --config c1 --no-cpu-baseline
bench fwd_c1 --config c1 --mode fwd --no-cpu-baseline
bench train_c0 --config c0 --no-cpu-baseline
bench train_c0_graph --config c0 --graph --no-cpu-baseline
bench train_c3_wide --shard --hidden 32,64 --steps 10 --warmup 3 --no-cpu-baseline
bench fwd_c3_wide --shard --hidden 32,64 --mode fwd --steps 10 --warmup 3 --no-cpu-baseline
echo collected
