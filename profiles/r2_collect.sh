#!/bin/bash
# Everything measured for this round's record, one step after the other, each under its own time limit;
# results land in gpurun_out/ (copied into profiles/r2_* afterwards).  Run from the repository root.
export TMPDIR=/tmp
set -e
bench() {   # bench <tag> <args...>
    tag=$1; shift
    timeout -k 10 200 python3 bench.py "$@" > gpurun_out/r2_${tag}_bench.json 2> gpurun_out/r2_${tag}_bench.err
    python3 -c "import json,sys; d=json.load(open('gpurun_out/r2_${tag}_bench.json')); print('${tag}:', round(d['ms_per_step'],4), 'ms/step', round(d['value']), d['unit'], d['dtype'])"
}
bench train_c3
bench fwd_c3 --mode fwd
bench train_c3_graph --graph --no-cpu-baseline
bench train_c3_gb32 --global-batch 32 --steps 10 --warmup 3 --no-cpu-baseline
bench train_c2 --config c2 --no-cpu-baseline
bench train_c4 --config c4 --no-cpu-baseline
bench train_c5_fp8 --config c5 --no-cpu-baseline
bench fwd_c5_fp8 --config c5 --mode fwd --no-cpu-baseline
bench train_c5_bf16 --config c5 --gate-dtype bf16 --no-cpu-baseline
bench fwd_c5_bf16 --config c5 --mode fwd --gate-dtype bf16 --no-cpu-baseline
bench train_p2 --config p2 --no-cpu-baseline
bench fwd_p2 --config p2 --mode fwd --no-cpu-baseline
bench train_c1 --config c1 --no-cpu-baseline
bench fwd_c1 --config c1 --mode fwd --no-cpu-baseline
bench train_c0 --config c0 --no-cpu-baseline
bench train_c3_wide --hidden 32,64 --steps 10 --warmup 3 --no-cpu-baseline
bench fwd_c3_wide --hidden 32,64 --mode fwd --steps 10 --warmup 3 --no-cpu-baseline
bash profiles/prof_train.sh > gpurun_out/r2_prof_train.txt
bash profiles/prof_fwd.sh > gpurun_out/r2_prof_fwd.txt
bash profiles/prof_cfg.sh c5 fwd > gpurun_out/r2_prof_c5_fwd.txt
bash profiles/prof_cfg.sh p2 train > gpurun_out/r2_prof_p2_train.txt
bash profiles/prof_cfg.sh c0 train > gpurun_out/r2_prof_c0_train.txt
bash profiles/pmc_train.sh fetch write
python3 profiles/traffic_table.py gpurun_out/pmc_train_fetch gpurun_out/pmc_train_write gpurun_out/train_stats.csv gpurun_out/r2_train_c3_kernel_traffic.csv > gpurun_out/r2_traffic.txt
python3 profiles/pw_traffic.py gpurun_out/pmc_train_fetch gpurun_out/pmc_train_write gpurun_out/r2_pointwise_bwd_traffic.json > /dev/null
python3 profiles/step_traffic.py gpurun_out/pmc_train_fetch gpurun_out/pmc_train_write gpurun_out/r2_train_c3_step_traffic.json > gpurun_out/r2_step_traffic.txt
# the fp8 cell kernel's traffic (c5 forward)
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_c5_$c && mkdir -p gpurun_out/pmc_c5_$c
    timeout -s KILL 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_c5_$c -o pmc -- python3 bench.py --config c5 --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_c5_$c.log 2>&1
    python3 profiles/pmc_summary.py gpurun_out/pmc_c5_$c plc_lstm_fp8 > gpurun_out/r2_pmc_c5_$c.txt
done
echo collected
