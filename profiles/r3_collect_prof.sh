#!/bin/bash
# Kernel statistics and PMC traffic of this round's record (rocprofv3; every pass a run of its own).
# Results land in gpurun_out/ (profiles/r3_copy.sh copies the summaries into profiles/).
export TMPDIR=/tmp
set -e
bash profiles/prof_run.sh r3_train_c3 --steps 10 --warmup 3
bash profiles/prof_run.sh r3_train_c3_shard --shard --steps 10 --warmup 3
bash profiles/prof_run.sh r3_fwd_c3 --mode fwd --steps 20 --warmup 3
bash profiles/prof_run.sh r3_fwd_c3_shard --mode fwd --shard --steps 20 --warmup 3
bash profiles/prof_run.sh r3_train_p2 --config p2 --steps 10 --warmup 3
echo stats done
# traffic: the default step (B = 32) and the shard (B = 4), FETCH_SIZE and WRITE_SIZE in separate passes
for m in full shard; do
    extra=""; [ $m = shard ] && extra="--shard"
    bash profiles/pmc_pass.sh train_${m}_fetch FETCH_SIZE $extra
    bash profiles/pmc_pass.sh train_${m}_write WRITE_SIZE $extra
    bash profiles/pmc_pass.sh fwd_${m}_fetch FETCH_SIZE --mode fwd $extra
    bash profiles/pmc_pass.sh fwd_${m}_write WRITE_SIZE --mode fwd $extra
    echo pmc $m done
done
cmd="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
python3 profiles/step_traffic.py gpurun_out/pmc_train_full_fetch gpurun_out/pmc_train_full_write gpurun_out/r3_train_c3_step_traffic.json 32 > gpurun_out/r3_step_traffic.txt
python3 profiles/step_traffic.py gpurun_out/pmc_train_shard_fetch gpurun_out/pmc_train_shard_write gpurun_out/r3_train_c3_shard_step_traffic.json 4 > gpurun_out/r3_shard_step_traffic.txt
rm -f gpurun_out/r3_pointwise_bwd_pair_traffic.json gpurun_out/r3_cell2_traffic.json
px=$((256 * 256))
python3 profiles/kernel_traffic.py gpurun_out/pmc_train_full_fetch gpurun_out/pmc_train_full_write plc_lstm_pointwise_bwd_pair_kernel 16 $((224 * 16 * 6 * px)) gpurun_out/r3_pointwise_bwd_pair_traffic.json "$cmd"
python3 profiles/kernel_traffic.py gpurun_out/pmc_train_shard_fetch gpurun_out/pmc_train_shard_write plc_lstm_pointwise_bwd_pair_kernel 2 $((224 * 2 * 6 * px)) gpurun_out/r3_pointwise_bwd_pair_traffic.json "$cmd --shard"
python3 profiles/kernel_traffic.py gpurun_out/pmc_fwd_full_fetch gpurun_out/pmc_fwd_full_write "plc_conv_mfma_kernel<2, 4, 4, 3, 0, 0" 16 $((416 * 16 * px)) gpurun_out/r3_cell2_traffic.json "$cmd --mode fwd"
python3 profiles/kernel_traffic.py gpurun_out/pmc_fwd_shard_fetch gpurun_out/pmc_fwd_shard_write "plc_conv_mfma_kernel<2, 4, 4, 3, 0, 0" 2 $((416 * 2 * px)) gpurun_out/r3_cell2_traffic.json "$cmd --mode fwd --shard"
python3 profiles/traffic_table.py gpurun_out/pmc_train_shard_fetch gpurun_out/pmc_train_shard_write gpurun_out/r3_train_c3_shard_stats.csv gpurun_out/r3_train_c3_shard_kernel_traffic.csv > gpurun_out/r3_shard_traffic.txt
python3 profiles/traffic_table.py gpurun_out/pmc_train_full_fetch gpurun_out/pmc_train_full_write gpurun_out/r3_train_c3_stats.csv gpurun_out/r3_train_c3_kernel_traffic.csv > gpurun_out/r3_full_traffic.txt
echo collected
# the same shard step with every kernel alone on the GPU (no half-batch chains, no side stream): per-kernel
# durations that do not overlap, for the kernel table
PLC_NO_BWD_SPLIT=1 PLC_NO_FWD_SPLIT=1 PLC_NO_SIDE_STREAM=1 bash profiles/prof_run.sh r3_train_c3_shard_single_stream --shard --steps 10 --warmup 3
PLC_NO_BWD_SPLIT=1 PLC_NO_FWD_SPLIT=1 PLC_NO_SIDE_STREAM=1 bash profiles/prof_run.sh r3_fwd_c3_shard_single_stream --shard --mode fwd --steps 20 --warmup 3
# ... and the default (global batch 32) step the same way: here the pointwise pair launch of grid 6 291 456 (16 images) is
# only bench.py's own roofline timing (the single-stream step launches it on the whole batch, grid 12 582 912), so that
# row of the by-grid table is the stand-alone duration the bench line's roofline object reports
PLC_NO_BWD_SPLIT=1 PLC_NO_FWD_SPLIT=1 PLC_NO_SIDE_STREAM=1 bash profiles/prof_run.sh r3_train_c3_single_stream --steps 10 --warmup 3
# the roofline kernels' stand-alone launches and nothing else under their names: these averages are what the bench
# lines' roofline objects report (launch_us)
bash profiles/prof_run.sh r3_roofline_launches --roofline-only
bash profiles/prof_run.sh r3_roofline_launches_shard --roofline-only --shard
echo single-stream stats done
# what travels back from the GPU box is limited to 64 MiB: drop the raw traces, keep the summaries
rm -rf gpurun_out/prof_r3_* gpurun_out/pmc_train_* gpurun_out/pmc_fwd_*
echo cleaned
