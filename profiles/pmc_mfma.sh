#!/bin/bash
# MFMA-utilisation counters of the forward bench (one --pmc pass, kernel trace only)
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_mfma && mkdir -p gpurun_out/pmc_mfma
timeout -s KILL 180 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o pmc -- python3 bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_mfma.log 2>&1
ls -R gpurun_out/pmc_mfma | head
