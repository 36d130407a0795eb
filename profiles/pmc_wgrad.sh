#!/bin/bash
# Counters of the weight-gradient kernels in the train bench: three separate --pmc passes (kernel
# trace only), each within the per-block counter limits.  Summaries -> gpurun_out/pmc_wgrad_*.txt
export TMPDIR=/tmp
run() {   # run <tag> <counters...>
    tag=$1; shift
    rm -rf gpurun_out/pmc_wgrad_$tag && mkdir -p gpurun_out/pmc_wgrad_$tag
    timeout -s KILL 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_wgrad_$tag -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_wgrad_$tag.log 2>&1 || return 1
    python3 profiles/pmc_summary.py gpurun_out/pmc_wgrad_$tag plc_wgrad_kernel > gpurun_out/pmc_wgrad_$tag.txt
}
run fetch FETCH_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum &&
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
