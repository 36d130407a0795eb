#!/bin/bash
# Counters of the library's kernels in the train bench: separate --pmc passes (kernel trace only),
# each within the per-block counter limits.  Summaries -> gpurun_out/pmc_train_<tag>.txt
#   bash profiles/pmc_train.sh            all passes
#   bash profiles/pmc_train.sh fetch write  only the named passes
export TMPDIR=/tmp
run() {   # run <tag> <counters...>
    tag=$1; shift
    rm -rf gpurun_out/pmc_train_$tag && mkdir -p gpurun_out/pmc_train_$tag
    timeout -s KILL 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_train_$tag -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_train_$tag.log 2>&1 || return 1
    python3 profiles/pmc_summary.py gpurun_out/pmc_train_$tag plc_ > gpurun_out/pmc_train_$tag.txt
}
passes=${@:-fetch write tcc sq}
for p in $passes; do
    case $p in
        fetch) run fetch FETCH_SIZE || exit 1 ;;
        write) run write WRITE_SIZE || exit 1 ;;
        tcc)   run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum || exit 1 ;;
        sq)    run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1 ;;
        *) echo "unknown pass $p"; exit 2 ;;
    esac
done
