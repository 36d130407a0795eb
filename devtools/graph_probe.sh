#!/bin/bash
# One pass of the graph-replay probe under the library's kernel-selection switches (devtools/graph_replay_probe.py).
set -u
out=gpurun_out/graph_probe
mkdir -p $out
run() {  # name, env..., then mode, when
    name=$1; shift
    echo "=== $name" | tee -a $out/summary.txt
    timeout -k 10 150 env "$@" > $out/$name.log 2>&1
    rc=$?
    echo "rc=$rc" >> $out/summary.txt
    grep -E "first deviating|^step|differ|^--" $out/$name.log | head -40 >> $out/summary.txt
    return $rc
}
run base        X=1 python3 -u devtools/graph_replay_probe.py rd 0 || exit 1
run every       X=1 python3 -u devtools/graph_replay_probe.py rd every || exit 1
run nopersist   PLC_NO_PERSISTENT=1 python3 -u devtools/graph_replay_probe.py rd 0 || exit 1
run nopair      PLC_NO_PAIRING=1 python3 -u devtools/graph_replay_probe.py rd 0 || exit 1
run noroll      PLC_NO_ROLLING=1 python3 -u devtools/graph_replay_probe.py rd 0 || exit 1
run nosdma      HSA_ENABLE_SDMA=0 python3 -u devtools/graph_replay_probe.py rd 0 || exit 1
cat $out/summary.txt
