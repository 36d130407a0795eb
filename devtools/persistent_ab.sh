set -e
for cfg in p2 c1; do
  for mode in train fwd; do
    for np in 0 1; do
      echo "== $cfg $mode no_persistent=$np"
      PLC_NO_PERSISTENT=$np timeout -k 10 120 python bench.py --config $cfg --mode $mode --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step', round(d['value']),'frames/s', 'fwd', round(d['fwd_frames_per_s']))"
    done
  done
done
