set -e
for cfg in c3 c2 c4; do for mode in fwd train; do for np in 0 1; do
  echo -n "$cfg $mode no_pairing=$np: "
  PLC_NO_PAIRING=$np timeout -k 10 120 python bench.py --config $cfg --mode $mode --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4),'ms/step')"
done; done; done
