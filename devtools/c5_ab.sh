set -e
for gd in fp8 bf16; do for mode in fwd train; do
  timeout -k 10 120 python bench.py --config c5 --gate-dtype $gd --mode $mode --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline_mfma', d['roofline']); print('$gd $mode', round(d['ms_per_step'],4), 'ms/step', d['dtype'], 'cell2 us', round(r['launch_us'],2), 'frac', round(r['frac'],3))"
done; done
