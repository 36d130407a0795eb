#!/bin/bash
# frames/s of the c3 geometry (256x256, T=10, scale 1) against the batch on one GPU
for B in 2 4 8 16 32; do
  timeout -k 10 200 python3 bench.py --shape $B,10,256,256,1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sweep_$B.json 2>gpurun_out/sweep_$B.err || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/sweep_$B.json')); print('B=$B train %.1f frames/s  %.3f ms/step  fwd %.1f frames/s' % (d['value'], d['ms_per_step'], d['fwd_frames_per_s']))"
done
