#!/bin/bash
# Does the power-of-two plane stride of a 256 x 256 image (H * W * 16 bytes = 1 MiB between the channel
# blocks of a tensor) cost bandwidth?  Train-step time per megapixel-frame at nearby heights.
set -e
mkdir -p gpurun_out
for shp in 4,10,256,256,1 4,10,255,256,1 4,10,250,256,1 4,10,264,256,1 4,10,256,288,1 4,10,256,224,1; do
    timeout -k 10 200 python3 bench.py --shape $shp --no-cpu-baseline > gpurun_out/stride_$shp.json
    python3 -c "
import json; d=json.load(open('gpurun_out/stride_$shp.json')); B,T,H,W,s=[int(v) for v in '$shp'.split(',')]
print('$shp', round(d['ms_per_step'],4), 'ms/step', round(d['ms_per_step']*1e3/(B*T*H*W/1e6),2), 'us per Mpx-frame')"
done
