export TMPDIR=/tmp
rm -rf gpurun_out/prof32 && mkdir -p gpurun_out/prof32
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof32 -o train -- python3 bench.py --global-batch 32 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof32.log 2>&1 || exit 1
python profiles/export_stats.py gpurun_out/prof32/train_results.db gpurun_out/prof32_stats.csv > gpurun_out/prof32_stats.txt
python profiles/by_grid.py gpurun_out/prof32/train_results.db gpurun_out/prof32_by_grid.csv >> gpurun_out/prof32_stats.txt 2>&1
rm -rf gpurun_out/prof32
timeout -k 10 200 python3 bench.py --global-batch 32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench32.json 2>gpurun_out/bench32.err || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench4.json 2>gpurun_out/bench4.err || exit 1
