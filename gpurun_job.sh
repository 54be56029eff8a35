timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-post-pass --no-simil 2>&1 | tail -1 > gpurun_out/bench_fm.json
