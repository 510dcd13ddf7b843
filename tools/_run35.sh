mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 --durations=6 2>&1 | grep -E "Error|FAILED|passed|failed|assert|s call" | head -30 > gpurun_out/t_all.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
