mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_crnn.py -q -m gpu --timeout 300 -k "sr_loss" 2>&1 | grep -E "Error|FAILED|passed|failed|assert" | head -20 > gpurun_out/t_crnn.log
timeout 300 python bench.py --arch tatt_tpg > gpurun_out/bench_tpg.json 2> gpurun_out/bench_tpg.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
