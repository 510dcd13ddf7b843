mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -x 2>&1 | tail -15 > gpurun_out/t_all.log
for b in 16 48 96 192; do timeout 100 python tools/bench_kernels.py --only conv3_fwd_ws_64_64 --batch $b; done > gpurun_out/kbench_ws_scan.log 2>&1
timeout 100 python tools/bench_kernels.py --match gru32 > gpurun_out/kbench_gru.log 2>&1
timeout 100 python tools/bench_kernels.py --match conv3_fwd > gpurun_out/kbench_conv3.log 2>&1
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_ws.json 2> gpurun_out/bench_ws.err
