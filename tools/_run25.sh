mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu --timeout 600 -k "row_streaming or weight_stationary or conv" 2>&1 | tail -8 > gpurun_out/t_new.log
for v in 1 2; do echo "WS_VARIANT=$v"; TATT_CONV3_WS_VARIANT=$v timeout 100 python tools/bench_kernels.py --match conv3_fwd_ws; done > gpurun_out/kbench_ws12.log 2>&1
timeout 100 python tools/bench_kernels.py --match linear > gpurun_out/kbench_lin.log 2>&1
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
TATT_ROWGEMM=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
TATT_CONV3_WS_VARIANT=1 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
timeout 200 python tools/aten_trace.py > gpurun_out/aten_trace.log 2>&1
