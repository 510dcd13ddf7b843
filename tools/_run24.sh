mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu --timeout 600 -k "row_streaming or linear or gru or tatt_train or tsrn_train" 2>&1 | tail -15 > gpurun_out/t_new.log
timeout 100 python tools/bench_kernels.py --match linear > gpurun_out/kbench_lin.log 2>&1
TATT_ROWGEMM=0 timeout 100 python tools/bench_kernels.py --match linear > gpurun_out/kbench_lin_old.log 2>&1
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_rg.json 2> gpurun_out/bench_rg.err
