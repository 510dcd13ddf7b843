mkdir -p gpurun_out
(nproc; cat /sys/fs/cgroup/cpu.max; python -c "import os;print(os.cpu_count(), len(os.sched_getaffinity(0)))"; free -g | head -2) > gpurun_out/cpuinfo.log 2>&1
timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "tbsrn or self_att or cat_pos or weight_stationary or layer_norm or ln or conv" 2>&1 | tail -25 > gpurun_out/t_new.log
timeout 200 python tools/bench_kernels.py --match conv3 > gpurun_out/kbench_conv3.log 2>&1
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_ws.json 2> gpurun_out/bench_ws.err
TATT_CONV3_WS=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_nows.json 2> gpurun_out/bench_nows.err
timeout 280 python bench.py --cpu-baseline-only --arch tatt --cpu-batch 48 > gpurun_out/cpu_base.json 2> gpurun_out/cpu_base.err
timeout 300 python bench.py --arch tbsrn --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/bench_tbsrn.json 2> gpurun_out/bench_tbsrn.err
