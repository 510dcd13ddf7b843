mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu --timeout 600 -k "tbsrn or self_att or cat_pos or weight_stationary" 2>&1 | tail -15 > gpurun_out/t_new.log
timeout 200 python tools/bench_kernels.py --match conv3_fwd_ws > gpurun_out/kbench_conv3.log 2>&1
timeout 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof6 -o r6 -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof6.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python $R/tools/bench_kernels.py --only conv3_fwd_ws_64_64 --iters 5 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- python $R/tools/bench_kernels.py --only conv3_fwd_ws_64_64 --iters 5 > $R/gpurun_out/pmc_write.log 2>&1
