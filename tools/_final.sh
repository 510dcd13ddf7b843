mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -3 > gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
timeout 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
timeout 300 python bench.py --arch tsrn --no-cpu-baseline > gpurun_out/bench_tsrn.json 2> gpurun_out/bench_tsrn.err
timeout 300 python bench.py --arch tbsrn --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_tbsrn.json 2> gpurun_out/bench_tbsrn.err
timeout 300 python bench.py --arch tatt_tpg > gpurun_out/bench_tpg.json 2> gpurun_out/bench_tpg.err
timeout 100 python tools/bench_kernels.py > gpurun_out/kbench_all.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof9 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof9 -o r9 -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof9.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python $R/tools/bench_kernels.py --only conv3_fwd_ws_64_64 --iters 5 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- python $R/tools/bench_kernels.py --only conv3_fwd_ws_64_64 --iters 5 > $R/gpurun_out/pmc_write.log 2>&1
