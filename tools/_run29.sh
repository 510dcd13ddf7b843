mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
timeout 300 python bench.py --arch tbsrn --steps 10 --warmup 3 > gpurun_out/bench_tbsrn.json 2> gpurun_out/bench_tbsrn.err
timeout 300 python bench.py --arch tsrn > gpurun_out/bench_tsrn.json 2> gpurun_out/bench_tsrn.err
timeout 100 python tools/bench_kernels.py > gpurun_out/kbench_all.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof7 $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof7 -o r7 -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof7.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- python $R/tools/bench_kernels.py --only conv3_fwd_ws_64_64 --iters 5 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- python $R/tools/bench_kernels.py --only conv3_fwd_ws_64_64 --iters 5 > $R/gpurun_out/pmc_write.log 2>&1
