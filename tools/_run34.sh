mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_crnn.py tests/test_kernels_gpu.py -q -m gpu --timeout 300 -k "crnn or conv or maxpool or stn or bilstm or sr_loss" 2>&1 | grep -E "Error|FAILED|passed|failed|assert" | head -30 > gpurun_out/t_crnn.log
timeout 300 python bench.py --arch tatt_tpg > gpurun_out/bench_tpg.json 2> gpurun_out/bench_tpg.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof8
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof8 -o r8 -- python $GRAFT_REPO_ROOT/bench.py --arch tatt_tpg --steps 10 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof8.log 2>&1
