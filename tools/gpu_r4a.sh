#!/bin/bash
# round-4 call A: validate what was written blind (split-bf16 conv3 weight gradient, query-GRU row-sum bias, second-generation BiGRU
# recurrences, fragment-stream weight gradients), then micro-benchmarks and a same-box A/B of the training step.
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -x -k "conv3_wgrad_split or gru32_v2 or gru_wgrad_frag or query_gru or gru_block or bigru32 or gru_wgrad_split" 2>&1 | tail -25 > gpurun_out/r4a_tests.log
tail -8 gpurun_out/r4a_tests.log
timeout 300 python tools/bench_kernels.py --match gru > gpurun_out/r4a_ubench_gru.txt 2>&1; cat gpurun_out/r4a_ubench_gru.txt | grep -v amdgpu.ids
timeout 200 python tools/bench_kernels.py --match conv3_wgrad > gpurun_out/r4a_ubench_conv3w.txt 2>&1; cat gpurun_out/r4a_ubench_conv3w.txt | grep -v amdgpu.ids
tools/ab_hooks.sh 2 "tatt_amd.ops.GRU32_V2=1" "tatt_amd.ops.GRU32_V2=0" "tatt_amd.functional.GRU_WGRAD_FRAG=0" "tatt_amd.ops.CONV3_WGRAD_SB=0" "tatt_amd.ops.CONV3_WGRAD_SB=0 tatt_amd.ops.GRU32_V2=0" 2>&1 | tee gpurun_out/r4a_ab.txt
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout=600 -x -k "fp64 or b48 or golden or replay" 2>&1 | tail -8 > gpurun_out/r4a_model_tests.log
tail -5 gpurun_out/r4a_model_tests.log
