mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_crnn.py -q -m gpu --timeout 300 --durations=8 2>&1 | grep -E "Error|FAILED|passed|failed|assert|s call|s setup" | head -40 > gpurun_out/t_crnn.log
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 300 -k "maxpool or gemm or stn" --durations=8 2>&1 | grep -E "Error|FAILED|passed|failed|s call|s setup" | head -30 > gpurun_out/t_k.log
