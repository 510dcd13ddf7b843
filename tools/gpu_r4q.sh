#!/bin/bash
# round 4, persistent query-GRU chains: parity + load tests, micro-benchmarks, same-box A/B of the step and of its pass groups
cd $GRAFT_REPO_ROOT
out=gpurun_out
F=tatt_amd.functional
(timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "query_gru" 2>&1 | tail -15) > $out/r4q_tests.txt
(timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "graph_replay or b48_parity or drift or tatt_train_step" -s 2>&1 | tail -25) >> $out/r4q_tests.txt
(timeout 300 python tools/bench_kernels.py --match qgru 2>&1 | grep -v amdgpu.ids) > $out/r4q_ubench.txt
bash tools/ab_hooks.sh 2 "$F.QGRU_CHAIN_FWD=0 $F.QGRU_CHAIN_BWD=0" "$F.QGRU_CHAIN_FWD=0 $F.QGRU_CHAIN_BWD=1" "$F.QGRU_CHAIN_FWD=1 $F.QGRU_CHAIN_BWD=0" "$F.QGRU_CHAIN_FWD=1 $F.QGRU_CHAIN_BWD=1" > $out/r4q_ab.txt 2>&1
for cfg in "$F.QGRU_CHAIN_FWD=0 $F.QGRU_CHAIN_BWD=0" "$F.QGRU_CHAIN_FWD=1 $F.QGRU_CHAIN_BWD=1"; do
  echo "$cfg $(timeout 200 python tools/ab_bench.py $cfg -- --steps 30 --warmup 5 --no-cpu-baseline --dp-selftest 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], [g["gpu_ms"] for g in d["collectives"]["pass_groups"]])' 2>&1 | tail -1)" >> $out/r4q_ab.txt
done
cat $out/r4q_tests.txt $out/r4q_ubench.txt $out/r4q_ab.txt
