#!/bin/bash
# same-box A/B of the step's TAIL: GPU time of the three groups of passes (DP self-test: one graph per group) in _base/ and here
rounds=${1:-2}
root=$(pwd)
for i in $(seq 1 $rounds); do
  for side in base new; do
    if [ $side == base ]; then dir=$root/_base; else dir=$root; fi
    ( cd $dir && cp $root/bench.py bench.py 2>/dev/null; cp $root/tatt_amd/train.py tatt_amd/train.py 2>/dev/null; timeout 240 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --dp-selftest 2>/dev/null | tail -1 > $root/gpurun_out/abt_$side.json )
    echo "$side $i $(python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["ms_per_step"], [g["gpu_ms"] for g in d["collectives"]["pass_groups"]])' gpurun_out/abt_$side.json 2>&1 | tail -1)"
  done
done
