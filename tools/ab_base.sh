#!/bin/bash
# Same-box A/B against the baseline checkout in _base/ (git worktree of an earlier commit, built in place): box-to-box and
# thermal variation between gpurun calls is ~5-15 %, larger than most single optimisations, so only interleaved runs inside one
# call are comparable.   usage: tools/ab_base.sh [rounds] [bench flags...]
# setup (once, in the build container): git worktree add _base <baseline commit> && (cd _base && python -m tatt_amd.build)
# (_base/ is git-ignored but travels to the GPU box with the snapshot; `git worktree remove --force _base` when done)
rounds=${1:-2}; shift
root=$(pwd)
mkdir -p gpurun_out
for i in $(seq 1 $rounds); do
  for side in base new; do
    if [ $side == base ]; then dir=$root/_base; else dir=$root; fi
    ( cd $dir && timeout 240 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$root/gpurun_out/ab_$side.err | tail -1 > $root/gpurun_out/ab_$side.json )
    echo "$side $i [$*] $(python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["ms_per_step"], d["value"], "kernel_ms", d["roofline"]["kernel_ms"])' gpurun_out/ab_$side.json 2>&1 | tail -1)"
  done
done
