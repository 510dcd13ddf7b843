#!/bin/bash
# one gpurun call: selected tests, bench configs, one rocprofv3 kernel trace + per-step timeline.  usage: tools/gpu_quick.sh <tag> "<pytest -k expr|none>" [bench cfgs "name:flags" ...]
tag=$1; ksel=$2; shift; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "$ksel" != "none" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -k "$ksel" 2>&1 | tail -40 > gpurun_out/${tag}_tests.log
  tail -12 gpurun_out/${tag}_tests.log
fi
for cfg in "$@"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  if [ "$name" == "prof" ]; then
    cd /tmp
    timeout 240 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --sustain 0 --no-exact-fp32 $flags > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1
    cd $GRAFT_REPO_ROOT
    db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
    python tools/prof_summary.py $db 15 > gpurun_out/${tag}_kernel_stats.txt 2>&1
    python tools/prof_timeline.py $db > gpurun_out/${tag}_timeline.txt 2>&1
    python tools/prof_timeline.py $db --dump --json gpurun_out/${tag}_step_kernel_table.json > gpurun_out/${tag}_timeline_dump.txt 2>&1
    rm -rf gpurun_out/${tag}_prof
    head -40 gpurun_out/${tag}_timeline.txt
    continue
  fi
  timeout 300 python bench.py --steps 30 --warmup 10 $flags > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err
  echo "bench $name rc=$? $(cut -c1-400 gpurun_out/${tag}_bench_${name}.json)"
done
