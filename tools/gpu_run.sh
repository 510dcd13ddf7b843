#!/bin/bash
# GPU-box driver script for one gpurun call: tests, benches, one rocprofv3 kernel trace.  Everything lands in gpurun_out/<tag>_*.
# usage: tools/gpu_run.sh <tag> [pytest -k expression | all | none] [bench configs...]
tag=${1:-r2a}; ksel=${2:-all}; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$ksel" != "none" ]; then
  if [ "$ksel" == "all" ]; then kflag=(); else kflag=(-k "$ksel"); fi
  python -m pytest tests -m gpu -q -x --timeout=900 "${kflag[@]}" > gpurun_out/${tag}_tests.log 2>&1
  echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
  tail -5 gpurun_out/${tag}_tests.log
fi
cfgs=("$@")
if [ ${#cfgs[@]} -eq 0 ]; then cfgs=("std:" "nodefer:--no-defer" "large:--tile large" "dp1:--dp-selftest --no-cpu-baseline" "tsrn:--arch tsrn --no-cpu-baseline" "tbsrn:--arch tbsrn --no-cpu-baseline"); fi
for cfg in "${cfgs[@]}"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  [ "$name" == "skip" ] && continue
  timeout 600 python bench.py --steps 20 --warmup 5 $flags > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err
  echo "bench $name rc=$? $(cut -c1-200 gpurun_out/${tag}_bench_${name}.json)"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
python tools/prof_summary.py $db 15 > gpurun_out/${tag}_kernel_stats.txt 2>&1
head -12 gpurun_out/${tag}_kernel_stats.txt
