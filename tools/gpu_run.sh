#!/bin/bash
# GPU-box driver script for one gpurun call: tests, benches, one rocprofv3 kernel trace.  Everything lands in gpurun_out/<tag>_*.
tag=${1:-r2a}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
for cfg in "std:" "noside:--no-side-stream" "large:--tile large" "dp1:--dp-selftest --no-cpu-baseline" "tsrn:--arch tsrn --no-cpu-baseline" "tbsrn:--arch tbsrn --no-cpu-baseline"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  timeout 600 python bench.py --steps 20 --warmup 5 $flags > gpurun_out/${tag}_bench_${name}.json 2> gpurun_out/${tag}_bench_${name}.err
  echo "bench $name rc=$? $(cut -c1-220 gpurun_out/${tag}_bench_${name}.json)"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
python tools/prof_summary.py $db 15 > gpurun_out/${tag}_kernel_stats.txt 2>&1
head -30 gpurun_out/${tag}_kernel_stats.txt
