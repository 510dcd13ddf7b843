#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/stn_head_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/r4u_stn_head_bench.txt
cd /tmp
rm -rf /tmp/prof_stn
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stn -o p -- python $GRAFT_REPO_ROOT/tools/stn_head_bench.py > /tmp/prof_stn.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/prof_stn -name "*.db" | head -1)
echo "db: $db"; tail -3 /tmp/prof_stn.log
python tools/prof_summary.py $db 1 2>&1 | cut -c1-200 | head -60 | tee $out/r4u_stn_kernel_stats.txt
