#!/bin/bash
# GPU box: kernel-trace durations of the generations of the split-bf16 3x3 kernels (tools/bench_kernels.py --match conv3_) and the SQ counter
# passes the round-5 verdict asked for, per bench entry.  usage: tools/conv3_prof.sh <tag> [bench entry ...]   -> gpurun_out/<tag>_*
tag=$1; shift
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/${tag}_kt -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --match conv3_ --iters 20 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/kt_by_grid.py /tmp/${tag}_kt conv3 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_trace.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_trace.txt
for name in "$@"; do
  out=/tmp/${tag}_pmc_${name}
  mkdir -p $out
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --only $name --iters 5 > $out/pass$i.log 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out 2>&1 | awk '/conv3_c64/{f=1} /^[a-z_A-Z]/ && !/conv3_c64/{f=0} f' > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_${name}.txt
  cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_${name}.txt
done
