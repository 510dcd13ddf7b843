#!/bin/bash
# GPU box: two rocprofv3 --pmc passes (wave-cycle breakdown, LDS) for single kernels of tools/bench_kernels.py.
# usage: tools/pmc_quick.sh <tag> <name> [<name> ...]     -> gpurun_out/<tag>_pmc_<name>.txt
tag=$1; shift
export TMPDIR=/tmp
for name in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_${name}
  mkdir -p $out
  cd /tmp
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --only $name --iters 5 > $out/pass$i.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $out > gpurun_out/${tag}_pmc_${name}.txt 2>&1
  rm -rf $out
  cat gpurun_out/${tag}_pmc_${name}.txt
done
