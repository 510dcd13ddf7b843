#!/bin/bash
# GPU box: rocprofv3 --pmc passes (counters only, one group per pass) for single kernels of tools/bench_kernels.py.
# usage: tools/pmc_collect.sh <tag> <bench_kernels name> [<batch> ["--H 32 --W 128"]]      -> gpurun_out/<tag>_pmc_<name>/pass*/...csv + summary
tag=$1; name=$2; batch=${3:-48}; extra=${4:-}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_${name}
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --only $name --iters 5 --batch $batch $extra > $out/pass$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out > gpurun_out/${tag}_pmc_${name}.txt 2>&1
cat gpurun_out/${tag}_pmc_${name}.txt
