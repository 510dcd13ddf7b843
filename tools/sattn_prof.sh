#!/bin/bash
# GPU box: A/B + kernel trace + SQ counter passes of the TBSRN self-attention kernels (tools/sattn_check.py).  usage: tools/sattn_prof.sh [pmc]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout 300 python $R/tools/sattn_check.py --bits 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/sattn_kt; mkdir -p /tmp/sattn_kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sattn_kt -o sa -- python $R/tools/sattn_check.py > /dev/null 2>&1
python $R/tools/kt_by_grid.py /tmp/sattn_kt sattn | grep "groups *8 \|groups *16 \|groups *768 " | tee $R/gpurun_out/sattn_kernel_trace.txt
if [ "$1" = pmc ]; then
  out=/tmp/sattn_pmc; rm -rf $out; mkdir -p $out
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_TRANS" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -o p -- python $R/tools/sattn_check.py --pmc > $out/pass$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $out 2>&1 | awk '/sattn/{f=1} /^[a-z_A-Z]/ && !/sattn/{f=0} f' | tee $R/gpurun_out/sattn_pmc.txt
fi
