#!/bin/bash
# GPU box: HBM traffic of single kernels of tools/bench_kernels.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes.
# usage: tools/pmc_traffic.sh <tag> <name> [<name> ...]     -> gpurun_out/<tag>_pmc_traffic_<name>.txt
tag=$1; shift
export TMPDIR=/tmp
for name in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_traffic_${name}
  mkdir -p $out
  cd /tmp
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pass$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --only $name --iters 5 > $out/pass$i.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_summary.py $out > gpurun_out/${tag}_pmc_traffic_${name}.txt 2>&1
  rm -rf $out
  grep -A3 -E "wgrad_sb|gru32_bwd2|gru_wgrad_frag|gru32_fwd2" gpurun_out/${tag}_pmc_traffic_${name}.txt | head -12
done
