#!/bin/bash
# A/B runs inside ONE gpurun call: tools/ab_r06.sh "<module.attr=val | none>" ... ; env hooks: TATT_QGRU_NARROW=0/1 via tools/ab_hooks
cd $GRAFT_REPO_ROOT
A="--steps 40 --warmup 10 --no-cpu-baseline --sustain 0 --no-exact-fp32"
for i in 1 2; do
for kv in "$@"; do
  echo "$kv $(python tools/ab_bench.py $kv -- $A 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
done; done
