#!/bin/bash
# same-box A/B of module-level hooks for any bench configuration: tools/ab_hooks_arch.sh "<bench flags>" "<mod.attr=val | none>" ...
cd $GRAFT_REPO_ROOT
flags=$1; shift
A="--steps 40 --warmup 10 --no-cpu-baseline --sustain 0 --no-exact-fp32 $flags"
for i in 1 2 3; do
for kv in "$@"; do
  if [ "$kv" == "none" ]; then k=""; else k=$kv; fi
  echo "$kv $(python tools/ab_bench.py $k -- $A 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
done; done
