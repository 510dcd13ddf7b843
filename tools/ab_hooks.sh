#!/bin/bash
# Same-box A/B of module-level hooks: tools/ab_hooks.sh <rounds> "<mod.attr=val[ mod.attr=val]>" ...   (one bench.py run per config and round)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    echo "$cfg $(timeout 200 python tools/ab_bench.py $cfg -- --steps 30 --warmup 5 --no-cpu-baseline --sustain 0 --no-exact-fp32 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>&1 | tail -1)"
  done
done
