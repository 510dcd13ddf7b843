#!/bin/bash
# A/B helper for one gpurun call: tools/ab_env.sh "<name>:<ENV=1 ...>:<bench flags>" ... -> one line per variant
for v in "$@"; do
  n=${v%%:*}; r=${v#*:}; e=${r%%:*}; f=${r#*:}
  env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline $f 2>gpurun_out/ab_$n.err | tail -1 > gpurun_out/ab_$n.json
  echo "$n [$e] [$f] $(python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["ms_per_step"], d["value"], "issue", d["config"].get("host_issue_ms_per_step"))' gpurun_out/ab_$n.json 2>&1 | tail -1)"
done
