#!/bin/bash
# A/B of environment switches on the GPU box: tools/ab_env.sh "<bench args>" "VAR=val [VAR=val]" ...   ("-" = none)
args=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e timeout 300 python3 bench.py --no-cpu --no-live-traffic $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-60s kernel %.5f ms  step %.5f ms' % ('$v', r['kernel_ms'], d['ms_per_step']))"
  done
done
