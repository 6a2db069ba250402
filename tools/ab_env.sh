#!/bin/bash
# GPU box: A/B of environment switches on one workload, twice each.  usage: tools/ab_env.sh "<bench args>" "VAR=1 VAR2=x" "-" ...   ("-" = no extra variables)
export TMPDIR=/tmp
args=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    vars=$cfg; [ "$cfg" = "-" ] && vars="KIWIGPU_NOP=1"
    env $vars timeout 300 python3 bench.py --full-line --no-cpu --no-live-traffic $args 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['step_ms_spread']; print('%-44s ms/step %.4f (min %.4f med %.4f) kernel_ms %.4f value %.1f' % ('$cfg', d['ms_per_step'], s['min'], s['median'], r['kernel_ms'], d['value']))"
  done
done
