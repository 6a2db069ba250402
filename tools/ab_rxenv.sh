#!/bin/bash
# GPU box: receivers under combinations of stream switches and HW queue counts (twice each).
export TMPDIR=/tmp
run() {
  env "$@" python3 bench.py --no-cpu --no-live-traffic --workload receivers --steps 40 --warmup 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['step_ms_spread']; print('   ms/step %.4f  cadence %.3f/%.3f/%.3f' % (d['ms_per_step'], s['min'], s['median'], s['max']))"
}
for cfg in "$@"; do
  echo "== $cfg"; run $cfg; run $cfg
done
