#!/bin/bash
# A/B on the GPU box: ddc14 and cfg2_chain in line (KIWIGPU_BENCH_DDC_DEFERRED=0) and in the streaming form (=1), twice each.
export TMPDIR=/tmp
for rep in 1 2; do
  for mode in 0 1; do
    for wl in ddc14 cfg2_chain; do
      KIWIGPU_BENCH_HOST_SPLIT=${HOST_SPLIT:-0} KIWIGPU_BENCH_DDC_DEFERRED=$mode timeout 300 python3 bench.py --full-line --no-cpu --no-live-traffic --workload $wl --steps 200 2> gpurun_out/ab_ddc_$wl.$mode.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['step_ms_spread']
print('%-10s deferred=$mode  step %.4f ms (min %.4f med %.4f max %.4f)  kernel_ms %.4f  ddc_alone %s frames_alone %s' % ('$wl', d['ms_per_step'], s['min'], s['median'], s['max'], r['kernel_ms'], r.get('ddc_ms_alone'), r.get('frames_ms_alone')))"
      grep -E "host enqueue|host split|repeat" gpurun_out/ab_ddc_$wl.$mode.err | sed 's/^/    /'
    done
  done
done
