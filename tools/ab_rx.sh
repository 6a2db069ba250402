#!/bin/bash
# GPU box: the ddc tests, then receivers with the one-shot capture (default) and round 3's continuous sampler, twice.
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_ddc_gpu.py tests/test_receivers_gpu.py tests/test_chain_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
for rep in 1 2; do
  for mode in 0 1; do
    KIWIGPU_BENCH_RX_CONTINUOUS=$mode timeout 300 python3 bench.py --no-cpu --no-live-traffic --workload receivers --steps 40 --warmup 4 2> gpurun_out/ab_rx.$mode.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['step_ms_spread']
print('receivers continuous=$mode  step %.4f ms (min %.4f med %.4f max %.4f)  x_realtime %.1f  checked %s' % (d['ms_per_step'], s['min'], s['median'], s['max'], d['x_realtime_all_receivers'], {k: v for k, v in d['checked'].items() if k != 'rule'}))"
    tail -3 gpurun_out/ab_rx.$mode.err | grep -v amdgpu.ids
  done
done
