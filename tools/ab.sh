#!/bin/bash
# A/B timing of library variants on the GPU box: tools/ab.sh "<bench args>" variant1 variant2 ...
# ("base" = the product build).  Prints the acq kernel time and step time of each, twice (ABAB).
args=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
    KIWIGPU_LIBRARY=$lib timeout 300 python3 bench.py --full-line --no-cpu --no-live-traffic $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-14s kernel %.5f ms  step %.5f ms  value %.1f  frac %.4f' % ('$v', r['kernel_ms'], d['ms_per_step'], d['value'], r['frac']))"
  done
done
