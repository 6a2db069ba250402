#!/bin/bash
# GPU box: A/B of library variants on chosen workloads.  usage: tools/ab_lib.sh "<workloads>" variant...   (base = the product build)
export TMPDIR=/tmp
wls=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
    for wl in $wls; do
      extra="--steps 200"; [ $wl = acq10ms ] && extra="--steps 20 --warmup 3"; case $wl in receivers*) extra="--steps 60 --warmup 4";; esac
      KIWIGPU_LIBRARY=$lib timeout 300 python3 bench.py --full-line --workload $wl --no-cpu --no-live-traffic $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['step_ms_spread']; print('%-8s %-10s value %.1f ms/step %.4f (min %.4f med %.4f max %.4f) kernel_ms %.4f frac %.4f' % ('$v', '$wl', d['value'], d['ms_per_step'], s['min'], s['median'], s['max'], r['kernel_ms'], r['frac']))"
    done
  done
done
