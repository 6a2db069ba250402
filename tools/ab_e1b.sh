#!/bin/bash
# GPU box: the 16368-lag (E1B) correlator, product build against libkiwigpu_old.so (the previous kernel), ABAB on acq59 / acq10ms,
# after the acquisition tests of the product build.
export TMPDIR=/tmp
python3 -m pytest tests/test_acq_gpu.py tests/test_acq10_gpu.py tests/test_ref_pins_gpu.py tests/test_example_gpu.py tests/test_golden_gpu.py tests/test_lifecycle_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for rep in 1 2; do
  for v in base old; do
    if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
    for wl in acq59 acq10ms; do
      extra="--steps 100"; [ $wl = acq10ms ] && extra="--steps 20 --warmup 3"
      KIWIGPU_LIBRARY=$lib python3 bench.py --full-line --workload $wl --no-cpu --no-live-traffic $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-5s %-8s value %.1f ms/step %.4f kernel_ms %.4f (min %.4f) frac %.4f' % ('$v', '$wl', d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['frac']))"
    done
  done
done
