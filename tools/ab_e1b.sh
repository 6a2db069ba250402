#!/bin/bash
# GPU box: the E1B (16368-lag) correlator, 512-thread radix-8 form against the 256-thread four-accumulator kernel.
export TMPDIR=/tmp
python3 -m pytest tests/test_acq_gpu.py tests/test_acq10_gpu.py tests/test_ref_pins_gpu.py tests/test_example_gpu.py tests/test_golden_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
for v in 1 0 1 0; do
  echo "== KIWIGPU_ACQ_E1B8=$v"
  KIWIGPU_ACQ_E1B8=$v python3 tools/time_e1b.py 32 2>&1 | tail -2
done
for v in 1 0; do
  for wl in acq59 acq10ms; do
    extra="--steps 40"; [ $wl = acq10ms ] && extra="--steps 10 --warmup 2"
    KIWIGPU_ACQ_E1B8=$v python3 bench.py --workload $wl --no-cpu --no-live-traffic $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('E1B8=$v $wl value %.1f ms/step %.4f kernel_ms %.4f frac %.4f found %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], d['found_svs']))"
  done
done
