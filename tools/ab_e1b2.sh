#!/bin/bash
# GPU box: variants of the 16368-lag correlator's cell end (libkiwigpu_<v>.so), ABAB on acq59 / acq10ms; then the FFT microbenchmark.
export TMPDIR=/tmp
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
    for wl in acq59 acq10ms; do
      extra="--steps 100"; [ $wl = acq10ms ] && extra="--steps 20 --warmup 3"
      KIWIGPU_LIBRARY=$lib python3 bench.py --full-line --workload $wl --no-cpu --no-live-traffic $extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s %-8s value %.1f ms/step %.4f kernel_ms %.4f (min %.4f) frac %.4f' % ('$v', '$wl', d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['frac']))"
    done
  done
done
cd tools/micro && ./fft4096_wave 2>&1 | tee ../../gpurun_out/fft4096_wave.txt
