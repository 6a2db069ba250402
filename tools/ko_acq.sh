#!/bin/bash
# GPU box: knock-out timing of the C/A correlator -- what is its time sensitive to?  Variants built with
#   make -C flydog_sdr_gps_amd/csrc VARIANT=ako$k EXTRA=-DACQ_KO=$k   (k = 1 no barriers inside the item, 3 no butterflies / products /
#   twiddles, 4 no operand rows of the next item, 8 no tile stores, 9 no tile loads, 10 no twiddle-accumulate, 11 no power / maximum scan);
#   results are wrong by construction, the bench line is tagged "invalid".  usage: tools/ko_acq.sh [workload] variant...   (base = the product)
export TMPDIR=/tmp KIWIGPU_BENCH_TIMING_EXPERIMENT=1
wl=acq; case "$1" in acq|acq59|acq10ms) wl=$1; shift;; esac
for v in "$@"; do
  if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
  extra="--steps 100"; [ $wl = acq10ms ] && extra="--steps 20 --warmup 3"
  KIWIGPU_LIBRARY=$lib timeout 300 python3 bench.py --workload $wl --no-cpu --no-live-traffic $extra 2>&1 | python3 -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads([l for l in t.splitlines() if l.startswith('{')][-1]); print('%-6s %s ms/step %.4f  kernel %.4f' % ('$v', '$wl', d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e:
    print('$v', 'no line:', t[-300:])
"
done
