#!/bin/bash
# GPU box: knock-out timing of the waterfall frame kernel -- what is its time sensitive to?  Variants built with
#   make -C flydog_sdr_gps_amd/csrc VARIANT=ko$k EXTRA="-DKG_WF_KO=$k -DKG_WF_FUSED_TW=0"   (k = 1 no barriers, 2 no tile stores / loads, 3 no butterflies and
#   twiddles, 4 no fetch of the next frame, 5 no pixel walk, 6 no combine / power, 7 static frame assignment instead of the claim, 8 no tile
#   stores, 9 no tile loads); rows are wrong by construction, the bench line is
#   tagged "invalid".  usage: tools/ko_wf.sh base-variant ko1 ko2 ...
export TMPDIR=/tmp KIWIGPU_BENCH_TIMING_EXPERIMENT=1
for v in "$@"; do
  for n in 2 1; do
    KIWIGPU_WF_WGS_PER_CU=$n KIWIGPU_LIBRARY=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so timeout 300 python3 bench.py --workload wf14 --no-cpu --no-live-traffic --steps 100 2>&1 | python3 -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads([l for l in t.splitlines() if l.startswith('{')][-1]); print('%-6s workgroups/CU %d  ms/step %.4f  kernel %.4f' % ('$v', $n, d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e:
    print('$v', 'no line:', t[-300:])
"
  done
done
