export TMPDIR=/tmp
for rep in 1 2; do
for v in base wflog; do
  if [ "$v" = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
  for wl in wf14 cfg2_chain; do
  KIWIGPU_LIBRARY=$lib timeout 300 python3 bench.py --full-line --workload $wl --no-cpu --no-live-traffic --steps 200 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['step_ms_spread']; print('$v $wl ms/step %.4f (min %.4f med %.4f) checked %s' % (d['ms_per_step'], s['min'], s['median'], d.get('checked')))"
  done
done
done
KIWIGPU_LIBRARY=$PWD/flydog_sdr_gps_amd/libkiwigpu_wflog.so timeout 600 python3 -m pytest tests/test_wf_gpu.py tests/test_ref_pins_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
