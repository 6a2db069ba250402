export TMPDIR=/tmp
KIWIGPU_ACQ_CA8=1 python3 -m pytest tests/test_acq_gpu.py tests/test_acq10_gpu.py tests/test_golden_gpu.py tests/test_example_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
for v in 1 0 1 0; do
  KIWIGPU_ACQ_CA8=$v python3 bench.py --workload acq --no-cpu --no-live-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('CA8=$v acq kernel_ms %.4f min %.4f frac %.4f' % (r['kernel_ms'], r['kernel_ms_min'], r['frac']))"
done
for v in 1 0; do
  KIWIGPU_ACQ_CA8=$v python3 bench.py --workload acq10ms --no-cpu --no-live-traffic --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('CA8=$v acq10ms kernel_ms %.4f frac %.4f' % (r['kernel_ms'], r['frac']))"
done
KIWIGPU_ACQ_CA8=1 KIWIGPU_ACQ_CA8_WGS=1 python3 bench.py --workload acq --no-cpu --no-live-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('CA8=1 one WG per CU: kernel_ms %.4f' % (r['kernel_ms']))"
