export TMPDIR=/tmp
python3 -m pytest tests/test_wf_gpu.py tests/test_chain_gpu.py tests/test_golden_gpu.py tests/test_receivers_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do python3 bench.py --workload wf14 --no-cpu --no-live-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('wf14 kernel_ms %.4f min %.4f frac %.4f' % (r['kernel_ms'], r['kernel_ms_min'], r['frac']))"; done
