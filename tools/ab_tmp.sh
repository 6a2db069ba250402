export TMPDIR=/tmp
python3 -m pytest tests/test_ddc_gpu.py tests/test_chain_gpu.py tests/test_receivers_gpu.py tests/test_fuzz_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do python3 bench.py --workload ddc14 --no-cpu --no-live-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ddc14 kernel_ms %.4f min %.4f' % (r['kernel_ms'], r['kernel_ms_min']))"; done
for i in 1 2; do python3 bench.py --workload receivers --no-cpu --no-live-traffic --steps 40 --warmup 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('receivers ms/step %.4f x realtime %.2f' % (d['ms_per_step'], d['x_realtime_all_receivers']))"; done
