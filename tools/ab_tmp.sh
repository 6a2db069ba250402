export TMPDIR=/tmp
python3 -m pytest tests/test_ddc_gpu.py tests/test_chain_gpu.py tests/test_receivers_gpu.py tests/test_fuzz_gpu.py tests/test_lifecycle_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
for v in 1 0 1 0; do
  KIWIGPU_DDC_SIDE=$v python3 bench.py --workload ddc14 --no-cpu --no-live-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('SIDE=$v ddc14 kernel_ms %.4f min %.4f median %.4f' % (r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))"
done
python3 bench.py --workload cfg2_chain --no-cpu --no-live-traffic --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg2_chain kernel_ms %.4f ddc alone %.4f frames alone %.4f' % (r['kernel_ms'], r['ddc_ms_alone'], r['frames_ms_alone']))"
python3 bench.py --workload receivers --no-cpu --no-live-traffic --steps 40 --warmup 4 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('receivers ms/step %.4f x realtime %.2f' % (d['ms_per_step'], d['x_realtime_all_receivers']))"
