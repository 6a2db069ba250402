export TMPDIR=/tmp
for v in base r02 base r02; do
 if [ $v = base ]; then lib=""; else lib=$PWD/flydog_sdr_gps_amd/libkiwigpu_$v.so; fi
 KIWIGPU_LIBRARY=$lib python3 - <<PY
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth
import flydog_sdr_gps_amd._lib as L
dev = torch.device("cuda", 0)
ctx = Context(0, torch.cuda.current_stream(dev).cuda_stream)
B = 32
s = Searcher(ctx, max_blocks=2 * B)
for sat in range(32): s.set_code(sat, prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2]))
iq = torch.from_numpy(np.stack([synth.config1_iq16(seed=b) for b in range(B)])).to(dev)
svs = list(range(32))
def step(first):
    s.sample_iq16_batch(int(iq.data_ptr()), B, first_block=first); s.correlate_async(svs, nblocks=B, first_block=first)
for i in range(300): step((i & 1) * B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): step((i & 1) * B)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("$v enqueue %.4f ms/step, total %.4f ms/step" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3))
PY
done
