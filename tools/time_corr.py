"""Correlator launch time, 32 blocks x 32 SVs x 41 bins, HIP events (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = Context(0)
s = Searcher(ctx, max_blocks=B)
for sat in range(32):
    s.set_code(sat, prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2]))
iq = synth.config1_iq16()
for b in range(B):
    s.sample_iq16(iq, block=b)
svs = list(range(32))
for _ in range(5):
    s.correlate_async(svs, nblocks=B)
ctx.sync()
ctx.timer_start()
for _ in range(100):
    s.correlate_async(svs, nblocks=B)
print("%.4f ms per launch of %d blocks (KIWIGPU_ACQ_WGS_PER_CU=%s)" % (ctx.timer_stop() / 100, B, os.environ.get("KIWIGPU_ACQ_WGS_PER_CU", "-")))
