"""Times Correlate() alone and Sample()+Correlate() for configs[1] without checking results
(for timing-only experimental builds: KIWIGPU_LIBRARY=<lib.so> python tools/time_acq.py [B ...])."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth   # noqa: E402

ctx = Context(0)
for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
    s = Searcher(ctx, max_blocks=2 * B)
    svs = list(range(32))
    for sat in svs:
        _, t1, t2, _ = sats.SATS[sat]
        s.set_code(sat, prn.cacode(t1, t2))
    iq = np.stack([synth.config1_iq16(seed=0x5EED0002 + b) for b in range(B)])
    d_iq = ctx.alloc(iq.nbytes)
    ctx.upload(d_iq, iq)
    s.sample_iq16_batch(d_iq, B, first_block=0)
    s.sample_iq16_batch(d_iq, B, first_block=B)
    for _ in range(3):
        s.correlate_async(svs, nblocks=B)
    ctx.sync()
    ctx.timer_start()
    for _ in range(50):
        s.correlate_async(svs, nblocks=B)
    corr = ctx.timer_stop() / 50 * 1e3
    par = 0
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        s.sample_iq16_batch(d_iq, B, first_block=par * B)
        s.correlate_async(svs, nblocks=B, first_block=par * B)
        par ^= 1
    ctx.sync()
    step = (time.perf_counter() - t0) / 100 * 1e6
    res, _ = s.fetch(want_cells=False)
    found = sorted(int(sv) + 1 for sv in svs if res[0, sv]["snr"] >= 16)
    print("B=%d  correlate %.1f us  step %.1f us  %.0f Msamples/s  found %s" % (
        B, corr, step, B * 65536 / step, found))
    s.close()
    ctx.free(d_iq)
