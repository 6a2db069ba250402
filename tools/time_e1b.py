"""Times Correlate() for Galileo E1B codes (4092 chips, 16368-sample window: the NQ = 4 kernel)
next to the same number of C/A codes.  Random memory codes: timing does not depend on them.
usage: python tools/time_e1b.py [B ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth   # noqa: E402

ctx = Context(0)
rng = np.random.default_rng(5)
NSV = 23
for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
    s = Searcher(ctx, max_blocks=B)
    for sat in range(NSV):
        _, t1, t2, _ = sats.SATS[sat]
        s.set_code(sat, prn.cacode(t1, t2))
        s.set_code(32 + sat, rng.integers(0, 2, 4092).astype(np.uint8), boc=True)
    iq = np.stack([synth.config1_iq16(seed=0x5EED0002 + b) for b in range(B)])
    d_iq = ctx.alloc(iq.nbytes)
    ctx.upload(d_iq, iq)
    s.sample_iq16_batch(d_iq, B, first_block=0)
    for name, svs in (("C/A", list(range(NSV))), ("E1B", list(range(32, 32 + NSV)))):
        for _ in range(3):
            s.correlate_async(svs, nblocks=B)
        ctx.sync()
        ctx.timer_start()
        for _ in range(20):
            s.correlate_async(svs, nblocks=B)
        us = ctx.timer_stop() / 20 * 1e3
        cells = B * NSV * 41
        print("B=%d %s: %8.1f us per Correlate() of %d cells = %6.1f ns/cell" % (B, name, us, cells, us * 1e3 / cells))
    s.close()
    ctx.free(d_iq)
