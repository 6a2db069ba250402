"""Debug: where do the correlator's workgroups land?  Needs a library built with -DKG_XCC_DEBUG
(histogram of (HW_REG_XCC_ID - blockIdx) & 7 over all correlator workgroups; one non-zero bin
means blockIdx % 8 labels the XCDs consistently).  usage: python tools/dbg_xcc.py <lib.so>"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth, _lib   # noqa: E402

hist_fn = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else _lib.library_path()).kg_acq_debug_xcc_hist


def hist():
    h = (C.c_int * 8)()
    assert hist_fn(h) == 0
    return list(h)


ctx = Context(0)
for B in (1, 8, 16):
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
    s.correlate_async(svs, nblocks=B)
    ctx.sync()
    hist()
    t0 = time.perf_counter()
    for _ in range(50):
        s.correlate_async(svs, nblocks=B)
    ctx.sync()
    t1 = time.perf_counter()
    print("B=%d correlate only: %.1f us/step, hist %s" % (B, (t1 - t0) / 50 * 1e6, hist()))
    par = 0
    t0 = time.perf_counter()
    for _ in range(50):
        s.sample_iq16_batch(d_iq, B, first_block=par * B)
        s.correlate_async(svs, nblocks=B, first_block=par * B)
        par ^= 1
    ctx.sync()
    t1 = time.perf_counter()
    print("B=%d sample+correlate: %.1f us/step, hist %s" % (B, (t1 - t0) / 50 * 1e6, hist()))
    s.close()
    ctx.free(d_iq)
