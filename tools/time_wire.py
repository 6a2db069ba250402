"""Times the wire-format kernels: sound ADPCM for a batch of channels (512 samples each) and
waterfall packets for a batch of rows.  usage: python tools/time_wire.py [nchan ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Adpcm, Context, wire   # noqa: E402

ctx = Context(0)
n = 512
for nchan in [int(a) for a in sys.argv[1:]] or [14, 1024, 8192]:
    rng = np.random.default_rng(1)
    x = rng.normal(0, 5000, (nchan, n)).astype(np.int16)
    A = Adpcm(ctx, nchan=nchan)
    d_x, d_o = ctx.alloc(x.nbytes), ctx.alloc(nchan * n // 2)
    ctx.upload(d_x, x)
    chans = np.arange(nchan, dtype=np.int32)
    for _ in range(3):
        A.encode_dev(chans, d_x, n, n, d_o, n // 2)
    ctx.sync()
    ctx.timer_start()
    for _ in range(20):
        A.encode_dev(chans, d_x, n, n, d_o, n // 2)
    us = ctx.timer_stop() / 20 * 1e3
    rows = rng.integers(0, 256, (nchan, 1024)).astype(np.uint8)
    d_r, d_p = ctx.alloc(rows.nbytes), ctx.alloc(nchan * wire.WF_PKT_MAX)
    ctx.upload(d_r, rows)
    infos = [(0, 3, i, True) for i in range(nchan)]
    for _ in range(2):
        wire.wf_packets_dev(ctx, d_r, 1024, infos, d_p)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        wire.wf_packets_dev(ctx, d_r, 1024, infos, d_p)
    usw = ctx.timer_stop() / 10 * 1e3
    print("nchan %5d: sound ADPCM %7.1f us per 512-sample block (%6.1f Msamples/s); "
          "waterfall packets %7.1f us per %d compressed rows (%6.1f Mpixels/s)"
          % (nchan, us, nchan * n / us, usw, nchan, nchan * 1024 / usw))
    for d in (d_x, d_o, d_r, d_p):
        ctx.free(d)
    A.close()
