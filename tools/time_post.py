"""Times kg_post_process_dev for a batch of receiver channels (512 samples per channel and
launch, the ns_out of c2s_sound()).  usage: python tools/time_post.py [nchan ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Post, post   # noqa: E402

ctx = Context(0)
n = 512
for nchan in [int(a) for a in sys.argv[1:]] or [14, 128, 1024, 8192]:
    P = Post(ctx, nchan=nchan)
    rng = np.random.default_rng(1)
    t = np.arange(n)
    x = (3000 * np.exp(2j * np.pi * rng.uniform(0.01, 0.2, (nchan, 1)) * t)
         + rng.normal(0, 30, (nchan, n)) + 1j * rng.normal(0, 30, (nchan, n))).astype(np.complex64)
    for mode in (post.MODE_SSB, post.MODE_AM, post.MODE_NBFM):
        for ch in range(nchan):
            P.set_agc(ch, True, ch & 1, -100, 50, 6, 1000, 12000.0)
            P.set_smeter(ch, 12000.0)
            P.set_mode(ch, mode)
            P.set_am_passband(ch, -4900, 4900, 12000.0); P.squelch_setup(ch, 12000.0); P.squelch_set(ch, 0, 0)
        chans = np.arange(nchan, dtype=np.int32)
        d_x = ctx.alloc(x.nbytes); ctx.upload(d_x, x)
        d_s = ctx.alloc(nchan * n * 2); d_d = ctx.alloc(nchan * n * 4); d_a = ctx.alloc(nchan * n * 8)
        for _ in range(3):
            P.process_dev(chans, d_x, n, n, d_s, d_d, d_a, n)
        ctx.sync()
        ctx.timer_start()
        reps = 20
        for _ in range(reps):
            P.process_dev(chans, d_x, n, n, d_s, d_d, d_a, n)
        us = ctx.timer_stop() / reps * 1e3
        rt = nchan * n / 12000.0 / (us * 1e-6)
        print("nchan %5d mode %d: %8.1f us per 512-sample pass = %7.1f Msamples/s, %8.0f x real time at 12 kHz"
              % (nchan, mode, us, nchan * n / us, rt / nchan))
        for d in (d_x, d_s, d_d, d_a):
            ctx.free(d)
    P.close()
