"""Waterfall DDC cost by channel count at one decimation (R = 64: the 64-bit path, no staging): how the run passes'
time follows the number of workgroups per CU (16 384 runs = 64 workgroups of 256 runs per channel, 2^24 ADC samples)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Ddc   # noqa: E402

ctx = Context(0)
n = 1 << 24
adc = (3000 * np.cos(2 * np.pi * 0.0123 * np.arange(n))).astype(np.int16)
d_adc = ctx.alloc(adc.nbytes)
ctx.upload(d_adc, adc)
R = 64
for nch in [int(a) for a in sys.argv[1:]] or [8, 10, 12, 13, 14, 15, 16, 17, 20]:
    d = Ddc(ctx, nchan=nch, max_samples=n)
    for ch in range(nch):
        d.set_wf(ch, (0x0123456789AB + 12345 * ch) & 0xFFFFFFFFFFFF, R)
    stride = n // R + 2
    d_out = ctx.alloc(nch * stride * 4)
    chans = list(range(nch))
    for _ in range(3):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ms = ctx.timer_stop() / 10
    print("%2d channels (%4d workgroups per run pass): %7.3f ms = %6.1f G channel-samples/s, %6.2f us per channel"
          % (nch, nch * 64, ms, nch * n / (ms * 1e-3) / 1e9, ms * 1e3 / nch))
    ctx.free(d_out)
    d.close()
ctx.free(d_adc)
