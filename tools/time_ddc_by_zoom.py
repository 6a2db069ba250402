"""Waterfall DDC cost by decimation: 14 channels all at the same zoom, 2^24 ADC samples."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Ddc, WfParams   # noqa: E402

ctx = Context(0)
n = 1 << 24
adc = (3000 * np.cos(2 * np.pi * 0.0123 * np.arange(n))).astype(np.int16)
d_adc = ctx.alloc(adc.nbytes)
ctx.upload(d_adc, adc)
nch = 14
for zoom in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5, 9, 10, 14]:
    d = Ddc(ctx, nchan=nch, max_samples=n)
    p = WfParams.for_zoom(zoom, 1000.0 * zoom, adc_clock=66.6666e6, ui_srate=30.0e6)
    for ch in range(nch):
        d.set_wf(ch, (p.i_offset + 12345 * ch) & 0xFFFFFFFFFFFF, p.decim)
    stride = n // p.decim + 2
    d_out = ctx.alloc(nch * stride * 4)
    chans = list(range(nch))
    for _ in range(2):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ms = ctx.timer_stop() / 5
    print("zoom %2d (R = %4d): %7.3f ms per 2^24 samples x %d channels = %6.1f G channel-samples/s"
          % (zoom, p.decim, ms, nch, nch * n / (ms * 1e-3) / 1e9))
    ctx.free(d_out)
    d.close()
ctx.free(d_adc)
