"""Times the audio DDC (kg_rxddc_push_dev) for a few channel counts on 2^22 ADC samples."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, RxDdc   # noqa: E402
from flydog_sdr_gps_amd.ddc import RX_DECIM, rx_phase_inc   # noqa: E402

ctx = Context(0)
n = 1 << 22
adc = (3000 * np.cos(2 * np.pi * 0.0123 * np.arange(n))).astype(np.int16)
d_adc = ctx.alloc(adc.nbytes)
ctx.upload(d_adc, adc)
for nch in [int(a) for a in sys.argv[1:]] or [4, 14, 128]:
    d = RxDdc(ctx, nchan=nch, max_samples=n)
    for ch in range(nch):
        d.set_freq(ch, rx_phase_inc(1.0e6 + 1.0e4 * ch, 66.6666e6))
    stride = n // RX_DECIM + 2
    d_out = ctx.alloc(nch * stride * 6)
    chans = list(range(nch))
    for _ in range(2):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ctx.sync()
    ctx.timer_start()
    for _ in range(10):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ms = ctx.timer_stop() / 10
    print("audio DDC %4d channels: %7.3f ms per 2^22 samples = %6.1f x real time at 66.67 MS/s, %6.1f G channel-samples/s"
          % (nch, ms, n / 66.6666e6 / (ms * 1e-3), nch * n / (ms * 1e-3) / 1e9))
    ctx.free(d_out)
    d.close()
ctx.free(d_adc)
