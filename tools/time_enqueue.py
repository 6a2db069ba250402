"""Host cost of enqueueing one waterfall-DDC push and one frames launch (GPU box): the C ABI call from Python, no sync inside the loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (the loader shares torch's HIP runtime when torch is imported first)
from flydog_sdr_gps_amd import Context, Ddc, Waterfall, WfParams, wf   # noqa: E402

ctx = Context(0)
n = 1 << 22
adc = (3000 * np.cos(2 * np.pi * 0.0123 * np.arange(n))).astype(np.int16)
d_adc = ctx.alloc(adc.nbytes)
ctx.upload(d_adc, adc)
nch = 14
d = Ddc(ctx, nchan=nch, max_samples=n)
for ch in range(nch):
    p = WfParams.for_zoom(ch, 1000.0 * ch, adc_clock=66.6666e6, ui_srate=30.0e6)
    d.set_wf(ch, p.i_offset, p.decim)
stride = n + 2
d_out = ctx.alloc(nch * stride * 4)
chans = list(range(nch))
for name, reps in (("push_dev", 300),):
    for _ in range(5):
        d.push_dev(d_adc, n, chans, d_out, stride)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        d.push_dev(d_adc, n, chans, d_out, stride)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("%s: host %.1f us per call, with the GPU work %.1f us per call" % (name, (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6))
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(2000):
        ctx.mark(1)
    t1 = time.perf_counter()
    ctx.sync()
    print("one empty kernel launch through the ABI: host %.2f us" % ((t1 - t0) / 2000 * 1e6))
