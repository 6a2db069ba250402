"""The device's log10f (csrc/kg_libm.h) and the oracle's restatement of the same algorithm against log10f of the image's libm.

    python tools/check_log10f.py                  the restatement on the CPU, every 16th float (seconds)
    python tools/check_log10f.py --exhaustive     every non-negative float (2 139 095 041 patterns) on the CPU restatement, fused
                                                  and unfused, AND on the GPU through kg_math_log10f_dev (needs a GPU)

The truth on both sides is libm's own log10f (oracle.libm_log10f_bits); nothing is tolerated: a difference is a bit difference.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kiwi_oracle as ko            # noqa: E402

exhaustive = "--exhaustive" in sys.argv
TOTAL = 0x7f800001                               # +0 .. +inf
threads = os.cpu_count() or 8
for fused in (True, False):
    t = time.time()
    done, bad_ln, bad_l10, where = ko.libm_check_range(0, TOTAL, 1 if exhaustive else 16, fused, threads)
    print("CPU restatement (%s multiply-adds): %d floats, logf differences %d, log10f differences %d%s  [%.1f s, %d threads]" % (
        "fused" if fused else "unfused", done, bad_ln, bad_l10, " first at 0x%08x" % where if bad_ln or bad_l10 else "", time.time() - t, threads))
    assert bad_ln == 0 and bad_l10 == 0
if exhaustive:
    from flydog_sdr_gps_amd import Context, post  # noqa: E402
    ctx = Context(0)
    CH, bad, t = 1 << 26, 0, time.time()
    for first in range(0, TOTAL + (1 << 23), CH):        # ... and the NaNs behind +inf
        n = min(CH, (TOTAL + (1 << 23)) - first)
        got, want = post.log10f(ctx, first_bits=first, n=n), ko.libm_log10f_bits(first, n)
        neq = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        bad += int(neq.sum())
        if neq.any():
            k = int(np.argmax(neq))
            print("  GPU differs at 0x%08x: %r vs %r" % (first + k, got[k], want[k]))
    print("GPU kg_math_log10f_dev: %d floats (all of +0 .. +inf and 2^23 NaNs), differences from libm's log10f: %d  [%.1f s]" % (
        TOTAL + (1 << 23), bad, time.time() - t))
    ctx.close()
    assert bad == 0
print("ok")
