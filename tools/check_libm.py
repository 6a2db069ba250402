"""The device's log10f / powf / expf (csrc/kg_libm.h) and the oracle's restatement of the same glibc 2.35 algorithms against the
image's libm.

    python tools/check_libm.py                  the restatements on the CPU, every 16th float (seconds)
    python tools/check_libm.py --exhaustive     every float on the CPU restatements (log10f / logf: all 2 139 095 041 non-negative
                                                patterns; powf(10, y) and expf: all 2^32; powf(x, y): 2^28 random pairs), AND on the
                                                GPU through kg_math_dev (needs a GPU)

The truth on both sides is libm's own function; nothing is tolerated: a difference is a bit difference (NaN equals NaN).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kiwi_oracle as ko            # noqa: E402

exhaustive = "--exhaustive" in sys.argv
step = 1 if exhaustive else 16
TOTAL = 0x7f800001                               # +0 .. +inf
threads = min(os.cpu_count() or 8, 64)
fma_host = "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
print("host: %d threads, FMA %s (glibc runs its %s build of expf / powf / logf)" % (threads, fma_host, "FMA" if fma_host else "baseline"))
for fused in (True, False):
    t = time.time()
    done, bad_ln, bad_l10, where = ko.libm_check_range(0, TOTAL, step, fused, threads)
    print("CPU restatement (%s multiply-adds): %d floats, logf differences %d, log10f differences %d%s  [%.1f s]" % (
        "fused" if fused else "unfused", done, bad_ln, bad_l10, " first at 0x%08x" % where if bad_ln or bad_l10 else "", time.time() - t))
    assert bad_ln == 0 and bad_l10 == 0
    t = time.time()
    done, bad_p, bad_e, bad_r, nr = ko.libm_check_pow_exp(0, 1 << 32, step, fused, fma_host, threads)
    print("CPU restatement (%s multiply-adds, expf residual %s): %d floats, powf(10, y) differences %d, expf differences %d, "
          "powf(x, y) differences %d of %d random pairs  [%.1f s]" % ("fused" if fused else "unfused", "fused" if fma_host else "unfused",
                                                                       done, bad_p, bad_e, bad_r, nr, time.time() - t))
    assert bad_p == 0 and bad_e == 0 and bad_r == 0
d = ko.libm_check_pow_exp(0, 1 << 32, step, True, not fma_host, threads)
print("(with the OTHER residual, expf differs on %d of %d arguments)" % (d[2], d[0]))
if exhaustive:
    from flydog_sdr_gps_amd import Context, post  # noqa: E402
    ctx = Context(0)
    CH = 1 << 26
    for name, fn, truth, total in (("log10f", post.MATH_LOG10F, ko.libm_log10f_bits, TOTAL + (1 << 23)),
                                   ("powf(10, y)", post.MATH_POWF, lambda f, n: ko.libm_powf_bits(10.0, f, n), 1 << 32),
                                   ("expf", post.MATH_EXPF, ko.libm_expf_bits, 1 << 32)):
        bad, t = 0, time.time()
        for first in range(0, total, CH):
            n = min(CH, total - first)
            got, want = post.math_dev(ctx, fn, first_bits=first, n=n, base=10.0), truth(first, n)
            neq = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
            bad += int(neq.sum())
            if neq.any():
                k = int(np.argmax(neq))
                print("  GPU %s differs at 0x%08x: %r vs %r" % (name, first + k, got[k], want[k]))
        print("GPU kg_math_dev %s: %d floats, differences from libm: %d  [%.1f s]" % (name, total, bad, time.time() - t))
        assert bad == 0
    ctx.close()
print("ok")
