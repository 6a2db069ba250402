"""Replays a failing `post` trial that tools/fuzz_parity.py saved (gpurun_out/fuzz_fail_post_<k>.npz) through the library and
the oracle and shows where the CAgc outputs part: usage python tools/replay_post.py <file.npz>"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Post   # noqa: E402
from oracle import kiwi_oracle as ko           # noqa: E402

d = np.load(sys.argv[1])
mode = int(d["mode"])
ctx = Context(0)
P = Post(ctx, nchan=1)
a = ko.Agc()
P.set_mode(0, mode); P.set_smeter(0, 12000.0); P.reset(0)
P.set_am_passband(0, -4900, 4900, 12000.0); P.squelch_setup(0, 12000.0); P.squelch_set(0, 0, 0)
seg = 0
while "x%d" % seg in d.files:
    prm = d["prm%d" % seg]
    args = (bool(prm[0]), bool(prm[1]), int(prm[2]), int(prm[3]), int(prm[4]), int(prm[5]), float(prm[6]))
    P.set_agc(0, *args); a.set_parameters(*args)
    x = d["x%d" % seg]
    s16, demod, agc = P.process([0], x[None, :])
    want = a.process_cpx(x)
    err = np.abs(agc[0] - want)
    scale = max(np.abs(want).max(), 1e-20)
    bad = np.nonzero(err > 2e-5 * scale)[0]
    print("segment %d: mode %d agc %s n %d  max |x| %.1f  max |want| %.3f  max err %.3e (%.2e of max)  samples over the bar: %d%s"
          % (seg, mode, args, x.size, np.abs(x).max(), scale, err.max(), err.max() / scale, bad.size, "" if not bad.size else " first %d last %d" % (bad[0], bad[-1])))
    if bad.size:
        i0 = max(0, bad[0] - 3)
        for i in range(i0, min(x.size, i0 + 10)):
            print("   i %4d  |x| %10.3f  got %s  want %s  rel %.2e" % (i, np.abs(x[i]), agc[0][i], want[i], err[i] / scale))
        # gain ratio got / want along the segment: a step in it = a branch taken differently
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.abs(agc[0]) / np.abs(want)
        fin = np.isfinite(ratio) & (np.abs(want) > 1e-3 * scale)
        r = ratio[fin]
        print("   gain ratio got / want over the segment: min %.6f max %.6f; at the first bad sample %.6f, at the last sample %.6f"
              % (r.min(), r.max(), ratio[bad[0]], ratio[np.nonzero(fin)[0][-1]]))
    seg += 1
P.close()
