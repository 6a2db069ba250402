"""Diagnostic: where the forward 16384-point FFT kernel spends its time (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flydog_sdr_gps_amd import Context, Searcher, synth
from flydog_sdr_gps_amd._lib import check, ptr

ctx = Context(0)
s = Searcher(ctx)
s.sample_iq16(synth.config1_iq16())
st = np.zeros(8, np.uint64)
check(s.lib.kg_acq_debug_fft_stamps(s.h, 0, ptr(st), 8), "stamps")
d = (st[1:4] - st[0]).astype(np.int64) * 10
print("fft_sub stamps (ns since start): loaded %d, transform %d, stored %d" % tuple(d))

# correlator: 32 SVs x 41 bins, one block
from flydog_sdr_gps_amd import prn, sats
for sat in range(32):
    s.set_code(sat, prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2]))
svs = np.arange(32, dtype=np.int32)
cs = np.zeros(512 + 4 * 1024, np.uint64)
check(s.lib.kg_acq_debug_corr_stamps(s.h, 1, ptr(svs), 32, ptr(cs), cs.size), "corr stamps")
cyc, rt = int(cs[2] - cs[0]), int(cs[3] - cs[1]) * 10
print("correlator WG life: %d cycles, %d ns -> clock %.2f GHz" % (cyc, rt, cyc / max(rt, 1)))
names = ["p0 radix", "p0 write", "barrier1", "p1 read", "p1 tw+radix", "p1 write+barrier2",
         "p2 read", "p2 tw+radix"]
order = [8, 9, 10, 0, 1, 2, 3, 4, 5, 6, 7, 11]
labels = ["conjmul(wait loads)", "prefetch issue", "p0 radix", "p0 write", "barrier1", "p1 read",
          "p1 tw+radix", "p1 write+bar2", "p2 read", "p2 tw+radix", "combine"]
for it in range(0, 12):
    v = cs[16 + 16 * it: 32 + 16 * it].astype(np.int64)
    if v[8] == 0:
        break
    seq = [v[i] for i in order]
    d = np.diff(seq)
    gap = (v[8] - prev_end) if it else 0
    prev_end = v[11]
    print("item %2d: total %5d cyc (gap %5d) | " % (it, seq[-1] - seq[0], gap) +
          ", ".join("%s %d" % (l, x) for l, x in zip(labels, d)))
