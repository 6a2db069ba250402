"""Diagnostic (GPU box): the correlator's in-kernel clock and per-item phase times under sustained
load (MI355X_MICROARCH.md, DVFS item 6: delta s_memtime / delta s_memrealtime x 100 MHz after >= 2 s of
back-to-back launches).  Uses the stamped instantiation of the kernel; the product kernel has no stamps."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth
from flydog_sdr_gps_amd._lib import check, ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = Context(0)
s = Searcher(ctx, max_blocks=B)
for sat in range(32):
    s.set_code(sat, prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2]))
iq = synth.config1_iq16()
for b in range(B):
    s.sample_iq16(iq, block=b)
svs = np.arange(32, dtype=np.int32)
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(50):
        s.correlate_async(svs, nblocks=B)
    ctx.sync()
    n += 50
print("%d launches of %d blocks in %.2f s: %.4f ms per launch" % (n, B, time.perf_counter() - t0, (time.perf_counter() - t0) / n * 1e3))
cs = np.zeros(512 + 4 * 1024, np.uint64)
check(s.lib.kg_acq_debug_corr_stamps(s.h, B, ptr(svs), 32, ptr(cs), cs.size), "corr stamps")
cyc, rt = int(cs[2] - cs[0]), int(cs[3] - cs[1]) * 10
print("stamped WG life: %d cycles, %d ns -> in-kernel clock %.3f GHz" % (cyc, rt, cyc / max(rt, 1)))
order = [8, 9, 10, 0, 1, 2, 3, 4, 5, 6, 7, 11]
labels = ["conjmul(wait loads)", "prefetch issue", "p0 radix", "p0 write", "barrier1", "p1 read",
          "p1 tw+radix", "p1 write+bar2", "p2 read", "p2 tw+radix", "combine"]
tot = np.zeros(len(labels))
cnt = 0
items = []
for it in range(24):
    v = cs[16 + 16 * it: 32 + 16 * it].astype(np.int64)
    if v[8] == 0:
        break
    seq = np.array([v[i] for i in order])
    tot += np.diff(seq)
    items.append((seq[0], seq[-1]))
    cnt += 1
print("mean over %d items (cycles): " % cnt + ", ".join("%s %.0f" % (l, x / cnt) for l, x in zip(labels, tot)))
per = [items[i + 1][0] - items[i][0] for i in range(len(items) - 1)]
print("item period (start to start): median %d cycles; in-item %.0f" % (int(np.median(per)), tot.sum() / cnt))

# every workgroup's life
nwg = ctx.num_cus * 2
w = cs[512:512 + 4 * nwg].astype(np.int64).reshape(nwg, 4)
w = w[w[:, 3] > 0]
t0 = w[:, 0].min()
start, end, xcc, cells = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, w[:, 2] & 0xF, w[:, 3]
life = end - start
print("%d workgroups: start skew %.1f us, life min %.0f median %.0f max %.0f us, last end %.0f us; cells per WG %d..%d"
      % (len(w), start.max(), life.min(), np.median(life), life.max(), end.max(), cells.min(), cells.max()))
for x in sorted(set(xcc)):
    m = xcc == x
    print("  xcc %d: %3d WGs, life median %.0f max %.0f us, us/cell median %.2f" % (x, m.sum(), np.median(life[m]), life[m].max(), np.median(life[m] / cells[m])))
late = start > 5.0
print("  workgroups starting later than 5 us: %d (first at %.0f us)" % (late.sum(), start[late].min() if late.any() else 0))
