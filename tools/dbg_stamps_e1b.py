"""Diagnostic (GPU box): in-kernel cycle stamps of the 512-thread four-quarter correlator (acq_correlate8_kernel),
lane 0 of waves 0 and 4 of one workgroup (they share a SIMD), per item: where the two phases of an item spend their time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flydog_sdr_gps_amd import Context, Searcher, synth
from flydog_sdr_gps_amd._lib import check, ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = Context(0)
s = Searcher(ctx, max_blocks=B)
rng = np.random.default_rng(5)
for sat in range(23):
    s.set_code(sat, rng.integers(0, 2, 4092).astype(np.uint8), boc=True)
iq = np.stack([synth.config1_iq16(seed=0x5EED0002 + b) for b in range(B)])
s.sample_iq16_host_batch(iq, 0)
svs = np.arange(23, dtype=np.int32)
cs = np.zeros(512 + 4 * 1024, np.uint64)
check(s.lib.kg_acq_debug_corr_stamps(s.h, B, ptr(svs), 23, ptr(cs), cs.size), "corr stamps")
labels = ["A: wait operands + conj-mul", "A: pass0 + T0 stores (2 rows)", "A: pass2 (2 rows)", "A: T2 stores", "A: stores land",
          "A: barrier", "B: T2/T0 reads + pass1 + T1 stores (2 rows)", "B: pass3 + accumulate (2 rows)", "B: stores land", "B: barrier"]
for wave, base in ((0, 0), (4, 1024)):
    print("wave %d" % wave)
    prev = None
    for it in range(4, 20):
        v = cs[base + 16 + 16 * it: base + 32 + 16 * it].astype(np.int64)
        if v[0] == 0:
            break
        d = np.diff(v[:11])
        gap = int(v[0] - prev) if prev is not None else 0
        prev = v[10]
        if v[11]:
            print("     cell end after item %d: scan %d, wave reductions + red %d, barrier %d, merge + store %d" % ((it,) + tuple(np.diff(v[11:16]))))
        print("  item %2d: %5d cyc (+%4d between items) | " % (it, v[10] - v[0], gap) + ", ".join("%s %d" % (l.split(":")[0] + ":" + l.split(":")[1][:18], x) for l, x in zip(labels, d)))
tot = cs[16 + 16 * 59 + 10] - cs[16 + 0] if cs[16 + 16 * 59] else 0
print("60 items: %d cycles = %.0f per item" % (tot, tot / 60.0))
