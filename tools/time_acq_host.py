"""PCIe-inclusive acquisition rate: host buffers handed over per block (kg_acq_sample_iq16 /
kg_acq_sample_bits), 32 SVs x 41 bins, results fetched per block (synchronous) or once per
batch of enqueued blocks (pipelined)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import Context, Searcher, prn, sats, synth   # noqa: E402

ctx = Context(0)
s = Searcher(ctx, max_blocks=8)
svs = list(range(32))
for sat in svs:
    _, t1, t2, _ = sats.SATS[sat]
    s.set_code(sat, prn.cacode(t1, t2))
iq = [synth.config1_iq16(seed=0x5EED0002 + b) for b in range(8)]
bits = synth.config0_bits()
for name, fn, arg, nsamp in (("int16 IQ, 256 KiB/block", s.sample_iq16, iq, 65536), ("1-bit IF, 8 KiB/block", s.sample, [bits] * 8, 65536)):
    for _ in range(3):
        fn(arg[0], block=0); s.correlate_async(svs, nblocks=1); s.fetch(want_cells=False)
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        fn(arg[i & 7], block=0)
        s.correlate_async(svs, nblocks=1)
        s.fetch(want_cells=False)
    t_sync = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for i in range(n):                        # 8 blocks in flight: sample into block b, correlate it, fetch every 8
        b = i & 7
        fn(arg[b], block=b)
        s.correlate_async(svs, nblocks=1, first_block=b)
        if b == 7:
            s.fetch(want_cells=False)
    ctx.sync()
    t_pipe = (time.perf_counter() - t0) / n
    print("%-26s synchronous %6.1f us/block = %6.1f Msamples/s;  8 in flight %6.1f us/block = %6.1f Msamples/s"
          % (name, t_sync * 1e6, nsamp / t_sync / 1e6, t_pipe * 1e6, nsamp / t_pipe / 1e6))
# The reference's own calling pattern (gps/search.cpp:571-575): ONE SV per Correlate() call, the SV list
# changing on every call.  Per call: enqueue (pair table staged through the ring, no stream sync),
# poll until idle (where the reference's coroutine yields), fetch the 16-byte result.
s.sample(bits, block=0)
ctx.sync()
lat, enq = [], []
for rep in range(20):
    for sv in svs:
        t0 = time.perf_counter()
        s.correlate_async([sv], nblocks=1)
        t1 = time.perf_counter()
        while not ctx.poll():
            pass
        s.fetch(want_cells=False)
        lat.append(time.perf_counter() - t0)
        enq.append(t1 - t0)
lat.sort(); enq.sort()
print("one SV per Correlate() call (41 cells): %6.1f us median, %6.1f us min per call; host enqueue %5.1f us median "
      "(never waits for the stream)" % (lat[len(lat) // 2] * 1e6, lat[0] * 1e6, enq[len(enq) // 2] * 1e6))
for B in (8, 32):
    s2 = Searcher(ctx, max_blocks=2 * B)
    for sat in svs:
        _, t1, t2, _ = sats.SATS[sat]
        s2.set_code(sat, prn.cacode(t1, t2))
    batch = np.stack([synth.config1_iq16(seed=0x5EED0002 + b) for b in range(B)])
    for _ in range(2):
        s2.sample_iq16_host_batch(batch, 0); s2.correlate_async(svs, nblocks=B); s2.fetch(want_cells=False)
    n = 40
    t0 = time.perf_counter()
    par = 0
    for i in range(n):
        s2.sample_iq16_host_batch(batch, par * B)
        s2.correlate_async(svs, nblocks=B, first_block=par * B)
        par ^= 1
    ctx.sync()
    t = (time.perf_counter() - t0) / n
    print("int16 IQ host batches of %2d blocks: %7.1f us/batch = %6.1f us/block = %6.1f Msamples/s"
          % (B, t * 1e6, t * 1e6 / B, B * 65536 / t / 1e6))
    s2.close()
s.close()
