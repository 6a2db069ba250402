"""Diagnostic (GPU box): where do the 10 ms correlator's cells differ from the oracle?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flydog_sdr_gps_amd import Context, Searcher, acq, sats, synth
from oracle import kiwi_oracle as ko
from tests.fixtures import e1b_chips

N10, FFT10 = acq.NSAMPLES_10MS, acq.FFT_LEN_10MS
codes = synth.all_sv_codes(e1b_chips())
ctx = Context(0)
s = Searcher(ctx, dop_lo=-128, dop_hi=127, max_blocks=1, nsamples=N10, fft_len=FFT10)
for sat, (chips, boc) in enumerate(codes):
    s.set_code(sat, chips, boc=boc)
iq = synth.config4_iq16(codes)
s.sample_iq16(iq)
data = ko.sample_iq16(iq, nsamples=N10, fft_len=FFT10)
for svs in ([0], [8], [0, 8], [36], [37, 38], list(range(16)), list(range(59))):
    res, cells = s.correlate_many(svs)
    oc = np.stack([ko.code_fft(codes[v][0], boc=codes[v][1], fft_len=FFT10) for v in svs])
    lim = [sats.E1B_LIMIT if codes[v][1] else sats.L1_LIMIT for v in svs]
    want, wcells = ko.correlate_many(oc, data, lim, dop_lo=-128, dop_hi=127, nthreads=16)
    bad = np.argwhere((cells[0]["idx"] != wcells["idx"]) |
                      (np.abs(cells[0]["max_pwr"] - wcells["max_pwr"]) > 1e-4 * wcells["max_pwr"]))
    print("svs %s: %d bad cells of %d" % (svs if len(svs) < 6 else "0..%d" % (len(svs) - 1), len(bad), cells[0].size))
    per = {}
    for i, d in bad:
        per.setdefault(svs[i], []).append(d - 128)
    for v, ds in per.items():
        print("   sv %d: %d bad bins, first %s" % (v, len(ds), ds[:12]))
    for i, d in bad[:4]:
        print("   e.g. sv %d dop %d: got %s want %s" % (svs[i], d - 128, cells[0][i, d], wcells[i, d]))
