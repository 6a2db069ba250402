"""Generates the committed fixtures under tests/golden/ (run in the build container).

  e1b_codes.npz   Galileo E1-B memory codes (hex, straight from the SIS ICD table the
                  reference carries in gps/e1bcode.h) for the PRNs the tests use:
                  E01/E02 (the reference's known answers, gps/search.cpp:295,302)
                  and E11 (an acquisition case).  Data only.
  acq_golden.npz  acquisition cases: packed 1-bit IF input + the CPU oracle's
                  (snr, dop, idx, valid) and per-Doppler peak indices.

The reference holds no golden vectors for this path and its FFT-dependent code is
unbuildable here, so the expected values come from the oracle
(oracle/kiwi_oracle.c, prec=1) -- they pin the oracle against regressions and
travel to the GPU box; they are not reference outputs.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flydog_sdr_gps_amd import prn, sats, synth      # noqa: E402
from oracle import kiwi_oracle as ko                  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

ref = "/root/reference/gps/e1bcode.h"
strings = re.findall(r'"([0-9A-F]{1023})"', open(ref).read())
assert len(strings) == 50
e1b = {"E01_hex": strings[0], "E02_hex": strings[1], "E11_hex": strings[10]}
np.savez_compressed(os.path.join(GOLD, "e1b_codes.npz"), **e1b)

cases = []


def add(sat, bits, chips=None):
    kind = sats.SATS[sat][3]
    if kind == sats.E1B:
        code, limit = ko.code_fft(chips, boc=True), sats.E1B_LIMIT
    else:
        code, limit = ko.code_fft(prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2])), sats.L1_LIMIT
    r, cells = ko.correlate(code, ko.sample_bits(bits), limit=limit)
    cases.append((sat, bits, chips, r, cells))
    print("sat %2d (%s %d): snr %.2f dop %d idx %d valid %d" %
          (sat, kind, sats.SATS[sat][0], r["snr"], r["dop"], r["idx"], r["valid"]))


def ca(sat):
    return prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2])


add(0, synth.config0_bits())                                                   # BASELINE configs[0]
add(16, synth.gps_scene_bits([(ca(16), 17.75, -4100.0, 2.0)], seed=11, cn0_dbhz=44.0))
add(32, synth.gps_scene_bits([(ca(32), 800.5, 2600.0, 1.0)], seed=12, cn0_dbhz=47.0))   # QZSS 194
add(4, synth.gps_scene_bits([(ca(0), 300.5, 1500.0, 0.7)], seed=13))           # PRN5 absent
add(8, synth.gps_scene_bits([(ca(8), 1022.9, 5000.0, 0.0)], seed=14, cn0_dbhz=41.0))    # edge Doppler, weak
e11 = prn.e1b_from_hex(e1b["E11_hex"])
sat_e11 = [i for i, s in enumerate(sats.SATS) if s[3] == sats.E1B and s[0] == 11][0]
add(sat_e11, synth.gps_scene_bits([(e11, 2000.25, -1250.0, 0.5, 48.0, True)], seed=15), e11)

out = {"ncases": len(cases)}
for k, (sat, bits, chips, r, cells) in enumerate(cases):
    out["case%d_sat" % k] = sat
    out["case%d_bits" % k] = bits
    if chips is not None:
        out["case%d_chips" % k] = chips
    out["case%d_result" % k] = np.array([r["snr"], r["dop"], r["idx"], r["valid"]], np.float64)
    out["case%d_cell_idx" % k] = cells["idx"]
    out["case%d_cell_snr" % k] = cells["snr"]
np.savez_compressed(os.path.join(GOLD, "acq_golden.npz"), **out)
print("wrote", GOLD)
