"""The compiled C++ host caller (examples/search_dropin.cpp: SearchInit -> per-SV Sample / Correlate /
ChanStart through include/kiwigpu.h, the INTEGRATION.md section 1 sequence) gives the results the
Python mirror gives for BASELINE configs[0]."""
import os
import re
import subprocess

import numpy as np
import pytest

from flydog_sdr_gps_amd import Searcher, handoff, prn, sats, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "search_dropin")


def run_example(tmp_path, bits, *args):
    f = tmp_path / "bits.bin"
    f.write_bytes(np.asarray(bits, np.uint8).tobytes())
    out = subprocess.run([EXE, str(f)] + list(args), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = {}
    for m in re.finditer(r"sat (\d+) prn (\d+) snr ([\d.]+) lo_shift (-?\d+) ca_shift (-?\d+) lo_rate (0x[0-9a-f]+) "
                         r"ca_rate (0x[0-9a-f]+) ca_pause (\d+)", out.stdout):
        rows[int(m.group(1))] = (float(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6), 16),
                                 int(m.group(7), 16), int(m.group(8)))
    lat = re.search(r"per-SV latency us: median ([\d.]+) min ([\d.]+)", out.stdout)
    return rows, (float(lat.group(1)), float(lat.group(2))), out.stdout


def test_cpp_caller_equals_python_mirror_config0(gpu_ctx, tmp_path):
    assert os.path.exists(EXE), "examples/search_dropin is not built (run __graft_entry__.build())"
    bits = synth.config0_bits()
    svs = [0, 5, 11]                                   # PRN 1 present; 6 and 12 absent
    s = Searcher(gpu_ctx)
    for sv in svs:
        s.set_code(sv, prn.cacode(sats.SATS[sv][1], sats.SATS[sv][2]))
    want = s.search(svs, packed=bits)
    s.close()
    for mode in ((), ("--batch",)):
        rows, lat, text = run_example(tmp_path, bits, "--sats", ",".join(map(str, svs)), *mode)
        assert set(rows) == set(svs), text
        for w in want:
            snr, lo, ca, lo_rate, ca_rate, pause = rows[w.sat]
            assert abs(snr - w.snr) <= 1e-3 * max(1.0, w.snr)          # printed with 4 decimals
            if w.snr >= 16:
                assert (lo, ca) == (w.lo_shift, w.ca_shift) == (6, 4808)
                cs = handoff.chan_start(False, lo, ca, 0.0)
                assert lo_rate == cs.lo_rate                          # the NCO word does not depend on the age
                assert ca_rate == cs.ca_rate and 0 < pause <= 16368
            else:
                assert (lo_rate, ca_rate, pause) == (0, 0, 0)          # ChanStart() not reached (:596-598)
        assert 0 < lat[1] <= lat[0] < 5e4


def test_cpp_caller_per_sv_latency_32_svs(gpu_ctx, tmp_path):
    """The reference's calling pattern -- one SV per Correlate() call, the SV list changing on every
    call -- must not stall on table rebuilds: a per-SV call (enqueue, poll, fetch) stays far below a
    millisecond."""
    bits = synth.config0_bits()
    rows, lat, text = run_example(tmp_path, bits, "--sats", ",".join(map(str, range(32))), "--repeat", "5")
    assert len(rows) == 32 and rows[0][1:3] == (6, 4808)
    print(text.splitlines()[-1])
    assert lat[0] < 500.0, text
