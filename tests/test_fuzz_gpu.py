"""A short run of the randomised differential soak (tools/fuzz_parity.py: random configurations of
the DDCs, CFastFIR, the S-meter/AGC/detector block and the wire formats against the oracle).
The full soak (150 s per module on the round's final library: 786k trials, 0 failures on MI355X, profiles/r04_fuzz_soak_final.txt; round 4: the waterfall DDC trial
mixes pushes, one-shot captures, resets, retunes, new phases, channel subsets and the deferred output stage -- it found the
stale-channel / re-base ordering bug tests/test_ddc_gpu.py::test_push_after_a_capture_of_some_channels now pins; round 5: a receiver-bank
module -- random banks with retunes between steps -- and 800k trials, 0 failures, profiles/r05_fuzz_soak_final.txt) is run by hand."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_soak():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "1.0", "7"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failures: 0" in r.stdout
