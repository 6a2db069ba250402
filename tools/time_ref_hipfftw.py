"""How fast is the REFERENCE ITSELF when its FFTW calls are served by a GPU FFT library?  oracle/_ref/search_ref is gps/search.cpp
compiled in place against hipFFTW (the FFTW3 interface over hipFFT, oracle/build_ref.sh): Sample() and Correlate() run on the host as
the reference wrote them, every fftwf_execute() is a hipFFT transform on the GPU with its copies.  This times Correlate() of 32 SVs
(41 Doppler bins each: 41 conjugate products on the host, 41 16384-point inverse transforms on the GPU, 41 power scans on the host)
by running the same block with 32 and with 320 Correlate() calls and taking the difference.  GPU box only; a context number for
DESIGN.md section 4, not a baseline (it is neither a CPU path nor ours).

    python tools/time_ref_hipfftw.py
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "search_ref")
rng = np.random.default_rng(1)
bits = rng.integers(0, 256, 8192, dtype=np.uint8)


def run(reps):
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "s.txt"), "w").write("S\n" + "".join("C %d\n" % sat for _ in range(reps) for sat in range(32)))
        bits.tofile(os.path.join(tmp, "in.bin"))
        t = time.time()
        subprocess.run([REF, os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return time.time() - t


run(1)
t1 = min(run(1) for _ in range(3))
t10 = min(run(10) for _ in range(3))
per_sv = (t10 - t1) / (9 * 32)
print("reference search.cpp + hipFFTW: start-up + SearchInit() + Sample() + 32 Correlate() %.2f s; one Correlate() (41 bins) %.3f ms"
      % (t1, per_sv * 1e3))
print("= %.2f Msamples/s for BASELINE configs[1]'s 32 SVs on one 65536-sample block (the library: ~2 650 on the same GPU)" % (65536 / (32 * per_sv) / 1e6))
