"""Seeded synthetic inputs for tests and bench.py.

Pure input generators (numpy): nothing here is on the measured path.
"""
import numpy as np

from . import prn as _prn
from . import sats as _sats

FS = 16.368e6     # gps/gps.h:43
FC = 4.092e6      # gps/gps.h:42
CPS = 1.023e6     # gps/gps.h:46
NSAMPLES = 65536


def _code_wave(chips, tau_chips, n):
    """(1 - 2 c[floor(n * CPS/FS + tau) mod L]) over n samples."""
    idx = np.floor(np.arange(n) * (CPS / FS) + tau_chips).astype(np.int64) % chips.size
    return 1.0 - 2.0 * chips[idx].astype(np.float64)


def gps_scene_bits(svs, seed, cn0_dbhz=45.0, n=NSAMPLES, boc_svs=()):
    """Packed 1-bit real IF (the reference's sampler format, LSB first).

    svs: list of (chips, tau_chips, doppler_hz, theta) with an optional 5th
    element cn0 (dB-Hz).  x = sum A*code*cos(2 pi (FC+fd) n/FS + theta) + N(0,1);
    bit = x < 0 (the reference maps bit 1 -> -1.0).  A = sqrt(4*10^(cn0/10)/FS)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n)
    x = rng.standard_normal(n)
    for sv in svs:
        chips, tau, fd, theta = sv[:4]
        cn0 = sv[4] if len(sv) > 4 else cn0_dbhz
        boc = sv[5] if len(sv) > 5 else False
        a = np.sqrt(4.0 * 10.0 ** (cn0 / 10.0) / FS)
        code = _code_wave(chips, tau, n)
        if boc:
            sub = (np.arange(n) * (CPS / FS) + tau) % 1.0 >= 0.5
            code = code * np.where(sub, -1.0, 1.0)
        x = x + a * code * np.cos(2 * np.pi * (FC + fd) * t / FS + theta)
    bits = (x < 0).astype(np.uint8)
    return np.packbits(bits, bitorder="little")


def gps_scene_iq16(svs, seed, cn0_dbhz=45.0, n=NSAMPLES, scale=2048.0):
    """Complex int16 IF samples (i,q interleaved), round(scale*x) clipped.
    x = sum A*code*exp(j(2 pi (FC+fd) n/FS + theta)) + CN(0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.0)
    for sv in svs:
        chips, tau, fd, theta = sv[:4]
        cn0 = sv[4] if len(sv) > 4 and sv[4] is not None else cn0_dbhz
        boc = sv[5] if len(sv) > 5 else False
        a = np.sqrt(10.0 ** (cn0 / 10.0) / FS)
        code = _code_wave(chips, tau, n)
        if boc:                                       # BOC(1,1): second half of every chip inverted
            sub = (np.arange(n) * (CPS / FS) + tau) % 1.0 >= 0.5
            code = code * np.where(sub, -1.0, 1.0)
        x = x + a * code * np.exp(1j * (2 * np.pi * (FC + fd) * t / FS + theta))
    iq = np.empty(2 * n, np.int16)
    iq[0::2] = np.clip(np.rint(scale * x.real), -32768, 32767)
    iq[1::2] = np.clip(np.rint(scale * x.imag), -32768, 32767)
    return iq


# BASELINE.json configs[1]: 8 SVs present out of the 32 Navstar PRNs searched
CONFIG1_PRESENT = [(1, 300.5, 1500.0, 0.7), (3, 12.25, -2250.0, 1.9), (7, 911.0, 4400.0, 0.1),
                   (11, 555.75, -4500.0, 2.8), (14, 71.5, 250.0, 4.0), (19, 1000.125, 3100.0, 5.5),
                   (22, 640.0, -700.0, 3.3), (30, 222.625, -3300.0, 1.2)]


def config1_iq16(seed=0x5EED0002, cn0_dbhz=47.0):
    svs = []
    for prn, tau, fd, th in CONFIG1_PRESENT:
        _, t1, t2, _ = _sats.SATS[_sats.navstar_index(prn)]
        svs.append((_prn.cacode(t1, t2), tau, fd, th))
    return gps_scene_iq16(svs, seed, cn0_dbhz)


def config0_bits(seed=0x5EED0001):
    """BASELINE.json configs[0]: PRN1 at tau = 300.5 chips, +1500 Hz, theta 0.7, 45 dB-Hz."""
    _, t1, t2, _ = _sats.SATS[0]
    return gps_scene_bits([(_prn.cacode(t1, t2), 300.5, 1500.0, 0.7)], seed, 45.0)


# BASELINE.json configs[4]: joint L1 C/A (+ QZSS) and Galileo E1B search on a 10 ms block
# (163680 samples at FS), 65536-point transforms, Doppler bin 4.092e6 / 65536 = 62.44 Hz.
NSAMPLES_10MS = 163680
BIN_10MS = 4.092e6 / 65536
# Detection threshold for this shape.  The reference's MIN_SIG = 16 (gps/gps.h:60) is sized for its
# 41 x 4092 trials per SV; 256 bins x 4092 (E1B: 16368) lags are 1.0 M (4.2 M) trials, whose noise
# maximum alone is ln(trials) + a few = 14..19 in units of the mean power.
MIN_SIG_10MS = 30.0
# (index into sats.SATS, code phase in chips, Doppler in Hz, carrier phase, C/N0 dB-Hz)
CONFIG4_PRESENT = [(0, 300.5, 1500.0, 0.7, 41.0), (6, 911.0, 7300.0, 0.1, 40.0), (13, 71.5, -250.0, 4.0, 39.0),
                   (21, 640.0, -7800.0, 3.3, 42.0), (29, 222.625, -3300.0, 1.2, 40.0),
                   (32, 800.5, 2600.0, 1.0, 41.0),                     # QZSS 194
                   (37, 1500.25, 3900.0, 2.2, 42.0), (45, 3333.5, -5100.0, 5.0, 43.0),   # E1B E03, E13
                   (58, 12.0, 60.0, 0.3, 42.0)]                        # E1B E36


def all_sv_codes(e1b):
    """[(chips, boc)] for every row of sats.SATS (36 C/A + QZSS rows, 23 E1B rows).
    e1b: {prn: chips uint8[4092]} -- the Galileo E1-B memory codes are ICD data the CALLER hands over
    (tests and bench.py take them from tests/fixtures.py)."""
    out = []
    for prn, t1, t2, kind in _sats.SATS:
        out.append((e1b[prn], True) if kind == _sats.E1B else (_prn.cacode(t1, t2), False))
    return out


def config4_iq16(codes, seed=0x5EED0005, present=CONFIG4_PRESENT):
    """One 10 ms block of complex int16 IF samples holding the SVs of CONFIG4_PRESENT
    (codes = all_sv_codes(e1b))."""
    svs = [(codes[sat][0], tau, fd, th, cn0, codes[sat][1]) for sat, tau, fd, th, cn0 in present]
    return gps_scene_iq16(svs, seed, n=NSAMPLES_10MS)


def wf_iq_frame(seed, tones=((0.05, -20.0), (0.21, -55.0), (0.33, -80.0)), noise_dbfs=-70.0,
                n=8192):
    """One waterfall DDC buffer: n complex int16 samples {i, q} (struct iq_t of
    rx/rx_waterfall.cpp:95-97).  tones: (cycles per sample, dBFS); noise rms in dBFS."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * (10.0 ** (noise_dbfs / 20.0) / np.sqrt(2))
    for f, dbfs in tones:
        x = x + 10.0 ** (dbfs / 20.0) * np.exp(2j * np.pi * (f * t + rng.random()))
    iq = np.empty((n, 2), np.int16)
    iq[:, 0] = np.clip(np.rint(32767.0 * x.real), -32768, 32767)
    iq[:, 1] = np.clip(np.rint(32767.0 * x.imag), -32768, 32767)
    return iq


def adc_stream(n, seed):
    """n samples of the synthetic 16-bit ADC stream of BASELINE configs[2] / [3] (bench.py, tests): four CW tones at
    0.0123 / 0.071 / 0.2003 / 0.31 of the sample rate (3000 ... 3 LSB) in Gaussian noise of 10 LSB rms."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n, dtype=np.float64)
    x = rng.normal(0, 10.0, n)
    for f, a in ((0.0123, 3000.0), (0.071, 300.0), (0.2003, 30.0), (0.31, 3.0)):
        x += a * np.cos(2 * np.pi * f * t)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)
