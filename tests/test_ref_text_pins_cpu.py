"""The literal tables of the reference files that cannot be compiled here (they include <fftw3.h>, absent from the image),
pinned by TEXT: the numbers are parsed from /root/reference at test time, in the build container, and compared with what the
oracle and the product compute from them -- as tests/test_ref_pins_cpu.py does for fir_iq.sv's taps.  Nothing of the reference
is kept under tests/: on a box without the tree the whole file skips.

  gps/search.cpp:100-136    COEF[31][2], column FT      -> the oracle's half-band decimator (impulse response),
                                                           kg_acq.hip's c_hb_even / HB_CENTRE
  gps/search.cpp:383-384    lo_sin / lo_cos             -> oracle/kiwi_oracle.c, kg_acq.hip's mixer
  rx/rx_waterfall.cpp:136-171 window constants          -> ko.wf_window, wf.window_functions (recomputed from the parsed constants)
  rx/rx_waterfall.cpp:175-185 CIC_comp p1 / p2, scaling -> ko.wf_cic_comp, wf.cic_comp_table
  rx/CuteSDR/fastfir.cpp:61-95, 102-146  CIC p1 / p2 (both rates), the five window functions' constants
                                                        -> ko.fir_window / ko.fir_cic_coeffs, kg_snd.hip's literals
"""
import os
import re

import numpy as np
import pytest

REFERENCE = os.environ.get("REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "gps", "search.cpp")), reason="reference tree not present")

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def ref_lines(rel, lo, hi):
    return "".join(open(os.path.join(REFERENCE, rel)).read().splitlines(True)[lo - 1:hi])


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def our(rel):
    return open(os.path.join(ROOT, rel)).read()


def test_half_band_taps_are_search_cpp_coef_column_ft(oracle):
    text = open(os.path.join(REFERENCE, "gps", "search.cpp")).read()
    ft = int(re.search(r"#define FT\s+(\d+)", text).group(1))
    ntaps = int(re.search(r"#define NTAPS\s+(\d+)", text).group(1))
    body = re.search(r"static float COEF\[NTAPS\]\[2\][^=]*=\s*\{(.*?)\}\s*;", text, re.S).group(1)
    vals = [float(v) for v in re.findall(NUM, strip_comments(body))]
    assert ft == 0 and ntaps == 31 and len(vals) == 62
    coef = np.array(vals, np.float32).reshape(31, 2)[:, ft]
    assert np.array_equal(coef, coef[::-1]) and np.all(coef[1:15:2] == 0) and coef[15] == np.float32(0.500009)
    # the oracle: y[o] = sum_j c[j] x[2o + j] (search.cpp:140-166) -- an impulse at x[30] reads the even taps, at x[31] the odd ones
    for at in (30, 31):
        x = np.zeros(64, np.complex64)
        x[at] = 1.0
        y = oracle.decimate_by2(x).real
        want = np.array([coef[at - 2 * o] if 0 <= at - 2 * o < 31 else 0.0 for o in range(y.size)], np.float32)
        assert np.array_equal(y, want), at
    # the kernel's constant table (kg_acq.hip): the 16 even taps and the centre
    src = our("flydog_sdr_gps_amd/csrc/kg_acq.hip")
    even = [float(v.rstrip("f")) for v in re.findall(NUM + "f", re.search(r"c_hb_even\[16\]\s*=\s*\{(.*?)\};", src, re.S).group(1))]
    centre = float(re.search(r"#define HB_CENTRE\s+(%s)f" % NUM, src).group(1))
    assert np.array_equal(np.array(even, np.float32), coef[0::2]) and np.float32(centre) == coef[15]
    # and the oracle's own literal table
    osrc = our("oracle/kiwi_oracle.c")
    tab = [float(v.rstrip("f")) for v in re.findall(NUM + "f", re.search(r"HB_COEF\[KO_NTAPS\]\s*=\s*\{(.*?)\};", osrc, re.S).group(1))]
    assert np.array_equal(np.array(tab, np.float32), coef)


def test_quadrature_oscillator_tables_are_search_cpp_lo_sin_lo_cos(oracle):
    text = ref_lines("gps/search.cpp", 380, 392)
    lo_sin = [int(v) for v in re.search(r"lo_sin\[\]\s*=\s*\{([^}]*)\}", text).group(1).split(",")]
    lo_cos = [int(v) for v in re.search(r"lo_cos\[\]\s*=\s*\{([^}]*)\}", text).group(1).split(",")]
    assert (lo_sin, lo_cos) == ([1, 1, 0, 0], [1, 0, 0, 1])
    osrc = our("oracle/kiwi_oracle.c")
    assert [int(v) for v in re.search(r"lo_sin\[4\]\s*=\s*\{([^}]*)\}", osrc).group(1).split(",")] == lo_sin
    assert [int(v) for v in re.search(r"lo_cos\[4\]\s*=\s*\{([^}]*)\}", osrc).group(1).split(",")] == lo_cos
    # behaviour: all-zero bits mix to I = lo_sin, Q = lo_cos (search.cpp:419-420), Bipolar(1) = -1; the decimators are linear and
    # symmetric, so the decimated block of a constant-bit input is the filtered oscillator -- the same for the host mirror
    td0 = oracle.sample_bits(np.zeros(8192, np.uint8), want_td=True)[1]
    td1 = oracle.sample_bits(np.full(8192, 0xFF, np.uint8), want_td=True)[1]
    assert np.array_equal(td0[40:-40], -td1[40:-40])                  # bit ^ lo: all ones is the negated oscillator
    ph = np.arange(65536) % 4                                          # lo_rate = 4 FC / FS = 1.0 exactly (:386)
    i0 = np.where(np.array(lo_sin)[ph] == 1, -1.0, 1.0)
    q0 = np.where(np.array(lo_cos)[ph] == 1, -1.0, 1.0)
    want = oracle.decimate_by2(oracle.decimate_by2((i0 + 1j * q0).astype(np.complex64)))
    assert np.array_equal(td0, want)


_libm = []


def _cosf(x):
    if not _libm:
        import ctypes
        import ctypes.util
        m = ctypes.CDLL(ctypes.util.find_library("m"))
        m.cosf.argtypes, m.cosf.restype = [ctypes.c_float], ctypes.c_float
        _libm.append(m)
    return float(_libm[0].cosf(x))


def window_from_constants(consts, n, denom, f32_cos):
    """sum_k (-1)^k a_k cos(k 2 pi i / denom): the reference's expression, evaluated its way (double, or float cosines)."""
    i = np.arange(n, dtype=np.float64)
    K_2PI = 2.0 * 3.14159265358979323846
    acc = np.full(n, consts[0], np.float64)
    for k, a in enumerate(consts[1:], 1):
        arg = ((1.0 if k == 1 else float(k)) * K_2PI * i) / denom
        c = np.array([_cosf(float(np.float32(v))) for v in arg]) if f32_cos else np.cos(arg)      # MCOS = cosf of the C library
        acc = acc + (-a if k % 2 else a) * c
    return acc


def test_waterfall_windows_and_cic_comp_are_rx_waterfall_cpp_constants(oracle):
    from flydog_sdr_gps_amd import wf
    text = strip_comments(ref_lines("rx/rx_waterfall.cpp", 128, 190))
    scale = float(re.search(r"adc_scale_decim = powf\(2, (-?\d+)\)", text).group(1))
    gain = float(re.search(r"#define WINDOW_GAIN\s+(%s)" % NUM, text).group(1))
    cases = re.split(r"case WINF_WF_", text)
    consts = {}
    for c in cases[1:]:
        name = re.match(r"(\w+)", c).group(1)
        body = c.split("break;")[0]
        vals = [float(v) for v in re.findall(r"(?<![\w.])(0\.\d+)(?![\w.])", body)]
        consts[name] = vals
    assert consts["HANNING"] == [0.5, 0.5] and consts["HAMMING"] == [0.54, 0.46]
    assert consts["BLACKMAN_HARRIS"] == [0.35875, 0.48829, 0.14128, 0.01168] and scale == -16 and gain == 1.0
    base = np.float32(2.0 ** scale * gain)
    denom = float(np.float32(8192 - 1))                                # (float)(WF_C_NSAMPS-1)
    got_o = [oracle.wf_window(k) for k in range(4)]
    got_p = wf.window_functions()
    for k, name in enumerate(("HANNING", "HAMMING", "BLACKMAN_HARRIS")):
        want = (float(base) * window_from_constants(consts[name], 8192, denom, False)).astype(np.float32)
        assert np.array_equal(got_o[k], want), name
        assert np.array_equal(got_p[k], want), name
    assert np.all(got_o[3] == base) and np.all(got_p[3] == base)
    # CIC_comp (:175-185)
    p1 = float(re.search(r"p1 = (%s)f" % NUM, text).group(1))
    p2 = float(re.search(r"p2 = (%s)f" % NUM, text).group(1))
    m = re.search(r"CIC_comp\[i\] = (%s) \+ cic_comp / (%s);" % (NUM, NUM), text)
    assert (p1, p2, float(m.group(1)), float(m.group(2))) == (-2.969, 36.26, 0.5, 2.0)
    assert "pow(sincf, -5)" in text
    # (the host mirror evaluates sinf / pow / exp with numpy instead of the C library: a few float steps, far inside the 1e-5 bar)
    assert np.allclose(wf.cic_comp_table(), oracle.wf_cic_comp(), rtol=2e-6, atol=0)
    osrc = our("oracle/kiwi_oracle_wf.c")
    assert ("p1 = %sf" % repr(p1)) in osrc and ("p2 = %sf" % repr(p2)) in osrc


def test_fastfir_windows_and_cic_constants_are_fastfir_cpp_text(oracle):
    text = strip_comments(ref_lines("rx/CuteSDR/fastfir.cpp", 61, 146))
    names = {"BLACKMAN_NUTTALL": 0, "BLACKMAN_HARRIS": 1, "NUTTALL": 2, "HANNING": 3, "HAMMING": 4}
    hdr = open(os.path.join(REFERENCE, "rx", "rx_sound.h")).read()
    for name, k in names.items():
        assert int(re.search(r"#define WINF_SND_%s\s+(\d+)" % name, hdr).group(1)) == k
    consts = {}
    for c in re.split(r"case WINF_SND_", text)[1:]:
        name = re.match(r"(\w+)", c).group(1)
        consts[name] = [float(v) for v in re.findall(r"(?<![\w.])(0\.\d+)(?![\w.])", c.split("break;")[0])]
    assert consts == {"BLACKMAN_NUTTALL": [0.3635819, 0.4891775, 0.1365995, 0.0106411],
                      "BLACKMAN_HARRIS": [0.35875, 0.48829, 0.14128, 0.01168],
                      "NUTTALL": [0.355768, 0.487396, 0.144232, 0.012604], "HANNING": [0.5, 0.5], "HAMMING": [0.54, 0.46]}
    assert "window_func = WINF_SND_BLACKMAN_NUTTALL" in text           # window_func < 0 (:105-106)
    for name, k in names.items():
        # m_pWindowTbl[i] = a0 - a1 MCOS((K_2PI*i)/(CONV_FIR_SIZE-1)) + ...: TYPEREAL float, MCOS = cosf, the sum in double
        want = window_from_constants(consts[name], 513, 512.0, True).astype(np.float32)
        assert np.array_equal(oracle.fir_window(k), want), name
    assert np.array_equal(oracle.fir_window(-1), oracle.fir_window(0))
    # the product designs its taps on the host with the same literals (kg_snd.hip): every constant, nothing else
    src = our("flydog_sdr_gps_amd/csrc/kg_snd.hip")
    body = src[src.index("SetupWindowFunction"):]
    body = body[:body.index("SetupCICFilter")] if "SetupCICFilter" in body else body[:4000]
    ours = sorted(set(float(v) for v in re.findall(r"(?<![\w.])(0\.\d+)(?![\w.])", strip_comments(body))))
    theirs = sorted(set(v for vals in consts.values() for v in vals))
    assert all(v in ours for v in theirs), (theirs, ours)
    # the CIC compensation constants of both sound rates (:70-73)
    p = re.search(r"p1 = \(snd_rate == SND_RATE_3CH \? (%s)f : (%s)f\)" % (NUM, NUM), text)
    q = re.search(r"p2 = \(snd_rate == SND_RATE_3CH \? (%s)f\s*: (%s)f\s*\)" % (NUM, NUM), text)
    assert [float(v) for v in p.groups() + q.groups()] == [-3.107, -2.969, 32.04, 36.26]
    for s in (our("oracle/kiwi_oracle_snd.c"), src):
        assert "-3.107f : -2.969f" in s and "32.04f : 36.26f" in s
