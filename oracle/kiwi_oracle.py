"""ctypes front for the CPU ORACLE (oracle/libkiwi_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (flydog_sdr_gps_amd) never
imports this module.  See oracle/kiwi_oracle.h for what each entry point
restates (reference file:line) and for the pinning status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkiwi_oracle.so")

NSAMPLES = 65536
FFT_LEN = 16384
DECIM = 4
NTAPS = 31
L1_CODELEN = 1023
E1B_CODELEN = 4092
L1_LIMIT = 4092          # SAMPLE_RATE/1000 * L1_CODE_PERIOD, gps/search.cpp:486
E1B_LIMIT = 16368        # SAMPLE_RATE/1000 * E1B_CODE_PERIOD
DOP_LO, DOP_HI = -20, 20

cpx = np.complex64


class AcqResult(C.Structure):
    _fields_ = [("snr", C.c_float), ("dop", C.c_int), ("idx", C.c_int), ("valid", C.c_int)]


class AcqCell(C.Structure):
    _fields_ = [("snr", C.c_float), ("max_pwr", C.c_float), ("tot_pwr", C.c_float),
                ("idx", C.c_int)]


cell_dtype = np.dtype([("snr", "<f4"), ("max_pwr", "<f4"), ("tot_pwr", "<f4"), ("idx", "<i4")])
result_dtype = np.dtype([("snr", "<f4"), ("dop", "<i4"), ("idx", "<i4"), ("valid", "<i4")])


def build(force=False):
    """(Re)build libkiwi_oracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # KIWI_ORACLE_LIBRARY: another build of the same sources (the sanitizer build, oracle/Makefile `asan`;
        # bench.py's -march=native build for the cpu_baseline leg)
        L = C.CDLL(os.environ.get("KIWI_ORACLE_LIBRARY") or _LIB_PATH)
        vp = C.c_void_p
        L.ko_cacode.argtypes = [C.c_int, C.c_int, vp]
        L.ko_e1b_from_hex.argtypes = [C.c_char_p, vp]
        L.ko_e1b_from_hex.restype = C.c_int
        L.ko_fft.argtypes = [C.c_int, C.c_int, vp, vp, C.c_int]
        L.ko_set_fft_hook.argtypes = [vp]
        L.ko_decimate_by2_float.argtypes = [C.c_int, vp]
        L.ko_decimate_by2_float.restype = C.c_int
        L.ko_code_fft.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), vp, C.c_int]
        L.ko_code_replica.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), vp]
        L.ko_sample_bits.argtypes = [vp, vp, vp, C.c_int]
        L.ko_sample_iq16.argtypes = [vp, vp, vp, C.c_int]
        L.ko_correlate.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.ko_correlate.restype = AcqResult
        L.ko_correlate_many.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp,
                                        C.c_int, C.c_int]
        L.ko_code_fft_n.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), vp, C.c_int, C.c_int]
        L.ko_code_replica_n.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), vp, C.c_int]
        L.ko_sample_bits_n.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
        L.ko_sample_iq16_n.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
        L.ko_correlate_n.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
        L.ko_correlate_n.restype = AcqResult
        L.ko_correlate_many_n.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp,
                                          C.c_int, C.c_int, C.c_int]
        L.ko_wf_window.argtypes = [C.c_int, vp]
        L.ko_wf_cic_comp.argtypes = [vp]
        L.ko_wf_params_for.argtypes = [C.c_int, C.c_float, C.c_double, C.c_double, C.c_int, vp]
        L.ko_wf_build_maps.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.ko_wf_window_iq.argtypes = [vp, vp, vp]
        L.ko_wf_compute_frame.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cacode(t0, t1):
    out = np.zeros(L1_CODELEN, np.uint8)
    lib().ko_cacode(int(t0), int(t1), _p(out))
    return out


def e1b_from_hex(hexstr):
    out = np.zeros(E1B_CODELEN, np.uint8)
    rc = lib().ko_e1b_from_hex(hexstr.encode("ascii"), _p(out))
    if rc != 0:
        raise ValueError("bad hex digit in E1B memory code")
    return out


def fft(x, sign=-1, prec=1):
    x = np.ascontiguousarray(x, cpx)
    out = np.empty_like(x)
    lib().ko_fft(x.size, int(sign), _p(x), _p(out), int(prec))
    return out


_FFT_HOOK_T = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
_fft_hook_keep = []


def set_fft_hook(fn):
    """prec=2 of every entry point then transforms with fn(x: complex64[n], sign) -> complex64[n] (unnormalised;
    sign -1 forward, +1 backward): how the tests run the restated reference loops over an FFT that is not the
    oracle's own.  None removes the hook."""
    if fn is None:
        lib().ko_set_fft_hook(None)
        _fft_hook_keep.clear()
        return

    def tramp(n, sign, pin, pout):
        x = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_float)), shape=(2 * n,)).view(cpx).copy()
        y = np.ascontiguousarray(fn(x, sign), cpx)
        C.memmove(pout, y.ctypes.data, 8 * n)
    cb = _FFT_HOOK_T(tramp)
    _fft_hook_keep[:] = [cb]
    lib().ko_set_fft_hook(C.cast(cb, C.c_void_p))


def decimate_by2(x):
    """DecimateBy2float on a copy; returns the size/2 outputs."""
    x = np.asarray(x, cpx)
    buf = np.zeros(x.size + NTAPS, cpx)
    buf[:x.size] = x
    n = lib().ko_decimate_by2_float(x.size, _p(buf))
    return buf[:n].copy()


# Shape arguments (nsamples, fft_len): the reference's (65536, 16384) by default; the _n entry
# points restate the same loops for BASELINE configs[4]'s (163680, 65536).
NSAMPLES_10MS, FFT_LEN_10MS = 163680, 65536


def code_fft(chips, boc=False, prec=1, phase=0.0, fft_len=FFT_LEN):
    chips = np.ascontiguousarray(chips, np.uint8)
    out = np.empty(fft_len, cpx)
    ph = C.c_float(phase)
    lib().ko_code_fft_n(_p(chips), chips.size, int(bool(boc)), C.byref(ph), _p(out), int(prec), int(fft_len))
    return out


def code_replica(chips, boc=False, phase=0.0, fft_len=FFT_LEN):
    chips = np.ascontiguousarray(chips, np.uint8)
    out = np.empty(fft_len, cpx)
    ph = C.c_float(phase)
    lib().ko_code_replica_n(_p(chips), chips.size, int(bool(boc)), C.byref(ph), _p(out), int(fft_len))
    return out, ph.value


def sample_bits(packed, prec=1, want_td=False, nsamples=NSAMPLES, fft_len=FFT_LEN):
    packed = np.ascontiguousarray(packed, np.uint8)
    assert packed.size == nsamples // 8 and nsamples <= DECIM * fft_len
    out = np.empty(fft_len, cpx)
    td = np.empty(fft_len, cpx) if want_td else None
    lib().ko_sample_bits_n(_p(packed), _p(out), _p(td) if want_td else None, int(prec), int(nsamples),
                           int(fft_len))
    return (out, td) if want_td else out


def sample_iq16(iq, prec=1, want_td=False, nsamples=NSAMPLES, fft_len=FFT_LEN):
    iq = np.ascontiguousarray(iq, np.int16).reshape(-1)
    assert iq.size == 2 * nsamples and nsamples <= DECIM * fft_len
    out = np.empty(fft_len, cpx)
    td = np.empty(fft_len, cpx) if want_td else None
    lib().ko_sample_iq16_n(_p(iq), _p(out), _p(td) if want_td else None, int(prec), int(nsamples),
                           int(fft_len))
    return (out, td) if want_td else out


def correlate(code, data, limit=L1_LIMIT, dop_lo=DOP_LO, dop_hi=DOP_HI, prec=1, code_next=None):
    """-> (result dict, cells structured array[dop_hi-dop_lo+1]); the transform length is code.size.
    code_next: the spectrum in the NEXT row of the reference's code[][] table (the satellite numbered one higher), which a
    negative Doppler bin reads into (ko_correlate_next_n); None = a row never written (zeros)."""
    code = np.ascontiguousarray(code, cpx)
    data = np.ascontiguousarray(data, cpx)
    assert code.size == data.size
    cells = np.zeros(dop_hi - dop_lo + 1, cell_dtype)
    L = lib()
    L.ko_correlate_next_n.restype = AcqResult
    nxt = None if code_next is None else np.ascontiguousarray(code_next, cpx)
    assert nxt is None or nxt.size == code.size
    r = L.ko_correlate_next_n(_p(code), _p(nxt) if nxt is not None else None, _p(data), C.c_int(int(limit)), C.c_int(dop_lo),
                              C.c_int(dop_hi), _p(cells), C.c_int(int(prec)), C.c_int(int(code.size)))
    return dict(snr=r.snr, dop=r.dop, idx=r.idx, valid=r.valid), cells


def correlate_many(codes, data, limits, dop_lo=DOP_LO, dop_hi=DOP_HI, prec=1, nthreads=1,
                   want_cells=True, nexts=None):
    """nexts: per SV the spectrum of the next row of the reference's table (see correlate), None entries / None = never written."""
    codes = np.ascontiguousarray(codes, cpx)
    data = np.ascontiguousarray(data, cpx)
    nsv = codes.shape[0]
    assert codes.shape[1] == data.size
    limits = np.ascontiguousarray(limits, np.int32)
    nd = dop_hi - dop_lo + 1
    out = np.zeros(nsv, result_dtype)
    cells = np.zeros((nsv, nd), cell_dtype) if want_cells else None
    nx, has = None, None
    if nexts is not None:
        assert len(nexts) == nsv
        has = np.array([0 if n is None else 1 for n in nexts], np.uint8)
        nx = np.zeros(codes.shape, cpx)
        for i, n in enumerate(nexts):
            if n is not None:
                nx[i] = n
    lib().ko_correlate_many_next_n(_p(codes), _p(nx) if nx is not None else None, _p(has) if has is not None else None, C.c_int(nsv),
                                   _p(data), _p(limits), C.c_int(dop_lo), C.c_int(dop_hi), _p(out), _p(cells) if want_cells else None,
                                   C.c_int(int(prec)), C.c_int(int(nthreads)), C.c_int(int(data.size)))
    return out, cells


# ---- waterfall -----------------------------------------------------------------
WF_NFFT, WF_WIDTH = 8192, 1024


class WfParams(C.Structure):
    _fields_ = [("zoom", C.c_int), ("decim", C.c_int), ("fft_used", C.c_int),
                ("plot_width", C.c_int), ("plot_width_clamped", C.c_int), ("start", C.c_float),
                ("fft_scale", C.c_float), ("fft_offset", C.c_float), ("i_offset", C.c_uint64)]


class WfCfg(C.Structure):
    _fields_ = [("zoom", C.c_int), ("window_func", C.c_int), ("interp", C.c_int),
                ("cic_comp", C.c_int), ("overlapped", C.c_int), ("fft_used", C.c_int),
                ("plot_width", C.c_int), ("plot_width_clamped", C.c_int),
                ("fft2wf_map", C.c_void_p), ("drop_sample", C.c_void_p),
                ("fft_scale", C.c_void_p), ("fft_scale_div2", C.c_void_p),
                ("fft_offset", C.c_float), ("CIC_comp", C.c_void_p)]


def wf_window(winf):
    out = np.empty(WF_NFFT, np.float32)
    lib().ko_wf_window(int(winf), _p(out))
    return out


def wf_cic_comp():
    out = np.empty(WF_NFFT, np.float32)
    lib().ko_wf_cic_comp(_p(out))
    return out


def wf_params(zoom, start, adc_clock=125.0e6, ui_srate=32.0e6, spectral_inversion=False):
    p = WfParams()
    lib().ko_wf_params_for(int(zoom), float(start), float(adc_clock), float(ui_srate),
                           int(bool(spectral_inversion)), C.byref(p))
    return p


def wf_build_maps(fft_used, plot_width, plot_width_clamped, spectral_inversion=False):
    m = np.zeros(fft_used, np.uint16)
    d = np.zeros(WF_WIDTH, np.uint16)
    lib().ko_wf_build_maps(fft_used, plot_width, plot_width_clamped, int(bool(spectral_inversion)),
                           _p(m), _p(d))
    return m, d


def wf_window_iq(iq, window):
    iq = np.ascontiguousarray(iq, np.int16).reshape(-1)
    window = np.ascontiguousarray(window, np.float32)
    out = np.empty(WF_NFFT, cpx)
    lib().ko_wf_window_iq(_p(iq), _p(window), _p(out))
    return out


def wf_compute_frame(samps, zoom, window_func, interp, cic_comp_on, overlapped, fft_used, plot_width,
                     plot_width_clamped, fft2wf_map, drop_sample, fft_scale, fft_scale_div2,
                     fft_offset, cic_table, prec=1):
    """-> (out u8[1024], pwr[fft_used], pwr_out[1024], dB[1024])"""
    samps = np.ascontiguousarray(samps, cpx)
    keep = [np.ascontiguousarray(fft2wf_map, np.uint16), np.ascontiguousarray(drop_sample, np.uint16),
            np.ascontiguousarray(fft_scale, np.float32), np.ascontiguousarray(fft_scale_div2, np.float32),
            np.ascontiguousarray(cic_table, np.float32)]
    cfg = WfCfg(zoom, window_func, interp, int(cic_comp_on), int(overlapped), fft_used, plot_width,
                plot_width_clamped, keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data,
                keep[3].ctypes.data, float(fft_offset), keep[4].ctypes.data)
    out = np.empty(WF_WIDTH, np.uint8)
    pwr = np.zeros(fft_used, np.float32)
    pwr_out = np.empty(WF_WIDTH, np.float32)
    dB = np.empty(WF_WIDTH, np.float32)
    lib().ko_wf_compute_frame(C.byref(cfg), _p(samps), _p(out), _p(pwr), _p(pwr_out), _p(dB), int(prec))
    return out, pwr, pwr_out, dB


# ---- audio front ------------------------------------------------------------------
FIR_FFT_SIZE, FIR_SIZE = 1024, 513


class FirState(C.Structure):
    _fields_ = [("in_pos", C.c_int), ("buf", C.c_float * (2 * 1024)), ("overlap", C.c_float * (2 * 513))]


def dpump_rescale(use_cicf=True):
    L = lib()
    L.ko_dpump_rescale.restype = C.c_float
    L.ko_dpump_rescale.argtypes = [C.c_int]
    return L.ko_dpump_rescale(int(use_cicf))


def dpump_unpack(raw, nsamps, nchans, enabled=None, rescale=None, dc_i=0.0, dc_q=0.0,
                 spectral_inversion=False):
    L = lib()
    L.ko_dpump_unpack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float,
                                  C.c_float, C.c_int, C.c_void_p, C.c_int]
    raw = np.ascontiguousarray(raw, np.uint8)
    en = np.ones(nchans, np.uint8) if enabled is None else np.ascontiguousarray(enabled, np.uint8)
    out = np.zeros((nchans, nsamps), cpx)
    L.ko_dpump_unpack(_p(raw), nsamps, nchans, _p(en), dpump_rescale() if rescale is None else rescale,
                      dc_i, dc_q, int(bool(spectral_inversion)), _p(out), nsamps)
    return out


def fir_cic_coeffs(snd_rate_3ch=False):
    out = np.empty(FIR_FFT_SIZE, np.float32)
    L = lib(); L.ko_fir_cic_coeffs.argtypes = [C.c_int, C.c_void_p]
    L.ko_fir_cic_coeffs(int(snd_rate_3ch), _p(out))
    return out


def fir_window(window_func=-1):
    out = np.empty(FIR_SIZE, np.float32)
    L = lib(); L.ko_fir_window.argtypes = [C.c_int, C.c_void_p]
    L.ko_fir_window(int(window_func), _p(out))
    return out


def fir_design(lo, hi, offset, fs, window=None, do_cic_comp=False, cic=None, prec=1):
    """-> (coef[1024], coef_cic[1024], time_coef[1024]) or None when rejected (:193-200)."""
    L = lib()
    L.ko_fir_design.argtypes = [C.c_float] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_int]
    L.ko_fir_design.restype = C.c_int
    window = fir_window() if window is None else np.ascontiguousarray(window, np.float32)
    cic = fir_cic_coeffs() if cic is None else np.ascontiguousarray(cic, np.float32)
    coef, coef_cic, tcoef = (np.empty(FIR_FFT_SIZE, cpx) for _ in range(3))
    rc = L.ko_fir_design(lo, hi, offset, fs, _p(window), int(do_cic_comp), _p(cic), _p(coef),
                         _p(coef_cic), _p(tcoef), int(prec))
    return None if rc else (coef, coef_cic, tcoef)


def fir_new_state():
    L = lib(); L.ko_fir_reset.argtypes = [C.c_void_p]
    st = FirState()
    L.ko_fir_reset(C.byref(st))
    return st


def fir_process(state, coef_cic, x, prec=1):
    L = lib()
    L.ko_fir_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.ko_fir_process.restype = C.c_int
    x = np.ascontiguousarray(x, cpx)
    coef_cic = np.ascontiguousarray(coef_cic, cpx)
    out = np.empty(x.size + 1024, cpx)
    n = L.ko_fir_process(C.byref(state), _p(coef_cic), _p(x), x.size, _p(out), int(prec))
    return out[:n].copy(), state.in_pos - (FIR_SIZE - 1)          # FirPos(), fastfir.h:33


def ddc_shape(which, r=0):
    """Structural constants of the pruned CICs as the oracle uses them (see ko_ddc_shape)."""
    o = (C.c_int * 64)()
    n = lib().ko_ddc_shape(C.c_int(which), C.c_int(r), o)
    return list(o[:n])


# ---- S-meter, CAgc, AM / NBFM detectors (kiwi_oracle_post.c) -----------------------
class Agc:
    """One CAgc instance (agc.cpp).  State lives in an opaque buffer of ko_agc_state_size()."""

    def __init__(self):
        L = lib()
        L.ko_agc_state_size.restype = C.c_size_t
        self._buf = C.create_string_buffer(L.ko_agc_state_size())
        L.ko_agc_init(self._buf)

    def set_parameters(self, agc_on, use_hang, threshold, manual_gain, slope, decay, sample_rate):
        L = lib()
        L.ko_agc_set_parameters.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_float]
        L.ko_agc_set_parameters(self._buf, int(agc_on), int(use_hang), int(threshold), int(manual_gain),
                                int(slope), int(decay), float(sample_rate))

    def delay(self):
        return int(lib().ko_agc_delay(self._buf))

    def process_cpx(self, x):
        x = np.ascontiguousarray(x, cpx)
        out = np.empty(x.size, cpx)
        lib().ko_agc_process_cpx(self._buf, C.c_int(x.size), _p(x), _p(out))
        return out

    def process_s16(self, x):
        x = np.ascontiguousarray(x, cpx)
        out = np.empty(x.size, np.int16)
        lib().ko_agc_process_s16(self._buf, C.c_int(x.size), _p(x), _p(out))
        return out


def smeter_alpha(frate):
    L = lib()
    L.ko_smeter_alpha.argtypes = [C.c_float]
    L.ko_smeter_alpha.restype = C.c_float
    return float(L.ko_smeter_alpha(float(frate)))


def smeter_process(avg_dB, alpha, x):
    """-> (new average, (value at j == 0, value at j == n/2))"""
    L = lib()
    L.ko_smeter_process.argtypes = [C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    L.ko_smeter_process.restype = C.c_float
    x = np.ascontiguousarray(x, cpx)
    tap = np.zeros(2, np.float32)
    avg = L.ko_smeter_process(float(avg_dB), float(alpha), x.size, _p(x), _p(tap))
    return float(avg), (float(tap[0]), float(tap[1]))


def am_detect(z1, agc):
    """-> (demod float32[n], new z1)"""
    agc = np.ascontiguousarray(agc, cpx)
    z = C.c_double(z1)
    out = np.empty(agc.size, np.float32)
    lib().ko_am_detect(C.byref(z), C.c_int(agc.size), _p(agc), _p(out))
    return out, z.value


def nbfm_detect(last, agc):
    """last = (re, im) of conn->last_sample -> (demod float32[n], new last)"""
    agc = np.ascontiguousarray(agc, cpx)
    l = np.array([complex(last[0], last[1])], cpx)
    out = np.empty(agc.size, np.float32)
    lib().ko_nbfm_detect(_p(l), C.c_int(agc.size), _p(agc), _p(out))
    return out, (float(l[0].real), float(l[0].imag))


# ---- CFir, CSquelch (kiwi_oracle_cfir.c) ---------------------------------------------
class CFir:
    """One CFir instance (rx/CuteSDR/fir.cpp), real-valued paths."""

    def __init__(self):
        L = lib()
        L.ko_cfir_state_size.restype = C.c_size_t
        self._buf = C.create_string_buffer(L.ko_cfir_state_size())
        L.ko_cfir_init(self._buf)

    def init_const(self, coef, fs=12000.0):
        coef = np.ascontiguousarray(coef, np.float32)
        L = lib()
        L.ko_cfir_init_const.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float]
        L.ko_cfir_init_const(self._buf, coef.size, _p(coef), float(fs))

    def _design(self, fn, numtaps, scale, astop, fpass, fstop, fs):
        fn.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 5
        fn.restype = C.c_int
        return int(fn(self._buf, int(numtaps), float(scale), float(astop), float(fpass), float(fstop), float(fs)))

    def init_lp(self, numtaps, scale, astop, fpass, fstop, fs):
        return self._design(lib().ko_cfir_init_lp, numtaps, scale, astop, fpass, fstop, fs)

    def init_hp(self, numtaps, scale, astop, fpass, fstop, fs):
        return self._design(lib().ko_cfir_init_hp, numtaps, scale, astop, fpass, fstop, fs)

    def taps(self):
        L = lib()
        n = int(L.ko_cfir_num_taps(self._buf))
        t = np.empty(n, np.float32)
        L.ko_cfir_get_taps(self._buf, _p(t))
        return t

    def process_rr(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.size, np.float32)
        lib().ko_cfir_process_rr(self._buf, C.c_int(x.size), _p(x), _p(out))
        return out

    def process_rm(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.size, np.int16)
        lib().ko_cfir_process_rm(self._buf, C.c_int(x.size), _p(x), _p(out))
        return out

    def process_mm(self, x):
        x = np.ascontiguousarray(x, np.int16)
        out = np.empty(x.size, np.int16)
        lib().ko_cfir_process_mm(self._buf, C.c_int(x.size), _p(x), _p(out))
        return out


class Squelch:
    """One CSquelch instance (rx/CuteSDR/squelch.cpp), the NBFM noise squelch."""

    def __init__(self):
        L = lib()
        L.ko_squelch_state_size.restype = C.c_size_t
        self._buf = C.create_string_buffer(L.ko_squelch_state_size())
        L.ko_squelch_init(self._buf)

    def setup(self, rate):
        L = lib()
        L.ko_squelch_setup.argtypes = [C.c_void_p, C.c_float]
        L.ko_squelch_setup(self._buf, float(rate))

    def set_squelch(self, value, squelch_max):
        lib().ko_squelch_set(self._buf, C.c_int(int(value)), C.c_int(int(squelch_max)))

    def reset(self):
        lib().ko_squelch_reset(self._buf)

    def squelched(self):
        return bool(lib().ko_squelch_is_squelched(self._buf))

    def ave(self):
        L = lib()
        L.ko_squelch_ave.restype = C.c_float
        return float(L.ko_squelch_ave(self._buf))

    def perform_fm(self, x):
        """-> (mono16 out, nsq_nc_sq)"""
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(x.size, np.int16)
        L = lib()
        L.ko_squelch_perform_fm.restype = C.c_int
        rc = L.ko_squelch_perform_fm(self._buf, C.c_int(x.size), _p(x), _p(out))
        return out, int(rc)


# ---- wire formats (kiwi_oracle_wire.c) ----------------------------------------------
class AdpcmState(C.Structure):
    _fields_ = [("index", C.c_int), ("previous", C.c_int)]


def adpcm_encode_i16(x, state=None):
    """encode_ima_adpcm_i16_e8 -> (bytes uint8[n/2], state)"""
    st = state if state is not None else AdpcmState(0, 0)
    x = np.ascontiguousarray(x, np.int16)
    out = np.empty(x.size // 2, np.uint8)
    lib().ko_adpcm_encode_i16(_p(x), _p(out), C.c_int(x.size), C.byref(st))
    return out, st


def adpcm_encode_u8(x, state=None):
    st = state if state is not None else AdpcmState(0, 0)
    x = np.ascontiguousarray(x, np.uint8)
    out = np.empty(x.size // 2, np.uint8)
    lib().ko_adpcm_encode_u8(_p(x), _p(out), C.c_int(x.size), C.byref(st))
    return out, st


def adpcm_decode_i16(b, state=None):
    st = state if state is not None else AdpcmState(0, 0)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.empty(b.size * 2, np.int16)
    lib().ko_adpcm_decode_i16(_p(b), _p(out), C.c_int(b.size), C.byref(st))
    return out, st


def adpcm_decode_u8(b, state=None):
    st = state if state is not None else AdpcmState(0, 0)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.empty(b.size * 2, np.uint8)
    lib().ko_adpcm_decode_u8(_p(b), _p(out), C.c_int(b.size), C.byref(st))
    return out, st


def adpcm_step_table():
    L = lib()
    L.ko_adpcm_step_table.restype = C.POINTER(C.c_int * 89)
    return np.array(L.ko_adpcm_step_table().contents[:], np.int32)


def wf_packet(row, x_bin_server, zoom, seq, use_compression):
    L = lib()
    L.ko_wf_packet.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.ko_wf_packet.restype = C.c_int
    row = np.ascontiguousarray(row, np.uint8)
    assert row.size == 1024
    pkt = np.zeros(16 + 1034, np.uint8)
    n = L.ko_wf_packet(_p(row), int(x_bin_server), int(zoom), int(seq), int(bool(use_compression)), _p(pkt))
    return pkt[:n].copy()


def snd_iq_payload(x, little_endian):
    """complex64[n] -> uint8[4n]: the IQ modes' sound payload (rx_sound.cpp:1076-1096)"""
    x = np.ascontiguousarray(x, cpx)
    out = np.zeros(4 * x.size, np.uint8)
    lib().ko_snd_iq_payload(_p(x), C.c_int(x.size), C.c_int(int(bool(little_endian))), _p(out))
    return out


def snd_header(flags, seq, smeter_dBm):
    L = lib()
    L.ko_snd_header.argtypes = [C.c_uint8, C.c_uint32, C.c_float, C.c_void_p]
    h = np.zeros(10, np.uint8)
    L.ko_snd_header(int(flags), int(seq) & 0xFFFFFFFF, float(smeter_dBm), _p(h))
    return h


class GpsState(C.Structure):
    _fields_ = [("gpssec", C.c_double), ("last_gpssec", C.c_double), ("gps_init", C.c_int), ("pad", C.c_int)]


def gps_begin(st, clk_gps_secs, dticks, adc_clock_base, gps_delay, gps_delay2):
    L = lib()
    L.ko_snd_gps_begin.argtypes = [C.c_void_p] + [C.c_double] * 5
    L.ko_snd_gps_begin(C.byref(st), clk_gps_secs, dticks, adc_clock_base, gps_delay, gps_delay2)


def gps_stamp(st, norm_nrx_samps, fir_pos, agc_on, agc_delay, rx_decim, adc_clock_base, clk_gps_secs, clk_ticks):
    L = lib()
    L.ko_snd_gps_stamp.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_double, C.c_double, C.c_uint64] + [C.c_void_p] * 3
    a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint8()
    L.ko_snd_gps_stamp(C.byref(st), norm_nrx_samps, fir_pos, int(bool(agc_on)), agc_delay, rx_decim, adc_clock_base,
                       clk_gps_secs, clk_ticks, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


# ---- hand-off arithmetic, waterfall autoscale (kiwi_oracle_handoff.c) ----------------
class ChanStart(C.Structure):
    _fields_ = [("lo_dop", C.c_double), ("ca_dop", C.c_double), ("lo_rate", C.c_uint32), ("ca_rate", C.c_uint32),
                ("ca_pause", C.c_uint32), ("code_creep", C.c_int32)]


def chan_start(is_e1b, lo_shift, ca_shift, secs):
    L = lib()
    L.ko_chan_start.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    o = ChanStart()
    L.ko_chan_start(int(bool(is_e1b)), int(lo_shift), int(ca_shift), float(secs), C.byref(o))
    return o


def aper_update(avg_pwr, row, algo, param, clear=False, start=0, stop=1024, waterfall_cal=-13):
    L = lib()
    L.ko_aper_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    avg = np.ascontiguousarray(avg_pwr, np.float32).copy()
    row = np.ascontiguousarray(row, np.uint8)
    L.ko_aper_update(_p(avg), _p(row), int(algo), float(param), int(bool(clear)), int(start), int(stop),
                     int(waterfall_cal))
    return avg


def aper_report(avg_pwr, start=0, stop=1024):
    avg = np.ascontiguousarray(avg_pwr, np.float32)
    s, n = C.c_int(), C.c_int()
    lib().ko_aper_report(_p(avg), C.c_int(start), C.c_int(stop), C.byref(s), C.byref(n))
    return s.value, n.value


def fir_process_taps(state, coef_cic, cic, x, prec=1):
    """-> (out, FirPos, pre [nblk, 1024], post [nblk, 1024]): ProcessData with its extension taps"""
    L = lib()
    L.ko_fir_process_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p]
    L.ko_fir_process_taps.restype = C.c_int
    x = np.ascontiguousarray(x, cpx)
    coef_cic = np.ascontiguousarray(coef_cic, cpx)
    cic = np.ascontiguousarray(cic, np.float32)
    out = np.empty(x.size + 1024, cpx)
    maxblk = x.size // 512 + 2
    pre, post = np.zeros((maxblk, 1024), cpx), np.zeros((maxblk, 1024), cpx)
    n = L.ko_fir_process_taps(C.byref(state), _p(coef_cic), _p(cic), _p(x), x.size, _p(out), int(prec), _p(pre), _p(post))
    nblk = n // 512
    return out[:n].copy(), state.in_pos - (FIR_SIZE - 1), pre[:nblk].copy(), post[:nblk].copy()


# ---- waterfall DDC ---------------------------------------------------------------
class DdcCicState(C.Structure):
    _fields_ = [("integ", (C.c_uint64 * 2) * 4), ("integ5", C.c_uint32),
                ("comb_prev", C.c_int64 * 5)]


class DdcWfState(C.Structure):
    _fields_ = [("phase", C.c_uint64), ("n", C.c_uint64), ("sample_no", C.c_uint32),
                ("cic", DdcCicState * 2)]


def ddc_nco_table():
    c = np.empty(8192, np.int16)
    s = np.empty(8192, np.int16)
    lib().ko_ddc_nco_table(_p(c), _p(s))
    return c, s


def ddc_wf(adc, phase_inc, log2r, state=None):
    """-> (iq int16 [nout, 2], state).  state None = reset (rst_wf_samp_wr)."""
    L = lib()
    L.ko_ddc_wf.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_uint64, C.c_int, C.c_void_p]
    L.ko_ddc_wf.restype = C.c_int
    adc = np.ascontiguousarray(adc, np.int16)
    if state is None:
        state = DdcWfState()
    out = np.empty((adc.size >> log2r) + 2, dtype=[("i", "<i2"), ("q", "<i2")])
    n = L.ko_ddc_wf(C.byref(state), _p(adc), adc.size, C.c_uint64(phase_inc & ((1 << 48) - 1)),
                    int(log2r), _p(out))
    return out[:n].view(np.int16).reshape(n, 2).copy(), state


class DdcRxState(C.Structure):
    _fields_ = [("phase", C.c_uint64), ("cnt1", C.c_uint32), ("cnt2", C.c_uint32), ("decim_by_2", C.c_int),
                ("i1", C.c_uint64 * 2), ("i2", C.c_uint64 * 2), ("i3", C.c_uint32 * 2),
                ("comb1_prev", (C.c_int64 * 3) * 2), ("j", (C.c_int64 * 5) * 2),
                ("comb2_prev", (C.c_int64 * 5) * 2), ("fir_buf", (C.c_int32 * 65) * 2)]


RX_DECIM = 1736 * 3 * 2          # RX_DECIM_4CH, kiwi.config:141


RX_STD, RX_WIDE, RX_14 = 0, 1, 2     # the RX instances of kiwi.config / fir_iq.sv (KO_RX_*)


def ddc_rx_decim(mode=RX_STD):
    return int(lib().ko_ddc_rx_decim(int(mode)))


def ddc_rx(adc, phase_inc, state=None, mode=RX_STD):
    """-> (rx_iq_t bytes uint8[nout*6], state)."""
    L = lib()
    L.ko_ddc_rx_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_uint64, C.c_void_p, C.c_int]
    L.ko_ddc_rx_mode.restype = C.c_int
    adc = np.ascontiguousarray(adc, np.int16)
    if state is None:
        state = DdcRxState()
    out = np.zeros((adc.size // ddc_rx_decim(mode) + 2) * 6, np.uint8)
    n = L.ko_ddc_rx_mode(C.byref(state), _p(adc), adc.size, C.c_uint64(phase_inc & ((1 << 48) - 1)), _p(out),
                         int(mode))
    return out[:6 * n].copy(), state


def ref_cacode(t0, t1):
    """Chips from the REFERENCE's own gps/cacode.h (oracle/_ref/cacode_ref), or None."""
    exe = os.path.join(_HERE, "_ref", "cacode_ref")
    if not os.path.exists(exe):
        return None
    s = subprocess.check_output([exe, str(t0), str(t1)]).decode().strip()
    return np.frombuffer(s.encode(), np.uint8) - ord("0")


# ---- the platform's log10f (kiwi_oracle_libm.c) ------------------------------------------
def libm_log10f(x):
    """log10f of the libm this oracle is linked against (the reference's S-meter / CAgc call it), elementwise."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().ko_libm_log10f(_p(x), _p(y), C.c_size_t(x.size))
    return y


def libm_log10f_bits(first, n):
    """log10f of the floats whose bit patterns are first .. first + n - 1"""
    y = np.empty(int(n), np.float32)
    lib().ko_libm_log10f_bits(C.c_uint32(int(first)), C.c_size_t(int(n)), _p(y))
    return y


def log10f_restated(x, fused=True):
    L = lib()
    L.ko_log10f_restated.argtypes = [C.c_float, C.c_int]
    L.ko_log10f_restated.restype = C.c_float
    return np.array([L.ko_log10f_restated(float(v), int(fused)) for v in np.atleast_1d(np.asarray(x, np.float32))], np.float32)


def libm_check_range(first, n, step=1, fused=True, threads=8):
    """The restated logf / log10f (csrc/kg_libm.h carries the same) against libm over bit patterns first, first + step, ...
    -> (values compared, logf differences, log10f differences, a differing pattern or 0)"""
    L = lib()
    L.ko_libm_check_range.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ko_libm_check_range.restype = C.c_uint64
    a, b, u = C.c_uint64(), C.c_uint64(), C.c_uint32()
    done = L.ko_libm_check_range(int(first), int(n), int(step), int(bool(fused)), int(threads), C.byref(a), C.byref(b), C.byref(u))
    return int(done), int(a.value), int(b.value), int(u.value)


def libm_powf_bits(base, first, n):
    y = np.empty(int(n), np.float32)
    lib().ko_libm_powf_bits(C.c_float(float(base)), C.c_uint32(int(first)), C.c_size_t(int(n)), _p(y))
    return y


def libm_expf_bits(first, n):
    y = np.empty(int(n), np.float32)
    lib().ko_libm_expf_bits(C.c_uint32(int(first)), C.c_size_t(int(n)), _p(y))
    return y


def libm_powf(base, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().ko_libm_powf(C.c_float(float(base)), _p(x), _p(y), C.c_size_t(x.size))
    return y


def libm_expf(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().ko_libm_expf(_p(x), _p(y), C.c_size_t(x.size))
    return y


def libm_check_pow_exp(first, n, step=1, fused=True, fused_residual=True, threads=8):
    """The restated powf(10, .) / expf / powf(random pairs) against libm over bit patterns first, first + step, ...
    -> (values compared, powf(10, .) differences, expf differences, random-pair differences, random pairs)"""
    L = lib()
    L.ko_libm_check_pow_exp.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4
    L.ko_libm_check_pow_exp.restype = C.c_uint64
    a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    done = L.ko_libm_check_pow_exp(int(first), int(n), int(step), int(bool(fused)), int(bool(fused_residual)), int(threads),
                                   C.byref(a), C.byref(b), C.byref(c), C.byref(d))
    return int(done), int(a.value), int(b.value), int(c.value), int(d.value)
