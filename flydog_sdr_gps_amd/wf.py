"""Host-side mirror of the reference's waterfall entry points over the C ABI.

Reference (rx/rx_waterfall.cpp)                 here
  c2s_waterfall_init() tables      :122-203  ->  window_functions(), cic_comp_table()
  c2s_waterfall(): zoom/start      :410-515  ->  WfParams.for_zoom()
                   maps, scales    :756-928  ->  build_maps(), WfParams.fft_scale
  sample_wf() window + compute_frame()       ->  Waterfall.frames()  (GPU, libkiwigpu.so)
The table/map builders below are host logic (they run once per tune, on the
CPU, in the reference too); every per-frame computation is in the library.
"""
import ctypes as C

import numpy as np

from ._lib import Context, WfChanCfgC, check, ptr

WF_NFFT = 8192          # rx/rx_waterfall.h:61-62
WF_WIDTH = 1024         # rx/rx_waterfall.h:65
MAX_ZOOM = 14           # kiwi.config:196
WINF_HANNING, WINF_HAMMING, WINF_BLACKMAN_HARRIS, WINF_NONE = 0, 1, 2, 3   # rx_waterfall.h:160-163
WF_MAX, WF_MIN, WF_LAST, WF_DROP, WF_CMA = 0, 1, 2, 3, 4                   # rx_waterfall.h:116

_f32 = np.float32


def window_functions():
    """wf_shmem_t.window_function[4][8192] (:136-171): 2^-16 * window, float storage,
    the cosine argument (K_2PI*i)/(float)(8191) evaluated in double."""
    i = np.arange(WF_NFFT, dtype=np.float64)
    a = (2.0 * 3.14159265358979323846 * i) / float(_f32(WF_NFFT - 1))
    base = np.full(WF_NFFT, _f32(2.0 ** -16), _f32)
    shapes = [0.5 - 0.5 * np.cos(a),
              0.54 - 0.46 * np.cos(a),
              0.35875 - 0.48829 * np.cos(a) + 0.14128 * np.cos(2.0 * a) - 0.01168 * np.cos(3.0 * a),
              None]
    out = np.empty((4, WF_NFFT), _f32)
    for k, sh in enumerate(shapes):
        out[k] = base if sh is None else (base.astype(np.float64) * sh).astype(_f32)
    return out


def cic_comp_table():
    """wf_shmem_t.CIC_comp[8192] (:175-185); TYPEREAL is float, MSIN is sinf."""
    K_PI = 3.14159265358979323846
    i = np.arange(WF_NFFT)
    x = (i.astype(_f32) / _f32(WF_NFFT)).astype(np.float64) + 0.5           # TYPEREAL(i)/N + 0.5f
    f = np.abs(np.fmod(x, 1.0) - 0.5).astype(_f32)                          # const TYPEREAL f
    fpi = f.astype(np.float64) * K_PI                                       # f*K_PI (double)
    sin_f = np.sin(fpi.astype(_f32)).astype(_f32).astype(np.float64)        # sinf((float)(f*K_PI))
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(f != 0, sin_f / fpi, 1.0).astype(_f32)              # const TYPEREAL sincf
    e = (_f32(36.26) * (f - _f32(0.5))).astype(np.float64)                  # p2*(f-0.5f): float
    cic = (np.power(sinc.astype(np.float64), -5.0) + float(_f32(-2.969)) * np.exp(e)).astype(_f32)
    return (0.5 + cic.astype(np.float64) / 2.0).astype(_f32)


class WfParams:
    """What c2s_waterfall() derives from `SET zoom= start=` (:410-515, :756-773, :889-903)."""

    def __init__(self, zoom, decim, start, i_offset, fft_used, plot_width, plot_width_clamped,
                 fft_scale, fft_offset):
        self.zoom, self.decim, self.start, self.i_offset = zoom, decim, start, i_offset
        self.fft_used, self.plot_width, self.plot_width_clamped = fft_used, plot_width, plot_width_clamped
        self.fft_scale, self.fft_offset = fft_scale, fft_offset

    @staticmethod
    def for_zoom(zoom, start, adc_clock=125.0e6, ui_srate=32.0e6, spectral_inversion=False):
        hz_per_start = _f32(ui_srate / (WF_WIDTH << MAX_ZOOM))                    # :262
        zm1 = zoom - 1 if zoom else 0                                              # :411
        decim = 1 << zm1                                                           # :415-426
        maxstart = (WF_WIDTH << MAX_ZOOM) - (WF_WIDTH << (MAX_ZOOM - zoom))        # :69
        start = _f32(min(max(_f32(start), _f32(0)), _f32(maxstart)))               # :486
        off = start * hz_per_start                                                 # :490
        off_inv = (_f32(maxstart) - start) * hz_per_start                          # :491
        v = float(off_inv if spectral_inversion else off) / adc_clock * 2.0 ** 48  # :498
        i_offset = (-int(v)) & 0xFFFFFFFFFFFF                                      # :499,:507
        fft_used = WF_NFFT // 2                                                    # :756
        if zoom != 0:
            fft_used //= 2                                                         # :762
        span = _f32(adc_clock / 2 / (1 << zoom))                                   # :765
        disp_fs = _f32(ui_srate / (1 << zoom))                                     # :766
        plot_width = int(_f32(_f32(WF_WIDTH) * span) / disp_fs)                    # :772
        pwc = min(plot_width, WF_WIDTH)                                            # :773
        maxmag = _f32(fft_used if zoom else fft_used // 2)                         # :891
        fft_scale = _f32(5.0 / float(maxmag * maxmag))                             # :897
        fft_offset = _f32(-0.08 if zoom else -0.8)                                 # :898
        return WfParams(zoom, decim, float(start), i_offset, fft_used, plot_width, pwc,
                        fft_scale, fft_offset)


def start_of_cf(zoom, cf_khz, ui_srate=32.0e6):
    """`SET zoom=<z> cf=<kHz>` -> the start the `start=` form would carry (rx/rx_waterfall.cpp:379-383), float arithmetic."""
    hz_per_start = _f32(ui_srate / (WF_WIDTH << MAX_ZOOM))
    zoom = min(max(int(zoom), 0), MAX_ZOOM)
    half_span = _f32((ui_srate / (1 << zoom)) / 2)
    cf = _f32(_f32(cf_khz) * _f32(1000.0))
    return float(_f32(_f32(cf - half_span) / hz_per_start))


def scale_arrays(params, ui_srate=32.0e6, masked=()):
    """wf_inst_t.fft_scale[] / .fft_scale_div2[] of the first plot_width_clamped pixels (rx/rx_waterfall.cpp:888-925): the zoom's
    scale, 0 where the pixel's frequency lies inside one of the admin's masked ranges [(lo_Hz, hi_Hz), ...]."""
    hz_per_start = _f32(ui_srate / (WF_WIDTH << MAX_ZOOM))
    n = params.plot_width_clamped
    scale = np.full(n, params.fft_scale, np.float32)
    if len(masked):
        i = np.arange(n, dtype=np.int64)
        f = np.rint((int(params.start) + (i << (MAX_ZOOM - params.zoom))).astype(np.float32) * hz_per_start).astype(np.int64)   # :907
        for lo, hi in masked:
            scale[(f >= int(lo)) & (f <= int(hi))] = 0
    return scale, (scale / np.float32(2)).astype(np.float32)


def build_maps(fft_used, plot_width, plot_width_clamped, spectral_inversion=False):
    """wf_inst_t.fft2wf_map[] and .drop_sample[] for the "FFT >= plot" case (:798-830)."""
    i = np.arange(fft_used, dtype=np.int64)
    j = plot_width * i // fft_used                                                 # :800
    if spectral_inversion:
        j = np.where(j < WF_WIDTH, WF_WIDTH - 1 - j, -1)                           # :801-802
    fft2wf = (j & 0xFFFF).astype(np.uint16)                                        # u2_t
    inv = int(np.rint(_f32(_f32(fft_used) * _f32(plot_width_clamped - 1)) / _f32(plot_width)))   # :814
    k = np.arange(WF_WIDTH, dtype=np.int64)
    d = np.rint((_f32(fft_used) * k.astype(_f32)) / _f32(plot_width)).astype(np.int64)           # :821
    if spectral_inversion:
        d = inv - d
    drop = np.zeros(WF_WIDTH, np.uint16)
    drop[:plot_width_clamped] = (d[:plot_width_clamped] & 0xFFFF).astype(np.uint16)
    return fft2wf, drop


class Waterfall:
    """GPU waterfall engine for nchan channels on one device (kg_wf)."""

    def __init__(self, ctx=None, nchan=4, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan = nchan
        h = C.c_void_p()
        check(self.lib.kg_wf_create(self.ctx.h, int(nchan), C.byref(h)), "kg_wf_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None) and not getattr(self, "_borrowed", False):          # an object must not outlive its context
                self.lib.kg_wf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tables(self, windows=None, cic_comp=None):
        """c2s_waterfall_init(): upload the window functions and the CIC compensation."""
        windows = window_functions() if windows is None else np.ascontiguousarray(windows, _f32)
        cic_comp = cic_comp_table() if cic_comp is None else np.ascontiguousarray(cic_comp, _f32)
        assert windows.shape == (4, WF_NFFT) and cic_comp.shape == (WF_NFFT,)
        check(self.lib.kg_wf_set_tables(self.h, ptr(windows), ptr(cic_comp)), "kg_wf_set_tables")

    def set_channel(self, ch, params, interp=WF_CMA, window_func=WINF_HANNING, cic_comp=True,
                    overlapped=False, spectral_inversion=False, fft_scale=None, maps=None):
        """The per-channel state compute_frame() reads (new map / new scale, :775-928)."""
        fft2wf, drop = maps if maps is not None else build_maps(
            params.fft_used, params.plot_width, params.plot_width_clamped, spectral_inversion)
        scale = np.full(WF_WIDTH, params.fft_scale, _f32) if fft_scale is None \
            else np.ascontiguousarray(fft_scale, _f32)
        scale2 = (scale / _f32(2)).astype(_f32)                                    # :921,:926
        cfg = WfChanCfgC(params.zoom, window_func, interp, int(cic_comp), int(overlapped),
                         params.fft_used, params.plot_width, params.plot_width_clamped,
                         float(params.fft_offset))
        check(self.lib.kg_wf_set_channel(self.h, int(ch), C.byref(cfg), ptr(fft2wf), ptr(drop),
                                         ptr(scale), ptr(scale2)), "kg_wf_set_channel")

    def frames(self, chan_of, iq):
        """iq: [nframes, 8192, 2] int16 (host).  -> [nframes, 1024] uint8."""
        chan_of = np.ascontiguousarray(chan_of, np.int32)
        iq = np.ascontiguousarray(iq, np.int16).reshape(chan_of.size, WF_NFFT, 2)
        out = np.empty((chan_of.size, WF_WIDTH), np.uint8)
        check(self.lib.kg_wf_frames(self.h, chan_of.size, ptr(chan_of), ptr(iq), ptr(out)),
              "kg_wf_frames")
        return out

    def frames_dev(self, chan_of, d_iq, d_out, frame_off=None, iq_len=None):
        """Device pointers (ints); enqueue only.  frame_off: where each frame starts, in iq_t pairs
        after d_iq (default: frames back to back); iq_len: how many pairs d_iq points at (required
        with frame_off: a frame that would run past it is rejected, never read)."""
        chan_of = np.ascontiguousarray(chan_of, np.int32)
        if frame_off is None:
            check(self.lib.kg_wf_frames_dev(self.h, chan_of.size, ptr(chan_of), ptr(int(d_iq)),
                                            ptr(int(d_out))), "kg_wf_frames_dev")
            return
        frame_off = np.ascontiguousarray(frame_off, np.uint64)
        if frame_off.size != chan_of.size:
            raise ValueError("frame_off and chan_of differ in length")
        if iq_len is None:
            raise ValueError("frames_dev(frame_off=...) needs iq_len, the extent of d_iq in iq_t pairs")
        check(self.lib.kg_wf_frames_at_dev(self.h, chan_of.size, ptr(chan_of), ptr(frame_off), int(iq_len),
                                           ptr(int(d_iq)), ptr(int(d_out))), "kg_wf_frames_at_dev")

    def debug_frame(self, ch, iq):
        iq = np.ascontiguousarray(iq, np.int16).reshape(WF_NFFT, 2)
        out = np.empty(WF_WIDTH, np.uint8)
        pwr = np.zeros(4096, _f32)
        pwr_out = np.empty(WF_WIDTH, _f32)
        dB = np.empty(WF_WIDTH, _f32)
        check(self.lib.kg_wf_debug_frame(self.h, int(ch), ptr(iq), ptr(out), ptr(pwr), ptr(pwr_out),
                                         ptr(dB)), "kg_wf_debug_frame")
        return out, pwr, pwr_out, dB
