"""Host-side mirror of the hand-off pieces over the C ABI.

Reference                                                             here
  CHANNEL::Start() rates / creep / pause   gps/channel.cpp:281-311 -> chan_start
  aperture_auto() averaging                rx/rx_waterfall.cpp:1183-1222 -> Aperture.update
  aperture_auto() signal / noise bands     rx/rx_waterfall.cpp:1233-1272 -> Aperture.report
"""
import ctypes as C

import numpy as np

from ._lib import Context, check, load_library, ptr

IIR, MMA, EMA = range(3)                               # aper_algo_t, rx/rx_waterfall.h:113
DECIM = 4                                              # gps.h: ca_shift = idx * DECIM (search.cpp:575)


class ChanStart(C.Structure):
    _fields_ = [("lo_dop", C.c_double), ("ca_dop", C.c_double), ("lo_rate", C.c_uint32), ("ca_rate", C.c_uint32),
                ("ca_pause", C.c_uint32), ("code_creep", C.c_int32)]


class AperCfg(C.Structure):
    _fields_ = [("algo", C.c_int32), ("param", C.c_float), ("clear", C.c_int32), ("audio_fft", C.c_int32)]


def chan_start(is_e1b, lo_shift, ca_shift, secs, lib=None):
    """-> ChanStart for an acquisition result: lo_shift = result dop bin, ca_shift = idx * DECIM."""
    lib = lib if lib is not None else load_library()
    o = ChanStart()
    lib.kg_acq_chan_start(int(bool(is_e1b)), int(lo_shift), int(ca_shift), float(secs), C.byref(o))
    return o


class Aperture:
    """avg_pwr[1024] of nchan waterfalls on the GPU (kg_aper)."""

    def __init__(self, ctx=None, nchan=4, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan = nchan
        h = C.c_void_p()
        check(self.lib.kg_aper_create(self.ctx.h, int(nchan), C.byref(h)), "kg_aper_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):          # an object must not outlive its context
                self.lib.kg_aper_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_dev(self, chans, d_rows, row_stride, cfgs, waterfall_cal=-13):
        """cfgs: list of (algo, param, clear, audio_fft) per row"""
        chans = np.ascontiguousarray(chans, np.int32)
        arr = (AperCfg * len(cfgs))(*[AperCfg(int(a), float(p), int(bool(c)), int(bool(f))) for a, p, c, f in cfgs])
        check(self.lib.kg_aper_update_dev(self.h, ptr(chans), chans.size, C.c_void_p(int(d_rows)), int(row_stride),
                                          arr, int(waterfall_cal)), "kg_aper_update_dev")

    def update(self, chans, rows, cfgs, waterfall_cal=-13):
        rows = np.ascontiguousarray(rows, np.uint8).reshape(len(cfgs), 1024)
        d = self.ctx.alloc(rows.nbytes)
        try:
            self.ctx.upload(d, rows)
            self.update_dev(chans, d, 1024, cfgs, waterfall_cal)
            self.ctx.sync()
        finally:
            self.ctx.free(d)

    def report(self, chans, audio_fft=None):
        chans = np.ascontiguousarray(chans, np.int32)
        af = np.zeros(chans.size, np.int32) if audio_fft is None else np.ascontiguousarray(audio_fft, np.int32)
        sig, noise = np.zeros(chans.size, np.int32), np.zeros(chans.size, np.int32)
        check(self.lib.kg_aper_report(self.h, ptr(chans), chans.size, ptr(af), ptr(sig), ptr(noise)), "kg_aper_report")
        return sig, noise

    def get(self, ch):
        out = np.zeros(1024, np.float32)
        check(self.lib.kg_aper_get(self.h, int(ch), ptr(out)), "kg_aper_get")
        return out
