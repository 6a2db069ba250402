"""Host-side mirror of the audio front over the C ABI.

Reference                                               here
  snd_service() unpack     rx/data_pump.cpp:145-208  ->  unpack(ctx, raw, nsamps, nchans, ...)
  rescale constant         rx/data_pump.cpp:73-74    ->  RESCALE
  m_PassbandFIR[ch].SetupParameters(...)  fastfir.cpp:171  ->  FastFir.setup(ch, lo, hi, offset, fs)
  m_PassbandFIR[ch].ProcessData(...)      fastfir.cpp:241  ->  FastFir.process(ch, samples)
  m_PassbandFIR[ch].FirPos()              fastfir.h:33     ->  FastFir.pos(ch)
"""
import ctypes as C

import numpy as np

from ._lib import Context, check, ptr, own_rows

# rescale = MPOW(2, -RXOUT_SCALE + CUTESDR_SCALE) * MPOW(10, CICF_GAIN_dB/20)  (float arithmetic)
RESCALE = float(np.float32(2.0 ** -8) * np.float32(np.power(np.float32(10.0), np.float32(4.5 / 20.0))))
WINF_BLACKMAN_NUTTALL, WINF_BLACKMAN_HARRIS, WINF_NUTTALL, WINF_HANNING, WINF_HAMMING = range(5)  # rx_sound.h:72-76


def pack_rx_iq(i24, q24):
    """Build the wire records of rx/data_pump.h:27-30 from 24-bit I/Q arrays shaped
    [nsamps, nchans] (test/bench helper; the FPGA produces these in the reference)."""
    i24 = np.asarray(i24, np.int64) & 0xFFFFFF
    q24 = np.asarray(q24, np.int64) & 0xFFFFFF
    rec = np.zeros(i24.shape + (6,), np.uint8)
    rec[..., 0] = i24 & 0xFF
    rec[..., 1] = (i24 >> 8) & 0xFF
    rec[..., 2] = q24 & 0xFF
    rec[..., 3] = (q24 >> 8) & 0xFF
    rec[..., 4] = (q24 >> 16) & 0xFF        # q3
    rec[..., 5] = (i24 >> 16) & 0xFF        # i3
    return rec.reshape(-1)


def unpack(ctx, raw, nsamps, nchans, enabled=None, rescale=RESCALE, dc_i=0.0, dc_q=0.0,
           spectral_inversion=False, out=None):
    """raw: uint8[nsamps*nchans*6] (host).  -> complex64 [nchans, nsamps]."""
    raw = np.ascontiguousarray(raw, np.uint8)
    en = np.ones(nchans, np.uint8) if enabled is None else np.ascontiguousarray(enabled, np.uint8)
    host = np.zeros((nchans, nsamps), np.complex64) if out is None else out
    d_raw = ctx.alloc(raw.nbytes)
    d_out = ctx.alloc(host.nbytes)
    try:
        ctx.upload(d_raw, raw)
        ctx.upload(d_out, host)
        check(ctx.lib.kg_dpump_unpack_dev(ctx.h, C.c_void_p(d_raw), nsamps, nchans, ptr(en), rescale, dc_i,
                                          dc_q, int(bool(spectral_inversion)), C.c_void_p(d_out), nsamps),
              "kg_dpump_unpack_dev")
        ctx.download(d_out, host)
    finally:
        ctx.free(d_raw)
        ctx.free(d_out)
    return host


def unpack_rows_dev(ctx, d_raw, raw_stride, nsamps, nchans, d_out, out_stride, enabled=None, rescale=RESCALE,
                    dc_i=0.0, dc_q=0.0, spectral_inversion=False):
    """Device buffers; records one row per channel (kg_rxddc_push_dev's layout).  Enqueue only."""
    en = np.ones(nchans, np.uint8) if enabled is None else np.ascontiguousarray(enabled, np.uint8)
    check(ctx.lib.kg_dpump_unpack_rows_dev(ctx.h, C.c_void_p(int(d_raw)), int(raw_stride), int(nsamps), int(nchans),
                                           ptr(en), rescale, dc_i, dc_q, int(bool(spectral_inversion)),
                                           C.c_void_p(int(d_out)), int(out_stride)), "kg_dpump_unpack_rows_dev")


class FastFir:
    """The m_PassbandFIR[] array of rx/rx_sound.cpp:150 on the GPU (kg_fir)."""

    def __init__(self, ctx=None, nchan=4, max_in=4096, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan, self.max_in = nchan, max_in
        h = C.c_void_p()
        check(self.lib.kg_fir_create(self.ctx.h, int(nchan), int(max_in), C.byref(h)), "kg_fir_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None) and not getattr(self, "_borrowed", False):          # an object must not outlive its context
                self.lib.kg_fir_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setup(self, ch, lo, hi, offset, fs, window_func=-1, do_cic_comp=False, snd_rate_3ch=False):
        """SetupParameters; returns False when the reference's sanity check rejects the call."""
        rc = check(self.lib.kg_fir_setup(self.h, int(ch), lo, hi, offset, fs, int(window_func),
                                         int(bool(do_cic_comp)), int(bool(snd_rate_3ch))), "kg_fir_setup")
        return rc == 0

    def set_coef(self, ch, coef_fft):
        coef_fft = np.ascontiguousarray(coef_fft, np.complex64)
        assert coef_fft.size == 1024
        check(self.lib.kg_fir_set_coef(self.h, int(ch), ptr(coef_fft)), "kg_fir_set_coef")

    def get_coef(self, ch):
        out = np.empty(1024, np.complex64)
        check(self.lib.kg_fir_get_coef(self.h, int(ch), ptr(out)), "kg_fir_get_coef")
        return out

    def reset(self, ch):
        check(self.lib.kg_fir_reset(self.h, int(ch)), "kg_fir_reset")

    def pos(self, ch):
        return check(self.lib.kg_fir_pos(self.h, int(ch)), "kg_fir_pos")

    def process(self, ch, x):
        """ProcessData: -> the 0 or 512*k output samples."""
        own_rows(self, "process()")
        x = np.ascontiguousarray(x, np.complex64)
        out = np.empty(x.size + 512, np.complex64)
        n = check(self.lib.kg_fir_process(self.h, int(ch), ptr(x), x.size, ptr(out)), "kg_fir_process")
        return out[:n].copy()

    def process_dev(self, chans, d_in, in_stride, n, d_out, out_stride):
        chans = np.ascontiguousarray(chans, np.int32)
        nout = np.zeros(chans.size, np.int32)
        check(self.lib.kg_fir_process_dev(self.h, ptr(chans), chans.size, ptr(int(d_in)), int(in_stride),
                                          int(n), ptr(int(d_out)), int(out_stride), ptr(nout)),
              "kg_fir_process_dev")
        return nout

    def process_taps(self, ch, x):
        """ProcessData with its extension taps (fastfir.cpp:278-302), host arrays:
        -> (out, pre [nblk, 1024], post [nblk, 1024])"""
        own_rows(self, "process_taps()")
        x = np.ascontiguousarray(x, np.complex64)
        maxblk = x.size // 512 + 2
        ctx = self.ctx
        d_in, d_out = ctx.alloc(max(x.nbytes, 8)), ctx.alloc((x.size + 512) * 8)
        d_pre, d_post = ctx.alloc(maxblk * 1024 * 8), ctx.alloc(maxblk * 1024 * 8)
        try:
            ctx.upload(d_in, x)
            chans = np.array([ch], np.int32)
            nout = np.zeros(1, np.int32)
            check(self.lib.kg_fir_process_taps_dev(self.h, ptr(chans), 1, ptr(int(d_in)), x.size, x.size,
                                                   ptr(int(d_out)), x.size + 512, ptr(nout), ptr(int(d_pre)),
                                                   ptr(int(d_post)), maxblk * 1024), "kg_fir_process_taps_dev")
            ctx.sync()
            n = int(nout[0])
            out = np.zeros(max(n, 1), np.complex64)
            pre, post = np.zeros((maxblk, 1024), np.complex64), np.zeros((maxblk, 1024), np.complex64)
            if n:
                ctx.download(d_out, out[:n])
                ctx.download(d_pre, pre)
                ctx.download(d_post, post)
        finally:
            for d in (d_in, d_out, d_pre, d_post):
                ctx.free(d)
        return out[:n].copy(), pre[:n // 512].copy(), post[:n // 512].copy()

    def process_taps_edit(self, ch, x, edit):
        """ProcessData when a PRE_FILTERED extension rewrites the spectrum it is handed (`buf_modified`,
        fastfir.cpp:286-290): process with taps, apply edit(pre) -> edited blocks (host callable standing in
        for the extension's device code), filter the edited blocks again.  -> (out, pre, edited)"""
        own_rows(self, "process_taps_edit()")
        x = np.ascontiguousarray(x, np.complex64)
        maxblk = x.size // 512 + 2
        ctx = self.ctx
        d_in, d_out = ctx.alloc(max(x.nbytes, 8)), ctx.alloc((x.size + 512) * 8)
        d_pre = ctx.alloc(maxblk * 1024 * 8)
        try:
            ctx.upload(d_in, x)
            chans = np.array([ch], np.int32)
            nout = np.zeros(1, np.int32)
            check(self.lib.kg_fir_process_taps_dev(self.h, ptr(chans), 1, ptr(int(d_in)), x.size, x.size,
                                                   ptr(int(d_out)), x.size + 512, ptr(nout), ptr(int(d_pre)), None,
                                                   maxblk * 1024), "kg_fir_process_taps_dev")
            ctx.sync()
            n = int(nout[0])
            nblk = np.array([n // 512], np.int32)
            pre = np.zeros((maxblk, 1024), np.complex64)
            ctx.download(d_pre, pre)
            edited = pre.copy()
            edited[:n // 512] = edit(pre[:n // 512])
            ctx.upload(d_pre, edited)
            check(self.lib.kg_fir_refilter_dev(self.h, ptr(chans), 1, ptr(nblk), ptr(int(d_pre)), maxblk * 1024,
                                               ptr(int(d_out)), x.size + 512), "kg_fir_refilter_dev")
            ctx.sync()
            out = np.zeros(max(n, 1), np.complex64)
            if n:
                ctx.download(d_out, out[:n])
        finally:
            for d in (d_in, d_out, d_pre):
                ctx.free(d)
        return out[:n].copy(), pre[:n // 512].copy(), edited[:n // 512].copy()
