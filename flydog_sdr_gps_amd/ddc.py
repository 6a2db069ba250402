"""Host-side mirror of the waterfall DDC control path over the C ABI.

In the reference the DDC is FPGA fabric (verilog/rx/waterfall_1cic.v) driven by
SPI commands from c2s_waterfall()/sample_wf() (rx/rx_waterfall.cpp):
  spi_set(CmdSetWFDecim, rx_chan, decim)                 :466   ->  Ddc.set_wf(ch, inc, decim)
  spi_set3(CmdSetWFFreq, rx_chan, i_offset hi32, lo16)   :507   ->  Ddc.set_wf(...)
  spi_set(CmdWFReset, rx_chan, WF_SAMP_RD_RST|WR_RST..)  :1005  ->  Ddc.reset(ch)
  spi_get_noduplex(CmdGetWFSamples, ...) x 9             :1036  ->  Ddc.push(adc) -> iq_t arrays
push_dev() takes device pointers (from kg_dev_alloc or any HIP allocator).
"""
import ctypes as C

import numpy as np

from ._lib import Context, check, ptr, own_rows


class Ddc:
    def __init__(self, ctx=None, nchan=4, max_samples=1 << 24, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan, self.max_samples = nchan, max_samples
        h = C.c_void_p()
        check(self.lib.kg_ddc_create(self.ctx.h, int(nchan), int(max_samples), C.byref(h)),
              "kg_ddc_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None) and not getattr(self, "_borrowed", False):          # an object must not outlive its context
                self.lib.kg_ddc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_wf(self, ch, phase_inc, decim):
        check(self.lib.kg_ddc_set_wf(self.h, int(ch), int(phase_inc) & ((1 << 48) - 1), int(decim)),
              "kg_ddc_set_wf")

    def reset(self, ch):
        check(self.lib.kg_ddc_reset_wf(self.h, int(ch)), "kg_ddc_reset_wf")

    def set_phase(self, ch, phase):
        check(self.lib.kg_ddc_set_phase(self.h, int(ch), int(phase) & ((1 << 48) - 1)),
              "kg_ddc_set_phase")

    def outputs(self, ch, n):
        return check(self.lib.kg_ddc_wf_outputs(self.h, int(ch), int(n)), "kg_ddc_wf_outputs")

    def set_deferred(self, on=True):
        """Deferred output stage (kg_ddc_wf_set_deferred): push_dev then leaves the stage that writes the rows on a stream
        of the object; join() makes a stream wait for it."""
        check(self.lib.kg_ddc_wf_set_deferred(self.h, int(bool(on))), "kg_ddc_wf_set_deferred")
        self.deferred = bool(on)

    def join(self, stream=None):
        """`stream` (a raw hipStream_t handle as int; None: the context's stream) waits for the last push's outputs."""
        check(self.lib.kg_ddc_wf_join(self.h, ptr(int(stream)) if stream else None), "kg_ddc_wf_join")

    def tail_after(self, event):
        """The next push's writers of the output rows start only after `event` (a raw hipEvent_t handle as int)."""
        check(self.lib.kg_ddc_wf_tail_after(self.h, ptr(int(event)) if event else None), "kg_ddc_wf_tail_after")

    def push_dev(self, d_adc, n, chans, d_out, out_stride):
        """Device pointers (ints).  Returns the per-channel output counts."""
        chans = np.ascontiguousarray(chans, np.int32)
        nouts = np.zeros(chans.size, np.int64)
        check(self.lib.kg_ddc_wf_push_dev(self.h, ptr(int(d_adc)), int(n), ptr(chans), chans.size,
                                          ptr(int(d_out)), int(out_stride), ptr(nouts)),
              "kg_ddc_wf_push_dev")
        return nouts

    def capture_dev(self, d_adc, n, chans, d_out, out_stride, max_out=8192):
        """The reference's non-overlapped frame (CmdWFReset + one-shot sampler, kg_ddc_wf_capture_dev): CICs reset at the
        block's first sample, each channel's first max_out outputs written, NCOs advanced by the whole block."""
        chans = np.ascontiguousarray(chans, np.int32)
        nouts = np.zeros(chans.size, np.int64)
        check(self.lib.kg_ddc_wf_capture_dev(self.h, ptr(int(d_adc)), int(n), ptr(chans), chans.size,
                                             ptr(int(d_out)), int(out_stride), int(max_out), ptr(nouts)),
              "kg_ddc_wf_capture_dev")
        return nouts

    def capture(self, adc, chans, max_out=8192):
        """Convenience for tests: host int16 array in, list of [<= max_out, 2] int16 arrays out."""
        own_rows(self, "capture()")
        adc = np.ascontiguousarray(adc, np.int16)
        d_adc = self.ctx.alloc(adc.nbytes)
        d_out = self.ctx.alloc(len(chans) * max_out * 4)
        try:
            self.ctx.upload(d_adc, adc)
            nouts = self.capture_dev(d_adc, adc.size, chans, d_out, max_out, max_out)
            if getattr(self, "deferred", False):
                self.join()
            host = np.zeros((len(chans), max_out, 2), np.int16)
            self.ctx.download(d_out, host)
        finally:
            self.ctx.free(d_adc)
            self.ctx.free(d_out)
        return [host[i, :int(nouts[i])].copy() for i in range(len(chans))]

    def push(self, adc, chans):
        """Convenience for tests: host int16 array in, list of [nout, 2] int16 arrays out."""
        own_rows(self, "push()")
        adc = np.ascontiguousarray(adc, np.int16)
        stride = max(int(self.outputs(c, adc.size)) for c in chans) + 1
        d_adc = self.ctx.alloc(adc.nbytes)
        d_out = self.ctx.alloc(len(chans) * stride * 4)
        try:
            self.ctx.upload(d_adc, adc)
            nouts = self.push_dev(d_adc, adc.size, chans, d_out, stride)
            if getattr(self, "deferred", False):
                self.join()
            host = np.zeros((len(chans), stride, 2), np.int16)
            self.ctx.download(d_out, host)
        finally:
            self.ctx.free(d_adc)
            self.ctx.free(d_out)
        return [host[i, :int(nouts[i])].copy() for i in range(len(chans))]


RX_DECIM = 1736 * 3 * 2      # RX_DECIM_4CH (kiwi.config:141): RX1_STD_DECIM * RX2_STD_DECIM * CICF_DECIM_BY_2
RX_DECIM_WIDE = 1543 * 2 * 2   # RX_DECIM_3CH (kiwi.config:140): the rx3 / 20.25 kHz instance
# the RX instances the reference builds (KG_RXDDC_* in kiwigpu.h): rx4 / rx8, rx3, rx14
RX_STD, RX_WIDE, RX_14 = 0, 1, 2


def rx_phase_inc(freq_hz, adc_clock=125.0e6, spectral_inversion=False, ui_srate=32.0e6):
    """rx_sound_set_freq (rx/rx_sound_cmd.cpp:80-90): i_phase = (u64) round(f / adc_clk * 2^48), f mirrored about the displayed
    bandwidth (ui_srate - f) when the admin's spectral inversion is on; C's round() (half away from zero), 48 bits to the FPGA."""
    import math
    f = (ui_srate - freq_hz) if spectral_inversion else freq_hz
    v = f / adc_clock * 2.0 ** 48
    return int(math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1)) & ((1 << 48) - 1)


class RxDdc:
    """The per-channel audio DDC (verilog/rx/rx.v) for nchan channels (kg_rxddc):
    CmdSetRXFreq -> set_freq(ch, inc);  CmdGetRX -> push(adc) -> rx_iq_t records."""

    def __init__(self, ctx=None, nchan=4, max_samples=1 << 24, device=0, mode=RX_STD):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan, self.max_samples, self.mode = nchan, max_samples, int(mode)
        h = C.c_void_p()
        check(self.lib.kg_rxddc_create_mode(self.ctx.h, int(nchan), int(max_samples), int(mode), C.byref(h)),
              "kg_rxddc_create_mode")
        self.h = h
        self.decim = check(self.lib.kg_rxddc_decim(h), "kg_rxddc_decim")

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None) and not getattr(self, "_borrowed", False):          # an object must not outlive its context
                self.lib.kg_rxddc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_freq(self, ch, phase_inc):
        check(self.lib.kg_rxddc_set_freq(self.h, int(ch), int(phase_inc) & ((1 << 48) - 1)),
              "kg_rxddc_set_freq")

    def reset(self, ch):
        check(self.lib.kg_rxddc_reset(self.h, int(ch)), "kg_rxddc_reset")

    def outputs(self, ch, n):
        return check(self.lib.kg_rxddc_outputs(self.h, int(ch), int(n)), "kg_rxddc_outputs")

    def push_dev(self, d_adc, n, chans, d_out, out_stride):
        chans = np.ascontiguousarray(chans, np.int32)
        nouts = np.zeros(chans.size, np.int32)
        check(self.lib.kg_rxddc_push_dev(self.h, ptr(int(d_adc)), int(n), ptr(chans), chans.size,
                                         ptr(int(d_out)), int(out_stride), ptr(nouts)),
              "kg_rxddc_push_dev")
        return nouts

    def push(self, adc, chans):
        """Host int16 array in; per channel the rx_iq_t bytes (uint8[nout*6]) out."""
        own_rows(self, "push()")
        adc = np.ascontiguousarray(adc, np.int16)
        stride = max(int(self.outputs(c, adc.size)) for c in chans) + 1
        d_adc = self.ctx.alloc(adc.nbytes)
        d_out = self.ctx.alloc(len(chans) * stride * 6)
        try:
            self.ctx.upload(d_adc, adc)
            nouts = self.push_dev(d_adc, adc.size, chans, d_out, stride)
            host = np.zeros((len(chans), stride * 6), np.uint8)
            self.ctx.download(d_out, host)
        finally:
            self.ctx.free(d_adc)
            self.ctx.free(d_out)
        return [host[i, :6 * int(nouts[i])].copy() for i in range(len(chans))]
