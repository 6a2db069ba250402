"""Host-side mirror of a bank of receivers over the C ABI (kg_rxbank, include/kiwigpu.h).

In the reference every connection runs two coroutines over what the data pump hands them:
  c2s_sound()      rx/rx_sound.cpp:333-601      in_samps -> CFastFIR -> S-meter / AGC / demod -> compression
  c2s_waterfall()  rx/rx_waterfall.cpp:930-1170  sample_wf() -> compute_frame() -> wf_pkt_t
RxBank is nrx such connections on one GPU; step() is ONE C call that enqueues a whole step of all of them.  The
per-seam objects (ddc, wf, rxddc, fir, post, adpcm) are the bank's own and are configured through the same wrapper
classes the single-seam tests use.
"""
import ctypes as C

import numpy as np

from . import post as post_mod
from . import wf as wf_mod
from ._lib import Context, borrow, check, load_library, ptr
from .ddc import RX_DECIM, RX_STD, Ddc, RxDdc, rx_phase_inc
from .post import Post
from .snd import RESCALE, FastFir
from .wf import Waterfall, WfParams
from .wire import Adpcm

WF_NFFT = 8192


class StepInfoC(C.Structure):
    _fields_ = [("step", C.c_uint64), ("nframes", C.c_int32), ("nrec", C.c_int32), ("nfir", C.c_int32),
                ("fir_pos", C.c_int32), ("snd_seq", C.c_uint32), ("table_bytes", C.c_int32), ("nmoves", C.c_int32)]


class BufsC(C.Structure):
    _fields_ = [("wf_iq", C.c_void_p), ("wf_iq_stride", C.c_size_t), ("wf_rows", C.c_void_p),
                ("wf_pkts", C.c_void_p), ("wf_pkt_stride", C.c_size_t), ("rx_raw", C.c_void_p), ("rx_stride", C.c_size_t),
                ("rx_in", C.c_void_p), ("fir_out", C.c_void_p), ("fir_stride", C.c_size_t), ("s16", C.c_void_p),
                ("adpcm", C.c_void_p), ("agc", C.c_void_p), ("iq_pay", C.c_void_p)]


ADC_CLOCK, UI_SRATE = 66.6666e6, 30.0e6


def survey_mix(nrx, first_rx=0, n=1 << 22, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE):
    """BASELINE configs[3] as SURVEY.md 8(d) defines it: receiver k of the 1024 listens at f_k = 100 kHz + k 29 kHz, its
    waterfall at zoom 8 + (k mod 4) centred on f_k, audio passband 300-2700 Hz.  With a step of n = 2^22 ADC samples zooms
    8..10 (R = 128..512) fill a frame inside the step -- the non-overlapped frame -- and zoom 11 (R = 1024, 2^23 samples
    per frame) is what sample_wf() switches to overlapped sampling for (rx_waterfall.cpp:962-983).
    -> [(WfParams, overlapped, audio phase increment)] for receivers first_rx .. first_rx + nrx - 1"""
    hz_per_start = ui_srate / (1024 << 14)
    out = []
    for i in range(nrx):
        k = first_rx + i
        f = 100.0e3 + 29.0e3 * k
        zoom = 8 + k % 4
        span = ui_srate / (1 << zoom)
        p = WfParams.for_zoom(zoom, max(f - span / 2, 0.0) / hz_per_start, adc_clock=adc_clock, ui_srate=ui_srate)
        out.append((p, WF_NFFT * p.decim > n, rx_phase_inc(f, adc_clock)))
    return out


def light_mix(nrx, first_rx=0, n=1 << 22, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE):
    """Rounds 2-4's receiver set, kept as a second named workload: zoom 1 + (k mod 10) (R = 1 .. 512, every frame one-shot),
    starts spread over the band, audio NCOs 10 Hz apart near 0.0123 f_adc."""
    hz_per_start = ui_srate / (1024 << 14)
    out = []
    for i in range(nrx):
        k = first_rx + i
        p = WfParams.for_zoom(1 + k % 10, (1.0e6 + 0.2e6 * (k % 97)) / hz_per_start, adc_clock=adc_clock, ui_srate=ui_srate)
        out.append((p, WF_NFFT * p.decim > n, rx_phase_inc(0.0123 * adc_clock - 1000.0 - 10.0 * k, adc_clock)))
    return out


MIXES = {"survey": survey_mix, "light": light_mix}


class RxBank:
    """nrx virtual receivers on one GPU (kg_rxbank)."""

    def __init__(self, nrx, n=1 << 22, device=0, rx_mode=RX_STD):
        self.lib = load_library()
        h = C.c_void_p()
        check(self.lib.kg_rxbank_create(int(device), int(nrx), int(n), int(rx_mode), C.byref(h)), "kg_rxbank_create")
        self.h, self.nrx, self.n, self.device = h, nrx, n, device
        self.ctx = Context.borrow(self.lib.kg_rxbank_ctx(h), device)
        self.ddc = borrow(Ddc, self.ctx, self.lib.kg_rxbank_ddc(h), nchan=nrx, max_samples=n)
        self.wf = borrow(Waterfall, self.ctx, self.lib.kg_rxbank_wf(h), nchan=nrx)
        self.rxddc = borrow(RxDdc, self.ctx, self.lib.kg_rxbank_rxddc(h), nchan=nrx, max_samples=n, mode=rx_mode)
        self.rxddc.decim = check(self.lib.kg_rxddc_decim(self.rxddc.h), "kg_rxddc_decim")
        self.rx_mode = int(rx_mode)
        self.fir = borrow(FastFir, self.ctx, self.lib.kg_rxbank_fir(h), nchan=nrx)
        self.post = borrow(Post, self.ctx, self.lib.kg_rxbank_post(h), nchan=nrx)
        self.adpcm = borrow(Adpcm, self.ctx, self.lib.kg_rxbank_adpcm(h), nchan=nrx)
        b = BufsC()
        check(self.lib.kg_rxbank_buffers(h, C.byref(b)), "kg_rxbank_buffers")
        self.bufs = b
        check(self.lib.kg_rxbank_set_unpack(h, RESCALE, 0.0, 0.0, 0), "kg_rxbank_set_unpack")
        self.params = [None] * nrx
        self.overlapped = [False] * nrx
        self.rx_inc = [0] * nrx
        self.little_endian = [False] * nrx
        self.audio = [None] * nrx                       # (mode, lo, hi, fs, de_emp, squelch) as set_audio configured it
        self.fs = ADC_CLOCK / self.rxddc.decim          # RX_DECIM (rx4 / rx8, rx14) or RX_DECIM_WIDE (rx3)

    def close(self):
        if getattr(self, "h", None):
            self.lib.kg_rxbank_destroy(self.h)
            self.h = None
            for o in (self.ddc, self.wf, self.rxddc, self.fir, self.post, self.adpcm, self.ctx):
                o.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration (the reference's per-connection command handlers)
    def set_wf(self, rx, params, overlapped=False, interp=wf_mod.WF_MAX, window_func=wf_mod.WINF_HANNING, cic_comp=True,
               use_compression=True):
        """`SET zoom= start=` of receiver rx: DDC frequency / decimation / sampler mode, compute_frame()'s per-channel
        state, the wf_pkt_t header fields."""
        check(self.lib.kg_rxbank_set_wf(self.h, int(rx), int(params.i_offset) & ((1 << 48) - 1), int(params.decim),
                                        int(bool(overlapped))), "kg_rxbank_set_wf")
        self.wf.set_channel(rx, params, interp=interp, window_func=window_func, cic_comp=cic_comp, overlapped=overlapped)
        check(self.lib.kg_rxbank_set_wf_pkt(self.h, int(rx), int(params.start), int(params.zoom), int(bool(use_compression))),
              "kg_rxbank_set_wf_pkt")
        self.params[rx], self.overlapped[rx] = params, bool(overlapped)

    def set_audio(self, rx, phase_inc, lo=300.0, hi=2700.0, fs=None, mode=post_mod.MODE_SSB, de_emp=0, squelch=0):
        """A connection's `SET mod= low_cut= high_cut= freq=`, `SET de_emp=`, `SET squelch=` for receiver rx: NCO, passband
        filter and the post-AM-detector filter designed with it (rx/rx_sound_cmd.cpp:268-282), AGC / S-meter / detector, the
        squelch of a new connection (rx/rx_sound.cpp:261-262) then the command's value, the mode's de-emphasis filter."""
        fs = self.fs if fs is None else fs
        fmax = int(fs / 2 - 1)                          # the handler clamps the client's cuts first (rx_sound_cmd.cpp:248-250)
        lo, hi = max(float(lo), float(-fmax)), min(float(hi), float(fmax))
        self.rxddc.set_freq(rx, phase_inc)
        if not self.fir.setup(rx, lo, hi, 0.0, fs):
            raise ValueError("set_audio: CFastFIR::SetupParameters rejects the passband %g .. %g Hz at %g Hz (fastfir.cpp:193-200)" % (lo, hi, fs))
        self.post.set_am_passband(rx, lo, hi, fs)
        self.post.set_agc(rx, True, False, -100, 50, 6, 1000, fs)
        self.post.set_smeter(rx, fs)
        self.post.set_mode(rx, mode)
        self.post.reset(rx)
        self.post.squelch_setup(rx, fs)
        self.post.squelch_set(rx, 0, 0)
        if squelch:
            self.post.squelch_set(rx, squelch, 0)
        nfm = mode == post_mod.MODE_NBFM
        self.post.set_deemp(rx, True, 0)
        self.post.set_deemp(rx, False, 0)
        if de_emp:
            self.post.set_de_emp(rx, de_emp, nfm, snd_rate_12k=abs(fs - 12000.0) < abs(fs - 20250.0), frate=fs)
        self.rx_inc[rx] = int(phase_inc)
        self.audio[rx] = (int(mode), float(lo), float(hi), float(fs), int(de_emp), int(squelch))

    def configure(self, mix):
        """mix: [(WfParams, overlapped, audio phase increment)] per receiver (survey_mix / light_mix)."""
        self.wf.set_tables()
        for rx, (p, ov, inc) in enumerate(mix):
            self.set_wf(rx, p, ov)
            self.set_audio(rx, inc)

    # ---- connections come and go (kg_rxbank_join / _leave)
    def join(self, rx, wf_setting=None, phase_inc=None, **audio):
        """A connection starts on receiver rx: its state starts from zero, nobody else's is touched.  wf_setting = (WfParams,
        overlapped) and phase_inc / **audio (set_audio's arguments) configure it as its first commands would."""
        check(self.lib.kg_rxbank_join(self.h, int(rx)), "kg_rxbank_join")
        if wf_setting is not None:
            self.set_wf(rx, wf_setting[0], wf_setting[1])
        if phase_inc is not None:
            self.set_audio(rx, phase_inc, **audio)

    def set_little_endian(self, rx, little_endian):
        check(self.lib.kg_rxbank_set_little_endian(self.h, int(rx), int(bool(little_endian))), "kg_rxbank_set_little_endian")
        self.little_endian[rx] = bool(little_endian)

    def leave(self, rx):
        check(self.lib.kg_rxbank_leave(self.h, int(rx)), "kg_rxbank_leave")

    def is_active(self, rx):
        return check(self.lib.kg_rxbank_is_active(self.h, int(rx)), "kg_rxbank_is_active") == 1

    def audio_map(self):
        """-> (nrec, nfir, fir_pos, snd_seq) per receiver after the last step"""
        nrec, nfir, pos = (np.zeros(self.nrx, np.int32) for _ in range(3))
        seq = np.zeros(self.nrx, np.uint32)
        check(self.lib.kg_rxbank_audio_map(self.h, ptr(nrec), ptr(nfir), ptr(pos), ptr(seq)), "kg_rxbank_audio_map")
        return nrec, nfir, pos, seq

    def ready(self):
        """False: the next step() would wait for its table slot (the host is KG_RXBANK_SLOTS steps ahead of the GPU)."""
        return check(self.lib.kg_rxbank_ready(self.h), "kg_rxbank_ready") == 1

    # ---- the step
    def step(self, d_adc, adc_ready_event=None):
        """One step over n ADC samples at d_adc (device pointer, int).  Enqueue only.  -> StepInfoC"""
        info = StepInfoC()
        check(self.lib.kg_rxbank_step(self.h, ptr(int(d_adc)), ptr(int(adc_ready_event)) if adc_ready_event else None,
                                      C.byref(info)), "kg_rxbank_step")
        return info

    def step_fast(self, d_adc):
        """step() without the info structure: the timed loop."""
        check(self.lib.kg_rxbank_step(self.h, C.c_void_p(d_adc), None, None), "kg_rxbank_step")

    def sync(self):
        check(self.lib.kg_rxbank_sync(self.h), "kg_rxbank_sync")

    def poll(self):
        return check(self.lib.kg_rxbank_poll(self.h), "kg_rxbank_poll") == 1

    def adc_done(self, stream, steps_back=1):
        """`stream` waits for the readers of the ADC block of the step `steps_back` steps ago (2: the buffer a double-buffered
        ring refills next)."""
        check(self.lib.kg_rxbank_adc_done(self.h, C.c_void_p(int(stream)), int(steps_back)), "kg_rxbank_adc_done")

    def host_profile(self):
        """Where the host's share of the steps since the last call went (text; kg_rxbank_host_profile)."""
        buf = C.create_string_buffer(1024)
        check(self.lib.kg_rxbank_host_profile(self.h, buf, 1024), "kg_rxbank_host_profile")
        return buf.value.decode()

    def frame_map(self):
        """-> (rx_of_frame, frame_off, pkt_bytes) of the last step"""
        rx_of = np.zeros(self.nrx, np.int32)
        off = np.zeros(self.nrx, np.uint64)
        nb = np.zeros(self.nrx, np.int32)
        nf = check(self.lib.kg_rxbank_frame_map(self.h, ptr(rx_of), ptr(off), ptr(nb)), "kg_rxbank_frame_map")
        return rx_of[:nf], off[:nf], nb[:nf]

    # ---- results to the host (after sync())
    def _rows(self, dptr, row_bytes, rows, dtype, shape_tail):
        """rows `rows` (a list of indices) of a [*, row_bytes] device array -> numpy [len(rows), ...]"""
        out = np.zeros((len(rows), row_bytes), np.uint8)
        for i, r in enumerate(rows):
            self.ctx.download(dptr + int(r) * row_bytes, out[i])
        return out.view(dtype).reshape((len(rows),) + shape_tail)

    def fetch(self, what, rows):
        b = self.bufs
        if what == "wf_iq":
            return self._rows(b.wf_iq, b.wf_iq_stride * 4, rows, np.int16, (b.wf_iq_stride, 2))
        if what == "rows":
            return self._rows(b.wf_rows, 1024, rows, np.uint8, (1024,))
        if what == "pkts":
            return self._rows(b.wf_pkts, b.wf_pkt_stride, rows, np.uint8, (b.wf_pkt_stride,))
        if what == "raw":
            return self._rows(b.rx_raw, b.rx_stride * 6, rows, np.uint8, (b.rx_stride * 6,))
        if what == "xin":
            return self._rows(b.rx_in, b.rx_stride * 8, rows, np.float32, (b.rx_stride, 2))
        if what == "firo":
            return self._rows(b.fir_out, b.fir_stride * 8, rows, np.float32, (b.fir_stride, 2))
        if what == "s16":
            return self._rows(b.s16, b.fir_stride * 2, rows, np.int16, (b.fir_stride,))
        if what == "pay":
            return self._rows(b.adpcm, b.fir_stride // 2, rows, np.uint8, (b.fir_stride // 2,))
        if what == "agc":
            return self._rows(b.agc, b.fir_stride * 8, rows, np.float32, (b.fir_stride, 2))
        if what == "iq_pay":
            return self._rows(b.iq_pay, b.fir_stride * 4, rows, np.uint8, (b.fir_stride * 4,))
        raise KeyError(what)
