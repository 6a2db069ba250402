"""Host-side mirror of what consumes the CFastFIR output in c2s_sound(), over the C ABI.

Reference                                                              here
  sMeterAlpha / sMeterAvg_dB loop   rx/rx_sound.cpp:248-250, 676-696 -> Post.set_smeter, Post.smeter
  m_Agc[ch].SetParameters(...)      rx/CuteSDR/agc.cpp:98-163        -> Post.set_agc
  m_Agc[ch].GetDelaySamples()       rx/CuteSDR/agc.h:27              -> Post.agc_delay
  m_Agc[ch].ProcessData(n, in, out) rx/CuteSDR/agc.cpp:259-292       -> Post.process (modes IQ / SSB)
  AM detector + DC removal          rx/rx_sound.cpp:766-783          -> Post.process (mode AM)
  m_AM_FIR.InitLPFilter(...)        rx/rx_sound_cmd.cpp:268-282      -> Post.set_am_passband (Post.cfir_init_lp)
  m_AM_FIR.ProcessFilter            rx/rx_sound.cpp:787              -> Post.process (mode AM, s16)
  NBFM fmdemod_quadri + clipper     rx/rx_sound.cpp:845-875          -> Post.process (mode NBFM)
  m_Squelch.SetupParameters / SetSquelch / Reset
                                    rx/rx_sound.cpp:261-262, rx/rx_sound_cmd.cpp:238,430
                                                                     -> Post.squelch_setup / squelch_set / squelch_reset
  m_Squelch.PerformFMSquelch        rx/rx_sound.cpp:876-877          -> Post.process (mode NBFM, s16), Post.squelch_state
  "SET de_emp=%d nfm=%d"            rx/rx_sound_cmd.cpp:543-585      -> Post.set_de_emp (tables: deemp.py)
  m_*_deemp_FIR.ProcessFilter       rx/rx_sound.cpp:898-907          -> Post.process (s16, in place)
"""
import ctypes as C

import numpy as np

from ._lib import Context, check, ptr, own_rows

MODE_IQ, MODE_SSB, MODE_AM, MODE_NBFM = range(4)      # KG_POST_* of include/kiwigpu.h
MAX_SAMPLES = 1024                                    # KG_POST_MAX_SAMPLES
CFIR_AM, CFIR_DEEMP_NFM, CFIR_DEEMP_AM_SSB, CFIR_SQUELCH_HP = range(4)  # KG_CFIR_*
CFIR_REAL_REAL, CFIR_REAL_MONO16, CFIR_MONO16_MONO16 = range(3)        # the ProcessFilter overloads


class Post:
    """S-meter + CAgc + detector state of nchan receiver channels on the GPU (kg_post)."""

    def __init__(self, ctx=None, nchan=4, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan = nchan
        h = C.c_void_p()
        check(self.lib.kg_post_create(self.ctx.h, int(nchan), C.byref(h)), "kg_post_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None) and not getattr(self, "_borrowed", False):          # an object must not outlive its context
                self.lib.kg_post_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_agc(self, ch, agc_on, use_hang, threshold, manual_gain, slope, decay, sample_rate):
        check(self.lib.kg_post_set_agc(self.h, int(ch), int(bool(agc_on)), int(bool(use_hang)), int(threshold),
                                       int(manual_gain), int(slope), int(decay), float(sample_rate)),
              "kg_post_set_agc")

    def agc_delay(self, ch):
        return check(self.lib.kg_post_agc_delay(self.h, int(ch)), "kg_post_agc_delay")

    def set_smeter(self, ch, frate):
        check(self.lib.kg_post_set_smeter(self.h, int(ch), float(frate)), "kg_post_set_smeter")

    def set_mode(self, ch, mode):
        check(self.lib.kg_post_set_mode(self.h, int(ch), int(mode)), "kg_post_set_mode")

    def reset(self, ch):
        check(self.lib.kg_post_reset(self.h, int(ch)), "kg_post_reset")

    # ---- CFir objects (rx/CuteSDR/fir.cpp) ----
    def cfir_init_lp(self, ch, which, numtaps, scale, astop, fpass, fstop, fs):
        """CFir::InitLPFilter -> tap count"""
        return check(self.lib.kg_post_cfir_init_lp(self.h, int(ch), int(which), int(numtaps), float(scale), float(astop), float(fpass),
                                                   float(fstop), float(fs)), "kg_post_cfir_init_lp")

    def cfir_init_const(self, ch, which, coef, fs=12000.0):
        """CFir::InitConstFir -> tap count"""
        coef = np.ascontiguousarray(coef, np.float32)
        return check(self.lib.kg_post_cfir_init_const(self.h, int(ch), int(which), coef.size, ptr(coef), float(fs)),
                     "kg_post_cfir_init_const")

    def cfir_taps(self, ch, which):
        taps = np.zeros(97, np.float32)
        n = check(self.lib.kg_post_cfir_get_taps(self.h, int(ch), int(which), ptr(taps)), "kg_post_cfir_get_taps")
        return taps[:n].copy()

    def cfir_process(self, chans, which, kind, x):
        """m_*_FIR[ch].ProcessFilter on host rows x [len(chans), n] (float32, or int16 for CFIR_MONO16_MONO16)."""
        own_rows(self, "cfir_process()")
        chans = np.ascontiguousarray(chans, np.int32)
        x = np.ascontiguousarray(x, np.int16 if kind == CFIR_MONO16_MONO16 else np.float32).reshape(chans.size, -1)
        n = x.shape[1]
        out = np.zeros((chans.size, n), np.float32 if kind == CFIR_REAL_REAL else np.int16)
        ctx = self.ctx
        bi, bo = ctx.alloc(x.nbytes), ctx.alloc(out.nbytes)
        try:
            ctx.upload(bi, x)
            check(self.lib.kg_post_cfir_process_dev(self.h, ptr(chans), chans.size, int(which), int(kind), ptr(int(bi)), n, n, ptr(int(bo)), n),
                  "kg_post_cfir_process_dev")
            ctx.sync()
            ctx.download(bo, out)
        finally:
            ctx.free(bi)
            ctx.free(bo)
        return out

    def squelch_perform(self, chans, x):
        """m_Squelch[ch].PerformFMSquelch on host rows x float32 [len(chans), n] -> (mono16 [len(chans), n], nsq_nc_sq int32[len(chans)])"""
        own_rows(self, "squelch_perform()")
        chans = np.ascontiguousarray(chans, np.int32)
        x = np.ascontiguousarray(x, np.float32).reshape(chans.size, -1)
        n = x.shape[1]
        out = np.zeros((chans.size, n), np.int16)
        ctx = self.ctx
        bi, bo = ctx.alloc(x.nbytes), ctx.alloc(out.nbytes)
        try:
            ctx.upload(bi, x)
            check(self.lib.kg_post_squelch_perform_dev(self.h, ptr(chans), chans.size, ptr(int(bi)), n, n, ptr(int(bo)), n),
                  "kg_post_squelch_perform_dev")
            ctx.sync()
            ctx.download(bo, out)
        finally:
            ctx.free(bi)
            ctx.free(bo)
        return out, self.squelch_state(chans)[0]

    def set_am_passband(self, ch, locut, hicut, frate):
        """The m_AM_FIR design of a passband change (rx/rx_sound_cmd.cpp:268-282) -> tap count"""
        return check(self.lib.kg_post_set_am_passband(self.h, int(ch), float(locut), float(hicut), float(frate)), "kg_post_set_am_passband")

    def set_deemp(self, ch, nfm, de_emp):
        check(self.lib.kg_post_set_deemp(self.h, int(ch), int(bool(nfm)), int(de_emp)), "kg_post_set_deemp")

    def set_de_emp(self, ch, de_emp, nfm, snd_rate_12k=True, frate=None):
        """`SET de_emp=<de_emp> nfm=<nfm>` as rx/rx_sound_cmd.cpp:543-585 handles it: the flag, and for de_emp 1 / 2 the
        coefficients of rx/rx_filter.h's table into the mode's filter (InitConstFir clears its samples)."""
        from . import deemp
        self.set_deemp(ch, nfm, de_emp)
        if de_emp:
            rate = frate if frate is not None else (12000.0 if snd_rate_12k else 20250.0)
            self.cfir_init_const(ch, CFIR_DEEMP_NFM if nfm else CFIR_DEEMP_AM_SSB, deemp.table(nfm, snd_rate_12k)[de_emp - 1], rate)

    # ---- CSquelch (rx/CuteSDR/squelch.cpp) ----
    def squelch_setup(self, ch, samplerate):
        check(self.lib.kg_post_squelch_setup(self.h, int(ch), float(samplerate)), "kg_post_squelch_setup")

    def squelch_set(self, ch, value, squelch_max=0):
        check(self.lib.kg_post_squelch_set(self.h, int(ch), int(value), int(squelch_max)), "kg_post_squelch_set")

    def squelch_reset(self, ch):
        check(self.lib.kg_post_squelch_reset(self.h, int(ch)), "kg_post_squelch_reset")

    def squelch_state(self, chans):
        """-> (nsq_nc_sq int32[n], squelched int32[n], m_SquelchAve float32[n]) after the last pass"""
        chans = np.ascontiguousarray(chans, np.int32)
        rc = np.zeros(chans.size, np.int32)
        sq = np.zeros(chans.size, np.int32)
        ave = np.zeros(chans.size, np.float32)
        check(self.lib.kg_post_squelch_state(self.h, ptr(chans), chans.size, ptr(rc), ptr(sq), ptr(ave)), "kg_post_squelch_state")
        return rc, sq, ave

    def process_dev(self, chans, d_fir, in_stride, nsamps, d_s16=0, d_demod=0, d_agc=0, out_stride=None):
        chans = np.ascontiguousarray(chans, np.int32)
        check(self.lib.kg_post_process_dev(self.h, ptr(chans), chans.size, ptr(int(d_fir)), int(in_stride),
                                           int(nsamps), ptr(int(d_s16)) if d_s16 else None,
                                           ptr(int(d_demod)) if d_demod else None,
                                           ptr(int(d_agc)) if d_agc else None,
                                           int(out_stride if out_stride is not None else nsamps)),
              "kg_post_process_dev")

    def process(self, chans, x):
        """x: complex64 [len(chans), n] FIR output (host).  -> (s16 int16, demod float32, agc complex64),
        each [len(chans), n]; a row is meaningful where the channel's mode produces it."""
        own_rows(self, "process()")
        chans = np.ascontiguousarray(chans, np.int32)
        x = np.ascontiguousarray(x, np.complex64).reshape(chans.size, -1)
        n = x.shape[1]
        s16 = np.zeros((chans.size, n), np.int16)
        demod = np.zeros((chans.size, n), np.float32)
        agc = np.zeros((chans.size, n), np.complex64)
        ctx = self.ctx
        bufs = [ctx.alloc(a.nbytes) for a in (x, s16, demod, agc)]
        try:
            ctx.upload(bufs[0], x)
            for b, a in zip(bufs[1:], (s16, demod, agc)):
                ctx.upload(b, a)
            self.process_dev(chans, bufs[0], n, n, bufs[1], bufs[2], bufs[3], n)
            ctx.sync()
            for b, a in zip(bufs[1:], (s16, demod, agc)):
                ctx.download(b, a)
        finally:
            for b in bufs:
                ctx.free(b)
        return s16, demod, agc

    def smeter(self, chans):
        """-> (sMeterAvg_dB float32[len(chans)], taps float32[len(chans), 2])"""
        chans = np.ascontiguousarray(chans, np.int32)
        avg = np.zeros(chans.size, np.float32)
        taps = np.zeros((chans.size, 2), np.float32)
        check(self.lib.kg_post_smeter(self.h, ptr(chans), chans.size, ptr(avg), ptr(taps)), "kg_post_smeter")
        return avg, taps


MATH_LOG10F, MATH_POWF, MATH_EXPF = 0, 1, 2          # KG_MATH_*


def math_dev(ctx, fn, x=None, first_bits=0, n=None, base=10.0):
    """kg_math_dev: the device's log10f / powf(base, .) / expf -- the host libm's algorithms, csrc/kg_libm.h -- over an array (x), or
    over the n floats whose bit patterns start at first_bits.  -> float32[n]"""
    if x is not None:
        x = np.ascontiguousarray(x, np.float32)
        n = x.size
    n = int(n)
    out = np.empty(n, np.float32)
    d_y = ctx.alloc(4 * n)
    d_x = ctx.alloc(4 * n) if x is not None else 0
    try:
        if x is not None:
            ctx.upload(d_x, x)
        check(ctx.lib.kg_math_dev(ctx.h, int(fn), float(base), C.c_void_p(d_x) if d_x else None, int(first_bits) & 0xFFFFFFFF, n,
                                  C.c_void_p(d_y)), "kg_math_dev")
        ctx.sync()
        ctx.download(d_y, out)
    finally:
        ctx.free(d_y)
        if d_x:
            ctx.free(d_x)
    return out


def log10f(ctx, x=None, first_bits=0, n=None):
    return math_dev(ctx, MATH_LOG10F, x, first_bits, n)
