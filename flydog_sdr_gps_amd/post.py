"""Host-side mirror of what consumes the CFastFIR output in c2s_sound(), over the C ABI.

Reference                                                              here
  sMeterAlpha / sMeterAvg_dB loop   rx/rx_sound.cpp:248-250, 676-696 -> Post.set_smeter, Post.smeter
  m_Agc[ch].SetParameters(...)      rx/CuteSDR/agc.cpp:98-163        -> Post.set_agc
  m_Agc[ch].GetDelaySamples()       rx/CuteSDR/agc.h:27              -> Post.agc_delay
  m_Agc[ch].ProcessData(n, in, out) rx/CuteSDR/agc.cpp:259-292       -> Post.process (modes IQ / SSB)
  AM detector + DC removal          rx/rx_sound.cpp:766-783          -> Post.process (mode AM)
  NBFM fmdemod_quadri + clipper     rx/rx_sound.cpp:845-881          -> Post.process (mode NBFM)
"""
import ctypes as C

import numpy as np

from ._lib import Context, check, ptr

MODE_IQ, MODE_SSB, MODE_AM, MODE_NBFM = range(4)      # KG_POST_* of include/kiwigpu.h
MAX_SAMPLES = 1024                                    # KG_POST_MAX_SAMPLES


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
