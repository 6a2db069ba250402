"""Host-side mirror of the reference's acquisition entry points over the C ABI.

Reference (gps/search.cpp)            here
  SearchInit()          :183-350  ->  Searcher(...)/Searcher.search_init()
  Sample()              :382-449  ->  Searcher.sample(packed_bits)
  Correlate(sat, data,  :453-499  ->  Searcher.correlate(sat) -> (snr, lo_shift, ca_shift/DECIM)
            &dop, &i)
  SearchTask() body     :571-575  ->  Searcher.search(sats)  (Sample once, Correlate many,
                                      ca_shift *= DECIM)
All arithmetic runs in libkiwigpu.so on the GPU; nothing here computes.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import sats as _sats
from . import prn as _prn
from ._lib import Context, cell_dtype, check, ptr, result_dtype

NSAMPLES = 65536      # gps/gps.h:73
FFT_LEN = 16384       # gps/gps.h:72
DECIM = 4             # gps/gps.h:62
BIN_SIZE = 249.755859375   # Hz, gps/gps.h:69
# BASELINE.json configs[4]: 10 ms coherent (163680 samples at FS), 65536-point transforms,
# Doppler bin SAMPLE_RATE / 65536 = 62.44 Hz, 256 bins = -128..127 (KG_ACQ10_* in kiwigpu.h)
NSAMPLES_10MS = 163680
FFT_LEN_10MS = 65536
BIN_SIZE_10MS = 4.092e6 / 65536
DOP_LO, DOP_HI = -20, 20   # gps/search.cpp:465
MIN_SIG = 16               # gps/gps.h:60

AcqResult = namedtuple("AcqResult", "sat snr lo_shift ca_shift valid")


class Searcher:
    """GPU acquisition engine for one device (kg_acq)."""

    def __init__(self, ctx=None, max_sats=_sats.MAX_SATS, dop_lo=DOP_LO, dop_hi=DOP_HI,
                 max_blocks=1, device=0, nsamples=NSAMPLES, fft_len=FFT_LEN):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.max_sats, self.dop_lo, self.dop_hi = max_sats, dop_lo, dop_hi
        self.ndop = dop_hi - dop_lo + 1
        self.max_blocks = max_blocks
        self.nsamples, self.fft_len = int(nsamples), int(fft_len)
        h = C.c_void_p()
        if (self.nsamples, self.fft_len) == (NSAMPLES, FFT_LEN):
            check(self.lib.kg_acq_create(self.ctx.h, max_sats, dop_lo, dop_hi, max_blocks,
                                         C.byref(h)), "kg_acq_create")
        else:
            check(self.lib.kg_acq_create_shape(self.ctx.h, max_sats, dop_lo, dop_hi, max_blocks,
                                               self.nsamples, self.fft_len, C.byref(h)), "kg_acq_create_shape")
        self.h = h
        self._last = (0, 0)
        # what the host has written into each row of the code table (the reference's code[sat][]): Correlate() reads the row BEHIND
        # a satellite's own for a negative Doppler bin (kg_acq.hip, acq_code_overrun_kernel), so what row sat + 1 holds matters
        self.rows = {}

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):          # an object must not outlive its context
                self.lib.kg_acq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- SearchInit ---------------------------------------------------------
    def set_code(self, sat, chips, boc=False, limit=None):
        chips = np.ascontiguousarray(chips, np.uint8)
        if limit is None:
            limit = _sats.E1B_LIMIT if boc else _sats.L1_LIMIT
        check(self.lib.kg_acq_set_code(self.h, int(sat), ptr(chips), chips.size, int(bool(boc)),
                                       int(limit)), "kg_acq_set_code")
        self.rows[int(sat)] = ("chips", chips.copy(), bool(boc))

    def set_code_fft(self, sat, code_fft, limit=_sats.L1_LIMIT):
        code_fft = np.ascontiguousarray(code_fft, np.complex64)
        if code_fft.size != self.fft_len:
            raise ValueError("code_fft must hold %d bins" % self.fft_len)
        check(self.lib.kg_acq_set_code_fft(self.h, int(sat), ptr(code_fft), int(limit)),
              "kg_acq_set_code_fft")
        self.rows[int(sat)] = ("fft", code_fft.copy(), False)

    def get_code_fft(self, sat):
        out = np.empty(self.fft_len, np.complex64)
        check(self.lib.kg_acq_get_code_fft(self.h, int(sat), ptr(out)), "kg_acq_get_code_fft")
        return out

    def search_init(self, e1b_hex=None):
        """Build the code table of every row of sats.SATS (SearchInit()).  E1B rows
        need the Galileo memory codes: e1b_hex maps prn -> 1023-digit hex string
        (gps/e1bcode.h:10-60); rows without one are skipped."""
        built = []
        for sat, (prn, t1, t2, kind) in enumerate(_sats.SATS):
            if kind == _sats.E1B:
                if not e1b_hex or prn not in e1b_hex:
                    continue
                self.set_code(sat, _prn.e1b_from_hex(e1b_hex[prn]), boc=True)
            else:
                self.set_code(sat, _prn.cacode(t1, t2), boc=False)
            built.append(sat)
        return built

    # ---- Sample -------------------------------------------------------------
    def sample(self, packed, block=0):
        """nsamples / 8 (8192) bytes of packed 1-bit IF -> data spectrum of `block` (Sample())."""
        if isinstance(packed, int):
            check(self.lib.kg_acq_sample_bits_dev(self.h, block, ptr(packed)),
                  "kg_acq_sample_bits_dev")
            return
        packed = np.ascontiguousarray(packed, np.uint8)
        if packed.size != self.nsamples // 8:
            raise ValueError("need %d bytes of packed samples" % (self.nsamples // 8))
        check(self.lib.kg_acq_sample_bits(self.h, block, ptr(packed)), "kg_acq_sample_bits")
        self.ctx.sync()      # the host buffer may be released by the caller

    def sample_iq16(self, iq, block=0):
        """nsamples (65536) complex int16 samples at the FS/4 IF (extension), host array or device ptr."""
        if isinstance(iq, int):
            check(self.lib.kg_acq_sample_iq16_dev(self.h, block, ptr(iq)),
                  "kg_acq_sample_iq16_dev")
            return
        iq = np.ascontiguousarray(iq, np.int16).reshape(-1)
        if iq.size != 2 * self.nsamples:
            raise ValueError("need %d int16 values" % (2 * self.nsamples))
        check(self.lib.kg_acq_sample_iq16(self.h, block, ptr(iq)), "kg_acq_sample_iq16")
        self.ctx.sync()

    def sample_iq16_host_batch(self, iq, first_block=0):
        """iq: int16 [nblocks, 2 * 65536] in host memory -> data spectra of blocks first_block .. (one
        transfer, one front-end launch; the array may be reused as soon as the call returns)."""
        iq = np.ascontiguousarray(iq, np.int16)
        iq = iq.reshape(-1, 2 * self.nsamples)
        check(self.lib.kg_acq_sample_iq16_batch(self.h, int(first_block), iq.shape[0], ptr(iq), self.nsamples),
              "kg_acq_sample_iq16_batch")

    def sample_iq16_batch(self, d_iq, nblocks, first_block=0, stride_bytes=None):
        """nblocks blocks from one device array (int pointer), one launch set."""
        if stride_bytes is None:
            stride_bytes = 4 * self.nsamples
        check(self.lib.kg_acq_sample_iq16_batch_dev(self.h, int(first_block), int(nblocks),
                                                    ptr(int(d_iq)), int(stride_bytes)),
              "kg_acq_sample_iq16_batch_dev")

    def set_data_fft(self, data_fft, block=0):
        data_fft = np.ascontiguousarray(data_fft, np.complex64)
        if data_fft.size != self.fft_len:
            raise ValueError("data_fft must hold %d bins" % self.fft_len)
        check(self.lib.kg_acq_set_data_fft(self.h, block, ptr(data_fft)), "kg_acq_set_data_fft")

    def get_data_fft(self, block=0):
        out = np.empty(self.fft_len, np.complex64)
        check(self.lib.kg_acq_get_data_fft(self.h, block, ptr(out)), "kg_acq_get_data_fft")
        return out

    def get_data_td(self, block=0):
        out = np.empty(self.fft_len, np.complex64)
        check(self.lib.kg_acq_get_data_td(self.h, block, ptr(out)), "kg_acq_get_data_td")
        return out

    # ---- Correlate ----------------------------------------------------------
    def correlate_async(self, sats, nblocks=1, first_block=0):
        """Enqueue Correlate() over blocks first_block..first_block+nblocks-1."""
        sats = np.ascontiguousarray(sats, np.int32)
        check(self.lib.kg_acq_correlate_blocks_async(self.h, int(first_block), int(nblocks),
                                                     ptr(sats), sats.size),
              "kg_acq_correlate_blocks_async")
        self._last = (int(nblocks), sats.size)

    def fetch(self, want_cells=True):
        nb, ns = self._last
        res = np.zeros((nb, ns), result_dtype)
        cells = np.zeros((nb, ns, self.ndop), cell_dtype) if want_cells else None
        check(self.lib.kg_acq_fetch(self.h, ptr(res), ptr(cells)), "kg_acq_fetch")
        return res, cells

    def correlate_many(self, sats, nblocks=1, want_cells=True, first_block=0):
        self.correlate_async(sats, nblocks, first_block)
        return self.fetch(want_cells)

    def correlate(self, sat, block_data=None):
        """Correlate(sat, data, &max_snr_dop, &max_snr_i) -> (snr, dop, idx, valid)."""
        if block_data is not None:
            self.set_data_fft(block_data, 0)
        res, _ = self.correlate_many([sat], 1, want_cells=False)
        r = res[0, 0]
        return float(r["snr"]), int(r["dop"]), int(r["idx"]), int(r["valid"])

    def results_dev(self):
        return self.lib.kg_acq_results_dev(self.h)

    # ---- SearchTask body ----------------------------------------------------
    def search(self, sats, packed=None, iq16=None, lo_shift=0, ca_shift=0):
        """One pass of the SearchTask loop body (:571-575) over `sats` for one sample
        block: Sample(); Correlate(); ca_shift *= DECIM.  lo_shift/ca_shift carry
        over when a result is not valid, as the reference's locals do (:513,:495)."""
        if packed is not None:
            self.sample(packed)
        elif iq16 is not None:
            self.sample_iq16(iq16)
        res, _ = self.correlate_many(list(sats), 1, want_cells=False)
        out = []
        for sat, r in zip(sats, res[0]):
            if r["valid"]:
                lo_shift, ca_shift = int(r["dop"]), int(r["idx"]) * DECIM
            out.append(AcqResult(int(sat), float(r["snr"]), lo_shift, ca_shift, int(r["valid"])))
        return out
