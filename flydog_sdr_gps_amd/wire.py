"""Host-side mirror of the wire formats over the C ABI.

Reference                                                                    here
  encode_ima_adpcm_i16_e8(out_samps_s2, bp, ns_out, &s->adpcm_snd)
                                  rx/csdr/ima_adpcm.cpp:185, rx_sound.cpp:1122 -> Adpcm.encode
  s->adpcm_snd (index, previousValue), "audio_adpcm_state"   rx_sound.cpp:1314 -> Adpcm.get_state / set_state
  uncompressed payload, LE / network order                   rx_sound.cpp:1126-1140 -> snd_payload
  snd_pkt_real_t header                     rx_sound.h:42-48, rx_sound.cpp:1219-1254 -> snd_header
  wf_pkt_t + encode_ima_adpcm_u8_e8 of pad + row
                                  rx_waterfall.h:73-89, rx_waterfall.cpp:1602-1639 -> wf_packets
"""
import ctypes as C

import numpy as np

from ._lib import Context, check, ptr, own_rows

WF_PKT_HDR, WF_ADPCM_PAD, WF_PKT_MAX = 16, 10, 16 + 10 + 1024
WF_FLAGS_COMPRESSION = 0x00010000                      # rx_waterfall.h:77
SND_FLAG_LPF, SND_FLAG_ADC_OVFL, SND_FLAG_NEW_FREQ, SND_FLAG_MODE_IQ = 0x01, 0x02, 0x04, 0x08   # rx_sound.cpp:461-468
SND_FLAG_COMPRESSED, SND_FLAG_RESTART, SND_FLAG_SQUELCH_UI, SND_FLAG_LITTLE_ENDIAN = 0x10, 0x20, 0x40, 0x80


class WfPktInfo(C.Structure):
    _fields_ = [("x_bin_server", C.c_uint32), ("zoom", C.c_uint32), ("seq", C.c_uint32),
                ("use_compression", C.c_int32)]


class Adpcm:
    """The adpcm_snd coder state of nchan sound connections on the GPU (kg_adpcm)."""

    def __init__(self, ctx=None, nchan=4, device=0):
        self.ctx = ctx if ctx is not None else Context(device)
        self.lib = self.ctx.lib
        self.nchan = nchan
        h = C.c_void_p()
        check(self.lib.kg_adpcm_create(self.ctx.h, int(nchan), C.byref(h)), "kg_adpcm_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None) and not getattr(self, "_borrowed", False):          # an object must not outlive its context
                self.lib.kg_adpcm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, ch, index=0, previous=0):
        check(self.lib.kg_adpcm_set_state(self.h, int(ch), int(index), int(previous)), "kg_adpcm_set_state")

    def get_state(self, ch):
        i, p = C.c_int(), C.c_int()
        check(self.lib.kg_adpcm_get_state(self.h, int(ch), C.byref(i), C.byref(p)), "kg_adpcm_get_state")
        return i.value, p.value

    def encode_dev(self, chans, d_s16, in_stride, nsamps, d_out, out_stride):
        chans = np.ascontiguousarray(chans, np.int32)
        check(self.lib.kg_adpcm_encode_dev(self.h, ptr(chans), chans.size, ptr(int(d_s16)), int(in_stride),
                                           int(nsamps), ptr(int(d_out)), int(out_stride)), "kg_adpcm_encode_dev")

    def encode(self, chans, x):
        """x: int16 [len(chans), n] (host) -> uint8 [len(chans), n/2]"""
        own_rows(self, "encode()")
        chans = np.ascontiguousarray(chans, np.int32)
        x = np.ascontiguousarray(x, np.int16).reshape(chans.size, -1)
        n = x.shape[1]
        out = np.zeros((chans.size, n // 2), np.uint8)
        ctx = self.ctx
        d_in, d_out = ctx.alloc(max(x.nbytes, 2)), ctx.alloc(max(out.nbytes, 1))
        try:
            ctx.upload(d_in, x)
            self.encode_dev(chans, d_in, n, n, d_out, n // 2)
            ctx.sync()
            ctx.download(d_out, out)
        finally:
            ctx.free(d_in)
            ctx.free(d_out)
        return out


def snd_header(ctx, flags, seq, smeter_dBm):
    h = np.zeros(10, np.uint8)
    ctx.lib.kg_snd_header(int(flags) & 0xFF, int(seq) & 0xFFFFFFFF, float(smeter_dBm), ptr(h))
    return h


def snd_iq_payload(ctx, x, little_endian):
    """x: complex64 [nch, n] (host: out_samps_c of the IQ modes, rx_sound.cpp:1076-1096) -> uint8 [nch, 4n]"""
    x = np.ascontiguousarray(x, np.complex64)
    x = x.reshape(1, -1) if x.ndim == 1 else x
    out = np.zeros((x.shape[0], 4 * x.shape[1]), np.uint8)
    d_in, d_out = ctx.alloc(x.nbytes), ctx.alloc(out.nbytes)
    try:
        ctx.upload(d_in, x)
        check(ctx.lib.kg_snd_iq_payload_dev(ctx.h, None, x.shape[0], C.c_void_p(d_in), x.shape[1], x.shape[1],
                                            int(bool(little_endian)), C.c_void_p(d_out), 4 * x.shape[1]), "kg_snd_iq_payload_dev")
        ctx.sync()
        ctx.download(d_out, out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    return out


def snd_payload(ctx, x, little_endian):
    """x: int16 [nch, n] (host) -> uint8 [nch, 2n]"""
    x = np.ascontiguousarray(x, np.int16)
    x = x.reshape(1, -1) if x.ndim == 1 else x
    out = np.zeros((x.shape[0], 2 * x.shape[1]), np.uint8)
    d_in, d_out = ctx.alloc(x.nbytes), ctx.alloc(out.nbytes)
    try:
        ctx.upload(d_in, x)
        check(ctx.lib.kg_snd_payload_dev(ctx.h, C.c_void_p(d_in), x.shape[1], x.shape[0], x.shape[1],
                                         int(bool(little_endian)), C.c_void_p(d_out), 2 * x.shape[1]),
              "kg_snd_payload_dev")
        ctx.sync()
        ctx.download(d_out, out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    return out


def wf_packets_dev(ctx, d_rows, row_stride, infos, d_pkts, pkt_stride=WF_PKT_MAX):
    """infos: list of (x_bin_server, zoom, seq, use_compression).  -> wire byte count per packet."""
    arr = (WfPktInfo * len(infos))(*[WfPktInfo(int(a), int(b), int(c), int(bool(d))) for a, b, c, d in infos])
    nb = np.zeros(len(infos), np.int32)
    check(ctx.lib.kg_wf_packets_dev(ctx.h, C.c_void_p(int(d_rows)), int(row_stride), len(infos), arr,
                                    C.c_void_p(int(d_pkts)), int(pkt_stride), ptr(nb)), "kg_wf_packets_dev")
    return nb


def wf_packets(ctx, rows, infos):
    """rows: uint8 [nrows, 1024] (host) -> list of packet byte arrays as they go on the wire"""
    rows = np.ascontiguousarray(rows, np.uint8).reshape(len(infos), 1024)
    pk = np.zeros((len(infos), WF_PKT_MAX), np.uint8)
    d_rows, d_pk = ctx.alloc(rows.nbytes), ctx.alloc(pk.nbytes)
    try:
        ctx.upload(d_rows, rows)
        ctx.upload(d_pk, pk)
        nb = wf_packets_dev(ctx, d_rows, 1024, infos, d_pk)
        ctx.sync()
        ctx.download(d_pk, pk)
    finally:
        ctx.free(d_rows)
        ctx.free(d_pk)
    return [pk[i, :nb[i]].copy() for i in range(len(infos))]


class GpsState(C.Structure):
    """snd_t::gpssec, last_gpssec, gps_init (rx/rx_sound.h:120-122) of one sound connection."""
    _fields_ = [("gpssec", C.c_double), ("last_gpssec", C.c_double), ("gps_init", C.c_int32), ("pad", C.c_int32)]


class IqStamp(C.Structure):
    _fields_ = [("gpssec", C.c_uint32), ("gpsnsec", C.c_uint32), ("last_gps_solution", C.c_uint8), ("pad", C.c_uint8 * 3)]


def gps_begin(lib, st, clk_gps_secs, dticks, adc_clock_base, gps_delay, gps_delay2):
    """rx/rx_sound.cpp:557, once per data-pump buffer (host arithmetic in libkiwigpu)."""
    lib.kg_snd_gps_begin(C.byref(st), float(clk_gps_secs), float(dticks), float(adc_clock_base), float(gps_delay),
                         float(gps_delay2))


def gps_stamp(lib, st, norm_nrx_samps, fir_pos, agc_on, agc_delay, rx_decim, adc_clock_base, clk_gps_secs, clk_ticks):
    """rx/rx_sound.cpp:636-661, once per FIR output block -> (gpssec, gpsnsec, last_gps_solution)"""
    o = IqStamp()
    lib.kg_snd_gps_stamp(C.byref(st), int(norm_nrx_samps), int(fir_pos), int(bool(agc_on)), int(agc_delay),
                         int(rx_decim), float(adc_clock_base), float(clk_gps_secs), int(clk_ticks), C.byref(o))
    return o.gpssec, o.gpsnsec, o.last_gps_solution
