"""Input data shared by the tests and bench.py (never imported by the product package)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def e1b_chips(path=None):
    """{prn: chips uint8[4092]} from tests/golden/e1b_ref.npz: the 50 Galileo E1-B memory codes as the
    reference's own gps/e1bcode.h produced them (tools/make_ref_golden.py).  In a deployment the caller
    hands over its own table."""
    g = np.load(path or os.path.join(GOLDEN, "e1b_ref.npz"))
    chips = np.unpackbits(g["chips_packed"], axis=1)[:, :4092]
    return {i + 1: chips[i].copy() for i in range(chips.shape[0])}


def run_fir_script(fir, script, x):
    """The script language of oracle/ref/ref_fir_main.cpp on an object with init_lp / init_hp / init_const /
    process_rr / process_rm / process_mm.  Returns the floats the reference driver would have written."""
    import numpy as np
    out, pos = [], 0
    for line in script:
        f = str(line).split()
        if f[0] in "LH":
            fn = fir.init_lp if f[0] == "L" else fir.init_hp
            out.append(np.array([fn(int(f[1]), *[float(v) for v in f[2:7]])], np.float32))
        elif f[0] == "K":
            n = int(f[1])
            fir.init_const(x[pos:pos + n], float(f[2]))
            pos += n
        else:
            n = int(f[1])
            blk = x[pos:pos + n]
            pos += n
            if f[0] == "R":
                out.append(np.asarray(fir.process_rr(blk), np.float32))
            elif f[0] == "M":
                out.append(np.asarray(fir.process_rm(blk)).astype(np.float32))
            else:
                out.append(np.asarray(fir.process_mm(blk.astype(np.int16))).astype(np.float32))
    assert pos == x.size
    return np.concatenate(out)


def run_squelch_script(sq, script, x):
    """The script language of oracle/ref/ref_squelch_main.cpp on an object with setup / set_squelch / reset /
    perform_fm (-> mono16 out, nsq_nc_sq)."""
    import numpy as np
    out, pos = [], 0
    for line in script:
        f = str(line).split()
        if f[0] == "P":
            sq.setup(float(f[1]))
        elif f[0] == "Q":
            sq.set_squelch(int(f[1]), int(f[2]))
        elif f[0] == "Z":
            sq.reset()
        else:
            n = int(f[1])
            y, rc = sq.perform_fm(x[pos:pos + n])
            pos += n
            out.append(np.concatenate([np.asarray(y).astype(np.float32), np.array([rc], np.float32)]))
    assert pos == x.size
    return np.concatenate(out)


def am_passband(lo, hi, rate):
    """The passband statements of the `SET mod= low_cut= high_cut=` handler (rx/rx_sound_cmd.cpp:248-250, 268-282; pinned by the
    reference's own statements: tests/golden/sndcmd_ref.npz): -> (locut, hicut as clamped to +-(int)(rate / 2 - 1), hbw, stop of m_AM_FIR)"""
    import numpy as np
    fmax = int(rate / 2 - 1)
    hi, lo = min(float(hi), float(fmax)), max(float(lo), float(-fmax))
    hbw = np.float32(max(np.float32(abs(hi)), np.float32(abs(lo))))
    if float(hbw) > rate / 2:
        hbw = np.float32(rate / 2)
    stop = np.float32(float(hbw) * 1.8)
    if float(stop) > rate / 2:
        stop = np.float32(rate / 2)
    return lo, hi, hbw, stop


def arm_audio_tail(P, ch, rate=12000.0, hbw=4900.0, squelch=0):
    """What the reference has always done before a channel can demodulate AM / NBFM: the post-detector filter of the passband
    command (rx/rx_sound_cmd.cpp:268-282) and the squelch of a new connection (rx/rx_sound.cpp:261-262)."""
    P.set_am_passband(ch, -hbw, hbw, rate)
    P.squelch_setup(ch, rate)
    P.squelch_set(ch, squelch, 0)


def run_fastfir_script(f, script, x):
    """The script language of oracle/ref/ref_fastfir_main.cpp on an object with window(w) / cic(on) / setup(inst, lo, hi, off, fs) /
    process(x) -> (out complex64, FirPos).  Returns the floats the reference driver would have written."""
    import numpy as np
    out, pos = [], 0
    for line in script:
        t = str(line).split()
        if t[0] == "W":
            f.window(int(t[1]))
        elif t[0] == "C":
            f.cic(int(t[1]) != 0)
        elif t[0] == "P":
            f.setup(int(t[1]), *[float(v) for v in t[2:6]])
        else:
            n = int(t[1])
            y, fp = f.process(x[pos:pos + n])
            pos += n
            out.append(np.array([y.size, fp], np.float32))
            out.append(np.ascontiguousarray(y, np.complex64).view(np.float32))
    assert pos == x.size
    return np.concatenate(out)


def fastfir_blocks(script, flat):
    """Splits the driver's float stream into [(count, FirPos, complex64 outputs)] per `D` line."""
    import numpy as np
    blocks, k = [], 0
    for line in script:
        if str(line).split()[0] != "D":
            continue
        cnt, fp = int(flat[k]), int(flat[k + 1])
        k += 2
        blocks.append((cnt, fp, flat[k:k + 2 * cnt].view(np.complex64)))
        k += 2 * cnt
    assert k == flat.size
    return blocks


def oracle_row(oracle, searcher, sat):
    """The oracle's version of what the host wrote into row `sat` of a Searcher's code table (Searcher.rows), or None when the
    row was never written -- the `code_next` of sat - 1 (the reference's Correlate() reads the next satellite's row behind a
    satellite's own for a negative Doppler bin: gps/search.cpp:471 over the doubled rows of :54)."""
    r = searcher.rows.get(int(sat))
    if r is None or sat >= searcher.max_sats:
        return None
    kind, arr, boc = r
    return oracle.code_fft(arr, boc=boc, fft_len=searcher.fft_len) if kind == "chips" else arr


def oracle_next_rows(oracle, searcher, svs):
    return [oracle_row(oracle, searcher, int(s) + 1) for s in svs]


def dpump_ref_cases(g):
    """tests/golden/dpump_ref.npz -> [(name, nch, ns, inversion, dc_i, dc_q, enabled list, bufs uint8[nbuf, bytes], rescale,
    [per buffer: {ch: (wr_pos after, ticks48, complex64[ns])}])] as the reference's own snd_service() produced them."""
    import numpy as np
    out = []
    for name in (str(n) for n in g["names"]):
        nch, ns, inv, dci, dcq, mask, nbuf = g[name + "_cfg"]
        nch, ns, inv, mask, nbuf = int(nch), int(ns), int(inv), int(mask), int(nbuf)
        en = [(mask >> ch) & 1 for ch in range(nch)]
        y, k = g[name + "_out"], 1
        per = []
        for b in range(nbuf):
            d = {}
            for ch in range(nch):
                if not en[ch]:
                    continue
                wr, hi, lo = (int(v) for v in y[k:k + 3])
                k += 3
                d[ch] = (wr, (hi << 24) | lo, y[k:k + 2 * ns].view(np.complex64))
                k += 2 * ns
            per.append(d)
        assert k == y.size
        out.append((name, nch, ns, bool(inv), float(dci), float(dcq), en, g[name + "_bufs"], float(y[0]), per))
    return out


def aper_ref_replay(g, update, report):
    """Walks tests/golden/aper_fftref.npz -- one wf_inst_t's aperture fields over the frames the reference's compute_frame() ->
    aperture_auto() (rx/rx_waterfall.cpp:1173-1272) was given -- with the caller's two halves:
        update(avg, row, algo, param, clear, start, stop, waterfall_cal) -> avg      (:1183-1222)
        report(avg, start, stop) -> (signal, noise)                                   (:1233-1271)
    and the control flow between them restated here (need / done counters :1176, the audio FFT's pixel range :1180-1181, the
    single-shot rule :1177, :1192-1195, the report clock :1224-1229).  Yields (frame index, avg, (signal, noise, done,
    report_sec, avg_clear), expected avg, expected state) after every frame."""
    import numpy as np
    IIR, MMA, EMA, OFF = 0, 1, 2, 3
    avg = np.zeros(1024, np.float32)
    st = dict(done=0, clear=0, signal=0, noise=0, report_sec=0)
    for k, (run, f, algo, param, cal, nwf, now, need) in enumerate(g["frames"]):
        algo, cal, nwf, now, need, f = int(algo), int(cal), int(nwf), int(now), int(need), int(f)
        if f == 0:
            st["clear"] = 1
        row = g["rows"][k]
        if need > st["done"]:
            single = algo == OFF
            start, stop = (256, 768) if 0 >= nwf else (0, 1024)
            if st["clear"]:
                avg = update(avg, row, MMA, 8.0, True, start, stop, cal)
                st["report_sec"], st["clear"] = now, 0
            else:
                a, p = (MMA, 8.0) if single else (algo, float(param))
                avg = update(avg, row, a, p, False, start, stop, cal)
            if now >= st["report_sec"] + (1 if single else 3):
                st["report_sec"] = now
                st["done"] += 1
                st["signal"], st["noise"] = report(avg, start, stop)
        yield k, avg, (st["signal"], st["noise"], st["done"], st["report_sec"], st["clear"]), g["avg_pwr"][k], \
            tuple(int(v) for v in g["states"][k])


# ---- c2s_sound()'s signal path as the reference's own statements ran it (tests/golden/sndpath_ref.npz) -------------------------
SND_AM_MODES, SND_FM_MODES, SND_SSB_MODES, SND_IQ_MODES = (0, 1), (6, 16), (2, 3, 4, 5, 9, 10), (7, 8)      # rx/mode.h:69-70
S_METER_CAL = -13
GPS_DELAY = 28926.838e-6                                  # rx_sound.cpp:92 (tests/test_ref_text_pins_cpu.py compares it by text)


def time_diff48(nxt, prev):
    """support/timing.cpp:90-100 (the host's; the library takes dticks as an argument)"""
    return nxt - prev if nxt >= prev else (0x0000FFFFFFFFFFFF - prev) + nxt
SND_FLAG_ADC_OVFL, SND_FLAG_MODE_IQ, SND_FLAG_COMPRESSED, SND_FLAG_SQUELCH_UI, SND_FLAG_LITTLE_ENDIAN = 0x02, 0x08, 0x10, 0x40, 0x80   # rx_sound.cpp:461-468


class OracleSoundPath:
    """One connection's path from the CFastFIR output to the bytes of the sound packet, oracle side, driven by sndpath_ref's script."""

    def __init__(self, ko, rate):
        import numpy as np
        self.ko, self.rate, self.np = ko, rate, np
        self.a, self.am, self.sq, self.de_am, self.de_nfm = ko.Agc(), ko.CFir(), ko.Squelch(), ko.CFir(), ko.CFir()
        self.sq.setup(rate)
        self.sq.set_squelch(0, 0)
        self.alpha, self.avg, self.z1, self.last = ko.smeter_alpha(rate), 0.0, 0.0, (0.0, 0.0)
        self.mode, self.de, self.de_fm, self.squelched = 2, 0, 0, False
        self.comp, self.le, self.ad = 1, False, None

    def agc(self, *prm):
        self.a.set_parameters(*prm, self.rate)

    def passband(self, lo, hi, hbw, stop):
        self.am.init_lp(0, 1.0, 50.0, self.np.float32(hbw), self.np.float32(stop), self.rate)

    def squelch(self, v, mx):
        self.sq.set_squelch(v, mx)

    def de_emp(self, de, de_fm):
        from flydog_sdr_gps_amd import deemp
        r12k = abs(self.rate - 12000.0) < abs(self.rate - 20250.0)
        self.de, self.de_fm = de, de_fm
        if de:
            self.de_am.init_const(deemp.table(False, r12k)[de - 1], self.rate)
        if de_fm:
            self.de_nfm.init_const(deemp.table(True, r12k)[de_fm - 1], self.rate)

    def set_mode(self, m):
        self.mode = m

    def wire(self, comp, le):
        self.comp, self.le = comp, bool(le)

    def block(self, x):
        ko, np = self.ko, self.np
        self.avg, taps = ko.smeter_process(self.avg, self.alpha, x)
        out = None
        if self.mode in SND_AM_MODES:
            d, self.z1 = ko.am_detect(self.z1, self.a.process_cpx(x))
            out = self.am.process_rm(d)
        elif self.mode in SND_FM_MODES:
            d, self.last = ko.nbfm_detect(self.last, self.a.process_cpx(x))
            out, rc = self.sq.perform_fm(d)
            if rc != 0:
                self.squelched = rc == 1
        elif self.mode in SND_SSB_MODES:
            out = self.a.process_s16(x)
        else:
            out = self.a.process_cpx(x)                                              # the IQ modes' AGC, in the packet section (:1052)
        if self.mode not in SND_IQ_MODES:
            fm = self.mode in SND_FM_MODES
            if fm and self.de_fm:
                out = self.de_nfm.process_mm(out)
            elif not fm and self.de:
                out = self.de_am.process_mm(out)
        return np.float32(self.avg), np.float32(taps[0]), np.float32(taps[1]), self.squelched, out

    def payload(self, out):
        ko, np = self.ko, self.np
        if self.mode in SND_IQ_MODES:
            return ko.snd_iq_payload(out, self.le)
        s16 = np.asarray(out).astype(np.int16)
        if self.comp:
            enc, self.ad = ko.adpcm_encode_i16(s16, self.ad)
            return enc
        return s16.astype("<i2" if self.le else ">i2").view(np.uint8)

    def header(self, flags, seq, smeter_dbm):
        return self.ko.snd_header(flags, seq, smeter_dbm)

    def gps(self, dticks, fir_pos, agc_on, cfg, clk_ticks, clk_secs):
        ko = self.ko
        if not hasattr(self, "gst"):
            self.gst = ko.GpsState()
        norm, delay2, decim, adc = cfg
        ko.gps_begin(self.gst, clk_secs, float(dticks), adc, GPS_DELAY, delay2)
        return ko.gps_stamp(self.gst, norm, fir_pos, agc_on, self.a.delay(), decim, adc, clk_secs, clk_ticks)


class GpuSoundPath:
    """The same script through kg_post, kg_adpcm / kg_snd_payload / kg_snd_iq_payload and kg_snd_header (one channel)."""

    def __init__(self, P, rate):
        import numpy as np
        from flydog_sdr_gps_amd import post, wire
        self.P, self.rate, self.np, self.post, self.wire_mod = P, rate, np, post, wire
        self.r12k = abs(rate - 12000.0) < abs(rate - 20250.0)
        P.set_smeter(0, rate); P.set_mode(0, post.MODE_SSB); P.reset(0)
        P.squelch_setup(0, rate); P.squelch_set(0, 0, 0)
        self.mode, self.comp, self.le = 2, 1, False
        self.ad = wire.Adpcm(P.ctx, nchan=1)

    def close(self):
        self.ad.close()

    def agc(self, *prm):
        self.P.set_agc(0, *prm, self.rate)

    def passband(self, lo, hi, hbw, stop):
        self.P.set_am_passband(0, lo, hi, self.rate)

    def squelch(self, v, mx):
        self.P.squelch_set(0, v, mx)

    def de_emp(self, de, de_fm):
        self.P.set_de_emp(0, de, 0, snd_rate_12k=self.r12k, frate=self.rate)
        self.P.set_de_emp(0, de_fm, 1, snd_rate_12k=self.r12k, frate=self.rate)

    def set_mode(self, m):
        post = self.post
        self.mode = m
        self.P.set_mode(0, post.MODE_AM if m in SND_AM_MODES else post.MODE_NBFM if m in SND_FM_MODES else post.MODE_SSB if m in SND_SSB_MODES
                        else post.MODE_IQ)

    def wire(self, comp, le):
        self.comp, self.le = comp, bool(le)

    def block(self, x):
        s16, _, agc = self.P.process([0], x[None, :])
        avg, taps = self.P.smeter([0])
        _, sq, _ = self.P.squelch_state([0])
        out = agc[0] if self.mode in SND_IQ_MODES else s16[0]
        return self.np.float32(avg[0]), self.np.float32(taps[0, 0]), self.np.float32(taps[0, 1]), bool(sq[0]), out

    def payload(self, out):
        np, wire, ctx = self.np, self.wire_mod, self.P.ctx
        if self.mode in SND_IQ_MODES:
            return np.asarray(wire.snd_iq_payload(ctx, np.asarray(out, np.complex64)[None, :], self.le)).reshape(-1)
        s16 = np.asarray(out, np.int16)
        if self.comp:
            return np.asarray(self.ad.encode([0], s16[None, :])).reshape(-1)
        return np.asarray(wire.snd_payload(ctx, s16[None, :], self.le)).reshape(-1)

    def header(self, flags, seq, smeter_dbm):
        return self.np.asarray(self.wire_mod.snd_header(self.P.ctx, flags, seq, smeter_dbm))

    def gps(self, dticks, fir_pos, agc_on, cfg, clk_ticks, clk_secs):
        wire, lib = self.wire_mod, self.P.ctx.lib
        if not hasattr(self, "gst"):
            self.gst = wire.GpsState()
        norm, delay2, decim, adc = cfg
        wire.gps_begin(lib, self.gst, clk_secs, float(dticks), adc, GPS_DELAY, delay2)
        return wire.gps_stamp(lib, self.gst, norm, fir_pos, agc_on, self.P.agc_delay(0), decim, adc, clk_secs, clk_ticks)


def sndpath_check(g, name, make_chain):
    """Runs scenario `name` of sndpath_ref.npz through make_chain(rate) and asserts EQUALITY with what the reference's statements
    produced: per CFastFIR block sMeterAvg_dB, the two S-meter taps (the reference adds S_meter_cal on the way to the hook),
    s->squelched, out_samps_s2 resp. the IQ modes' AGC output; per packet the payload bytes and the header (flags, sequence number,
    S-meter field).  The flag rules of rx_sound.cpp:1228-1233 and the packet's block grouping are the host's and restated here.
    -> (packets, blocks, samples compared)"""
    import numpy as np
    rate, lo, hi = (float(v) for v in g[name + "_band"])
    x, y = g[name + "_in"], g[name + "_out"]
    chain = make_chain(rate)
    import struct
    pos = npkt = nblk = samples = seq = 0
    mode, comp, le, ovfl, sq, agc_on = 2, 1, False, 0, False, 1
    cal = np.float32(S_METER_CAL)
    # what the reference's firmware-mode switch (rx_sound.cpp:306-319) gave: inputs of the library's stamp functions
    gps_cfg, clk_ticks, clk_secs, tq, stamp = None, 0, 0.0, [], None
    norm, delay2 = int(y[0]), float(np.float64(y[1]) + np.float64(y[2]) + np.float64(y[3]))
    ypos = 4
    for line in (str(l) for l in g[name + "_script"]):
        f = line.split()
        if f[0] == "R":
            assert float(f[1]) == rate
            gps_cfg = (norm, delay2, int(f[4]), float(f[5]))
        elif f[0] == "C":
            clk_ticks, clk_secs = int(f[1]), float(f[2])
        elif f[0] == "T":
            tq.append((int(f[1]), int(f[2])))
        elif f[0] == "A":
            chain.agc(*[int(v) for v in f[1:7]])
            agc_on = int(f[1])
        elif f[0] == "L":
            chain.passband(lo, hi, float(f[1]), float(f[2]))
        elif f[0] == "Q":
            chain.squelch(int(f[1]), int(f[2]))
        elif f[0] == "E":
            chain.de_emp(int(f[1]), int(f[2]))
        elif f[0] == "M":
            mode = int(f[1])
            chain.set_mode(mode)
        elif f[0] == "W":
            comp, le = int(f[1]), bool(int(f[2]))
            chain.wire(comp, le)
        elif f[0] == "V":
            ovfl = int(f[1])
        else:
            iq = mode in SND_IQ_MODES
            payload, dbm = [], None
            for n in (int(v) for v in f[1:]):
                ticks, fir_pos = tq.pop(0) if tq else (0, 0)
                stamp = chain.gps(time_diff48(ticks, clk_ticks), fir_pos, agc_on, gps_cfg, clk_ticks, clk_secs)     # before the AGC runs, as :636-661
                avg, t0, t1, sq, out = chain.block(x[pos:pos + n])
                pos += n
                w_avg, w_dbm, w_t0, w_t1, w_sq = y[ypos:ypos + 5]
                ypos += 5
                where = (name, npkt, nblk, mode, n)
                assert np.float32(avg).view(np.uint32) == np.float32(w_avg).view(np.uint32), (where, "sMeterAvg_dB", avg, w_avg)
                assert np.float32(w_avg) + cal == w_dbm                                           # rx_sound.cpp:696
                assert np.float32(t0) + cal == w_t0, (where, "S-meter tap j = 0")
                if n >= 2:
                    assert np.float32(t1) + cal == w_t1, (where, "S-meter tap j = n / 2")
                if mode in SND_FM_MODES:
                    assert sq == bool(w_sq), (where, "s->squelched", sq, w_sq)
                if iq:
                    want = y[ypos:ypos + 2 * n].view(np.complex64)
                    ypos += 2 * n
                    assert np.array_equal(np.ascontiguousarray(out, np.complex64).view(np.uint64), np.ascontiguousarray(want).view(np.uint64)), (where, "IQ AGC")
                else:
                    want = y[ypos:ypos + n].astype(np.int32)
                    ypos += n
                    assert np.array_equal(np.asarray(out).astype(np.int32), want), (where, "out_samps_s2", int(np.abs(np.asarray(out).astype(np.int32) - want).max()))
                payload.append(np.asarray(chain.payload(out), np.uint8))
                dbm = np.float32(w_dbm)
                samples += n
                nblk += 1
            hsize, bc = int(y[ypos]), int(y[ypos + 1])
            ypos += 2
            pkt = y[ypos:ypos + hsize + bc].astype(np.uint8)
            ypos += hsize + bc
            payload = np.concatenate(payload)
            assert payload.size == bc and np.array_equal(payload, pkt[hsize:]), (name, npkt, mode, "payload", payload.size, bc)
            seq += 1
            flags = (SND_FLAG_ADC_OVFL if ovfl else 0) | (SND_FLAG_MODE_IQ if iq else 0) | (SND_FLAG_COMPRESSED if comp and not iq else 0) \
                | (SND_FLAG_SQUELCH_UI if sq else 0) | (SND_FLAG_LITTLE_ENDIAN if le else 0)
            hdr = np.asarray(chain.header(flags, seq, float(dbm)), np.uint8)
            assert hsize == (20 if iq else 10) and np.array_equal(hdr[:10], pkt[:10]), (name, npkt, mode, "header", hdr[:10], pkt[:10])
            if iq:                                                                                # snd_pkt_iq_t's GPS stamp (rx_sound.h:61-64): the last block's
                want_stamp = struct.pack("<BBII", stamp[2], 0, stamp[0], stamp[1])
                assert bytes(pkt[10:20]) == want_stamp, (name, npkt, "GPS stamp", stamp, struct.unpack("<BBII", bytes(pkt[10:20])))
            npkt += 1
    assert pos == x.size and ypos == y.size
    if hasattr(chain, "close"):
        chain.close()
    return npkt, nblk, samples


def fastfir_taps_cases(g):
    """tests/golden/fastfir_taps_fftref.npz -> [(name, script lines, x, [per D line: (n, count, FirPos, [(flag, bins complex64[1024])...],
    out complex64[count])])]: what the reference's CFastFIR::ProcessData handed a registered extension hook and returned."""
    import numpy as np
    cases = []
    for name in (str(n) for n in g["names"]):
        script, x, y = [str(l) for l in g[name + "_script"]], g[name + "_in"], g[name + "_out"]
        pos, hooked, per = 0, False, []
        for line in script:
            f = line.split()
            if f[0] == "H":
                hooked = int(f[1]) != 0
            elif f[0] == "D":
                count, firpos = int(y[pos]), int(y[pos + 1])
                pos += 2
                taps = []
                if hooked:
                    nt = int(y[pos]); pos += 1
                    for _ in range(nt):
                        flag = int(y[pos]); pos += 1
                        taps.append((flag, y[pos:pos + 2048].view(np.complex64).copy())); pos += 2048
                out = y[pos:pos + 2 * count].view(np.complex64).copy(); pos += 2 * count
                per.append((int(f[1]), count, firpos, taps, out))
        assert pos == y.size
        cases.append((name, script, x, per))
    return cases
