"""The stage-by-stage comparison of a receiver bank (flydog_sdr_gps_amd.rxbank.RxBank = kg_rxbank) with the oracle.

Test infrastructure: imported by tests/test_receivers_gpu.py (every receiver) and by bench.py AFTER its timed region (a
sample of the receivers).  The oracle is fed the same ADC stream with its state carried from step to step:
  waterfall   one-shot receivers: CICs reset at the block's first sample, NCO running on (rx_waterfall.cpp:1005-1041);
              overlapped receivers: the continuous sampler, frame = the newest 8192 outputs (:967-991); the frame bit for
              bit where the bank says it read it, the u8 row (tests/test_wf_gpu.py's rule), the wf_pkt_t byte for byte
  audio       rx_iq_t records bit for bit, the unpacked samples bit for bit, CFastFIR (1e-5 of max: a transform), then on the GPU's
              own CFastFIR output everything behind it BIT FOR BIT: CAgc, detector, m_AM_FIR / squelch / de-emphasis -> mono16, the
              IQ mode's AGC output, the ADPCM and IQ payloads (log10f / powf are the host libm's on the device, csrc/kg_libm.h)
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class AudioTail:
    """What follows the AGC of one receiver up to out_samps_s2 (rx/rx_sound.cpp:762-907), oracle side, by the mode
    RxBank.set_audio configured: (mode, lo, hi, fs, de_emp, squelch).  block() -> (mono16 block or the IQ mode's complex block,
    is it the IQ mode)."""

    def __init__(self, ko, audio):
        from flydog_sdr_gps_amd import deemp, post
        self.ko, self.post = ko, post
        self.mode, lo, hi, fs, self.de_emp, squelch = audio if audio is not None else (post.MODE_SSB, 300.0, 2700.0, 12000.0, 0, 0)
        self.z1, self.last = 0.0, (0.0, 0.0)
        self.am = ko.CFir()
        from tests.fixtures import am_passband
        _, _, hbw, stop = am_passband(lo, hi, fs)                                       # rx_sound_cmd.cpp:248-250, 268-282
        self.am.init_lp(0, 1.0, 50.0, hbw, stop, fs)
        self.sq = ko.Squelch()
        self.sq.setup(fs)
        self.sq.set_squelch(0, 0)
        if squelch:
            self.sq.set_squelch(squelch, 0)
        self.de = ko.CFir()
        if self.de_emp:
            self.de.init_const(deemp.table(self.mode == post.MODE_NBFM, abs(fs - 12000.0) < abs(fs - 20250.0))[self.de_emp - 1], fs)
        self.rcs = []

    def block(self, agc, y):
        post, ko = self.post, self.ko
        if self.mode == post.MODE_IQ:
            return agc.process_cpx(y), True       # the AGC's complex output, its payload byte for byte
        if self.mode == post.MODE_SSB:
            s = agc.process_s16(y)
        elif self.mode == post.MODE_AM:
            d, self.z1 = ko.am_detect(self.z1, agc.process_cpx(y))
            s = self.am.process_rm(d)
        else:
            d, self.last = ko.nbfm_detect(self.last, agc.process_cpx(y))
            s, rc = self.sq.perform_fm(d)
            self.rcs.append(rc)
        if self.de_emp:
            s = self.de.process_mm(s)
        return s, False


def check_bank(bank, adc_of_step, d_adc_of_step, rxs, steps=3, threads=8, events=None):
    """Steps a FRESHLY CONFIGURED RxBank `steps` times (step k over adc_of_step(k), a host int16 array of bank.n samples whose
    device copy is d_adc_of_step(k)) and checks every stage of the receivers `rxs`.
    events: {step: [("wf", rx, WfParams, overlapped) | ("freq", rx, phase_inc) | ("leave", rx) |
    ("join", rx, WfParams, overlapped, phase_inc, {set_audio keywords}) | ("audio", rx, {set_audio keywords})]} -- the
    connection's `SET zoom= start=` / `SET freq=` / `SET mod=` commands and connections that end and start, applied to the bank
    and to the oracle's state before that step: a new waterfall setting resets the receiver's sampler (CmdWFReset, fill pipe), a
    new audio frequency leaves the filters running (rx_sound_cmd.cpp:41-51), a joining receiver starts from zero -- its audio
    DDC, CFastFIR position, detector state, ADPCM state and sequence number; m_Agc[] persists (rx/rx_sound.cpp:152, 236-269) --
    while every other receiver's state runs on: from then on its sound blocks complete on steps of its own.
    -> {"receivers", "steps", "frames", "audio_blocks", "overlapped_frames", "ring_moves"}"""
    from flydog_sdr_gps_amd import wf
    from oracle import kiwi_oracle as ko
    from tests.test_wf_gpu import check_row, db_bound, oracle_frame
    ko.lib()
    rxs = list(rxs)
    n = bank.n
    tables = (wf.window_functions(), wf.cic_comp_table())
    wf_st = {rx: None for rx in rxs}                       # overlapped: the continuous sampler's state
    hist = {rx: np.zeros((0, 2), np.int16) for rx in rxs}  # overlapped: the newest outputs (<= 8192)
    rx_st = {rx: None for rx in rxs}
    fir_st = {rx: ko.fir_new_state() for rx in rxs}
    agcs = {rx: ko.Agc() for rx in rxs}
    for a in agcs.values():
        a.set_parameters(True, False, -100, 50, 6, 1000, bank.fs)
    tails = {rx: AudioTail(ko, bank.audio[rx]) for rx in rxs}
    ad_st = {rx: None for rx in rxs}
    coef = {rx: bank.fir.get_coef(rx) for rx in rxs}
    frames = audio_blocks = ov_frames = moves = 0
    snd_seq = {rx: 0 for rx in rxs}
    active = {rx: bank.is_active(rx) for rx in rxs}
    blocks_of = {rx: [] for rx in rxs}                     # the steps on which a receiver's sound blocks completed
    nco0 = {rx: 0 for rx in rxs}                           # one-shot receivers: ADC samples since the waterfall NCO was last set
    total_n = 0
    with ThreadPoolExecutor(threads) as pool:              # the oracle's C calls release the GIL
        for step in range(steps):
            for ev in (events or {}).get(step, ()):
                if ev[0] == "wf":
                    _, rx, p, ov = ev
                    bank.set_wf(rx, p, ov)                 # kg_rxbank_set_wf: phase = 0, sampler reset
                    if rx in wf_st:
                        wf_st[rx], hist[rx], nco0[rx] = None, np.zeros((0, 2), np.int16), total_n
                elif ev[0] == "freq":
                    _, rx, inc = ev
                    bank.rxddc.set_freq(rx, inc)
                    bank.rx_inc[rx] = int(inc)
                elif ev[0] == "leave":
                    bank.leave(ev[1])
                    if ev[1] in active:
                        active[ev[1]] = False
                elif ev[0] == "join":
                    _, rx, p, ov, inc, kw = ev
                    bank.join(rx, (p, ov), inc, **kw)
                    if rx in active:
                        active[rx] = True
                        wf_st[rx], hist[rx], nco0[rx] = None, np.zeros((0, 2), np.int16), total_n
                        rx_st[rx], fir_st[rx], ad_st[rx], snd_seq[rx] = None, ko.fir_new_state(), None, 0
                        tails[rx] = AudioTail(ko, bank.audio[rx])
                        coef[rx] = bank.fir.get_coef(rx)
                elif ev[0] == "audio":
                    _, rx, kw = ev
                    bank.set_audio(rx, bank.rx_inc[rx], **kw)          # `SET mod= low_cut= high_cut=`: new filters, the AGC runs on
                    if rx in active:
                        tails[rx] = AudioTail(ko, bank.audio[rx])
                        coef[rx] = bank.fir.get_coef(rx)
            adc = adc_of_step(step)
            info = bank.step(d_adc_of_step(step))
            bank.sync()
            assert info.step == step, (step, info.step)
            moves += info.nmoves
            rx_of, f_off, pkt_bytes = bank.frame_map()
            frame_of = {int(r): f for f, r in enumerate(rx_of)}
            m_nrec, m_nfir, m_pos, m_seq = bank.audio_map()
            live = [rx for rx in rxs if active[rx]]
            for rx in rxs:
                if not active[rx]:
                    assert m_nrec[rx] == 0 and m_nfir[rx] == 0 and rx not in frame_of, (step, rx, "an inactive receiver was stepped")

            def wf_ref(rx):
                p = bank.params[rx]
                l2 = int(np.log2(p.decim))
                if bank.overlapped[rx]:
                    return ko.ddc_wf(adc, p.i_offset, l2, wf_st[rx])
                st = ko.DdcWfState()                                   # CmdWFReset: CICs zero, the NCO running on
                st.phase = ((total_n - nco0[rx]) * p.i_offset) & ((1 << 48) - 1)
                return ko.ddc_wf(adc[:8192 * p.decim], p.i_offset, l2, st)

            wf_out = list(pool.map(wf_ref, live))
            rx_out = list(pool.map(lambda rx: ko.ddc_rx(adc, bank.rx_inc[rx], rx_st[rx], bank.rx_mode), live))
            with_frame = [rx for rx in live if rx in frame_of]
            g_rows = dict(zip(with_frame, bank.fetch("rows", [frame_of[rx] for rx in with_frame])))
            g_pkts = dict(zip(with_frame, bank.fetch("pkts", [frame_of[rx] for rx in with_frame])))
            g = {k: bank.fetch(k, live) for k in ("wf_iq", "raw", "xin", "firo", "s16", "pay", "agc", "iq_pay")} if live else {}
            stride = bank.bufs.wf_iq_stride
            for i, rx in enumerate(live):
                p = bank.params[rx]
                nrec, nfir = int(m_nrec[rx]), int(m_nfir[rx])
                assert int(m_seq[rx]) == snd_seq[rx] + nfir // 512, (step, rx, m_seq[rx], snd_seq[rx], nfir)
                iq, st_new = wf_out[i]
                if bank.overlapped[rx]:
                    wf_st[rx] = st_new
                    assert iq.shape[0] == n // p.decim, (step, rx, iq.shape)
                    hist[rx] = np.concatenate([hist[rx], iq])[-8192:]
                    want_frame = hist[rx] if hist[rx].shape[0] == 8192 else None     # "fill pipe": no frame yet
                else:
                    assert iq.shape[0] == 8192, (step, rx, iq.shape)
                    want_frame = iq
                assert (want_frame is not None) == (rx in frame_of), (step, rx, "frame taken / not taken")
                if want_frame is not None:
                    f = frame_of[rx]
                    off = int(f_off[f]) - rx * stride
                    assert 0 <= off <= stride - 8192, (step, rx, off)
                    assert np.array_equal(g["wf_iq"][i, off:off + 8192], want_frame), (step, rx, "frame samples")
                    w_out, _, w_pwr_out, w_dB = oracle_frame(ko, tables, want_frame, p, wf.WF_MAX, wf.WINF_HANNING, True,
                                                             bank.overlapped[rx], False)
                    check_row(g_rows[rx], w_out, w_dB, db_bound(w_pwr_out))
                    want_pkt = ko.wf_packet(g_rows[rx], int(p.start), p.zoom, snd_seq[rx], True)
                    assert int(pkt_bytes[f]) == want_pkt.size, (step, rx, pkt_bytes[f], want_pkt.size)
                    assert np.array_equal(g_pkts[rx][:want_pkt.size], want_pkt), (step, rx, "wf_pkt_t")
                    frames += 1
                    ov_frames += int(bank.overlapped[rx])
                # audio DDC -> rx_iq_t records -> unpack
                raw, rx_st[rx] = rx_out[i]
                assert raw.size == 6 * nrec, (step, rx, raw.size, nrec)
                assert np.array_equal(g["raw"][i, :raw.size], raw), (step, rx, "rx_iq_t")
                got_x = np.zeros(0, np.complex64)
                if nrec:
                    x = ko.dpump_unpack(raw, nrec, 1)[0]
                    got_x = np.ascontiguousarray(g["xin"][i, :nrec]).view(np.complex64).ravel()
                    assert np.array_equal(got_x.view(np.uint32), x.view(np.uint32)), (step, rx, "unpack")
                # CFastFIR on the GPU's own input, then CAgc and ADPCM on the GPU's own upstream output
                want_y, _ = ko.fir_process(fir_st[rx], coef[rx], got_x, prec=0)
                assert want_y.size == nfir, (step, rx, want_y.size, nfir)
                for blk in range(nfir // 512):
                    sl = slice(512 * blk, 512 * (blk + 1))
                    got_y = np.ascontiguousarray(g["firo"][i, sl]).view(np.complex64).ravel()
                    assert np.abs(got_y - want_y[sl]).max() <= 1e-5 * np.abs(want_y[sl]).max(), (step, rx, "CFastFIR")
                    want_s, is_iq = tails[rx].block(agcs[rx], got_y)
                    if is_iq:                                             # MODE_IQ (rx_sound.cpp:1040-1096)
                        got_a = np.ascontiguousarray(g["agc"][i, sl]).view(np.complex64).ravel()
                        assert np.array_equal(got_a.view(np.uint64), np.ascontiguousarray(want_s, np.complex64).view(np.uint64)), (step, rx, "IQ-mode AGC")
                        want_p = ko.snd_iq_payload(want_s, bank.little_endian[rx])
                        assert np.array_equal(g["iq_pay"][i, 2048 * blk:2048 * (blk + 1)], want_p), (step, rx, "IQ payload")
                        audio_blocks += 1
                        blocks_of[rx].append(step)
                        continue
                    assert np.array_equal(g["s16"][i, sl].astype(int), np.asarray(want_s).astype(int)), (
                        step, rx, "mono16", int(np.abs(g["s16"][i, sl].astype(int) - np.asarray(want_s).astype(int)).max()))
                    want_enc, ad_st[rx] = ko.adpcm_encode_i16(np.asarray(want_s, np.int16), ad_st[rx])
                    assert np.array_equal(g["pay"][i, 256 * blk:256 * (blk + 1)], want_enc), (step, rx, "ADPCM")
                    audio_blocks += 1
                    blocks_of[rx].append(step)
                assert int(m_pos[rx]) == fir_st[rx].in_pos - 512, (step, rx, "FirPos", m_pos[rx], fir_st[rx].in_pos)
                snd_seq[rx] += nfir // 512
            total_n += n
    check_bank.blocks_of = blocks_of                        # (for the caller that asks when each receiver's blocks completed)
    return {"receivers": len(rxs), "steps": steps, "frames": frames, "audio_blocks": audio_blocks,
            "overlapped_frames": ov_frames, "ring_moves": moves}
