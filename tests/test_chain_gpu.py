"""The audio path end to end on the GPU, 4 receiver channels in one batch per stage:
ADC stream -> audio DDC (rx1/rx2/CICF, rx_iq_t records) -> data-pump unpack -> CFastFIR ->
S-meter + CAgc (mono16) -> IMA ADPCM -> sound packet.  Every stage is compared with the
oracle fed exactly what the GPU fed that stage, at the stage's own bar (integer stages
bit-exact, CFastFIR 1e-5, mono16 1 LSB), and the decoded packet audio is the AM-modulated
tone that went in."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import Adpcm, FastFir, Post, RxDdc, post, snd, wire
from flydog_sdr_gps_amd.ddc import RX_DECIM, rx_phase_inc

pytestmark = pytest.mark.gpu


def test_audio_chain_adc_to_packet(gpu_ctx, oracle):
    fs = 125e6 / RX_DECIM                                   # 12 kHz
    nch, nrec = 4, 170 * 15                                  # 15 data-pump buffers of 170 samples (config.h:40)
    n = RX_DECIM * nrec
    t = np.arange(n)
    carriers = [0.0371, 0.0913, 0.1502, 0.2207]              # x 125 MHz
    adc = np.zeros(n)
    for k, fc in enumerate(carriers):                        # each receiver's signal: a tone 1 kHz (+k*200 Hz) above its dial
        am = 1 + 0.5 * np.cos(2 * np.pi * 3.0 / 125e6 * RX_DECIM * t / 40.0)
        adc += 4000 * am * np.cos(2 * np.pi * (fc + (1000 + 200 * k) / 125e6) * t)
    adc = np.rint(adc).astype(np.int16)

    d = RxDdc(gpu_ctx, nchan=nch, max_samples=n)
    fir = FastFir(gpu_ctx, nchan=nch, max_in=nrec)
    P = Post(gpu_ctx, nchan=nch)
    A = Adpcm(gpu_ctx, nchan=nch)
    try:
        chans = list(range(nch))
        for ch, fc in enumerate(carriers):
            d.set_freq(ch, rx_phase_inc(fc * 125e6))
            fir.setup(ch, 300.0, 2700.0, 0.0, fs)
            P.set_agc(ch, True, False, -100, 50, 6, 1000, fs)
            P.set_smeter(ch, fs); P.set_mode(ch, post.MODE_SSB); P.reset(ch)

        # stage 1: DDC, bit-exact
        raws = d.push(adc, chans)
        for ch, fc in enumerate(carriers):
            want = oracle.ddc_rx(adc, rx_phase_inc(fc * 125e6))[0]
            assert np.array_equal(raws[ch], want), ch
        assert all(r.size == 6 * nrec for r in raws)

        # stage 2: unpack, bit-exact (one record stream per channel here)
        xs = [snd.unpack(gpu_ctx, raws[ch], nrec, 1)[0] for ch in range(nch)]
        for ch in range(nch):
            assert np.array_equal(xs[ch].view(np.uint32), oracle.dpump_unpack(raws[ch], nrec, 1)[0].view(np.uint32))

        # stage 3: CFastFIR in 170-sample calls, as c2s_sound() makes them
        ys = [[] for _ in range(nch)]
        states = [oracle.fir_new_state() for _ in range(nch)]
        for k in range(nrec // 170):
            for ch in range(nch):
                blk = xs[ch][170 * k:170 * (k + 1)]
                got = fir.process(ch, blk)
                want, _ = oracle.fir_process(states[ch], fir.get_coef(ch), blk, prec=0)
                assert got.size == want.size
                if got.size:
                    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
                    ys[ch].append(got)
        nblk = len(ys[0])
        assert nblk == 4 and all(len(y) == nblk and all(b.size == 512 for b in y) for y in ys)

        # stages 4-5: S-meter + AGC (mono16, all channels per launch), then ADPCM (bit-exact on the GPU's mono16)
        agcs = [oracle.Agc() for _ in range(nch)]
        for a in agcs:
            a.set_parameters(True, False, -100, 50, 6, 1000, fs)
        ad_states = [None] * nch
        payload = [[] for _ in range(nch)]
        for b in range(nblk):
            x = np.stack([ys[ch][b] for ch in range(nch)])
            s16, _, _ = P.process(chans, x)
            enc = A.encode(chans, s16)
            for ch in range(nch):
                want = agcs[ch].process_s16(x[ch])
                dlt = np.abs(s16[ch].astype(int) - want.astype(int))
                assert dlt.max() <= 1 and (dlt == 0).mean() > 0.99
                want_enc, ad_states[ch] = oracle.adpcm_encode_i16(s16[ch], ad_states[ch])
                assert np.array_equal(enc[ch], want_enc)
                payload[ch].append(enc[ch])
        avg, _ = P.smeter(chans)

        # stage 6: the packet a compressed connection gets (4 blocks = LOOP_BC 1024 bytes), decoded as the client does
        for ch in range(nch):
            hdr = wire.snd_header(gpu_ctx, wire.SND_FLAG_COMPRESSED, 1, float(avg[ch]) - 13)
            pkt = np.concatenate([hdr] + payload[ch])
            assert pkt.size == 10 + 1024 and bytes(pkt[:3]) == b"SND"
            audio, _ = oracle.adpcm_decode_i16(pkt[10:])
            tail = audio[-1024:].astype(float)
            spec = np.abs(np.fft.rfft(tail * np.hanning(1024)))
            peak_hz = np.argmax(spec) * fs / 1024
            assert abs(peak_hz - (1000 + 200 * ch)) < 2 * fs / 1024, (ch, peak_hz)       # USB: the offset comes out as audio
            assert 3000 < np.abs(tail).max() < 32767
    finally:
        d.close(); fir.close(); P.close(); A.close()


def test_waterfall_chain_adc_to_packet(gpu_ctx, oracle):
    """ADC stream -> waterfall DDC (zoom 4: R = 16) -> 8192-sample frame -> u8 row -> autoscale
    averages -> wf_pkt_t, each stage against the oracle fed the GPU's own upstream output; the
    carrier that went in sits at its pixel in the decoded packet."""
    from flydog_sdr_gps_amd import Aperture, Ddc, Waterfall, WfParams, handoff, wf
    from tests.test_wf_gpu import check_row, db_bound, oracle_frame
    zoom, start_hz, ui_srate = 4, 6.0e6, 30.0e6
    hz_per_start = ui_srate / (1024 << 14)                         # rx_waterfall.cpp:262: `start` counts these
    p = WfParams.for_zoom(zoom, start_hz / hz_per_start, adc_clock=66.6666e6, ui_srate=ui_srate)
    assert p.decim == 8
    n = 8192 * p.decim + 4096
    t = np.arange(n)
    f_sig = p.start * hz_per_start + 0.3 * ui_srate / (1 << zoom)  # 30 % into the displayed span
    rng = np.random.default_rng(4)
    adc = np.rint(6000 * np.cos(2 * np.pi * f_sig / 66.6666e6 * t) + rng.normal(0, 20, n)).astype(np.int16)

    d = Ddc(gpu_ctx, nchan=1, max_samples=n)
    W = Waterfall(gpu_ctx, nchan=1)
    A = Aperture(gpu_ctx, nchan=1)
    tables = (wf.window_functions(), wf.cic_comp_table())
    try:
        d.set_wf(0, p.i_offset, p.decim)
        iq = d.push(adc, [0])[0][:8192]                              # stage 1: bit-exact
        want_iq = oracle.ddc_wf(adc, p.i_offset, int(np.log2(p.decim)))[0][:8192]
        assert np.array_equal(iq, want_iq)

        W.set_tables(*tables)                                         # stage 2: the frame
        W.set_channel(0, p, interp=wf.WF_MAX, window_func=wf.WINF_HANNING, cic_comp=True)
        row = W.frames([0], iq[None])[0]
        w_out, _, w_pwr_out, w_dB = oracle_frame(oracle, tables, iq, p, wf.WF_MAX, wf.WINF_HANNING, True, False, False)
        check_row(row, w_out, w_dB, db_bound(w_pwr_out))

        A.update([0], row[None], [(handoff.MMA, 8.0, True, False)])   # stage 3: autoscale, exact
        assert np.array_equal(A.get(0), oracle.aper_update(np.zeros(1024, np.float32), row, handoff.MMA, 8.0, True))
        sig, noise = A.report([0])
        assert (int(sig[0]), int(noise[0])) == oracle.aper_report(A.get(0))

        from flydog_sdr_gps_amd import wire                           # stage 4: the packet, bit-exact
        pk = wire.wf_packets(gpu_ctx, row[None], [(int(p.start), zoom, 9, True)])[0]
        assert np.array_equal(pk, oracle.wf_packet(row, int(p.start), zoom, 9, True))
        dec, _ = oracle.adpcm_decode_u8(pk[16:])
        peak = int(np.argmax(dec[10:].astype(int)))
        assert abs(peak - 0.3 * 1024) <= 2, peak
        assert sig[0] > noise[0] + 30
    finally:
        d.close(); W.close(); A.close()
