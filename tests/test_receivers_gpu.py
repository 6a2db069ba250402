"""BASELINE configs[3] through the C ABI's receiver bank (kg_rxbank, include/kiwigpu.h): the object `bench.py --workload
receivers` times, stepped over its 2^22-sample ADC block on SURVEY.md 8(d)'s receiver mix (f_k = 100 kHz + k 29 kHz, zoom
8 + (k mod 4): zooms 8..10 take the reference's non-overlapped frame, zoom 11 its overlapped / continuous sampler) and
EVERY stage of EVERY receiver checked against the oracle (tests/rxbank_check.py); the same on rounds 2-4's lighter mix;
and a small-step bank stepped often enough for the overlapped receivers' sample rings to wrap."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEPS = 3
N = 1 << 22


def _bank(nrx, n, mix):
    from flydog_sdr_gps_amd.rxbank import RxBank
    bank = RxBank(nrx, n)
    bank.configure(mix)
    return bank


@pytest.mark.parametrize("mix_name,NR,FIRST", [("survey", 10, 123),     # straddles the rank 0 / rank 1 boundary of the 8 x 128 sharding
                                               ("survey", 128, 128),    # rank 1's whole slice: the per-GPU shape of configs[3]
                                               ("light", 10, 123)])     # every zoom 1..10 (R = 1 bypass, staged small R, large R)
def test_receiver_bank_every_stage_every_receiver(oracle, mix_name, NR, FIRST):
    from flydog_sdr_gps_amd import synth
    from flydog_sdr_gps_amd.rxbank import MIXES
    from tests.rxbank_check import check_bank
    mix = MIXES[mix_name](NR, FIRST, N)
    zooms = sorted(set(p.zoom for p, _, _ in mix))
    assert zooms == ([8, 9, 10, 11] if mix_name == "survey" else list(range(1, 11)))
    n_ov = sum(1 for _, ov, _ in mix if ov)
    assert n_ov == (sum(1 for p, _, _ in mix if p.zoom == 11) if mix_name == "survey" else 0)
    adc = synth.adc_stream(N, 0x5EED0004)
    bank = _bank(NR, N, mix)
    try:
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc, lambda k: d_adc, range(NR), STEPS)
        # 402 records a step: a 512-sample block on steps 2 and 3; an overlapped receiver (4096 outputs a step) takes its first
        # frame on step 2
        assert got == {"receivers": NR, "steps": STEPS, "frames": STEPS * NR - n_ov, "audio_blocks": 2 * NR,
                       "overlapped_frames": (STEPS - 1) * n_ov, "ring_moves": 0}, got
        bank.ctx.free(d_adc)
    finally:
        bank.close()


def test_overlapped_rings_wrap_and_streams_differ_per_step(oracle):
    """A small step (2^17 samples) so that the rings wrap inside the test: R = 16 one-shot, R = 32 / 64 / 128 overlapped with
    4096 / 2048 / 1024 outputs per step; 48 steps over 48 DIFFERENT blocks of one stream; audio: 12.6 records per step, one
    512-sample block per receiver near the end."""
    from flydog_sdr_gps_amd import synth
    from flydog_sdr_gps_amd.ddc import rx_phase_inc
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, UI_SRATE
    from flydog_sdr_gps_amd.wf import WfParams
    from tests.rxbank_check import check_bank
    n, steps = 1 << 17, 48
    hz = UI_SRATE / (1024 << 14)
    mix = []
    for k, zoom in enumerate([5, 6, 7, 8, 6, 5, 8, 7]):
        # (every span holds the stream's strong carrier at 0.0123 f_adc = 820 kHz: the rows' tolerance is relative to the
        # largest bin, and a span of decimated noise alone is dominated by the DC bin compute_frame() then blanks)
        span = UI_SRATE / (1 << zoom)
        p = WfParams.for_zoom(zoom, (0.0123 * ADC_CLOCK - span * (0.2 + 0.07 * k)) / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        mix.append((p, 8192 * p.decim > n, rx_phase_inc(0.0123 * ADC_CLOCK - 900.0 - 35.0 * k, ADC_CLOCK)))
    assert [ov for _, ov, _ in mix] == [False, True, True, True, True, False, True, True]
    adc = synth.adc_stream(n * steps, 0x5EED0044)
    bank = _bank(len(mix), n, mix)
    try:
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc[k * n:(k + 1) * n], lambda k: d_adc + 2 * k * n, range(len(mix)), steps)
        def wraps(m, ring=65536):                          # the ring discipline of kg_rxbank.hip, restated
            w = cnt = 0
            for _ in range(steps):
                if w + m > ring:
                    w, cnt = 8192 - m, cnt + 1
                w += m
            return cnt
        assert wraps(4096) >= 2 and wraps(2048) >= 1
        assert got["ring_moves"] == 2 * (wraps(4096) + wraps(2048) + wraps(1024)), got
        assert got["frames"] == 2 * steps + 2 * (steps - 1) + 2 * (steps - 3) + 2 * (steps - 7), got
        assert got["audio_blocks"] == len(mix), got         # 603 records in 48 steps: one block each
        bank.ctx.free(d_adc)
    finally:
        bank.close()


def test_bank_error_paths():
    from flydog_sdr_gps_amd import KiwiGpuError
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, UI_SRATE, RxBank
    from flydog_sdr_gps_amd.wf import WfParams
    hz = UI_SRATE / (1024 << 14)
    bank = RxBank(2, 1 << 17)
    try:
        bank.wf.set_tables()
        p16 = WfParams.for_zoom(5, 1.0e6 / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        p32 = WfParams.for_zoom(6, 1.0e6 / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        with pytest.raises(KiwiGpuError):
            bank.set_wf(0, p32, overlapped=False)            # a one-shot frame at R = 32 takes 2^18 samples
        p8 = WfParams.for_zoom(4, 1.0e6 / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        with pytest.raises(KiwiGpuError):
            bank.set_wf(0, p8, overlapped=True)              # 16384 outputs a step: a frame per step, one-shot is the mode
        with pytest.raises(KiwiGpuError):
            bank.set_wf(2, p16)                              # receiver out of range
        bank.set_wf(0, p16)
        d = bank.ctx.alloc(2 << 17)
        with pytest.raises(KiwiGpuError):
            bank.step(d)                                     # receiver 1 has no waterfall set
        bank.ctx.free(d)
        # a bank-owned object's host-buffer conveniences refuse (its rows go by receiver number), in C as in Python
        with pytest.raises(KiwiGpuError, match="receiver bank"):
            bank.fir.process(0, np.zeros(16, np.complex64))
        # an inactive receiver needs no waterfall setting; a joined one needs a new one
        bank.leave(1)
        assert not bank.is_active(1) and bank.is_active(0) and bank.ready()
        d = bank.ctx.alloc(2 << 17)
        from flydog_sdr_gps_amd.ddc import rx_phase_inc
        bank.set_audio(0, rx_phase_inc(1.0e6, ADC_CLOCK))
        bank.step(d)
        bank.join(0)
        with pytest.raises(KiwiGpuError, match="since it joined"):
            bank.step(d)
        bank.sync()
        bank.ctx.free(d)
    finally:
        bank.close()
    # settings that could only fail later, on every step, fail when they are made (the advisor's round-5 finding): an overlapped
    # sampler that yields an odd number of outputs per step.  (The other one -- the largest step, 2^27 samples of the rx3
    # instance, is 43 sound blocks per receiver and needed more table entries than a step had -- is sized for now: the step table
    # holds 58 blocks, and kg_rxbank_create checks its step against that.)
    bank = RxBank(1, 8192)
    try:
        p14 = WfParams.for_zoom(14, 1.0e6 / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)     # R = 8192: one output per step
        with pytest.raises(KiwiGpuError, match="even divisor"):
            bank.set_wf(0, p14, overlapped=True)
    finally:
        bank.close()


def _small_mix(zooms, n):
    from flydog_sdr_gps_amd.ddc import rx_phase_inc
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, UI_SRATE
    from flydog_sdr_gps_amd.wf import WfParams
    hz = UI_SRATE / (1024 << 14)
    mix = []
    for k, zoom in enumerate(zooms):
        span = UI_SRATE / (1 << zoom)
        p = WfParams.for_zoom(zoom, (0.0123 * ADC_CLOCK - span * (0.2 + 0.05 * (k % 9))) / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        mix.append((p, 8192 * p.decim > n, rx_phase_inc(0.0123 * ADC_CLOCK - 900.0 - 35.0 * k, ADC_CLOCK)))
    return mix


def test_connections_retune_between_steps(oracle):
    """`SET zoom= start=` and `SET freq=` while the bank runs (rx_waterfall.cpp:410-515, rx_sound_cmd.cpp:41-51): a receiver
    goes from the one-shot sampler to the overlapped one and back, another changes its span inside the overlapped mode
    (its ring starts empty again: "fill pipe"), a third moves its audio NCO (filters keep running).  12 steps of 2^17
    samples over one stream, every stage of every receiver against the oracle with the same events applied."""
    from flydog_sdr_gps_amd import synth
    from flydog_sdr_gps_amd.ddc import rx_phase_inc
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK
    from tests.rxbank_check import check_bank
    n, steps = 1 << 17, 12
    mix = _small_mix([5, 6, 7, 5, 8], n)
    alt = _small_mix([6, 5, 8, 4, 7], n)                     # what the receivers are retuned to
    adc = synth.adc_stream(n * steps, 0x5EED0047)
    bank = _bank(len(mix), n, mix)
    try:
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        events = {3: [("wf", 0, alt[0][0], alt[0][1])],                                   # one-shot -> overlapped
                  5: [("wf", 2, alt[2][0], alt[2][1]), ("freq", 4, rx_phase_inc(0.0123 * ADC_CLOCK - 1500.0, ADC_CLOCK))],
                  8: [("wf", 0, mix[0][0], mix[0][1]), ("wf", 3, alt[3][0], alt[3][1])]}  # back to one-shot; another one-shot zoom
        assert alt[0][1] and not mix[0][1] and alt[2][1] and mix[2][1]
        got = check_bank(bank, lambda k: adc[k * n:(k + 1) * n], lambda k: d_adc + 2 * k * n, range(len(mix)), steps, events=events)
        assert got["frames"] > 0 and got["overlapped_frames"] > 0, got
        bank.ctx.free(d_adc)
    finally:
        bank.close()


@pytest.mark.parametrize("mode_name", ["wide", "rx14"])
def test_bank_of_the_other_rx_instances(oracle, mode_name):
    """The rx3 (20.25 kHz, ADC / 6172) and rx14 (17-tap CICF) audio DDCs under the bank (KG_RXDDC_WIDE / KG_RXDDC_RX14), 70
    receivers -- not a multiple of a wave -- over 44 steps of 2^17 samples (one or two sound blocks each); a sample of the
    receivers against the oracle."""
    from flydog_sdr_gps_amd import synth
    from flydog_sdr_gps_amd.ddc import RX_14, RX_WIDE
    from flydog_sdr_gps_amd.rxbank import RxBank
    from tests.rxbank_check import check_bank
    n, steps, nrx = 1 << 17, 44, 70
    mode = RX_WIDE if mode_name == "wide" else RX_14
    mix = _small_mix([5 + k % 4 for k in range(nrx)], n)
    adc = synth.adc_stream(n * steps, 0x5EED0048)
    bank = RxBank(nrx, n, rx_mode=mode)
    try:
        bank.configure(mix)
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc[k * n:(k + 1) * n], lambda k: d_adc + 2 * k * n, [0, 1, 2, 3, 33, 63, 64, 69], steps)
        assert got["frames"] > 0 and got["audio_blocks"] >= 8, got
        bank.ctx.free(d_adc)
    finally:
        bank.close()


def test_overlapped_mode_at_its_edges(oracle):
    """The overlapped sampler at the ends of what kg_rxbank_set_wf accepts: a receiver whose step yields exactly one frame
    (R = 16 with 2^17 samples: 8192 outputs a step -- the ring wraps without anything to keep) and two that fill their pipes
    slowly (R = 2048 / 4096: 64 / 32 outputs a step, no frame inside the test), beside a one-shot receiver.  12 steps."""
    from flydog_sdr_gps_amd import synth
    from flydog_sdr_gps_amd.ddc import rx_phase_inc
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, UI_SRATE
    from flydog_sdr_gps_amd.wf import WfParams
    from tests.rxbank_check import check_bank
    n, steps = 1 << 17, 12
    hz = UI_SRATE / (1024 << 14)
    mix = []
    for k, (zoom, ov) in enumerate([(5, True), (5, False), (12, True), (13, True)]):
        span = UI_SRATE / (1 << zoom)
        p = WfParams.for_zoom(zoom, (0.0123 * ADC_CLOCK - span * 0.3) / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        mix.append((p, ov, rx_phase_inc(0.0123 * ADC_CLOCK - 800.0 - 50.0 * k, ADC_CLOCK)))
    assert [p.decim for p, _, _ in mix] == [16, 16, 2048, 4096]
    adc = synth.adc_stream(n * steps, 0x5EED0049)
    bank = _bank(len(mix), n, mix)
    try:
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc[k * n:(k + 1) * n], lambda k: d_adc + 2 * k * n, range(len(mix)), steps)
        assert got["frames"] == 2 * steps and got["overlapped_frames"] == steps and got["ring_moves"] == 0, got
        bank.ctx.free(d_adc)
    finally:
        bank.close()


def test_long_steps_carry_several_sound_blocks(oracle):
    """A step of 2^24 ADC samples is 1610 rx_iq_t records per receiver: three (then four) 512-sample CFastFIR blocks -- and as
    many S-meter / CAgc / ADPCM calls -- inside ONE kg_rxbank_step; R = 4096 is the overlapped sampler there (4096 outputs a
    step), R = 256 ... 2048 one-shot.  Two steps over two different blocks."""
    from flydog_sdr_gps_amd import synth
    from tests.rxbank_check import check_bank
    n, steps = 1 << 24, 2
    mix = _small_mix([9, 11, 12, 13], n)
    assert [ov for _, ov, _ in mix] == [False, False, False, True]
    adc = synth.adc_stream(n * steps, 0x5EED004A)
    bank = _bank(len(mix), n, mix)
    try:
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc[k * n:(k + 1) * n], lambda k: d_adc + 2 * k * n, range(len(mix)), steps)
        assert got["audio_blocks"] == len(mix) * (2 * n // 10416 // 512) and got["frames"] == 3 * steps + 1, got
        bank.ctx.free(d_adc)
    finally:
        bank.close()


@pytest.mark.parametrize("depth", [2, 9])
def test_adc_ring_refilled_without_host_synchronisation(oracle, depth):
    """The streaming shape INTEGRATION.md 5a describes: a ring of device buffers refilled in turn on the caller's own stream,
    no host synchronisation between steps -- `adc_ready_event` orders a step behind the writer of its block; with TWO buffers
    kg_rxbank_adc_done orders the writer of a buffer behind the step that last read it, with NINE (KG_RXBANK_SLOTS + 1) the
    writer needs no ordering at all, because kg_rxbank_step returns only when the step eight before it has completed.  Ten steps enqueued back to back; a missing edge in
    either direction would let block k + 2 overwrite block k while step k is still running: 64 receivers and 2^22 samples make
    a step 0.7 ms of GPU work against 0.1 ms of host time, so the host IS steps ahead (with kg_rxbank_adc_done made a no-op
    the comparison below fails).  The audio DDC's records of the LAST step (state carried through all six blocks) and the
    last step's frames of four of the receivers must equal the oracle's over the same stream."""
    import torch
    from flydog_sdr_gps_amd import synth
    n, steps = 1 << 22, 6 if depth == 2 else 20
    mix = _small_mix([8 + k % 4 for k in range(64)], n)       # zooms 8 .. 10: one-shot; zoom 11: overlapped (4096 a step)
    check = [0, 1, 2, 3]
    adc = synth.adc_stream(n * steps, 0x5EED004B)
    host = torch.from_numpy(adc).pin_memory()
    dev = torch.device("cuda", 0)
    bank = _bank(len(mix), n, mix)
    try:
        bufs = [torch.zeros(n, dtype=torch.int16, device=dev) for _ in range(depth)]
        up = torch.cuda.Stream(device=dev)
        evs = [torch.cuda.Event() for _ in range(steps)]
        torch.cuda.synchronize(dev)
        infos = []
        for k in range(steps):
            with torch.cuda.stream(up):
                if depth == 2:
                    bank.adc_done(up.cuda_stream, 2)         # the step that read bufs[k % 2], two steps ago, is behind us
                bufs[k % depth].copy_(host[k * n:(k + 1) * n], non_blocking=True)
                evs[k].record(up)
            infos.append(bank.step(bufs[k % depth].data_ptr(), adc_ready_event=evs[k].cuda_event))
        bank.sync()
        torch.cuda.synchronize(dev)
        # the oracle over the whole stream
        nrec_last = infos[-1].nrec
        raw = bank.fetch("raw", check)
        rx_of, f_off, _ = bank.frame_map()
        wf_iq = bank.fetch("wf_iq", check)
        stride = bank.bufs.wf_iq_stride
        for rx in check:
            p, ov, inc = mix[rx]
            st = None
            for k in range(steps):
                rec, st = oracle.ddc_rx(adc[k * n:(k + 1) * n], inc, st)
            assert rec.size == 6 * nrec_last and np.array_equal(raw[rx, :rec.size], rec), ("audio records of the last step", rx)
            l2 = int(np.log2(p.decim))
            if ov:
                iq, _ = oracle.ddc_wf(adc, p.i_offset, l2)
                want = iq[-8192:]
            else:
                s0 = oracle.DdcWfState()
                s0.phase = ((steps - 1) * n * p.i_offset) & ((1 << 48) - 1)
                want, _ = oracle.ddc_wf(adc[(steps - 1) * n:(steps - 1) * n + 8192 * p.decim], p.i_offset, l2, s0)
            f = list(rx_of).index(rx)
            off = int(f_off[f]) - rx * stride
            assert np.array_equal(wf_iq[rx, off:off + 8192], want), ("frame of the last step", rx)
    finally:
        bank.close()


def test_two_banks_stepped_from_two_threads_equal_their_solo_runs():
    """Two connections' worth of banks on ONE device, each stepped by its own host thread at the same time (the C calls release
    the interpreter lock): every output buffer of every step equals, byte for byte, what the same bank produces alone.  Holds
    the process-wide pieces -- the stream pools, the library's tables, the contexts' staging -- to 'no shared mutable state'."""
    import threading
    from flydog_sdr_gps_amd import synth
    n, steps = 1 << 20, 7
    mixes = [_small_mix([6, 7, 8, 5, 7, 6], n), _small_mix([4, 5, 6, 7, 8, 3, 2, 1, 6, 7], n)]
    adcs = [synth.adc_stream(n * steps, 0x5EED0070 + b) for b in range(2)]
    keys = ("rows", "pkts", "raw", "xin", "firo", "s16", "pay")

    def run(b, out, barrier=None):
        bank = _bank(len(mixes[b]), n, mixes[b])
        try:
            d_adc = bank.ctx.alloc(adcs[b].nbytes)
            bank.ctx.upload(d_adc, adcs[b])
            if barrier is not None:
                barrier.wait()
            for k in range(steps):
                info = bank.step(d_adc + 2 * k * n)
                bank.sync()
                rec = {"info": (info.nframes, info.nrec, info.nfir, info.snd_seq, info.nmoves)}
                nrx = len(mixes[b])
                rx_of, f_off, pkt_bytes = bank.frame_map()
                rec["map"] = (rx_of.tolist(), f_off.tolist(), pkt_bytes.tolist())
                # what a step DEFINES of each buffer (the rest of a row is whatever the allocation held)
                valid = {"rows": 1024, "pkts": None, "raw": 6 * info.nrec, "xin": info.nrec, "firo": info.nfir, "s16": info.nfir,
                         "pay": info.nfir // 2}
                for key in keys:
                    rows = range(info.nframes) if key in ("rows", "pkts") else range(nrx)
                    got = bank.fetch(key, rows)
                    if key == "pkts":
                        rec[key] = [got[f, :int(pkt_bytes[f])].copy() for f in range(info.nframes)]
                    else:
                        rec[key] = got[:, :valid[key]].copy()
                out.append(rec)
            bank.ctx.free(d_adc)
        finally:
            bank.close()

    solo = [[], []]
    for b in range(2):
        run(b, solo[b])
    both = [[], []]
    errs = []
    barrier = threading.Barrier(2)

    def guarded(b):
        try:
            run(b, both[b], barrier)
        except BaseException as e:        # noqa: BLE001 -- re-raised below, in the test's thread
            errs.append(e)
            barrier.abort()
    threads = [threading.Thread(target=guarded, args=(b,)) for b in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    frames = 0
    for b in range(2):
        assert len(both[b]) == steps
        for k in range(steps):
            assert both[b][k]["info"] == solo[b][k]["info"], (b, k)
            frames += both[b][k]["info"][0]
            assert both[b][k]["map"] == solo[b][k]["map"], (b, k)
            for key in keys:
                if key == "pkts":
                    assert all(np.array_equal(x, y) for x, y in zip(both[b][k][key], solo[b][k][key])), (b, k, key)
                else:
                    assert np.array_equal(both[b][k][key], solo[b][k][key]), (b, k, key)
    assert frames > 0 and any(r["info"][2] for r in both[0]) and any(r["info"][2] for r in both[1])


def test_bank_of_mixed_demodulators_reaches_the_sound_payload(oracle):
    """One bank whose receivers listen in different modes -- SSB, AM (wide and narrow), NBFM (squelch open, a threshold the
    channel's noise crosses, forced shut), with and without de-emphasis -- stepped six times: every receiver's out_samps_s2
    (rx/rx_sound.cpp:762-907: AGC -> detector -> m_AM_FIR resp. m_Squelch -> de-emphasis) against the oracle's chain on the
    GPU's own CFastFIR output, and its ADPCM sound payload byte for byte on the GPU's own mono16 block."""
    from flydog_sdr_gps_amd import post, synth
    from flydog_sdr_gps_amd.rxbank import MIXES
    from tests.rxbank_check import check_bank
    modes = [   # mode, low_cut, high_cut, de_emp, squelch
        (post.MODE_SSB, 300.0, 2700.0, 0, 0),
        (post.MODE_AM, -2500.0, 2500.0, 0, 0),
        (post.MODE_NBFM, -3000.0, 3000.0, 0, 0),
        (post.MODE_AM, -2500.0, 2500.0, 1, 0),
        (post.MODE_NBFM, -3000.0, 3000.0, 2, 80),
        (post.MODE_SSB, -2700.0, -300.0, 2, 0),
        (post.MODE_AM, -1200.0, 1250.0, 2, 0),
        (post.MODE_NBFM, -3000.0, 3000.0, 1, 99),
        (post.MODE_SSB, 300.0, 2700.0, 0, 0),
        (post.MODE_NBFM, -3000.0, 3000.0, 0, 99),
        (post.MODE_IQ, -3000.0, 3000.0, 0, 0),                 # the IQ mode: the AGC's complex output as (s2_t) pairs, network order
        (post.MODE_IQ, -2500.0, 2500.0, 0, 0),                 # ... and little-endian
    ]
    NR, steps = len(modes), 6
    mix = MIXES["survey"](NR, 123, N)
    adc = synth.adc_stream(N, 0x5EED0004)
    bank = _bank(NR, N, mix)
    try:
        for rx, (mode, lo, hi, de, sq) in enumerate(modes):
            bank.set_audio(rx, mix[rx][2], lo, hi, mode=mode, de_emp=de, squelch=sq)
        bank.set_little_endian(11, True)
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc, lambda k: d_adc, range(NR), steps)
        assert got["audio_blocks"] == 4 * NR, got              # 402 records a step: blocks on steps 2, 3, 4 and 6
        rc, sq, ave = bank.post.squelch_state(list(range(NR)))
        assert not sq[2] and sq[7] and sq[9], (sq, ave)         # value 0: always open; value 99 (threshold 0): forced shut
        s16 = bank.fetch("s16", [7, 9])
        assert np.all(s16[1][:512] == 1)                        # squelch.cpp:205-207: the squelched block is all ones ...
        assert not np.all(s16[0][:512] == 1)                    # ... which receiver 7's de-emphasis filter then filters (rx_sound.cpp:898-900)
        bank.ctx.free(d_adc)
    finally:
        bank.close()


def test_connections_join_and_leave_at_different_times(oracle):
    """Receivers of ONE bank whose connections start, end, start again and change mode on different steps (the reference gives
    every connection its own c2s_sound() / c2s_waterfall() loop, CFastFIR position and sequence numbers: rx/rx_sound.cpp:264-269,
    503-613): fourteen steps, every stage of every active receiver against an oracle that saw the same events.  A receiver that
    joined mid-block completes its 512-sample sound blocks on steps of its OWN; nobody else's state is touched by a join, a leave
    or a `SET mod=` (rounds 4-5: the bank demanded one record count and one FIR position of all its receivers)."""
    from flydog_sdr_gps_amd import post, synth
    from flydog_sdr_gps_amd.ddc import rx_phase_inc
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, UI_SRATE
    from flydog_sdr_gps_amd.wf import WfParams
    from tests.rxbank_check import check_bank
    n, steps, NR = 1 << 21, 14, 8
    hz = UI_SRATE / (1024 << 14)
    mix = []
    for k, zoom in enumerate([8, 9, 10, 8, 9, 10, 8, 9]):          # R = 128 one-shot (8192 R = 2^20 <= n), R = 256: one-shot, R = 512: overlapped
        span = UI_SRATE / (1 << zoom)
        p = WfParams.for_zoom(zoom, (0.0123 * ADC_CLOCK - span * (0.2 + 0.07 * k)) / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        mix.append((p, 8192 * p.decim > n, rx_phase_inc(0.0123 * ADC_CLOCK - 900.0 - 35.0 * k, ADC_CLOCK)))
    assert [ov for _, ov, _ in mix] == [False, False, True, False, False, True, False, False]
    ssb = dict(lo=300.0, hi=2700.0, mode=post.MODE_SSB)
    am = dict(lo=-2500.0, hi=2500.0, mode=post.MODE_AM, de_emp=1)
    nbfm = dict(lo=-3000.0, hi=3000.0, mode=post.MODE_NBFM, squelch=80)

    def join(rx, kw):
        return ("join", rx, mix[rx][0], mix[rx][1], mix[rx][2], kw)
    events = {
        0: [("leave", 4), ("leave", 5), ("leave", 6), ("leave", 7)],     # four connections at first
        2: [join(4, am)],
        3: [join(5, nbfm), ("leave", 1)],
        5: [join(6, ssb), ("audio", 2, am)],                            # receiver 2: SET mod=am on a running connection
        7: [join(1, nbfm)],                                             # receiver 1 comes back
        8: [join(7, am), ("wf", 3, mix[0][0], False)],                  # and a retune of receiver 3's waterfall among it all
        9: [("leave", 0)],
        11: [("audio", 5, ssb)],
    }
    adc = synth.adc_stream(n * 2, 0x5EED0061)
    bank = _bank(NR, n, mix)
    try:
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        got = check_bank(bank, lambda k: adc[(k % 2) * n:(k % 2 + 1) * n], lambda k: d_adc + 2 * (k % 2) * n, range(NR), steps, events=events)
        blocks = check_bank.blocks_of
        # 201.3 records a step: the four first connections complete their blocks on the same steps; the later ones on their own
        assert blocks[2] == blocks[3] and len(blocks[2]) == 5, blocks
        assert blocks[0] == [s for s in blocks[2] if s < 9], blocks
        for rx, at in ((4, 2), (5, 3), (6, 5), (1, 7), (7, 8)):
            mine = [s for s in blocks[rx] if s >= at]                  # (receiver 1 had a first connection: steps 0 .. 2)
            assert mine and mine[0] == at + 2, (rx, blocks[rx])        # 512 records take 2.54 steps from a standing start
        assert blocks[1][0] == blocks[2][0] and not [s for s in blocks[1] if 3 <= s < 7], blocks
        assert len({tuple(blocks[rx]) for rx in (2, 4, 5, 6, 1, 7)}) >= 4, blocks     # really different cadences
        assert got["audio_blocks"] == sum(len(v) for v in blocks.values()), got
        rc, sq, _ = bank.post.squelch_state([1, 5])
        bank.ctx.free(d_adc)
    finally:
        bank.close()
