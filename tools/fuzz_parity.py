"""Randomised differential soak: many random configurations of every stateful module through the
C ABI against the oracle (bit-exact where the module is integer, the module's bar otherwise).
usage: python tools/fuzz_parity.py [seconds per module] [seed] [module name: only that one]
A failing `post` trial leaves its story (mode, parameter sets, inputs) in gpurun_out/fuzz_fail_post_<k>.npz."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flydog_sdr_gps_amd import (Adpcm, Context, Ddc, FastFir, Post, RxDdc, Searcher, post, prn, sats, snd, synth,   # noqa: E402
                                wire)
from flydog_sdr_gps_amd.ddc import RX_DECIM   # noqa: E402
from oracle import kiwi_oracle as ko          # noqa: E402
from tests.fixtures import am_passband, arm_audio_tail, oracle_row    # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
only = sys.argv[3] if len(sys.argv) > 3 else None
last_story = {}
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
ctx = Context(0)
fails = 0


def adc_block(n):
    t = np.arange(n)
    x = rng.normal(0, rng.uniform(1, 3000), n)
    for _ in range(int(rng.integers(0, 4))):
        x += rng.uniform(10, 12000) * np.cos(2 * np.pi * rng.uniform(0, 0.5) * t + rng.uniform(0, 6))
    if rng.random() < 0.1:
        x[:] = rng.choice([-32768, 32767, 0])
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def soak(name, trial):
    global fails
    if only is not None and name.replace(" ", "") != only.replace(" ", ""):
        return
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        try:
            trial()
        except AssertionError as e:
            fails += 1
            print("FAIL %s trial %d: %s" % (name, n, e))
            if last_story.get("module") == name:
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez("gpurun_out/fuzz_fail_%s_%d.npz" % (name, fails), **{k: v for k, v in last_story.items() if k != "module"})
            if fails > 5:
                sys.exit(1)
        n += 1
    print("%-10s %5d trials ok" % (name, n))


def trial_wfddc():
    """Random sequences of everything the waterfall DDC's object does between two samples: continuous pushes (all channels or a
    subset), one-shot captures (CmdWFReset + sampler, random sampler size), CIC resets, retunes, new phases -- in line or
    with the deferred output stage -- against an oracle channel model that is told the same story."""
    nch = int(rng.integers(1, 5))
    big = rng.random() < 0.15                        # long blocks: the carry scan cut into chunks across workgroups
    d = Ddc(ctx, nchan=nch, max_samples=1 << (20 if big else 17))
    incs = [int(rng.integers(0, 1 << 48)) for _ in range(nch)]
    l2 = [int(rng.integers(0, 14)) for _ in range(nch)]
    M48 = (1 << 48) - 1
    states = [None] * nch                            # oracle filter state (None: reset)
    phase = [0] * nch                                # NCO phase of each channel's next sample
    stale = [False] * nch                            # a capture cut the filters short: the next continuous push starts from reset
    for ch in range(nch):
        d.set_wf(ch, incs[ch], 1 << l2[ch])
    d.set_deferred(bool(rng.random() < 0.4))

    def fresh(ch):
        st = ko.DdcWfState()
        st.phase = phase[ch] & M48
        return st
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.integers(1 << 17, 1 << 20)) if big else int(rng.integers(1, 1 << 16))
        adc = adc_block(n)
        chans = list(range(nch))
        if nch > 1 and rng.random() < 0.3:
            chans = sorted(rng.choice(nch, int(rng.integers(1, nch + 1)), replace=False).tolist(), key=lambda _: rng.random())
        what = rng.random()
        if what < 0.35:                              # capture
            mo = int(rng.choice([1, 7, 300, 8192]))
            got = d.capture(adc, chans, mo)
            for i, ch in enumerate(chans):
                need = min(n, mo << l2[ch])
                want, _ = ko.ddc_wf(adc[:need], incs[ch], l2[ch], fresh(ch))
                assert np.array_equal(got[i], want[:mo]), ("wf capture", l2[ch], n, mo)
                phase[ch] = (phase[ch] + n * incs[ch]) & M48
                stale[ch], states[ch] = True, None
        else:
            got = d.push(adc, chans)
            for i, ch in enumerate(chans):
                if stale[ch] or states[ch] is None:
                    states[ch] = fresh(ch)
                    stale[ch] = False
                want, states[ch] = ko.ddc_wf(adc, incs[ch], l2[ch], states[ch])
                assert np.array_equal(got[i], want), ("wf ddc", l2[ch], n)
                phase[ch] = (phase[ch] + n * incs[ch]) & M48
        ev = rng.random()                            # something the host does between blocks
        ch = int(rng.integers(0, nch))
        if ev < 0.15:
            d.reset(ch); states[ch], stale[ch] = None, False
        elif ev < 0.25:
            incs[ch] = int(rng.integers(0, 1 << 48)); l2[ch] = int(rng.integers(0, 14))
            d.set_wf(ch, incs[ch], 1 << l2[ch]); states[ch], stale[ch], phase[ch] = None, False, 0
        elif ev < 0.32:
            phase[ch] = int(rng.integers(0, 1 << 48))
            d.set_phase(ch, phase[ch])
            if states[ch] is not None:
                states[ch].phase = phase[ch]
    d.close()


def trial_rxddc():
    nch = int(rng.integers(1, 4))
    d = RxDdc(ctx, nchan=nch, max_samples=1 << 17)
    incs = [int(rng.integers(0, 1 << 48)) for _ in range(nch)]
    states = [None] * nch
    for ch in range(nch):
        d.set_freq(ch, incs[ch])
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 6 * RX_DECIM))
        adc = adc_block(n)
        got = d.push(adc, list(range(nch)))
        for ch in range(nch):
            want, states[ch] = ko.ddc_rx(adc, incs[ch], states[ch])
            assert np.array_equal(got[ch], want), ("rx ddc", n)
    d.close()


def trial_fir():
    f = FastFir(ctx, nchan=1, max_in=4096)
    lo = float(rng.uniform(-5000, 4000))
    hi = float(lo + rng.uniform(100, 5000))
    if not f.setup(0, lo, min(hi, 5900.0), float(rng.uniform(-200, 200)), 12000.0, window_func=int(rng.integers(-1, 5)),
                   do_cic_comp=bool(rng.integers(0, 2))):
        f.close()
        return
    coef = f.get_coef(0)
    st = ko.fir_new_state()
    scale = 0.0
    for _ in range(int(rng.integers(2, 8))):
        n = int(rng.integers(1, 1500))
        x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * rng.uniform(1, 20000)).astype(np.complex64)
        scale = max(scale, float(np.abs(x).max()))
        got = f.process(0, x)
        want, pos = ko.fir_process(st, coef, x, prec=0)
        assert got.size == want.size and f.pos(0) == pos, "fir sizes"
        if got.size:
            # 1e-5 of the block's maximum, but not below the fp32 rounding of what went through the
            # transform: a block that is mostly start-up zeros or stop-band content is tiny next to
            # its input (observed: errors ~1e-7 of the input amplitude, outputs 0.2 % of it)
            bar = 1e-5 * max(float(np.abs(want).max()), 0.05 * scale)
            assert np.abs(got - want).max() <= bar, ("fir values", float(np.abs(got - want).max()), bar)
    f.close()


def trial_post():
    P = Post(ctx, nchan=1)
    a = ko.Agc()
    mode = int(rng.integers(0, 4))
    P.set_mode(0, mode)
    arm_audio_tail(P, 0)
    P.set_smeter(0, 12000.0)
    P.reset(0)
    avg, alpha, z1, last = 0.0, ko.smeter_alpha(12000.0), 0.0, (0.0, 0.0)
    last_story.clear()
    last_story.update(module="post", mode=np.int64(mode))
    for seg in range(int(rng.integers(1, 6))):
        prm = (bool(rng.integers(0, 5)), bool(rng.integers(0, 2)), int(rng.integers(-130, -20)), int(rng.integers(0, 90)),
               int(rng.integers(0, 11)), int(rng.choice([20, 100, 500, 1000, 5000])), float(rng.choice([12000.0, 20250.0])))
        P.set_agc(0, *prm)
        a.set_parameters(*prm)
        n = int(rng.integers(1, 1025))
        t = np.arange(n)
        env = rng.uniform(1, 20000) * np.exp(-t / rng.uniform(50, 5000)) if rng.random() < 0.5 else np.full(n, rng.uniform(0, 9000))
        x = (env * np.exp(2j * np.pi * rng.uniform(-0.4, 0.4) * t) + rng.normal(0, rng.uniform(0, 50), n)).astype(np.complex64)
        last_story["prm%d" % seg] = np.array([float(v) for v in prm])
        last_story["x%d" % seg] = x
        s16, demod, agc = P.process([0], x[None, :])
        avg, _ = ko.smeter_process(avg, alpha, x)
        # BIT-EXACT (round 6): log10f (which CAgc branches on) and powf are the host libm's own algorithms on the device
        # (csrc/kg_libm.h).  Rounds 4-5 held 1 LSB / 2e-5 / 2e-4 dB here and let a once-in-10^6 gain step through.
        if mode == post.MODE_SSB:
            want = a.process_s16(x)
            assert np.array_equal(s16[0].astype(int), want.astype(int)), ("post s16", np.abs(s16[0].astype(int) - want.astype(int)).max())
        else:
            want = a.process_cpx(x)
            assert np.array_equal(np.ascontiguousarray(agc[0]).view(np.uint64), np.ascontiguousarray(want, np.complex64).view(np.uint64)), "post cpx"
            if mode == post.MODE_AM:
                wd, z1 = ko.am_detect(z1, want)
                assert np.array_equal(np.ascontiguousarray(demod[0]).view(np.uint32), np.asarray(wd, np.float32).view(np.uint32)), "post am"
            elif mode == post.MODE_NBFM:
                wd, last = ko.nbfm_detect(last, want)
                assert np.array_equal(np.ascontiguousarray(demod[0]).view(np.uint32), np.asarray(wd, np.float32).view(np.uint32)), "post nbfm"
        got_avg, _ = P.smeter([0])
        assert np.float32(got_avg[0]).view(np.uint32) == np.float32(avg).view(np.uint32), ("smeter", got_avg[0], avg)
    P.close()


def trial_tail():
    """What follows the detectors (round 6): a CFir with a random Kaiser design, forced tap count or caller's coefficients driven in
    ragged blocks through its three ProcessFilter paths, and a CSquelch with random thresholds over noise / quiet stretches, both
    through their standalone C-ABI entry points -- BIT-EXACT against the oracle (which rx/CuteSDR/fir.cpp and squelch.cpp, built
    in place, pin bit for bit)."""
    P = Post(ctx, nchan=2)
    try:
        f = ko.CFir()
        rate = float(rng.choice([12000.0, 20250.0]))
        kind = int(rng.integers(0, 3))
        if kind == 0:                                              # the AM filter of a passband change, any half bandwidth
            lo = -float(rng.uniform(50, rate / 2 + 500))
            hi = float(rng.uniform(50, rate / 2 + 500))
            n1 = P.set_am_passband(0, lo, hi, rate)
            _, _, hbw, stop = am_passband(lo, hi, rate)
            assert n1 == f.init_lp(0, 1.0, 50.0, hbw, stop, rate), "am fir taps"
        elif kind == 1:                                            # any low-pass the interface takes
            a = (int(rng.choice([0, 0, 5, 33, 96, 97])), float(rng.uniform(0.2, 2.0)), float(rng.uniform(10, 80)),
                 float(rng.uniform(100, 3000)), float(rng.uniform(3100, 5900)), rate)
            assert P.cfir_init_lp(0, post.CFIR_AM, *a) == f.init_lp(*a), "lp taps"
        else:
            taps = rng.normal(0, 0.2, int(rng.integers(1, 98))).astype(np.float32)
            P.cfir_init_const(0, post.CFIR_AM, taps, rate)
            f.init_const(taps, rate)
        assert np.array_equal(P.cfir_taps(0, post.CFIR_AM).view(np.uint32), f.taps().view(np.uint32)), "taps"
        for _ in range(int(rng.integers(1, 7))):
            n = int(rng.integers(1, 1025))
            k = int(rng.integers(0, 3))
            if k == 2:
                x = rng.integers(-32768, 32768, n).astype(np.int16)
                got, want = P.cfir_process([0], post.CFIR_AM, post.CFIR_MONO16_MONO16, x[None])[0], f.process_mm(x)
            else:
                x = (rng.normal(0, 10.0 ** rng.uniform(0, 4.6), n)).astype(np.float32)
                if k == 0:
                    got, want = P.cfir_process([0], post.CFIR_AM, post.CFIR_REAL_REAL, x[None])[0], f.process_rr(x)
                    got, want = got.view(np.uint32), want.view(np.uint32)
                else:
                    got, want = P.cfir_process([0], post.CFIR_AM, post.CFIR_REAL_MONO16, x[None])[0], f.process_rm(x)
            assert np.array_equal(got, want), ("cfir", kind, k, n)
        q = ko.Squelch()
        P.squelch_setup(1, rate)
        q.setup(rate)
        for _ in range(int(rng.integers(2, 10))):
            if rng.random() < 0.4:
                v, m = int(rng.integers(0, 100)), int(rng.choice([0, 0, 3000, 8192, 20000]))
                P.squelch_set(1, v, m)
                q.set_squelch(v, m)
            elif rng.random() < 0.1:
                P.squelch_reset(1)
                q.reset()
            n = int(rng.integers(1, 1025))
            x = np.clip(rng.normal(0, rng.uniform(10, 7000), n), -8192, 8192).astype(np.float32)
            try:
                got, rc = P.squelch_perform([1], x[None])
            except Exception as e:                                  # before the first SetSquelch the library refuses; the oracle's
                if "squelch" in str(e):                             # state then has indeterminate value / threshold in the reference
                    P.squelch_set(1, 0, 0)
                    q.set_squelch(0, 0)
                    got, rc = P.squelch_perform([1], x[None])
                else:
                    raise
            want, wrc = q.perform_fm(x)
            assert np.array_equal(got[0], want) and int(rc[0]) == wrc, ("squelch", n, int(rc[0]), wrc)
    finally:
        P.close()


def trial_wire():
    nch = int(rng.integers(1, 6))
    A = Adpcm(ctx, nchan=nch)
    sts = [None] * nch
    for _ in range(int(rng.integers(1, 4))):
        n = 2 * int(rng.integers(1, 600))
        x = np.clip(np.rint(rng.normal(0, rng.uniform(1, 20000), (nch, n))), -32768, 32767).astype(np.int16)
        got = A.encode(list(range(nch)), x)
        for ch in range(nch):
            want, sts[ch] = ko.adpcm_encode_i16(x[ch], sts[ch])
            assert np.array_equal(got[ch], want), "adpcm"
    A.close()
    rows = rng.integers(0, 256, (3, 1024)).astype(np.uint8)
    infos = [(int(rng.integers(0, 1 << 31)), int(rng.integers(0, 15)), int(rng.integers(0, 1 << 31)), bool(rng.integers(0, 2)))
             for _ in range(3)]
    for r, p in enumerate(wire.wf_packets(ctx, rows, infos)):
        assert np.array_equal(p, ko.wf_packet(rows[r], *infos[r])), "wf packet"


_acq = {}


def trial_acq():
    """A block of 1-bit IF holding one SV -- C/A or Galileo E1B (BOC(1,1), all 16368 lags or a random window longer
    than 4096: the 512-thread correlator) -- at a random code phase / Doppler / C/N0, or absent; a random list of 1 .. 12 SVs
    of both kinds is searched (fewer than 8 pairs: cells dealt over the XCDs; more: pairs).  For the SV that is there the GPU's
    (valid, Doppler bin, index) must be the oracle's and snr within 2e-5; for the others the level must agree."""
    from tests.fixtures import e1b_chips
    if not _acq:
        _acq["s"] = Searcher(ctx, max_blocks=1)
        _acq["codes"] = {}
        _acq["e1b"] = e1b_chips()
    s = _acq["s"]

    def code_of(sat, limit=None):
        prn_no, t1, t2, kind = sats.SATS[sat]
        boc = kind == sats.E1B
        chips = _acq["e1b"][prn_no] if boc else prn.cacode(t1, t2)
        lim = limit if limit is not None else (sats.E1B_LIMIT if boc else sats.L1_LIMIT)
        if _acq["codes"].get(sat, (None, None))[1] != lim:
            s.set_code(sat, chips, boc=boc, limit=lim)
            _acq["codes"][sat] = (chips, lim)
        return chips, boc, lim
    nsv = int(rng.integers(1, 13))
    svs = [int(x) for x in rng.choice(len(sats.SATS), nsv, replace=False)]
    target = svs[int(rng.integers(0, nsv))]
    tkind = sats.SATS[target][3] == sats.E1B
    tlimit = int(rng.integers(4097, 16385)) if tkind and rng.random() < 0.5 else None
    info = {sat: code_of(sat, tlimit if sat == target else None) for sat in svs}
    chips, boc, _ = info[target]
    present = rng.random() < 0.85
    other = prn.cacode(*sats.SATS[(target + 7) % 32][1:3])
    scene = [(chips if present else other, float(rng.uniform(0, chips.size)), float(rng.uniform(-5000, 5000)),
              float(rng.uniform(0, 6.28)), float(rng.uniform(40, 50)), bool(boc and present))]
    bits = synth.gps_scene_bits(scene, int(rng.integers(0, 1 << 31)))
    s.sample(bits, block=0)
    got, _ = s.correlate_many(svs, nblocks=1, want_cells=False)
    data = ko.sample_bits(bits, prec=1)
    for k, sat in enumerate(svs):
        c, b, lim = info[sat]
        g = got[0, k]
        res = ko.correlate(ko.code_fft(c, boc=b, prec=1), data, limit=lim, code_next=oracle_row(ko, s, sat + 1))
        w_snr, w_dop, w_idx, w_valid = (res[0][x] for x in ("snr", "dop", "idx", "valid"))
        if w_snr >= 24:                                 # a detection: everything must agree
            assert (int(g["valid"]), int(g["dop"]), int(g["idx"])) == (int(w_valid), int(w_dop), int(w_idx)), ("acq", sat, lim, g, res[0])
            assert abs(float(g["snr"]) - w_snr) <= 2e-5 * w_snr, ("acq snr", sat, g["snr"], w_snr)
        else:                                           # noise: the maximum may sit on a near-tie; the level must agree
            assert abs(float(g["snr"]) - w_snr) <= 1e-3 * max(w_snr, 1.0), ("acq noise snr", sat, g["snr"], w_snr)


_wf = {}


def trial_wf():
    """Waterfall frames: random zoom / start / interpolation / window / CIC compensation / inversion on
    random frames, batched over a random frame -> channel map (the device copy of the map is cached)."""
    from flydog_sdr_gps_amd import Waterfall, WfParams, wf
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.test_wf_gpu import check_row, db_bound, oracle_frame
    if not _wf:
        _wf["tables"] = (wf.window_functions(), wf.cic_comp_table())
        _wf["w"] = Waterfall(ctx, nchan=6)
        _wf["w"].set_tables(*_wf["tables"])
    w, tables = _wf["w"], _wf["tables"]
    cfg = []
    for ch in range(6):
        zoom = int(rng.integers(0, 15))
        inv = bool(rng.integers(0, 2))
        p = WfParams.for_zoom(zoom, float(rng.uniform(0, 2.0e7)), spectral_inversion=inv)
        interp = int(rng.choice([wf.WF_MAX, wf.WF_MIN, wf.WF_LAST, wf.WF_DROP, wf.WF_CMA]))
        win = int(rng.integers(0, 4))
        comp = bool(rng.integers(0, 2))
        w.set_channel(ch, p, interp=interp, window_func=win, cic_comp=comp, spectral_inversion=inv)
        cfg.append((p, interp, win, comp, inv))
    nfr = int(rng.integers(1, 9))
    chan_of = [int(c) for c in rng.integers(0, 6, nfr)]
    iqs = [synth.wf_iq_frame(seed=int(rng.integers(0, 1 << 30)), noise_dbfs=float(rng.uniform(-70, -20))) for _ in range(nfr)]
    out = w.frames(chan_of, np.stack(iqs))
    for k, ch in enumerate(chan_of):
        p, interp, win, comp, inv = cfg[ch]
        w_out, _, w_po, w_dB = oracle_frame(ko, tables, iqs[k], p, interp, win, comp, False, inv)
        check_row(out[k], w_out, w_dB, db_bound(w_po))


def trial_rxbank():
    """A receiver bank (kg_rxbank): random receiver count, step length, zooms, sampler modes (wherever both are allowed a coin
    decides), audio NCOs, a handful of steps over a random stream with retunes in between; every stage of every receiver
    against the oracle (tests/rxbank_check.py)."""
    from flydog_sdr_gps_amd.ddc import rx_phase_inc
    from flydog_sdr_gps_amd.rxbank import ADC_CLOCK, UI_SRATE, RxBank
    from flydog_sdr_gps_amd.wf import WfParams
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.rxbank_check import check_bank
    log2n = int(rng.integers(15, 19))
    n = 1 << log2n
    nrx = int(rng.integers(1, 9))
    steps = int(rng.integers(2, 12))
    hz = UI_SRATE / (1024 << 14)

    def receiver():
        zoom = int(rng.integers(1, 15))
        decim = 1 << max(zoom - 1, 0)
        span = UI_SRATE / (1 << zoom)
        # (a span that holds the stream's strong carrier: the rows' tolerance is relative to the largest bin)
        p = WfParams.for_zoom(zoom, max(0.0123 * ADC_CLOCK - span * rng.uniform(0.1, 0.6), 0.0) / hz, adc_clock=ADC_CLOCK, ui_srate=UI_SRATE)
        can_shot, can_ov = 8192 * decim <= n, n // decim <= 8192 and n // decim >= 2
        ov = bool(rng.integers(0, 2)) if (can_shot and can_ov) else can_ov
        return p, ov, rx_phase_inc(0.0123 * ADC_CLOCK - rng.uniform(400.0, 2500.0), ADC_CLOCK)
    mix = [receiver() for _ in range(nrx)]
    events = {}
    modes = [dict(lo=300.0, hi=2700.0, mode=post.MODE_SSB), dict(lo=-2500.0, hi=2500.0, mode=post.MODE_AM, de_emp=int(rng.integers(0, 3))),
             dict(lo=-3000.0, hi=3000.0, mode=post.MODE_NBFM, squelch=int(rng.choice([0, 60, 85, 99])), de_emp=int(rng.integers(0, 3))),
             dict(lo=-3000.0, hi=3000.0, mode=post.MODE_IQ)]
    gone = set()
    for st in range(0, steps):
        if st == 0:
            for rx in range(nrx):                                     # some connections are not there at first
                if nrx > 1 and rng.random() < 0.25:
                    events.setdefault(0, []).append(("leave", rx))
                    gone.add(rx)
            continue
        r = rng.random()
        rx = int(rng.integers(0, nrx))
        if r < 0.25 and rx not in gone:
            p, ov, inc = receiver()
            events.setdefault(st, []).append(("wf", rx, p, ov) if rng.random() < 0.6 else ("freq", rx, inc))
        elif r < 0.45 and gone:
            rx = int(rng.choice(sorted(gone)))
            p, ov, inc = receiver()
            mix[rx] = (p, ov, inc)
            events.setdefault(st, []).append(("join", rx, p, ov, inc, dict(modes[int(rng.integers(0, 4))])))
            gone.discard(rx)
        elif r < 0.55 and rx not in gone and len(gone) < nrx - 1:
            events.setdefault(st, []).append(("leave", rx))
            gone.add(rx)
        elif r < 0.7 and rx not in gone:
            events.setdefault(st, []).append(("audio", rx, dict(modes[int(rng.integers(0, 4))])))
    t = np.arange(n * steps)
    x = rng.normal(0, 10.0, n * steps) + 3000.0 * np.cos(2 * np.pi * 0.0123 * t) + rng.uniform(0, 2000) * np.cos(2 * np.pi * rng.uniform(0, 0.5) * t)
    adc = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    bank = RxBank(nrx, n)
    try:
        bank.configure(mix)
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        try:
            check_bank(bank, lambda k: adc[k * n:(k + 1) * n], lambda k: d_adc + 2 * k * n, range(nrx), steps, threads=4, events=events)
        except AssertionError as e:
            raise AssertionError("n 2^%d, %d receivers %s, %d steps, events %s: %s" % (
                log2n, nrx, [(p.zoom, ov) for p, ov, _ in mix], steps, {k: [(e_[0], e_[1]) for e_ in v] for k, v in events.items()}, e))
        bank.ctx.free(d_adc)
    finally:
        bank.close()


for name, fn in (("rxbank", trial_rxbank), ("tail", trial_tail), ("acq", trial_acq), ("wf ddc", trial_wfddc), ("rx ddc", trial_rxddc), ("fastfir", trial_fir), ("post", trial_post), ("wire", trial_wire), ("wf frames", trial_wf)):
    soak(name, fn)
print("failures:", fails)
if _acq:
    _acq["s"].close()
if _wf:
    _wf["w"].close()
ctx.close()
sys.exit(1 if fails else 0)
