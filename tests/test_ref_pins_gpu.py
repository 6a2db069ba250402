"""The HIP path against vectors the REFERENCE ITSELF produced (tests/golden/*_ref.*: outputs of the
reference's own rx/CuteSDR/agc.cpp, rx/csdr/ima_adpcm.cpp and gps/e1bcode.h, compiled from their
sources in place by oracle/build_ref.sh; generator tools/make_ref_golden.py).  The same vectors pin the
oracle in tests/test_ref_pins_cpu.py."""
import os

import numpy as np
import pytest

from flydog_sdr_gps_amd import Adpcm, Post, Searcher, post, sats, wire

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_cagc_on_the_gpu_matches_reference_agc_cpp(gpu_ctx):
    """m_Agc[ch].SetParameters / ProcessData (complex and mono16) / GetDelaySamples sequences against the outputs of
    rx/CuteSDR/agc.cpp ITSELF (agc_ref.npz): BIT-EXACT since round 6 -- log10f and powf are the host libm's algorithms on the
    device (csrc/kg_libm.h) -- complex outputs, mono16, delays (rounds 2-5: 1e-5 of full scale, one LSB)."""
    g = np.load(os.path.join(GOLD, "agc_ref.npz"))
    names = [str(n) for n in g["names"]]
    P = Post(gpu_ctx, nchan=len(names))
    for ch, name in enumerate(names):
        P.set_smeter(ch, 12000.0)
        P.reset(ch)
        x, want, pos, wpos = g[name + "_in"], g[name + "_out"], 0, 0
        for line in g[name + "_script"]:
            f = str(line).split()
            if f[0] == "P":
                P.set_agc(ch, *[int(v) for v in f[1:7]], float(f[7]))
            elif f[0] == "D":
                assert P.agc_delay(ch) == int(want[wpos]), name
                wpos += 1
            else:
                n = int(f[1])
                blk = x[pos:pos + n]
                pos += n
                P.set_mode(ch, post.MODE_IQ if f[0] == "C" else post.MODE_SSB)
                s16, _, agc = P.process([ch], blk.reshape(1, -1))
                if f[0] == "C":
                    w = want[wpos:wpos + 2 * n].view(np.complex64)
                    wpos += 2 * n
                    assert np.array_equal(np.ascontiguousarray(agc[0]).view(np.uint64), np.ascontiguousarray(w).view(np.uint64)), (name, np.abs(agc[0] - w).max())
                else:
                    w = want[wpos:wpos + n]
                    wpos += n
                    assert np.array_equal(s16[0].astype(np.int32), w.astype(np.int32)), (name, np.abs(s16[0].astype(np.int32) - w.astype(np.int32)).max())
        assert wpos == want.size and pos == x.size
    P.close()


def test_adpcm_on_the_gpu_matches_reference_ima_adpcm_cpp(gpu_ctx):
    g = np.load(os.path.join(GOLD, "adpcm_ref.npz"))
    x = g["i16_in"]
    A = Adpcm(gpu_ctx, nchan=3)
    A.set_state(1, 0, 0)
    enc = [A.encode([1], x[k:k + 512].reshape(1, -1))[0] for k in range(0, x.size, 512)]   # state carried on the device
    assert np.array_equal(np.concatenate(enc), g["i16_enc"])
    assert A.get_state(1) == tuple(int(v) for v in g["i16_enc_state"])
    A.set_state(2, 37, -1234)                                  # the audio_adpcm_state resume
    enc = [A.encode([2], x[k:k + 170].reshape(1, -1))[0] for k in range(0, 3400, 170)]
    assert np.array_equal(np.concatenate(enc), g["i16_enc_resumed"])
    assert A.get_state(2) == tuple(int(v) for v in g["i16_enc_resumed_state"])
    A.close()
    # compute_frame()'s compressed rows: 10 pad bytes + the row through encode_ima_adpcm_u8_e8
    rows = g["u8_rows"]
    pkts = wire.wf_packets(gpu_ctx, rows, [(100 + i, 3, i, True) for i in range(rows.shape[0])])
    for i, pk in enumerate(pkts):
        assert pk.size == 16 + 517 and np.array_equal(pk[16:], g["u8_enc"][i])


def test_e1b_code_tables_from_reference_chips(gpu_ctx, oracle):
    """SearchInit()'s E1B rows built from the chips the reference's E1BCODE produced."""
    g = np.load(os.path.join(GOLD, "e1b_ref.npz"))
    chips = np.unpackbits(g["chips_packed"], axis=1)[:, :4092]
    s = Searcher(gpu_ctx)
    for sat in (36, 44, 58):                                   # E02, E11, E36 rows of Sats[]
        prn = sats.SATS[sat][0]
        assert sats.SATS[sat][3] == sats.E1B
        s.set_code(sat, chips[prn - 1], boc=True)
        want = oracle.code_fft(chips[prn - 1], boc=True)
        got = s.get_code_fft(sat)
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    s.close()


class GpuCFir:
    """One CFir of a kg_post channel behind the script interface of tests/fixtures.run_fir_script."""

    def __init__(self, P, ch):
        self.P, self.ch, self.which = P, ch, post.CFIR_AM

    def init_lp(self, numtaps, scale, astop, fpass, fstop, fs):
        self.which = post.CFIR_AM
        return self.P.cfir_init_lp(self.ch, self.which, numtaps, scale, astop, fpass, fstop, fs)

    def init_hp(self, numtaps, scale, astop, fpass, fstop, fs):
        # the one high-pass the path designs: CSquelch::InitNoiseSquelch (squelch.cpp:137), through kg_post_squelch_setup
        assert (numtaps, scale, astop, fpass, fstop) == (0, 1.0, 50.0, 2400.0, 1950.0)
        self.which = post.CFIR_SQUELCH_HP
        self.P.squelch_setup(self.ch, fs)
        return self.P.cfir_taps(self.ch, self.which).size

    def init_const(self, coef, fs):
        self.which = post.CFIR_DEEMP_NFM
        self.P.cfir_init_const(self.ch, self.which, coef, fs)

    def process_rr(self, x):
        return self.P.cfir_process([self.ch], self.which, post.CFIR_REAL_REAL, x[None, :])[0]

    def process_rm(self, x):
        return self.P.cfir_process([self.ch], self.which, post.CFIR_REAL_MONO16, x[None, :])[0]

    def process_mm(self, x):
        return self.P.cfir_process([self.ch], self.which, post.CFIR_MONO16_MONO16, x[None, :])[0]


def test_cfir_on_the_gpu_matches_reference_fir_cpp(gpu_ctx):
    """m_AM_FIR / the de-emphasis filters / the squelch's high-pass against rx/CuteSDR/fir.cpp itself (fir_ref.npz): the designs
    (host arithmetic, taps read back through an impulse on the reference side, through the GPU here) and the three real-valued
    ProcessFilter paths, whose float sums the kernel forms in the reference's rotating order: BIT-EXACT."""
    from tests.fixtures import run_fir_script
    g = np.load(os.path.join(GOLD, "fir_ref.npz"))
    names = [str(n) for n in g["names"]]
    P = Post(gpu_ctx, nchan=len(names))
    try:
        for ch, name in enumerate(names):
            got = run_fir_script(GpuCFir(P, ch), g[name + "_script"], g[name + "_in"])
            want = g[name + "_out"]
            assert got.shape == want.shape, name
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
                "%s: %d of %d differ, max |diff| %g" % (name, np.count_nonzero(got != want), got.size, np.abs(got - want).max())
    finally:
        P.close()


class GpuSquelch:
    def __init__(self, P, ch):
        self.P, self.ch = P, ch

    def setup(self, rate):
        self.P.squelch_setup(self.ch, rate)

    def set_squelch(self, value, squelch_max):
        self.P.squelch_set(self.ch, value, squelch_max)

    def reset(self):
        self.P.squelch_reset(self.ch)

    def perform_fm(self, x):
        out, rc = self.P.squelch_perform([self.ch], x[None, :])
        return out[0], int(rc[0])


def test_squelch_on_the_gpu_matches_reference_squelch_cpp(gpu_ctx):
    """m_Squelch[ch].SetupParameters / SetSquelch / Reset / PerformFMSquelch sequences against rx/CuteSDR/squelch.cpp itself
    (squelch_ref.npz): outputs and return values BIT-EXACT (the noise average is a float/double recursion with no libm call)."""
    from tests.fixtures import run_squelch_script
    g = np.load(os.path.join(GOLD, "squelch_ref.npz"))
    names = [str(n) for n in g["names"]]
    P = Post(gpu_ctx, nchan=len(names))
    try:
        for ch, name in enumerate(names):
            got = run_squelch_script(GpuSquelch(P, ch), g[name + "_script"], g[name + "_in"])
            want = g[name + "_out"]
            assert got.shape == want.shape and np.array_equal(got, want), "%s: %d differ" % (name, np.count_nonzero(got != want))
    finally:
        P.close()


# ---- the FFT-dependent rows against the reference's own gps/search.cpp and rx/CuteSDR/fastfir.cpp ---------------------------
# (tests/golden/acq_fftref.npz, fastfir_fftref.npz: compiled in place against hipFFTW, run on a GPU box; the same vectors pin
# the oracle in tests/test_ref_pins_cpu.py)
FFT_TOL = 1e-5


def test_acquisition_on_the_gpu_matches_reference_search_cpp(gpu_ctx):
    """kg_acq behind the Searcher mirror, set up as SearchInit() leaves the reference (all 59 rows of Sats[]), against what
    gps/search.cpp ITSELF computed: code tables and Sample() spectra to 1e-5 of their largest bin, Correlate()'s Doppler bin and
    peak index EQUAL for all eleven (scene, SV) pairs -- present and absent SVs, both signs of Doppler (a negative bin reads the
    next satellite's row behind its own: search.cpp:471 over :54's doubled rows), E1B over 16368 lags -- and snr to 1e-5."""
    from flydog_sdr_gps_amd import synth
    from tests.fixtures import e1b_chips
    g = np.load(os.path.join(GOLD, "acq_fftref.npz"))
    keep = int(g["keep_every"])
    codes = synth.all_sv_codes(e1b_chips())
    s = Searcher(gpu_ctx, max_blocks=1)
    try:
        for sat, (chips, boc) in enumerate(codes):
            s.set_code(sat, chips, boc=boc)
        for sat in (int(v) for v in g["code_sats"]):
            mine = s.get_code_fft(sat)
            assert np.abs(mine[::keep] - g["code_%d_bins" % sat]).max() <= FFT_TOL * g["code_%d_max" % sat], sat
        negative = 0
        for name in (str(n) for n in g["scene_names"]):
            s.sample(g[name + "_bits"])
            data = s.get_data_fft()
            assert np.abs(data[::keep] - g[name + "_spec_bins"]).max() <= FFT_TOL * g[name + "_spec_max"], name
            asked = [int(v) for v in g[name + "_sats"]]
            res, _ = s.correlate_many(asked)
            for k, sat in enumerate(asked):
                r = res[0, k]
                assert (int(r["dop"]), int(r["idx"])) == (int(g[name + "_dop"][k]), int(g[name + "_idx"][k])), (name, sat, r)
                assert abs(float(r["snr"]) - g[name + "_snr"][k]) <= FFT_TOL * g[name + "_snr"][k], (name, sat, r["snr"], g[name + "_snr"][k])
                negative += int(r["dop"]) < 0
        assert negative >= 4
        # the row behind a satellite's own is what the table holds THERE: with satellite 20's row rewritten (to PRN 1's code) the
        # negative-Doppler result of satellite 19 changes, a positive-Doppler one does not
        name = "prn20_negative_doppler"
        s.sample(g[name + "_bits"])
        before, _ = s.correlate_many([19, 7])
        s.set_code(20, codes[0][0])
        after, _ = s.correlate_many([19, 7])
        assert float(after[0, 0]["snr"]) != float(before[0, 0]["snr"]) and float(after[0, 1]["snr"]) == float(before[0, 1]["snr"])
        assert (int(after[0, 0]["dop"]), int(after[0, 0]["idx"])) == (int(before[0, 0]["dop"]), int(before[0, 0]["idx"]))
    finally:
        s.close()


class GpuFastFir:
    def __init__(self, F, ch):
        self.F, self.ch, self.win, self.cic_on = F, ch, -1, False

    def window(self, w):
        self.win = w

    def cic(self, on):
        self.cic_on = on

    def setup(self, inst, lo, hi, off, fs):
        self.F.setup(self.ch, lo, hi, off, fs, window_func=self.win, do_cic_comp=self.cic_on)

    def process(self, x):
        y = self.F.process(self.ch, x)
        return y, self.F.pos(self.ch)


def test_fastfir_on_the_gpu_matches_reference_fastfir_cpp(gpu_ctx):
    """kg_fir against what rx/CuteSDR/fastfir.cpp ITSELF computed (fastfir_fftref.npz): counts and FirPos() equal, samples to
    1e-5 of the scenario's largest output."""
    from flydog_sdr_gps_amd import FastFir
    from tests.fixtures import fastfir_blocks, run_fastfir_script
    g = np.load(os.path.join(GOLD, "fastfir_fftref.npz"))
    names = [str(n) for n in g["names"]]
    F = FastFir(gpu_ctx, nchan=len(names), max_in=1024)
    try:
        for ch, name in enumerate(names):
            got = fastfir_blocks(g[name + "_script"], run_fastfir_script(GpuFastFir(F, ch), g[name + "_script"], g[name + "_in"]))
            want = fastfir_blocks(g[name + "_script"], g[name + "_out"])
            scale = max(float(np.abs(w[2]).max()) for w in want if w[0])
            for b, ((gc, gp, gy), (wc, wp, wy)) in enumerate(zip(got, want)):
                assert (gc, gp) == (wc, wp), (name, b, gc, gp, wc, wp)
                if wc:
                    assert np.abs(gy - wy).max() <= FFT_TOL * scale, (name, b, float(np.abs(gy - wy).max()), scale)
    finally:
        F.close()


def test_waterfall_on_the_gpu_matches_reference_rx_waterfall_cpp(gpu_ctx):
    """kg_wf with the reference's own tables against compute_frame() of rx/rx_waterfall.cpp ITSELF (wf_fftref.npz), eight frames:
    the 1024 output bytes under SURVEY section 7's (int)-edge rule (a pixel may differ by one count where the reference's dB
    value sits within 1e-5-of-maximum power of an integer; at most three per frame), and -- where the row is identical -- the
    wf_pkt_t payload (ADPCM where compression is on) byte for byte."""
    from flydog_sdr_gps_amd import Waterfall, wf
    g = np.load(os.path.join(GOLD, "wf_fftref.npz"))
    W = Waterfall(gpu_ctx, nchan=1)
    try:
        W.set_tables(g["window_function"], g["cic_comp"])
        exact = 0
        for k in range(int(g["ncases"])):
            zoom, start, interp, winf, cic, ovl, inv, comp = g["cases"][k]
            p = wf.WfParams.for_zoom(int(zoom), float(start), spectral_inversion=bool(inv))
            W.set_channel(0, p, interp=int(interp), window_func=int(winf), cic_comp=bool(cic), overlapped=bool(ovl), spectral_inversion=bool(inv))
            row = W.frames([0], g["case%d_iq" % k][None])[0]
            want = g["case%d_row" % k]
            diff = row.astype(int) - want.astype(int)
            assert np.count_nonzero(diff) <= 3 and np.abs(diff).max() <= 1, (k, int(np.count_nonzero(diff)), int(np.abs(diff).max()))
            if not np.any(diff):
                exact += 1
                nbytes, limit, xbin, flags, seq = (int(v) for v in g["case%d_hdr" % k])
                pkt = wire.wf_packets(gpu_ctx, row[None], [(xbin, p.zoom, seq, bool(comp))])[0]
                assert pkt.size - 16 == nbytes and np.array_equal(pkt[16:], g["case%d_payload" % k]), k
        assert exact >= 6, exact
    finally:
        W.close()


def test_dpump_unpack_on_the_gpu_matches_reference_data_pump_cpp(gpu_ctx):
    """kg_dpump_unpack_dev against snd_service() of rx/data_pump.cpp ITSELF (dpump_ref.npz): BIT-EXACT for 4 / 8 / 14 / 3 channels,
    disabled channels, spectral inversion, DC offsets, the -2^23 edge of the sign extension."""
    from flydog_sdr_gps_amd import snd
    from tests.fixtures import dpump_ref_cases
    g = np.load(os.path.join(GOLD, "dpump_ref.npz"))
    for name, nch, ns, inv, dci, dcq, en, bufs, rescale, per in dpump_ref_cases(g):
        assert np.float32(rescale) == np.float32(snd.RESCALE), name
        for b, d in enumerate(per[:4]):
            got = snd.unpack(gpu_ctx, bufs[b, :6 * ns * nch], ns, nch, enabled=en, dc_i=dci, dc_q=dcq, spectral_inversion=inv)
            for ch, (_, _, samps) in d.items():
                assert np.array_equal(got[ch].view(np.uint32), samps.view(np.uint32)), (name, b, ch)


def test_chan_start_matches_reference_channel_cpp():
    """kg_acq_chan_start (host arithmetic of the library) against the SPI commands CHANNEL::Start of gps/channel.cpp ITSELF sent
    (chan_ref.npz, 205 calls): lo_rate, ca_rate, ca_pause equal."""
    from flydog_sdr_gps_amd import handoff
    from tests.test_ref_pins_cpu import chan_ref_expect
    g = np.load(os.path.join(GOLD, "chan_ref.npz"))
    for is_e1b, lo_shift, ca_shift, secs, lo_rate, ca_rate, pause in chan_ref_expect(g):
        o = handoff.chan_start(is_e1b, lo_shift, ca_shift, secs)
        assert (o.lo_rate, o.ca_rate, o.ca_pause) == (lo_rate, ca_rate, pause), (is_e1b, lo_shift, ca_shift, secs)


def test_aperture_on_the_gpu_matches_reference_aperture_auto(gpu_ctx):
    """kg_aper_update_dev / kg_aper_report against aperture_auto() of rx/rx_waterfall.cpp ITSELF (aper_fftref.npz, made by the
    reference's compute_frame() on the GPU box): every average BIT-EXACT -- the IIR's too, expf being the host libm's algorithm on
    the device (csrc/kg_libm.h) -- and every report -- signal, noise, when -- equal."""
    from flydog_sdr_gps_amd import Aperture
    from tests.fixtures import aper_ref_replay
    g = np.load(os.path.join(GOLD, "aper_fftref.npz"))
    A = Aperture(gpu_ctx, nchan=1)
    try:
        def update(avg, row, algo, param, clear, start, stop, cal):
            A.update([0], row, [(algo, param, clear, start == 256)], waterfall_cal=cal)
            return A.get(0)

        def report(avg, start, stop):
            sig, noise = A.report([0], [start == 256])
            return int(sig[0]), int(noise[0])

        n = 0
        for k, avg, st, want_avg, want_st in aper_ref_replay(g, update, report):
            assert np.array_equal(avg.view(np.uint32), want_avg.view(np.uint32)), (k, g["frames"][k])
            assert st == want_st, (k, st, want_st)
            n += 1
        assert n == len(g["frames"])
    finally:
        A.close()


def test_sound_path_on_the_gpu_matches_the_references_own_statements(gpu_ctx):
    """kg_post, the payload kernels and kg_snd_header against rx/rx_sound.cpp:676-908, 1035-1140, 1222-1253 ITSELF (sndpath_ref.npz: the
    reference's statements, cut at build time, around its own CAgc / CFir / CSquelch / ADPCM coder): sMeterAvg_dB and its taps,
    out_samps_s2 of every mode family with and without de-emphasis, the IQ modes' AGC output, the squelch verdict, every payload byte
    (ADPCM with carried state, raw and IQ pairs in either byte order) and every header -- BIT-EXACT on the GPU (log10f / powf are the
    host libm's algorithms on the device, csrc/kg_libm.h)."""
    from tests.fixtures import GpuSoundPath, sndpath_check
    g = np.load(os.path.join(GOLD, "sndpath_ref.npz"))
    packets = blocks = samples = 0
    for name in (str(n) for n in g["names"]):
        P = Post(gpu_ctx, nchan=1)
        try:
            p, b, s = sndpath_check(g, name, lambda rate: GpuSoundPath(P, rate))
        finally:
            P.close()
        packets, blocks, samples = packets + p, blocks + b, samples + s
    assert packets >= 60 and blocks >= 100 and samples >= 50000, (packets, blocks, samples)


def test_fastfir_extension_taps_on_the_gpu_match_reference_fastfir_cpp(gpu_ctx):
    """kg_fir_process_taps_dev / kg_fir_refilter_dev against what CFastFIR::ProcessData of rx/CuteSDR/fastfir.cpp ITSELF handed a
    registered extension hook and returned (fastfir_taps_fftref.npz): PRE_FILTERED and POST_FILTERED buffers, and the output
    filtered from the buffer the PRE hook edited (`buf_modified`): counts, FirPos(), the hook calls equal; bins and samples to 1e-5
    of the largest."""
    from flydog_sdr_gps_amd import FastFir
    from tests.fixtures import fastfir_taps_cases
    from tests.test_ref_pins_cpu import _notch
    g = np.load(os.path.join(GOLD, "fastfir_taps_fftref.npz"))
    cases = fastfir_taps_cases(g)
    F = FastFir(gpu_ctx, nchan=len(cases), max_in=2048)
    try:
        for ch, (name, script, x, per) in enumerate(cases):
            pos = flags = edit = k = 0
            cic_on = False
            for line in script:
                t = line.split()
                if t[0] == "C":
                    cic_on = int(t[1]) != 0
                elif t[0] == "P":
                    F.setup(ch, *[float(v) for v in t[2:6]], do_cic_comp=cic_on)
                elif t[0] == "H":
                    flags, edit = int(t[1]), int(t[2])
                else:
                    n, count, firpos, taps, want = per[k]
                    k += 1
                    blk = x[pos:pos + n]
                    pos += n
                    if flags & 1 and edit:
                        out, pre, _ = F.process_taps_edit(ch, blk, _notch)
                        post = None
                    else:
                        out, pre, post = F.process_taps(ch, blk)
                    assert (out.size, F.pos(ch)) == (count, firpos), (name, k)
                    per_blk = len([1 for fl in (1, 2) if flags & fl])
                    for i, (fl, bins) in enumerate(taps):
                        got = pre[i // per_blk] if fl == 1 else post[i // per_blk]
                        assert np.abs(got - bins).max() <= FFT_TOL * np.abs(bins).max(), (name, k, i, fl)
                    if count:
                        assert np.abs(out - want).max() <= FFT_TOL * np.abs(want).max(), (name, k, float(np.abs(out - want).max()))
    finally:
        F.close()


def test_am_passband_on_the_gpu_matches_reference_rx_sound_cmd_cpp(gpu_ctx):
    """kg_post_set_am_passband against the reference's own passband statements (rx/rx_sound_cmd.cpp:243-286 cut at build time around
    its own CFir; sndcmd_ref.npz): the taps the library designs for the client's cuts -- beyond the Nyquist limit included, where the
    handler clamps them first -- equal m_AM_FIR's, bit for bit."""
    g = np.load(os.path.join(GOLD, "sndcmd_ref.npz"))
    P = Post(gpu_ctx, nchan=1)
    try:
        for (lo, hi, rate), w_taps in zip(g["bands"], g["am_fir"]):
            if lo == 0 and hi == 0:
                continue
            n = P.set_am_passband(0, lo, hi, rate)
            taps = P.cfir_taps(0, post.CFIR_AM)
            assert n == taps.size and np.array_equal(taps.view(np.uint32), w_taps[:n].view(np.uint32)) and not w_taps[n:].any(), (lo, hi, rate, n)
    finally:
        P.close()
