"""The oracle against vectors the REFERENCE ITSELF produced (tests/golden/*_ref.*, written by
tools/make_ref_golden.py from oracle/_ref/, which oracle/build_ref.sh compiles from the reference's
own sources in place).  These are the pins DESIGN.md section 3 lists; the HIP path is compared with
the same vectors in tests/test_ref_pins_gpu.py."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def chips_to_hex(chips):
    """4092 chips -> the 1023 hex digits of the ICD table, MSB first per nibble (gps/e1bcode.h:84, bit = 3 - (i % 4))."""
    v = np.asarray(chips, np.uint8).reshape(-1, 4)
    return "".join("%X" % (8 * a + 4 * b + 2 * c + d) for a, b, c, d in v)


def test_e1b_all_50_codes_match_reference_e1bcode_h(oracle):
    """gps/e1bcode.h:62-92 E1BCODE(prn) for prn 1..50 (NUM_E1B_SATS): e1b_ref.npz holds the chips the reference's own
    class produced (outputs only).  The hex table is the reference's text and is NOT kept under tests/: where
    $REFERENCE is present (the build container) it is read from gps/e1bcode.h here; elsewhere the digits are
    re-derived from the chips, which still pins the nibble order of both from-hex routines."""
    import re
    from flydog_sdr_gps_amd import prn
    g = np.load(os.path.join(GOLD, "e1b_ref.npz"))
    assert sorted(g.files) == ["chips_packed"], "outputs of the reference only -- no reference text under tests/"
    chips = np.unpackbits(g["chips_packed"], axis=1)[:, :4092]
    assert chips.shape == (50, 4092)
    hdr = os.path.join(os.environ.get("REFERENCE", "/root/reference"), "gps", "e1bcode.h")
    if os.path.isfile(hdr):
        hexes = re.findall(r'"([0-9A-F]{1023})"', open(hdr).read())
        assert len(hexes) == 50
        assert hexes == [chips_to_hex(c) for c in chips]
    else:
        hexes = [chips_to_hex(c) for c in chips]
    for i in range(50):
        assert np.array_equal(oracle.e1b_from_hex(hexes[i]), chips[i]), "oracle, E%02d" % (i + 1)
        assert np.array_equal(prn.e1b_from_hex(hexes[i]), chips[i]), "host mirror, E%02d" % (i + 1)
    # the reference's own known answers (gps/search.cpp:295,302)
    assert int("".join(map(str, chips[0, :20])), 2) == 0xf5d71
    assert int("".join(map(str, chips[1, :20])), 2) == 0x96b85
    assert chips_to_hex(chips[0])[:5] == "F5D71"


def test_constants_match_reference_headers(oracle):
    """gps/gps.h, kiwi.h and the kiwi.gen.h the reference's assembler generates from kiwi.config."""
    from flydog_sdr_gps_amd import acq, ddc, sats
    c = json.load(open(os.path.join(GOLD, "consts_ref.json")))
    assert (oracle.NSAMPLES, oracle.FFT_LEN, oracle.DECIM) == (c["NSAMPLES"], c["FFT_LEN"], c["DECIM"])
    assert (acq.NSAMPLES, acq.FFT_LEN, acq.DECIM, acq.BIN_SIZE, acq.MIN_SIG) == \
        (c["NSAMPLES"], c["FFT_LEN"], c["DECIM"], c["BIN_SIZE"], c["MIN_SIG"])
    assert (oracle.L1_CODELEN, oracle.E1B_CODELEN) == (c["L1_CODELEN"], c["E1B_CODELEN"])
    assert oracle.L1_LIMIT == c["SAMPLE_RATE"] // 1000 * c["L1_CODE_PERIOD"]          # search.cpp:486
    assert oracle.E1B_LIMIT == c["SAMPLE_RATE"] // 1000 * c["E1B_CODE_PERIOD"]
    assert (sats.L1_LIMIT, sats.E1B_LIMIT, sats.MAX_SATS) == (oracle.L1_LIMIT, oracle.E1B_LIMIT, c["MAX_SATS"])
    assert (acq.DOP_LO, acq.DOP_HI) == (int(-5000 / c["BIN_SIZE"]), int(5000 / c["BIN_SIZE"]))   # search.cpp:465
    assert ddc.RX_DECIM == c["RX1_STD_DECIM"] * c["RX2_STD_DECIM"] * c["VAL_CICF_DECIM_BY_2"] == oracle.RX_DECIM
    assert (c["FS"], c["FC"], c["CPS"]) == (16.368e6, 4.092e6, 1.023e6)
    # the hand-off restatement (gps/channel.cpp:281-311) uses exactly these
    o = oracle.chan_start(False, 1, 0, 0.0)
    assert o.lo_dop == c["BIN_SIZE"] and o.ca_dop == c["BIN_SIZE"] / c["L1_f"] * c["CPS"]
    assert o.lo_rate == int((c["FC"] + o.lo_dop) / c["FS"] * 2.0 ** 32)
    assert oracle.chan_start(True, 0, 0, 0.0).ca_pause == c["FS_I"] // 1000 * c["E1B_CODE_PERIOD"]


def run_agc_script(agc, script, x):
    """The script language of oracle/ref/ref_agc_main.cpp on an object with set_parameters /
    process_cpx / process_s16 / delay."""
    out, pos = [], 0
    for line in script:
        f = str(line).split()
        if f[0] == "P":
            agc.set_parameters(*[int(v) for v in f[1:7]], float(f[7]))
        elif f[0] == "D":
            out.append(np.array([agc.delay()], np.float32))
        else:
            n = int(f[1])
            blk = x[pos:pos + n]
            pos += n
            if f[0] == "C":
                out.append(agc.process_cpx(blk).view(np.float32))
            else:
                out.append(agc.process_s16(blk).astype(np.float32))
    return np.concatenate(out)


class OracleAgc:
    def __init__(self, oracle):
        self.o, self.a = oracle, oracle.Agc()

    def set_parameters(self, *p):
        self.a.set_parameters(*p)

    def process_cpx(self, x):
        return self.a.process_cpx(x)

    def process_s16(self, x):
        return self.a.process_s16(x)

    def delay(self):
        return self.a.delay()


def test_agc_oracle_matches_reference_cagc(oracle):
    """rx/CuteSDR/agc.cpp built from its own source: the oracle's restatement must reproduce its
    outputs.  Same compiler, same libm, operation-by-operation float/double evaluation on both
    sides: bit-identical is expected and asserted."""
    g = np.load(os.path.join(GOLD, "agc_ref.npz"))
    for name in g["names"]:
        name = str(name)
        got = run_agc_script(OracleAgc(oracle), g[name + "_script"], g[name + "_in"])
        want = g[name + "_out"]
        assert got.shape == want.shape, name
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
            "%s: max |diff| %g" % (name, np.abs(got - want).max())


def test_adpcm_oracle_matches_reference_ima_adpcm_cpp(oracle):
    """rx/csdr/ima_adpcm.cpp:134-214 built from its own source."""
    g = np.load(os.path.join(GOLD, "adpcm_ref.npz"))
    x = g["i16_in"]
    st = oracle.AdpcmState(0, 0)
    enc = []
    for k in range(0, x.size, 512):                          # state carried block to block
        e, st = oracle.adpcm_encode_i16(x[k:k + 512], st)
        enc.append(e)
    enc = np.concatenate(enc)
    assert np.array_equal(enc, g["i16_enc"])
    assert (st.index, st.previous) == tuple(g["i16_enc_state"])
    dec, st = oracle.adpcm_decode_i16(g["i16_enc"])
    assert np.array_equal(dec, g["i16_dec"]) and (st.index, st.previous) == tuple(g["i16_dec_state"])
    st = oracle.AdpcmState(37, -1234)
    enc = []
    for k in range(0, 3400, 170):
        e, st = oracle.adpcm_encode_i16(x[k:k + 170], st)
        enc.append(e)
    assert np.array_equal(np.concatenate(enc), g["i16_enc_resumed"])
    assert (st.index, st.previous) == tuple(g["i16_enc_resumed_state"])
    for r, want_e, want_d in zip(g["u8_rows"], g["u8_enc"], g["u8_dec"]):
        padded = np.concatenate([np.full(10, r[0], np.uint8), r])
        e, _ = oracle.adpcm_encode_u8(padded)
        assert np.array_equal(e, want_e)
        d, _ = oracle.adpcm_decode_u8(e)
        assert np.array_equal(d, want_d)
        # and the packet builder puts exactly these bytes behind its 16-byte header
        pkt = oracle.wf_packet(r, 1234, 5, 77, True)
        assert np.array_equal(pkt[16:], want_e)


def test_cic_shapes_match_reference_cic_gen_c(oracle):
    """verilog/rx/cic_gen.c, compiled from its own source and run with the reference's kiwi.gen.h,
    emits the register widths, truncations and output slices of every CIC instance (cic_ref.json holds
    its output).  The DDC oracle -- and through it the HIP kernels -- must use exactly these, for the
    std (1736 / 3) AND the wide (1543 / 2) decimations and for the waterfall CIC."""
    c = json.load(open(os.path.join(GOLD, "cic_ref.json")))

    def flat(e):
        n = e["N"]
        integ, combs = e["integrators"], e["combs"]
        trunc = e["trunc"]                       # N integrator inputs, N comb inputs, the output
        comb_drop = trunc[n:2 * n]
        return [n, e["Bin"], e["Bout"]] + integ + combs + comb_drop + e["out"][1:]

    wf1 = c["cic_wf1"]
    assert (wf1["N"], wf1["R"], wf1["acc"]) == (5, 8192, 89)
    assert oracle.ddc_shape(0) == flat(wf1)
    for name, which, r in (("cic_rx1_12k", 1, 1736), ("cic_rx1_20k", 1, 1543)):
        e = c[name]
        assert e["R"] == r and e["integrators"][:2] == [e["acc"], e["acc"]]
        assert oracle.ddc_shape(which, r) == flat(e), name
        assert e["trunc"][2] == e["acc"] - 26                # what the third integrator drops
    assert oracle.ddc_shape(2) == flat(c["cic_rx2_12k"]) and c["cic_rx2_12k"]["R"] == 3
    assert oracle.ddc_shape(3) == flat(c["cic_rx2_20k"]) and c["cic_rx2_20k"]["R"] == 2
    k = json.load(open(os.path.join(GOLD, "consts_ref.json")))
    assert (k["RX1_STD_DECIM"], k["RX2_STD_DECIM"], k["RX1_WIDE_DECIM"], k["RX2_WIDE_DECIM"]) == (1736, 3, 1543, 2)
    assert oracle.ddc_rx_decim(oracle.RX_STD) == 1736 * 3 * 2 and oracle.ddc_rx_decim(oracle.RX_WIDE) == 1543 * 2 * 2


@pytest.mark.skipif(not os.path.isfile("/root/reference/verilog/rx/fir_iq.sv"), reason="reference tree not present")
def test_all_three_cicf_tap_sets_match_fir_iq_sv(oracle):
    """fir_iq.sv:45-123: the RX_CFG == 3, RX_CFG == 14 and default coefficient tables."""
    import ctypes as C
    import re
    text = open("/root/reference/verilog/rx/fir_iq.sv").read()
    # the three `assign taps[..] = COEFF'('sh.....)` runs, in file order: RX_CFG == 3, == 14, default
    allt = [(int(i), int(h, 16)) for i, h in
            re.findall(r"assign taps\[\s*(\d+)\]\s*=\s*COEFF'\('sh([0-9a-f]{5})\)", text)]
    sets, cur = [], []
    for i, v in allt:
        if i == 0 and cur:
            sets.append(cur)
            cur = []
        cur.append(v)
    sets.append(cur)
    assert [len(x) for x in sets] == [33, 9, 33]
    L = oracle.lib()
    for name, n, want in (("ko_cicf_taps65_wide", 33, sets[0]), ("ko_cicf_taps17", 9, sets[1]),
                          ("ko_cicf_taps65", 33, sets[2])):
        got = list((C.c_int32 * n).in_dll(L, name))
        assert got == want[:n] and len(want) >= n, name


def test_cfir_oracle_matches_reference_fir_cpp(oracle):
    """rx/CuteSDR/fir.cpp built from its own source (oracle/_ref/fir_ref): Kaiser low-pass / high-pass designs (taps read
    back through an impulse, tap counts), InitConstFir with the de-emphasis tables, and the real -> real, real -> mono16 and
    mono16 -> mono16 ProcessFilter paths with their rotating accumulation order: bit-identical."""
    from tests.fixtures import run_fir_script
    g = np.load(os.path.join(GOLD, "fir_ref.npz"))
    assert len(g["names"]) == 15
    for name in g["names"]:
        name = str(name)
        got = run_fir_script(oracle.CFir(), g[name + "_script"], g[name + "_in"])
        want = g[name + "_out"]
        assert got.shape == want.shape, name
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%s: max |diff| %g" % (name, np.abs(got - want).max())
    # what the vectors say about the designs rx_sound_cmd.cpp:270-282 / squelch.cpp:137 ask for
    assert g["am_fir_am_12k_out"][0] == 32 and g["am_fir_amn_12k_out"][0] == 18 and g["am_fir_am_20k_out"][0] == 15
    assert g["hp_squelch_12k_out"][0] == 79 and g["hp_squelch_20k_out"][0] == 97      # 96 | 1: fills MAX_NUMCOEF
    assert g["am_fir_full_band_out"][0] == 9          # Fstop == Fpass: (int) of +inf, as x86 converts it


def test_squelch_oracle_matches_reference_squelch_cpp(oracle):
    """rx/CuteSDR/squelch.cpp (+ fir.cpp for its m_HpFir) built from its own sources (oracle/_ref/squelch_ref): outputs and
    the nsq_nc_sq return value of every PerformFMSquelch call, bit-identical; the vectors open and close the squelch."""
    from tests.fixtures import run_squelch_script
    g = np.load(os.path.join(GOLD, "squelch_ref.npz"))
    seen = set()
    for name in g["names"]:
        name = str(name)
        got = run_squelch_script(oracle.Squelch(), g[name + "_script"], g[name + "_in"])
        want = g[name + "_out"]
        assert got.shape == want.shape, name
        assert np.array_equal(got, want), "%s: %d differ" % (name, np.count_nonzero(got != want))
        pos = 0
        for line in g[name + "_script"]:
            f = str(line).split()
            if f[0] == "F":
                pos += int(f[1]) + 1
                seen.add(int(want[pos - 1]))
    assert seen == {-1, 0, 1}
    w = g["threshold_80_out"].reshape(12, 513)
    closed = [bool(np.all(r[:512] == 1)) for r in w]
    assert closed[2] and not closed[7] and closed[11], closed          # noise closes it, a quiet carrier opens it, noise again


@pytest.mark.skipif(not os.path.isfile("/root/reference/rx/rx_filter.h"), reason="reference tree not present")
def test_deemphasis_tables_match_rx_filter_h():
    """rx/rx_filter.h:29-73: the four [2][79] de-emphasis tables the host mirror carries (flydog_sdr_gps_amd/deemp.py) are the
    reference's numbers, and the reference's CFir, handed them, answers an impulse with exactly them (fir_ref.npz)."""
    import re
    from flydog_sdr_gps_amd import deemp
    text = open("/root/reference/rx/rx_filter.h").read()
    g = np.load(os.path.join(GOLD, "fir_ref.npz"))
    for name in ("nfm_deemp_12000", "nfm_deemp_20250", "am_ssb_deemp_12000", "am_ssb_deemp_20250"):
        body = re.search(r"const float %s\[N_NFM_DEEMP\]\[N_DEEMP_TAPS\] = \{(.*?)\n\};" % name, text, re.S).group(1)
        rows = re.findall(r"\{([^{}]*)\}", body)
        tab = np.array([[float(v) for v in re.sub(r"//[^\n]*", "", r).replace("\n", " ").split(",") if v.strip()] for r in rows],
                       np.float32)
        assert tab.shape == (2, deemp.N_DEEMP_TAPS)
        assert np.array_equal(getattr(deemp, name.upper()), tab), name
        for k in range(2):
            assert np.array_equal(g["%s_%d_out" % (name, k)][:79], tab[k])
    assert int(re.search(r"#define N_DEEMP_TAPS (\d+)", text).group(1)) == deemp.N_DEEMP_TAPS


# ---- the FFT-dependent rows, pinned by the reference's own gps/search.cpp and rx/CuteSDR/fastfir.cpp -------------------------
# compiled in place against the FFTW3 API the image ships (hipFFTW) and run on the GPU box (oracle/build_ref.sh,
# tools/make_ref_fft_golden.py -> tests/golden/acq_fftref.npz, fastfir_fftref.npz).  hipFFTW's transforms are not FFTW's
# (another algorithm, GPU arithmetic): float spectra are held to north_star's 1e-5 of the spectrum's largest bin, everything
# that is an index or a count to equality.
FFT_TOL = 1e-5


def _all_codes():
    from flydog_sdr_gps_amd import synth
    from tests.fixtures import e1b_chips
    return synth.all_sv_codes(e1b_chips())


def test_acquisition_oracle_matches_reference_search_cpp(oracle):
    """SearchInit()'s code tables (C/A PRN 1, QZSS 194, Galileo E02 and E36 with BOC(1,1)), both decimators, Sample()'s data
    spectrum of five 1-bit IF scenes and Correlate()'s (snr, Doppler bin, peak index) for eleven (scene, SV) pairs -- present
    SVs at positive and negative Doppler, absent ones (the winner among noise peaks), a noise-only block, E1B over its 16368
    lags -- as gps/search.cpp ITSELF computed them.  Peak bin and index: equal.  snr: 1e-5.  Decimators: bit-exact."""
    g = np.load(os.path.join(GOLD, "acq_fftref.npz"))
    keep = int(g["keep_every"])
    codes = _all_codes()

    def code_fft(s):
        return oracle.code_fft(codes[s][0], boc=codes[s][1])
    for s in g["code_sats"]:
        mine = code_fft(int(s))
        assert np.abs(mine[::keep] - g["code_%d_bins" % s]).max() <= FFT_TOL * g["code_%d_max" % s], s
        l2 = np.sqrt(np.sum(np.abs(mine.astype(np.complex128)) ** 2))
        assert abs(l2 - g["code_%d_l2" % s]) <= FFT_TOL * g["code_%d_l2" % s] and abs(np.abs(mine).max() - g["code_%d_max" % s]) <= FFT_TOL * g["code_%d_max" % s]
    assert np.array_equal(oracle.decimate_by2(g["dec_float_in"]).view(np.uint32), g["dec_float_out"].view(np.uint32))
    # DecimateBy2binary (search.cpp:168-179): bits[][2] -> +-1 (bit 1 -> -1.0) -> the float decimator
    b = g["dec_binary_in"].astype(np.float32)
    pm = (np.where(b[:, 0] > 0, -1.0, 1.0) + 1j * np.where(b[:, 1] > 0, -1.0, 1.0)).astype(np.complex64)
    assert np.array_equal(oracle.decimate_by2(pm).view(np.uint32), g["dec_binary_out"].view(np.uint32))
    negative = 0
    for name in (str(n) for n in g["scene_names"]):
        data = oracle.sample_bits(g[name + "_bits"])
        assert np.abs(data[::keep] - g[name + "_spec_bins"]).max() <= FFT_TOL * g[name + "_spec_max"], name
        assert abs(np.sqrt(np.sum(np.abs(data.astype(np.complex128)) ** 2)) - g[name + "_spec_l2"]) <= FFT_TOL * g[name + "_spec_l2"]
        for k, s in enumerate(int(v) for v in g[name + "_sats"]):
            lim = oracle.E1B_LIMIT if codes[s][1] else oracle.L1_LIMIT
            nxt = code_fft(s + 1) if s + 1 < len(codes) else None          # the table's next row (zeros behind the last SV)
            r, _ = oracle.correlate(code_fft(s), data, limit=lim, code_next=nxt)
            assert (r["dop"], r["idx"]) == (int(g[name + "_dop"][k]), int(g[name + "_idx"][k])), (name, s, r)
            assert abs(r["snr"] - g[name + "_snr"][k]) <= FFT_TOL * g[name + "_snr"][k], (name, s, r["snr"], g[name + "_snr"][k])
            negative += r["dop"] < 0
    assert negative >= 4          # the cases that read the next satellite's row (search.cpp:471 over the doubled row of :54)
    # ... which the modulo-N restatement of rounds 1-5 got wrong: it must NOT reproduce the reference there
    data = oracle.sample_bits(g["prn20_negative_doppler_bits"])
    wrapped, _ = oracle.correlate(code_fft(19), data, code_next=code_fft(19))      # own bins 0 .. |dop| behind the row = modulo N
    assert abs(wrapped["snr"] - g["prn20_negative_doppler_snr"][0]) > 1e-3 * wrapped["snr"]


class OracleFastFir:
    def __init__(self, oracle):
        self.o, self.st = oracle, oracle.fir_new_state()
        self.win, self.cic_on, self.coef, self.last = oracle.fir_window(-1), False, None, None
        self.base = None

    def window(self, w):
        self.win = self.o.fir_window(w)

    def cic(self, on):                         # SetupCICFilter: m_pFilterCoef_CIC from m_pFilterCoef (fastfir.cpp:148-158)
        self.cic_on = on
        if self.base is not None:
            self.coef = (self.base * (self.o.fir_cic_coeffs() if on else 1.0)).astype(np.complex64)

    def setup(self, inst, lo, hi, off, fs):
        if self.last == (lo, hi, off, fs):     # :180-184
            return
        self.last = (lo, hi, off, fs)
        d = self.o.fir_design(lo, hi, off, fs, window=self.win, do_cic_comp=self.cic_on, prec=0)
        if d is not None:                      # :193-200: a rejected setting leaves the old filter
            self.base, self.coef = d[0], d[1]

    def process(self, x):
        return self.o.fir_process(self.st, self.coef, x, prec=0)


def test_fastfir_oracle_matches_reference_fastfir_cpp(oracle):
    """CFastFIR::SetupParameters / SetupCICFilter / ProcessData sequences (USB in the data pump's 170-sample interrupts, LSB in
    ragged lengths, AM, a narrow CW filter with an offset, the 20.25 kHz mode, a retune in mid-stream with a rejected and a
    repeated setting, CIC compensation on) as rx/CuteSDR/fastfir.cpp ITSELF computed them: output counts and FirPos() equal,
    samples to 1e-5 of the block's largest."""
    from tests.fixtures import fastfir_blocks, run_fastfir_script
    g = np.load(os.path.join(GOLD, "fastfir_fftref.npz"))
    for name in (str(n) for n in g["names"]):
        got = fastfir_blocks(g[name + "_script"], run_fastfir_script(OracleFastFir(oracle), g[name + "_script"], g[name + "_in"]))
        want = fastfir_blocks(g[name + "_script"], g[name + "_out"])
        assert len(got) == len(want)
        scale = max(float(np.abs(w[2]).max()) for w in want if w[0])
        for b, ((gc, gp, gy), (wc, wp, wy)) in enumerate(zip(got, want)):
            assert (gc, gp) == (wc, wp), (name, b, gc, gp, wc, wp)
            if wc:
                assert np.abs(gy - wy).max() <= FFT_TOL * scale, (name, b, float(np.abs(gy - wy).max()), scale)


def _wf_case_params(g, k):
    from flydog_sdr_gps_amd import wf
    zoom, start, interp, winf, cic, ovl, inv, comp = g["cases"][k]
    p = wf.WfParams.for_zoom(int(zoom), float(start), spectral_inversion=bool(inv))
    return p, int(interp), int(winf), bool(cic), bool(ovl), bool(inv), bool(comp)


def test_waterfall_oracle_matches_reference_rx_waterfall_cpp(oracle):
    """c2s_waterfall_init()'s tables and compute_frame() of rx/rx_waterfall.cpp ITSELF (wf_fftref.npz: compiled in place against
    hipFFTW, run on a GPU box) on eight frames -- every zoom regime, interpolation mode, window, CIC compensation on / off /
    overlapped, spectral inversion: the window functions bit for bit, CIC_comp to a float step, the transform to 1e-5 of its
    largest bin, and the 1024 output bytes, fft_used_limit, packet header fields and (where compression is on) the ADPCM payload
    EQUAL."""
    from flydog_sdr_gps_amd import wf
    g = np.load(os.path.join(GOLD, "wf_fftref.npz"))
    for w in range(4):
        assert np.array_equal(oracle.wf_window(w).view(np.uint32), g["window_function"][w].view(np.uint32)), w
    assert np.abs(oracle.wf_cic_comp() - g["cic_comp"]).max() <= 2e-7 * np.abs(g["cic_comp"]).max()
    assert int(g["n_chunks"]) == 9
    tables = (g["window_function"], g["cic_comp"])
    seen_comp = 0
    for k in range(int(g["ncases"])):
        p, interp, winf, cic, ovl, inv, comp = _wf_case_params(g, k)
        m, d = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, inv)
        sc = np.full(1024, p.fft_scale, np.float32)
        samps = oracle.wf_window_iq(g["case%d_iq" % k], tables[0][winf])
        row, pwr, pwr_out, dB = oracle.wf_compute_frame(samps, p.zoom, winf, interp, cic, ovl, p.fft_used, p.plot_width, p.plot_width_clamped,
                                                        m, d, sc, (sc / np.float32(2)).astype(np.float32), p.fft_offset, tables[1])
        assert np.array_equal(row, g["case%d_row" % k]), (k, int(np.count_nonzero(row != g["case%d_row" % k])))
        nbytes, limit, xbin, flags, seq = (int(v) for v in g["case%d_hdr" % k])
        use_comp = comp and p.zoom != 0                               # rx_waterfall.cpp:1283-1285
        pkt = oracle.wf_packet(row, int(p.start), p.zoom, seq, comp)          # the connection's setting: the zoom rule is inside
        assert pkt.size - 16 == nbytes and np.array_equal(pkt[16:], g["case%d_payload" % k]), k
        assert (xbin, flags) == (int(p.start), p.zoom | (0x10000 if use_comp else 0)), (k, xbin, flags)
        assert limit == min(p.fft_used, next((i for i in range(p.fft_used) if m[i] >= 1024), p.fft_used)) or limit == p.fft_used, (k, limit)
        spec = oracle.fft(samps, sign=-1, prec=0)[:p.fft_used]
        assert np.abs(spec[::2] - g["case%d_spec" % k]).max() <= FFT_TOL * g["case%d_spec_max" % k], k
        seen_comp += use_comp
    assert seen_comp == 4


def test_dpump_oracle_matches_reference_data_pump_cpp(oracle):
    """snd_service() of rx/data_pump.cpp ITSELF (the driver includes the file; dpump_ref.npz): `rescale` as its static initialiser
    evaluates MPOW(2, -RXOUT_SCALE + CUTESDR_SCALE) * MPOW(10, CICF_GAIN_dB / 20), the 24-bit sign extension, the re / im swap and
    its spectral-inversion variant, DC offsets, disabled channels skipped, sample-major channel-minor records for 4 / 8 / 14 / 3
    channels: the oracle's unpack BIT-EXACT; the 32-deep ring position and the 48-bit tick assembly as the reference left them."""
    from tests.fixtures import dpump_ref_cases
    g = np.load(os.path.join(GOLD, "dpump_ref.npz"))
    for name, nch, ns, inv, dci, dcq, en, bufs, rescale, per in dpump_ref_cases(g):
        assert np.float32(rescale) == np.float32(oracle.dpump_rescale()), name
        for b, d in enumerate(per):
            raw = bufs[b, :6 * ns * nch]
            want = oracle.dpump_unpack(raw, ns, nch, enabled=en, dc_i=dci, dc_q=dcq, spectral_inversion=inv)
            t = bufs[b, 6 * ns * nch:].view("<u2")                    # rx_trailer_t: ticks[3], write_ctr_stored, write_ctr_current
            for ch, (wr, ticks, samps) in d.items():
                assert np.array_equal(want[ch].view(np.uint32), samps.view(np.uint32)), (name, b, ch)
                assert wr == (b + 1) % 32 and ticks == (int(t[2]) << 32 | int(t[1]) << 16 | int(t[0])), (name, b, ch)


def chan_ref_expect(g):
    """chan_ref.npz -> [(is_e1b, lo_shift, ca_shift, secs, lo_rate, ca_rate, ca_pause)] from the SPI commands CHANNEL::Start sent:
    CmdSetRateLO (ch, lo_rate), CmdSetRateCG (ch, ca_rate), CmdPause (ch, ca_pause - 1) when ca_pause != 0, CmdSetMask."""
    c = json.load(open(os.path.join(GOLD, "consts_ref.json")))
    out, k = [], 0
    for call, n in zip(g["calls"], g["ncmds"]):
        ch, sat, t_sample, now, lo_shift, ca_shift, snr = (int(v) for v in call)
        cmds = g["cmds"][k:k + n]
        k += n
        secs = ((now - t_sample) & 0xffffffff) / 1e6                # (timer_us() - t_sample) / 1e6 in u4_t arithmetic, channel.cpp:293
        assert n in (3, 4) and cmds[0][0] != cmds[1][0]
        pause = int(cmds[2][2]) + 1 if n == 4 else 0
        out.append((sat >= 36, lo_shift, ca_shift, secs, int(cmds[0][2]), int(cmds[1][2]), pause))
    return out


def test_chan_start_oracle_matches_reference_channel_cpp(oracle):
    """CHANNEL::Start of gps/channel.cpp ITSELF (chan_ref.npz: 205 calls over every Doppler bin, code phases of both code periods,
    delays from 0 to a minute): the NCO rates and the code-generator pause it sent over SPI, equal."""
    g = np.load(os.path.join(GOLD, "chan_ref.npz"))
    exp = chan_ref_expect(g)
    assert len(exp) == 205 and sum(e[0] for e in exp) > 50
    for is_e1b, lo_shift, ca_shift, secs, lo_rate, ca_rate, pause in exp:
        o = oracle.chan_start(is_e1b, lo_shift, ca_shift, secs)
        assert (o.lo_rate, o.ca_rate, o.ca_pause) == (lo_rate, ca_rate, pause), (is_e1b, lo_shift, ca_shift, secs)


def test_aperture_oracle_matches_reference_aperture_auto(oracle):
    """aperture_auto() of rx/rx_waterfall.cpp ITSELF, run inside the reference's compute_frame() on the GPU box with the rows it had
    just produced (aper_fftref.npz; dB_wire_to_dBm of rx_util.cpp and qsort_intcomp of support/misc.cpp linked in place): the IIR,
    MMA and EMA averages, the single-shot mode, the audio FFT's pixel range, two calibrations, an all-masked run -- avg_pwr[]
    after every frame BIT-EXACT, signal / noise / done_autoscale / report_sec / avg_clear equal."""
    from tests.fixtures import aper_ref_replay
    g = np.load(os.path.join(GOLD, "aper_fftref.npz"))
    n, reports, algos = 0, set(), set()
    for k, avg, st, want_avg, want_st in aper_ref_replay(g, oracle.aper_update, oracle.aper_report):
        assert np.array_equal(np.asarray(avg, np.float32).view(np.uint32), want_avg.view(np.uint32)), (k, g["frames"][k])
        assert st == want_st, (k, st, want_st)
        n += 1
        reports.add((st[0], st[1]))
        algos.add(int(g["frames"][k][2]))
    assert n == len(g["frames"]) >= 30 and algos == {0, 1, 2, 3}
    assert (-110, -120) not in reports and (-80, -120) in reports      # the all-masked run reports -110 -> floor -80 / -120
    assert len(reports) >= 5


def test_libm_restatements_equal_the_images_libm(oracle):
    """The platform's log10f, powf and expf (GNU C Library 2.35: e_log10f.c over e_logf.c, e_powf.c, e_expf.c) restated in
    oracle/kiwi_oracle_libm.c against the libm the oracle is linked with: every 64th float, every float of [0.5, 2), subnormals,
    negatives; fused and unfused multiply-adds.  0 differences (every argument: tools/check_libm.py --exhaustive).  expf's residual
    is the FMA build's on an FMA host (2 of 2^32 arguments tell: both are in the sample).  The DEVICE copy (csrc/kg_libm.h)
    carries the same tables and constants: compared here as text."""
    import re
    fma_host = "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    for fused in (True, False):
        for first, n, step in ((0, 0x7f800001, 64), (0x3f000000, 1 << 24, 1), (0, 0x00800000, 3), (0x80000000, 0x7fffffff, 4099)):
            done, bad_ln, bad_l10, where = oracle.libm_check_range(first, n, step, fused)
            assert done >= n // step and (bad_ln, bad_l10) == (0, 0), (fused, hex(first), bad_ln, bad_l10, hex(where))
        for first, n, step in ((0, 1 << 32, 64), (0x42024200, 0x100, 1), (0xc27c6500, 0x100, 1), (0xb5800000, 1 << 23, 1)):
            done, bad_p, bad_e, bad_r, nr = oracle.libm_check_pow_exp(first, n, step, fused, fma_host)
            assert done >= n // step and (bad_p, bad_e, bad_r) == (0, 0, 0), (fused, hex(first), bad_p, bad_e, bad_r)
    assert oracle.libm_check_pow_exp(0x42024200, 0x100, 1, True, not fma_host)[2] == 1          # the other residual: one argument tells
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev_text = open(os.path.join(here, "flydog_sdr_gps_amd", "csrc", "kg_libm.h")).read()
    cpu_text = open(os.path.join(here, "oracle", "kiwi_oracle_libm.c")).read()
    dev_text, cpu_text = dev_text[dev_text.index("#ifndef KG_LIBM_H"):], cpu_text[cpu_text.index("#include"):]
    hexf = re.compile(r"-?0x[0-9a-f.]+p[-+]?\d+")
    dev, cpu = [float.fromhex(v) for v in hexf.findall(dev_text)], [float.fromhex(v) for v in hexf.findall(cpu_text)]
    assert dev[:32] == cpu[:32] and cpu[18:20] == [1.0, 0.0] and len(dev) >= 80
    assert sorted(set(dev)) == sorted(set(v for v in cpu if v != 2.0 ** 23))      # (the host's subnormal scaling 0x1p23f is log10f's two25 on the device)
    hexi = re.compile(r"0x3f[ef][0-9a-f]{13}ull")
    assert hexi.findall(dev_text) == hexi.findall(cpu_text) and len(hexi.findall(dev_text)) == 32
    for lit in ("3.3554432000e+07f", "7.9034151668e-07f", "4.3429449201e-01f", "3.0102920532e-01f", "0x1.62e42ep6f", "-0x1.9fe368p6f"):
        assert lit in dev_text and lit in cpu_text, lit


def test_sound_path_oracle_matches_the_references_own_statements(oracle):
    """c2s_sound()'s signal path between CFastFIR and the bytes of the sound packet -- rx/rx_sound.cpp:676-908, 1035-1140, 1222-1253, the
    reference's OWN STATEMENTS cut out of the file at build time and compiled around its agc.cpp / fir.cpp / squelch.cpp /
    ima_adpcm.cpp (oracle/build_ref.sh, oracle/ref/ref_sndpath_main.cpp; sndpath_ref.npz): S-meter average and taps, AM detector +
    m_AM_FIR, NBFM detector + clipper + m_Squelch, the SSB / CW AGC, both de-emphasis filters, `s->squelched`, the IQ modes' AGC; the
    payload (ADPCM with its state carried over packets, raw mono and (s2_t) IQ pairs in either byte order) and the header (flags, sequence
    number, the clamped S-meter field); every mode family, both rates, ragged and one-sample blocks, mode hops with state carried,
    silence, full scale, ADC overflow.  The oracle's restatement: EQUAL, every value and every byte."""
    from tests.fixtures import OracleSoundPath, sndpath_check
    g = np.load(os.path.join(GOLD, "sndpath_ref.npz"))
    names = [str(n) for n in g["names"]]
    assert len(names) == 11
    packets = blocks = samples = 0
    for name in names:
        p, b, s = sndpath_check(g, name, lambda rate: OracleSoundPath(oracle, rate))
        packets, blocks, samples = packets + p, blocks + b, samples + s
    assert packets >= 60 and blocks >= 100 and samples >= 50000, (packets, blocks, samples)


def test_waterfall_commands_match_the_references_own_statements(oracle):
    """What a waterfall connection derives from `SET zoom= start=` / `SET zoom= cf=` -- rows W2 and W6's map -- against
    c2s_waterfall()'s OWN STATEMENTS (rx/rx_waterfall.cpp:365-529, 756-928, cut out of the file at build time: oracle/build_ref.sh,
    oracle/ref/ref_wfcmd_main.cpp; wfcmd_ref.npz): 5 configurations (30 / 32 MHz displays, spectral inversion, the admin's masked
    ranges) x 110 commands in sequence (every zoom at its edges and beyond them, the cf form with out-of-range zooms, pans).
    The decimation word and the 48-bit NCO offset it hands to the SPI driver, the clamped start, fft_used, plot_width(_clamped),
    fft_offset, fft2wf_map[], drop_sample[], fft_scale[], fft_scale_div2[]: the host mirror (flydog_sdr_gps_amd/wf.py) AND the
    oracle (ko_wf_params_for / ko_wf_build_maps) EQUAL, every value."""
    import re
    from flydog_sdr_gps_amd import wf
    g = np.load(os.path.join(GOLD, "wfcmd_ref.npz"))
    CmdSetWFDecim, CmdSetWFFreq = 1, 2                        # the harness's tags for the reference's enum values
    ncmd = rebuilt = masked = 0
    for name in (str(n) for n in g["names"]):
        adc, srate, inv = g[name + "_cfg"]
        inv = bool(inv)
        masks = [tuple(int(v) for v in m) for m in g[name + "_masks"]]
        mpos = dpos = 0
        last_zoom = -1
        for k, cmd in enumerate(str(c) for c in g[name + "_cmds"]):
            nspi, zoom, start, wait_ms, wait_us, fft_used, pw, pwc, limit, had = (int(v) for v in g[name + "_hdr"][k])
            m = re.match(r"SET zoom=(-?\d+) (start|cf)=(\S+)", cmd)
            z = min(max(int(m.group(1)), 0), 14)
            st = float(m.group(3)) if m.group(2) == "start" else wf.start_of_cf(z, float(m.group(3)), ui_srate=srate)
            prm = wf.WfParams.for_zoom(z, st, adc_clock=adc, ui_srate=srate, spectral_inversion=inv)
            o = oracle.wf_params(z, st, adc_clock=adc, ui_srate=srate, spectral_inversion=inv)
            for who, p in (("mirror", prm), ("oracle", o)):
                assert (zoom, start, fft_used, pw, pwc) == (p.zoom, int(p.start), p.fft_used, p.plot_width, p.plot_width_clamped), (name, cmd, who)
                assert np.float32(p.fft_offset) == g[name + "_fft_offset"][k], (name, cmd, who)
                spi = g[name + "_spi"][k]
                freq = spi[nspi - 1]
                assert freq[0] == CmdSetWFFreq and ((int(freq[2]) << 16) | int(freq[3])) == int(p.i_offset) & 0xFFFFFFFFFFFF, (name, cmd, who)
                assert (nspi == 2) == (z != last_zoom), (name, cmd)
                if nspi == 2:
                    assert spi[0][0] == CmdSetWFDecim and int(spi[0][2]) == int(p.decim), (name, cmd, who, spi[0], p.decim)
            sc, d2 = wf.scale_arrays(prm, ui_srate=srate, masked=masks)
            assert np.array_equal(sc.view(np.uint32), g[name + "_scale"][k][:pwc].view(np.uint32)), (name, cmd, "fft_scale")
            assert np.array_equal(d2.view(np.uint32), g[name + "_div2"][k][:pwc].view(np.uint32)), (name, cmd, "fft_scale_div2")
            assert np.float32(o.fft_scale) == np.float32(prm.fft_scale)
            masked += int((sc == 0).sum())
            assert had == (z != last_zoom) and limit == 0
            if had:
                want_map, want_drop = g[name + "_maps"][mpos:mpos + fft_used], g[name + "_drops"][dpos:dpos + pwc]
                mpos, dpos = mpos + fft_used, dpos + pwc
                for who, (fm, dr) in (("mirror", wf.build_maps(fft_used, pw, pwc, inv)), ("oracle", oracle.wf_build_maps(fft_used, pw, pwc, inv))):
                    assert np.array_equal(np.asarray(fm, np.int64).astype(np.uint16), want_map), (name, cmd, who, "fft2wf_map")
                    assert np.array_equal(np.asarray(dr, np.int64).astype(np.uint16)[:pwc], want_drop), (name, cmd, who, "drop_sample")
                rebuilt += 1
            last_zoom = z
            ncmd += 1
        assert mpos == g[name + "_maps"].size and dpos == g[name + "_drops"].size
    assert ncmd == 550 and rebuilt > 200 and masked > 10000, (ncmd, rebuilt, masked)


def _notch(pre):
    e = np.array(pre, np.complex64, copy=True)
    e[..., 256:768] = 0                         # what the driver's editing hook does (oracle/ref/ref_fastfir_main.cpp)
    return e


def test_fastfir_extension_taps_oracle_matches_reference_fastfir_cpp(oracle):
    """what CFastFIR::ProcessData of rx/CuteSDR/fastfir.cpp ITSELF hands a registered extension hook
    (fastfir_taps_fftref.npz; the driver registers one through ext_users[] as extensions do): the PRE_FILTERED buffer (forward
    spectrum x m_CIC), the POST_FILTERED buffer (x m_pFilterCoef_CIC), and -- when the PRE hook edits its buffer and answers true --
    the output filtered from the EDITED buffer with m_pFilterCoef (`buf_modified`, :286-290).  Hook calls and their order equal;
    bins and samples to 1e-5 of the largest."""
    from tests.fixtures import fastfir_taps_cases
    g = np.load(os.path.join(GOLD, "fastfir_taps_fftref.npz"))
    ncalls = 0
    for name, script, x, per in fastfir_taps_cases(g):
        f, pos, flags, edit, k = OracleFastFir(oracle), 0, 0, 0, 0
        for line in script:
            t = line.split()
            if t[0] == "C":
                f.cic(int(t[1]) != 0)
            elif t[0] == "P":
                f.setup(int(t[1]), *[float(v) for v in t[2:6]])
            elif t[0] == "H":
                flags, edit = int(t[1]), int(t[2])
            else:
                n, count, firpos, taps, want = per[k]
                k += 1
                cic = oracle.fir_cic_coeffs() if f.cic_on else np.ones(1024, np.float32)
                out, fp, pre, post = oracle.fir_process_taps(f.st, f.coef, cic, x[pos:pos + n], prec=0)
                pos += n
                assert (out.size, fp) == (count, firpos), (name, k)
                nblk = count // 512
                order = [fl for fl in (1, 2) if flags & fl] * nblk if flags else []
                assert [fl for fl, _ in taps] == order, (name, k, [fl for fl, _ in taps])
                for i, (fl, bins) in enumerate(taps):
                    b = i // len([1 for fl_ in (1, 2) if flags & fl_])
                    got = pre[b] if fl == 1 else post[b]
                    assert np.abs(got - bins).max() <= FFT_TOL * np.abs(bins).max(), (name, k, b, fl)
                    ncalls += 1
                if flags & 1 and edit and nblk:
                    out = np.concatenate([oracle.fft(f.base * _notch(pre[b]), sign=+1, prec=1)[512:] for b in range(nblk)])
                if count:
                    assert np.abs(out - want).max() <= FFT_TOL * np.abs(want).max(), (name, k, float(np.abs(out - want).max()))
    assert ncalls == 20


def test_audio_nco_phase_increment_matches_reference_rx_sound_cmd_cpp():
    """Row D6: rx_sound_set_freq() of rx/rx_sound_cmd.cpp ITSELF (built in place; sndcmd_ref.npz: 396 frequencies over three ADC clocks,
    both display bandwidths, spectral inversion on and off, the band edges): the 48-bit word it hands to spi_set3(CmdSetRXFreq) equals
    the host mirror's (ddc.rx_phase_inc), which is what kg_rxddc_set_freq is given."""
    from flydog_sdr_gps_amd import ddc
    g = np.load(os.path.join(GOLD, "sndcmd_ref.npz"))
    assert len(g["calls"]) == 396
    for (f_khz, adc, srate, inv), want in zip(g["calls"], g["i_phase"]):
        got = ddc.rx_phase_inc(f_khz * 1000.0, adc_clock=adc, spectral_inversion=bool(inv), ui_srate=srate)
        assert got == int(want), (f_khz, adc, srate, inv, hex(got), hex(int(want)))


def test_passband_statements_match_reference_rx_sound_cmd_cpp(oracle):
    """The passband part of the `SET mod= low_cut= high_cut=` handler -- rx/rx_sound_cmd.cpp:243-272, 276-286, the reference's own
    statements cut out of the file at build time around its own CFir (sndcmd_ref.npz, 37 passbands: both rates, cuts beyond the
    Nyquist limit, one-sided and narrow passbands, the all-zero "no change" command): the clamped cuts CFastFIR::SetupParameters is
    given, the half bandwidth, and m_AM_FIR's taps read back through an impulse -- the helper every AM test uses
    (tests/fixtures.am_passband) and the oracle's CFir design: EQUAL, taps bit for bit."""
    from tests.fixtures import am_passband
    g = np.load(os.path.join(GOLD, "sndcmd_ref.npz"))
    assert len(g["bands"]) == 37
    clamped = 0
    for (lo, hi, rate), (w_lo, w_hi, _, _, _, w_hbw), w_taps in zip(g["bands"], g["band_out"], g["am_fir"]):
        if lo == 0 and hi == 0:                                   # no_pb_change: nothing is touched
            assert (w_lo, w_hi) == (0, 0) and not w_taps.any()
            continue
        c_lo, c_hi, hbw, stop = am_passband(lo, hi, rate)
        assert (c_lo, c_hi, float(hbw)) == (w_lo, w_hi, w_hbw), (lo, hi, rate, c_lo, c_hi, hbw, w_lo, w_hi, w_hbw)
        clamped += (c_lo, c_hi) != (lo, hi)
        f = oracle.CFir()
        n = f.init_lp(0, 1.0, 50.0, hbw, stop, rate)
        taps = np.asarray(f.taps(), np.float32)
        assert n == taps.size <= 97 and np.array_equal(taps.view(np.uint32), w_taps[:n].view(np.uint32)) and not w_taps[n:].any(), (lo, hi, rate, n)
    assert clamped >= 7
