"""The reference's INTEGER outputs on this path do not depend on which correct FFT computes them.

Why this file exists: gps/search.cpp, rx/rx_waterfall.cpp and rx/CuteSDR/fastfir.cpp transform with FFTW3f
(gps/search.cpp:240-241,447,481; rx/rx_waterfall.cpp:133,1291), which is absent from this image.  (Round 6 pins those rows
with the reference's own files built against hipFFTW, the FFTW3 API the image does ship: DESIGN.md section 3.  hipFFTW is one
more transform that is not FFTW3f's, which is what this file is about.)  What can be bounded here is the risk a transform's
identity carries: the same restated reference loops (oracle/kiwi_oracle*.c) are run over THREE independent
transforms --

    prec=1  the oracle's double-precision radix-2, rounded to fp32 on store,
    prec=0  the oracle's fp32 radix-4 Stockham,
    prec=2  scipy.fft (pocketfft) complex64 through the oracle's FFT hook (the seat FFTW3f has in the reference),

and every integer the path hands on must be identical: the peak index of EVERY (SV, Doppler) cell, the winning
(Doppler bin, code phase) per SV -- BASELINE configs[0], [1] (all 1 312 cells) and [4] (all 15 104 cells + the 59
winners) -- and the u8 waterfall rows of the eight tests/test_wf_gpu.py cases (<= 2 one-LSB flips at (int) edges,
the same rule the GPU tests use).  The margins are reported: the smallest relative gap between a winner and its
runner-up, against the largest relative difference the three transforms produce in the same quantity.

`python -m tests.test_fft_invariance_cpu` prints the report committed as profiles/r04_fft_invariance.txt."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import WfParams, acq, prn, sats, synth, wf
from tests.fixtures import e1b_chips

PRECS = (1, 0, 2)
NAMES = {1: "oracle f64 radix-2", 0: "oracle f32 radix-4", 2: "scipy.fft complex64"}


def _hook(ko):
    import scipy.fft as sf
    ko.set_fft_hook(lambda x, sign: sf.fft(x) if sign < 0 else sf.ifft(x, norm="forward"))


def acq_three_ways(ko, codes, data_of, fft_len, dop_lo, dop_hi, nthreads=8):
    """-> {prec: (results[nsv], cells[nsv][ndop])}: SearchInit code tables, Sample() and Correlate() all with that FFT."""
    _hook(ko)
    limits = [ko.E1B_LIMIT if b else ko.L1_LIMIT for _, b in codes]
    out = {}
    for prec in PRECS:
        spectra = np.stack([ko.code_fft(c, boc=b, prec=prec, fft_len=fft_len) for c, b in codes])
        data = data_of(prec)
        out[prec] = ko.correlate_many(spectra, data, limits, dop_lo=dop_lo, dop_hi=dop_hi, prec=prec, nthreads=nthreads)
    ko.set_fft_hook(None)
    return out


def acq_margins(runs):
    """Per-SV winner: relative snr gap to the runner-up Doppler bin (what a different FFT would have to bridge to
    change `dop`), and the largest relative snr difference between the transforms over all cells."""
    ref_res, ref_cells = runs[1]
    snr = ref_cells["snr"].astype(np.float64)
    top2 = np.sort(snr, axis=1)[:, -2:]
    gap = (top2[:, 1] - top2[:, 0]) / top2[:, 1]
    spread = 0.0
    for prec in (0, 2):
        spread = max(spread, float(np.max(np.abs(runs[prec][1]["snr"].astype(np.float64) - snr) / snr)))
    return gap, spread


def cell_peak_margin(ko, code_spec, data_spec, limit, dop_lo, dop_hi):
    """Smallest relative gap, over the cells of one SV, between the cell's peak power and the runner-up lag (numpy
    restatement of gps/search.cpp:471-495 on the oracle's own spectra; double precision)."""
    N = data_spec.size
    d = np.conj(data_spec.astype(np.complex128))
    c = code_spec.astype(np.complex128)
    worst = 1.0
    for dop in range(dop_lo, dop_hi + 1):
        y = np.fft.ifft(d * np.roll(c, dop))[:limit] * N
        pw = y.real ** 2 + y.imag ** 2
        a, b = np.partition(pw, -2)[-2:]
        worst = min(worst, (b - a) / b)
    return worst


def assert_same_integers(runs, what):
    r1, c1 = runs[1]
    diffs = {}
    for prec in (0, 2):
        r, c = runs[prec]
        diffs[prec] = (int(np.count_nonzero(c["idx"] != c1["idx"])),
                       int(np.count_nonzero((r["dop"] != r1["dop"]) | (r["idx"] != r1["idx"]) | (r["valid"] != r1["valid"]))))
        assert diffs[prec] == (0, 0), "%s: %s differs from the f64 transform in %d cell indices, %d SV results" % (
            what, NAMES[prec], *diffs[prec])
        # float outputs: inside 2e-5 between FFT implementations (the GPU parity tests hold 1e-5 against the oracle)
        assert np.max(np.abs(r["snr"] - r1["snr"]) / r1["snr"]) < 2e-5
    return diffs


def config0(ko):
    _, t1, t2, _ = sats.SATS[0]
    codes = [(prn.cacode(t1, t2), False)]
    bits = synth.config0_bits()
    return acq_three_ways(ko, codes, lambda prec: ko.sample_bits(bits, prec=prec), acq.FFT_LEN, -20, 20, 1), codes


def config1(ko):
    codes = [(prn.cacode(sats.SATS[s][1], sats.SATS[s][2]), False) for s in range(32)]
    iq = synth.config1_iq16(seed=0x5EED0002)
    return acq_three_ways(ko, codes, lambda prec: ko.sample_iq16(iq, prec=prec), acq.FFT_LEN, -20, 20), (codes, iq)


def config4(ko):
    codes = synth.all_sv_codes(e1b_chips())
    iq = synth.config4_iq16(codes, seed=0x5EED0005)
    ns, N = acq.NSAMPLES_10MS, acq.FFT_LEN_10MS
    return acq_three_ways(ko, codes, lambda prec: ko.sample_iq16(iq, prec=prec, nsamples=ns, fft_len=N), N, -128, 127), codes


def test_config0_prn1_winner_is_fft_invariant(oracle):
    runs, _ = config0(oracle)
    assert_same_integers(runs, "configs[0]")
    assert runs[1][0][0]["valid"] == 1 and runs[1][0][0]["snr"] > acq.MIN_SIG


def test_config1_all_1312_cells_are_fft_invariant(oracle):
    runs, _ = config1(oracle)
    assert runs[1][1].shape == (32, 41)
    assert_same_integers(runs, "configs[1]")
    gap, spread = acq_margins(runs)
    # the runner-up Doppler bin of every SV is farther from the winner than the transforms are from each other
    assert gap.min() > 10 * spread, (gap.min(), spread)


def test_config4_all_15104_cells_and_59_winners_are_fft_invariant(oracle):
    runs, _ = config4(oracle)
    assert runs[1][1].shape == (59, 256)
    assert_same_integers(runs, "configs[4]")
    gap, spread = acq_margins(runs)
    assert gap.min() > 10 * spread, (gap.min(), spread)


# ---- waterfall rows ------------------------------------------------------------------------------------
from tests.test_wf_gpu import CASES, DB_EDGE, MAX_FLIPS          # noqa: E402  (the same eight cases, the same rule)


def wf_three_ways(ko, case, tables):
    zoom, start, interp, window_func, cic_comp, overlapped, inv = case
    p = WfParams.for_zoom(zoom, start, spectral_inversion=inv)
    iq = synth.wf_iq_frame(seed=1000 + zoom)
    m, d = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, inv)
    sc = np.full(1024, p.fft_scale, np.float32)
    samps = ko.wf_window_iq(iq, tables[0][window_func])
    _hook(ko)
    out = {prec: ko.wf_compute_frame(samps, p.zoom, window_func, interp, cic_comp, overlapped, p.fft_used, p.plot_width,
                                     p.plot_width_clamped, m, d, sc, (sc / np.float32(2)).astype(np.float32),
                                     p.fft_offset, tables[1], prec=prec) for prec in PRECS}
    ko.set_fft_hook(None)
    return out


@pytest.mark.parametrize("case", CASES, ids=lambda c: "z%d-i%d-w%d-c%d%d-inv%d" % (c[0], c[2], c[3], c[4], c[5], c[6]))
def test_waterfall_rows_are_fft_invariant(oracle, case):
    """u8 rows (rx/rx_waterfall.cpp:1489-1554): identical across the three transforms except one-LSB flips where the f64
    run's dB lies within the power-error bound of an (int) edge -- at most MAX_FLIPS per row among the strong pixels
    (within 45 dB of the row's maximum), the rule tests/test_wf_gpu.py applies to the GPU."""
    from tests.test_wf_gpu import db_bound
    tables = (wf.window_functions(), wf.cic_comp_table())
    runs = wf_three_ways(oracle, case, tables)
    w_out, w_pwr, w_pwr_out, w_dB = runs[1]
    tol = db_bound(w_pwr_out)
    strong = w_pwr_out > 10 ** -4.5 * w_pwr_out.max()
    for prec in (0, 2):
        out, pwr, pwr_out, dB = runs[prec]
        assert np.abs(pwr - w_pwr).max() <= 1e-5 * w_pwr.max()
        diff = out.astype(int) - w_out.astype(int)
        bad = np.nonzero(diff)[0]
        assert np.count_nonzero(diff[strong]) <= MAX_FLIPS, (NAMES[prec], bad[:10])
        clamped = np.clip(w_dB.astype(np.float64), -200.0, 0.0)
        for i in bad:
            assert abs(diff[i]) <= 1 + int(tol[i]), (i, out[i], w_out[i])
            assert abs(clamped[i] - np.rint(clamped[i])) < tol[i] + DB_EDGE, (i, w_dB[i], tol[i])


def report():
    from oracle import kiwi_oracle as ko
    ko.lib()
    lines = ["FFT-implementation invariance of the reference's integer outputs (tests/test_fft_invariance_cpu.py)",
             "transforms: " + "; ".join("prec=%d %s" % (p, NAMES[p]) for p in PRECS), ""]
    for name, fn, dlo, dhi in (("configs[0] PRN1, 1-bit IF, 41 bins", config0, -20, 20),
                               ("configs[1] 32 SVs x 41 bins", config1, -20, 20),
                               ("configs[4] 59 SVs x 256 bins, 10 ms", config4, -128, 127)):
        runs, extra = fn(ko)
        diffs = assert_same_integers(runs, name)
        gap, spread = acq_margins(runs)
        ncell = runs[1][1].size
        lines.append("%s: %d cells, %d SVs" % (name, ncell, runs[1][0].size))
        for prec in (0, 2):
            lines.append("   %-22s vs f64: %d of %d cell peak indices differ, %d of %d SV results differ" % (
                NAMES[prec], diffs[prec][0], ncell, diffs[prec][1], runs[1][0].size))
        lines.append("   smallest relative snr gap winner / runner-up Doppler bin over the SVs: %.3e (SV %d); largest relative snr "
                     "difference between transforms over all cells: %.3e  -> margin %.0fx" % (
                         gap.min(), int(gap.argmin()), spread, gap.min() / spread))
        if name.startswith("configs[1]"):
            codes, iq = extra
            data = ko.sample_iq16(iq, prec=1)
            worst = min(cell_peak_margin(ko, ko.code_fft(c, boc=b, prec=1), data, ko.L1_LIMIT, dlo, dhi) for c, b in codes)
            lines.append("   smallest relative gap peak power / runner-up lag over all %d cells: %.3e" % (ncell, worst))
        lines.append("")
    tables = (wf.window_functions(), wf.cic_comp_table())
    for case in CASES:
        runs = wf_three_ways(ko, case, tables)
        w = runs[1][0].astype(int)
        lines.append("waterfall z%-2d interp %d window %d comp %d ovl %d inv %d: u8 pixels differing from the f64 row: %s" % (
            case[0], case[2], case[3], case[4], case[5], case[6],
            ", ".join("%s %d" % (NAMES[p], int(np.count_nonzero(runs[p][0].astype(int) - w))) for p in (0, 2))))
    return "\n".join(lines)


if __name__ == "__main__":
    print(report())
