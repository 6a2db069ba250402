"""GPU parity: libkiwigpu acquisition (through the C ABI) vs the CPU oracle.

Bars (BASELINE.json north_star): peak index / Doppler bin bit-exact; float
results within 1e-5 relative (spectra: relative to the spectrum's max magnitude).
"""
import numpy as np
import pytest

from flydog_sdr_gps_amd import Searcher, prn, sats, synth
from tests.fixtures import oracle_next_rows, oracle_row      # the row BEHIND a satellite's own, which a negative Doppler bin reads

pytestmark = pytest.mark.gpu

RTOL = 1e-5
L1 = sats.L1_LIMIT


def relmax(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def searcher(gpu_ctx):
    s = Searcher(gpu_ctx, max_blocks=4)
    yield s
    s.close()


@pytest.fixture(scope="module")
def navstar_codes(searcher, oracle):
    codes = {}
    for sat in range(32):
        _, t1, t2, _ = sats.SATS[sat]
        chips = prn.cacode(t1, t2)
        searcher.set_code(sat, chips)
        codes[sat] = oracle.code_fft(chips)
    return codes


@pytest.fixture(scope="module")
def navstar_codes_for(oracle):
    def make(searcher, sat_list):
        out = []
        for sat in sat_list:
            _, t1, t2, _ = sats.SATS[sat]
            chips = prn.cacode(t1, t2)
            searcher.set_code(sat, chips)
            out.append(oracle.code_fft(chips))
        return np.stack(out)
    return make


def test_code_table_matches_oracle(searcher, navstar_codes):
    for sat in (0, 8, 31):
        got = searcher.get_code_fft(sat)
        assert relmax(got, navstar_codes[sat]) < RTOL


def test_sample_bits_time_domain_bit_exact(searcher, oracle):
    bits = synth.config0_bits()
    searcher.sample(bits)
    _, td = oracle.sample_bits(bits, want_td=True)
    got = searcher.get_data_td()
    # same operations in the same order, no FMA contraction on either side
    assert np.array_equal(got.view(np.uint32), td.view(np.uint32))


def test_sample_bits_spectrum(searcher, oracle):
    bits = synth.config0_bits()
    searcher.sample(bits)
    want = oracle.sample_bits(bits)
    assert relmax(searcher.get_data_fft(), want) < RTOL


def test_sample_iq16(searcher, oracle):
    iq = synth.config1_iq16()
    searcher.sample_iq16(iq)
    want, td = oracle.sample_iq16(iq, want_td=True)
    assert np.array_equal(searcher.get_data_td().view(np.uint32), td.view(np.uint32))
    assert relmax(searcher.get_data_fft(), want) < RTOL


SNR_RTOL = RTOL           # north_star's 1e-5 on snr too: the achieved maxima are 3.1e-6 (configs[1]) and 7.3e-6 (configs[4]), profiles/r05_float_errors.txt


def check_cells(got_cells, want_cells, tag="cells"):
    from tests.errlog import record
    assert np.array_equal(got_cells["idx"], want_cells["idx"])          # bit-exact peak index
    for k, tol in (("max_pwr", RTOL), ("tot_pwr", RTOL), ("snr", SNR_RTOL)):
        record("%s.%s" % (tag, k), got_cells[k], want_cells[k], tol)
    np.testing.assert_allclose(got_cells["max_pwr"], want_cells["max_pwr"], rtol=RTOL)
    np.testing.assert_allclose(got_cells["tot_pwr"], want_cells["tot_pwr"], rtol=RTOL)
    np.testing.assert_allclose(got_cells["snr"], want_cells["snr"], rtol=SNR_RTOL)


def test_config0_prn1(searcher, navstar_codes, oracle):
    """BASELINE.json configs[0]: PRN1 cold acquisition from 4 ms of 1-bit IF."""
    bits = synth.config0_bits()
    searcher.sample(bits)
    res, cells = searcher.correlate_many([0])
    data = oracle.sample_bits(bits)
    want, wcells = oracle.correlate(navstar_codes[0], data, code_next=oracle_row(oracle, searcher, 1))
    r = res[0, 0]
    assert (int(r["dop"]), int(r["idx"]), int(r["valid"])) == (want["dop"], want["idx"], 1)
    assert abs(r["snr"] - want["snr"]) <= SNR_RTOL * want["snr"]
    assert int(r["dop"]) == 6 and int(r["idx"]) == 1202           # injected +1500 Hz, 300.5 chips
    check_cells(cells[0, 0], wcells, "configs[0] 41 cells")
    # the SearchTask view: ca_shift *= DECIM
    out = searcher.search([0], packed=bits)
    assert out[0].lo_shift == 6 and out[0].ca_shift == 4808 and out[0].snr >= 16


def test_config1_32sv_41bins(searcher, navstar_codes, oracle):
    """BASELINE.json configs[1]: 32 SVs x 41 Doppler bins on int16 IQ."""
    iq = synth.config1_iq16()
    searcher.sample_iq16(iq)
    svs = list(range(32))
    res, cells = searcher.correlate_many(svs)
    data = oracle.sample_iq16(iq)           # the oracle chain end to end, from the int16 samples
    codes = np.stack([navstar_codes[s] for s in svs])
    want, wcells = oracle.correlate_many(codes, data, [L1] * 32, nthreads=8, nexts=oracle_next_rows(oracle, searcher, svs))
    assert np.array_equal(res[0]["dop"], want["dop"])
    assert np.array_equal(res[0]["idx"], want["idx"])
    assert np.array_equal(res[0]["valid"], want["valid"])
    np.testing.assert_allclose(res[0]["snr"], want["snr"], rtol=SNR_RTOL)
    check_cells(cells[0], wcells, "configs[1] 1312 cells")
    found = {s + 1 for s in svs if res[0, s]["snr"] >= 16}
    assert found == {p for p, *_ in synth.CONFIG1_PRESENT}
    for p, tau, fd, _ in synth.CONFIG1_PRESENT:
        r = res[0, p - 1]
        assert int(r["dop"]) == int(round(fd / 249.755859375))
        assert abs(int(r["idx"]) - tau * 4) <= 1


def test_injected_spectrum_and_shifts(searcher, navstar_codes, oracle):
    """Correlate() on a hand-made data spectrum: data = code shifted by d bins and
    advanced by `delay` samples -> the peak must come back at exactly (d, delay)
    (prod = conj(data)*code[k-dop], backward FFT: search.cpp:471,481)."""
    code = navstar_codes[4]
    n = np.arange(16384)
    for d, delay in ((-20, 0), (20, 4091), (0, 17), (-7, 2048), (13, 1)):
        data = np.roll(code, d) * np.exp(2j * np.pi * n * delay / 16384)
        searcher.set_data_fft(data.astype(np.complex64))
        res, cells = searcher.correlate_many([4])
        want, wcells = oracle.correlate(code, data.astype(np.complex64), code_next=oracle_row(oracle, searcher, 5))
        assert (int(res[0, 0]["dop"]), int(res[0, 0]["idx"])) == (want["dop"], want["idx"])
        assert (want["dop"], want["idx"]) == (d, delay)
        # only the matched-Doppler cell: this noise-free input makes the other
        # cells' outputs mirror-symmetric (exact power ties between two lags)
        di = d - searcher.dop_lo
        check_cells(cells[0, 0][di:di + 1], wcells[di:di + 1])


def test_exact_ties_go_to_the_first_lag_and_the_first_bin(gpu_ctx, oracle):
    """search.cpp:486-495: both scans are strict `>`, so of equal maxima the FIRST one wins -- the lowest
    lag inside a cell, the lowest Doppler bin among cells.  A product spectrum that is a single DC bin
    transforms to the same value at every lag, exactly (every butterfly adds zeros, the only non-zero
    input of each pass has twiddle 1): every lag of every cell ties, across lanes, waves and the
    workgroup merge of the reduction, and every cell's snr is exactly 1."""
    s = Searcher(gpu_ctx, max_sats=2)
    try:
        n = s.fft_len
        code = np.ones(n, np.complex64)
        data = np.zeros(n, np.complex64)
        data[0] = 3 + 4j
        s.set_data_fft(data)
        for limit in (sats.L1_LIMIT, sats.E1B_LIMIT, 1000):      # one accumulator, four (all 16368 lags), a short window
            s.set_code_fft(0, code, limit=limit)
            res, cells = s.correlate_many([0])
            want, wcells = oracle.correlate(code, data, limit=limit)
            assert np.all(wcells["idx"] == 0) and np.all(wcells["snr"] == 1.0) and (want["dop"], want["idx"]) == (s.dop_lo, 0)
            assert np.all(cells[0, 0]["idx"] == 0), limit
            assert np.all(cells[0, 0]["max_pwr"] == 25.0) and np.all(cells[0, 0]["snr"] == 1.0), limit
            assert (int(res[0, 0]["dop"]), int(res[0, 0]["idx"]), int(res[0, 0]["valid"])) == (s.dop_lo, 0, 1)
    finally:
        s.close()


def test_all_zero_input_is_invalid(searcher, navstar_codes):
    """search.cpp:455,495: snr = 0/0 is never > 0, so nothing is reported."""
    searcher.set_data_fft(np.zeros(16384, np.complex64))
    res, _ = searcher.correlate_many([0, 1])
    assert np.all(res["valid"] == 0) and np.all(res["snr"] == 0)
    out = searcher.search([0], lo_shift=3, ca_shift=44)
    assert (out[0].lo_shift, out[0].ca_shift, out[0].valid) == (3, 44, 0)


def test_multi_block_batch(searcher, navstar_codes, oracle):
    """Several resident blocks ("receivers") searched in one launch."""
    svs = [0, 2, 6, 10, 13, 18, 21, 29, 5]
    datas = []
    for b in range(3):
        iq = synth.config1_iq16(seed=100 + b, cn0_dbhz=46.0 + b)
        searcher.sample_iq16(iq, block=b)
        datas.append(searcher.get_data_fft(b))
    res, cells = searcher.correlate_many(svs, nblocks=3)
    codes = np.stack([navstar_codes[s] for s in svs])
    for b in range(3):
        want, wcells = oracle.correlate_many(codes, datas[b], [L1] * len(svs), nthreads=8, nexts=oracle_next_rows(oracle, searcher, svs))
        assert np.array_equal(res[b]["dop"], want["dop"])
        assert np.array_equal(res[b]["idx"], want["idx"])
        check_cells(cells[b], wcells)


def test_custom_limit_and_narrow_doppler(gpu_ctx, oracle):
    s = Searcher(gpu_ctx, max_sats=4, dop_lo=-3, dop_hi=5)
    chips = prn.cacode(3, 7)
    s.set_code(1, chips, limit=1000)
    bits = synth.gps_scene_bits([(chips, 100.25, 700.0, 0.3)], seed=7)
    s.sample(bits)
    res, cells = s.correlate_many([1])
    want, wcells = oracle.correlate(oracle.code_fft(chips), oracle.sample_bits(bits), limit=1000,
                                    dop_lo=-3, dop_hi=5)
    assert (int(res[0, 0]["dop"]), int(res[0, 0]["idx"])) == (want["dop"], want["idx"])
    check_cells(cells[0, 0], wcells)
    s.close()


@pytest.mark.parametrize("limit", [1, 255, 256, 257, 3839, 3840, 3841, 4095, 4096, 4097, 4607, 4608, 4609, 8191, 8192, 12288,
                                   15871, 15872, 15873, 16367, 16384])   # (> 4096: the 512-thread kernel, rows of 512 lags)
def test_search_window_boundaries(gpu_ctx, oracle, limit):
    """The peak-search window (search.cpp:486 `limit`) on and around the 256-lag rows of the cell-end scan
    (rows wholly inside the window skip the per-lane test) and around 4096, where the four-accumulator
    correlator takes over: every cell's first maximum and totals against the oracle, with the true peak
    both inside and outside the window."""
    s = Searcher(gpu_ctx, max_sats=2, dop_lo=-4, dop_hi=4)
    try:
        chips = prn.cacode(2, 6)
        s.set_code(0, chips, limit=limit)
        for delay in (100.25, 900.5):           # code delay in chips: lag 401 / 3602 of the 4.092 MS/s grid
            bits = synth.gps_scene_bits([(chips, delay, 0.0, 0.5)], seed=int(delay))
            s.sample(bits)
            res, cells = s.correlate_many([0])
            want, wcells = oracle.correlate(oracle.code_fft(chips), oracle.sample_bits(bits), limit=limit,
                                            dop_lo=-4, dop_hi=4)
            assert (int(res[0, 0]["dop"]), int(res[0, 0]["idx"])) == (want["dop"], want["idx"]), (limit, delay)
            check_cells(cells[0, 0], wcells)
            assert np.all(cells[0, 0]["idx"] < limit)
    finally:
        s.close()


def test_error_paths(gpu_ctx):
    from flydog_sdr_gps_amd import KiwiGpuError
    s = Searcher(gpu_ctx, max_sats=4)
    with pytest.raises(KiwiGpuError):
        s.correlate_many([0])                 # no code set
    with pytest.raises(KiwiGpuError):
        s.set_code(9, prn.cacode(2, 6))       # sat out of range
    with pytest.raises(KiwiGpuError):
        s.set_code(0, np.full(1023, 2, np.uint8))
    with pytest.raises(ValueError):
        s.sample(np.zeros(10, np.uint8))
    s.close()


def test_golden_fixtures_incl_e1b_and_qzss(gpu_ctx, oracle):
    """tests/golden/acq_golden.npz: Navstar, QZSS (G2-init code), absent SV, edge
    Doppler and a Galileo E1B (BOC(1,1), 16368-sample window) case."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "acq_golden.npz"))
    s = Searcher(gpu_ctx)
    for k in range(int(g["ncases"])):
        sat = int(g["case%d_sat" % k])
        _, t1, t2, kind = sats.SATS[sat]
        if kind == sats.E1B:
            chips, boc = g["case%d_chips" % k], True
        else:
            chips, boc = prn.cacode(t1, t2), False
        s.set_code(sat, chips, boc=boc)
        want_code = oracle.code_fft(chips, boc=boc)
        assert relmax(s.get_code_fft(sat), want_code) < RTOL
        s.sample(g["case%d_bits" % k])
        res, cells = s.correlate_many([sat])
        want = g["case%d_result" % k]
        r = res[0, 0]
        assert (int(r["dop"]), int(r["idx"]), int(r["valid"])) == tuple(int(v) for v in want[1:])
        assert abs(r["snr"] - want[0]) <= 3 * RTOL * want[0]
        assert np.array_equal(cells[0, 0]["idx"], g["case%d_cell_idx" % k])
        np.testing.assert_allclose(cells[0, 0]["snr"], g["case%d_cell_snr" % k], rtol=3 * RTOL)
    s.close()


def test_mixed_l1_e1b_batch(gpu_ctx, oracle):
    """One launch list mixing 4092-window (C/A) and 16368-window (E1B) SVs."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "acq_golden.npz"))
    e1b = g["case5_chips"]
    s = Searcher(gpu_ctx, max_sats=8)
    ca = prn.cacode(2, 6)
    s.set_code(0, ca)
    s.set_code(1, e1b, boc=True)
    s.set_code(2, prn.cacode(3, 7))
    s.sample(g["case5_bits"])
    res, cells = s.correlate_many([1, 0, 2, 1])
    data = s.get_data_fft()
    w_e, c_e = oracle.correlate(oracle.code_fft(e1b, boc=True), data, limit=sats.E1B_LIMIT, code_next=oracle_row(oracle, s, 2))
    w_0, c_0 = oracle.correlate(oracle.code_fft(ca), data, code_next=oracle_row(oracle, s, 1))
    for pos, (w, c) in zip((0, 1, 3), ((w_e, c_e), (w_0, c_0), (w_e, c_e))):
        assert (int(res[0, pos]["dop"]), int(res[0, pos]["idx"])) == (w["dop"], w["idx"])
        check_cells(cells[0, pos], c)
    s.close()


def test_two_block_sets_pipelined(gpu_ctx, navstar_codes_for, oracle):
    """Alternating block sets: Sample() of set B overlaps Correlate() of set A on the
    library's two streams; results must be those of the right block every time."""
    s = Searcher(gpu_ctx, max_blocks=4)
    svs = [0, 2, 6, 10, 13]
    codes = navstar_codes_for(s, svs)
    iqs = [synth.config1_iq16(seed=200 + i, cn0_dbhz=45.0 + i) for i in range(6)]
    wants = []
    for iq in iqs:
        w, _ = oracle.correlate_many(codes, oracle.sample_iq16(iq), [L1] * len(svs), nthreads=4, nexts=oracle_next_rows(oracle, s, svs))
        wants.append(w)
    got = []
    for step in range(3):                       # steps use blocks {0,1}, {2,3}, {0,1}
        first = (step & 1) * 2
        for b in range(2):
            s.sample_iq16(iqs[2 * step + b], block=first + b)
        res, _ = s.correlate_many(svs, nblocks=2, want_cells=False, first_block=first)
        got.append(res.copy())
    for step in range(3):
        for b in range(2):
            w = wants[2 * step + b]
            assert np.array_equal(got[step][b]["dop"], w["dop"])
            assert np.array_equal(got[step][b]["idx"], w["idx"])
            np.testing.assert_allclose(got[step][b]["snr"], w["snr"], rtol=3 * RTOL)
    s.close()


def test_host_batch_sample_equals_per_block(gpu_ctx):
    """kg_acq_sample_iq16_batch (host buffers, one transfer per batch) gives the spectra the
    per-block host entry point gives, and the caller's array may be overwritten at once."""
    s = Searcher(gpu_ctx, max_blocks=6)
    try:
        blocks = np.stack([synth.config1_iq16(seed=77 + b) for b in range(3)])
        s.sample_iq16_host_batch(blocks, first_block=3)
        keep = blocks.copy()
        blocks[:] = 0                                   # "enqueue only": the library has its own copy
        for b in range(3):
            s.sample_iq16(keep[b], block=b)
        for b in range(3):
            assert np.array_equal(s.get_data_fft(block=b).view(np.uint32), s.get_data_fft(block=3 + b).view(np.uint32))
    finally:
        s.close()
