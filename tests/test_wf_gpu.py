"""GPU parity: libkiwigpu waterfall frames (through the C ABI) vs the CPU oracle's
restatement of sample_wf() + compute_frame() (rx/rx_waterfall.cpp).

Bars.  North star: float spectra within 1e-5 of the spectrum's maximum.  So
  pwr, pwr_out:  |gpu - oracle| <= 1e-5 * max(oracle)
  dB:            the same bound pushed through 10*log10: a pixel of power p may move by
                 10*log10(1 + 1e-5*max/p) (+1e-4 for log10f's last ulp) -- tight on strong
                 pixels, loose 90 dB below a strong carrier, where the fp32 FFT's rounding
                 noise (relative to the carrier) is what the reference's FFTW float FFT has too;
  u8 row:        identical, except that a pixel may differ by one LSB when the oracle's dB
                 is closer to an integer (the (int) truncation edge, :1546) than that
                 pixel's dB bound.
test_low_dynamic_range_is_tight pins the tight regime: every pixel within 45 dB of the
maximum, at most MAX_FLIPS one-LSB differences per frame, each within DB_EDGE of an edge."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import Waterfall, WfParams, synth, wf

pytestmark = pytest.mark.gpu

RTOL = 1e-5
DB_EDGE = 2e-4
MAX_FLIPS = 2


@pytest.fixture(scope="module")
def tables():
    return wf.window_functions(), wf.cic_comp_table()


@pytest.fixture(scope="module")
def engine(gpu_ctx, tables):
    w = Waterfall(gpu_ctx, nchan=16)
    w.set_tables(*tables)
    yield w
    w.close()


def oracle_frame(oracle, tables, iq, p, interp, window_func, cic_comp, overlapped, inv, scale=None):
    windows, cic = tables
    m, d = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, inv)
    sc = np.full(1024, p.fft_scale, np.float32) if scale is None else scale
    samps = oracle.wf_window_iq(iq, windows[window_func])
    return oracle.wf_compute_frame(samps, p.zoom, window_func, interp, cic_comp, overlapped, p.fft_used,
                                   p.plot_width, p.plot_width_clamped, m, d, sc,
                                   (sc / np.float32(2)).astype(np.float32), p.fft_offset, cic)


def db_bound(w_pwr_out, scale_like=None):
    """Per-pixel dB tolerance implied by a power error of RTOL * max."""
    dp = RTOL * float(w_pwr_out.max())
    p = np.maximum(w_pwr_out.astype(np.float64), 1e-300)
    return 10.0 * np.log10(1.0 + dp / p) + 1e-4


def check_row(got, want, want_db, tol_db=None, max_flips=None):
    """tol_db None: the tight regime (DB_EDGE, MAX_FLIPS)."""
    diff = got.astype(int) - want.astype(int)
    bad = np.nonzero(diff)[0]
    if tol_db is None:
        assert bad.size <= MAX_FLIPS, "too many differing pixels: %s" % bad[:10]
        tol_db = np.full(want.size, DB_EDGE)
    elif max_flips is not None:
        assert bad.size <= max_flips
    clamped = np.clip(want_db.astype(np.float64), -200.0, 0.0)
    for i in bad:
        assert abs(diff[i]) <= 1 + int(tol_db[i]), (i, got[i], want[i], tol_db[i])
        assert abs(clamped[i] - np.rint(clamped[i])) < tol_db[i] + DB_EDGE, (i, want_db[i], tol_db[i])


CASES = [
    # zoom, start, interp, window, cic_comp, overlapped, inversion
    (0, 0.0, wf.WF_CMA, wf.WINF_HANNING, True, False, False),
    (3, 2.0e6, wf.WF_MAX, wf.WINF_BLACKMAN_HARRIS, True, False, False),
    (10, 9.0e6, wf.WF_DROP, wf.WINF_HANNING, True, False, True),
    (0, 0.0, wf.WF_MAX, wf.WINF_BLACKMAN_HARRIS, False, False, False),   # dc = 4 bins
    (1, 1.0e6, wf.WF_MIN, wf.WINF_HAMMING, True, False, False),          # zoom 1: never compensated
    (5, 5.0e6, wf.WF_LAST, wf.WINF_NONE, True, True, False),             # overlapped: no comp
    (7, 3.0e6, wf.WF_CMA, wf.WINF_HANNING, True, False, True),
    (14, 1.6e7, wf.WF_MIN, wf.WINF_HANNING, False, False, False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "z%d-i%d-w%d-c%d%d-inv%d" % (c[0], c[2], c[3], c[4], c[5], c[6]))
def test_frame_matches_oracle(engine, oracle, tables, case):
    zoom, start, interp, window_func, cic_comp, overlapped, inv = case
    p = WfParams.for_zoom(zoom, start, spectral_inversion=inv)
    engine.set_channel(0, p, interp=interp, window_func=window_func, cic_comp=cic_comp,
                       overlapped=overlapped, spectral_inversion=inv)
    iq = synth.wf_iq_frame(seed=1000 + zoom)
    out, pwr, pwr_out, dB = engine.debug_frame(0, iq)
    w_out, w_pwr, w_pwr_out, w_dB = oracle_frame(oracle, tables, iq, p, interp, window_func, cic_comp,
                                                 overlapped, inv)
    n = p.fft_used
    assert np.abs(pwr[:n] - w_pwr).max() <= RTOL * w_pwr.max()
    assert np.abs(pwr_out - w_pwr_out).max() <= RTOL * max(w_pwr_out.max(), 1e-30)
    tol = db_bound(w_pwr_out)
    floor = w_dB > -290                       # 10*log10f(1e-30): untouched / masked pixels
    assert np.all(np.abs(dB - w_dB)[floor] <= tol[floor]), np.abs(dB - w_dB)[floor].max()
    check_row(out, w_out, w_dB, tol)
    # the plain batched entry point gives the same row
    assert np.array_equal(engine.frames([0], iq[None])[0], out)


def test_strong_pixels_are_tight(engine, oracle, tables):
    """Pixels within 45 dB of the frame's maximum: dB within 5e-4, and the bytes agree
    except at (int) truncation edges (at most MAX_FLIPS per frame, each within DB_EDGE)."""
    for k, (zoom, interp) in enumerate(((0, wf.WF_CMA), (2, wf.WF_MAX), (6, wf.WF_CMA), (9, wf.WF_LAST))):
        p = WfParams.for_zoom(zoom, 1.0e6)
        engine.set_channel(4, p, interp=interp)
        iq = synth.wf_iq_frame(seed=300 + k, tones=((0.11, -52.0), (0.3, -58.0)), noise_dbfs=-30.0)
        out, pwr, pwr_out, dB = engine.debug_frame(4, iq)
        w_out, w_pwr, w_pwr_out, w_dB = oracle_frame(oracle, tables, iq, p, interp, wf.WINF_HANNING,
                                                     True, False, False)
        strong = w_pwr_out > 10 ** -4.5 * w_pwr_out.max()
        assert strong.sum() > 800
        np.testing.assert_allclose(dB[strong], w_dB[strong], rtol=0, atol=5e-4)
        check_row(out[strong], w_out[strong], w_dB[strong])      # tight: MAX_FLIPS, DB_EDGE
        check_row(out, w_out, w_dB, db_bound(w_pwr_out))


def test_zero_input_row_is_floor(engine):
    """All-zero samples: 10*log10f(1e-30) + offset clamps at -200 -> byte 55 (:1539-1546)."""
    p = WfParams.for_zoom(4, 0.0)
    engine.set_channel(1, p)
    out = engine.frames([1], np.zeros((1, 8192, 2), np.int16))
    assert np.all(out == 55)


def test_full_scale_clamps_at_top(engine, oracle, tables):
    """A full-scale carrier exceeds 0 dB -> clamp -> byte 255 = (u1_t)(int)(-1.0)."""
    p = WfParams.for_zoom(0, 0.0)
    engine.set_channel(2, p, interp=wf.WF_MAX, window_func=wf.WINF_NONE, cic_comp=False)
    t = np.arange(8192)
    iq = np.empty((8192, 2), np.int16)
    iq[:, 0] = np.rint(32767 * np.cos(2 * np.pi * 400 * t / 8192))
    iq[:, 1] = np.rint(32767 * np.sin(2 * np.pi * 400 * t / 8192))
    out = engine.frames([2], iq[None])[0]
    w_out, _, w_po, w_dB = oracle_frame(oracle, tables, iq, p, wf.WF_MAX, wf.WINF_NONE, False, False, False)
    # an exact-bin carrier leaves every other bin at quantisation-noise level, 100+ dB down
    check_row(out, w_out, w_dB, db_bound(w_po) + 200.0 * (w_dB < -100))
    assert out.max() == 255 and w_dB.max() > 0 and out[400 * p.plot_width // p.fft_used] == 255


def test_masked_scale_and_custom_scale_array(engine, oracle, tables):
    """fft_scale[] carries the masked-frequency zeros of :905-919."""
    p = WfParams.for_zoom(2, 4.0e6)
    scale = np.full(1024, p.fft_scale, np.float32)
    scale[100:140] = 0
    engine.set_channel(3, p, interp=wf.WF_CMA, fft_scale=scale)
    iq = synth.wf_iq_frame(seed=5)
    out = engine.frames([3], iq[None])[0]
    w_out, _, w_po, w_dB = oracle_frame(oracle, tables, iq, p, wf.WF_CMA, wf.WINF_HANNING, True, False, False,
                                        scale=scale)
    check_row(out, w_out, w_dB, db_bound(w_po))
    assert np.all(out[100:140] == 55)


def test_batch_of_frames_over_mixed_channels(engine, oracle, tables):
    """14 channels (BASELINE configs[2] zoom set) x 3 frames in one launch."""
    zooms = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14]
    ps = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch)
        ps.append(p)
        engine.set_channel(ch, p, interp=wf.WF_CMA, window_func=wf.WINF_HANNING, cic_comp=True)
    chan_of, iqs = [], []
    for fr in range(3):
        for ch in range(len(zooms)):
            chan_of.append(ch)
            iqs.append(synth.wf_iq_frame(seed=77 * fr + ch))
    out = engine.frames(chan_of, np.stack(iqs))
    for k, (ch, iq) in enumerate(zip(chan_of, iqs)):
        w_out, _, w_po, w_dB = oracle_frame(oracle, tables, iq, ps[ch], wf.WF_CMA, wf.WINF_HANNING, True, False, False)
        check_row(out[k], w_out, w_dB, db_bound(w_po))


def test_long_list_with_windows_and_compensation_changing(engine, oracle, tables):
    """A workgroup keeps its window values and CIC factors in registers from frame to frame and re-reads them only when the
    next frame's channel wants others (round 4).  1 600 frames -- three and more per workgroup -- over six channels that
    differ in window function, in compensation on / off and in zoom 0 (all sixteen factor rows) against zoomed (eight), in a
    random order: every row against the oracle for ITS channel."""
    cfgs = [(0, wf.WINF_HANNING, True), (5, wf.WINF_HAMMING, True), (3, wf.WINF_HANNING, False),
            (0, wf.WINF_BLACKMAN_HARRIS, False), (9, wf.WINF_NONE, True), (1, wf.WINF_HANNING, True)]
    ps = []
    for ch, (z, wfn, comp) in enumerate(cfgs):
        p = WfParams.for_zoom(z, 1.5e6 * ch)
        ps.append(p)
        engine.set_channel(ch, p, interp=wf.WF_MAX, window_func=wfn, cic_comp=comp)
    niq = 12
    iqs = [synth.wf_iq_frame(seed=4100 + i) for i in range(niq)]
    rng = np.random.default_rng(2024)
    n = 1600
    chan_of = rng.integers(0, len(cfgs), n)
    chan_of[:40] = 0                         # a run without any change, then changes of every kind
    chan_of[40:80] = np.tile([0, 5], 20)     # same window, same compensation, 16 rows <-> 8 rows... both zoom 0 / 1
    which = rng.integers(0, niq, n)
    out = engine.frames([int(c) for c in chan_of], np.stack([iqs[i] for i in which]))
    want = {}
    for k in range(n):
        key = (int(chan_of[k]), int(which[k]))
        if key not in want:
            z, wfn, comp = cfgs[key[0]]
            want[key] = oracle_frame(oracle, tables, iqs[key[1]], ps[key[0]], wf.WF_MAX, wfn, comp, False, False)
        w_out, _, w_po, w_dB = want[key]
        check_row(out[k], w_out, w_dB, db_bound(w_po))


def test_channel_map_changing_between_calls(engine, oracle, tables):
    """The frame -> channel map of a batch is kept on the device while it does not change
    (kg_stage_cache): same map again, a permuted one of the same length, a longer one, the first one
    again -- every call must use ITS map."""
    zooms = [0, 3, 7, 10]
    ps = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch)
        ps.append(p)
        engine.set_channel(ch, p, interp=wf.WF_MAX, window_func=wf.WINF_HANNING, cic_comp=True)
    iqs = [synth.wf_iq_frame(seed=500 + i) for i in range(12)]
    want = {}

    def ref(ch, i):
        if (ch, i) not in want:
            want[(ch, i)] = oracle_frame(oracle, tables, iqs[i], ps[ch], wf.WF_MAX, wf.WINF_HANNING, True, False, False)
        return want[(ch, i)]

    maps = [[0, 1, 2, 3], [0, 1, 2, 3], [3, 2, 1, 0], [1, 1, 0, 2], list(range(4)) * 3, [0, 1, 2, 3], [2]]
    for m in maps:
        out = engine.frames(m, np.stack(iqs[:len(m)]))
        for k, ch in enumerate(m):
            w_out, _, w_po, w_dB = ref(ch, k)
            check_row(out[k], w_out, w_dB, db_bound(w_po))


def test_frames_read_in_place_at_offsets(engine, gpu_ctx, oracle, tables):
    """kg_wf_frames_at_dev: frames that are not back to back (taken where the DDC left them: per-channel rows, a
    frame starting anywhere on an even sample).  Five frames scattered over a buffer in a shuffled order, offsets
    changing between two calls of the same length -- every row against the oracle on the samples at ITS offset;
    an odd offset, or a frame that would run past the stated extent of the buffer, is refused."""
    from flydog_sdr_gps_amd import KiwiGpuError
    zooms = [0, 4, 9]
    ps = []
    for ch, z in enumerate(zooms):
        p = WfParams.for_zoom(z, 1.0e6 * ch)
        ps.append(p)
        engine.set_channel(ch, p, interp=wf.WF_CMA, window_func=wf.WINF_HANNING, cic_comp=True)
    nbuf = 6 * 8192 + 4096
    rng = np.random.default_rng(11)
    buf = np.zeros((nbuf, 2), np.int16)
    frames = [synth.wf_iq_frame(seed=900 + i) for i in range(5)]
    d_iq = gpu_ctx.alloc(buf.nbytes)
    d_out = gpu_ctx.alloc(5 * 1024)
    try:
        for offs in ([2 * 8192 + 2, 0, 5 * 8192 + 4096, 8192 + 2048, 4 * 8192 - 6], [2, 8192 + 4, 2 * 8192 + 6, 3 * 8192 + 8, 5 * 8192]):
            buf[:] = rng.integers(-100, 100, buf.shape)
            for fr, off in zip(frames, offs):
                buf[off:off + 8192] = fr
            chan_of = [1, 0, 2, 1, 0]
            gpu_ctx.upload(d_iq, buf)
            engine.frames_dev(chan_of, d_iq, d_out, frame_off=offs, iq_len=nbuf)
            out = np.zeros((5, 1024), np.uint8)
            gpu_ctx.download(d_out, out)
            for k, ch in enumerate(chan_of):
                w_out, _, w_po, w_dB = oracle_frame(oracle, tables, buf[offs[k]:offs[k] + 8192], ps[ch], wf.WF_CMA,
                                                    wf.WINF_HANNING, True, False, False)
                check_row(out[k], w_out, w_dB, db_bound(w_po))
        with pytest.raises(KiwiGpuError):
            engine.frames_dev([0], d_iq, d_out, frame_off=[3], iq_len=nbuf)
        # a frame must end inside the extent the caller states: the last admissible start, then one pair past it
        engine.frames_dev([0], d_iq, d_out, frame_off=[nbuf - 8192], iq_len=nbuf)
        with pytest.raises(KiwiGpuError):
            engine.frames_dev([0], d_iq, d_out, frame_off=[nbuf - 8190], iq_len=nbuf)
        with pytest.raises(KiwiGpuError):
            engine.frames_dev([0, 1], d_iq, d_out, frame_off=[0, 8192], iq_len=8192 + 8190)
    finally:
        gpu_ctx.free(d_iq)
        gpu_ctx.free(d_out)


def test_wf_error_paths(gpu_ctx, tables):
    from flydog_sdr_gps_amd import KiwiGpuError
    w = Waterfall(gpu_ctx, nchan=2)
    p = WfParams.for_zoom(0, 0.0)
    with pytest.raises(KiwiGpuError):
        w.frames([0], np.zeros((1, 8192, 2), np.int16))          # tables not set
    w.set_tables(*tables)
    with pytest.raises(KiwiGpuError):
        w.frames([0], np.zeros((1, 8192, 2), np.int16))          # channel not configured
    with pytest.raises(KiwiGpuError):
        w.set_channel(5, p)                                      # channel out of range
    w.set_channel(0, p)
    with pytest.raises(KiwiGpuError):
        w.frames([1], np.zeros((1, 8192, 2), np.int16))
    w.close()
