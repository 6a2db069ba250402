"""GPU parity: waterfall DDC (kg_ddc, through the C ABI) vs oracle/kiwi_oracle_ddc.c, the
sequential restatement of verilog/rx/{iq_mixer,cic_prune_var,waterfall_1cic}.v + cic_wf1.vh.
Integer path: every output sample must be BIT-EXACT."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import Ddc

pytestmark = pytest.mark.gpu


def adc_stream(n, seed, tones=((0.0123, 9000.0), (0.201, 700.0)), noise=40.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n)
    x = rng.normal(0, noise, n)
    for f, a in tones:
        x = x + a * np.cos(2 * np.pi * f * t + rng.random() * 6)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def inc_for(f):                      # -f cycles/sample as a 48-bit phase increment (i_offset = -offset, :499)
    return (-int(round(f * 2 ** 48))) & ((1 << 48) - 1)


@pytest.mark.parametrize("log2r", [0, 1, 2, 5, 9, 13])
def test_single_channel_matches_oracle(gpu_ctx, oracle, log2r):
    n = max(1 << 16, (1 << log2r) * 40)
    adc = adc_stream(n, seed=10 + log2r)
    inc = inc_for(0.0123 + 2.0 ** -22)       # 0.24 Hz-per-unit offset: inside every CIC passband
    d = Ddc(gpu_ctx, nchan=1, max_samples=n)
    d.set_wf(0, inc, 1 << log2r)
    got = d.push(adc, [0])[0]
    want, _ = oracle.ddc_wf(adc, inc, log2r)
    assert got.shape == want.shape and got.shape[0] == n >> log2r
    assert np.array_equal(got, want)
    assert np.abs(want[8:].astype(int)).max() > 1000        # a real signal came through
    d.close()


def test_pieces_of_any_length_carry_state(gpu_ctx, oracle):
    """Pushing the stream in ragged pieces == pushing it whole == the oracle."""
    n = 200_000
    adc = adc_stream(n, seed=3)
    inc = inc_for(0.2007)
    for log2r in (3, 7, 11):
        d = Ddc(gpu_ctx, nchan=1, max_samples=n)
        d.set_wf(0, inc, 1 << log2r)
        want, _ = oracle.ddc_wf(adc, inc, log2r)
        parts, pos = [], 0
        for step in (1, 63, 64, 65, 4097, 9, 70001, 1000, n):
            if pos >= n:
                break
            parts.append(d.push(adc[pos:pos + step], [0])[0])
            pos += step
        got = np.concatenate(parts)
        assert np.array_equal(got, want), log2r
        d.close()


def test_long_blocks_take_the_chunked_state_scan(gpu_ctx, oracle):
    """2^21 samples in two unequal pushes: 8192+ runs per channel, so the integrator carry scan of a
    (channel, I/Q) pair is cut into the maximum number of chunks across workgroups (look-back through
    published aggregates), with the state carried from the first push into the second; R = 1 rides
    along on the bypass kernel's persistent loop."""
    n = 1 << 21
    adc = adc_stream(n, seed=77, tones=((0.0123, 9000.0), (0.31, 1500.0)))
    chans = [(inc_for(0.0123 + 2.0 ** -20), 6), (inc_for(0.31), 12), (inc_for(0.05), 0)]
    d = Ddc(gpu_ctx, nchan=len(chans), max_samples=n)
    try:
        for ch, (inc, log2r) in enumerate(chans):
            d.set_wf(ch, inc, 1 << log2r)
        cut = (1 << 20) + 12345
        a = d.push(adc[:cut], list(range(len(chans))))
        b = d.push(adc[cut:], list(range(len(chans))))
        for ch, (inc, log2r) in enumerate(chans):
            want, _ = oracle.ddc_wf(adc, inc, log2r)
            got = np.concatenate([a[ch], b[ch]])
            assert got.shape == want.shape and np.array_equal(got, want), ch
    finally:
        d.close()


def test_reset_restarts_cic_but_not_phase(gpu_ctx, oracle):
    n = 50_000
    adc = adc_stream(2 * n, seed=5)
    inc = inc_for(0.05)
    d = Ddc(gpu_ctx, nchan=1, max_samples=n)
    d.set_wf(0, inc, 16)
    d.push(adc[:n], [0])
    d.reset(0)
    got = d.push(adc[n:], [0])[0]
    st = oracle.DdcWfState()
    st.phase = (n * inc) & ((1 << 48) - 1)               # NCO kept running, CIC zeroed
    want, _ = oracle.ddc_wf(adc[n:], inc, 4, st)
    assert np.array_equal(got, want)
    d.close()


def test_fourteen_channels_baseline_zoom_set(gpu_ctx, oracle):
    """BASELINE configs[2] decimations: zooms {0,0,1,2,..,10,12,14} -> R = 1 << max(zoom-1, 0)."""
    zooms = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14]
    n = 1 << 18
    adc = adc_stream(n, seed=8, tones=((0.031, 8000.0), (0.1234, 2000.0), (0.3, 300.0)))
    d = Ddc(gpu_ctx, nchan=len(zooms), max_samples=n)
    incs = []
    for ch, z in enumerate(zooms):
        incs.append(inc_for(0.03 + 0.007 * ch))
        d.set_wf(ch, incs[-1], 1 << max(z - 1, 0))
    got = d.push(adc, list(range(len(zooms))))
    for ch, z in enumerate(zooms):
        want, _ = oracle.ddc_wf(adc, incs[ch], max(z - 1, 0))
        assert np.array_equal(got[ch], want), (ch, z)
    d.close()


def test_adc_block_at_any_alignment(gpu_ctx, oracle):
    """The ADC block's device address need not be 16-byte aligned (the run passes then walk it sample by sample
    instead of in 16-byte groups loaded one group ahead; the bypass kernel takes its scalar loads): a block that starts
    1, 3 and 4 samples into an allocation, every kind of channel (bypass, staged R <= 8, 64-bit and 96-bit pass B),
    state carried from push to push."""
    log2rs = [0, 1, 3, 6, 10, 13]
    n = 3 * 40_000
    adc = adc_stream(n, seed=21, tones=((0.0123, 9000.0), (0.25, 1200.0)))
    incs = [inc_for(0.011 + 0.013 * k) for k in range(len(log2rs))]
    d = Ddc(gpu_ctx, nchan=len(log2rs), max_samples=n)
    d_adc = gpu_ctx.alloc(2 * (n + 16))
    chans = list(range(len(log2rs)))
    try:
        for ch, lr in enumerate(log2rs):
            d.set_wf(ch, incs[ch], 1 << lr)
        stride = n + 2
        d_out = gpu_ctx.alloc(len(chans) * stride * 4)
        parts = [[] for _ in chans]
        for piece, shift in enumerate((1, 3, 4)):
            blk = np.ascontiguousarray(adc[piece * 40_000:(piece + 1) * 40_000])
            pad = np.zeros(blk.size + 16, np.int16)
            pad[shift:shift + blk.size] = blk
            gpu_ctx.upload(d_adc, pad)
            nouts = d.push_dev(d_adc + 2 * shift, blk.size, chans, d_out, stride)
            host = np.zeros((len(chans), stride, 2), np.int16)
            gpu_ctx.download(d_out, host)
            for ch in chans:
                parts[ch].append(host[ch, :int(nouts[ch])].copy())
        gpu_ctx.free(d_out)
        for ch, lr in enumerate(log2rs):
            want, _ = oracle.ddc_wf(adc, incs[ch], lr)
            got = np.concatenate(parts[ch])
            assert got.shape == want.shape and np.array_equal(got, want), (ch, lr)
    finally:
        gpu_ctx.free(d_adc)
        d.close()


@pytest.mark.parametrize("nby,stride_pad,shift", [(1, 4, 0), (2, 1, 0), (3, 8, 0), (5, 4, 0), (2, 4, 1), (4, 3, 2), (2, 0, 4)])
def test_bypass_channels_rows_and_samples_at_any_alignment(gpu_ctx, oracle, nby, stride_pad, shift):
    """R = 1 channels (ddc_wf_bypass_kernel): whole 4096-sample blocks go through a straight-line loop -- 16-byte stores when the
    caller's rows and the sample stream are aligned, a transposed lane mapping with coalesced 4-byte stores when they are not
    (out_stride = n + 1 is enough) -- the ragged end through the general form; one to five bypass channels (four per pass over
    the samples) beside a filtered one, two pushes so that the phase carries."""
    n = 5 * 4096 + 1234
    log2rs = [0] * nby + [4]
    adc = adc_stream(2 * n, seed=77, tones=((0.031, 7000.0), (0.2, 900.0)))
    incs = [inc_for(0.017 + 0.019 * k) for k in range(len(log2rs))]
    d = Ddc(gpu_ctx, nchan=len(log2rs), max_samples=n)
    d_adc = gpu_ctx.alloc(2 * (n + 16))
    chans = list(range(len(log2rs)))
    stride = n + stride_pad
    d_out = gpu_ctx.alloc(len(chans) * stride * 4)
    try:
        for ch, lr in enumerate(log2rs):
            d.set_wf(ch, incs[ch], 1 << lr)
        parts = [[] for _ in chans]
        for piece in range(2):
            blk = np.ascontiguousarray(adc[piece * n:(piece + 1) * n])
            pad = np.zeros(n + 16, np.int16)
            pad[shift:shift + n] = blk
            gpu_ctx.upload(d_adc, pad)
            nouts = d.push_dev(d_adc + 2 * shift, n, chans, d_out, stride)
            host = np.zeros((len(chans), stride, 2), np.int16)
            gpu_ctx.download(d_out, host)
            for ch in chans:
                parts[ch].append(host[ch, :int(nouts[ch])].copy())
        for ch, lr in enumerate(log2rs):
            want, _ = oracle.ddc_wf(adc, incs[ch], lr)
            got = np.concatenate(parts[ch])
            assert got.shape == want.shape and np.array_equal(got, want), (ch, lr)
    finally:
        gpu_ctx.free(d_out)
        gpu_ctx.free(d_adc)
        d.close()


def test_extreme_inputs_wrap_like_the_registers(gpu_ctx, oracle):
    """Full-scale square wave at DC offset: the integrators wrap many times."""
    n = 1 << 17
    adc = np.where((np.arange(n) // 5000) % 2 == 0, 32767, -32768).astype(np.int16)
    d = Ddc(gpu_ctx, nchan=2, max_samples=n)
    d.set_wf(0, 0, 8192)                     # NCO at 0 Hz: cos = 16383, sin = 0
    d.set_wf(1, inc_for(0.25), 64)
    got = d.push(adc, [0, 1])
    for ch, (inc, l2) in enumerate(((0, 13), (inc_for(0.25), 6))):
        want, _ = oracle.ddc_wf(adc, inc, l2)
        assert np.array_equal(got[ch], want)
    d.close()


def test_ddc_error_paths(gpu_ctx):
    from flydog_sdr_gps_amd import KiwiGpuError
    d = Ddc(gpu_ctx, nchan=2, max_samples=4096)
    with pytest.raises(KiwiGpuError):
        d.set_wf(0, 0, 3)                    # not a power of two
    with pytest.raises(KiwiGpuError):
        d.set_wf(0, 0, 16384)                # beyond WF_1CIC_MAXD
    with pytest.raises(KiwiGpuError):
        d.set_wf(2, 0, 4)
    with pytest.raises(KiwiGpuError):
        d.outputs(1, 100)                    # channel not configured
    d.close()


# ---- audio DDC (verilog/rx/rx.v) ------------------------------------------------------
from flydog_sdr_gps_amd import RxDdc                       # noqa: E402
from flydog_sdr_gps_amd.ddc import RX_DECIM, rx_phase_inc  # noqa: E402


def test_rx_ddc_matches_oracle_bit_exact(gpu_ctx, oracle):
    n = RX_DECIM * 90 + 1234
    adc = adc_stream(n, seed=21, tones=((0.0371 + 900 / 125e6, 15000.0), (0.0371 - 2500 / 125e6, 900.0), (0.2, 5000.0)))
    inc = (-rx_phase_inc(0.0371 * 125e6)) & ((1 << 48) - 1)
    d = RxDdc(gpu_ctx, nchan=1, max_samples=n)
    d.set_freq(0, inc)
    got = d.push(adc, [0])[0]
    want, _ = oracle.ddc_rx(adc, inc)
    assert got.size == want.size == 6 * 90
    assert np.array_equal(got, want)
    d.close()


def test_rx_ddc_ragged_pushes_and_retune(gpu_ctx, oracle):
    n = RX_DECIM * 40
    adc = adc_stream(n, seed=22, tones=((0.11, 12000.0),))
    inc1, inc2 = (-rx_phase_inc(0.11 * 125e6 - 600)) & ((1 << 48) - 1), (-rx_phase_inc(0.11 * 125e6 + 1500)) & ((1 << 48) - 1)
    d = RxDdc(gpu_ctx, nchan=1, max_samples=n)
    d.set_freq(0, inc1)
    st, parts, wants, pos = None, [], [], 0
    for k, step in enumerate((1, 1735, 1736, 5209, 10416, 64, 100001, 77, n)):
        if pos >= n:
            break
        if k == 5:                       # retune mid-stream: the filters keep their state
            d.set_freq(0, inc2)
        seg = adc[pos:pos + step]
        pos += len(seg)
        assert d.outputs(0, len(seg)) * 6 == len(oracle.ddc_rx(seg, inc2 if k >= 5 else inc1,
                                                                __import__("copy").deepcopy(st) if st else None)[0])
        parts.append(d.push(seg, [0])[0])
        w, st = oracle.ddc_rx(seg, inc2 if k >= 5 else inc1, st)
        wants.append(w)
    got, want = np.concatenate(parts), np.concatenate(wants)
    assert got.size == want.size == 6 * 40 and np.array_equal(got, want)
    d.close()


def test_rx_ddc_many_channels_and_reset(gpu_ctx, oracle):
    nch, n = 14, RX_DECIM * 24
    adc = adc_stream(n, seed=23, tones=((0.05, 9000.0), (0.0503, 4000.0), (0.3, 2000.0)))
    d = RxDdc(gpu_ctx, nchan=nch, max_samples=n)
    incs = [(-rx_phase_inc((0.05 + 1e-5 * ch) * 125e6)) & ((1 << 48) - 1) for ch in range(nch)]
    for ch in range(nch):
        d.set_freq(ch, incs[ch])
    got = d.push(adc, list(range(nch)))
    for ch in range(nch):
        assert np.array_equal(got[ch], oracle.ddc_rx(adc, incs[ch])[0]), ch
    d.reset(3)
    again = d.push(adc, [3])[0]
    assert np.array_equal(again, got[3])
    d.close()


def test_rx_ddc_feeds_the_unpack_and_fir(gpu_ctx, oracle):
    """DDC records -> data-pump unpack -> CFastFIR with the reference's conventions (positive
    phase increment, rx_sound_cmd.cpp:47; I/Q swapped in the unpack, data_pump.cpp:196-201):
    a tone 1 kHz above the tuned frequency comes out in the USB passband, not in the LSB one."""
    from flydog_sdr_gps_amd import FastFir, snd
    n = RX_DECIM * 1100
    t = np.arange(n)
    adc = np.rint(20000 * np.cos(2 * np.pi * (0.0371 + 1000 / 125e6) * t)).astype(np.int16)
    inc = rx_phase_inc(0.0371 * 125e6)
    d = RxDdc(gpu_ctx, nchan=1, max_samples=n)
    d.set_freq(0, inc)
    raw = d.push(adc, [0])[0]
    x = snd.unpack(gpu_ctx, raw, raw.size // 6, 1)[0]
    f = FastFir(gpu_ctx, nchan=1, max_in=x.size)
    f.setup(0, 300.0, 2700.0, 0.0, 125e6 / RX_DECIM)
    y = f.process(0, x)
    assert y.size == 1024
    p_in = np.mean(np.abs(x[200:]) ** 2)
    p_out = np.mean(np.abs(y[512:]) ** 2)
    assert 0.8 < p_out / p_in < 1.2                          # the tone is inside the passband
    f.setup(0, -2700.0, -300.0, 0.0, 125e6 / RX_DECIM)      # and rejected by the LSB filter
    f.reset(0)
    y2 = f.process(0, x)
    assert np.mean(np.abs(y2[512:]) ** 2) < 1e-4 * p_in
    d.close()
    f.close()


def test_many_small_decimations_take_the_staged_strobe_path(gpu_ctx, oracle):
    """26 channels, 24 of them with R = 2, 4, 8, on 2^20 + 37 samples (8193 runs of 128): 24 x 8193
    runs >= two waves per SIMD, so pass B uses its LDS-staged, transposed strobe stores for them
    (the last, short run of each channel and the second push -- decimation counter no longer
    aligned -- take the plain path).  Bit-exact per channel, state carried."""
    rng = np.random.default_rng(31)
    l2 = [1, 2, 3] * 8 + [4, 6]
    nch = len(l2)
    incs = [int(rng.integers(0, 1 << 48)) for _ in range(nch)]
    d = Ddc(gpu_ctx, nchan=nch, max_samples=(1 << 20) + 64)
    try:
        for ch in range(nch):
            d.set_wf(ch, incs[ch], 1 << l2[ch])
        states = [None] * nch
        for n in ((1 << 20) + 37, (1 << 19) + 5):
            t = np.arange(n)
            adc = np.clip(np.rint(9000 * np.cos(2 * np.pi * 0.0123 * t) + rng.normal(0, 200, n)), -32768, 32767).astype(np.int16)
            got = d.push(adc, list(range(nch)))
            for ch in range(nch):
                want, states[ch] = oracle.ddc_wf(adc, incs[ch], l2[ch], states[ch])
                assert np.array_equal(got[ch], want), (ch, l2[ch], n)
    finally:
        d.close()


# ---- the other RX instances: rx3 (wide, 20.25 kHz) and rx14 (17-tap CICF) ----------------------
from flydog_sdr_gps_amd.ddc import RX_14, RX_DECIM_WIDE, RX_STD, RX_WIDE   # noqa: E402


@pytest.mark.parametrize("mode", [RX_WIDE, RX_14, RX_STD])
def test_rx_ddc_instances_bit_exact_ragged_retune_multichannel(gpu_ctx, oracle, mode):
    """kiwi.config:101-105 / fir_iq.sv:39-123: 1543 x 2 x 2 with the RX_CFG == 3 taps, and 1736 x 3 x 2
    with the 17-tap RX_CFG == 14 filter, bit-exact against the sequential oracle for ragged pushes,
    a mid-stream retune, several channels and a reset."""
    decim = {RX_STD: RX_DECIM, RX_WIDE: RX_DECIM_WIDE, RX_14: RX_DECIM}[mode]
    nch, n = 5, decim * 36 + 777
    adc = adc_stream(n, seed=31 + mode, tones=((0.0912, 14000.0), (0.0912 + 3000 / 125e6, 1500.0), (0.27, 6000.0)))
    d = RxDdc(gpu_ctx, nchan=nch, max_samples=n, mode=mode)
    assert d.decim == decim == oracle.ddc_rx_decim(mode)
    incs = [(-rx_phase_inc((0.0912 + 2e-6 * ch) * 125e6)) & ((1 << 48) - 1) for ch in range(nch)]
    inc2 = (-rx_phase_inc(0.0912 * 125e6 + 2200)) & ((1 << 48) - 1)
    for ch in range(nch):
        d.set_freq(ch, incs[ch])
    chans = list(range(nch))
    sts, parts, wants, pos = [None] * nch, [[] for _ in chans], [[] for _ in chans], 0
    for k, step in enumerate((1, decim - 1, decim, 2 * decim + 1, 63, 100003, 5, n)):
        if pos >= n:
            break
        if k == 4:
            d.set_freq(2, inc2)                       # retune one channel: its filters keep running
            incs[2] = inc2
        seg = adc[pos:pos + step]
        pos += len(seg)
        got = d.push(seg, chans)
        for ch in chans:
            w, sts[ch] = oracle.ddc_rx(seg, incs[ch], sts[ch], mode=mode)
            parts[ch].append(got[ch])
            wants[ch].append(w)
    for ch in chans:
        g, w = np.concatenate(parts[ch]), np.concatenate(wants[ch])
        assert g.size == w.size and g.size >= 6 * 35, (ch, g.size, w.size)
        assert np.array_equal(g, w), (mode, ch)
    d.reset(1)
    d.set_freq(1, incs[1])
    again = d.push(adc[:decim * 20], [1])[0]
    assert np.array_equal(again, oracle.ddc_rx(adc[:decim * 20], incs[1], mode=mode)[0])
    d.close()


def test_rx_ddc_wide_mode_differs_and_rejects_bad_mode(gpu_ctx):
    from flydog_sdr_gps_amd import KiwiGpuError
    with pytest.raises(KiwiGpuError):
        RxDdc(gpu_ctx, nchan=1, max_samples=1 << 16, mode=7)
    d = RxDdc(gpu_ctx, nchan=1, max_samples=1 << 20, mode=RX_WIDE)
    with pytest.raises(KiwiGpuError):
        d.outputs(0, RX_DECIM_WIDE * 10)                    # no frequency set yet
    d.set_freq(0, 12345)
    assert d.outputs(0, RX_DECIM_WIDE * 10) == 10           # one record per 6172 ADC samples
    d.close()


def _state_after(oracle, adc, inc, log2r, st=None):
    return oracle.ddc_wf(adc, inc, log2r, st)


def test_deferred_output_stage_pipelines_pushes_bit_exact(gpu_ctx, oracle):
    """kg_ddc_wf_set_deferred: the output stage of a push (bypass, run-total prefix, combs) runs on a stream of the object
    while the context's stream already carries the next push's run passes, the two buffer sets alternating.  Twelve
    back-to-back device pushes of ragged lengths WITHOUT any host synchronisation in between, every push into its own
    output region, one join at the end: every channel's concatenated output must equal the oracle's on the whole stream,
    bit for bit -- BASELINE configs[2]'s decimation set (bypass, staged small decimations, 64- and 96-bit run passes)."""
    zooms = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14]
    lens = [1 << 18, (1 << 18) + 8, 70001, 1 << 17, 64, 65, 1 << 18, 99999, 4097, 1 << 18, 31, 1 << 16]
    n = sum(lens)
    adc = adc_stream(n, seed=21, tones=((0.031, 8000.0), (0.1234, 2000.0), (0.3, 300.0)))
    C = len(zooms)
    d = Ddc(gpu_ctx, nchan=C, max_samples=max(lens))
    incs = [inc_for(0.03 + 0.007 * ch) for ch in range(C)]
    d_adc = gpu_ctx.alloc(adc.nbytes)
    stride = max(lens) + 8
    d_out = gpu_ctx.alloc(len(lens) * C * stride * 4)
    try:
        for ch, z in enumerate(zooms):
            d.set_wf(ch, incs[ch], 1 << max(z - 1, 0))
        d.set_deferred(True)
        gpu_ctx.upload(d_adc, adc)
        pos, nouts = 0, []
        for k, ln in enumerate(lens):
            nouts.append(d.push_dev(d_adc + 2 * pos, ln, list(range(C)), d_out + 4 * k * C * stride, stride))
            pos += ln
        d.join()                                                   # the context's stream waits for the last output stage
        host = np.zeros((len(lens), C, stride, 2), np.int16)
        gpu_ctx.download(d_out, host)
        for ch, z in enumerate(zooms):
            want, _ = oracle.ddc_wf(adc, incs[ch], max(z - 1, 0))
            got = np.concatenate([host[k, ch, :int(nouts[k][ch])] for k in range(len(lens))])
            assert got.shape == want.shape and np.array_equal(got, want), (ch, z)
        # back to the in-line mode on the same object: state carries over
        d.set_deferred(False)
        more = adc_stream(50_000, seed=22)
        got2 = d.push(more, list(range(C)))
        for ch, z in enumerate(zooms):
            want, _ = oracle.ddc_wf(np.concatenate([adc, more]), incs[ch], max(z - 1, 0))
            assert np.array_equal(got2[ch], want[want.shape[0] - got2[ch].shape[0]:]), (ch, z)
    finally:
        gpu_ctx.free(d_adc)
        gpu_ctx.free(d_out)
        d.close()


@pytest.mark.parametrize("deferred", [False, True])
def test_channels_of_different_age_are_rebased(gpu_ctx, oracle, deferred):
    """The kernels take ONE samples-since-reference count per push.  Channels whose reference points differ -- one pushed
    alone for a while, one retuned, one reset, one given a new phase mid-stream -- are re-based on the next joint push;
    every channel must still equal the oracle fed ITS history."""
    n = 40_000
    adc = adc_stream(4 * n, seed=31)
    a, b, c, e = adc[:n], adc[n:2 * n], adc[2 * n:3 * n], adc[3 * n:]
    incs = [inc_for(0.05), inc_for(0.11), inc_for(0.2), inc_for(0.31)]
    l2 = [4, 1, 9, 0]
    d = Ddc(gpu_ctx, nchan=4, max_samples=n)
    try:
        for ch in range(4):
            d.set_wf(ch, incs[ch], 1 << l2[ch])
        d.set_deferred(deferred)
        st = [None] * 4
        want = [[] for _ in range(4)]

        def ref(ch, x):
            w, st[ch] = oracle.ddc_wf(x, incs[ch], l2[ch], st[ch])
            want[ch].append(w)
            return w
        g = d.push(a, [0, 1, 2, 3])
        for ch in range(4):
            assert np.array_equal(g[ch], ref(ch, a)), ("all", ch)
        g = d.push(b, [1, 3])                                      # channels 0 and 2 sit this one out (their NCOs do not advance)
        for i, ch in enumerate((1, 3)):
            assert np.array_equal(g[i], ref(ch, b)), ("subset", ch)
        d.reset(2)                                                 # CIC registers cleared, phase kept
        ph2 = (n * incs[2]) & ((1 << 48) - 1)
        st[2] = oracle.DdcWfState(); st[2].phase = ph2
        d.set_phase(1, 12345678901)                                # a new phase for channel 1; its filters keep running
        st[1].phase = 12345678901
        incs[0] = inc_for(0.07)                                    # retune channel 0: resets it
        d.set_wf(0, incs[0], 1 << l2[0])
        st[0] = None
        g = d.push(c, [0, 1, 2, 3])                                # four different ages -> one re-base
        for ch in range(4):
            assert np.array_equal(g[ch], ref(ch, c)), ("rebased", ch)
        g = d.push(e, [3, 2, 1, 0])                                # and on, in another list order
        for i, ch in enumerate((3, 2, 1, 0)):
            assert np.array_equal(g[i], ref(ch, e)), ("after", ch)
    finally:
        d.close()


def _capture_ref(oracle, adc, inc, log2r, phase0, max_out):
    """CmdWFReset (CIC registers and decimation counter cleared, NCO running on at phase0) + one-shot sampler."""
    st = oracle.DdcWfState()
    st.phase = phase0 & ((1 << 48) - 1)
    need = min(adc.size, max_out << log2r)
    w, _ = oracle.ddc_wf(adc[:need], inc, log2r, st)
    return w[:max_out]


@pytest.mark.parametrize("deferred", [False, True])
def test_capture_is_reset_plus_one_shot_sampler(gpu_ctx, oracle, deferred):
    """kg_ddc_wf_capture_dev = the reference's non-overlapped frame (rx/rx_waterfall.cpp:1005-1041,
    verilog/rx/waterfall_1cic.v:45-47): per block every channel's CICs restart, the NCO runs on, the first 8192 outputs are
    kept.  Three consecutive blocks of 2^21 samples, decimations 1 .. 512 (zooms 1 .. 10, the receivers' set) plus one too
    slow to fill the sampler (R = 2048: 1024 outputs per block): every block's capture against the oracle started from
    a cleared filter at that block's NCO phase -- bit for bit.  Then a continuous push on the same channels: it starts
    from the reset state (the reference resets when it changes sampler mode), NCO still running."""
    n = 1 << 21
    log2rs = [0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11]
    C = len(log2rs)
    adc = adc_stream(4 * n, seed=41, tones=((0.031, 8000.0), (0.1234, 2000.0), (0.3, 300.0)))
    incs = [inc_for(0.03 + 0.011 * ch) for ch in range(C)]
    d = Ddc(gpu_ctx, nchan=C, max_samples=n)
    try:
        for ch in range(C):
            d.set_wf(ch, incs[ch], 1 << log2rs[ch])
        d.set_deferred(deferred)
        for k in range(3):
            blk = adc[k * n:(k + 1) * n]
            got = d.capture(blk, list(range(C)), 8192)
            for ch in range(C):
                want = _capture_ref(oracle, blk, incs[ch], log2rs[ch], k * n * incs[ch], 8192)
                assert got[ch].shape[0] == min(8192, n >> log2rs[ch]) == want.shape[0], (k, ch)
                assert np.array_equal(got[ch], want), (k, ch, log2rs[ch])
        # a smaller sampler and a subset of the channels in another order
        blk = adc[3 * n:3 * n + 300_000]
        sub = [7, 2, 0, 11]
        got = d.capture(blk, sub, 1000)
        for i, ch in enumerate(sub):
            want = _capture_ref(oracle, blk, incs[ch], log2rs[ch], 3 * n * incs[ch], 1000)
            assert np.array_equal(got[i], want), ("subset", ch)
        # continuous push after captures: reset filters, phase = everything pushed so far (the subset's channels are
        # 300 000 samples further on than the others: one re-base)
        tail = adc[3 * n + 300_000:3 * n + 400_000]
        got = d.push(tail, list(range(C)))
        for ch in range(C):
            st = oracle.DdcWfState()
            st.phase = ((3 * n + (300_000 if ch in sub else 0)) * incs[ch]) & ((1 << 48) - 1)
            want, _ = oracle.ddc_wf(tail, incs[ch], log2rs[ch], st)
            assert np.array_equal(got[ch], want), ("push after capture", ch)
    finally:
        d.close()


def test_push_after_a_capture_of_some_channels(gpu_ctx, oracle):
    """A capture of a SUBSET leaves those channels stale (reset on their next continuous push) and every channel of the
    object at a different distance from its reference point than before: the joint push that follows must reset the stale
    ones first and only then compare ages (found by tools/fuzz_parity.py: the reset used to come after)."""
    n = 30_000
    adc = adc_stream(3 * n, seed=51)
    incs = [inc_for(0.04), inc_for(0.17), inc_for(0.29)]
    l2 = [5, 0, 2]
    M48 = (1 << 48) - 1
    d = Ddc(gpu_ctx, nchan=3, max_samples=n)
    try:
        for ch in range(3):
            d.set_wf(ch, incs[ch], 1 << l2[ch])
        st = [None] * 3
        g = d.push(adc[:n], [0, 1, 2])
        for ch in range(3):
            w, st[ch] = oracle.ddc_wf(adc[:n], incs[ch], l2[ch], st[ch])
            assert np.array_equal(g[ch], w)
        g = d.capture(adc[n:2 * n], [2, 0], 200)                  # channel 1 sits out: its NCO does not advance
        for i, ch in enumerate((2, 0)):
            assert np.array_equal(g[i], _capture_ref(oracle, adc[n:2 * n], incs[ch], l2[ch], n * incs[ch], 200)), ch
        g = d.push(adc[2 * n:], [0, 1, 2])                        # 0 and 2: from reset at phase 2 n inc; 1: carries on from n
        for ch in range(3):
            if ch != 1:
                st[ch] = oracle.DdcWfState(); st[ch].phase = (2 * n * incs[ch]) & M48
            w, st[ch] = oracle.ddc_wf(adc[2 * n:], incs[ch], l2[ch], st[ch])
            assert np.array_equal(g[ch], w), ch
    finally:
        d.close()


def test_capture_of_a_short_block_and_odd_sizes(gpu_ctx, oracle):
    """Blocks shorter than the sampler needs (fewer than max_out outputs come back), sizes that are no multiple of a run,
    max_out = 1."""
    d = Ddc(gpu_ctx, nchan=3, max_samples=1 << 18)
    try:
        incs = [inc_for(0.07), inc_for(0.2), inc_for(0.013)]
        l2 = [3, 6, 0]
        for ch in range(3):
            d.set_wf(ch, incs[ch], 1 << l2[ch])
        pos = 0
        adc = adc_stream(400_000, seed=43)
        for ln, mo in ((70_001, 8192), (64, 8192), (131_072, 1), (99_999, 300), (12_345, 8192)):
            blk = adc[pos:pos + ln]
            got = d.capture(blk, [0, 1, 2], mo)
            for ch in range(3):
                want = _capture_ref(oracle, blk, incs[ch], l2[ch], pos * incs[ch], mo)
                assert got[ch].shape == want.shape and np.array_equal(got[ch], want), (ln, mo, ch)
            pos += ln
    finally:
        d.close()


def test_deferred_push_then_the_sample_buffer_is_recycled_at_once(gpu_ctx, oracle):
    """in deferred mode the R = 1 bypass kernel (output stream) and pass B of the small decimations (second
    stream) read the caller's samples off the context's stream.  A streaming caller refills ONE sample buffer on the
    context's stream right behind each push (kg_dev_upload is ordered there): every push must still have seen its own
    block -- ten blocks through one buffer, no host synchronisation except the upload's own, bypass + small + large
    decimations, every channel's whole output bit for bit against the oracle."""
    l2 = [0, 0, 1, 2, 3, 5, 8]
    C, n, blocks = len(l2), 1 << 18, 10
    adc = adc_stream(n * blocks, seed=77, tones=((0.031, 8000.0), (0.21, 900.0)))
    incs = [inc_for(0.02 + 0.011 * ch) for ch in range(C)]
    d = Ddc(gpu_ctx, nchan=C, max_samples=n)
    d_adc = gpu_ctx.alloc(2 * n)
    stride = n + 8
    d_out = gpu_ctx.alloc(blocks * C * stride * 4)
    try:
        for ch in range(C):
            d.set_wf(ch, incs[ch], 1 << l2[ch])
        d.set_deferred(True)
        nouts = []
        for k in range(blocks):
            gpu_ctx.upload(d_adc, adc[k * n:(k + 1) * n])           # on the context's stream, behind the push before
            nouts.append(d.push_dev(d_adc, n, list(range(C)), d_out + 4 * k * C * stride, stride))
        gpu_ctx.upload(d_adc, np.zeros(n, np.int16))                # ... and once more behind the last push
        d.join()
        host = np.zeros((blocks, C, stride, 2), np.int16)
        gpu_ctx.download(d_out, host)
        for ch in range(C):
            want, _ = oracle.ddc_wf(adc, incs[ch], l2[ch])
            got = np.concatenate([host[k, ch, :int(nouts[k][ch])] for k in range(blocks)])
            assert got.shape == want.shape and np.array_equal(got, want), ch
    finally:
        gpu_ctx.free(d_adc)
        gpu_ctx.free(d_out)
        d.close()


def test_step_call_mixes_sampler_modes_and_row_offsets(gpu_ctx, oracle):
    """kg_ddc_wf_step_dev (round 5): ONE call in which some entries run the continuous sampler (max_out 0) and the others
    are captured (CmdWFReset + one-shot of max_out pairs), each writing at its own offset into its row.  Three blocks; the
    continuous entries' outputs concatenate to the oracle's on the whole stream, a captured entry's equal a fresh filter's
    at the block's NCO phase."""
    import ctypes as C
    from flydog_sdr_gps_amd._lib import check, ptr
    l2 = [0, 2, 5, 7, 9, 3, 0, 6]
    mo = [0, 0, 0, 0, 300, 8192, 1000, 0]                           # 0: continuous
    nch, n, blocks = len(l2), 1 << 19, 3
    adc = adc_stream(n * blocks, seed=91, tones=((0.05, 7000.0), (0.33, 500.0)))
    incs = [inc_for(0.013 + 0.017 * ch) for ch in range(nch)]
    d = Ddc(gpu_ctx, nchan=nch, max_samples=n)
    stride = n + 64
    d_adc = gpu_ctx.alloc(adc.nbytes)
    d_out = gpu_ctx.alloc(nch * stride * 4)
    chans = np.arange(nch, dtype=np.int32)
    try:
        for ch in range(nch):
            d.set_wf(ch, incs[ch], 1 << l2[ch])
        gpu_ctx.upload(d_adc, adc)
        cont = {ch: [] for ch in range(nch) if mo[ch] == 0}
        for k in range(blocks):
            off = np.array([(7 * ch + 16 * k) & ~1 for ch in range(nch)], np.int64)        # any (even) offset inside the row
            nouts = np.zeros(nch, np.int64)
            check(d.lib.kg_ddc_wf_step_dev(d.h, ptr(int(d_adc + 2 * k * n)), n, ptr(chans), nch, ptr(int(d_out)), stride, ptr(off),
                                           ptr(np.array(mo, np.int64)), ptr(nouts)), "kg_ddc_wf_step_dev")
            gpu_ctx.sync()
            host = np.zeros((nch, stride, 2), np.int16)
            gpu_ctx.download(d_out, host)
            blk = adc[k * n:(k + 1) * n]
            for ch in range(nch):
                got = host[ch, int(off[ch]):int(off[ch]) + int(nouts[ch])]
                if mo[ch] == 0:
                    cont[ch].append(got.copy())
                else:
                    want = _capture_ref(oracle, blk, incs[ch], l2[ch], k * n * incs[ch], mo[ch])
                    assert got.shape == want.shape and np.array_equal(got, want), (k, ch)
        for ch, parts in cont.items():
            want, _ = oracle.ddc_wf(adc, incs[ch], l2[ch])
            got = np.concatenate(parts)
            assert got.shape == want.shape and np.array_equal(got, want), ch
        # an offset that leaves no room for the entry's outputs is refused, nothing enqueued
        bad = np.full(nch, stride - 8, np.int64)
        rc = d.lib.kg_ddc_wf_step_dev(d.h, ptr(int(d_adc)), n, ptr(chans), nch, ptr(int(d_out)), stride, ptr(bad),
                                      ptr(np.array(mo, np.int64)), None)
        assert rc < 0
    finally:
        gpu_ctx.free(d_adc)
        gpu_ctx.free(d_out)
        d.close()


def test_outputs_predictor_after_a_capture(gpu_ctx):
    """after a capture the next continuous push resets the channel, so kg_ddc_wf_outputs must answer n >> log2 R
    (not what the un-reset counter would give)."""
    d = Ddc(gpu_ctx, nchan=1, max_samples=1 << 16)
    try:
        d.set_wf(0, inc_for(0.1), 64)
        adc = adc_stream(1000, seed=5)
        d.capture(adc, [0], 8192)                                   # 1000 samples at R = 64: the counter would stand at 40
        assert d.outputs(0, 100) == 1
        got = d.push(adc_stream(100, seed=6), [0])
        assert got[0].shape[0] == 1
    finally:
        d.close()
