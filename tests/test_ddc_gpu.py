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
