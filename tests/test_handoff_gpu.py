"""GPU/host parity of the hand-off row through the C ABI: CHANNEL::Start() arithmetic (host,
double) and aperture_auto() (device) against the oracle's restatement of gps/channel.cpp:281-311
and rx/rx_waterfall.cpp:1173-1273.  Every average (the IIR's expf is the host libm's algorithm on the device, csrc/kg_libm.h) and
the band report are exact."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import Aperture, KiwiGpuError, chan_start, handoff

pytestmark = pytest.mark.gpu


def test_chan_start_matches_oracle(gpu_ctx, oracle):
    rng = np.random.default_rng(0)
    cases = [(False, 0, 1000, 0.5), (True, -12, 65000, 1.25), (False, 20, 0, 0.0), (False, -20, 16367, 3.0),
             (True, 7, 65471, 10.0), (False, 3, 5, 40.0)]
    cases += [(bool(rng.integers(0, 2)), int(rng.integers(-20, 21)), int(rng.integers(0, 65472)), float(rng.uniform(0, 5)))
              for _ in range(200)]
    for c in cases:
        g, w = chan_start(*c, lib=gpu_ctx.lib), oracle.chan_start(*c)
        assert (g.lo_rate, g.ca_rate, g.ca_pause, g.code_creep, g.lo_dop, g.ca_dop) == \
               (w.lo_rate, w.ca_rate, w.ca_pause, w.code_creep, w.lo_dop, w.ca_dop), c
    g = chan_start(False, 0, 1000, 0.5, lib=gpu_ctx.lib)
    assert (g.lo_rate, g.ca_rate, g.ca_pause, g.code_creep) == (0x40000000, 0x10000000, 16368 - 1000, 0)
    assert chan_start(True, 0, 65000, 0.0, lib=gpu_ctx.lib).ca_pause == 65472 - 65000


def rows_for(rng, k):
    base = rng.normal(100 + 10 * (k % 3), 6, 1024)
    base[200 + 50 * (k % 5):230 + 50 * (k % 5)] += 90
    if k % 4 == 0:
        base[:100] = 0                                   # masked area: -255 + cal <= -190
    return np.clip(base, 0, 255).astype(np.uint8)


def test_aperture_update_and_report(gpu_ctx, oracle):
    rng = np.random.default_rng(17)
    nchan = 12
    A = Aperture(gpu_ctx, nchan=nchan)
    want = [np.zeros(1024, np.float32) for _ in range(nchan)]
    try:
        chans = np.arange(nchan, dtype=np.int32)
        algos = [handoff.MMA, handoff.EMA, handoff.IIR]
        for it in range(9):
            rows = np.stack([rows_for(rng, ch + it) for ch in range(nchan)])
            cfgs = [(algos[ch % 3], [8.0, 4.0, 2.5][ch % 3], it == 0 or (it == 5 and ch == 3), ch >= 9)
                    for ch in range(nchan)]
            A.update(chans, rows, cfgs, waterfall_cal=-13)
            for ch in range(nchan):
                a, p, clr, af = cfgs[ch]
                want[ch] = oracle.aper_update(want[ch], rows[ch], a, p, clr, 256 if af else 0, 768 if af else 1024, -13)
            sig, noise = A.report(chans, [ch >= 9 for ch in range(nchan)])
            for ch in range(nchan):
                got = A.get(ch)
                assert np.array_equal(got.view(np.uint32), want[ch].view(np.uint32)), (it, ch, algos[ch % 3])
                # the report is exact for the averages the device holds
                af = ch >= 9
                assert (int(sig[ch]), int(noise[ch])) == oracle.aper_report(got, 256 if af else 0, 768 if af else 1024)
    finally:
        A.close()


def test_aperture_report_edges(gpu_ctx, oracle):
    """Everything masked -> -110/-120 before the -80 floor; ties go to the lower band; a lone strong
    pixel sets the signal; exact multiples of 5 and values just below them."""
    A = Aperture(gpu_ctx, nchan=4)
    try:
        rows = np.zeros((4, 1024), np.uint8)
        rows[1, :512], rows[1, 512:] = 155, 165           # two bands, 512 each: -113 -> -115, -103 -> -105
        rows[2, :] = 140; rows[2, 77] = 250               # -128 -> -130 and one pixel at -18 -> -20
        rows[3, :] = 142                                  # -126 -> -130; then cal -12 puts it on -125 exactly
        A.update([0, 1, 2, 3], rows, [(handoff.MMA, 8.0, True, False)] * 4, waterfall_cal=-13)
        sig, noise = A.report([0, 1, 2, 3])
        assert (sig[0], noise[0]) == (-80, -120)
        assert (sig[1], noise[1]) == (-80, -115)
        assert (sig[2], noise[2]) == (-20, -130)
        for ch in range(4):
            assert (int(sig[ch]), int(noise[ch])) == oracle.aper_report(A.get(ch))
        A.update([3], rows[3:4], [(handoff.MMA, 8.0, True, False)], waterfall_cal=-12)
        assert A.report([3])[1][0] == -125 and oracle.aper_report(A.get(3))[1] == -125
        with pytest.raises(KiwiGpuError):
            A.update([0, 0], rows[:2], [(handoff.MMA, 8.0, True, False)] * 2)
        with pytest.raises(KiwiGpuError):
            A.update([0], rows[:1], [(7, 8.0, False, False)])
    finally:
        A.close()
