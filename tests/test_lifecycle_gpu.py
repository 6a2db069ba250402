"""Lifecycle: every object of the C ABI created, used once and destroyed, many times over; the
device's free memory must come back (no leak in create/destroy or in the per-call scratch)."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import (Adpcm, Aperture, Context, Ddc, FastFir, Post, RxDdc, Searcher, Waterfall, WfParams,
                                handoff, post, prn, sats, synth, wf, wire)

pytestmark = pytest.mark.gpu


def use_everything(ctx):
    s = Searcher(ctx, max_blocks=2)
    s.set_code(0, prn.cacode(*sats.SATS[0][1:3]))
    s.set_code(1, np.arange(4092, dtype=np.uint8) & 1, boc=True)      # a 16368-lag SV: the 512-thread correlator and its layout
    s.sample_iq16(synth.config1_iq16(seed=1), block=0)
    s.correlate_async([0, 1], nblocks=1)
    s.fetch()
    s.close()
    ctx.mark(7)                                                        # the profiling marker kernel
    ctx.sync()
    w = Waterfall(ctx, nchan=2)
    w.set_tables()
    p = WfParams.for_zoom(3, 1000.0)
    w.set_channel(0, p)
    row = w.frames([0], synth.wf_iq_frame(seed=2)[None])[0]
    w.close()
    d = Ddc(ctx, nchan=2, max_samples=1 << 16)
    d.set_wf(0, p.i_offset, p.decim)
    d.push(np.zeros(1 << 14, np.int16), [0])
    d.close()
    r = RxDdc(ctx, nchan=2, max_samples=1 << 16)
    r.set_freq(0, 12345)
    r.push(np.zeros(1 << 15, np.int16), [0])
    r.close()
    f = FastFir(ctx, nchan=2, max_in=1024)
    f.setup(0, 300.0, 2700.0, 0.0, 12000.0)
    y = f.process(0, np.ones(1024, np.complex64))
    f.close()
    P = Post(ctx, nchan=2)
    P.set_agc(0, True, False, -100, 50, 6, 1000, 12000.0)
    P.set_mode(0, post.MODE_SSB)
    s16, _, _ = P.process([0], y[None, :512])
    P.smeter([0])
    P.close()
    A = Adpcm(ctx, nchan=2)
    A.encode([0], s16)
    A.close()
    wire.wf_packets(ctx, row[None], [(1, 3, 5, True)])
    a = Aperture(ctx, nchan=2)
    a.update([0], row[None], [(handoff.MMA, 8.0, True, False)])
    a.report([0])
    a.close()


def test_create_use_destroy_returns_device_memory():
    ctx = Context(0)
    try:
        use_everything(ctx)                       # first round: code objects, lazily grown scratch
        free0, total = ctx.mem_info()
        for _ in range(15):
            use_everything(ctx)
        free1, _ = ctx.mem_info()
        assert total > 200 * 2 ** 30              # an MI355X: 288 GB
        assert free0 - free1 < 8 * 2 ** 20, "device memory shrank by %.1f MiB over 15 rounds" % ((free0 - free1) / 2 ** 20)
    finally:
        ctx.close()


def test_contexts_come_and_go():
    free = []
    for _ in range(6):
        c = Context(0)
        s = Searcher(c, max_blocks=1)
        s.close()
        free.append(c.mem_info()[0])
        c.close()
    assert max(free) - min(free[1:]) < 64 * 2 ** 20



def test_mark_rejects_tags_out_of_range(gpu_ctx):
    from flydog_sdr_gps_amd import KiwiGpuError
    gpu_ctx.mark(1)
    gpu_ctx.mark(65535)
    gpu_ctx.sync()
    for bad in (0, -3, 65536):
        with pytest.raises(KiwiGpuError):
            gpu_ctx.mark(bad)
