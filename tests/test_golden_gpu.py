"""GPU results vs the committed fixtures of tests/golden/ (made by tools/make_golden.py)."""
import os

import numpy as np
import pytest

from flydog_sdr_gps_amd import Ddc, FastFir, RxDdc, Waterfall, WfParams, snd, wf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_wf_rows(gpu_ctx):
    g = np.load(os.path.join(GOLD, "wf_golden.npz"))
    w = Waterfall(gpu_ctx, nchan=1)
    w.set_tables(wf.window_functions(), g["cic_comp"])
    for k in range(int(g["ncases"])):
        zoom, start, interp, winf, cic, inv = g["case%d_cfg" % k]
        p = WfParams.for_zoom(int(zoom), float(start), spectral_inversion=bool(inv))
        w.set_channel(0, p, interp=int(interp), window_func=int(winf), cic_comp=bool(cic), spectral_inversion=bool(inv))
        row, _, pwr_out, dB = w.debug_frame(0, g["case%d_iq" % k])
        want, want_dB = g["case%d_row" % k], g["case%d_dB" % k]
        assert np.abs(pwr_out - g["case%d_pwr_out" % k]).max() <= 1e-5 * g["case%d_pwr_out" % k].max()
        diff = np.nonzero(row != want)[0]
        assert diff.size <= 8
        for i in diff:                                      # one LSB at an (int) truncation edge
            assert abs(int(row[i]) - int(want[i])) == 1 and abs(want_dB[i] - np.rint(want_dB[i])) < 2e-3
    w.close()


def test_unpack_fir_ddc(gpu_ctx):
    g = np.load(os.path.join(GOLD, "snd_golden.npz"))
    got = snd.unpack(gpu_ctx, g["raw"], 8, 4, dc_i=0.5, dc_q=-0.25)
    assert np.array_equal(got.view(np.uint32), g["unpack_normal"].view(np.uint32))
    got = snd.unpack(gpu_ctx, g["raw"], 8, 4, dc_i=0.5, dc_q=-0.25, spectral_inversion=True)
    assert np.array_equal(got.view(np.uint32), g["unpack_inverted"].view(np.uint32))
    f = FastFir(gpu_ctx, nchan=1, max_in=256)
    f.set_coef(0, g["fir_coef"])                            # the reference's own m_pFilterCoef_CIC path
    outs, pos = [], []
    for k in range(7):
        outs.append(f.process(0, g["fir_in"][170 * k:170 * (k + 1)]))
        pos.append(f.pos(0))
    y = np.concatenate(outs)
    assert pos == list(g["fir_pos"]) and y.size == g["fir_out"].size
    assert np.abs(y - g["fir_out"]).max() <= 1e-5 * np.abs(g["fir_out"]).max()
    f.close()
    d = np.load(os.path.join(GOLD, "ddc_golden.npz"))
    for l2 in (0, 4, 11):
        e = Ddc(gpu_ctx, nchan=1, max_samples=d["adc"].size)
        e.set_wf(0, int(d["inc"]), 1 << l2)
        assert np.array_equal(e.push(d["adc"], [0])[0], d["wf_r%d" % l2])
        e.close()
    r = RxDdc(gpu_ctx, nchan=1, max_samples=d["adc_rx"].size)
    r.set_freq(0, int(d["inc_rx"]))
    assert np.array_equal(r.push(d["adc_rx"], [0])[0], d["rx_records"])
    r.close()


def test_post_golden(gpu_ctx):
    """S-meter / CAgc / detectors against the committed oracle outputs (tests/golden/post_golden.npz);
    bars as in tests/test_post_gpu.py."""
    from flydog_sdr_gps_amd import Post, post
    from tests.fixtures import arm_audio_tail
    g = np.load(os.path.join(GOLD, "post_golden.npz"))
    x, n = g["x"], g["x"].size
    P = Post(gpu_ctx, nchan=9)
    try:
        for k, args in enumerate(g["agc_args"]):
            for m, mode in enumerate((post.MODE_SSB, post.MODE_AM, post.MODE_NBFM)):
                ch = 3 * k + m
                P.set_agc(ch, *[int(v) for v in args], float(g["rate"]))
                P.set_smeter(ch, float(g["rate"])); P.set_mode(ch, mode); P.reset(ch)
                arm_audio_tail(P, ch, float(g["rate"]))
        outs = [P.process(np.arange(9), np.tile(x[i:i + 512], (9, 1))) for i in range(0, n, 512)]
        s16 = np.concatenate([o[0] for o in outs], axis=1)
        dem = np.concatenate([o[1] for o in outs], axis=1)
        agc = np.concatenate([o[2] for o in outs], axis=1)
        for k in range(3):
            want = g["agc_cpx_%d" % k]
            d = np.abs(s16[3 * k].astype(int) - g["agc_s16_%d" % k].astype(int))
            assert d.max() <= 1 and (d == 0).mean() >= 0.99
            for ch in (3 * k + 1, 3 * k + 2):
                assert np.abs(agc[ch] - want).max() <= 1e-5 * np.abs(want).max()
            zmax = 100.0 * float(np.abs(want).max())
            assert np.abs(dem[3 * k + 1] - g["am_%d" % k]).max() <= 4 * float(np.spacing(np.float32(zmax))) + 1e-5 * np.abs(g["am_%d" % k]).max()
            assert np.abs(dem[3 * k + 2] - g["nbfm_%d" % k]).max() <= 2e-5 * 8192
        avg, taps = P.smeter([0])
        assert abs(avg[0] - g["smeter"][1]) <= 1e-4 and np.abs(taps[0] - g["smeter"][2:]).max() <= 1e-4
    finally:
        P.close()


def test_wire_golden(gpu_ctx):
    """ADPCM sound coder, waterfall packets and the sound header against tests/golden/wire_golden.npz."""
    from flydog_sdr_gps_amd import Adpcm, wire
    g = np.load(os.path.join(GOLD, "wire_golden.npz"))
    A = Adpcm(gpu_ctx, nchan=1)
    try:
        enc = [A.encode([0], g["audio"][None, i:i + 512])[0] for i in range(0, g["audio"].size, 512)]
        assert np.array_equal(np.concatenate(enc), g["adpcm"]) and list(A.get_state(0)) == list(g["adpcm_state"])
    finally:
        A.close()
    pk = wire.wf_packets(gpu_ctx, np.stack([g["row"], g["row"]]), [(123456, 7, 4242, True), (123456, 7, 4242, False)])
    assert np.array_equal(pk[0], g["pkt_compressed"]) and np.array_equal(pk[1], g["pkt_raw"])
    assert np.array_equal(wire.snd_header(gpu_ctx, 0x10, 4242, -87.31), g["snd_header"])
