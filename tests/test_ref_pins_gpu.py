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
    """m_Agc[ch].SetParameters / ProcessData (complex and mono16) / GetDelaySamples sequences.  The GPU
    differs from the reference's x86 build only through log10f / powf (device libm): complex outputs
    within 1e-5 of full scale, mono16 within one LSB and >= 99 % identical, delays identical."""
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
                    assert np.abs(agc[0] - w).max() <= 1e-5 * 32767.0, (name, np.abs(agc[0] - w).max())
                else:
                    w = want[wpos:wpos + n]
                    wpos += n
                    d = np.abs(s16[0].astype(np.int32) - w.astype(np.int32))
                    assert d.max() <= 1 and np.mean(d == 0) >= 0.99, (name, d.max(), np.mean(d == 0))
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
