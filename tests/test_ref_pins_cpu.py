"""The oracle against vectors the REFERENCE ITSELF produced (tests/golden/*_ref.*, written by
tools/make_ref_golden.py from oracle/_ref/, which oracle/build_ref.sh compiles from the reference's
own sources in place).  These are the pins DESIGN.md section 3 lists; the HIP path is compared with
the same vectors in tests/test_ref_pins_gpu.py."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_e1b_all_50_codes_match_reference_e1bcode_h(oracle):
    """gps/e1bcode.h:62-92 E1BCODE(prn) for prn 1..50 (NUM_E1B_SATS)."""
    from flydog_sdr_gps_amd import prn
    g = np.load(os.path.join(GOLD, "e1b_ref.npz"))
    chips = np.unpackbits(g["chips_packed"], axis=1)[:, :4092]
    assert chips.shape == (50, 4092)
    for i in range(50):
        h = str(g["hex"][i])
        assert np.array_equal(oracle.e1b_from_hex(h), chips[i]), "oracle, E%02d" % (i + 1)
        assert np.array_equal(prn.e1b_from_hex(h), chips[i]), "host mirror, E%02d" % (i + 1)
    # the reference's own known answers (gps/search.cpp:295,302)
    assert int("".join(map(str, chips[0, :20])), 2) == 0xf5d71
    assert int("".join(map(str, chips[1, :20])), 2) == 0x96b85


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
