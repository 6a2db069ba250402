"""GPU parity of the wire formats (kg_wire) through the C ABI: IMA ADPCM sound and waterfall
coders, wf_pkt_t, sound payload and header -- integer work, bit-exact against the oracle's
restatement of rx/csdr/ima_adpcm.cpp, rx/rx_waterfall.cpp:1602-1639, rx/rx_sound.cpp:1122-1254."""
import numpy as np
import pytest

from flydog_sdr_gps_amd import Adpcm, KiwiGpuError, wire

pytestmark = pytest.mark.gpu


def audio(n, rng, kind):
    t = np.arange(n)
    if kind == 0:
        x = 9000 * np.sin(2 * np.pi * 0.013 * t) + rng.normal(0, 300, n)
    elif kind == 1:
        x = rng.normal(0, 12000, n)                                    # drives the index to 88 and the clamps
    elif kind == 2:
        x = np.where((t // 97) % 2 == 0, -32768, 32767)                # full-scale square wave
    elif kind == 3:
        x = np.zeros(n)                                                # index walks down to 0
    else:
        x = rng.integers(-3, 4, n)                                     # tiny steps
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def test_sound_adpcm_many_channels_bit_exact(gpu_ctx, oracle):
    """70 channels (more than one wave) x 5 signal kinds, four consecutive 512-sample blocks with
    the coder state kept on the device; bytes and final (index, previousValue) identical."""
    rng = np.random.default_rng(21)
    nch, n, nblk = 70, 512, 4
    x = np.stack([audio(n * nblk, rng, ch % 5) for ch in range(nch)])
    A = Adpcm(gpu_ctx, nchan=nch + 3)
    try:
        chans = np.arange(nch, dtype=np.int32)[::-1].copy() + 3        # not the identity mapping
        got = [A.encode(chans, x[:, b * n:(b + 1) * n]) for b in range(nblk)]
        for r, ch in enumerate(chans):
            st = None
            for b in range(nblk):
                want, st = oracle.adpcm_encode_i16(x[r, b * n:(b + 1) * n], st)
                assert np.array_equal(got[b][r], want), (r, b)
            assert A.get_state(int(ch)) == (st.index, st.previous)
        assert A.get_state(0) == (0, 0)                                 # untouched channel
    finally:
        A.close()


def test_sound_adpcm_channel_list_changing_between_calls(gpu_ctx, oracle):
    """The channel list of an encode call is kept on the device while it does not change; every
    call must still use its own list: same, permuted, shorter, longer, first again."""
    rng = np.random.default_rng(22)
    nch, n = 9, 256
    A = Adpcm(gpu_ctx, nchan=nch)
    st = [None] * nch
    try:
        lists = [[0, 1, 2, 3], [0, 1, 2, 3], [3, 2, 1, 0], [5, 4], list(range(nch)), [0, 1, 2, 3], [8]]
        for k, chans in enumerate(lists):
            x = np.stack([audio(n, rng, (ch + k) % 5) for ch in chans])
            got = A.encode(np.asarray(chans, np.int32), x)
            for r, ch in enumerate(chans):
                want, st[ch] = oracle.adpcm_encode_i16(x[r], st[ch])
                assert np.array_equal(got[r], want), (k, ch)
        for ch in range(nch):
            want = (st[ch].index, st[ch].previous) if st[ch] is not None else (0, 0)
            assert A.get_state(ch) == want
    finally:
        A.close()


@pytest.mark.parametrize("n", [2, 6, 170, 512, 2048])
def test_sound_adpcm_lengths_and_state(gpu_ctx, oracle, n):
    rng = np.random.default_rng(n)
    x = audio(n, rng, 1)
    A = Adpcm(gpu_ctx, nchan=2)
    try:
        A.set_state(1, 40, -1234)                                       # resume from a sent state
        got = A.encode([1], x[None, :])[0]
        want, st = oracle.adpcm_encode_i16(x, oracle.AdpcmState(40, -1234))
        assert np.array_equal(got, want) and A.get_state(1) == (st.index, st.previous)
        dec, _ = oracle.adpcm_decode_i16(got, oracle.AdpcmState(40, -1234))   # what the client reconstructs
        assert dec[-1] == st.previous
    finally:
        A.close()


def test_waterfall_packets_bit_exact(gpu_ctx, oracle):
    rng = np.random.default_rng(8)
    rows = [rng.integers(0, 256, 1024), np.zeros(1024), np.full(1024, 255),
            np.clip(120 + 60 * np.sin(np.arange(1024) / 9.0) + rng.normal(0, 6, 1024), 0, 255),
            np.where(np.arange(1024) % 64 < 32, 0, 255), rng.integers(40, 60, 1024)]
    rows = np.stack([np.asarray(r).astype(np.uint8) for r in rows] * 2)
    infos = [(rng.integers(0, 2 ** 31), z % 15, 1000 + z, z % 2 == 0) for z in range(len(rows))]
    pk = wire.wf_packets(gpu_ctx, rows, infos)
    for r, (info, p) in enumerate(zip(infos, pk)):
        want = oracle.wf_packet(rows[r], *info)
        comp = bool(info[3]) and info[1] != 0            # rx_waterfall.cpp:1283-1285: a row at zoom 0 is never compressed (z = 0 is in the list)
        assert p.size == want.size == 16 + (517 if comp else 1024)
        assert np.array_equal(p, want), r
        assert bytes(p[:4]) == b"W/F " and int.from_bytes(bytes(p[8:12]), "little") == info[1] | (0x10000 if comp else 0)
        if comp:                                                        # the client's decoder gets the row back, roughly
            dec, _ = oracle.adpcm_decode_u8(p[16:])
            assert dec.size == 1034


def test_sound_payload_and_header(gpu_ctx, oracle):
    rng = np.random.default_rng(3)
    x = rng.integers(-32768, 32768, (5, 512)).astype(np.int16)
    le = wire.snd_payload(gpu_ctx, x, True)
    be = wire.snd_payload(gpu_ctx, x, False)
    assert np.array_equal(le, x.view(np.uint8).reshape(5, -1))
    assert np.array_equal(be, x.byteswap().view(np.uint8).reshape(5, -1))
    for flags, seq, dbm in ((0x10, 1, -73.26), (0x98, 0xFFFFFFFF, 10.0), (0, 0x01020304, -200.0), (0x40, 7, 3.4),
                            (1, 2, -126.96), (1, 2, -0.05)):
        assert np.array_equal(wire.snd_header(gpu_ctx, flags, seq, dbm), oracle.snd_header(flags, seq, dbm))
    h = wire.snd_header(gpu_ctx, 0x10, 0x01020304, -73.26)
    assert bytes(h[:3]) == b"SND" and list(h[4:8]) == [4, 3, 2, 1] and (int(h[8]) << 8 | int(h[9])) == 537


def test_iq_mode_sound_payload(gpu_ctx, oracle):
    """The IQ modes' payload (rx/rx_sound.cpp:1076-1096): (s2_t) re, (s2_t) im per sample, little-endian or network order; values on
    and beyond the int16 range, negative fractions (truncation toward zero), NaN / infinity (the x86 conversion's indefinite value)."""
    rng = np.random.default_rng(4)
    x = (rng.normal(0, 9000, (4, 512)) + 1j * rng.normal(0, 9000, (4, 512))).astype(np.complex64)
    x[0, :8] = [32767.9, -32768.9, 32768.0, -32769.0, -0.9, 0.9, 70000.5 - 70000.5j, -1e12 + 1e12j]
    x[1, 0] = complex(np.nan, np.inf)
    for le in (True, False):
        got = wire.snd_iq_payload(gpu_ctx, x, le)
        for r in range(4):
            assert np.array_equal(got[r], oracle.snd_iq_payload(x[r], le)), (le, r)
    le = wire.snd_iq_payload(gpu_ctx, x[2:3], True)[0].view("<i2").reshape(-1, 2)
    assert np.array_equal(le[:, 0], np.trunc(x[2].real).astype(np.int16)) and np.array_equal(le[:, 1], np.trunc(x[2].imag).astype(np.int16))


def test_argument_errors(gpu_ctx):
    A = Adpcm(gpu_ctx, nchan=2)
    try:
        with pytest.raises(KiwiGpuError):
            A.encode([0], np.zeros((1, 511), np.int16))                 # odd length
        with pytest.raises(KiwiGpuError):
            A.encode([0, 0], np.zeros((2, 8), np.int16))
        with pytest.raises(KiwiGpuError):
            A.set_state(0, 89, 0)
        with pytest.raises(KiwiGpuError):
            A.get_state(5)
    finally:
        A.close()
