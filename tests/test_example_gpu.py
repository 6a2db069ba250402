"""The compiled C++ host caller (examples/search_dropin.cpp: SearchInit -> per-SV Sample / Correlate /
ChanStart through include/kiwigpu.h, the INTEGRATION.md section 1 sequence) gives the results the
Python mirror gives for BASELINE configs[0]."""
import os
import re
import subprocess

import numpy as np
import pytest

from flydog_sdr_gps_amd import Searcher, handoff, prn, sats, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "search_dropin")


def run_example(tmp_path, bits, *args):
    f = tmp_path / "bits.bin"
    f.write_bytes(np.asarray(bits, np.uint8).tobytes())
    out = subprocess.run([EXE, str(f)] + list(args), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = {}
    for m in re.finditer(r"sat (\d+) prn (\d+) snr ([\d.]+) lo_shift (-?\d+) ca_shift (-?\d+) lo_rate (0x[0-9a-f]+) "
                         r"ca_rate (0x[0-9a-f]+) ca_pause (\d+)", out.stdout):
        rows[int(m.group(1))] = (float(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6), 16),
                                 int(m.group(7), 16), int(m.group(8)))
    lat = re.search(r"per-SV latency us: median ([\d.]+) min ([\d.]+)", out.stdout)
    return rows, (float(lat.group(1)), float(lat.group(2))), out.stdout


def test_cpp_caller_equals_python_mirror_config0(gpu_ctx, tmp_path):
    assert os.path.exists(EXE), "examples/search_dropin is not built (run __graft_entry__.build())"
    bits = synth.config0_bits()
    svs = [0, 5, 11]                                   # PRN 1 present; 6 and 12 absent
    s = Searcher(gpu_ctx)
    for sv in svs:
        s.set_code(sv, prn.cacode(sats.SATS[sv][1], sats.SATS[sv][2]))
    want = s.search(svs, packed=bits)
    s.close()
    for mode in ((), ("--batch",)):
        rows, lat, text = run_example(tmp_path, bits, "--sats", ",".join(map(str, svs)), *mode)
        assert set(rows) == set(svs), text
        for w in want:
            snr, lo, ca, lo_rate, ca_rate, pause = rows[w.sat]
            assert abs(snr - w.snr) <= 1e-3 * max(1.0, w.snr)          # printed with 4 decimals
            if w.snr >= 16:
                assert (lo, ca) == (w.lo_shift, w.ca_shift) == (6, 4808)
                cs = handoff.chan_start(False, lo, ca, 0.0)
                assert lo_rate == cs.lo_rate                          # the NCO word does not depend on the age
                assert ca_rate == cs.ca_rate and 0 < pause <= 16368
            else:
                assert (lo_rate, ca_rate, pause) == (0, 0, 0)          # ChanStart() not reached (:596-598)
        assert 0 < lat[1] <= lat[0] < 5e4


def test_cpp_caller_per_sv_latency_32_svs(gpu_ctx, tmp_path):
    """The reference's calling pattern -- one SV per Correlate() call, the SV list changing on every
    call -- must not stall on table rebuilds: a per-SV call (enqueue, poll, fetch) stays far below a
    millisecond."""
    bits = synth.config0_bits()
    rows, lat, text = run_example(tmp_path, bits, "--sats", ",".join(map(str, range(32))), "--repeat", "5")
    assert len(rows) == 32 and rows[0][1:3] == (6, 4808)
    print(text.splitlines()[-1])
    assert lat[0] < 500.0, text


# ---- the waterfall seam from C++ ---------------------------------------------------------------------------------
WF_EXE = os.path.join(ROOT, "examples", "waterfall_dropin")


def write_wf_tables(path, params, interp, window_func):
    """tables.bin of examples/waterfall_dropin.cpp: the arrays a reference build would hand over (WF_SHMEM->window_function,
    ->CIC_comp, each wf_inst_t's fft2wf_map / drop_sample / fft_scale / fft_scale_div2), here from the host mirror wf.py."""
    import struct
    from flydog_sdr_gps_amd import wf
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(params)))
        f.write(np.ascontiguousarray(wf.window_functions(), np.float32).tobytes())
        f.write(np.ascontiguousarray(wf.cic_comp_table(), np.float32).tobytes())
        for p in params:
            m, d = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, False)
            m4096 = np.zeros(4096, np.uint16)
            m4096[:len(m)] = m
            scale = np.full(1024, p.fft_scale, np.float32)
            f.write(struct.pack("<8if", p.zoom, window_func, interp, 1, 0, p.fft_used, p.plot_width, p.plot_width_clamped,
                                float(p.fft_offset)))
            f.write(struct.pack("<QiI", p.i_offset & ((1 << 48) - 1), p.decim, int(p.start) & 0xFFFFFFFF))
            f.write(m4096.tobytes())
            f.write(np.ascontiguousarray(d, np.uint16).tobytes())
            f.write(scale.tobytes())
            f.write((scale / np.float32(2)).astype(np.float32).tobytes())


def test_cpp_waterfall_caller_against_the_oracle(gpu_ctx, oracle, tmp_path):
    """examples/waterfall_dropin.cpp -- c2s_waterfall_init / CmdSetWFFreq + CmdSetWFDecim / sample_wf (CmdWFReset + one-shot
    sampler) / compute_frame / the W/F packet, through include/kiwigpu.h from plain C++ -- on two consecutive ADC blocks and
    six channels (zooms 0 .. 9): every row and every packet against the ORACLE's chain on the same samples (the DDC model
    started from a cleared filter at each block's NCO phase, sample_wf window + compute_frame, wf_pkt_t with ADPCM)."""
    from flydog_sdr_gps_amd import WfParams, wf
    from tests.test_ddc_gpu import adc_stream
    from tests.test_wf_gpu import check_row, db_bound, oracle_frame
    assert os.path.exists(WF_EXE), "examples/waterfall_dropin is not built (run __graft_entry__.build())"
    zooms = [0, 1, 3, 5, 7, 9]
    params = [WfParams.for_zoom(z, 2.0e6 + 0.7e6 * k, adc_clock=66.6666e6, ui_srate=30.0e6) for k, z in enumerate(zooms)]
    n, steps = 8192 * 256, 2                                  # zoom 9 (R = 256) just fills the sampler
    adc = adc_stream(n * steps, seed=61, tones=((0.031, 8000.0), (0.1234, 2000.0), (0.3, 300.0)))
    tb, ab, ob = tmp_path / "tables.bin", tmp_path / "adc.bin", tmp_path / "out.bin"
    write_wf_tables(tb, params, wf.WF_MAX, wf.WINF_HANNING)
    ab.write_bytes(adc.tobytes())
    r = subprocess.run([WF_EXE, str(tb), str(ab), str(ob), str(steps)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rec = np.dtype([("row", np.uint8, 1024), ("nb", "<i4"), ("pkt", np.uint8, 16 + 10 + 1024)])
    out = np.frombuffer(ob.read_bytes(), rec).reshape(steps, len(zooms))
    tables = (wf.window_functions(), wf.cic_comp_table())
    for s in range(steps):
        blk = adc[s * n:(s + 1) * n]
        for c, p in enumerate(params):
            st = oracle.DdcWfState()
            st.phase = (s * n * p.i_offset) & ((1 << 48) - 1)
            iq, _ = oracle.ddc_wf(blk[:8192 * p.decim], p.i_offset, int(np.log2(p.decim)), st)
            assert iq.shape[0] == 8192
            w_out, _, w_po, w_dB = oracle_frame(oracle, tables, iq, p, wf.WF_MAX, wf.WINF_HANNING, True, False, False)
            check_row(out[s, c]["row"], w_out, w_dB, db_bound(w_po))
            want_pkt = oracle.wf_packet(out[s, c]["row"], int(p.start), p.zoom, s, True)
            assert int(out[s, c]["nb"]) == want_pkt.size
            assert np.array_equal(out[s, c]["pkt"][:want_pkt.size], want_pkt), (s, c)


# ---- a bank of receivers from C++ ---------------------------------------------------------------------------------
BANK_EXE = os.path.join(ROOT, "examples", "rxbank_dropin")


def write_bank_tables(path, mix, n, fs):
    """tables.bin of examples/rxbank_dropin.cpp: write_wf_tables' arrays plus, per receiver, the sampler mode, the audio
    NCO word and the passband."""
    import struct
    from flydog_sdr_gps_amd import wf
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", len(mix), n))
        f.write(np.ascontiguousarray(wf.window_functions(), np.float32).tobytes())
        f.write(np.ascontiguousarray(wf.cic_comp_table(), np.float32).tobytes())
        for p, ov, rx_inc in mix:
            m, d = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, False)
            m4096 = np.zeros(4096, np.uint16)
            m4096[:len(m)] = m
            scale = np.full(1024, p.fft_scale, np.float32)
            f.write(struct.pack("<8if", p.zoom, wf.WINF_HANNING, wf.WF_MAX, 1, int(ov), p.fft_used, p.plot_width, p.plot_width_clamped,
                                float(p.fft_offset)))
            f.write(struct.pack("<QiiI", p.i_offset & ((1 << 48) - 1), p.decim, int(ov), int(p.start) & 0xFFFFFFFF))
            f.write(m4096.tobytes())
            f.write(np.ascontiguousarray(d, np.uint16).tobytes())
            f.write(scale.tobytes())
            f.write((scale / np.float32(2)).astype(np.float32).tobytes())
            f.write(struct.pack("<Q3f", int(rx_inc), 300.0, 2700.0, fs))


def test_cpp_bank_caller_equals_the_python_mirror(tmp_path):
    """examples/rxbank_dropin.cpp -- kg_rxbank_create, the per-seam setters on the bank's objects, ONE kg_rxbank_step per
    step, kg_rxbank_poll, the results read back -- on 12 receivers of SURVEY's mix (three of them overlapped), three steps:
    every row, packet, mono16 block and ADPCM payload equals what flydog_sdr_gps_amd.rxbank.RxBank (checked stage by stage
    against the oracle in tests/test_receivers_gpu.py) produces from the same stream, and the host's share of a step is
    small."""
    from flydog_sdr_gps_amd import synth
    from flydog_sdr_gps_amd.rxbank import RxBank, survey_mix
    assert os.path.exists(BANK_EXE), "examples/rxbank_dropin is not built (run __graft_entry__.build())"
    NR, n, steps = 12, 1 << 22, 3
    mix = survey_mix(NR, 20, n)                                   # receivers 20..31: 680 kHz .. 999 kHz, the carrier at 820 kHz among them
    adc = synth.adc_stream(n * steps, 0x5EED0046)
    tb, ab, ob = tmp_path / "tables.bin", tmp_path / "adc.bin", tmp_path / "out.bin"
    bank = RxBank(NR, n)
    try:
        write_bank_tables(tb, mix, n, bank.fs)
        ab.write_bytes(adc.tobytes())
        r = subprocess.run([BANK_EXE, str(tb), str(ab), str(ob), str(steps)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        print(r.stdout.strip())
        enq_us = float(r.stdout.split("kg_rxbank_step ")[1].split(" us")[0])
        assert enq_us < 1000.0, r.stdout                          # (12 receivers on an idle GPU: ~100 us; the bound only catches a stall)
        blob = ob.read_bytes()
        bank.configure(mix)
        d_adc = bank.ctx.alloc(adc.nbytes)
        bank.ctx.upload(d_adc, adc)
        pos = 0
        info_dt = np.dtype([("step", "<u8"), ("nframes", "<i4"), ("nrec", "<i4"), ("nfir", "<i4"), ("fir_pos", "<i4"),
                            ("snd_seq", "<u4"), ("table_bytes", "<i4"), ("nmoves", "<i4"), ("pad", "<i4")])
        frame_dt = np.dtype([("rx", "<i4"), ("row", np.uint8, 1024), ("nb", "<i4"), ("pkt", np.uint8, 16 + 10 + 1024)])
        total_frames = total_blocks = 0
        for s in range(steps):
            info = bank.step(d_adc + 2 * n * s)
            bank.sync()
            got = np.frombuffer(blob, info_dt, 1, pos)[0]
            pos += info_dt.itemsize
            for k in ("step", "nframes", "nrec", "nfir", "fir_pos", "snd_seq", "table_bytes", "nmoves"):
                assert int(got[k]) == int(getattr(info, k)), (s, k, got[k], getattr(info, k))
            rx_of, _, nb = bank.frame_map()
            fr = np.frombuffer(blob, frame_dt, info.nframes, pos)
            pos += frame_dt.itemsize * info.nframes
            rows = bank.fetch("rows", range(info.nframes))
            pkts = bank.fetch("pkts", range(info.nframes))
            for f in range(info.nframes):
                assert int(fr[f]["rx"]) == int(rx_of[f]) and int(fr[f]["nb"]) == int(nb[f])
                assert np.array_equal(fr[f]["row"], rows[f]), (s, f)
                assert np.array_equal(fr[f]["pkt"][:nb[f]], pkts[f][:nb[f]]), (s, f)
            total_frames += info.nframes
            if info.nfir:
                s16 = bank.fetch("s16", range(NR))
                pay = bank.fetch("pay", range(NR))
                for k in range(NR):
                    a = np.frombuffer(blob, "<i2", info.nfir, pos); pos += 2 * info.nfir
                    b = np.frombuffer(blob, np.uint8, info.nfir // 2, pos); pos += info.nfir // 2
                    assert np.array_equal(a, s16[k, :info.nfir]) and np.array_equal(b, pay[k, :info.nfir // 2]), (s, k)
                    assert np.abs(a.astype(int)).max() > 0
                    total_blocks += info.nfir // 512
        assert pos == len(blob)
        n_ov = sum(1 for _, ov, _ in mix if ov)
        assert n_ov == 3 and total_frames == steps * NR - n_ov and total_blocks == 2 * NR
        bank.ctx.free(d_adc)
    finally:
        bank.close()


# ---- the audio chain from C++ ------------------------------------------------------------------------------------------
SND_EXE = os.path.join(ROOT, "examples", "sound_dropin")


def test_cpp_sound_caller_equals_the_python_mirror(gpu_ctx, tmp_path):
    """examples/sound_dropin.cpp -- snd_service()'s unpack, CFastFIR, the S-meter / AGC / detector / filter stage, the payload coders
    and the packet header through include/kiwigpu.h from plain C++, on 40 SPI buffers of 170 records for six connections (USB and AM
    with ADPCM, NBFM behind a squelch as raw little-endian audio, IQ in both byte orders, LSB with de-emphasis in network order):
    every packet it writes -- header, sequence number, S-meter field, flags, payload -- equals the Python mirror's, made with the
    same entry points.  (What those entry points compute is held to the reference elsewhere: tests/test_ref_pins_gpu.py.)"""
    import struct
    from flydog_sdr_gps_amd import FastFir, Post, deemp, post, snd, wire
    assert os.path.exists(SND_EXE), "examples/sound_dropin is not built (run __graft_entry__.build())"
    rate, nsamps, nbuf, cal = 12000.0, 170, 40, np.float32(-13)
    chans = [   # mode, agc (on, hang, thresh, manGain, slope, decay), de_emp, nfm, squelch, compression, little_endian, lo, hi
        (post.MODE_SSB, (1, 0, -100, 50, 6, 1000), 0, 0, 0, 1, 0, 300.0, 2700.0),
        (post.MODE_AM, (1, 0, -100, 50, 6, 1000), 1, 0, 0, 1, 0, -4900.0, 4900.0),
        (post.MODE_NBFM, (1, 0, -100, 50, 6, 1000), 0, 0, 80, 0, 1, -6000.0, 6000.0),
        (post.MODE_IQ, (1, 1, -90, 50, 3, 500), 0, 0, 0, 1, 0, -5000.0, 5000.0),
        (post.MODE_IQ, (0, 0, -100, 70, 6, 1000), 0, 0, 0, 1, 1, -5000.0, 5000.0),
        (post.MODE_SSB, (1, 0, -100, 50, 6, 1000), 2, 0, 0, 0, 0, -2700.0, -300.0),
    ]
    nch = len(chans)
    rng = np.random.default_rng(2024)
    t = np.arange(nsamps * nbuf)
    i24, q24 = np.zeros((t.size, nch), np.int64), np.zeros((t.size, nch), np.int64)
    for ch in range(nch):
        z = 2.0e5 * (1 + 0.5 * np.sin(2 * np.pi * t / (40.0 + ch))) * np.exp(2j * np.pi * (0.05 + 0.013 * ch) * t) + rng.normal(0, 2000, t.size) \
            + 1j * rng.normal(0, 2000, t.size)
        i24[:, ch], q24[:, ch] = np.rint(z.real).astype(np.int64), np.rint(z.imag).astype(np.int64)
    raw = snd.pack_rx_iq(i24, q24)
    cfg = struct.pack("<iiiff", nch, nsamps, nbuf, rate, snd.RESCALE)
    for mode, agc, de, nfm, sqv, comp, le, lo, hi in chans:
        taps = np.zeros(79, np.float32)
        if de:
            taps[:] = deemp.table(bool(nfm), True)[de - 1]
        cfg += struct.pack("<12iff", mode, *agc, de, nfm, sqv, comp, le, lo, hi) + taps.tobytes()
    cb, rb, ob = tmp_path / "cfg.bin", tmp_path / "raw.bin", tmp_path / "out.bin"
    cb.write_bytes(cfg); rb.write_bytes(raw.tobytes())
    r = subprocess.run([SND_EXE, str(cb), str(rb), str(ob)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    data, got, pos = ob.read_bytes(), [], 0
    while pos < len(data):
        ch, nb = struct.unpack_from("<ii", data, pos)
        got.append((ch, np.frombuffer(data, np.uint8, nb, pos + 8)))
        pos += 8 + nb

    # the same calls from Python
    F, P, A = FastFir(gpu_ctx, nchan=nch, max_in=nsamps), Post(gpu_ctx, nchan=nch), wire.Adpcm(gpu_ctx, nchan=nch)
    try:
        for ch, (mode, agc, de, nfm, sqv, comp, le, lo, hi) in enumerate(chans):
            lo, hi = max(lo, -(rate / 2 - 1)), min(hi, rate / 2 - 1)            # rx_sound_cmd.cpp:248-250
            assert F.setup(ch, lo, hi, 0.0, rate)
            P.set_smeter(ch, rate); P.set_agc(ch, *agc, rate); P.set_am_passband(ch, lo, hi, rate)
            P.squelch_setup(ch, rate); P.squelch_set(ch, sqv, 0)
            P.set_de_emp(ch, de, nfm, snd_rate_12k=True, frate=rate)
            P.set_mode(ch, mode); P.reset(ch)
        want, pend, seq = [], [np.zeros(0, np.uint8) for _ in chans], [0] * nch
        for b in range(nbuf):
            x = snd.unpack(gpu_ctx, raw[b * nsamps * nch * 6:(b + 1) * nsamps * nch * 6], nsamps, nch)
            for ch, (mode, agc, de, nfm, sqv, comp, le, lo, hi) in enumerate(chans):
                y = F.process(ch, x[ch])
                if y.size == 0:
                    continue
                assert y.size == 512
                s16, _, agc_out = P.process([ch], y[None, :])
                if mode == post.MODE_IQ:
                    pay = wire.snd_iq_payload(gpu_ctx, agc_out[0][None, :], le).reshape(-1)
                elif comp:
                    pay = A.encode([ch], s16[0][None, :]).reshape(-1)
                else:
                    pay = wire.snd_payload(gpu_ctx, s16[0][None, :], le).reshape(-1)
                pend[ch] = np.concatenate([pend[ch], pay])
                if pend[ch].size < 1024:
                    continue
                avg, _ = P.smeter([ch])
                _, sq, _ = P.squelch_state([ch])
                iq = mode == post.MODE_IQ
                flags = (0x08 if iq else 0) | (0x10 if comp and not iq else 0) | (0x40 if sq[0] else 0) | (0x80 if le else 0)
                seq[ch] += 1
                want.append((ch, np.concatenate([wire.snd_header(gpu_ctx, flags, seq[ch], float(np.float32(avg[0]) + cal)), pend[ch]])))
                pend[ch] = np.zeros(0, np.uint8)
    finally:
        A.close(); P.close(); F.close()
    assert len(got) == len(want) >= 40, (len(got), len(want))
    sizes = {ch: set() for ch in range(nch)}
    for k, ((gc, gp), (wc, wp)) in enumerate(zip(got, want)):
        assert gc == wc and gp.size == wp.size and np.array_equal(gp, wp), (k, gc, wc, gp.size, wp.size, int(np.argmax(gp[:min(gp.size, wp.size)] != wp[:min(gp.size, wp.size)])))
        assert bytes(gp[:3]) == b"SND"
        sizes[gc].add(gp.size)
    assert sizes == {0: {1034}, 1: {1034}, 2: {1034}, 3: {2058}, 4: {2058}, 5: {1034}}, sizes
