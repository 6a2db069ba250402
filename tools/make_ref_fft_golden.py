"""Generates tests/golden/acq_fftref.npz, fastfir_fftref.npz, wf_fftref.npz and aper_fftref.npz from the REFERENCE ITSELF -- its gps/search.cpp (SearchInit,
Sample, Correlate, the decimators) and rx/CuteSDR/fastfir.cpp (CFastFIR), compiled in place by oracle/build_ref.sh against the
FFTW3 API the image ships (hipFFTW) -- RUN ON THE GPU BOX, where hipFFTW's transforms execute:

    gpurun -- 'python tools/make_ref_fft_golden.py gpurun_out/fftref'      (then: cp gpurun_out/fftref/*.npz tests/golden/)

Needs oracle/_ref/search_ref and fastfir_ref (built in the build container, they travel with the snapshot) and nothing of
/root/reference.  What is committed is data: inputs made here from fixed seeds, the outputs the reference's code produced.
Spectra are kept as every 4th bin plus the full array's L2 norm and largest magnitude (a code table is 128 KiB).
The script also prints how far the oracle's restatement is from each vector (the tests assert it).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fftref")
os.makedirs(out_dir, exist_ok=True)

from flydog_sdr_gps_amd import prn, sats, synth            # noqa: E402  (pure numpy input generators)
from tests.fixtures import e1b_chips                         # noqa: E402

FFT_LEN, KEEP = 16384, 4


def run(binary, script, data):
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        np.asarray(data).tofile(os.path.join(tmp, "in.bin"))
        p = subprocess.run([os.path.join(REF, binary), os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p.returncode != 0:
            raise SystemExit("%s failed (%d): %s" % (binary, p.returncode, p.stderr.decode()[-2000:]))
        return np.fromfile(os.path.join(tmp, "out.bin"), np.float32)


def keep(spec):
    spec = np.asarray(spec, np.complex64)
    return {"bins": spec[::KEEP].copy(), "l2": np.float64(np.sqrt(np.sum(np.abs(spec.astype(np.complex128)) ** 2))),
            "max": np.float32(np.abs(spec).max())}


# ---- acquisition ------------------------------------------------------------------------------------------------------------
e1b = e1b_chips()
codes = synth.all_sv_codes(e1b)
E02, E05, E36 = 36, 39, 58                                   # rows of sats.SATS (E1B PRNs 2, 5, 36)
assert sats.SATS[E02][0] == 2 and sats.SATS[E05][0] == 5 and sats.SATS[E36][0] == 36 and sats.SATS[32][0] == 194
scenes = [
    # name, the SVs in the 1-bit IF block [(sat, tau chips, Doppler Hz, theta, C/N0)], seed, the SVs Correlate() is asked for
    ("config0_prn1", [(0, 300.5, 1500.0, 0.7, 45.0)], 0x5EED0001, [0, 5]),            # BASELINE configs[0]; PRN 6 is absent
    ("noise_only", [], 0x5EED00B0, [0, E02]),
    ("prn20_negative_doppler", [(19, 1000.25, -3200.0, 2.1, 47.0), (7, 12.0, 4900.0, 0.3, 44.0)], 0x5EED00B1, [19, 7, 32]),
    ("e1b_e02", [(E02, 1500.25, 3900.0, 2.2, 46.0)], 0x5EED00B2, [E02, E05]),
    ("e1b_e36_late_code_phase", [(E36, 4000.5, -1100.0, 0.9, 47.0), (2, 700.0, 250.0, 1.0, 45.0)], 0x5EED00B3, [E36, 2]),
]
acq = {"scene_names": np.array([s[0] for s in scenes]), "keep_every": np.int32(KEEP)}
# code tables (row G3): C/A PRN1, QZSS 194, Galileo E02 and E36
code_sats = [0, 32, E02, E36]
y = run("search_ref", ["T %d" % s for s in code_sats], np.zeros(1, np.uint8)).view(np.complex64).reshape(len(code_sats), FFT_LEN)
acq["code_sats"] = np.array(code_sats, np.int32)
for k, s in enumerate(code_sats):
    for key, v in keep(y[k]).items():
        acq["code_%d_%s" % (s, key)] = v
# decimators (rows G5, G6): 256-sample vectors, tail behaviour included (the filter reads 30 samples past the end: zero-filled)
rng = np.random.Generator(np.random.PCG64(0x5EED00B4))
dec_in = (rng.standard_normal(256) + 1j * rng.standard_normal(256)).astype(np.complex64)
bits_in = rng.integers(0, 2, (256, 2)).astype(np.int8)
y = run("search_ref", ["D 256", "B 256"], np.concatenate([dec_in.view(np.uint8), bits_in.view(np.uint8).ravel()]))
acq["dec_float_in"], acq["dec_float_out"] = dec_in, y[:256].view(np.complex64).copy()
acq["dec_binary_in"], acq["dec_binary_out"] = bits_in, y[256:512].view(np.complex64).copy()
for name, present, seed, asked in scenes:
    svs = [(codes[sat][0], tau, fd, th, cn0, codes[sat][1]) for sat, tau, fd, th, cn0 in present]
    packed = synth.gps_scene_bits(svs, seed)
    assert packed.size == 8192
    y = run("search_ref", ["S"] + ["C %d" % s for s in asked], packed)
    spec = y[:2 * FFT_LEN].view(np.complex64)
    res = y[2 * FFT_LEN:].reshape(len(asked), 3)
    acq[name + "_bits"] = packed
    for key, v in keep(spec).items():
        acq[name + "_spec_" + key] = v
    acq[name + "_sats"] = np.array(asked, np.int32)
    acq[name + "_snr"] = res[:, 0].copy()
    acq[name + "_dop"] = res[:, 1].astype(np.int32)
    acq[name + "_idx"] = res[:, 2].astype(np.int32)
    print("acq %-26s" % name, " ".join("sat %d: snr %.2f dop %d idx %d;" % (s, r[0], r[1], r[2]) for s, r in zip(asked, res)))
np.savez_compressed(os.path.join(out_dir, "acq_fftref.npz"), **acq)

# ---- CFastFIR ---------------------------------------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED00B5))


def audio_iq(n, amp=3000.0):
    t = np.arange(n)
    x = amp * (np.exp(2j * np.pi * 0.04 * t) + 0.3 * np.exp(-2j * np.pi * 0.11 * t)) + 40.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


ff_scen = [
    # USB 300-2700 Hz at 12 kHz in the data pump's 170-sample interrupts (4 channels, config.h:40): FirPos 0, 170, 340, 510 -> 168, ...
    ("usb_12k_170", ["P 0 300 2700 0 12000"] + ["D 170"] * 12),
    ("lsb_12k_ragged", ["P 0 -2700 -300 0 12000", "D 85", "D 226", "D 48", "D 512", "D 1", "D 600", "D 170", "D 170", "D 170"]),
    ("am_12k", ["P 0 -4900 4900 0 12000"] + ["D 170"] * 7),
    ("cw_narrow_offset", ["P 0 -250 250 500 12000"] + ["D 170"] * 7),
    ("wide_20k", ["P 0 -6000 6000 0 20250"] + ["D 226"] * 6),
    # a retune in mid-stream (the buffer keeps its samples, :171-232), a rejected setting (lo >= hi: the old filter stays, :193-200),
    # the same setting again (returns at once, :180-184)
    ("retune_and_rejected", ["P 0 300 2700 0 12000", "D 400", "P 0 -2700 -300 0 12000", "D 400", "P 0 500 400 0 12000", "D 400",
                             "P 0 500 400 0 12000", "D 400", "P 0 300 2700 0 12000", "D 448"]),
    ("cic_comp_on", ["C 1", "P 0 300 2700 0 12000"] + ["D 170"] * 7),
]
ff = {"names": np.array([s[0] for s in ff_scen])}
for name, script in ff_scen:
    n = sum(int(l.split()[1]) for l in script if l[0] == "D")
    x = audio_iq(n)
    y = run("fastfir_ref", script, x)
    ff[name + "_script"], ff[name + "_in"], ff[name + "_out"] = np.array(script), x, y
    print("fastfir %-22s %5d samples in, %5d floats out" % (name, n, y.size))
np.savez_compressed(os.path.join(out_dir, "fastfir_fftref.npz"), **ff)

# ---- CFastFIR's extension taps (fastfir.cpp:278-302) ------------------------------------------------------------
# the driver's H line registers an extension hook through ext_users[]: PRE_FILTERED gets the forward spectrum times m_CIC, POST_FILTERED the
# filtered spectrum; an editing PRE hook (zeroes bins 256 .. 767, answers true) makes ProcessData filter the edited buffer with m_pFilterCoef
tp_scen = [
    ("pre_post_cic_on", ["C 1", "P 0 300 2700 0 12000", "H 3 0"] + ["D 512"] * 4),
    ("post_only_cic_off_ragged", ["P 0 -2700 -300 0 12000", "H 2 0", "D 300", "D 724", "D 1024", "D 170", "D 342"]),
    ("pre_edit_cic_on", ["C 1", "P 0 -4900 4900 0 12000", "H 1 1"] + ["D 512"] * 4),
    ("pre_edit_cic_off", ["P 0 -4900 4900 0 12000", "H 1 1"] + ["D 512"] * 3 + ["H 0 0", "D 512"]),
]
tp = {"names": np.array([s_[0] for s_ in tp_scen])}
for name, script in tp_scen:
    n = sum(int(l.split()[1]) for l in script if l[0] == "D")
    x = audio_iq(n)
    y = run("fastfir_ref", script, x)
    tp[name + "_script"], tp[name + "_in"], tp[name + "_out"] = np.array(script), x, y
    print("fastfir taps %-26s %5d samples in, %6d floats out" % (name, n, y.size))
np.savez_compressed(os.path.join(out_dir, "fastfir_taps_fftref.npz"), **tp)

# ---- waterfall frames -------------------------------------------------------------------------------------------------------
from flydog_sdr_gps_amd import wf as wfm                    # noqa: E402  (host mirror: WfParams, maps -- the per-frame INPUTS)

WF_CASES = [
    # zoom, start, interp, window, cic_comp, overlapped, inversion, compression   (tests/test_wf_gpu.py's cases + compression)
    (0, 0.0, wfm.WF_CMA, wfm.WINF_HANNING, True, False, False, True),            # zoom 0: never compressed (:1285)
    (3, 2.0e6, wfm.WF_MAX, wfm.WINF_BLACKMAN_HARRIS, True, False, False, True),
    (10, 9.0e6, wfm.WF_DROP, wfm.WINF_HANNING, True, False, True, False),
    (0, 0.0, wfm.WF_MAX, wfm.WINF_BLACKMAN_HARRIS, False, False, False, False),  # dc = 4 bins
    (1, 1.0e6, wfm.WF_MIN, wfm.WINF_HAMMING, True, False, False, True),          # zoom 1: never compensated
    (5, 5.0e6, wfm.WF_LAST, wfm.WINF_NONE, True, True, False, True),             # overlapped: no compensation
    (7, 3.0e6, wfm.WF_CMA, wfm.WINF_HANNING, True, False, True, False),
    (14, 1.6e7, wfm.WF_MIN, wfm.WINF_HANNING, False, False, False, True),
]
script, blobs, meta = ["T"], [], []
for k, (zoom, start, interp, winf, cic, ovl, inv, comp) in enumerate(WF_CASES):
    prm = wfm.WfParams.for_zoom(zoom, start, spectral_inversion=inv)
    m, d = wfm.build_maps(prm.fft_used, prm.plot_width, prm.plot_width_clamped, inv)
    sc = np.full(1024, prm.fft_scale, np.float32)
    iq = synth.wf_iq_frame(seed=1000 + zoom + 100 * k)
    # fft2wf_map entries of -1 (inverted maps: "not plotted") are stored in the reference's u2_t table as 65535
    m16 = np.asarray(m, np.int64).astype(np.uint16)
    d16 = np.zeros(1024, np.uint16)
    d16[:np.asarray(d).size] = np.asarray(d, np.int64).astype(np.uint16)
    # every case twice: the uncompressed row (WF_COMP_OFF), then the packet payload as the case's compression setting makes it
    for c in (0, int(comp)):
        blobs += [m16.tobytes(), d16.tobytes(), sc.tobytes(), (sc / np.float32(2)).astype(np.float32).tobytes(), np.ascontiguousarray(iq, np.int16).tobytes()]
        script.append("F %d %d %d %d %d %d %d %d %r %d %d %d" % (zoom, winf, interp, int(cic), int(ovl), prm.fft_used, prm.plot_width,
                                                                 prm.plot_width_clamped, float(prm.fft_offset), c, int(prm.start), 4242 + k))
    meta.append((prm, iq))
y = run("wf_ref", script, np.frombuffer(b"".join(blobs), np.uint8))
wfg = {"ncases": np.int32(len(WF_CASES)), "cases": np.array([[c[0], c[1], c[2], c[3], int(c[4]), int(c[5]), int(c[6]), int(c[7])] for c in WF_CASES], np.float64)}
pos = 0
wfg["window_function"] = y[pos:pos + 4 * 8192].reshape(4, 8192).copy(); pos += 4 * 8192
wfg["cic_comp"] = y[pos:pos + 8192].copy(); pos += 8192
wfg["n_chunks"] = np.int32(y[pos]); pos += 1
for k, (prm, iq) in enumerate(meta):
    nbytes, limit, xbin, flags, seq = (int(v) for v in y[pos:pos + 5]); pos += 5
    assert nbytes == 1024
    pos += 2 * prm.fft_used
    wfg["case%d_row" % k] = y[pos:pos + nbytes].astype(np.uint8); pos += nbytes
    nbytes, limit, xbin, flags, seq = (int(v) for v in y[pos:pos + 5]); pos += 5
    spec = y[pos:pos + 2 * prm.fft_used].view(np.complex64); pos += 2 * prm.fft_used
    payload = y[pos:pos + nbytes].astype(np.uint8); pos += nbytes
    wfg["case%d_iq" % k] = np.ascontiguousarray(iq, np.int16)
    wfg["case%d_hdr" % k] = np.array([nbytes, limit, xbin, flags, seq], np.int64)
    wfg["case%d_spec" % k] = spec[::2].copy()                       # every 2nd of the fft_used bins
    wfg["case%d_spec_max" % k] = np.float32(np.abs(spec).max())
    wfg["case%d_payload" % k] = payload
    print("wf case %d (z%d): %d payload bytes, fft_used_limit %d, flags 0x%x, row min/max %d/%d" % (
        k, WF_CASES[k][0], nbytes, limit, flags, payload.min(), payload.max()))
assert pos == y.size
np.savez_compressed(os.path.join(out_dir, "wf_fftref.npz"), **wfg)

# ---- aperture_auto(): the static function compute_frame() calls when wf->aper == AUTO (rx_waterfall.cpp:1173-1272, :1619) --------
# One connection's life per run: the averages are loaded from the first row (avg_clear), every later row is averaged in with the
# run's algorithm, and the 5 dB histogram is reported when the script's clock has advanced far enough (3 s; 1 s single-shot).
# Every frame is an uncompressed row of case 1's setting (zoom 3) over its own seeded samples; a run of all-zero samples gives
# the masked-everywhere report.  Per frame the golden keeps the row the reference produced (aperture_auto's input) and the
# state it left: signal, noise, done_autoscale, report_sec, avg_clear, avg_pwr[1024].
IIR, MMA, EMA, OFF = 0, 1, 2, 3
APER_RUNS = [
    # algo, param, waterfall_cal, wf_chans, [clock of each frame], [need_autoscale of each frame], zero input?
    (IIR, 0.35, -13, 4, [100, 101, 102, 103, 104, 105, 107, 108], [1, 1, 1, 1, 2, 2, 2, 2], False),
    (MMA, 4.0, -13, 4, [200, 201, 203, 204, 206, 209], [3, 3, 3, 4, 4, 4], False),
    (EMA, 6.0, -20, 4, [300, 302, 303, 306, 307], [5, 5, 5, 6, 6], False),
    (OFF, 0.0, -13, 4, [400, 400, 401, 401, 402], [7, 7, 7, 8, 8], False),         # single shot: MMA 8, one second
    (MMA, 2.0, 0, 0, [500, 501, 503, 504], [9, 9, 9, 9], False),                    # rx14.wf0: the audio FFT's pixels 256 .. 767
    (EMA, 3.0, -13, 4, [600, 601, 603], [10, 10, 10], True),                         # nothing but masked bands: -110 / -120
    (IIR, 4.0, -13, 4, [700, 703, 704, 707, 710], [11, 11, 12, 12, 13], False),     # a fast IIR, reports every 3 s
]
zoom, start_hz, interp, winf, cic, ovl, inv, _ = WF_CASES[1]
prm = wfm.WfParams.for_zoom(zoom, start_hz, spectral_inversion=inv)
m, d = wfm.build_maps(prm.fft_used, prm.plot_width, prm.plot_width_clamped, inv)
sc = np.full(1024, prm.fft_scale, np.float32)
m16 = np.asarray(m, np.int64).astype(np.uint16)
d16 = np.zeros(1024, np.uint16)
d16[:np.asarray(d).size] = np.asarray(d, np.int64).astype(np.uint16)
script, blobs, frames = [], [], []
for r, (algo, param, cal, nwf, clocks, needs, zero) in enumerate(APER_RUNS):
    for f, (now, need) in enumerate(zip(clocks, needs)):
        script.append("P 1 %d %r %d %d %d %d %d" % (algo, float(param), 1 if f == 0 else -1, need, now, cal, nwf))
        iq = np.zeros((8192, 2), np.int16) if zero else synth.wf_iq_frame(seed=7000 + 37 * r + f)
        if not zero and f % 3 == 2:
            iq = (np.asarray(iq, np.int32) // 8).astype(np.int16)       # a quieter frame: the averages move
        blobs += [m16.tobytes(), d16.tobytes(), sc.tobytes(), (sc / np.float32(2)).astype(np.float32).tobytes(), np.ascontiguousarray(iq, np.int16).tobytes()]
        script.append("F %d %d %d %d %d %d %d %d %r %d %d %d" % (zoom, winf, interp, int(cic), int(ovl), prm.fft_used, prm.plot_width,
                                                                 prm.plot_width_clamped, float(prm.fft_offset), 0, int(prm.start), 9000 + f))
        frames.append((r, f, algo, param, cal, nwf, now, need))
y = run("wf_ref", script, np.frombuffer(b"".join(blobs), np.uint8))
pos, rows, states, avgs = 0, [], [], []
for _ in frames:
    nbytes = int(y[pos]); pos += 5
    assert nbytes == 1024
    pos += 2 * prm.fft_used
    rows.append(y[pos:pos + 1024].astype(np.uint8)); pos += 1024
    states.append(y[pos:pos + 5].astype(np.int64)); pos += 5
    avgs.append(y[pos:pos + 1024].copy()); pos += 1024
assert pos == y.size
aper = {"frames": np.array(frames, np.float64), "rows": np.array(rows), "states": np.array(states), "avg_pwr": np.array(avgs)}
np.savez_compressed(os.path.join(out_dir, "aper_fftref.npz"), **aper)
for (r, f, algo, param, cal, nwf, now, need), st in zip(frames, states):
    print("aper run %d frame %d (algo %d, t = %d, need %d): signal %d noise %d done %d report_sec %d" % (r, f, algo, now, need, *st[:4]))

# ---- how far is the oracle's restatement? (the tests assert these; here for the log) -----------------------------------------
try:
    from oracle import kiwi_oracle as ko
    ko.lib()
    g = np.load(os.path.join(out_dir, "acq_fftref.npz"))
    for s in code_sats:
        boc = codes[s][1]
        mine = ko.code_fft(codes[s][0], boc=boc)
        print("code table sat %2d: oracle vs reference %.2e of max; l2 %.3e vs %.3e" % (
            s, np.abs(mine[::KEEP] - g["code_%d_bins" % s]).max() / g["code_%d_max" % s], np.sqrt(np.sum(np.abs(mine.astype(np.complex128)) ** 2)), g["code_%d_l2" % s]))
    print("DecimateBy2float bit-exact:", np.array_equal(ko.decimate_by2(g["dec_float_in"]).view(np.uint32), g["dec_float_out"].view(np.uint32)))
    for name, present, seed, asked in scenes:
        data = ko.sample_bits(g[name + "_bits"])
        d = np.abs(data[::KEEP] - g[name + "_spec_bins"]).max() / g[name + "_spec_max"]
        line = "scene %-26s spectrum %.2e of max;" % (name, d)
        for k, s in enumerate(asked):
            lim = ko.E1B_LIMIT if codes[s][1] else ko.L1_LIMIT
            nxt = ko.code_fft(codes[s + 1][0], boc=codes[s + 1][1]) if s + 1 < len(codes) else None
            r, _ = ko.correlate(ko.code_fft(codes[s][0], boc=codes[s][1]), data, limit=lim, code_next=nxt)
            line += " sat %d: (%d, %d) vs (%d, %d) snr %.4f vs %.4f;" % (s, r["dop"], r["idx"], g[name + "_dop"][k], g[name + "_idx"][k], r["snr"], g[name + "_snr"][k])
        print(line)
    g = np.load(os.path.join(out_dir, "wf_fftref.npz"))
    tables = (g["window_function"], g["cic_comp"])
    for w in range(4):
        print("window %d: oracle bit-exact %s" % (w, np.array_equal(ko.wf_window(w).view(np.uint32), g["window_function"][w].view(np.uint32))))
    print("CIC_comp: oracle max rel diff %.2e" % (np.abs(ko.wf_cic_comp() - g["cic_comp"]).max() / np.abs(g["cic_comp"]).max()))
    for k, (zoom, start, interp, winf, cic, ovl, inv, comp) in enumerate(WF_CASES):
        prm = meta[k][0]
        m, d = wfm.build_maps(prm.fft_used, prm.plot_width, prm.plot_width_clamped, inv)
        sc = np.full(1024, prm.fft_scale, np.float32)
        samps = ko.wf_window_iq(g["case%d_iq" % k], tables[0][winf])
        row, pwr, pwr_out, dB = ko.wf_compute_frame(samps, prm.zoom, winf, interp, cic, ovl, prm.fft_used, prm.plot_width, prm.plot_width_clamped,
                                                    m, d, sc, (sc / np.float32(2)).astype(np.float32), prm.fft_offset, tables[1])
        use_comp = comp and zoom != 0
        want_pkt = ko.wf_packet(row, int(prm.start), prm.zoom, 4242 + k, use_comp)[16:]
        pay = g["case%d_payload" % k]
        dd = np.abs(row.astype(int) - g["case%d_row" % k].astype(int))
        print("wf case %d: row differs in %d of 1024 pixels (max %d); payload (%d bytes) equal: %s" % (
            k, np.count_nonzero(dd), dd.max(), pay.size, np.array_equal(want_pkt, pay)))
except Exception as e:                                                          # the vectors are written either way
    print("oracle comparison skipped:", repr(e))
