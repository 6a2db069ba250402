"""Generates the committed fixtures under tests/golden/ (run in the build container).

  acq_golden.npz  acquisition cases: packed 1-bit IF input + the CPU oracle's
                  (snr, dop, idx, valid) and per-Doppler peak indices.

The expected values here come from the oracle (oracle/kiwi_oracle.c, prec=1): they pin the oracle against regressions
and travel to the GPU box; they are not reference outputs.  (What the reference itself computed -- its gps/search.cpp,
rx/rx_waterfall.cpp, rx/CuteSDR/fastfir.cpp built in place against hipFFTW -- is tests/golden/*_fftref.npz,
tools/make_ref_fft_golden.py; the reference holds no golden vectors of its own for this path.)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flydog_sdr_gps_amd import prn, sats, synth      # noqa: E402
from oracle import kiwi_oracle as ko                  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

from tests.fixtures import e1b_chips                   # noqa: E402  (the reference's own E1BCODE outputs, e1b_ref.npz)

cases = []


def add(sat, bits, chips=None):
    kind = sats.SATS[sat][3]
    if kind == sats.E1B:
        code, limit = ko.code_fft(chips, boc=True), sats.E1B_LIMIT
    else:
        code, limit = ko.code_fft(prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2])), sats.L1_LIMIT
    r, cells = ko.correlate(code, ko.sample_bits(bits), limit=limit)
    cases.append((sat, bits, chips, r, cells))
    print("sat %2d (%s %d): snr %.2f dop %d idx %d valid %d" %
          (sat, kind, sats.SATS[sat][0], r["snr"], r["dop"], r["idx"], r["valid"]))


def ca(sat):
    return prn.cacode(sats.SATS[sat][1], sats.SATS[sat][2])


add(0, synth.config0_bits())                                                   # BASELINE configs[0]
add(16, synth.gps_scene_bits([(ca(16), 17.75, -4100.0, 2.0)], seed=11, cn0_dbhz=44.0))
add(32, synth.gps_scene_bits([(ca(32), 800.5, 2600.0, 1.0)], seed=12, cn0_dbhz=47.0))   # QZSS 194
add(4, synth.gps_scene_bits([(ca(0), 300.5, 1500.0, 0.7)], seed=13))           # PRN5 absent
add(8, synth.gps_scene_bits([(ca(8), 1022.9, 5000.0, 0.0)], seed=14, cn0_dbhz=41.0))    # edge Doppler, weak
e11 = e1b_chips()[11]
sat_e11 = [i for i, s in enumerate(sats.SATS) if s[3] == sats.E1B and s[0] == 11][0]
add(sat_e11, synth.gps_scene_bits([(e11, 2000.25, -1250.0, 0.5, 48.0, True)], seed=15), e11)

out = {"ncases": len(cases)}
for k, (sat, bits, chips, r, cells) in enumerate(cases):
    out["case%d_sat" % k] = sat
    out["case%d_bits" % k] = bits
    if chips is not None:
        out["case%d_chips" % k] = chips
    out["case%d_result" % k] = np.array([r["snr"], r["dop"], r["idx"], r["valid"]], np.float64)
    out["case%d_cell_idx" % k] = cells["idx"]
    out["case%d_cell_snr" % k] = cells["snr"]
np.savez_compressed(os.path.join(GOLD, "acq_golden.npz"), **out)
print("wrote", GOLD)

# ---------------------------------------------------------------------------------
# waterfall / audio front / DDC fixtures (same status: oracle outputs, regression pins)
# ---------------------------------------------------------------------------------
from flydog_sdr_gps_amd import wf, snd          # noqa: E402

tables = (wf.window_functions(), wf.cic_comp_table())
out = {"cic_comp": tables[1]}
cases_wf = [(0, 0.0, wf.WF_CMA, wf.WINF_HANNING, True, False),
            (3, 2.0e6, wf.WF_MAX, wf.WINF_BLACKMAN_HARRIS, True, False),
            (10, 9.0e6, wf.WF_DROP, wf.WINF_HANNING, True, True)]
out["ncases"] = len(cases_wf)
for k, (zoom, start, interp, winf, cic, inv) in enumerate(cases_wf):
    p = wf.WfParams.for_zoom(zoom, start, spectral_inversion=inv)
    m, d = wf.build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, inv)
    iq = synth.wf_iq_frame(seed=4000 + k, tones=((0.05, -50.0), (0.21, -60.0)), noise_dbfs=-45.0)
    sc = np.full(1024, p.fft_scale, np.float32)
    row, pwr, pwr_out, dB = ko.wf_compute_frame(ko.wf_window_iq(iq, tables[0][winf]), p.zoom, winf, interp, cic,
                                                False, p.fft_used, p.plot_width, p.plot_width_clamped, m, d, sc,
                                                (sc / np.float32(2)).astype(np.float32), p.fft_offset, tables[1])
    out["case%d_cfg" % k] = np.array([zoom, start, interp, winf, int(cic), int(inv)], np.float64)
    out["case%d_iq" % k] = iq
    out["case%d_row" % k] = row
    out["case%d_dB" % k] = dB
    out["case%d_pwr_out" % k] = pwr_out
    print("wf case %d: z%d row min/max %d/%d" % (k, zoom, row.min(), row.max()))
np.savez_compressed(os.path.join(GOLD, "wf_golden.npz"), **out)

rng = np.random.default_rng(0x5EED0A)
i24 = rng.integers(-2 ** 23, 2 ** 23, (8, 4)); q24 = rng.integers(-2 ** 23, 2 ** 23, (8, 4))
i24[0, 0], q24[0, 0], i24[1, 1], q24[1, 1] = -2 ** 23, 2 ** 23 - 1, -1, -2
raw = snd.pack_rx_iq(i24, q24)
coef, coef_cic, tcoef = ko.fir_design(300.0, 2700.0, 0.0, 12000.0)
x = ((rng.standard_normal(170 * 7) + 1j * rng.standard_normal(170 * 7)) * 2000).astype(np.complex64)
st, outs, pos = ko.fir_new_state(), [], []
for k in range(7):
    o, pp = ko.fir_process(st, coef_cic, x[170 * k:170 * (k + 1)])
    outs.append(o); pos.append(pp)
np.savez_compressed(os.path.join(GOLD, "snd_golden.npz"), raw=raw,
                    unpack_normal=ko.dpump_unpack(raw, 8, 4, dc_i=0.5, dc_q=-0.25),
                    unpack_inverted=ko.dpump_unpack(raw, 8, 4, dc_i=0.5, dc_q=-0.25, spectral_inversion=True),
                    fir_coef=coef_cic, fir_in=x, fir_out=np.concatenate(outs), fir_pos=np.array(pos))
print("snd: fir outputs", sum(o.size for o in outs), "pos", pos)

t = np.arange(1 << 15)
adc = np.clip(np.rint(9000 * np.cos(2 * np.pi * 0.0123 * t) + 700 * np.cos(2 * np.pi * 0.201 * t + 1)
                      + rng.normal(0, 40, t.size)), -32768, 32767).astype(np.int16)
inc = (-int(round((0.0123 + 2.0 ** -22) * 2 ** 48))) & ((1 << 48) - 1)
dd = {"adc": adc, "inc": np.uint64(inc)}
for l2 in (0, 4, 11):
    dd["wf_r%d" % l2] = ko.ddc_wf(adc, inc, l2)[0]
n_rx = 10416 * 5
t = np.arange(n_rx)
adc_rx = np.clip(np.rint(15000 * np.cos(2 * np.pi * (0.0371 + 900 / 125e6) * t) + rng.normal(0, 30, n_rx)), -32768, 32767).astype(np.int16)
inc_rx = int(round(0.0371 * 2 ** 48)) & ((1 << 48) - 1)
dd["adc_rx"] = adc_rx
dd["inc_rx"] = np.uint64(inc_rx)
dd["rx_records"] = ko.ddc_rx(adc_rx, inc_rx)[0]
np.savez_compressed(os.path.join(GOLD, "ddc_golden.npz"), **dd)
print("ddc: wf outputs", {k: v.shape for k, v in dd.items() if k.startswith("wf_")}, "rx records", dd["rx_records"].size // 6)

# ---- S-meter / CAgc / detectors (kiwi_oracle_post.c) ----------------------------------
rng = np.random.default_rng(0x5EED0B)
n = 4 * 512
t = np.arange(n)
env = np.where((t // 600) % 2 == 0, 400.0, 7000.0) * np.exp(-(t % 600) / 500.0)
xp = (env * np.exp(2j * np.pi * 0.043 * t) + rng.normal(0, 8, n) + 1j * rng.normal(0, 8, n)).astype(np.complex64)
pg = {"x": xp, "agc_args": np.array([[1, 0, -100, 50, 6, 1000], [1, 1, -90, 50, 3, 500], [0, 0, -100, 60, 6, 1000]]),
      "rate": np.float32(12000.0)}
for k, args in enumerate(pg["agc_args"]):
    a = ko.Agc(); a.set_parameters(*[int(v) for v in args], 12000.0)
    b = ko.Agc(); b.set_parameters(*[int(v) for v in args], 12000.0)
    cp = np.concatenate([a.process_cpx(xp[i:i + 512]) for i in range(0, n, 512)])
    pg["agc_cpx_%d" % k] = cp
    pg["agc_s16_%d" % k] = np.concatenate([b.process_s16(xp[i:i + 512]) for i in range(0, n, 512)])
    pg["am_%d" % k] = ko.am_detect(0.0, cp)[0]
    pg["nbfm_%d" % k] = ko.nbfm_detect((0.0, 0.0), cp)[0]
al = ko.smeter_alpha(12000.0)
avg, taps = 0.0, None
for i in range(0, n, 512):
    avg, taps = ko.smeter_process(avg, al, xp[i:i + 512])
pg["smeter"] = np.array([al, avg, taps[0], taps[1]], np.float32)
np.savez_compressed(os.path.join(GOLD, "post_golden.npz"), **pg)
print("post: smeter", pg["smeter"], "agc |out| max", [float(np.abs(pg["agc_cpx_%d" % k]).max()) for k in range(3)])

# ---- wire formats (kiwi_oracle_wire.c) ------------------------------------------------
rng = np.random.default_rng(0x5EED0C)
t = np.arange(2048)
aud = np.clip(np.rint(7000 * np.sin(2 * np.pi * 0.011 * t) * (1 + 0.8 * np.sin(2 * np.pi * t / 400.0))
                      + rng.normal(0, 500, t.size)), -32768, 32767).astype(np.int16)
st, enc = None, []
for i in range(0, aud.size, 512):
    e, st = ko.adpcm_encode_i16(aud[i:i + 512], st)
    enc.append(e)
row = np.clip(110 + 70 * np.sin(np.arange(1024) / 11.0) + rng.normal(0, 9, 1024), 0, 255).astype(np.uint8)
np.savez_compressed(os.path.join(GOLD, "wire_golden.npz"), audio=aud, adpcm=np.concatenate(enc),
                    adpcm_state=np.array([st.index, st.previous]), row=row,
                    pkt_compressed=ko.wf_packet(row, 123456, 7, 4242, True),
                    pkt_raw=ko.wf_packet(row, 123456, 7, 4242, False),
                    snd_header=ko.snd_header(0x10, 4242, -87.31))
print("wire: adpcm bytes", sum(e.size for e in enc), "state", (st.index, st.previous))
