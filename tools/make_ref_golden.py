"""Generates tests/golden/*_ref.* from the REFERENCE ITSELF (run in the build container, where
/root/reference is mounted and `make -C oracle` has built oracle/_ref/ from the reference's own
sources, see oracle/build_ref.sh).  What is committed is data: inputs made here from fixed seeds and
the outputs the reference's code produced for them.

  e1b_ref.npz     the 50 Galileo E1-B memory codes: the chips E1BCODE(prn) of gps/e1bcode.h produces
                  (outputs only; the header's hex table is read from $REFERENCE at test time)
  consts_ref.json the constants of gps/gps.h, kiwi.h and the generated kiwi.gen.h
  cic_ref.json    register widths / truncations / output slices verilog/rx/cic_gen.c emits for every
                  CIC instance of kiwi.config (rx1 std + wide, rx2 std + wide, wf1)
  agc_ref.npz     CAgc (rx/CuteSDR/agc.cpp): SetParameters / ProcessData scripts, inputs, outputs
  adpcm_ref.npz   rx/csdr/ima_adpcm.cpp: i16 and u8 coder + decoder, inputs, outputs, end states
  fir_ref.npz     CFir (rx/CuteSDR/fir.cpp): InitLPFilter / InitHPFilter / InitConstFir + the three real-valued
                  ProcessFilter paths -- the designed taps (read back through an impulse), inputs, outputs
  squelch_ref.npz CSquelch (rx/CuteSDR/squelch.cpp): SetupParameters / SetSquelch / Reset / PerformFMSquelch scripts
  sndpath_ref.npz c2s_sound()'s own statements between CFastFIR and the sound packet (rx/rx_sound.cpp:676-908, cut at build time):
                  S-meter, AM / NBFM detectors with m_AM_FIR / m_Squelch, SSB AGC, de-emphasis -- every mode family, both rates
  wfcmd_ref.npz   c2s_waterfall()'s own statements for `SET zoom= start= / cf=` (rx/rx_waterfall.cpp:365-529, 756-928, cut at build time):
                  the SPI words (decimation, 48-bit NCO offset), fft_used / plot_width, fft2wf_map[], drop_sample[], fft_scale[] with masks
  sndcmd_ref.npz  rx_sound_set_freq() (rx/rx_sound_cmd.cpp): frequency -> the 48-bit phase increment it hands to spi_set3(CmdSetRXFreq)
  dpump_ref.npz   snd_service() (rx/data_pump.cpp): SPI buffers of rx_iq_t records + trailer -> in_samps rings, ticks, rescale
  chan_ref.npz    CHANNEL::Start (gps/channel.cpp): acquisition results -> the SPI commands that program a tracking channel
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("REFERENCE", "/root/reference")


def run(args, stdin=None):
    p = subprocess.run(args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return p.stdout, p.stderr


# ---- E1B ----------------------------------------------------------------------------------
out, _ = run([os.path.join(REF, "e1b_ref")])
chips = np.zeros((50, 4092), np.uint8)
for line in out.decode().splitlines():
    prn, bits = line.split()
    chips[int(prn) - 1] = np.frombuffer(bits.encode(), np.uint8) - ord("0")
# outputs only: the ICD hex table itself (text of gps/e1bcode.h) is NOT kept under tests/ -- the tests read it
# from $REFERENCE at run time where the reference is present (tests/test_ref_pins_cpu.py)
np.savez_compressed(os.path.join(GOLD, "e1b_ref.npz"), chips_packed=np.packbits(chips, axis=1))
print("e1b_ref.npz: 50 codes, E01 first 20 chips 0x%x" % int("".join(map(str, chips[0, :20])), 2))

# ---- constants ----------------------------------------------------------------------------
out, _ = run([os.path.join(REF, "gpsconst_ref")])
consts = json.loads(out.decode())
consts.pop("_")
json.dump(consts, open(os.path.join(GOLD, "consts_ref.json"), "w"), indent=1, sort_keys=True)
print("consts_ref.json: %d constants" % len(consts))

# ---- CIC shapes ---------------------------------------------------------------------------
cic = {}
for name in ("cic_rx1_12k", "cic_rx1_20k", "cic_rx2_12k", "cic_rx2_20k", "cic_wf1"):
    text = open(os.path.join(REF, "cic", name + ".vh")).read()
    m = re.search(r"N=(\d+) R=(\d+) M=1 Bin=(\d+) Bout=(\d+)", text)
    n, r, bin_, bout = map(int, m.groups())
    acc = int(re.search(r"acc_max (\d+)", text).group(1))
    integ = [int(h) + 1 for h in re.findall(r"wire signed \[(\d+):0\] integrator[1-9]\d*_data", text)]
    comb = [int(h) + 1 for h in re.findall(r"wire signed \[(\d+):0\] comb[1-9]\d*_data", text)]
    comb0 = int(re.search(r"wire signed \[(\d+):0\] comb0_data", text).group(1)) + 1
    trunc = [int(t) for t in re.findall(r"// trunc (\d+) bits", text)]
    # "out = combN[hi -: w] + combN[round_bit]", or without the rounding term when nothing is dropped
    o = re.search(r"assign out = comb(\d+)_data\[(\d+) -:(\d+)\](?: \+ comb\d+_data\[(\d+)\])?", text)
    cic[name] = {"N": n, "R": r, "Bin": bin_, "Bout": bout, "acc": acc, "integrators": integ,
                 "comb0": comb0, "combs": comb, "trunc": trunc,
                 "out": [int(x) if x is not None else -1 for x in o.groups()]}
json.dump(cic, open(os.path.join(GOLD, "cic_ref.json"), "w"), indent=1, sort_keys=True)
for k, v in cic.items():
    print("cic_ref.json:", k, v)

# ---- CAgc ---------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED00A6))


def envelope(kind, n):
    t = np.arange(n)
    if kind == "steps":
        e = np.where((t // 700) % 2 == 0, 3000.0, 30.0)
    elif kind == "decay":
        e = 20000.0 * np.exp(-t / 900.0) + 5.0
    elif kind == "silence":
        e = np.where(t < n // 3, 800.0, 0.0)
    elif kind == "ties":
        e = np.full(n, 1000.0)
    else:
        e = 10.0 ** rng.uniform(0.5, 4.3, n)
    ph = np.cumsum(rng.uniform(0.1, 0.5, n))
    x = e * np.exp(1j * ph) + (0 if kind == "ties" else 1) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


scenarios = [
    # (name, script lines; C/M lengths consume the input in order)
    ("agc_slow_cpx_steps", "steps", ["P 1 0 -100 50 6 1000 12000", "D"] + ["C 512"] * 4),
    ("agc_fast_s16_decay", "decay", ["P 1 0 -90 50 4 250 12000"] + ["M 512"] * 4),
    ("agc_hang_s16_steps", "steps", ["P 1 1 -100 50 6 500 12000", "D"] + ["M 512"] * 4),
    ("manual_gain_cpx", "noise", ["P 0 0 -100 70 6 1000 12000"] + ["C 512"] * 2),
    ("wide_rate_s16_silence", "silence", ["P 1 0 -100 50 6 1000 20250", "D"] + ["M 512"] * 4),
    ("retune_sequence", "noise", ["P 1 0 -100 50 6 1000 12000", "C 512", "P 1 0 -100 50 6 1000 12000", "C 300",
                                  "P 1 0 -80 50 2 100 12000", "C 212", "P 0 0 -80 35 2 100 12000", "M 512",
                                  "P 1 1 -110 35 10 2000 20250", "D", "M 512", "C 1"]),
    ("exact_ties_cpx", "ties", ["P 1 0 -100 50 6 1000 12000"] + ["C 512"] * 2),
]
agc = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, kind, script in scenarios:
        n = sum(int(l.split()[1]) for l in script if l[0] in "CM")
        x = envelope(kind, n)
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        x.tofile(os.path.join(tmp, "in.bin"))
        run([os.path.join(REF, "agc_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"),
             os.path.join(tmp, "out.bin")])
        y = np.fromfile(os.path.join(tmp, "out.bin"), np.float32)
        agc[name + "_script"] = np.array(script)
        agc[name + "_in"] = x
        agc[name + "_out"] = y
        print("agc_ref.npz: %-24s %5d samples in, %5d floats out" % (name, n, y.size))
agc["names"] = np.array([s[0] for s in scenarios])
np.savez_compressed(os.path.join(GOLD, "agc_ref.npz"), **agc)

# ---- IMA ADPCM ------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED00AD))
t = np.arange(4096)
sig = np.concatenate([
    8000.0 * np.sin(2 * np.pi * 0.013 * t) + 200.0 * rng.standard_normal(t.size),
    np.where((t // 37) % 2 == 0, 32767.0, -32768.0),           # full-scale square wave
    np.zeros(t.size),                                            # silence
    rng.integers(-32768, 32768, t.size).astype(np.float64),     # white, full range
]).round().clip(-32768, 32767).astype(np.int16)
ad = {"i16_in": sig}
enc, err = run([os.path.join(REF, "adpcm_ref"), "enc_i16", "0", "0", "512"], sig.tobytes())
ad["i16_enc"] = np.frombuffer(enc, np.uint8)
ad["i16_enc_state"] = np.array([int(v) for v in err.split()], np.int32)
dec, err = run([os.path.join(REF, "adpcm_ref"), "dec_i16", "0", "0", "256"], enc)
ad["i16_dec"] = np.frombuffer(dec, np.int16)
ad["i16_dec_state"] = np.array([int(v) for v in err.split()], np.int32)
# resumed state (the audio_adpcm_state message), odd block length
enc2, err = run([os.path.join(REF, "adpcm_ref"), "enc_i16", "37", "-1234", "170"], sig[:3400].tobytes())
ad["i16_enc_resumed"] = np.frombuffer(enc2, np.uint8)
ad["i16_enc_resumed_state"] = np.array([int(v) for v in err.split()], np.int32)
rows = np.stack([
    np.clip(120 + 60 * np.sin(2 * np.pi * np.arange(1024) / 200.0) + rng.normal(0, 8, 1024), 0, 255),
    rng.integers(0, 256, 1024).astype(np.float64),
    np.full(1024, 55.0),
    np.where(np.arange(1024) % 2 == 0, 255.0, 0.0),
]).round().astype(np.uint8)
ad["u8_rows"] = rows
u8enc, u8dec = [], []
for r in rows:
    # compute_frame(): 10 pad bytes (copies of the first pixel) + the row, fresh state per row
    padded = np.concatenate([np.full(10, r[0], np.uint8), r])
    e, _ = run([os.path.join(REF, "adpcm_ref"), "enc_u8", "0", "0", "1034"], padded.tobytes())
    u8enc.append(np.frombuffer(e, np.uint8))
    d, _ = run([os.path.join(REF, "adpcm_ref"), "dec_u8", "0", "0", "517"], e)
    u8dec.append(np.frombuffer(d, np.uint8))
ad["u8_enc"] = np.stack(u8enc)
ad["u8_dec"] = np.stack(u8dec)
np.savez_compressed(os.path.join(GOLD, "adpcm_ref.npz"), **ad)
print("adpcm_ref.npz: %d int16 samples, %d rows; end state %s" % (sig.size, rows.shape[0], ad["i16_enc_state"]))

# ---- CFir -----------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED00F1))
# the de-emphasis tables (rx/rx_filter.h:29-73): read as numbers, handed to the reference's InitConstFir
text = open(os.path.join(REFERENCE, "rx", "rx_filter.h")).read()
deemp = {}
for name in ("nfm_deemp_12000", "nfm_deemp_20250", "am_ssb_deemp_12000", "am_ssb_deemp_20250"):
    body = re.search(r"const float %s\[N_NFM_DEEMP\]\[N_DEEMP_TAPS\] = \{(.*?)\n\};" % name, text, re.S).group(1)
    rows = re.findall(r"\{([^{}]*)\}", body)
    tab = np.array([[float(v) for v in re.sub(r"//[^\n]*", "", r).replace("\n", " ").split(",") if v.strip()] for r in rows], np.float32)
    assert tab.shape == (2, 79), (name, tab.shape)
    deemp[name] = tab


def audio(kind, n):
    t = np.arange(n)
    if kind == "am":                      # what the AM detector hands over: envelope minus DC, a few thousand counts
        return (6000.0 * np.sin(2 * np.pi * 0.031 * t) * (1 + 0.3 * np.sin(2 * np.pi * 0.0007 * t)) + 400.0 * rng.standard_normal(n)).astype(np.float32)
    if kind == "loud":                    # past the int16 range: the (TYPEMONO16) cast of an out-of-range float
        return (60000.0 * np.sin(2 * np.pi * 0.011 * t) + 100.0 * rng.standard_normal(n)).astype(np.float32)
    if kind == "s16":                     # mono16 audio (SSB / detector outputs)
        return np.round(9000.0 * np.sin(2 * np.pi * 0.05 * t) + 3000.0 * rng.standard_normal(n)).clip(-32768, 32767).astype(np.float32)
    if kind == "fm":                      # the NBFM detector's output: clipped at +-8192 (rx_sound.cpp:860)
        return np.clip(5000.0 * rng.standard_normal(n), -8192, 8192).astype(np.float32)
    raise ValueError(kind)


def impulse(n):
    x = np.zeros(n, np.float32)
    x[0] = 1.0
    return x


# (name, [(script line, input floats or None)]): `R` after a design line with an impulse reads the taps back
fir_scen = [
    # m_AM_FIR as rx_sound_cmd.cpp:270-282 designs it: AM 9.8 kHz (hbw 4900, stop min(8820, 6000)), AMN 5 kHz, 12 kHz rate
    ("am_fir_am_12k", [("L 0 1.0 50.0 4900 6000 12000", None), ("R 100", impulse(100)),
                       ("L 0 1.0 50.0 4900 6000 12000", None)] + [("M 512", audio("am", 512)) for _ in range(4)]),
    ("am_fir_amn_12k", [("L 0 1.0 50.0 2500 4500 12000", None), ("R 100", impulse(100)),
                        ("L 0 1.0 50.0 2500 4500 12000", None), ("M 512", audio("am", 512)), ("M 300", audio("am", 300)),
                        ("M 212", audio("am", 212)), ("M 1", audio("am", 1)), ("M 511", audio("loud", 511))]),
    ("am_fir_am_20k", [("L 0 1.0 50.0 6000 10125 20250", None), ("R 100", impulse(100)),
                       ("L 0 1.0 50.0 6000 10125 20250", None)] + [("M 512", audio("am", 512)) for _ in range(2)]),
    # hbw == frate / 2: Fstop == Fpass, an infinite tap estimate whose (int) is undefined; this build (x86-64) gives 9 taps
    ("am_fir_full_band", [("L 0 1.0 50.0 6000 6000 12000", None), ("R 20", impulse(20)), ("M 256", audio("am", 256))]),
    ("lp_forced_taps_weak_stop", [("L 33 0.5 35.0 1000 3000 12000", None), ("R 40", impulse(40)), ("R 300", audio("am", 300)),
                                  ("L 0 1.0 15.0 1000 1200 12000", None), ("R 100", impulse(100))]),
    # CSquelch::InitNoiseSquelch (squelch.cpp:137) at both sound rates
    ("hp_squelch_12k", [("H 0 1.0 50.0 2400 1950 12000", None), ("R 100", impulse(100)),
                        ("H 0 1.0 50.0 2400 1950 12000", None), ("R 512", audio("fm", 512)), ("R 170", audio("fm", 170)),
                        ("R 512", audio("fm", 512))]),
    ("hp_squelch_20k", [("H 0 1.0 50.0 2400 1950 20250", None), ("R 100", impulse(100)), ("R 512", audio("fm", 512))]),
]
for tname, tab in deemp.items():
    fs = 12000 if tname.endswith("12000") else 20250
    for k in range(2):
        fir_scen.append(("%s_%d" % (tname, k), [("K 79 %d" % fs, tab[k]), ("R 80", impulse(80)), ("K 79 %d" % fs, tab[k]),
                                                ("S 512", audio("s16", 512)), ("S 340", audio("s16", 340)),
                                                ("S 172", audio("s16", 172)), ("S 512", audio("s16", 512))]))
fir = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, steps in fir_scen:
        script = [s for s, _ in steps]
        x = np.concatenate([np.asarray(v, np.float32) for _, v in steps if v is not None])
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        x.tofile(os.path.join(tmp, "in.bin"))
        run([os.path.join(REF, "fir_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")])
        y = np.fromfile(os.path.join(tmp, "out.bin"), np.float32)
        fir[name + "_script"] = np.array(script)
        fir[name + "_in"] = x
        fir[name + "_out"] = y
        print("fir_ref.npz: %-24s %5d floats in, %5d floats out" % (name, x.size, y.size))
fir["names"] = np.array([s[0] for s in fir_scen])
np.savez_compressed(os.path.join(GOLD, "fir_ref.npz"), **fir)

# ---- CSquelch -------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED0059))


def fm_noise(n, level=5000.0):
    return np.clip(level * rng.standard_normal(n), -8192, 8192).astype(np.float32)


def fm_voice(n, f=0.07):
    return (4000.0 * np.sin(2 * np.pi * f * np.arange(n)) + 30.0 * rng.standard_normal(n)).astype(np.float32)


sq_scen = [
    # rx_sound.cpp:261-262: SetupParameters, SetSquelch(0, 0) -- always open
    ("open_default", ["P 12000", "Q 0 0"] + ["F 512"] * 3, [fm_noise(512), fm_voice(512), fm_noise(512)]),
    # mid threshold: noise closes it, a quiet carrier opens it, noise closes it again (hysteresis both ways)
    ("threshold_80", ["P 12000", "Q 80 0"] + ["F 512"] * 12,
     [fm_noise(512)] * 0 + [fm_noise(512) for _ in range(3)] + [fm_voice(512) for _ in range(5)] + [fm_noise(512) for _ in range(4)]),
    ("threshold_99_forced", ["P 12000", "Q 0 0", "F 512", "Q 99 0", "F 512", "F 512", "Q 0 0", "F 512"],
     [fm_voice(512) for _ in range(4)]),
    ("explicit_max_and_reset", ["P 12000", "Q 30 3000", "F 512", "F 512", "F 512", "Z", "F 512", "Q 60 3000", "F 170", "F 342", "F 512"],
     [fm_voice(512), fm_noise(512, 1500.0), fm_noise(512, 6000.0), fm_voice(512), fm_noise(170), fm_voice(342), fm_voice(512)]),
    ("wide_rate_20250", ["P 20250", "Q 75 0"] + ["F 512"] * 6,
     [fm_noise(512) for _ in range(2)] + [fm_voice(512, 0.03) for _ in range(3)] + [fm_noise(512)]),
    ("near_threshold_levels", ["P 12000", "Q 85 0"] + ["F 512"] * 16,
     [fm_noise(512, 1200.0 + 250.0 * k) for k in range(10)] + [fm_noise(512, 3450.0 - 450.0 * k) for k in range(6)]),
]
sq = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, script, blocks in sq_scen:
        x = np.concatenate(blocks)
        assert x.size == sum(int(l.split()[1]) for l in script if l[0] == "F"), name
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        x.tofile(os.path.join(tmp, "in.bin"))
        run([os.path.join(REF, "squelch_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")])
        y = np.fromfile(os.path.join(tmp, "out.bin"), np.float32)
        sq[name + "_script"] = np.array(script)
        sq[name + "_in"] = x
        sq[name + "_out"] = y
        print("squelch_ref.npz: %-24s %5d floats in, %5d floats out" % (name, x.size, y.size))
sq["names"] = np.array([s[0] for s in sq_scen])
np.savez_compressed(os.path.join(GOLD, "squelch_ref.npz"), **sq)

# ---- c2s_sound()'s signal path, the reference's own statements (rx/rx_sound.cpp:676-908 cut at build time) ---------------------
# mode numbers: rx/mode.h:69-70
M_AM, M_AMN, M_USB, M_LSB, M_CW, M_CWN, M_NBFM, M_IQ, M_DRM, M_USN, M_LSN, M_SAM, M_SAU, M_SAL, M_SAS, M_QAM, M_NNFM = range(17)
rng = np.random.Generator(np.random.PCG64(0x5EED0061))


def cpx_noise(n, level):
    return (level * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


def am_sig(n, amp=4000.0, depth=0.6):
    t = np.arange(n)
    env = amp * (1 + depth * np.sin(2 * np.pi * t / 37.0) + 0.2 * np.sin(2 * np.pi * t / 11.0))
    return (env * np.exp(2j * np.pi * 0.07 * t) + cpx_noise(n, 20.0)).astype(np.complex64)


def fm_sig(n, quiet=True):
    t = np.arange(n)
    ph = 2 * np.pi * np.cumsum(0.02 * np.sin(2 * np.pi * t / 40.0))
    return (5000.0 * np.exp(1j * ph) + cpx_noise(n, 6.0)).astype(np.complex64) if quiet else cpx_noise(n, 3000.0)


def ssb_sig(n, amp=2500.0):
    t = np.arange(n)
    env = amp * np.where((t // 700) % 2 == 0, 0.1, 1.0) * np.exp(-t / 3000.0)
    return (env * (np.exp(2j * np.pi * 0.031 * t) + 0.4 * np.exp(2j * np.pi * 0.113 * t)) + cpx_noise(n, 8.0)).astype(np.complex64)


def passband(lo, hi, rate):
    """hbw / stop of m_AM_FIR as rx_sound_cmd.cpp:268-282 derives them from the passband (float hbw, double frate)"""
    hbw = np.float32(min(float(np.float32(max(abs(hi), abs(lo)))), rate / 2))
    stop = np.float32(min(float(np.float32(float(hbw) * 1.8)), rate / 2))
    return "L %r %r" % (float(hbw), float(stop))


def blocks_of(sig, lens):
    out, pos = [], 0
    for n in lens:
        out.append(sig[pos:pos + n]); pos += n
    return out


B6 = [512] * 6
RAG = [512, 300, 212, 170, 342, 512, 1, 511, 512]


def B(n=512, k=1):
    return ["B %d" % n] * k


sp_scen = [
    # name, rate, (lo, hi) of the passband, script (without R / L; "B n" = a CFastFIR block, grouped into packets below), input blocks in B order
    ("am_default", 12000.0, (-4900.0, 4900.0), ["A 1 0 -100 50 6 1000", "M %d" % M_AM] + B(512, 8), blocks_of(am_sig(4096), [512] * 8)),
    ("amn_deemp", 12000.0, (-2500.0, 2500.0), ["A 1 0 -100 50 6 1000", "E 1 0", "M %d" % M_AMN] + B(512, 4) + ["E 2 0"] + B(512, 4),
     blocks_of(am_sig(4096, 9000.0, 0.9), [512] * 8)),
    ("usb_hang_ragged_raw_le", 12000.0, (300.0, 2700.0), ["W 0 1", "A 1 1 -90 50 3 500", "M %d" % M_USB] + ["B %d" % n for n in RAG], blocks_of(ssb_sig(sum(RAG)), RAG)),
    ("lsb_cw_manual_raw_be", 12000.0, (-2700.0, -300.0), ["W 0 0", "A 0 0 -100 70 6 1000", "M %d" % M_LSB, "B 512", "B 512", "M %d" % M_CW, "B 512", "A 1 0 -130 50 0 100",
                                                         "M %d" % M_CWN, "B 512", "B 512", "M %d" % M_USN, "B 512", "M %d" % M_LSN, "B 512"], blocks_of(ssb_sig(3584, 6000.0), [512] * 7)),
    ("ssb_deemp", 12000.0, (300.0, 2700.0), ["A 1 0 -100 50 6 1000", "E 2 0", "M %d" % M_USB] + B(512, 8), blocks_of(ssb_sig(4096, 12000.0), [512] * 8)),
    ("nbfm_squelch_80", 12000.0, (-6000.0, 6000.0), ["A 1 0 -100 50 6 1000", "Q 80 0", "M %d" % M_NBFM] + B(512, 12),
     [fm_sig(512, False) for _ in range(3)] + [fm_sig(512) for _ in range(5)] + [fm_sig(512, False) for _ in range(4)]),
    ("nnfm_deemp_forced_shut", 12000.0, (-3000.0, 3000.0), ["A 1 0 -100 50 6 1000", "E 0 1", "Q 0 0", "M %d" % M_NNFM] + B(512, 4) + ["Q 99 0"] + B(512, 4)
     + ["Q 0 0"] + B(512, 4) + ["E 0 2"] + B(512, 4), [fm_sig(512) for _ in range(16)]),
    ("iq_le_then_be", 12000.0, (-5000.0, 5000.0), ["A 1 0 -100 50 6 1000", "W 1 1", "M %d" % M_IQ] + B(512, 3) + ["W 1 0"] + B(512, 3) + ["A 0 0 -100 90 6 1000"] + B(512, 2),
     blocks_of(am_sig(4096, 300.0), [512] * 8)),
    ("rate_20250_every_family", 20250.0, (-6000.0, 6000.0), ["A 1 1 -60 50 10 2000", "E 1 1", "M %d" % M_AM] + B(512, 4) + ["M %d" % M_NBFM, "Q 75 0"] + B(512, 4)
     + ["M %d" % M_USB] + B(512, 4) + ["M %d" % M_IQ] + B(512, 2),
     blocks_of(am_sig(2048), [512] * 4) + [fm_sig(512, False), fm_sig(512), fm_sig(512), fm_sig(512)] + blocks_of(ssb_sig(2048), [512] * 4) + blocks_of(am_sig(1024), [512] * 2)),
    ("mode_hops_state_carried", 12000.0, (-4000.0, 4000.0), ["W 0 1", "A 1 0 -100 50 6 1000", "E 1 2", "Q 70 0"] + sum([["M %d" % m, "B 400", "B 112"] for m in
     (M_AM, M_NBFM, M_USB, M_AM, M_IQ, M_NNFM, M_CW, M_AMN, M_IQ)], []), sum([blocks_of(sg, [400, 112]) for sg in
     (am_sig(512), fm_sig(512), ssb_sig(512), am_sig(512, 800.0), am_sig(512), fm_sig(512, False), ssb_sig(512), am_sig(512), am_sig(512))], [])),
    ("silence_full_scale_overflow", 12000.0, (-4900.0, 4900.0), ["A 1 0 -100 50 6 1000", "W 0 0", "M %d" % M_AM, "B 512", "V 1", "B 512", "M %d" % M_NBFM, "B 512", "V 0", "B 512",
                                                                "M %d" % M_USB, "B 512", "B 512", "M %d" % M_IQ, "B 512", "B 512"],
     [np.zeros(512, np.complex64), (30000.0 * np.exp(2j * np.pi * 0.05 * np.arange(512))).astype(np.complex64)] * 4),
]
sp = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, rate, (lo, hi), lines, blocks in sp_scen:
        # group the blocks into packets as the server's `do { ... } while (bc < LOOP_BC)` does: 4 x 512 samples of compressed mono audio,
        # one block otherwise (1024 bytes of raw mono, 2048 of IQ); a command between blocks ends the packet (the scenarios place them there)
        # the firmware mode of the scenario (config.h:31-34; nrx_samps = 680 / channels), the audio decimation, the ADC clock
        fw_sel, nrx, adc_base = {"am_default": (0, 170, 66.6666e6), "amn_deemp": (1, 85, 66.6666e6), "iq_le_then_be": (3, 48, 66.6672e6),
                                 "mode_hops_state_carried": (1, 85, 66.6660e6)}.get(name, (2, 226, 66.6666e6) if rate > 15000 else (0, 170, 66.6666e6))
        decim = int(round(adc_base / rate))
        script, pend, comp, mode = ["R %r %d %d %d %r" % (rate, fw_sel, nrx, decim, adc_base), passband(lo, hi, rate)], [], 1, M_USB
        tick, nblk_seen = [(1 << 48) - 40 * decim * nrx], [0]       # the 48-bit tick counter wraps inside every scenario

        def flush():
            for _ in pend:                                         # one T line per block: the buffer's ticks, CFastFIR's position
                tick[0] = (tick[0] + 3 * decim * nrx) & ((1 << 48) - 1)
                script.append("T %d %d" % (tick[0], (nblk_seen[0] * 166) % nrx))
                nblk_seen[0] += 1
                if nblk_seen[0] == 3:                              # the first GPS solution arrives after three blocks, a later one after nine
                    script.insert(len(script) - 1, "C %d %r" % ((tick[0] - 7 * decim * nrx) & ((1 << 48) - 1), 345600.25))
                if nblk_seen[0] == 9:
                    script.insert(len(script) - 1, "C %d %r" % ((tick[0] - 2 * decim * nrx) & ((1 << 48) - 1), 604799.9995))
            script.append("P " + " ".join(str(v) for v in pend))
            del pend[:]

        for ln in lines + ["#end"]:
            if ln[0] == "B":
                pend.append(int(ln.split()[1]))
                if len(pend) == (4 if comp and mode not in (M_IQ, M_DRM) else 1):
                    flush()
                continue
            if pend:
                flush()
            if ln[0] == "W":
                comp = int(ln.split()[1])
            if ln[0] == "M":
                mode = int(ln.split()[1])
            if ln[0] != "#":
                script.append(ln)
        x = np.concatenate(blocks).astype(np.complex64)
        assert x.size == sum(sum(int(v) for v in l.split()[1:]) for l in script if l[0] == "P"), name
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        x.tofile(os.path.join(tmp, "in.bin"))
        run([os.path.join(REF, "sndpath_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")])
        y = np.fromfile(os.path.join(tmp, "out.bin"), np.float32)
        sp[name + "_script"] = np.array(script)
        sp[name + "_band"] = np.array([rate, lo, hi], np.float64)
        sp[name + "_in"] = x
        sp[name + "_out"] = y
        print("sndpath_ref.npz: %-28s %5d samples in, %2d packets, %5d floats out" % (name, x.size, sum(l[0] == "P" for l in script), y.size))
sp["names"] = np.array([s_[0] for s_ in sp_scen])
np.savez_compressed(os.path.join(GOLD, "sndpath_ref.npz"), **sp)

# ---- c2s_waterfall()'s `SET zoom=` case and its map / scale / mask construction, the reference's own statements -------------------
# (rx/rx_waterfall.cpp:365-529, 756-928 cut at build time).  One connection per configuration; commands in sequence, as a client zooms
# and pans (the case block sends the decimation word only when the zoom changes).
rng = np.random.Generator(np.random.PCG64(0x5EED0062))
MAXZ, WIDTH = 14, 1024
wc_scen = []
for cname, adc, srate, inv, masks in (("kiwi_30MHz", 66.6666e6, 30.0e6, 0, []), ("kiwi_32MHz", 66.672e6, 32.0e6, 0, []),
                                       ("inverted_30MHz", 66.6660e6, 30.0e6, 1, []),
                                       ("masked_30MHz", 66.6666e6, 30.0e6, 0, [(7000000, 7100000), (14250000, 14250500), (0, 30000)]),
                                       ("masked_inverted_32MHz", 66.670e6, 32.0e6, 1, [(3500000, 3600000), (28000000, 29700000)])):
    cmds = []
    for z in range(MAXZ + 1):                                        # every zoom: the left edge, a random start, the right edge and beyond
        maxstart = (WIDTH << MAXZ) - (WIDTH << (MAXZ - z))
        for st in (0.0, float(rng.integers(0, maxstart + 1)), float(maxstart), float(maxstart) + 4096.0, float(rng.integers(0, maxstart + 1)) + 0.5):
            cmds.append("SET zoom=%d start=%r" % (z, st))
    for _ in range(30):                                              # the cf= form, zoom out of range clamped
        cmds.append("SET zoom=%d cf=%r" % (int(rng.integers(-1, MAXZ + 3)), float(np.round(rng.uniform(0.0, srate / 1000.0), 3))))
    for z, st in ((5, 1234567.0), (5, 1234567.0), (5, 1234999.0), (6, 1234999.0), (0, 0.0)):     # pans without a zoom change, a repeat
        cmds.append("SET zoom=%d start=%r" % (z, st))
    wc_scen.append((cname, adc, srate, inv, masks, cmds))
wc = {}
with tempfile.TemporaryDirectory() as tmp:
    for cname, adc, srate, inv, masks, cmds in wc_scen:
        script = ["C %r %r %d 9" % (adc, srate, inv)] + ["X %d %d" % m for m in masks] + ["K " + c for c in cmds]
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        run([os.path.join(REF, "wfcmd_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "out.bin")])
        y = np.fromfile(os.path.join(tmp, "out.bin"), np.float64)
        # typed and compact: one header row per command, the maps only where the reference rebuilt them
        hdr, spi, offs, maps, drops, scales, div2s, pos = [], [], [], [], [], [], [], 0
        for _ in cmds:
            nspi = int(y[pos]); pos += 1
            calls = y[pos:pos + 4 * nspi].reshape(nspi, 4).astype(np.int64); pos += 4 * nspi
            zoom, start, wait_ms, wait_us, fft_used, pw, pwc = (int(v) for v in y[pos:pos + 7])
            off, limit, had = np.float32(y[pos + 7]), int(y[pos + 8]), int(y[pos + 9]); pos += 10
            assert fft_used >= pw, "FFT < plot does not occur at these clocks"
            if had:
                maps.append(y[pos:pos + fft_used].astype(np.int64).astype(np.uint16)); pos += fft_used     # u2_t table: -1 is 65535
                drops.append(y[pos:pos + pwc].astype(np.uint16)); pos += pwc
            sc = np.zeros(WIDTH, np.float32); sc[:pwc] = y[pos:pos + pwc]; pos += pwc
            d2 = np.zeros(WIDTH, np.float32); d2[:pwc] = y[pos:pos + pwc]; pos += pwc
            hdr.append([nspi, zoom, start, wait_ms, wait_us, fft_used, pw, pwc, limit, had])
            spi.append(np.vstack([calls, np.full((2 - nspi, 4), -2, np.int64)]) if nspi < 2 else calls)
            offs.append(off); scales.append(sc); div2s.append(d2)
        assert pos == y.size
        wc[cname + "_cfg"] = np.array([adc, srate, inv], np.float64)
        wc[cname + "_masks"] = np.array(masks, np.int64).reshape(-1, 2)
        wc[cname + "_cmds"] = np.array(cmds)
        wc[cname + "_hdr"] = np.array(hdr, np.int64)
        wc[cname + "_spi"] = np.array(spi, np.int64)                  # [ncmd, 2, (cmd, wparam, lparam, w2param)]; -2 rows: no call
        wc[cname + "_fft_offset"] = np.array(offs, np.float32)
        wc[cname + "_maps"] = np.concatenate(maps)
        wc[cname + "_drops"] = np.concatenate(drops)
        wc[cname + "_scale"] = np.array(scales)
        wc[cname + "_div2"] = np.array(div2s)
        print("wfcmd_ref.npz: %-24s %3d commands, %d map rebuilds, %d masked pixels" % (
            cname, len(cmds), len(maps), int(sum((s_[:h[7]] == 0).sum() for s_, h in zip(scales, hdr)))))
wc["names"] = np.array([c[0] for c in wc_scen])
np.savez_compressed(os.path.join(GOLD, "wfcmd_ref.npz"), **wc)

# ---- rx_sound_set_freq(): the audio NCO's phase increment (row D6) -------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED0063))
fq = []
for adc, srate in ((66.6666e6, 30.0e6), (66.672e6, 32.0e6), (66.66599e6, 30.0e6)):
    for inv in (0, 1):
        for f in [0.0, 0.001, 7020.0, 14250.5, 29999.999, srate / 1000.0] + [float(np.round(rng.uniform(0.0, srate / 1000.0), 3)) for _ in range(60)]:
            fq.append((f, adc, srate, inv))
with tempfile.TemporaryDirectory() as tmp:
    open(os.path.join(tmp, "s.txt"), "w").write("".join("F %r %r %r %d\n" % q for q in fq))
    run([os.path.join(REF, "sndcmd_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "out.bin")])
    y = np.fromfile(os.path.join(tmp, "out.bin"), np.float64).reshape(-1, 4)
assert y.shape[0] == len(fq) and np.all(y[:, 0] == 1) and np.all(y[:, 1] == 3)
# ... and the passband statements of the `SET mod= low_cut= high_cut=` handler (rx_sound_cmd.cpp:243-272, 276-286, cut at build time)
bands = [(-4900.0, 4900.0, 12000.0), (-6000.0, 6000.0, 12000.0), (300.0, 2700.0, 12000.0), (-2700.0, -300.0, 12000.0), (-2500.0, 2500.0, 12000.0),
         (-1200.0, 1250.0, 12000.0), (450.0, 550.0, 12000.0), (-5999.0, 5999.0, 12000.0), (-9000.0, 2000.0, 12000.0), (100.0, 8000.0, 12000.0),
         (0.0, 0.0, 12000.0), (-6000.0, 6000.0, 20250.0), (-10124.0, 10124.0, 20250.0), (-12000.0, 12000.0, 20250.0), (300.0, 3600.0, 20250.0),
         (-5623.0, 5629.5, 20250.0), (0.0, 40.0, 12000.0)]
bands += [(float(-np.round(rng.uniform(0, 7000), 1)), float(np.round(rng.uniform(0, 7000), 1)), 12000.0) for _ in range(20)]
with tempfile.TemporaryDirectory() as tmp:
    open(os.path.join(tmp, "s.txt"), "w").write("".join("B %r %r %r\n" % b for b in bands))
    run([os.path.join(REF, "sndcmd_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "out.bin")])
    yb = np.fromfile(os.path.join(tmp, "out.bin"), np.float64).reshape(len(bands), 134)
np.savez_compressed(os.path.join(GOLD, "sndcmd_ref.npz"), calls=np.array(fq, np.float64),
                    i_phase=(y[:, 2].astype(np.uint64) << np.uint64(16)) | y[:, 3].astype(np.uint64),
                    bands=np.array(bands, np.float64), band_out=yb[:, :6], am_fir=yb[:, 6:].astype(np.float32))
print("sndcmd_ref.npz: %d rx_sound_set_freq calls, %d passbands" % (len(fq), len(bands)))

# ---- data pump unpack -----------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED00D9))
dp_scen = [
    # name, rx_chans, nrx_samps ((2047 - 5) / 3 / rx_chans, config.h:40), inversion, dc_i, dc_q, enabled mask, buffers
    ("rx4_170", 4, 170, 0, 0.0, 0.0, 0b1111, 3),
    ("rx8_85_some_disabled", 8, 85, 0, 0.0, 0.0, 0b10110101, 2),
    ("rx14_48", 14, 48, 0, 0.0, 0.0, (1 << 14) - 1, 2),
    ("rx3_226_inverted_dc", 3, 226, 1, 12.5, -7.25, 0b111, 34),          # spectral inversion, DC offsets, the 32-deep ring wraps
]
dp = {"names": np.array([s[0] for s in dp_scen])}
with tempfile.TemporaryDirectory() as tmp:
    for name, nch, ns, inv, dci, dcq, mask, nbuf in dp_scen:
        bufs = rng.integers(0, 256, (nbuf, ns * nch * 6 + 10), dtype=np.uint8)
        bufs[0, :6 * nch] = np.tile(np.array([0x00, 0x00, 0x00, 0x80, 0x00, 0x80], np.uint8), nch)   # i = q = -2^23: the sign extension's edge
        script = ["G %d %d %d %r %r %d" % (nch, ns, inv, dci, dcq, mask), "R"] + ["S"] * nbuf
        open(os.path.join(tmp, "s.txt"), "w").write("\n".join(script) + "\n")
        bufs.tofile(os.path.join(tmp, "in.bin"))
        run([os.path.join(REF, "dpump_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")])
        y = np.fromfile(os.path.join(tmp, "out.bin"), np.float32)
        dp[name + "_cfg"] = np.array([nch, ns, inv, dci, dcq, mask, nbuf], np.float64)
        dp[name + "_bufs"] = bufs
        dp[name + "_out"] = y
        print("dpump_ref.npz: %-24s %d buffers of %d bytes, %d floats out, rescale %.9g" % (name, nbuf, bufs.shape[1], y.size, y[0]))
np.savez_compressed(os.path.join(GOLD, "dpump_ref.npz"), **dp)

# ---- CHANNEL::Start ---------------------------------------------------------------------------------
rng = np.random.Generator(np.random.PCG64(0x5EED00C5))
calls = [(0, 0, 1000, 501000, 6, 4808, 55)]                       # BASELINE configs[0]'s result, half a second after the samples
for k in range(200):
    e1b = k % 3 == 2
    sat = int(rng.integers(36, 59)) if e1b else int(rng.integers(0, 36))
    t0 = int(rng.integers(0, 1 << 30))
    dt = int(rng.choice([0, 1, 999, 20000, 500000, 3000000, 60000000]) + rng.integers(0, 1000))
    calls.append((int(rng.integers(0, 12)), sat, t0, (t0 + dt) & 0xffffffff, int(rng.integers(-20, 21)),
                  4 * int(rng.integers(0, 16368 if e1b else 4092)), int(rng.integers(16, 200))))
calls += [(1, 5, 0, 0, 0, 0, 20), (2, 40, 0, 0, 0, 0, 20), (3, 7, 0, 1000000, 20, 16364, 30), (4, 50, 0, 1000000, -20, 65468, 30)]
with tempfile.TemporaryDirectory() as tmp:
    open(os.path.join(tmp, "s.txt"), "w").write("".join("S %d %d %d %d %d %d %d\n" % c for c in calls))
    run([os.path.join(REF, "chan_ref"), os.path.join(tmp, "s.txt"), os.path.join(tmp, "out.bin")])
    y = np.fromfile(os.path.join(tmp, "out.bin"), np.float64)
cmds, k = [], 0
for c in calls:
    n = int(y[k]); k += 1
    cmds.append(y[k:k + 3 * n].reshape(n, 3).astype(np.int64)); k += 3 * n
assert k == y.size
np.savez_compressed(os.path.join(GOLD, "chan_ref.npz"), calls=np.array(calls, np.int64), ncmds=np.array([c.shape[0] for c in cmds], np.int32),
                    cmds=np.concatenate(cmds))
print("chan_ref.npz: %d ChanStart calls, %d SPI commands; first: %s" % (len(calls), sum(c.shape[0] for c in cmds), cmds[0].tolist()))
