"""Input data shared by the tests and bench.py (never imported by the product package)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def e1b_chips(path=None):
    """{prn: chips uint8[4092]} from tests/golden/e1b_ref.npz: the 50 Galileo E1-B memory codes as the
    reference's own gps/e1bcode.h produced them (tools/make_ref_golden.py).  In a deployment the caller
    hands over its own table."""
    g = np.load(path or os.path.join(GOLDEN, "e1b_ref.npz"))
    chips = np.unpackbits(g["chips_packed"], axis=1)[:, :4092]
    return {i + 1: chips[i].copy() for i in range(chips.shape[0])}


def run_fir_script(fir, script, x):
    """The script language of oracle/ref/ref_fir_main.cpp on an object with init_lp / init_hp / init_const /
    process_rr / process_rm / process_mm.  Returns the floats the reference driver would have written."""
    import numpy as np
    out, pos = [], 0
    for line in script:
        f = str(line).split()
        if f[0] in "LH":
            fn = fir.init_lp if f[0] == "L" else fir.init_hp
            out.append(np.array([fn(int(f[1]), *[float(v) for v in f[2:7]])], np.float32))
        elif f[0] == "K":
            n = int(f[1])
            fir.init_const(x[pos:pos + n], float(f[2]))
            pos += n
        else:
            n = int(f[1])
            blk = x[pos:pos + n]
            pos += n
            if f[0] == "R":
                out.append(np.asarray(fir.process_rr(blk), np.float32))
            elif f[0] == "M":
                out.append(np.asarray(fir.process_rm(blk)).astype(np.float32))
            else:
                out.append(np.asarray(fir.process_mm(blk.astype(np.int16))).astype(np.float32))
    assert pos == x.size
    return np.concatenate(out)


def run_squelch_script(sq, script, x):
    """The script language of oracle/ref/ref_squelch_main.cpp on an object with setup / set_squelch / reset /
    perform_fm (-> mono16 out, nsq_nc_sq)."""
    import numpy as np
    out, pos = [], 0
    for line in script:
        f = str(line).split()
        if f[0] == "P":
            sq.setup(float(f[1]))
        elif f[0] == "Q":
            sq.set_squelch(int(f[1]), int(f[2]))
        elif f[0] == "Z":
            sq.reset()
        else:
            n = int(f[1])
            y, rc = sq.perform_fm(x[pos:pos + n])
            pos += n
            out.append(np.concatenate([np.asarray(y).astype(np.float32), np.array([rc], np.float32)]))
    assert pos == x.size
    return np.concatenate(out)


def arm_audio_tail(P, ch, rate=12000.0, hbw=4900.0, squelch=0):
    """What the reference has always done before a channel can demodulate AM / NBFM: the post-detector filter of the passband
    command (rx/rx_sound_cmd.cpp:268-282) and the squelch of a new connection (rx/rx_sound.cpp:261-262)."""
    P.set_am_passband(ch, -hbw, hbw, rate)
    P.squelch_setup(ch, rate)
    P.squelch_set(ch, squelch, 0)


def run_fastfir_script(f, script, x):
    """The script language of oracle/ref/ref_fastfir_main.cpp on an object with window(w) / cic(on) / setup(inst, lo, hi, off, fs) /
    process(x) -> (out complex64, FirPos).  Returns the floats the reference driver would have written."""
    import numpy as np
    out, pos = [], 0
    for line in script:
        t = str(line).split()
        if t[0] == "W":
            f.window(int(t[1]))
        elif t[0] == "C":
            f.cic(int(t[1]) != 0)
        elif t[0] == "P":
            f.setup(int(t[1]), *[float(v) for v in t[2:6]])
        else:
            n = int(t[1])
            y, fp = f.process(x[pos:pos + n])
            pos += n
            out.append(np.array([y.size, fp], np.float32))
            out.append(np.ascontiguousarray(y, np.complex64).view(np.float32))
    assert pos == x.size
    return np.concatenate(out)


def fastfir_blocks(script, flat):
    """Splits the driver's float stream into [(count, FirPos, complex64 outputs)] per `D` line."""
    import numpy as np
    blocks, k = [], 0
    for line in script:
        if str(line).split()[0] != "D":
            continue
        cnt, fp = int(flat[k]), int(flat[k + 1])
        k += 2
        blocks.append((cnt, fp, flat[k:k + 2 * cnt].view(np.complex64)))
        k += 2 * cnt
    assert k == flat.size
    return blocks


def oracle_row(oracle, searcher, sat):
    """The oracle's version of what the host wrote into row `sat` of a Searcher's code table (Searcher.rows), or None when the
    row was never written -- the `code_next` of sat - 1 (the reference's Correlate() reads the next satellite's row behind a
    satellite's own for a negative Doppler bin: gps/search.cpp:471 over the doubled rows of :54)."""
    r = searcher.rows.get(int(sat))
    if r is None or sat >= searcher.max_sats:
        return None
    kind, arr, boc = r
    return oracle.code_fft(arr, boc=boc, fft_len=searcher.fft_len) if kind == "chips" else arr


def oracle_next_rows(oracle, searcher, svs):
    return [oracle_row(oracle, searcher, int(s) + 1) for s in svs]


def dpump_ref_cases(g):
    """tests/golden/dpump_ref.npz -> [(name, nch, ns, inversion, dc_i, dc_q, enabled list, bufs uint8[nbuf, bytes], rescale,
    [per buffer: {ch: (wr_pos after, ticks48, complex64[ns])}])] as the reference's own snd_service() produced them."""
    import numpy as np
    out = []
    for name in (str(n) for n in g["names"]):
        nch, ns, inv, dci, dcq, mask, nbuf = g[name + "_cfg"]
        nch, ns, inv, mask, nbuf = int(nch), int(ns), int(inv), int(mask), int(nbuf)
        en = [(mask >> ch) & 1 for ch in range(nch)]
        y, k = g[name + "_out"], 1
        per = []
        for b in range(nbuf):
            d = {}
            for ch in range(nch):
                if not en[ch]:
                    continue
                wr, hi, lo = (int(v) for v in y[k:k + 3])
                k += 3
                d[ch] = (wr, (hi << 24) | lo, y[k:k + 2 * ns].view(np.complex64))
                k += 2 * ns
            per.append(d)
        assert k == y.size
        out.append((name, nch, ns, bool(inv), float(dci), float(dcq), en, g[name + "_bufs"], float(y[0]), per))
    return out


def aper_ref_replay(g, update, report):
    """Walks tests/golden/aper_fftref.npz -- one wf_inst_t's aperture fields over the frames the reference's compute_frame() ->
    aperture_auto() (rx/rx_waterfall.cpp:1173-1272) was given -- with the caller's two halves:
        update(avg, row, algo, param, clear, start, stop, waterfall_cal) -> avg      (:1183-1222)
        report(avg, start, stop) -> (signal, noise)                                   (:1233-1271)
    and the control flow between them restated here (need / done counters :1176, the audio FFT's pixel range :1180-1181, the
    single-shot rule :1177, :1192-1195, the report clock :1224-1229).  Yields (frame index, avg, (signal, noise, done,
    report_sec, avg_clear), expected avg, expected state) after every frame."""
    import numpy as np
    IIR, MMA, EMA, OFF = 0, 1, 2, 3
    avg = np.zeros(1024, np.float32)
    st = dict(done=0, clear=0, signal=0, noise=0, report_sec=0)
    for k, (run, f, algo, param, cal, nwf, now, need) in enumerate(g["frames"]):
        algo, cal, nwf, now, need, f = int(algo), int(cal), int(nwf), int(now), int(need), int(f)
        if f == 0:
            st["clear"] = 1
        row = g["rows"][k]
        if need > st["done"]:
            single = algo == OFF
            start, stop = (256, 768) if 0 >= nwf else (0, 1024)
            if st["clear"]:
                avg = update(avg, row, MMA, 8.0, True, start, stop, cal)
                st["report_sec"], st["clear"] = now, 0
            else:
                a, p = (MMA, 8.0) if single else (algo, float(param))
                avg = update(avg, row, a, p, False, start, stop, cal)
            if now >= st["report_sec"] + (1 if single else 3):
                st["report_sec"] = now
                st["done"] += 1
                st["signal"], st["noise"] = report(avg, start, stop)
        yield k, avg, (st["signal"], st["noise"], st["done"], st["report_sec"], st["clear"]), g["avg_pwr"][k], \
            tuple(int(v) for v in g["states"][k])
