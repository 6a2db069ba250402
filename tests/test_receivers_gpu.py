"""BASELINE configs[3] at the bench's own shape: bench.ReceiverBank (the object `bench.py --workload
receivers` times) stepped three times over its 2^22-sample ADC block with (a) a slice of the
receiver set that straddles a rank boundary (receivers 123..132 of the 1024: every zoom 1..10, ranks 0
and 1 of the 8 x 128 sharding) and (b) rank 1's whole slice of 128 receivers, and EVERY stage of EVERY receiver checked against the oracle fed the same stream
with its state carried from step to step: both DDCs bit-exact on all their output, the frame the
waterfall took, the u8 row, the wf_pkt_t, the unpacked audio samples, CFastFIR, CAgc mono16 and the
ADPCM payload."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEPS = 3


@pytest.mark.parametrize("NR,FIRST", [(10, 123),      # every zoom once, straddling the rank 0 / rank 1 boundary
                                      (128, 128)])    # rank 1's whole slice: the per-GPU shape of configs[3]
def test_receiver_bank_every_stage_every_receiver(gpu_ctx, oracle, NR, FIRST):
    import torch
    import bench
    from flydog_sdr_gps_amd import wf
    from tests.test_wf_gpu import check_row, db_bound, oracle_frame
    dev = torch.device("cuda", 0)
    bank = bench.ReceiverBank(0, dev, NR, 1 << 22, FIRST)
    try:
        assert sorted(set(p.zoom for p in bank.params)) == list(range(1, 11))
        adc = bank.adc_host
        tables = (wf.window_functions(), wf.cic_comp_table())
        wf_st, rx_st = [None] * NR, [None] * NR
        fir_st = [oracle.fir_new_state() for _ in range(NR)]
        agcs = [oracle.Agc() for _ in range(NR)]
        for a in agcs:
            a.set_parameters(True, False, -100, 50, 6, 1000, bank.fs)
        ad_st = [None] * NR
        pool = ThreadPoolExecutor(8)                             # the oracle's C calls release the GIL
        audio_blocks = 0
        for step in range(STEPS):
            bank.step()
            torch.cuda.synchronize(dev)
            nrec, nout, nw = bank.last["nrec"], bank.last["nout"], bank.last["nw"]
            g = {k: getattr(bank, k).cpu().numpy() for k in ("wf_iq", "frames", "rows", "pkts", "raw", "xin", "firo", "s16", "pay")}

            def wf_ref(ch):
                p = bank.params[ch]
                return oracle.ddc_wf(adc, p.i_offset, int(np.log2(p.decim)), wf_st[ch])

            def rx_ref(ch):
                return oracle.ddc_rx(adc, bank.rx_inc[ch], rx_st[ch])

            wf_out = list(pool.map(wf_ref, range(NR)))
            rx_out = list(pool.map(rx_ref, range(NR)))
            for ch in range(NR):
                p = bank.params[ch]
                # waterfall DDC: all of this step's output, then the frame the waterfall took
                iq, wf_st[ch] = wf_out[ch]
                assert iq.shape[0] == int(nw[ch]), (step, ch)
                assert np.array_equal(g["wf_iq"][ch, :iq.shape[0]], iq), (step, ch)
                assert np.array_equal(g["frames"][ch], iq[:8192])
                w_out, _, w_pwr_out, w_dB = oracle_frame(oracle, tables, iq[:8192], p, wf.WF_MAX, wf.WINF_HANNING, True, False, False)
                check_row(g["rows"][ch], w_out, w_dB, db_bound(w_pwr_out))
                want_pkt = oracle.wf_packet(g["rows"][ch], int(p.start), p.zoom, 0, True)
                assert np.array_equal(g["pkts"][ch, :want_pkt.size], want_pkt), (step, ch)

                # audio DDC -> rx_iq_t records -> unpack
                raw, rx_st[ch] = rx_out[ch]
                assert raw.size == 6 * nrec, (step, ch, raw.size, nrec)
                assert np.array_equal(g["raw"][ch, :raw.size], raw), (step, ch)
                x = oracle.dpump_unpack(raw, nrec, 1)[0]
                got_x = np.ascontiguousarray(g["xin"][ch, :nrec]).view(np.complex64).ravel()
                assert np.array_equal(got_x.view(np.uint32), x.view(np.uint32)), (step, ch)

                # CFastFIR on the GPU's own input, then CAgc and ADPCM on the GPU's own upstream output
                want_y, _ = oracle.fir_process(fir_st[ch], bank.fir.get_coef(ch), got_x, prec=0)
                assert want_y.size == nout, (step, ch, want_y.size, nout)
                if nout:
                    got_y = np.ascontiguousarray(g["firo"][ch, :nout]).view(np.complex64).ravel()
                    assert np.abs(got_y - want_y).max() <= 1e-5 * np.abs(want_y).max(), (step, ch)
                    want_s = agcs[ch].process_s16(got_y)
                    dlt = np.abs(g["s16"][ch].astype(int) - want_s.astype(int))
                    assert dlt.max() <= 1 and (dlt == 0).mean() > 0.99, (step, ch, dlt.max())
                    want_enc, ad_st[ch] = oracle.adpcm_encode_i16(g["s16"][ch], ad_st[ch])
                    assert np.array_equal(g["pay"][ch], want_enc), (step, ch)
                    audio_blocks += 1
        assert audio_blocks == 2 * NR                            # 402 records a step: a 512-sample block on steps 2 and 3
        assert bank.counts == {"frames": STEPS * NR, "audio_blocks": 2 * NR}
        pool.shutdown()
    finally:
        bank.close()
