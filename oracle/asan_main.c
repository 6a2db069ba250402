/* Sanitizer driver for the CPU oracle (test infrastructure): one pass through every family of entry
 * points on small inputs, built with -fsanitize=address,undefined (make -C oracle asan-run).  The GPU
 * box cannot run sanitizers; the oracle is the code that CAN be checked this way. */
#include "kiwi_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void)
{
    unsigned seed = 12345;
    /* PRN + acquisition, reference shape */
    static uint8_t chips[KO_L1_CODELEN], e1b[KO_E1B_CODELEN], bits[KO_NSAMPLES / 8];
    ko_cacode(2, 6, chips);
    char hex[1024];
    for (int i = 0; i < 1023; i++) hex[i] = "0123456789ABCDEF"[lcg(&seed) & 15];
    hex[1023] = 0;
    if (ko_e1b_from_hex(hex, e1b)) return 1;
    for (size_t i = 0; i < sizeof bits; i++) bits[i] = (uint8_t) lcg(&seed);
    ko_cpx *code = malloc(sizeof(ko_cpx) * KO_FFT_LEN), *data = malloc(sizeof(ko_cpx) * KO_FFT_LEN);
    ko_cpx *td = malloc(sizeof(ko_cpx) * KO_FFT_LEN);
    float ph = 0;
    ko_code_fft(chips, KO_L1_CODELEN, 0, &ph, code, 1);
    ko_sample_bits(bits, data, td, 1);
    ko_acq_cell cells[5];
    ko_acq_result r = ko_correlate(code, data, 4092, -2, 2, cells, 0);
    ko_acq_result rr[2];
    int limits[2] = {4092, 16368};
    ko_cpx *codes2 = malloc(sizeof(ko_cpx) * 2 * KO_FFT_LEN);
    memcpy(codes2, code, sizeof(ko_cpx) * KO_FFT_LEN);
    ph = 0; ko_code_fft(e1b, KO_E1B_CODELEN, 1, &ph, codes2 + KO_FFT_LEN, 0);
    ko_correlate_many(codes2, 2, data, limits, -1, 1, rr, NULL, 0, 2);
    static int16_t iq[2 * KO_NSAMPLES];
    for (size_t i = 0; i < 2 * KO_NSAMPLES; i++) iq[i] = (int16_t) (lcg(&seed) % 4001) - 2000;
    ko_sample_iq16(iq, data, NULL, 0);
    /* the 10 ms shape, one cell */
    const int N10 = 65536, NS10 = 163680;
    ko_cpx *c10 = malloc(sizeof(ko_cpx) * N10), *d10 = malloc(sizeof(ko_cpx) * N10);
    int16_t *iq10 = malloc(sizeof(int16_t) * 2 * NS10);
    for (int i = 0; i < 2 * NS10; i++) iq10[i] = (int16_t) (lcg(&seed) % 4001) - 2000;
    ph = 0; ko_code_fft_n(chips, KO_L1_CODELEN, 0, &ph, c10, 0, N10);
    ko_sample_iq16_n(iq10, d10, NULL, 0, NS10, N10);
    ko_acq_result r10 = ko_correlate_n(c10, d10, 4092, -128, -128, NULL, 1, N10);
    /* waterfall */
    static float win[KO_WF_NFFT], cic[KO_WF_NFFT], sc[1024], sc2[1024], pwr[4096], pwro[1024], db[1024];
    static uint16_t map[4096], drop[1024];
    static int16_t wiq[2 * KO_WF_NFFT];
    static uint8_t row[1024];
    ko_wf_window(KO_WINF_HANNING, win); ko_wf_cic_comp(cic);
    ko_wf_params p;
    ko_wf_params_for(3, 1.0e6f, 125.0e6, 32.0e6, 0, &p);
    ko_wf_build_maps(p.fft_used, p.plot_width, p.plot_width_clamped, 0, map, drop);
    for (int i = 0; i < 1024; i++) { sc[i] = p.fft_scale; sc2[i] = p.fft_scale / 2; }
    for (int i = 0; i < 2 * KO_WF_NFFT; i++) wiq[i] = (int16_t) (lcg(&seed) % 2001) - 1000;
    ko_cpx *samps = malloc(sizeof(ko_cpx) * KO_WF_NFFT);
    ko_wf_window_iq(wiq, win, samps);
    for (int interp = 0; interp <= KO_WF_CMA; interp++) {
        ko_wf_cfg w = {p.zoom, KO_WINF_HANNING, interp, 1, 0, p.fft_used, p.plot_width, p.plot_width_clamped,
                       map, drop, sc, sc2, p.fft_offset, cic};
        ko_wf_compute_frame(&w, samps, row, pwr, pwro, db, interp & 1);
    }
    /* audio front */
    static float fwin[KO_FIR_SIZE], fcic[KO_FIR_FFT_SIZE];
    static ko_cpx coef[1024], coefc[1024], tco[1024], fin[700], fout[2048], pre[4096], post[4096];
    ko_fir_window(-1, fwin); ko_fir_cic_coeffs(0, fcic);
    if (ko_fir_design(300.f, 2700.f, 0.f, 12000.f, fwin, 1, fcic, coef, coefc, tco, 1)) return 2;
    ko_fir_state fs; ko_fir_reset(&fs);
    for (int i = 0; i < 700; i++) { fin[i].re = (float) (lcg(&seed) % 1000); fin[i].im = (float) (lcg(&seed) % 1000); }
    int nf = ko_fir_process_taps(&fs, coefc, fcic, fin, 700, fout, 0, pre, post);
    nf += ko_fir_process(&fs, coefc, fin, 700, fout, 1);
    /* DDCs */
    int16_t *adc = malloc(sizeof(int16_t) * 70000);
    for (int i = 0; i < 70000; i++) adc[i] = (int16_t) (lcg(&seed) % 60001) - 30000;
    ko_ddc_wf_state ws; ko_ddc_wf_reset(&ws);
    int16_t *wout = malloc(4 * 70002);
    int nw = ko_ddc_wf(&ws, adc, 70000, 0x123456789abULL, 0, wout);
    ko_ddc_wf_reset(&ws); nw += ko_ddc_wf(&ws, adc, 70000, 0xfedcba98765ULL, 13, wout);
    uint8_t *rxo = malloc(6 * 32);
    int nrx = 0;
    for (int mode = 0; mode < 3; mode++) { ko_ddc_rx_state rs; ko_ddc_rx_reset(&rs); nrx += ko_ddc_rx_mode(&rs, adc, 70000, 0x3456789abcdULL, rxo, mode); }
    /* post */
    ko_agc_state *ag = malloc(ko_agc_state_size());
    ko_agc_init(ag); ko_agc_set_parameters(ag, 1, 1, -100, 50, 6, 500, 12000.f);
    static ko_cpx ao[700]; static int16_t as16[700]; static float dem[700];
    ko_agc_process_cpx(ag, 700, fin, ao); ko_agc_process_s16(ag, 700, fin, as16);
    float tap[2]; (void) ko_smeter_process(0.f, ko_smeter_alpha(12000.f), 700, fin, tap);
    double z1 = 0; ko_am_detect(&z1, 700, ao, dem);
    ko_cpx last = {0, 0}; ko_nbfm_detect(&last, 700, ao, dem);
    /* wire + hand-off */
    static uint8_t enc[350], pkt[16 + 1034], hdr[10];
    ko_adpcm_state as = {0, 0};
    ko_adpcm_encode_i16(as16, enc, 700, &as);
    ko_adpcm_decode_i16(enc, as16, 350, &as);
    int np = ko_wf_packet(row, 1, 2, 3, 1, pkt);
    ko_snd_header(0x10, 77, -50.5f, hdr);
    ko_chan_start_out cs; ko_chan_start(1, -7, 9000, 0.25, &cs);
    static float avg[1024]; int sig, noi;
    ko_aper_update(avg, row, 1, 8.f, 1, 0, 1024, -13); ko_aper_report(avg, 0, 1024, &sig, &noi);
    ko_gps_state gs = {0, 0, 0, 0}; uint32_t a, b; uint8_t c;
    ko_snd_gps_begin(&gs, 604799.9, 1.0e6, 66.6666e6, 4e-4, 1e-7);
    ko_snd_gps_stamp(&gs, 85, 168, 1, 180, 5555, 66.6666e6, 604790.0, 0, &a, &b, &c);
    printf("asan driver ok: acq %d/%d/%d, fir %d, wf ddc %d, rx %d, pkt %d, snr %.2f\n", r.idx, rr[1].idx, r10.idx, nf, nw, nrx, np, r.snr);
    free(code); free(data); free(td); free(codes2); free(c10); free(d10); free(iq10); free(samps); free(adc); free(wout); free(rxo); free(ag);
    return 0;
}
