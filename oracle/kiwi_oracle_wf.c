/*
 * kiwi_oracle_wf.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 2: waterfall (rx/rx_waterfall.cpp).  See kiwi_oracle.h.
 * c2s_waterfall_init()'s tables and compute_frame() are PINNED by rx/rx_waterfall.cpp itself, built in place against hipFFTW and
 * run on the GPU box (tests/golden/wf_fftref.npz: every output byte and ADPCM payload of eight frames equal); the zoom / start
 * parameter formulas and the map construction sit inside the c2s_waterfall() coroutine and stay a restatement.
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NFFT KO_WF_NFFT
#define WIDTH KO_WF_WIDTH

/* rx/CuteSDR/datatypes.h:103-104 */
#define K_2PI (2.0 * 3.14159265358979323846)
#define K_PI (3.14159265358979323846)

/* rx/rx_waterfall.cpp:136-171.  adc_scale_decim = 2^-16 and WINDOW_GAIN = 1.0 are
 * folded in (:136,:142,:146); window[i] is float, the cos() expression double. */
void ko_wf_window(int winf, float *window)
{
    const float adc_scale_decim = powf(2, -16);
    int i;
    for (i = 0; i < NFFT; i++) {
        window[i] = adc_scale_decim * 1.0;
        switch (winf) {
        case KO_WINF_HANNING:
            window[i] *= (0.5 - 0.5 * cos((K_2PI * i) / (float) (NFFT - 1)));
            break;
        case KO_WINF_HAMMING:
            window[i] *= (0.54 - 0.46 * cos((K_2PI * i) / (float) (NFFT - 1)));
            break;
        case KO_WINF_BLACKMAN_HARRIS:
            window[i] *= (0.35875
                - 0.48829 * cos((K_2PI * i) / (float) (NFFT - 1))
                + 0.14128 * cos((2.0 * K_2PI * i) / (float) (NFFT - 1))
                - 0.01168 * cos((3.0 * K_2PI * i) / (float) (NFFT - 1)));
            break;
        default:
            break;
        }
    }
}

/* rx/rx_waterfall.cpp:175-185; TYPEREAL is float, MSIN sinf (datatypes.h:73) */
void ko_wf_cic_comp(float *comp)
{
    int i;
    for (i = 0; i < NFFT; i++) {
        const float f = fabs(fmod((float) i / NFFT + 0.5f, 1.0f) - 0.5f);
        const float p1 = -2.969f;
        const float p2 = 36.26f;
        const float sincf_ = f ? sinf(f * K_PI) / (f * K_PI) : 1.0f;
        float cic_comp = pow(sincf_, -5) + p1 * exp(p2 * (f - 0.5f));
        comp[i] = 0.5 + cic_comp / 2.0;
    }
}

/* zoom/start -> hardware and display parameters:
 * decimation :410-426 (USE_WF_1CIC, WF_USING_HALF_CIC == 2), start clamp and NCO
 * offset :69,:485-499, fft_used/plot_width :756-773, scale/offset :889-903 */
void ko_wf_params_for(int zoom, float start, double adc_clock, double ui_srate,
                      int spectral_inversion, ko_wf_params *o)
{
    const int MAX_ZOOM = 14;
    const float HZperStart = ui_srate / (WIDTH << MAX_ZOOM);             /* :262 */
    const int zm1 = zoom ? (zoom - 1) : 0;                                /* :411 */
    o->zoom = zoom;
    o->decim = 1 << zm1;                                                  /* :415-426 */
    const int maxstart = (WIDTH << MAX_ZOOM) - (WIDTH << (MAX_ZOOM - zoom));   /* :69 */
    if (start < 0) start = 0;
    if (start > maxstart) start = maxstart;                               /* :486 */
    o->start = start;
    const float off_freq = start * HZperStart;                            /* :490 */
    const float off_freq_inv = ((float) maxstart - start) * HZperStart;   /* :491 */
    uint64_t i_offset = (uint64_t) (int64_t)
        ((spectral_inversion ? off_freq_inv : off_freq) / adc_clock * pow(2, 48));   /* :498 */
    i_offset = -i_offset;                                                 /* :499 */
    o->i_offset = i_offset & 0xffffffffffffULL;                           /* 32 + 16 bits, :507 */
    o->fft_used = NFFT / 2;                                               /* :756 */
    if (zoom != 0) o->fft_used /= 2;                                      /* :762 */
    const float span = adc_clock / 2 / (1 << zoom);                       /* :765 */
    const float disp_fs = ui_srate / (1 << zoom);                         /* :766 */
    o->plot_width = WIDTH * span / disp_fs;                               /* :772 */
    o->plot_width_clamped = (o->plot_width > WIDTH) ? WIDTH : o->plot_width;
    const float maxmag = zoom ? o->fft_used : o->fft_used / 2;            /* :891 */
    o->fft_scale = 5.0 / (maxmag * maxmag);                               /* :897 */
    o->fft_offset = zoom ? -0.08 : -0.8;                                  /* :898 */
}

/* rx/rx_waterfall.cpp:798-830 ("FFT >= plot", no unwrap) */
void ko_wf_build_maps(int fft_used, int plot_width, int plot_width_clamped,
                      int spectral_inversion, uint16_t *fft2wf_map, uint16_t *drop_sample)
{
    int i, j;
    for (i = 0; i < fft_used; i++) {
        j = plot_width * i / fft_used;                                    /* :800 */
        if (spectral_inversion) j = (j < WIDTH) ? (WIDTH - 1 - j) : -1;   /* :801-802 */
        fft2wf_map[i] = (uint16_t) j;                                     /* u2_t */
    }
    const int fft_used_inv = roundf((float) fft_used * (plot_width_clamped - 1) / plot_width);  /* :814 */
    for (i = 0; i < plot_width_clamped; i++) {
        j = roundf((float) fft_used * i / plot_width);                    /* :821 */
        if (spectral_inversion) j = fft_used_inv - j;
        drop_sample[i] = (uint16_t) j;
    }
}

/* rx/rx_waterfall.cpp:1049-1066 (the SPI chunking quirk of :1036-1041 is not
 * reproduced: SURVEY.md row W3) */
void ko_wf_window_iq(const int16_t *iq, const float *window, ko_cpx *out)
{
    int sn;
    for (sn = 0; sn < NFFT; sn++) {
        out[sn].re = (float) (int32_t) iq[2 * sn] * window[sn];
        out[sn].im = (float) (int32_t) iq[2 * sn + 1] * window[sn];
    }
}

/* rx/rx_waterfall.cpp:1275-1575, the "FFT >= plot" branch (the other is
 * unreachable in this configuration: SURVEY.md row W9). */
void ko_wf_compute_frame(const ko_wf_cfg *wf, const ko_cpx *samps, uint8_t *out,
                         float *pwr_dbg, float *pwr_out_dbg, float *dB_dbg, int prec)
{
    ko_cpx *fft = (ko_cpx *) malloc(sizeof(ko_cpx) * NFFT);
    float *pwr = (float *) calloc(NFFT, sizeof(float));
    float pwr_out[WIDTH];
    uint8_t cma_avgs[WIDTH];
    int i, fft_used_limit = wf->fft_used;                                 /* :1298 */
    ko_fft(NFFT, -1, samps, fft, prec);                                   /* :1291 */

    const int bin_dc_offset =
        (wf->zoom == 0 && wf->window_func == KO_WINF_BLACKMAN_HARRIS) ? 4 : 2;   /* :1303 */
    for (i = 0; i < bin_dc_offset; i++) pwr[i] = 0;                       /* :1304 */
    if (wf->zoom <= 1) {                                                  /* :1324 */
        for (i = bin_dc_offset; i < fft_used_limit; i++) {
            float re = fft[i].re, im = fft[i].im;
            pwr[i] = re * re + im * im;
        }
    } else {
        const int no_cic_comp = (wf->overlapped || !wf->cic_comp);        /* :1335 */
        for (i = bin_dc_offset; i < fft_used_limit; i++) {
            float re, im;
            if (no_cic_comp) {
                re = fft[i].re; im = fft[i].im;
            } else {
                float comp = wf->CIC_comp[i];
                re = fft[i].re * comp; im = fft[i].im * comp;             /* :1342 */
            }
            pwr[i] = re * re + im * im;
        }
    }

    int bin = 0, _bin = -1;
    float p, dB;
    memset(pwr_out, 0, sizeof(pwr_out));                                  /* :1385 */
    memset(cma_avgs, 0, sizeof(cma_avgs));

    if (wf->interp == KO_WF_DROP) {                                       /* :1409-1420 */
        for (i = 0; i < wf->plot_width_clamped; i++) pwr_out[i] = pwr[wf->drop_sample[i]];
    } else {
        for (i = 0; i < fft_used_limit; i++) {                            /* :1430 */
            p = pwr[i];
            bin = wf->fft2wf_map[i];
            if (bin >= WIDTH || bin < 0) {                                /* :1433 */
                fft_used_limit = i;                                       /* :1446 */
                break;
            }
            if (bin == _bin) {                                            /* :1458 */
                switch (wf->interp) {
                case KO_WF_CMA:  pwr_out[bin] += p; cma_avgs[bin]++; break;
                case KO_WF_MAX:  if (p > pwr_out[bin]) pwr_out[bin] = p; break;
                case KO_WF_MIN:  if (p < pwr_out[bin]) pwr_out[bin] = p; break;
                case KO_WF_LAST: pwr_out[bin] = p; break;
                default: break;
                }
            } else {                                                      /* :1468 */
                pwr_out[bin] = p;
                if (wf->interp == KO_WF_CMA) cma_avgs[bin] = 1;
                _bin = bin;
            }
        }
    }

    for (i = 0; i < WIDTH; i++) {                                         /* :1489 */
        float scale;
        if (wf->interp == KO_WF_CMA) {
            int avgs = cma_avgs[i];                                       /* :1499-1500 */
            scale = (avgs == 1) ? wf->fft_scale[i]
                  : ((avgs == 2) ? wf->fft_scale_div2[i] : (wf->fft_scale[i] / avgs));
        } else {
            scale = wf->fft_scale[i];
        }
        p = pwr_out[i];
        dB = 10.0 * log10f(p * scale + 1e-30F) + wf->fft_offset;          /* :1507 */
        if (dB_dbg) dB_dbg[i] = dB;
        if (dB > 0) dB = 0;                                               /* :1543 */
        if (dB < -200.0) dB = -200.0;
        dB--;
        /* :1546 (u1_t)(int)dB; NaN (an untouched CMA pixel: 0*inf) converts to
         * INT_MIN on x86 and 0 on ARM, both truncate to byte 0 */
        out[i] = (dB != dB) ? 0 : (uint8_t) (int) dB;
    }
    if (pwr_dbg) memcpy(pwr_dbg, pwr, sizeof(float) * wf->fft_used);
    if (pwr_out_dbg) memcpy(pwr_out_dbg, pwr_out, sizeof(pwr_out));
    free(fft); free(pwr);
}
