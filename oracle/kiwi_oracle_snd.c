/*
 * kiwi_oracle_snd.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 4: audio front -- rx/data_pump.cpp snd_service() unpack and
 * rx/CuteSDR/fastfir.cpp CFastFIR.  See kiwi_oracle.h.
 * The unpack is an integer/float-exact restatement (snd_service() does not link without the SPI runtime); CFastFIR is PINNED
 * by rx/CuteSDR/fastfir.cpp itself, built in place against hipFFTW and run on the GPU box (tests/golden/fastfir_fftref.npz).
 * TYPEREAL is float and MSIN/MCOS/MPOW are sinf/cosf/powf (datatypes.h:69-82).
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define K_2PI (2.0 * 3.14159265358979323846)      /* datatypes.h:103 */
#define K_PI (3.14159265358979323846)
#define FFT_SIZE KO_FIR_FFT_SIZE
#define FIR_SIZE KO_FIR_SIZE

/* rx/data_pump.cpp:73-74 */
float ko_dpump_rescale(int use_cicf)
{
    return powf(2, -23 + 15) * (use_cicf ? powf(10, 4.5 / 20.0) : 1);
}

/* rx/data_pump.cpp:145-208.  raw: nsamps * nchans rx_iq_t {u16 i, u16 q, u8 q3, u8 i3}
 * (data_pump.h:27-30), sample-major, channel-minor.  out[ch][nsamps]. */
void ko_dpump_unpack(const uint8_t *raw, int nsamps, int nchans, const uint8_t *enabled,
                     float rescale, float dc_i, float dc_q, int spectral_inversion,
                     ko_cpx *out, int out_stride)
{
    const uint8_t *p = raw;
    for (int j = 0; j < nsamps; j++) {
        for (int ch = 0; ch < nchans; ch++, p += 6) {
            if (!enabled[ch]) continue;
            const uint16_t lo_i = (uint16_t) (p[0] | (p[1] << 8)), lo_q = (uint16_t) (p[2] | (p[3] << 8));
            const uint8_t q3 = p[4], i3 = p[5];
            /* S24_8_16(h8, l16), types.h:44 */
            const int32_t i = (int32_t) (((uint32_t) i3 << 16) | lo_i | ((i3 & 0x80) ? 0xff000000u : 0));
            const int32_t q = (int32_t) (((uint32_t) q3 << 16) | lo_q | ((q3 & 0x80) ? 0xff000000u : 0));
            ko_cpx *o = &out[(size_t) ch * out_stride + j];
            if (spectral_inversion) {                     /* :181-182 */
                o->re = i * rescale + dc_i;
                o->im = q * rescale + dc_q;
            } else {                                      /* :200-201: I/Q swapped */
                o->re = q * rescale + dc_i;
                o->im = i * rescale + dc_q;
            }
        }
    }
}

/* fastfir.cpp:61-79 (constructor): CIC compensation table */
void ko_fir_cic_coeffs(int snd_rate_3ch, float *cic)
{
    for (int i = 0; i < FFT_SIZE; i++) {
        const float f = fabs(fmod((float) i / FFT_SIZE + 0.5f, 1.0f) - 0.5f);
        const float p1 = (snd_rate_3ch ? -3.107f : -2.969f);
        const float p2 = (snd_rate_3ch ? 32.04f : 36.26f);
        const float sincf_ = f ? sinf(f * K_PI) / (f * K_PI) : 1.0f;
        cic[i] = pow(sincf_, -5) + p1 * exp(p2 * (f - 0.5f));
    }
}

/* fastfir.cpp:102-146 SetupWindowFunction; window_func < 0 -> Blackman-Nuttall */
void ko_fir_window(int window_func, float *tbl)
{
    if (window_func < 0) window_func = 0;
    for (int i = 0; i < FIR_SIZE; i++) {
        const int D = FIR_SIZE - 1;
        switch (window_func) {
        case 0: tbl[i] = (0.3635819 - 0.4891775 * cosf((K_2PI * i) / D) + 0.1365995 * cosf((2.0 * K_2PI * i) / D)
                          - 0.0106411 * cosf((3.0 * K_2PI * i) / D)); break;
        case 1: tbl[i] = (0.35875 - 0.48829 * cosf((K_2PI * i) / D) + 0.14128 * cosf((2.0 * K_2PI * i) / D)
                          - 0.01168 * cosf((3.0 * K_2PI * i) / D)); break;
        case 2: tbl[i] = (0.355768 - 0.487396 * cosf((K_2PI * i) / D) + 0.144232 * cosf((2.0 * K_2PI * i) / D)
                          - 0.012604 * cosf((3.0 * K_2PI * i) / D)); break;
        case 3: tbl[i] = (0.5 - 0.5 * cosf((K_2PI * i) / D)); break;
        default: tbl[i] = (0.54 - 0.46 * cosf((K_2PI * i) / D)); break;
        }
    }
}

/* fastfir.cpp:171-232 SetupParameters + :148-158 SetupCICFilter.
 * Returns 0, or -1 when the sanity check at :193-200 rejects the parameters
 * (the reference then leaves the previous coefficients in place).
 * time_coef (may be NULL): the 1024 zero-padded time-domain taps before the FFT. */
int ko_fir_design(float FLoCut, float FHiCut, float Offset, float SampleRate, const float *window,
                  int do_cic_comp, const float *cic_coeffs, ko_cpx *coef, ko_cpx *coef_cic,
                  ko_cpx *time_coef, int prec)
{
    FLoCut += Offset;
    FHiCut += Offset;
    if ((FLoCut >= FHiCut) || (FLoCut >= SampleRate / 2.0) || (FLoCut <= -SampleRate / 2.0) ||
        (FHiCut >= SampleRate / 2.0) || (FHiCut <= -SampleRate / 2.0))
        return -1;
    float nFL = FLoCut / SampleRate;
    float nFH = FHiCut / SampleRate;
    float nFc = (nFH - nFL) / 2.0;
    float nFs = K_2PI * (nFH + nFL) / 2.0;
    float fCenter = 0.5 * (float) (FIR_SIZE - 1);
    ko_cpx *t = (ko_cpx *) calloc(FFT_SIZE, sizeof(ko_cpx));
    for (int i = 0; i < FIR_SIZE; i++) {
        float x = (float) i - fCenter;
        float z;
        if ((float) i == fCenter) z = 2.0 * nFc;
        else z = (float) sinf(K_2PI * x * nFc) / (K_PI * x) * window[i];
        t[i].re = z * cosf(nFs * x) / (float) FFT_SIZE;
        t[i].im = z * sinf(nFs * x) / (float) FFT_SIZE;
    }
    if (time_coef) memcpy(time_coef, t, sizeof(ko_cpx) * FFT_SIZE);
    ko_fft(FFT_SIZE, -1, t, coef, prec);                              /* :229 */
    for (int i = 0; i < FFT_SIZE; i++) {                              /* :153-157 */
        coef_cic[i].re = coef[i].re * (do_cic_comp ? cic_coeffs[i] : 1.0);
        coef_cic[i].im = coef[i].im * (do_cic_comp ? cic_coeffs[i] : 1.0);
    }
    free(t);
    return 0;
}

void ko_fir_reset(ko_fir_state *s)
{
    memset(s, 0, sizeof *s);
    s->in_pos = FIR_SIZE - 1;                                         /* :58 */
}

/* fastfir.cpp:241-324 ProcessData (no extension FFT hooks).  Returns the number
 * of samples written to out (a multiple of 512). */
int ko_fir_process(ko_fir_state *s, const ko_cpx *coef_cic, const ko_cpx *in, int n, ko_cpx *out, int prec)
{
    return ko_fir_process_taps(s, coef_cic, NULL, in, n, out, prec, NULL, NULL);
}

/* The same with the extension taps of fastfir.cpp:278-302: per 1024-point block, pre = forward
 * spectrum x m_CIC (simd_multiply_cfc, :280-283: what receive_FFT(PRE_FILTERED) is handed) and
 * post = the filtered spectrum (:299-302: receive_FFT(POST_FILTERED) / specAF_FFT).  An
 * extension that edits the PRE buffer (buf_modified, :286-290) is outside this restatement. */
int ko_fir_process_taps(ko_fir_state *s, const ko_cpx *coef_cic, const float *cic, const ko_cpx *in, int n,
                        ko_cpx *out, int prec, ko_cpx *pre, ko_cpx *post)
{
    int outpos = 0, blk = 0;
    ko_cpx tmp[FFT_SIZE];
    for (int i = 0; i < n; i++) {
        int j = s->in_pos - (FFT_SIZE - FIR_SIZE + 1);
        if (j >= 0) s->overlap[j] = in[i];                            /* :265-268 */
        s->buf[s->in_pos++] = in[i];
        if (s->in_pos >= FFT_SIZE) {
            ko_fft(FFT_SIZE, -1, s->buf, tmp, prec);                  /* :274 */
            if (pre && cic)
                for (int k = 0; k < FFT_SIZE; k++) {                  /* simd_multiply_cfc :280-283 */
                    pre[(size_t) blk * FFT_SIZE + k].re = tmp[k].re * cic[k];
                    pre[(size_t) blk * FFT_SIZE + k].im = tmp[k].im * cic[k];
                }
            for (int k = 0; k < FFT_SIZE; k++) {                      /* simd_multiply_ccc :293 */
                const float ar = coef_cic[k].re, ai = coef_cic[k].im, br = tmp[k].re, bi = tmp[k].im;
                s->buf[k].re = ar * br - ai * bi;
                s->buf[k].im = ar * bi + ai * br;
            }
            if (post) memcpy(post + (size_t) blk * FFT_SIZE, s->buf, sizeof(ko_cpx) * FFT_SIZE);   /* :299-302 */
            blk++;
            ko_fft(FFT_SIZE, +1, s->buf, tmp, prec);                  /* :304 */
            if (out) for (j = FIR_SIZE - 1; j < FFT_SIZE; j++) out[outpos++] = tmp[j];   /* :307-310 */
            for (j = 0; j < FIR_SIZE - 1; j++) s->buf[j] = s->overlap[j];                /* :313-316 */
            s->in_pos = FIR_SIZE - 1;
        }
    }
    return outpos;
}
