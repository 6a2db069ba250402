/*
 * kiwi_oracle_post.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 6: what consumes the CFastFIR output per receiver channel:
 *   S-meter           rx/rx_sound.cpp:248-250, 676-696
 *   CAgc              rx/CuteSDR/agc.cpp:77-86 (ctor), :98-163 (SetParameters),
 *                     :165-254 (the per-sample recursion), :259-292 (both ProcessData)
 *   AM detector       rx/rx_sound.cpp:766-783 (envelope + DC-removal IIR)
 *   NBFM detector     rx/rx_sound.cpp:845-881 (csdr fmdemod_quadri + clipper)
 * TYPEREAL is float (datatypes.h:18,46); literals such as 1.0, 0.5, 10.0, 1e-16 are
 * double, so the reference's expressions mix float and double and round to float on
 * assignment.  Every expression below keeps the operand types of the line it cites.
 * CAgc: PINNED -- rx/CuteSDR/agc.cpp is built from its own source against the kiwi.gen.h the
 * reference's assembler generates (oracle/build_ref.sh) and tests/golden/agc_ref.npz holds its
 * outputs; tests/test_ref_pins_cpu.py requires this restatement to reproduce them bit for bit.
 * S-meter and the AM / NBFM detectors live inside c2s_sound() (rx/rx_sound.cpp), which does not
 * link without the task / SPI runtime; PINNED (round 6) by the coroutine's OWN STATEMENTS -- lines 676-908, cut out of the file at
 * build time and compiled around the reference's agc.cpp / fir.cpp / squelch.cpp (oracle/build_ref.sh, oracle/ref/ref_sndpath_main.cpp;
 * tests/golden/sndpath_ref.npz, 11 scenarios over every mode family): tests/test_ref_pins_cpu.py requires this restatement to
 * reproduce sMeterAvg_dB, its taps and every out_samps_s2 sample bit for bit.  Transcendentals are this host's libm
 * (log10f, powf, expf; kiwi_oracle_libm.c restates them for the device's sake).
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <string.h>

#define DELAY_TIMECONST .015            /* agc.cpp:47 */
#define WINDOW_TIMECONST .018           /* :50 */
#define ATTACK_RISE_TIMECONST .002      /* :54 */
#define ATTACK_FALL_TIMECONST .005      /* :55 */
#define DECAY_RISEFALL_RATIO .3         /* :57 */
#define RELEASE_TIMECONST .05           /* :61 */
#define AGC_OUTSCALE 0.7                /* :64 */
#define MAX_AMPLITUDE 32767.0           /* :66 */
#define MAX_MANUAL_AMPLITUDE 32767.0    /* :67 */

size_t ko_agc_state_size(void) { return sizeof(ko_agc_state); }
int ko_agc_delay(const ko_agc_state *s) { return s->delay_samples; }     /* GetDelaySamples(), agc.h:27 */

/* agc.cpp:77-86.  The reference leaves the dynamic state (averagers, buffers) unset
 * until the first SetParameters() with a sample rate other than 100.0; the oracle
 * defines that case as "freshly reset" (same values as :119-131). */
static void agc_reset_dynamic(ko_agc_state *s)
{
    for (int i = 0; i < KO_AGC_MAX_DELAY_BUF; i++) {
        s->sig_delay_buf[i].re = 0.0f;
        s->sig_delay_buf[i].im = 0.0f;
        s->mag_buf[i] = -16.0f;
    }
    s->sig_delay_ptr = 0;
    s->hang_timer = 0;
    s->peak = -16.0f;
    s->decay_ave = -5.0f;
    s->attack_ave = -5.0f;
    s->mag_buf_pos = 0;
}

void ko_agc_init(ko_agc_state *s)
{
    memset(s, 0, sizeof *s);
    s->agc_on = 1;
    s->sample_rate = 100.0f;
    agc_reset_dynamic(s);
}

/* agc.cpp:98-163 */
void ko_agc_set_parameters(ko_agc_state *s, int agc_on, int use_hang, int threshold, int manual_gain,
                           int slope_factor, int decay, float sample_rate)
{
    if (agc_on == s->agc_on && use_hang == s->use_hang && threshold == s->threshold &&
        manual_gain == s->manual_gain && slope_factor == s->slope_factor && decay == s->decay &&
        sample_rate == s->sample_rate)
        return;                                                       /* :101-106 */
    s->agc_on = agc_on;
    s->use_hang = use_hang;
    s->threshold = threshold;
    s->manual_gain = manual_gain;
    s->slope_factor = slope_factor;                                   /* int -> TYPEREAL member, :112 */
    s->decay = decay;
    if (s->sample_rate != sample_rate) {                              /* :115-131 */
        s->sample_rate = sample_rate;
        agc_reset_dynamic(s);
    }
    s->manual_agc_gain = MAX_MANUAL_AMPLITUDE * powf(10.0, -(100 - (float) s->manual_gain) / 20.0);   /* :134 */
    s->knee = (float) s->threshold / 20.0;                            /* :138 */
    s->gain_slope = s->slope_factor / 100.0;                          /* :139 */
    s->fixed_gain = AGC_OUTSCALE * powf(10.0, s->knee * (s->gain_slope - 1.0));                       /* :140 */
    s->attack_rise_alpha = (1.0 - expf(-1.0 / (s->sample_rate * ATTACK_RISE_TIMECONST)));             /* :144 */
    s->attack_fall_alpha = (1.0 - expf(-1.0 / (s->sample_rate * ATTACK_FALL_TIMECONST)));             /* :145 */
    s->decay_rise_alpha =
        (1.0 - expf(-1.0 / (s->sample_rate * (float) s->decay * .001 * DECAY_RISEFALL_RATIO)));      /* :147 */
    s->hang_time = (int) (s->sample_rate * (float) s->decay * .001);                                  /* :148 */
    if (s->use_hang)
        s->decay_fall_alpha = (1.0 - expf(-1.0 / (s->sample_rate * RELEASE_TIMECONST)));              /* :151 */
    else
        s->decay_fall_alpha = (1.0 - expf(-1.0 / (s->sample_rate * (float) s->decay * .001)));        /* :153 */
    s->delay_samples = (int) (s->sample_rate * DELAY_TIMECONST);      /* :155 */
    s->window_samples = (int) (s->sample_rate * WINDOW_TIMECONST);    /* :156 */
    if (s->delay_samples >= KO_AGC_MAX_DELAY_BUF - 1) s->delay_samples = KO_AGC_MAX_DELAY_BUF - 1;    /* :159-160 */
}

/* agc.cpp:172-252: one input sample -> the delayed sample and the gain to apply to it */
static inline float agc_step(ko_agc_state *s, ko_cpx in, ko_cpx *delayed)
{
    *delayed = s->sig_delay_buf[s->sig_delay_ptr];                    /* :175 */
    s->sig_delay_buf[s->sig_delay_ptr++] = in;                        /* :178 */
    if (s->sig_delay_ptr >= s->delay_samples) s->sig_delay_ptr = 0;   /* :179-180 */

    float mag = in.re * in.re + in.im * in.im;                        /* :189 */
    mag = 0.5 * log10f(mag / (MAX_AMPLITUDE * MAX_AMPLITUDE) + 1e-16);   /* :191 */

    float tmp = s->mag_buf[s->mag_buf_pos];                           /* :194 oldest */
    s->mag_buf[s->mag_buf_pos++] = mag;                               /* :195 */
    if (s->mag_buf_pos >= s->window_samples) s->mag_buf_pos = 0;      /* :196-197 */
    if (mag > s->peak) {                                              /* :199-200 */
        s->peak = mag;
    } else if (tmp == s->peak) {                                      /* :202-210: the peak left the window */
        s->peak = -8.0f;
        for (int k = 0; k < s->window_samples; k++)
            if (s->mag_buf[k] > s->peak) s->peak = s->mag_buf[k];
    }

    const float peak = s->peak;
    if (peak > s->attack_ave)                                         /* :215-218 and :232-235, same in both modes */
        s->attack_ave = (1.0 - s->attack_rise_alpha) * s->attack_ave + s->attack_rise_alpha * peak;
    else
        s->attack_ave = (1.0 - s->attack_fall_alpha) * s->attack_ave + s->attack_fall_alpha * peak;
    if (s->use_hang) {                                                /* :220-229 */
        if (peak > s->decay_ave) {
            s->decay_ave = (1.0 - s->decay_rise_alpha) * s->decay_ave + s->decay_rise_alpha * peak;
            s->hang_timer = 0;
        } else if (s->hang_timer < s->hang_time) {
            s->hang_timer++;
        } else {
            s->decay_ave = (1.0 - s->decay_fall_alpha) * s->decay_ave + s->decay_fall_alpha * peak;
        }
    } else {                                                          /* :237-240 */
        if (peak > s->decay_ave)
            s->decay_ave = (1.0 - s->decay_rise_alpha) * s->decay_ave + s->decay_rise_alpha * (peak);
        else
            s->decay_ave = (1.0 - s->decay_fall_alpha) * s->decay_ave + s->decay_fall_alpha * (peak);
    }
    mag = s->attack_ave > s->decay_ave ? s->attack_ave : s->decay_ave;   /* :244-247 */
    if (mag <= s->knee) return s->fixed_gain;                         /* :250-251 */
    return AGC_OUTSCALE * powf(10.0, mag * (s->gain_slope - 1.0));    /* :253 */
}

/* agc.cpp:259-271 */
void ko_agc_process_cpx(ko_agc_state *s, int n, const ko_cpx *in, ko_cpx *out)
{
    if (s->agc_on) {
        for (int i = 0; i < n; i++) {
            ko_cpx d;
            const float gain = agc_step(s, in[i], &d);
            out[i].re = d.re * gain;
            out[i].im = d.im * gain;
        }
    } else {
        for (int i = 0; i < n; i++) {
            out[i].re = s->manual_agc_gain * in[i].re;
            out[i].im = s->manual_agc_gain * in[i].im;
        }
    }
}

/* (TYPEMONO16) of a float: truncation toward zero; out of the s2_t range the
 * reference's cast is undefined -- the oracle (and the GPU path) fix it as the x86
 * behaviour of int conversion followed by taking the low 16 bits. */
static inline int16_t to_mono16(float v)
{
    int32_t w;
    if (!(v > -2147483648.0f && v < 2147483648.0f)) w = (int32_t) 0x80000000u;   /* cvttss2si indefinite */
    else w = (int32_t) v;
    return (int16_t) (uint16_t) (uint32_t) w;
}

/* agc.cpp:281-292 */
void ko_agc_process_s16(ko_agc_state *s, int n, const ko_cpx *in, int16_t *out)
{
    if (s->agc_on) {
        for (int i = 0; i < n; i++) {
            ko_cpx d;
            const float gain = agc_step(s, in[i], &d);
            out[i] = to_mono16(d.re * gain);
        }
    } else {
        for (int i = 0; i < n; i++) out[i] = to_mono16(s->manual_agc_gain * in[i].re);
    }
}

/* rx_sound.cpp:248-249 */
float ko_smeter_alpha(float frate)
{
    return 1.0 - expf(-1.0 / ((float) frate * .01));
}

/* rx_sound.cpp:676-696.  Returns the new average; tap[0], tap[1] = the values at j == 0
 * and j == n/2 (what receive_S_meter() would be handed, before S_meter_cal). */
float ko_smeter_process(float avg_dB, float alpha, int n, const ko_cpx *in, float *tap)
{
    const float snd_max_val = (float) ((1 << (15 - 2)) - 1);          /* :683, CUTESDR_SCALE 15 (kiwi.h:42) */
    const float snd_max_pwr = snd_max_val * snd_max_val;              /* :684 */
    for (int j = 0; j < n; j++) {
        const float re = (float) in[j].re, im = (float) in[j].im;
        const float pwr = re * re + im * im;
        const float pwr_dB = 10.0 * log10f((pwr / snd_max_pwr) + 1e-30);     /* :687 */
        avg_dB = (1.0 - alpha) * avg_dB + alpha * pwr_dB;                     /* :688 */
        if (tap && (j == 0 || j == n / 2)) tap[j == 0 ? 0 : 1] = avg_dB;      /* :693 */
    }
    return avg_dB;
}

/* rx_sound.cpp:766-783; z1 is the `double z1` of :244 */
void ko_am_detect(double *z1, int n, const ko_cpx *agc, float *demod)
{
    for (int j = 0; j < n; j++) {
        const float pwr = agc[j].re * agc[j].re + agc[j].im * agc[j].im;
        const float mag = sqrtf(pwr);                                 /* :771 (C++ sqrt(float)) */
        const float z0 = mag + (*z1 * 0.99f);                         /* :777 */
        demod[j] = z0 - *z1;                                          /* :778 */
        *z1 = z0;                                                     /* :779 */
    }
}

/* rx_sound.cpp:845-881; last = conn->last_sample */
void ko_nbfm_detect(ko_cpx *last, int n, const ko_cpx *agc, float *demod)
{
    const float max_val = 32767, clipper_val = 8192;                  /* :839, rx_sound.h:37 */
    for (int j = 0; j < n; j++) {
        const float i = agc[j].re, q = agc[j].im;
        const float iL = j ? agc[j - 1].re : last->re, qL = j ? agc[j - 1].im : last->im;
        const float pwr = i * i + q * q;
        float out = pwr ? (max_val * 0.340447550238101026565118445432744920253753662109375 *
                           (i * (q - qL) - q * (i - iL)) / pwr) : 0;   /* :851,859 */
        if (clipper_val > 0) out = out < -clipper_val ? -clipper_val : (out > clipper_val ? clipper_val : out);
        demod[j] = out;
    }
    if (n > 0) *last = agc[n - 1];                                    /* :875 */
}
