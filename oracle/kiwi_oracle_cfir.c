/*
 * kiwi_oracle_cfir.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 6b: the tail of the AM / NBFM audio chains:
 *   CFir       rx/CuteSDR/fir.cpp:74-92, 176-214 (ProcessFilter real->real, real->mono16,
 *              mono16->mono16), :220-240 (InitConstFir), :282-384 (InitLPFilter),
 *              :403-486 (InitHPFilter), :538-555 (Izero)
 *              -- m_AM_FIR (rx/rx_sound.cpp:787, designed at rx/rx_sound_cmd.cpp:270-282) and the
 *              de-emphasis filters m_nfm_deemp_FIR / m_am_ssb_deemp_FIR (rx/rx_sound.cpp:898-907,
 *              rx/rx_sound_cmd.cpp:556-585, tables rx/rx_filter.h)
 *   CSquelch   rx/CuteSDR/squelch.cpp:67-77 (Reset), :84-116 (SetupParameters),
 *              :122-129 (SetSquelch), :135-139 (InitNoiseSquelch), :151-231 (PerformFMSquelch)
 *              -- m_Squelch (rx/rx_sound.cpp:876)
 * TYPEREAL is float (datatypes.h:46) and the M*() macros are the float libm calls (:73-85); the
 * literals are double: every expression keeps the operand types of the line it cites.
 * PINNED: fir.cpp and squelch.cpp are built from their own sources (oracle/build_ref.sh ->
 * oracle/_ref/fir_ref, squelch_ref) and tests/golden/fir_ref.npz / squelch_ref.npz hold their
 * outputs; tests/test_ref_pins_cpu.py requires this restatement to reproduce them bit for bit.
 * The double -> int conversion of an infinite tap estimate (Fstop == Fpass, fir.cpp:304, :425)
 * is undefined in C; restated as what x86 does (INT_MIN), which the vectors hold.
 */
#include "kiwi_oracle.h"

#include <limits.h>
#include <math.h>
#include <string.h>

#define K_2PI (2.0 * 3.14159265358979323846)   /* datatypes.h:103 */
#define K_PI (3.14159265358979323846)          /* :104 */

size_t ko_cfir_state_size(void) { return sizeof(ko_cfir_state); }
int ko_cfir_num_taps(const ko_cfir_state *f) { return f->num_taps; }
void ko_cfir_get_taps(const ko_cfir_state *f, float *taps) { memcpy(taps, f->coef, sizeof(float) * (size_t) f->num_taps); }

void ko_cfir_init(ko_cfir_state *f)            /* fir.cpp:60-64 */
{
    memset(f, 0, sizeof *f);
    f->num_taps = 1;
    f->state = 0;
}

static void cfir_clear(ko_cfir_state *f)       /* :230-236, :344-350 */
{
    for (int i = 0; i < f->num_taps; i++) f->zbuf[i] = 0.0f;
    f->state = 0;
}

/* fir.cpp:220-240 */
void ko_cfir_init_const(ko_cfir_state *f, int num_taps, const float *coef, float fs)
{
    f->sample_rate = fs;
    f->num_taps = num_taps > KO_CFIR_MAX_NUMCOEF ? KO_CFIR_MAX_NUMCOEF : num_taps;
    for (int i = 0; i < f->num_taps; i++) {
        f->coef[i] = coef[i];
        f->coef[f->num_taps + i] = coef[i];
    }
    cfir_clear(f);
}

/* fir.cpp:538-555 */
static float izero(float x)
{
    float x2 = x / 2.0;
    float sum = 1.0;
    float ds = 1.0;
    float di = 1.0;
    float errorlimit = 1e-9;
    float tmp;
    do {
        tmp = x2 / di;
        tmp *= tmp;
        ds *= tmp;
        sum += ds;
        di += 1.0;
    } while (ds >= errorlimit * sum);
    return sum;
}

static float kaiser_beta(float Astop)          /* fir.cpp:294-301, :415-422 */
{
    float Beta;
    if (Astop < 20.96)
        Beta = 0;
    else if (Astop >= 50.0)
        Beta = .1102 * (Astop - 8.71);
    else
        Beta = .5842 * powf((Astop - 20.96), 0.4) + .07886 * (Astop - 20.96);
    return Beta;
}

static int taps_estimate(double v)             /* the (int) of :304 / :425 as x86 converts it */
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int) v;
}

static void cfir_finish(ko_cfir_state *f)      /* :332-350, :452-470 */
{
    for (int n = 0; n < f->num_taps; n++) f->coef[n + f->num_taps] = f->coef[n];
    cfir_clear(f);
}

/* fir.cpp:282-384 */
int ko_cfir_init_lp(ko_cfir_state *f, int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate)
{
    int n;
    float Beta;
    f->sample_rate = Fsamprate;
    float normFpass = Fpass / Fsamprate;
    float normFstop = Fstop / Fsamprate;
    float normFcut = (normFstop + normFpass) / 2.0;
    Beta = kaiser_beta(Astop);
    f->num_taps = taps_estimate((Astop - 8.0) / (2.285 * K_2PI * (normFstop - normFpass)) + 1);     /* :304 */
    if (f->num_taps > KO_CFIR_MAX_NUMCOEF) f->num_taps = KO_CFIR_MAX_NUMCOEF;                         /* :307-310 */
    if (f->num_taps < 9) f->num_taps = 9;
    if (NumTaps) f->num_taps = NumTaps;                                                               /* :312-313 */
    float fCenter = .5 * (float) (f->num_taps - 1);
    float izb = izero(Beta);
    for (n = 0; n < f->num_taps; n++) {
        float x = (float) n - fCenter;
        float c;
        if ((float) n == fCenter)
            c = 2.0 * normFcut;                                                                       /* :322-323 */
        else
            c = (float) sinf(K_2PI * x * normFcut) / (K_PI * x);                                      /* :325 */
        x = ((float) n - ((float) f->num_taps - 1.0) / 2.0) / (((float) f->num_taps - 1.0) / 2.0);    /* :327 */
        f->coef[n] = Scale * c * izero(Beta * sqrtf(1 - (x * x))) / izb;                              /* :328 */
    }
    cfir_finish(f);
    return f->num_taps;
}

/* fir.cpp:403-486 */
int ko_cfir_init_hp(ko_cfir_state *f, int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate)
{
    int n;
    float Beta;
    f->sample_rate = Fsamprate;
    float normFpass = Fpass / Fsamprate;
    float normFstop = Fstop / Fsamprate;
    float normFcut = (normFstop + normFpass) / 2.0;
    Beta = kaiser_beta(Astop);
    f->num_taps = taps_estimate((Astop - 8.0) / (2.285 * K_2PI * (normFpass - normFstop)) + 1);     /* :425 */
    if (f->num_taps > (KO_CFIR_MAX_NUMCOEF - 1)) f->num_taps = KO_CFIR_MAX_NUMCOEF - 1;               /* :428-431 */
    if (f->num_taps < 3) f->num_taps = 3;
    f->num_taps |= 1;                                                                                 /* :433 */
    if (NumTaps) f->num_taps = NumTaps;
    float izb = izero(Beta);
    float fCenter = .5 * (float) (f->num_taps - 1);
    for (n = 0; n < f->num_taps; n++) {
        float x = (float) n - (float) (f->num_taps - 1) / 2.0;                                        /* :442 */
        float c;
        if ((float) n == fCenter)
            c = 1.0 - 2.0 * normFcut;                                                                 /* :446 */
        else
            c = (float) (sinf(K_PI * x) / (K_PI * x) - sinf(K_2PI * x * normFcut) / (K_PI * x));      /* :448 */
        x = ((float) n - ((float) f->num_taps - 1.0) / 2.0) / (((float) f->num_taps - 1.0) / 2.0);    /* :451 */
        f->coef[n] = Scale * c * izero(Beta * sqrtf(1 - (x * x))) / izb;                              /* :452 */
    }
    cfir_finish(f);
    return f->num_taps;
}

/* one sample of fir.cpp:79-91 (the same loop in :181-193 and :204-216) */
static inline float cfir_step(ko_cfir_state *f, float in)
{
    f->zbuf[f->state] = in;
    const float *H = &f->coef[f->num_taps - f->state];
    const float *Z = f->zbuf;
    float acc = (*H++ * *Z++);
    for (int j = 1; j < f->num_taps; j++) acc += (*H++ * *Z++);
    if (--f->state < 0) f->state += f->num_taps;
    return acc;
}

/* (TYPEMONO16) of a float: as kiwi_oracle_post.c's to_mono16 (x86's conversion, low 16 bits) */
static inline int16_t cfir_mono16(float v)
{
    int32_t w;
    if (!(v > -2147483648.0f && v < 2147483648.0f)) w = (int32_t) 0x80000000u;
    else w = (int32_t) v;
    return (int16_t) (uint16_t) (uint32_t) w;
}

void ko_cfir_process_rr(ko_cfir_state *f, int n, const float *in, float *out)          /* fir.cpp:74-92 */
{
    for (int i = 0; i < n; i++) out[i] = cfir_step(f, in[i]);
}

void ko_cfir_process_rm(ko_cfir_state *f, int n, const float *in, int16_t *out)        /* fir.cpp:176-194 */
{
    for (int i = 0; i < n; i++) out[i] = cfir_mono16(cfir_step(f, in[i]));
}

void ko_cfir_process_mm(ko_cfir_state *f, int n, const int16_t *in, int16_t *out)      /* fir.cpp:199-217; in == out allowed */
{
    for (int i = 0; i < n; i++) out[i] = cfir_mono16(cfir_step(f, in[i]));
}

/* ---- CSquelch ---- */
#define VOICE_BANDWIDTH 3000.0          /* squelch.cpp:46 */
#define SQUELCH_MAX 8192                /* :57 = CLIPPER_NBFM_VAL, rx/rx_sound.h:37 */
#define SQUELCHAVE_TIMECONST .02        /* :58 */
#define SQUELCH_HYSTERESIS 50.0         /* :59 */

size_t ko_squelch_state_size(void) { return sizeof(ko_squelch_state); }

void ko_squelch_reset(ko_squelch_state *s)     /* squelch.cpp:67-77 (the PLL words are not used by PerformFMSquelch) */
{
    s->squelch_ave = 0.0;
    s->squelch_state = 1;
    s->set_squelch = 0;
}

void ko_squelch_init(ko_squelch_state *s)      /* :62-65; value / threshold are indeterminate there until SetSquelch: 0 here */
{
    memset(s, 0, sizeof *s);
    ko_cfir_init(&s->hp);
    ko_squelch_reset(s);
}

void ko_squelch_setup(ko_squelch_state *s, float samplerate)      /* :84-116 */
{
    s->sample_rate = samplerate;
    s->squelch_hp_freq = VOICE_BANDWIDTH;                                                             /* :106 */
    s->squelch_alpha = (1.0 - expf(-1.0 / (s->sample_rate * SQUELCHAVE_TIMECONST)));                  /* :107 */
    ko_cfir_init_hp(&s->hp, 0, 1.0, 50.0, s->squelch_hp_freq * .8, s->squelch_hp_freq * .65, s->sample_rate);   /* :137 */
    ko_squelch_reset(s);
}

void ko_squelch_set(ko_squelch_state *s, int Value, int SquelchMax)      /* :122-129 */
{
    s->squelch_value = Value;
    if (SquelchMax == 0) SquelchMax = SQUELCH_MAX;
    s->squelch_threshold = (float) (SquelchMax - ((SquelchMax * Value) / 99));
    s->set_squelch = 1;
}

int ko_squelch_is_squelched(const ko_squelch_state *s) { return s->squelch_state; }
float ko_squelch_ave(const ko_squelch_state *s) { return s->squelch_ave; }

/* squelch.cpp:151-231 */
int ko_squelch_perform_fm(ko_squelch_state *s, int n, const float *in, int16_t *out)
{
    int nsq_nc_sq = 0;
    if (n > KO_SQ_MAX_SQBUF_SIZE) return 0;                                                           /* :155-156 */
    float sqbuf[KO_SQ_MAX_SQBUF_SIZE];
    ko_cfir_process_rr(&s->hp, n, in, sqbuf);                                                         /* :161 */
    for (int i = 0; i < n; i++) {
        float mag = fabsf(sqbuf[i]);
        s->squelch_ave = (1.0 - s->squelch_alpha) * s->squelch_ave + s->squelch_alpha * mag;          /* :166 */
    }
    if (s->squelch_value == 0) {                                                                      /* :176-179 */
        if (s->squelch_state) nsq_nc_sq = -1;
        s->squelch_state = 0;
    } else if (s->squelch_threshold == 0) {                                                           /* :182-185 */
        if (!s->squelch_state) nsq_nc_sq = 1;
        s->squelch_state = 1;
    } else if (s->squelch_state) {                                                                    /* :188-193 */
        if (s->squelch_ave < (s->squelch_threshold - SQUELCH_HYSTERESIS)) {
            nsq_nc_sq = -1;
            s->squelch_state = 0;
        }
    } else {                                                                                          /* :195-200 */
        if (s->squelch_ave >= (s->squelch_threshold + SQUELCH_HYSTERESIS)) {
            nsq_nc_sq = 1;
            s->squelch_state = 1;
        }
    }
    if (s->squelch_state) {
        for (int i = 0; i < n; i++) out[i] = 1;                                                       /* :205-207 */
    } else {
        for (int i = 0; i < n; i++) out[i] = cfir_mono16(in[i]);                                      /* :214-215 */
    }
    if (s->set_squelch) {                                                                             /* :218-221 */
        nsq_nc_sq = s->squelch_state ? 1 : -1;
        s->set_squelch = 0;
    }
    return nsq_nc_sq;
}
