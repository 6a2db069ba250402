/*
 * kiwi_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See kiwi_oracle.h for the scope statement and the pinning status.
 * Part 1: FFT, PRN generators, GPS acquisition (gps/search.cpp).
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* FFT                                                                       */
/* ======================================================================== */

typedef struct { double re, im; } cpxd;

#define TW_CACHE 16
static struct { int n, sign; cpxd *d; ko_cpx *f; } tw_cache[TW_CACHE];
static pthread_mutex_t tw_lock = PTHREAD_MUTEX_INITIALIZER;

/* exp(sign*2*pi*i*k/n) for k in [0,n), every entry straight from cos/sin in
 * double (no recurrences), plus its fp32 rounding. */
static void twiddles(int n, int sign, const cpxd **d, const ko_cpx **f)
{
    pthread_mutex_lock(&tw_lock);
    int i, slot = -1;
    for (i = 0; i < TW_CACHE; i++) {
        if (tw_cache[i].n == n && tw_cache[i].sign == sign) { slot = i; break; }
        if (tw_cache[i].n == 0 && slot < 0) slot = i;
    }
    if (slot < 0) slot = 0;     /* evict; callers hold pointers only briefly */
    if (tw_cache[slot].n != n || tw_cache[slot].sign != sign) {
        free(tw_cache[slot].d); free(tw_cache[slot].f);
        cpxd *td = (cpxd *) malloc(sizeof(cpxd) * n);
        ko_cpx *tf = (ko_cpx *) malloc(sizeof(ko_cpx) * n);
        for (i = 0; i < n; i++) {
            double a = (double) sign * 2.0 * M_PI * (double) i / (double) n;
            td[i].re = cos(a); td[i].im = sin(a);
            tf[i].re = (float) td[i].re; tf[i].im = (float) td[i].im;
        }
        tw_cache[slot].n = n; tw_cache[slot].sign = sign;
        tw_cache[slot].d = td; tw_cache[slot].f = tf;
    }
    *d = tw_cache[slot].d; *f = tw_cache[slot].f;
    pthread_mutex_unlock(&tw_lock);
}

static int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

/* double-precision radix-2 decimation-in-time, natural in / natural out */
static void fft_f64(int n, int sign, const ko_cpx *in, ko_cpx *out)
{
    const cpxd *tw; const ko_cpx *twf;
    twiddles(n, sign, &tw, &twf);
    int lg = ilog2(n), i, j, s;
    cpxd *x = (cpxd *) malloc(sizeof(cpxd) * n);
    for (i = 0; i < n; i++) {
        unsigned r = 0, v = (unsigned) i;
        for (j = 0; j < lg; j++) { r = (r << 1) | (v & 1); v >>= 1; }
        x[r].re = in[i].re; x[r].im = in[i].im;
    }
    for (s = 1; s <= lg; s++) {
        int m = 1 << s, h = m >> 1, step = n / m;
        for (i = 0; i < n; i += m) {
            for (j = 0; j < h; j++) {
                cpxd w = tw[j * step];
                cpxd a = x[i + j], b = x[i + j + h];
                double tr = b.re * w.re - b.im * w.im;
                double ti = b.re * w.im + b.im * w.re;
                x[i + j].re = a.re + tr;     x[i + j].im = a.im + ti;
                x[i + j + h].re = a.re - tr; x[i + j + h].im = a.im - ti;
            }
        }
    }
    for (i = 0; i < n; i++) { out[i].re = (float) x[i].re; out[i].im = (float) x[i].im; }
    free(x);
}

/* fp32 Stockham autosort, radix-4 passes plus one radix-2 pass when log2(n)
 * is odd.  All arithmetic in fp32, twiddles = fp32 roundings of exact values. */
static void fft_f32(int n, int sign, const ko_cpx *in, ko_cpx *out)
{
    const cpxd *twd; const ko_cpx *tw;
    twiddles(n, sign, &twd, &tw);
    ko_cpx *a = (ko_cpx *) malloc(sizeof(ko_cpx) * n);
    ko_cpx *b = (ko_cpx *) malloc(sizeof(ko_cpx) * n);
    memcpy(a, in, sizeof(ko_cpx) * n);
    const float sg = (float) sign;
    int ns = 1;
    if (ilog2(n) & 1) {                       /* radix-2 first, ns = 1: no twiddles */
        int h = n / 2, t;
        for (t = 0; t < h; t++) {
            ko_cpx u = a[t], v = a[t + h];
            b[2 * t].re = u.re + v.re;     b[2 * t].im = u.im + v.im;
            b[2 * t + 1].re = u.re - v.re; b[2 * t + 1].im = u.im - v.im;
        }
        { ko_cpx *tmp = a; a = b; b = tmp; }
        ns = 2;
    }
    for (; ns < n; ns *= 4) {
        int q = n / 4, stride = n / (ns * 4), t;
        for (t = 0; t < q; t++) {
            int k = t % ns;
            ko_cpx x0 = a[t], x1 = a[t + q], x2 = a[t + 2 * q], x3 = a[t + 3 * q];
            if (k) {
                ko_cpx w1 = tw[k * stride], w2 = tw[2 * k * stride], w3 = tw[3 * k * stride];
                ko_cpx y;
                y.re = x1.re * w1.re - x1.im * w1.im; y.im = x1.re * w1.im + x1.im * w1.re; x1 = y;
                y.re = x2.re * w2.re - x2.im * w2.im; y.im = x2.re * w2.im + x2.im * w2.re; x2 = y;
                y.re = x3.re * w3.re - x3.im * w3.im; y.im = x3.re * w3.im + x3.im * w3.re; x3 = y;
            }
            ko_cpx s02 = { x0.re + x2.re, x0.im + x2.im }, d02 = { x0.re - x2.re, x0.im - x2.im };
            ko_cpx s13 = { x1.re + x3.re, x1.im + x3.im }, d13 = { x1.re - x3.re, x1.im - x3.im };
            /* sign*j*d13 */
            ko_cpx jd = { -sg * d13.im, sg * d13.re };
            int o = (t / ns) * ns * 4 + k;
            b[o].re = s02.re + s13.re;          b[o].im = s02.im + s13.im;
            b[o + ns].re = d02.re + jd.re;      b[o + ns].im = d02.im + jd.im;
            b[o + 2 * ns].re = s02.re - s13.re; b[o + 2 * ns].im = s02.im - s13.im;
            b[o + 3 * ns].re = d02.re - jd.re;  b[o + 3 * ns].im = d02.im - jd.im;
        }
        { ko_cpx *tmp = a; a = b; b = tmp; }
    }
    memcpy(out, a, sizeof(ko_cpx) * n);
    free(a); free(b);
}

/* prec = 2: the transform is done by whoever registered a hook (tests/test_fft_invariance_cpu.py plugs the tuned
 * FFT that IS in the image, scipy.fft / pocketfft complex64, into every reference loop restated here -- the place
 * FFTW3f has in the reference).  The hook may be entered from ko_correlate_many's worker threads. */
static ko_fft_hook g_fft_hook;
void ko_set_fft_hook(ko_fft_hook fn) { g_fft_hook = fn; }

void ko_fft(int n, int sign, const ko_cpx *in, ko_cpx *out, int prec)
{
    if (prec == 2 && g_fft_hook) g_fft_hook(n, sign, in, out);
    else if (prec) fft_f64(n, sign, in, out);
    else fft_f32(n, sign, in, out);
}

/* ======================================================================== */
/* PRN generators                                                            */
/* ======================================================================== */

/* gps/cacode.h:23-53.  G1 = x^10+x^3+1, G2 = x^10+x^9+x^8+x^6+x^3+x^2+1,
 * registers indexed 1..10, shift towards higher index (:47-52). */
void ko_cacode(int t0, int t1, uint8_t *chips)
{
    uint8_t g1[11], g2[11];
    int i, n, use_taps = !(t0 > 10 || t1 > 10);       /* cacode.h:28 */
    if (use_taps) {
        for (i = 1; i <= 10; i++) g2[i] = 1;           /* cacode.h:38 */
    } else {
        int v = t1;
        for (i = 1; i <= 10; i++, v >>= 1) g2[i] = (uint8_t) (v & 1);   /* cacode.h:32-33 */
    }
    for (i = 1; i <= 10; i++) g1[i] = 1;               /* cacode.h:40 */
    for (n = 0; n < KO_L1_CODELEN; n++) {
        chips[n] = use_taps ? (uint8_t) (g1[10] ^ g2[t0] ^ g2[t1])     /* cacode.h:44 */
                            : (uint8_t) (g1[10] ^ g2[10]);
        g1[0] = g1[3] ^ g1[10];                                         /* cacode.h:48 */
        g2[0] = g2[2] ^ g2[3] ^ g2[6] ^ g2[8] ^ g2[9] ^ g2[10];        /* cacode.h:49 */
        for (i = 10; i >= 1; i--) { g1[i] = g1[i - 1]; g2[i] = g2[i - 1]; }
    }
}

/* gps/e1bcode.h:70-76 */
int ko_e1b_from_hex(const char *hex, uint8_t *chips)
{
    int i;
    for (i = 0; i < KO_E1B_CODELEN; i++) {
        char c = hex[i / 4];
        int nib;
        if (c >= '0' && c <= '9') nib = c - '0';
        else if (c >= 'A' && c <= 'F') nib = c - 'A' + 10;
        else return -1;
        chips[i] = (uint8_t) ((nib >> (3 - (i % 4))) & 1);
    }
    return 0;
}

/* ======================================================================== */
/* GPS acquisition                                                           */
/* ======================================================================== */

/* gps/search.cpp:62-66 */
static inline float bipolar(int bit) { return bit ? -1.0f : 1.0f; }

/* gps/search.cpp:101-136, column FT=0 ("remez"); decimal literals rounded to
 * float exactly as the reference's float COEF[][] initialiser does. */
static const float HB_COEF[KO_NTAPS] = {
    -0.010233f, 0.0f,  0.010668f, 0.0f, -0.016324f, 0.0f,  0.024377f, 0.0f,
    -0.036482f, 0.0f,  0.056990f, 0.0f, -0.101993f, 0.0f,
     0.316926f, 0.500009f, 0.316926f,
     0.0f, -0.101993f, 0.0f,  0.056990f, 0.0f, -0.036482f, 0.0f,  0.024377f,
     0.0f, -0.016324f, 0.0f,  0.010668f, 0.0f, -0.010233f,
};

/* gps/search.cpp:140-166.  Accumulation order kept: c0 term, then j = 2,4..30,
 * then the centre tap; separate multiply and add (built with -ffp-contract=off). */
int ko_decimate_by2_float(int size, ko_cpx *buf)
{
    const float coef_0 = HB_COEF[0], coef_m = HB_COEF[(KO_NTAPS - 1) / 2];
    int i, o, j;
    memset(&buf[size], 0, KO_NTAPS * sizeof(ko_cpx));             /* :145 */
    for (i = 0, o = 0; i < size; i += 2, ++o) {
        float accI = buf[i].re * coef_0;
        float accQ = buf[i].im * coef_0;
        for (j = 2; j < KO_NTAPS; j += 2) {
            const float coef = HB_COEF[j];
            accI += buf[i + j].re * coef;
            accQ += buf[i + j].im * coef;
        }
        accI += buf[i + (KO_NTAPS - 1) / 2].re * coef_m;
        accQ += buf[i + (KO_NTAPS - 1) / 2].im * coef_m;
        buf[o].re = accI;
        buf[o].im = accQ;
    }
    return size / 2;
}

/* gps/search.cpp:250-267 (boc=0) and :315-329 (boc=1), then :269-276 */
void ko_code_replica_n(const uint8_t *chips, int nchips, int boc, float *phase,
                       ko_cpx *out, int fft_len)
{
    /* the replica covers NSAMPLES = DECIM * FFT_LEN samples (:250; gps/gps.h:72-73) */
    const int nsamples = KO_DECIM * fft_len;
    ko_cpx *buf = (ko_cpx *) malloc(sizeof(ko_cpx) * ((size_t) nsamples + 2 * KO_NTAPS));
    const float rate = (float) (1.023e6 / 16.368e6);     /* CPS/FS, :205,:306 */
    float ph = *phase;
    int codep = 0, i, n;
    for (i = 0; i < nsamples; i++) {
        float chip;
        if (!boc) {
            chip = bipolar(chips[codep]);                 /* :252 */
            ph += rate;                                   /* :254 */
            if (ph >= 1.0) {                              /* :256 */
                ph -= 1.0;
                if (++codep >= nchips) codep = 0;         /* ca.Clock(), period 1023 */
                chip *= 1.0 - ph;                         /* :261 */
                chip += ph * bipolar(chips[codep]);       /* :262 */
            }
        } else {
            int boc11 = (ph >= 0.5) ? 1 : 0;              /* :317 */
            chip = bipolar(chips[codep] ^ boc11);         /* :318 */
            ph += rate;                                   /* :320 */
            if (ph >= 1.0) {                              /* :322 */
                ph -= 1.0;
                if (++codep >= nchips) codep = 0;         /* e1bcode.h:86-90 */
            }
        }
        buf[i].re = chip; buf[i].im = 0;
    }
    *phase = ph;
    n = nsamples;
    for (i = KO_DECIM; i > 1; i >>= 1) n = ko_decimate_by2_float(n, buf);   /* :273-275 */
    memcpy(out, buf, sizeof(ko_cpx) * fft_len);
    free(buf);
}

void ko_code_replica(const uint8_t *chips, int nchips, int boc, float *phase, ko_cpx *out)
{
    ko_code_replica_n(chips, nchips, boc, phase, out, KO_FFT_LEN);
}

void ko_code_fft_n(const uint8_t *chips, int nchips, int boc, float *phase,
                   ko_cpx *out, int prec, int fft_len)
{
    ko_cpx *td = (ko_cpx *) malloc(sizeof(ko_cpx) * fft_len);
    ko_code_replica_n(chips, nchips, boc, phase, td, fft_len);
    ko_fft(fft_len, -1, td, out, prec);                   /* :280 / :342 */
    free(td);
}

void ko_code_fft(const uint8_t *chips, int nchips, int boc, float *phase,
                 ko_cpx *out, int prec)
{
    ko_code_fft_n(chips, nchips, boc, phase, out, prec, KO_FFT_LEN);
}

/* buf holds DECIM * fft_len samples, zero from `nsamples` on (the reference's block is full:
 * nsamples == DECIM * fft_len) */
static void finish_sample(ko_cpx *buf, ko_cpx *out, ko_cpx *td, int prec, int fft_len)
{
    int n = KO_DECIM * fft_len, i;
    /* DecimateBy2binary's float stage + DecimateBy2float, :437-442 */
    for (i = KO_DECIM; i > 1; i >>= 1) n = ko_decimate_by2_float(n, buf);
    if (td) memcpy(td, buf, sizeof(ko_cpx) * fft_len);
    ko_fft(fft_len, -1, buf, out, prec);                  /* :447 */
}

/* gps/search.cpp:382-449 */
void ko_sample_bits_n(const uint8_t *packed, ko_cpx *out, ko_cpx *td, int prec, int nsamples, int fft_len)
{
    static const int lo_sin[4] = {1, 1, 0, 0};            /* :383 */
    static const int lo_cos[4] = {1, 0, 0, 1};            /* :384 */
    const float lo_rate = (float) (4 * 4.092e6 / 16.368e6);     /* :386 */
    ko_cpx *buf = (ko_cpx *) calloc((size_t) KO_DECIM * fft_len + 2 * KO_NTAPS, sizeof(ko_cpx));
    float lo_phase = 0;
    int i = 0, j = 0, b;
    while (i < nsamples) {
        uint8_t byte = packed[j++];                       /* :408 */
        for (b = 0; b < 8 && i < nsamples; ++b, ++i, byte >>= 1) {
            const int bit = byte & 1;                     /* :411 LSB first */
            int bi = bit ^ lo_sin[(int) lo_phase];        /* :419 */
            int bq = bit ^ lo_cos[(int) lo_phase];        /* :420 */
            lo_phase += lo_rate;                          /* :422 */
            lo_phase -= 4 * (lo_phase >= 4);              /* :423 */
            /* simd_bit2float (support/simd.cpp:165-167) then Bipolar(f >= 0)
             * (:173-174): bit 1 -> -1.0, bit 0 -> +1.0 */
            float fi = (float) (2 * (bi > 0) - 1), fq = (float) (2 * (bq > 0) - 1);
            buf[i].re = bipolar(fi >= 0);
            buf[i].im = bipolar(fq >= 0);
        }
    }
    finish_sample(buf, out, td, prec, fft_len);
    free(buf);
}

void ko_sample_bits(const uint8_t *packed, ko_cpx *out, ko_cpx *td, int prec)
{
    ko_sample_bits_n(packed, out, td, prec, KO_NSAMPLES, KO_FFT_LEN);
}

void ko_sample_iq16_n(const int16_t *iq, ko_cpx *out, ko_cpx *td, int prec, int nsamples, int fft_len)
{
    ko_cpx *buf = (ko_cpx *) calloc((size_t) KO_DECIM * fft_len + 2 * KO_NTAPS, sizeof(ko_cpx));
    int i;
    for (i = 0; i < nsamples; i++) {
        float a = (float) iq[2 * i], b = (float) iq[2 * i + 1];
        switch (i & 3) {                                  /* (a+jb)*(-j)^i */
        case 0:  buf[i].re =  a; buf[i].im =  b; break;
        case 1:  buf[i].re =  b; buf[i].im = -a; break;
        case 2:  buf[i].re = -a; buf[i].im = -b; break;
        default: buf[i].re = -b; buf[i].im =  a; break;
        }
    }
    finish_sample(buf, out, td, prec, fft_len);
    free(buf);
}

void ko_sample_iq16(const int16_t *iq, ko_cpx *out, ko_cpx *td, int prec)
{
    ko_sample_iq16_n(iq, out, td, prec, KO_NSAMPLES, KO_FFT_LEN);
}

/* one (SV, Doppler) cell of the search.cpp:465-496 loop body.
 * The reference multiplies by code[sat] + FFT_LEN - dop (:471), a pointer into a row that holds the spectrum TWICE (:54,
 * :281-282): entry N - dop + i.  For dop >= 0 that is bin (i - dop) mod N.  For dop < 0 the last |dop| products run past the
 * row's 2 N entries into the NEXT satellite's row, code[sat + 1][0 .. |dop|) -- the rows of the static array are contiguous,
 * a row nobody wrote is zero -- and NOT into bins 0 .. |dop| - 1 of the satellite's own spectrum.  `next` is that row (NULL:
 * never written).  Pinned by the reference's own Correlate() (oracle/_ref/search_ref, tests/golden/acq_fftref.npz): until
 * round 6 this function wrapped modulo N, 0.3 % off in snr at dop = -13 and another winner among noise peaks. */
static ko_acq_cell correlate_cell(const ko_cpx *code, const ko_cpx *next, const ko_cpx *data, int limit, int dop,
                                  ko_cpx *prod, ko_cpx *rev, int prec, int N)
{
    float max_pwr = 0, tot_pwr = 0;                       /* :466 */
    int max_pwr_i = 0, i;
    for (i = 0; i < N; i++) {                             /* :471 = :473-477, simd.cpp:39-67 */
        const long e = (long) N - dop + i;                /* entry of the doubled row; dop in [-N, N] */
        ko_cpx c;
        if (e < 2L * N) c = code[e % N];
        else if (next) c = next[e - 2L * N];
        else { c.re = 0.0f; c.im = 0.0f; }
        prod[i].re = data[i].re * c.re + data[i].im * c.im;
        prod[i].im = data[i].re * c.im - data[i].im * c.re;
    }
    ko_fft(N, +1, prod, rev, prec);                       /* :481 */
    for (i = 0; i < limit; i++) {                         /* :486-490 */
        const float pwr = rev[i].re * rev[i].re + rev[i].im * rev[i].im;
        if (pwr > max_pwr) { max_pwr = pwr; max_pwr_i = i; }
        tot_pwr += pwr;
    }
    ko_acq_cell c;
    const float ave_pwr = tot_pwr / i;                    /* :493 */
    c.snr = max_pwr / ave_pwr;                            /* :494 */ c.max_pwr = max_pwr; c.tot_pwr = tot_pwr; c.idx = max_pwr_i;
    return c;
}

/* gps/search.cpp:453-499 */
ko_acq_result ko_correlate_next_n(const ko_cpx *code, const ko_cpx *next, const ko_cpx *data, int limit,
                                  int dop_lo, int dop_hi, ko_acq_cell *cells, int prec, int fft_len)
{
    ko_cpx *prod = (ko_cpx *) malloc(sizeof(ko_cpx) * fft_len);       /* rev_buf, :58,:454 */
    ko_cpx *rev = (ko_cpx *) malloc(sizeof(ko_cpx) * fft_len);
    ko_acq_result r = {0.0f, 0, 0, 0};
    float max_snr = 0;                                    /* :455 */
    int dop;
    for (dop = dop_lo; dop <= dop_hi; dop++) {            /* :465 */
        const ko_acq_cell c = correlate_cell(code, next, data, limit, dop, prod, rev, prec, fft_len);
        if (cells) cells[dop - dop_lo] = c;
        if (c.snr > max_snr) {                            /* :495 */
            max_snr = c.snr; r.dop = dop; r.idx = c.idx; r.valid = 1;
        }
    }
    r.snr = max_snr;                                      /* :498 */
    free(prod); free(rev);
    return r;
}

/* the next row never written (zeros) */
ko_acq_result ko_correlate_n(const ko_cpx *code, const ko_cpx *data, int limit,
                             int dop_lo, int dop_hi, ko_acq_cell *cells, int prec, int fft_len)
{
    return ko_correlate_next_n(code, NULL, data, limit, dop_lo, dop_hi, cells, prec, fft_len);
}

ko_acq_result ko_correlate(const ko_cpx *code, const ko_cpx *data, int limit,
                           int dop_lo, int dop_hi, ko_acq_cell *cells, int prec)
{
    return ko_correlate_n(code, data, limit, dop_lo, dop_hi, cells, prec, KO_FFT_LEN);
}

typedef struct {
    const ko_cpx *codes, *data; const int *limits; int nsv, dop_lo, dop_hi, prec;
    ko_acq_cell *cells; int tid, nthreads, fft_len;
    const ko_cpx *nexts; const unsigned char *has_next;   /* [nsv][fft_len], [nsv]: the row behind each SV's (see correlate_cell) */
} many_arg;

/* threads take (SV, Doppler) cells round-robin: all host cores stay busy even
 * when there are fewer SVs than cores (the CPU-baseline "T_all" figure) */
static void *many_worker(void *p)
{
    many_arg *a = (many_arg *) p;
    const int nd = a->dop_hi - a->dop_lo + 1, ncell = a->nsv * nd;
    ko_cpx *prod = (ko_cpx *) malloc(sizeof(ko_cpx) * a->fft_len);
    ko_cpx *rev = (ko_cpx *) malloc(sizeof(ko_cpx) * a->fft_len);
    int c;
    for (c = a->tid; c < ncell; c += a->nthreads) {
        const int s = c / nd, di = c - s * nd;
        const ko_cpx *next = (a->nexts && a->has_next && a->has_next[s]) ? a->nexts + (size_t) s * a->fft_len : NULL;
        a->cells[c] = correlate_cell(a->codes + (size_t) s * a->fft_len, next, a->data, a->limits[s],
                                     a->dop_lo + di, prod, rev, a->prec, a->fft_len);
    }
    free(prod); free(rev);
    return NULL;
}

void ko_correlate_many_next_n(const ko_cpx *codes, const ko_cpx *nexts, const unsigned char *has_next, int nsv, const ko_cpx *data,
                              const int *limits, int dop_lo, int dop_hi,
                              ko_acq_result *out, ko_acq_cell *cells, int prec,
                              int nthreads, int fft_len)
{
    const int nd = dop_hi - dop_lo + 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    if (nthreads > nsv * nd) nthreads = nsv * nd;
    { const cpxd *d; const ko_cpx *f; twiddles(fft_len, +1, &d, &f); }     /* warm cache */
    ko_acq_cell *cl = cells ? cells : (ko_acq_cell *) malloc(sizeof(ko_acq_cell) * (size_t) nsv * nd);
    pthread_t *th = (pthread_t *) malloc(sizeof(pthread_t) * nthreads);
    many_arg *args = (many_arg *) malloc(sizeof(many_arg) * nthreads);
    int t, s, di;
    for (t = 0; t < nthreads; t++) {
        many_arg a = { codes, data, limits, nsv, dop_lo, dop_hi, prec, cl, t, nthreads, fft_len, nexts, has_next };
        args[t] = a;
        if (t > 0) pthread_create(&th[t], NULL, many_worker, &args[t]);
    }
    many_worker(&args[0]);
    for (t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    for (s = 0; s < nsv; s++) {                           /* search.cpp:455,495 */
        ko_acq_result r = {0.0f, 0, 0, 0};
        float max_snr = 0;
        for (di = 0; di < nd; di++) {
            const ko_acq_cell *c = &cl[(size_t) s * nd + di];
            if (c->snr > max_snr) { max_snr = c->snr; r.dop = dop_lo + di; r.idx = c->idx; r.valid = 1; }
        }
        r.snr = max_snr;
        out[s] = r;
    }
    if (!cells) free(cl);
    free(th); free(args);
}

void ko_correlate_many_n(const ko_cpx *codes, int nsv, const ko_cpx *data,
                         const int *limits, int dop_lo, int dop_hi,
                         ko_acq_result *out, ko_acq_cell *cells, int prec,
                         int nthreads, int fft_len)
{
    ko_correlate_many_next_n(codes, NULL, NULL, nsv, data, limits, dop_lo, dop_hi, out, cells, prec, nthreads, fft_len);
}

void ko_correlate_many(const ko_cpx *codes, int nsv, const ko_cpx *data,
                       const int *limits, int dop_lo, int dop_hi,
                       ko_acq_result *out, ko_acq_cell *cells, int prec,
                       int nthreads)
{
    ko_correlate_many_n(codes, nsv, data, limits, dop_lo, dop_hi, out, cells, prec, nthreads, KO_FFT_LEN);
}
