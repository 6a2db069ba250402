/*
 * kiwi_oracle_libm.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 11: the platform's log10f, which the reference's S-meter and CAgc call (rx/rx_sound.cpp:687, rx/CuteSDR/agc.cpp:191) and
 * CAgc branches on (agc.cpp:215-240).  The reference has no log10f of its own; it links libm.  On this image that is the
 * GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11): __ieee754_log10f (sysdeps/ieee754/flt-32/e_log10f.c, the fdlibm
 * wrapper) over logf (sysdeps/ieee754/flt-32/e_logf.c: S. Nagy's 16-interval table method from ARM's optimized routines).
 * The two published algorithms are restated here; the device's copy of the same restatement is csrc/kg_libm.h.
 *
 * PINNED BY THE IMAGE'S libm ITSELF: ko_libm_check_range() compares the restatement with logf() / log10f() of the libm this
 * file is linked against, bit for bit; over ALL non-negative floats (tools/check_log10f.py --exhaustive) there are 0 differences,
 * with every multiply-add fused or none (the rounding to float hides the difference everywhere).  The oracle's own arithmetic
 * everywhere else keeps calling libm: this file exists to prove that the DEVICE function equals it.
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static const double LOGF_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

/* e_logf.c; fused = 1 evaluates every multiply-add with fma() (what an FMA build of libm and the device do) */
static float logf_restated(float x, int fused)
{
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {      /* x < 2^-126, inf or nan */
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN;
        ix = f2u(x * 0x1p23f);                                /* subnormal: normalise */
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int) ((tmp >> 19) % 16);
    const int k = (int32_t) tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = LOGF_TAB[i][0], logc = LOGF_TAB[i][1], z = (double) u2f(iz);
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, LN2 = 0x1.62e42fefa39efp-1;
    double r, y0, r2, y;
    if (fused) {
        r = fma(z, invc, -1.0); y0 = fma((double) k, LN2, logc); r2 = r * r;
        y = fma(A1, r, A2); y = fma(A0, r2, y); y = fma(y, r2, y0 + r);
    } else {
        volatile double t;
        t = z * invc; r = t - 1.0; t = (double) k * LN2; y0 = logc + t; r2 = r * r;
        t = A1 * r; y = t + A2; t = A0 * r2; y = t + y; t = y * r2; y = t + (y0 + r);
    }
    return (float) y;
}

/* e_log10f.c */
static float log10f_restated(float x, int fused)
{
    int32_t hx = (int32_t) f2u(x), k = 0, i;
    if (hx < 0x00800000) {
        if ((hx & 0x7fffffff) == 0) return -INFINITY;
        if (hx < 0) return NAN;
        k -= 25;
        x *= 3.3554432000e+07f;
        hx = (int32_t) f2u(x);
    }
    if (hx >= 0x7f800000) return x + x;
    k += (hx >> 23) - 127;
    i = (int32_t) (((uint32_t) k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float) (k + i);
    volatile float a = y * 7.9034151668e-07f, b = 4.3429449201e-01f * logf_restated(u2f((uint32_t) hx), fused);
    volatile float z = a + b, c = y * 3.0102920532e-01f;
    return z + c;
}

float ko_logf_restated(float x, int fused) { return logf_restated(x, fused); }
float ko_log10f_restated(float x, int fused) { return log10f_restated(x, fused); }

/* the image's libm over an array / over consecutive bit patterns (what the device function is compared with) */
void ko_libm_log10f(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = log10f(x[i]); }
void ko_libm_log10f_bits(uint32_t first, size_t n, float *y) { for (size_t i = 0; i < n; i++) y[i] = log10f(u2f(first + (uint32_t) i)); }

typedef struct { uint32_t first; uint64_t n, step; int fused; uint64_t bad_ln, bad_l10; uint32_t first_bad; } libm_job;

static int same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }

static void *libm_worker(void *p)
{
    libm_job *j = (libm_job *) p;
    for (uint64_t q = 0; q < j->n; q += j->step) {
        const uint32_t u = j->first + (uint32_t) q;
        const float x = u2f(u);
        if (!same(logf_restated(x, j->fused), logf(x))) { if (!j->bad_ln && !j->bad_l10) j->first_bad = u; j->bad_ln++; }
        if (!same(log10f_restated(x, j->fused), log10f(x))) { if (!j->bad_ln && !j->bad_l10) j->first_bad = u; j->bad_l10++; }
    }
    return NULL;
}

/* Bit patterns first, first + step, ... below first + n against libm's logf / log10f on `threads` threads.
 * -> the number of values compared; *bad_logf / *bad_log10f = differences, *first_bad = the lowest differing pattern of a thread. */
uint64_t ko_libm_check_range(uint32_t first, uint64_t n, uint64_t step, int fused, int threads, uint64_t *bad_logf,
                             uint64_t *bad_log10f, uint32_t *first_bad)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (step < 1) step = 1;
    pthread_t th[64];
    libm_job job[64];
    uint64_t done = 0;
    for (int t = 0; t < threads; t++) {
        const uint64_t lo = n * (uint64_t) t / (uint64_t) threads, hi = n * (uint64_t) (t + 1) / (uint64_t) threads;
        job[t] = (libm_job) {(uint32_t) (first + lo), hi - lo, step, fused, 0, 0, 0};
        pthread_create(&th[t], NULL, libm_worker, &job[t]);
    }
    *bad_logf = *bad_log10f = 0;
    *first_bad = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        *bad_logf += job[t].bad_ln;
        *bad_log10f += job[t].bad_l10;
        if ((job[t].bad_ln || job[t].bad_l10) && !*first_bad) *first_bad = job[t].first_bad;
        done += (job[t].n + step - 1) / step;
    }
    return done;
}
