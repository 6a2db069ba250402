/*
 * kiwi_oracle_libm.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 11: the platform's log10f, powf and expf.  log10f is what the reference's S-meter and CAgc call (rx/rx_sound.cpp:687, rx/CuteSDR/agc.cpp:191) and
 * CAgc branches on (agc.cpp:215-240).  The reference has no log10f of its own; it links libm.  On this image that is the
 * GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11): __ieee754_log10f (sysdeps/ieee754/flt-32/e_log10f.c, the fdlibm
 * wrapper) over logf (sysdeps/ieee754/flt-32/e_logf.c: S. Nagy's 16-interval table method from ARM's optimized routines).
 * powf is CAgc's gain, powf(10, mag * (slope - 1)) (agc.cpp:250-253), expf aperture_auto()'s IIR gain (rx/rx_waterfall.cpp:1199):
 * e_powf.c and e_expf.c of the same directory (log2 by a 16-interval table and a quintic, exp2 by a 32-entry table and a cubic).
 * The published algorithms are restated here; the device's copy of the same restatement is csrc/kg_libm.h.
 *
 * PINNED BY THE IMAGE'S libm ITSELF: ko_libm_check_range() compares the restatement with logf() / log10f() of the libm this
 * file is linked against, bit for bit; over ALL non-negative floats (tools/check_log10f.py --exhaustive) there are 0 differences,
 * with every multiply-add fused or none (the rounding to float hides the difference everywhere); ko_libm_check_pow_exp() does the
 * same for powf(10, y) over all 2^32 y, powf(x, y) over random pairs and expf over all 2^32 x: 0 differences on an FMA-capable
 * host, where glibc runs its FMA build (sysdeps/x86_64/fpu/multiarch) -- expf's residual z - kd is fma(InvLn2N, x, -kd) there, and 2
 * of the 2^32 arguments (0x4202422f, 0xc27c65d9) tell; on a host without FMA the unfused residual is the matching one.  The oracle's own arithmetic
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


/* ---- e_powf.c, e_expf.c ---- */
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

static const double POWF_LOG2_TAB[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
static const uint64_t EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

#define MAD(f, a, b, c) ((f) ? fma(a, b, c) : mad_unfused(a, b, c))
static inline double mad_unfused(double a, double b, double c) { volatile double t = a * b; return t + c; }

static double powf_log2(uint32_t ix, int fused)
{
    const uint32_t tmp = ix - 0x3f330000u, top = tmp & 0xff800000u;
    const int i = (int) ((tmp >> 19) % 16), k = (int32_t) top >> 23;
    const double invc = POWF_LOG2_TAB[i][0], logc = POWF_LOG2_TAB[i][1], z = (double) u2f(ix - top);
    const double r = MAD(fused, z, invc, -1.0), y0 = logc + (double) k, r2 = r * r;
    double y = MAD(fused, 0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = MAD(fused, 0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1), r4 = r2 * r2;
    double q = MAD(fused, 0x1.71547652ab82bp0, r, y0);
    q = MAD(fused, p, r2, q);
    return MAD(fused, y, r4, q);
}

static float exp2_tail(double r, uint64_t ki, uint64_t sign_bias, double c0, double c1, double c2, int fused)
{
    uint64_t t = EXP2F_TAB[ki % 32];
    t += (ki + sign_bias) << 47;
    const double s = u2d(t), z = MAD(fused, c0, r, c1), r2 = r * r;
    double y = MAD(fused, c2, r, 1.0);
    y = MAD(fused, z, r2, y);
    return (float) (y * s);
}

static int checkint(uint32_t iy)
{
    const int e = iy >> 23 & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
static int zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

float ko_powf_restated(float x, float y, int fused)
{
    uint32_t sign_bias = 0, ix = f2u(x), iy = f2u(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || zeroinfnan(iy)) {
        if (zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && checkint(iy) == 1) x2 = -x2;
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        if (ix & 0x80000000u) {
            const int yint = checkint(iy);
            if (yint == 0) return NAN;
            if (yint == 1) sign_bias = 1u << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) { ix = f2u(x * 0x1p23f); ix &= 0x7fffffffu; ix -= 23u << 23; }
    }
    const double ylogx = (double) y * powf_log2(ix, fused);
    if ((d2u(ylogx) >> 47 & 0xffff) >= d2u(126.0) >> 47) {
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -INFINITY : INFINITY;
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    const double SHIFT = 0x1.8p+52 / 32;
    volatile double kdv = ylogx + SHIFT;
    double kd = kdv;
    const uint64_t ki = d2u(kd);
    kd -= SHIFT;
    return exp2_tail(ylogx - kd, ki, sign_bias, 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1, fused);
}

/* fused_residual: r = fma(InvLn2N, x, -kd) (glibc's FMA build) or z - kd */
float ko_expf_restated(float x, int fused, int fused_residual)
{
    const uint32_t abstop = (f2u(x) >> 20) & 0x7ff;
    if (abstop >= (f2u(88.0f) >> 20)) {
        if (f2u(x) == f2u(-INFINITY)) return 0.0f;
        if (abstop >= (f2u(INFINITY) >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    const double xd = (double) x, InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
    volatile double zv = InvLn2N * xd;
    const double z = zv;
    volatile double kdv = z + SHIFT;
    double kd = kdv;
    const uint64_t ki = d2u(kd);
    kd -= SHIFT;
    const double r = fused_residual ? fma(InvLn2N, xd, -kd) : z - kd;
    return exp2_tail(r, ki, 0, 0x1.c6af84b912394p-5 / 32 / 32 / 32, 0x1.ebfce50fac4f3p-3 / 32 / 32, 0x1.62e42ff0c52d6p-1 / 32, fused);
}

void ko_libm_powf_bits(float base, uint32_t first, size_t n, float *y) { for (size_t i = 0; i < n; i++) y[i] = powf(base, u2f(first + (uint32_t) i)); }
void ko_libm_expf_bits(uint32_t first, size_t n, float *y) { for (size_t i = 0; i < n; i++) y[i] = expf(u2f(first + (uint32_t) i)); }
void ko_libm_powf(float base, const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = powf(base, x[i]); }
void ko_libm_expf(const float *x, float *y, size_t n) { for (size_t i = 0; i < n; i++) y[i] = expf(x[i]); }

typedef struct { uint32_t first; uint64_t n, step; int fused, fused_residual; uint64_t bad_pow, bad_exp, bad_rand, nrand; } pe_job;

static void *pe_worker(void *p)
{
    pe_job *j = (pe_job *) p;
    uint64_t s = 0x9e3779b97f4a7c15ull * ((uint64_t) j->first + 1);
    for (uint64_t q = 0; q < j->n; q += j->step) {
        const float v = u2f(j->first + (uint32_t) q);
        if (!same(ko_powf_restated(10.0f, v, j->fused), powf(10.0f, v))) j->bad_pow++;
        if (!same(ko_expf_restated(v, j->fused, j->fused_residual), expf(v))) j->bad_exp++;
        if ((q & 15) == 0) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const float a = u2f((uint32_t) s), b = u2f((uint32_t) (s >> 32));
            if (!same(ko_powf_restated(a, b, j->fused), powf(a, b))) j->bad_rand++;
            j->nrand++;
        }
    }
    return NULL;
}

/* Arguments with bit patterns first, first + step, ... below first + n: powf(10, .) and expf against libm, and powf(a, b) on one random
 * pair per 16 arguments.  -> values compared; differences by kind. */
uint64_t ko_libm_check_pow_exp(uint32_t first, uint64_t n, uint64_t step, int fused, int fused_residual, int threads,
                               uint64_t *bad_pow10, uint64_t *bad_exp, uint64_t *bad_rand, uint64_t *nrand)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (step < 1) step = 1;
    pthread_t th[64];
    pe_job job[64];
    uint64_t done = 0;
    for (int t = 0; t < threads; t++) {
        const uint64_t lo = n * (uint64_t) t / (uint64_t) threads, hi = n * (uint64_t) (t + 1) / (uint64_t) threads;
        job[t] = (pe_job) {(uint32_t) (first + lo), hi - lo, step, fused, fused_residual, 0, 0, 0, 0};
        pthread_create(&th[t], NULL, pe_worker, &job[t]);
    }
    *bad_pow10 = *bad_exp = *bad_rand = *nrand = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        *bad_pow10 += job[t].bad_pow; *bad_exp += job[t].bad_exp; *bad_rand += job[t].bad_rand; *nrand += job[t].nrand;
        done += (job[t].n + step - 1) / step;
    }
    return done;
}
