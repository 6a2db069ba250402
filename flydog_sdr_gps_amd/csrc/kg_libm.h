// kg_libm.h -- log10f as the reference's host computes it, on the device, bit for bit.
//
// Why: the S-meter and CAgc of c2s_sound() take log10f of every sample (rx/rx_sound.cpp:687, rx/CuteSDR/agc.cpp:191), and CAgc's
// averagers and hang timer BRANCH on those values (agc.cpp:215-240).  With the device's own log10f (1 ulp, but not the same
// ulp) one sample in ~10^6 random trials took the other branch and the output carried a constant 1.7e-4 gain step from there on
// (VERDICT r5, weak 1).  The reference has no log10f of its own: it links the platform's libm.  The checker and the reference
// pieces built in place (the test infrastructure) run on this image's GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11, libm.so.6), whose
// log10f is __ieee754_log10f of sysdeps/ieee754/flt-32/e_log10f.c (the fdlibm wrapper: exponent split, three float operations)
// over logf of sysdeps/ieee754/flt-32/e_logf.c (Szabolcs Nagy's table method from ARM's optimized routines: 16 intervals, a
// cubic in double, one rounding to float).  Both algorithms are restated here from their published form; nothing of glibc is
// in the repository.  Pinned: the same restatement in C (the test infrastructure's part 11) equals the image's logf AND log10f on ALL
// 2 139 095 041 non-negative floats (tools/check_log10f.py --exhaustive, 13 s on 8 cores; with and without fused
// multiply-adds: the final rounding to float hides the difference everywhere), and tests/test_libm_gpu.py compares this device
// function with the image's log10f through the C ABI (kg_math_log10f_dev).
#ifndef KG_LIBM_H
#define KG_LIBM_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kg_libm {
// logf_data: c is near the centre of the i-th of 16 intervals of one binade starting at OFF = 0x3f330000, invc = 1/c, logc = round(log(c))
__device__ __constant__ const double logf_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};

// logf for a finite positive normal x (what log10f hands it: 0.5 <= x < 2)
__device__ __forceinline__ float logf_normal(float x)
{
    const uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int) ((tmp >> 19) & 15u);
    const int k = (int32_t) tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = logf_tab[i][0], logc = logf_tab[i][1];
    const double z = (double) __uint_as_float(iz);
    // log(x) = log1p(z/c - 1) + log(c) + k ln 2; every multiply-add fused (the variant checked exhaustively)
    const double r = __fma_rn(z, invc, -1.0);
    const double y0 = __fma_rn((double) k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = __dmul_rn(r, r);
    double y = __fma_rn(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = __fma_rn(-0x1.00ea348b88334p-2, r2, y);
    y = __fma_rn(y, r2, __dadd_rn(y0, r));
    return __double2float_rn(y);
}

// __ieee754_log10f
__device__ __forceinline__ float log10f_glibc(float x)
{
    int32_t hx = (int32_t) __float_as_uint(x), k = 0;
    if (hx < 0x00800000) {                                    // x < 2^-126
        if ((hx & 0x7fffffff) == 0) return -__builtin_huge_valf();
        if (hx < 0) return __builtin_nanf("");
        k -= 25;
        x = __fmul_rn(x, 3.3554432000e+07f);                  // subnormal: scale up by 2^25
        hx = (int32_t) __float_as_uint(x);
    }
    if (hx >= 0x7f800000) return __fadd_rn(x, x);
    k += (hx >> 23) - 127;
    const int32_t i = (int32_t) (((uint32_t) k & 0x80000000u) >> 31);
    hx = (hx & 0x007fffff) | ((0x7f - i) << 23);
    const float y = (float) (k + i);
    const float m = __uint_as_float((uint32_t) hx);
    const float z = __fadd_rn(__fmul_rn(y, 7.9034151668e-07f), __fmul_rn(4.3429449201e-01f, logf_normal(m)));
    return __fadd_rn(z, __fmul_rn(y, 3.0102920532e-01f));
}
}  // namespace kg_libm
#endif
