// kg_fft.h -- in-LDS fp32 complex FFT building blocks for gfx950 (wave64).
//
// 4096-point transform by one 256-thread group: 16 points per thread, three
// radix-16 passes (Stockham autosort, natural order in -> natural order out),
// two exchanges through a 32 KiB LDS tile.  The tile is XOR-swizzled
// (P(e) = e ^ ((e >> 4) & 15)) so that every ds_write_b64 (16-lane groups,
// 32 banks) and ds_read_b64 (32-lane groups, 64 banks) of the three passes is
// bank-conflict free with no padding (checked exhaustively in
// tools/proto_fft.py).
//
// 16384 points = 4 x 4096 over the index residue mod 4 (callers combine the
// four sub-transforms with a radix-4 step; see kg_acq.hip).
//
// The whole library is compiled with -ffp-contract=off: every fused
// multiply-add below is explicit.
#pragma once

#include <hip/hip_runtime.h>
#include "kg_tables.h"

typedef float f2 __attribute__((ext_vector_type(2)));   // (re, im)

#define KG_DEV __device__ __forceinline__

KG_DEV f2 kg_splat(float v) { return f2{v, v}; }

// a * w
KG_DEV f2 kg_cmul(f2 a, f2 w)
{
    f2 r = a * kg_splat(w.x);
    return __builtin_elementwise_fma(f2{-a.y, a.x}, kg_splat(w.y), r);
}
// a * conj(w)
KG_DEV f2 kg_cmulc(f2 a, f2 w)
{
    f2 r = a * kg_splat(w.x);
    return __builtin_elementwise_fma(f2{a.y, -a.x}, kg_splat(w.y), r);
}
// a * w (SIGN > 0) or a * conj(w) (SIGN < 0): tables hold exp(+i...)
template <int SIGN> KG_DEV f2 kg_twmul(f2 a, f2 w)
{
    return SIGN > 0 ? kg_cmul(a, w) : kg_cmulc(a, w);
}
// (SIGN*j) * a
template <int SIGN> KG_DEV f2 kg_mulj(f2 a)
{
    return SIGN > 0 ? f2{-a.y, a.x} : f2{a.y, -a.x};
}

// y_c = sum_a x_a (SIGN*j)^(a*c)
template <int SIGN> KG_DEV void kg_radix4(f2 &x0, f2 &x1, f2 &x2, f2 &x3)
{
    f2 s02 = x0 + x2, d02 = x0 - x2;
    f2 s13 = x1 + x3, d13 = x1 - x3;
    f2 jd = kg_mulj<SIGN>(d13);
    x0 = s02 + s13;
    x1 = d02 + jd;
    x2 = s02 - s13;
    x3 = d02 - jd;
}

template <int SIGN, int K> KG_DEV f2 kg_w16mul(f2 a)
{
    if constexpr (K == 0) return a;
    else if constexpr (K == 4) return kg_mulj<SIGN>(a);
    else {
        const f2 w = f2{KG_W16[K][0], KG_W16[K][1]};
        return kg_twmul<SIGN>(a, w);
    }
}

// In: x[j].  Out: y[m] = sum_j x[j] * exp(SIGN*2*pi*i*j*m/16).
template <int SIGN> KG_DEV void kg_radix16(f2 (&x)[16], f2 (&y)[16])
{
    // stage 1: over a, for each b (j = 4a + b); result u_b[c] lands in x[4c + b]
#pragma unroll
    for (int b = 0; b < 4; b++) kg_radix4<SIGN>(x[b], x[4 + b], x[8 + b], x[12 + b]);
    // twiddle u_b[c] *= W16^(b*c)
    x[5]  = kg_w16mul<SIGN, 1>(x[5]);
    x[6]  = kg_w16mul<SIGN, 2>(x[6]);
    x[7]  = kg_w16mul<SIGN, 3>(x[7]);
    x[9]  = kg_w16mul<SIGN, 2>(x[9]);
    x[10] = kg_w16mul<SIGN, 4>(x[10]);
    x[11] = kg_w16mul<SIGN, 6>(x[11]);
    x[13] = kg_w16mul<SIGN, 3>(x[13]);
    x[14] = kg_w16mul<SIGN, 6>(x[14]);
    x[15] = kg_w16mul<SIGN, 9>(x[15]);
    // stage 2: over b, for each c; Y[c + 4d] lands in x[4c + d]
#pragma unroll
    for (int c = 0; c < 4; c++) kg_radix4<SIGN>(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
#pragma unroll
    for (int m = 0; m < 16; m++) y[m] = x[4 * (m & 3) + (m >> 2)];
}

// Per-thread inter-pass twiddles of the 4096-point transform, thread t of 256:
//   tw1[j-1] = exp(+2*pi*i * j*(t&15) / 256),  tw2[j-1] = exp(+2*pi*i * j*t / 4096)
// taken from a table tab4096[k] = exp(+2*pi*i*k/4096) whose entries are the
// fp32 roundings of double-precision values (host-built, HBM/L2 resident).
struct kg_tw4096 {
    f2 tw1[15];
    f2 tw2[15];
};

KG_DEV void kg_tw4096_load(kg_tw4096 &tw, const f2 *__restrict__ tab4096, int t)
{
#pragma unroll
    for (int j = 1; j < 16; j++) {
        tw.tw1[j - 1] = tab4096[(j * (t & 15)) << 4];
        tw.tw2[j - 1] = tab4096[j * t];
    }
}

// 4096-point transform of x (thread t holds X[t + 256 j], j = 0..15) by a
// 256-thread group sharing the 4096-element LDS tile `lds`.
// Out: y[m] = sum_k X[k] exp(SIGN*2*pi*i*k*n/4096) at n = t + 256 m.
// Contains four __syncthreads(): every thread of the workgroup must call it
// (groups of a larger workgroup run it in lockstep on their own tiles).
template <int SIGN>
KG_DEV void kg_subfft4096(f2 (&x)[16], f2 (&y)[16], f2 *lds, const kg_tw4096 &tw, int t)
{
    const int tl = t & 15, th = t >> 4;
    // pass 0 (no twiddles): out index 16 t + m
    kg_radix16<SIGN>(x, y);
    __syncthreads();                       // previous user of the tile is done reading
    {
        f2 *w = lds + 16 * t;
#pragma unroll
        for (int m = 0; m < 16; m++) w[m ^ tl] = y[m];
    }
    __syncthreads();
    const f2 *r = lds + (t ^ (th & 15));   // P(t + 256 j) = 256 j + (t ^ ((t >> 4) & 15))
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = r[256 * j];
    __syncthreads();
    // pass 1: twiddle W256^(j*(t&15)), out index (t>>4)*256 + (t&15) + 16 m
#pragma unroll
    for (int j = 1; j < 16; j++) x[j] = kg_twmul<SIGN>(x[j], tw.tw1[j - 1]);
    kg_radix16<SIGN>(x, y);
    {
        f2 *w = lds + th * 256;
#pragma unroll
        for (int m = 0; m < 16; m++) w[16 * m + (tl ^ m)] = y[m];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = r[256 * j];
    // pass 2: twiddle W4096^(j*t), out index t + 256 m (kept in registers)
#pragma unroll
    for (int j = 1; j < 16; j++) x[j] = kg_twmul<SIGN>(x[j], tw.tw2[j - 1]);
    kg_radix16<SIGN>(x, y);
}
