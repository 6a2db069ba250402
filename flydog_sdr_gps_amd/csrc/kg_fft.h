// kg_fft.h -- in-LDS fp32 complex FFT building blocks for gfx950 (wave64).
//
// 4096-point transform by one 256-thread group: 16 points per thread, three
// radix-16 passes (Stockham autosort, natural order in -> natural order out),
// two exchanges through two 32 KiB LDS tiles used alternately, so one
// workgroup barrier per exchange is enough (the tile being written was last
// read two barriers ago).  The tiles are XOR-swizzled (P(e) = e ^ ((e>>4)&15)):
// every ds_write_b64 (16-lane groups, 32 banks) and ds_read_b64 (32-lane
// groups, 64 banks) of the three passes is bank-conflict free with no padding
// (checked exhaustively in tools/proto_fft.py; SQ_LDS_BANK_CONFLICT = 0 on
// hardware, profiles/).
//
// Arithmetic: a complex value is one 64-bit VGPR pair and every operation is a
// packed v_pk_{add,mul,fma}_f32.  A wave issues one VALU instruction per four
// cycles, a packed one keeps its SIMD-32 busy for all four, so a single wave
// can saturate the SIMD -- which matters because the kernels built on this run
// at two waves per SIMD.  hipcc folds broadcasts into op_sel but materialises
// the swap-and-negate of a complex product / of a multiplication by +-j with
// v_xor + v_mov; those two patterns are therefore written as inline asm with
// op_sel / neg_lo / neg_hi modifiers (plain VALU: no manual wait states needed).
// The library is built with -ffp-contract=off (every fused multiply-add is
// explicit) and -fno-slp-vectorize.
//
// 16384 points = 4 x 4096 over the index residue mod 4 (callers combine the
// four sub-transforms with a radix-4 step; see kg_acq.hip).
#pragma once

#include <hip/hip_runtime.h>
#include "kg_tables.h"

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im) = float2 = fftwf_complex

#define KG_DEV __device__ __forceinline__

KG_DEV cf kg_scale(cf a, float s) { return a * cf{s, s}; }
KG_DEV cf kg_ld(const float2 *p) { return *reinterpret_cast<const cf *>(p); }
KG_DEV void kg_st(float2 *p, cf v) { *reinterpret_cast<cf *>(p) = v; }
// LDS tile read: one ds_read_b64 per element.  Left alone, hipcc pairs two of them into a
// ds_read2st64_b64, which occupies the LDS for 8 cycles against 2 + 2 (MI355X_MICROARCH.md, LDS table);
// the volatile LDS-address-space access keeps them apart (-1.6 % correlator, -1.8 % waterfall frames).
KG_DEV cf kg_ld_tile(const float2 *p) { return *(const volatile cf __attribute__((address_space(3))) *) p; }
// LDS tile store: one ds_write_b64 per element.  Left alone, hipcc pairs stores a constant stride apart into ds_write2_b64 /
// ds_write2st64_b64: 13 cycles of the store path against 6 + 6 (MI355X_MICROARCH.md, LDS table) -- and the pair leaves only when both
// values are there.  Round 5, A/B on one box: -2 % on the 16368-lag correlator (KG_TILE_ST_PAIRED=1 restores the paired form).
#ifndef KG_TILE_ST_PAIRED
#define KG_TILE_ST_PAIRED 0
#endif
KG_DEV void kg_st_tile(float2 *p, cf v)
{
#if KG_TILE_ST_PAIRED
    *reinterpret_cast<cf *>(p) = v;
#else
    *(volatile cf __attribute__((address_space(3))) *) p = v;
#endif
}

// a * w = (a.x w.x - a.y w.y, a.y w.x + a.x w.y)
KG_DEV cf kg_cmul(cf a, cf w)
{
    cf r, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=v"(d) : "v"(a), "v"(w), "v"(r));
    return d;
}
// a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
KG_DEV cf kg_cmulc(cf a, cf w)
{
    cf r, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(d) : "v"(a), "v"(w), "v"(r));
    return d;
}
// the same with w wave-uniform (an SGPR pair: compile-time or s_load'ed constants)
KG_DEV cf kg_cmul_s(cf a, cf w)
{
    cf r, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=v"(d) : "v"(a), "s"(w), "v"(r));
    return d;
}
KG_DEV cf kg_cmulc_s(cf a, cf w)
{
    cf r, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : "=v"(d) : "v"(a), "s"(w), "v"(r));
    return d;
}
// a + j b = (a.x - b.y, a.y + b.x)
KG_DEV cf kg_addj(cf a, cf b)
{
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a - j b = (a.x + b.y, a.y - b.x)
KG_DEV cf kg_subj(cf a, cf b)
{
    cf d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * w (SIGN > 0) or a * conj(w) (SIGN < 0): tables hold exp(+i...)
template <int SIGN> KG_DEV cf kg_twmul(cf a, cf w)
{
    return SIGN > 0 ? kg_cmul(a, w) : kg_cmulc(a, w);
}
template <int SIGN> KG_DEV cf kg_twmul_s(cf a, cf w)
{
    return SIGN > 0 ? kg_cmul_s(a, w) : kg_cmulc_s(a, w);
}
// a + (SIGN*j) b,  a - (SIGN*j) b
template <int SIGN> KG_DEV cf kg_add_sj(cf a, cf b) { return SIGN > 0 ? kg_addj(a, b) : kg_subj(a, b); }
template <int SIGN> KG_DEV cf kg_sub_sj(cf a, cf b) { return SIGN > 0 ? kg_subj(a, b) : kg_addj(a, b); }

// ---------------------------------------------------------------------------
// Batched complex products.  hipcc puts one wait state (s_nop) between an inline-asm definition and
// the instruction that reads it (it cannot see that the asm is not a partial-register write), and a
// product's multiply feeds its fused multiply-add directly: 85 of the 451 vector instructions of a
// correlator item were followed by an s_nop and a dependent issue.  These blocks hold four independent
// products each -- the four multiplies, then the four multiply-adds -- so neither happens.
// a_i <- a_i * w_i (CONJ = false) or a_i * conj(w_i) (CONJ = true), in place.
// ---------------------------------------------------------------------------
#define KG_MUL_(r, a, w) "v_pk_mul_f32 " r ", " a ", " w " op_sel_hi:[1,0]\n\t"
#define KG_FMA_(a, w, r, neg) "v_pk_fma_f32 " a ", " a ", " w ", " r " op_sel:[1,1,0] op_sel_hi:[0,1,1] " neg "\n\t"

// o_i = a * w_i for one a and three wave-uniform w_i, NOT in place (the in-place blocks need a copy of a
// per product when a is used again: 15 v_mov_b64 per correlator item went on that)
#define KG_FMA3_(o, a, w, neg) "v_pk_fma_f32 " o ", " a ", " w ", " o " op_sel:[1,1,0] op_sel_hi:[0,1,1] " neg "\n\t"
KG_DEV void kg_cmul1x3s(cf &o0, cf &o1, cf &o2, cf a, cf w0, cf w1, cf w2)
{
    asm(KG_MUL_("%0", "%3", "%4") KG_MUL_("%1", "%3", "%5") KG_MUL_("%2", "%3", "%6")
        KG_FMA3_("%0", "%3", "%4", "neg_lo:[0,1,0]") KG_FMA3_("%1", "%3", "%5", "neg_lo:[0,1,0]")
        KG_FMA3_("%2", "%3", "%6", "neg_lo:[0,1,0]")
        : "=&v"(o0), "=&v"(o1), "=&v"(o2) : "v"(a), "s"(w0), "s"(w1), "s"(w2));
}

// the same with the w_i in vector registers
KG_DEV void kg_cmul1x3v(cf &o0, cf &o1, cf &o2, cf a, cf w0, cf w1, cf w2)
{
    asm(KG_MUL_("%0", "%3", "%4") KG_MUL_("%1", "%3", "%5") KG_MUL_("%2", "%3", "%6")
        KG_FMA3_("%0", "%3", "%4", "neg_lo:[0,1,0]") KG_FMA3_("%1", "%3", "%5", "neg_lo:[0,1,0]")
        KG_FMA3_("%2", "%3", "%6", "neg_lo:[0,1,0]")
        : "=&v"(o0), "=&v"(o1), "=&v"(o2) : "v"(a), "v"(w0), "v"(w1), "v"(w2));
}

template <bool CONJ> KG_DEV void kg_cmul4v(cf &a0, cf &a1, cf &a2, cf &a3, cf w0, cf w1, cf w2, cf w3)
{
    cf r0, r1, r2, r3;
    if constexpr (!CONJ)
        asm(KG_MUL_("%4", "%0", "%8") KG_MUL_("%5", "%1", "%9") KG_MUL_("%6", "%2", "%10") KG_MUL_("%7", "%3", "%11")
            KG_FMA_("%0", "%8", "%4", "neg_lo:[0,1,0]") KG_FMA_("%1", "%9", "%5", "neg_lo:[0,1,0]")
            KG_FMA_("%2", "%10", "%6", "neg_lo:[0,1,0]") KG_FMA_("%3", "%11", "%7", "neg_lo:[0,1,0]")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    else
        asm(KG_MUL_("%4", "%0", "%8") KG_MUL_("%5", "%1", "%9") KG_MUL_("%6", "%2", "%10") KG_MUL_("%7", "%3", "%11")
            KG_FMA_("%0", "%8", "%4", "neg_hi:[0,1,0]") KG_FMA_("%1", "%9", "%5", "neg_hi:[0,1,0]")
            KG_FMA_("%2", "%10", "%6", "neg_hi:[0,1,0]") KG_FMA_("%3", "%11", "%7", "neg_hi:[0,1,0]")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
// o_i = a_i * conj(w_i), NOT in place (a_i and w_i stay live: no copy of an operand that is overwritten later anyway)
KG_DEV void kg_cmulc4_o(cf &o0, cf &o1, cf &o2, cf &o3, cf a0, cf a1, cf a2, cf a3, cf w0, cf w1, cf w2, cf w3)
{
#define KG_FMAO_(o, a, w) "v_pk_fma_f32 " o ", " a ", " w ", " o " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
    asm(KG_MUL_("%0", "%4", "%8") KG_MUL_("%1", "%5", "%9") KG_MUL_("%2", "%6", "%10") KG_MUL_("%3", "%7", "%11")
        KG_FMAO_("%0", "%4", "%8") KG_FMAO_("%1", "%5", "%9") KG_FMAO_("%2", "%6", "%10") KG_FMAO_("%3", "%7", "%11")
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
#undef KG_FMAO_
}
// the same with wave-uniform w_i (SGPR pairs)
template <bool CONJ> KG_DEV void kg_cmul4s(cf &a0, cf &a1, cf &a2, cf &a3, cf w0, cf w1, cf w2, cf w3)
{
    cf r0, r1, r2, r3;
    if constexpr (!CONJ)
        asm(KG_MUL_("%4", "%0", "%8") KG_MUL_("%5", "%1", "%9") KG_MUL_("%6", "%2", "%10") KG_MUL_("%7", "%3", "%11")
            KG_FMA_("%0", "%8", "%4", "neg_lo:[0,1,0]") KG_FMA_("%1", "%9", "%5", "neg_lo:[0,1,0]")
            KG_FMA_("%2", "%10", "%6", "neg_lo:[0,1,0]") KG_FMA_("%3", "%11", "%7", "neg_lo:[0,1,0]")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "s"(w0), "s"(w1), "s"(w2), "s"(w3));
    else
        asm(KG_MUL_("%4", "%0", "%8") KG_MUL_("%5", "%1", "%9") KG_MUL_("%6", "%2", "%10") KG_MUL_("%7", "%3", "%11")
            KG_FMA_("%0", "%8", "%4", "neg_hi:[0,1,0]") KG_FMA_("%1", "%9", "%5", "neg_hi:[0,1,0]")
            KG_FMA_("%2", "%10", "%6", "neg_hi:[0,1,0]") KG_FMA_("%3", "%11", "%7", "neg_hi:[0,1,0]")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "s"(w0), "s"(w1), "s"(w2), "s"(w3));
}
// three products (the last group of the fifteen inter-pass twiddles)
template <bool CONJ> KG_DEV void kg_cmul3v(cf &a0, cf &a1, cf &a2, cf w0, cf w1, cf w2)
{
    cf r0, r1, r2;
    if constexpr (!CONJ)
        asm(KG_MUL_("%3", "%0", "%6") KG_MUL_("%4", "%1", "%7") KG_MUL_("%5", "%2", "%8")
            KG_FMA_("%0", "%6", "%3", "neg_lo:[0,1,0]") KG_FMA_("%1", "%7", "%4", "neg_lo:[0,1,0]")
            KG_FMA_("%2", "%8", "%5", "neg_lo:[0,1,0]")
            : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(w0), "v"(w1), "v"(w2));
    else
        asm(KG_MUL_("%3", "%0", "%6") KG_MUL_("%4", "%1", "%7") KG_MUL_("%5", "%2", "%8")
            KG_FMA_("%0", "%6", "%3", "neg_hi:[0,1,0]") KG_FMA_("%1", "%7", "%4", "neg_hi:[0,1,0]")
            KG_FMA_("%2", "%8", "%5", "neg_hi:[0,1,0]")
            : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(w0), "v"(w1), "v"(w2));
}
// acc_i += y_i * w_i for four points: two fused multiply-adds per point (first the x halves, then the
// y halves), nothing reads the register written by the previous instruction
KG_DEV void kg_cmac4v(cf &c0, cf &c1, cf &c2, cf &c3, cf y0, cf y1, cf y2, cf y3, cf w0, cf w1, cf w2, cf w3)
{
#define KG_MAC1_(c, y, w) "v_pk_fma_f32 " c ", " y ", " w ", " c " op_sel_hi:[0,1,1]\n\t"
#define KG_MAC2_(c, y, w) "v_pk_fma_f32 " c ", " y ", " w ", " c " op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
    asm(KG_MAC1_("%0", "%4", "%8") KG_MAC1_("%1", "%5", "%9") KG_MAC1_("%2", "%6", "%10") KG_MAC1_("%3", "%7", "%11")
        KG_MAC2_("%0", "%4", "%8") KG_MAC2_("%1", "%5", "%9") KG_MAC2_("%2", "%6", "%10") KG_MAC2_("%3", "%7", "%11")
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
// the same with one factor w (a VGPR pair) for all four points
KG_DEV void kg_cmac4v1(cf &c0, cf &c1, cf &c2, cf &c3, cf y0, cf y1, cf y2, cf y3, cf w)
{
    asm(KG_MAC1_("%0", "%4", "%8") KG_MAC1_("%1", "%5", "%8") KG_MAC1_("%2", "%6", "%8") KG_MAC1_("%3", "%7", "%8")
        KG_MAC2_("%0", "%4", "%8") KG_MAC2_("%1", "%5", "%8") KG_MAC2_("%2", "%6", "%8") KG_MAC2_("%3", "%7", "%8")
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "v"(w));
}
// the same with one wave-uniform factor w (an SGPR pair) for all four points
KG_DEV void kg_cmac4s(cf &c0, cf &c1, cf &c2, cf &c3, cf y0, cf y1, cf y2, cf y3, cf w)
{
    asm(KG_MAC1_("%0", "%4", "%8") KG_MAC1_("%1", "%5", "%8") KG_MAC1_("%2", "%6", "%8") KG_MAC1_("%3", "%7", "%8")
        KG_MAC2_("%0", "%4", "%8") KG_MAC2_("%1", "%5", "%8") KG_MAC2_("%2", "%6", "%8") KG_MAC2_("%3", "%7", "%8")
        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
        : "v"(y0), "v"(y1), "v"(y2), "v"(y3), "s"(w));
}
#undef KG_MAC1_
#undef KG_MAC2_

// y_c = sum_a x_a (SIGN*j)^(a*c).  X2J: x2 still has to be multiplied by SIGN*j
// (the W16^4 twiddle of the radix-16, folded into this butterfly's first adds).
// (p, q) = (a + (SIGN*j) b, a - (SIGN*j) b) in one block: two inline-asm definitions that the next
// instruction reads would each cost a wait state
template <int SIGN> KG_DEV void kg_addsub_sj(cf &p, cf &q, cf a, cf b)
{
    if constexpr (SIGN > 0)
        asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
            : "=&v"(p), "=v"(q) : "v"(a), "v"(b));
    else
        asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
            : "=&v"(p), "=v"(q) : "v"(a), "v"(b));
}
template <int SIGN, bool X2J = false> KG_DEV void kg_radix4(cf &x0, cf &x1, cf &x2, cf &x3)
{
    cf s02, d02;
    if constexpr (X2J) kg_addsub_sj<SIGN>(s02, d02, x0, x2);
    else { s02 = x0 + x2; d02 = x0 - x2; }
    const cf s13 = x1 + x3, d13 = x1 - x3;
    x0 = s02 + s13;
    x2 = s02 - s13;
    kg_addsub_sj<SIGN>(x1, x3, d02, d13);
}

template <int SIGN, int K> KG_DEV cf kg_w16mul(cf a)
{
    return kg_twmul_s<SIGN>(a, cf{KG_W16[K][0], KG_W16[K][1]});
}

// In: x[j].  Out: y[m] = sum_j x[j] * exp(SIGN*2*pi*i*j*m/16).   80 packed instructions.
// (kg_w16mul: the single-product form of the internal twiddles, kept for the one-wave transforms.)
template <int SIGN> KG_DEV void kg_radix16(cf (&x)[16], cf (&y)[16])
{
    // stage 1: over a, for each b (j = 4a + b); u_b[c] lands in x[4c + b]
#pragma unroll
    for (int b = 0; b < 4; b++) kg_radix4<SIGN>(x[b], x[4 + b], x[8 + b], x[12 + b]);
    // u_b[c] *= W16^(b*c)   (b*c = 4 is folded into stage 2)
#define KG_W16C(K) cf{KG_W16[K][0], KG_W16[K][1]}
    kg_cmul4s<(SIGN < 0)>(x[5], x[6], x[7], x[9], KG_W16C(1), KG_W16C(2), KG_W16C(3), KG_W16C(2));
    kg_cmul4s<(SIGN < 0)>(x[11], x[13], x[14], x[15], KG_W16C(6), KG_W16C(3), KG_W16C(6), KG_W16C(9));
#undef KG_W16C
    // stage 2: over b, for each c; Y[c + 4d] lands in x[4c + d]
    kg_radix4<SIGN>(x[0], x[1], x[2], x[3]);
    kg_radix4<SIGN>(x[4], x[5], x[6], x[7]);
    kg_radix4<SIGN, true>(x[8], x[9], x[10], x[11]);
    kg_radix4<SIGN>(x[12], x[13], x[14], x[15]);
#pragma unroll
    for (int m = 0; m < 16; m++) y[m] = x[4 * (m & 3) + (m >> 2)];
}

struct kg_tw15 { cf w[15]; };     // the fifteen inter-pass twiddles of one pass (w[j - 1] for input j)

// ---------------------------------------------------------------------------
// Round 4: twiddles FUSED into the butterflies.  A radix-2 step on a twiddled operand, (u + w z, u - w z), is written
//     s = u + w z     two fused multiply-adds (the product is never formed on its own)
//     d = 2 u - s     one
// -- three packed instructions where product, sum and difference took four.  Every pair of a radix-4 with a twiddled
// member saves one: 8 in a first stage behind inter-pass twiddles (or behind the conjugate products of pass 0), 5 in a
// second stage (the internal W16 constants): 13 of the 110 packed instructions of twiddle16 + radix16.  d carries the
// rounding of s instead of the product's own -- the same size of error, a different value: results move in the last
// bits (the parity bars are 1e-5 of the spectrum's maximum; peak indices stay exact).
// Blocks of up to four independent operations per asm statement, as above (no wait state per product).
// ---------------------------------------------------------------------------
#define KG_CFMA1_(t, a, w, c) "v_pk_fma_f32 " t ", " a ", " w ", " c " op_sel_hi:[1,0,1]\n\t"
#define KG_CFMA2P_(t, a, w) "v_pk_fma_f32 " t ", " a ", " w ", " t " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
#define KG_CFMA2C_(t, a, w) "v_pk_fma_f32 " t ", " a ", " w ", " t " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
#define KG_2CMT_(d, c, two, t) "v_pk_fma_f32 " d ", " c ", " two ", " t " neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
// t_i = c_i + a_i w_i (CONJ: a_i conj(w_i)), four at a time, w_i in vector registers
template <bool CONJ> KG_DEV void kg_cfma4v(cf &t0, cf &t1, cf &t2, cf &t3, cf c0, cf c1, cf c2, cf c3,
                                           cf a0, cf a1, cf a2, cf a3, cf w0, cf w1, cf w2, cf w3)
{
    if constexpr (!CONJ)
        asm(KG_CFMA1_("%0", "%8", "%12", "%4") KG_CFMA1_("%1", "%9", "%13", "%5") KG_CFMA1_("%2", "%10", "%14", "%6") KG_CFMA1_("%3", "%11", "%15", "%7")
            KG_CFMA2P_("%0", "%8", "%12") KG_CFMA2P_("%1", "%9", "%13") KG_CFMA2P_("%2", "%10", "%14") KG_CFMA2P_("%3", "%11", "%15")
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    else
        asm(KG_CFMA1_("%0", "%8", "%12", "%4") KG_CFMA1_("%1", "%9", "%13", "%5") KG_CFMA1_("%2", "%10", "%14", "%6") KG_CFMA1_("%3", "%11", "%15", "%7")
            KG_CFMA2C_("%0", "%8", "%12") KG_CFMA2C_("%1", "%9", "%13") KG_CFMA2C_("%2", "%10", "%14") KG_CFMA2C_("%3", "%11", "%15")
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
// o_i = a_i w_i (CONJ: a_i conj(w_i)), NOT in place, four at a time
template <bool CONJ> KG_DEV void kg_cmul4v_o(cf &o0, cf &o1, cf &o2, cf &o3, cf a0, cf a1, cf a2, cf a3, cf w0, cf w1, cf w2, cf w3)
{
    if constexpr (!CONJ)
        asm(KG_MUL_("%0", "%4", "%8") KG_MUL_("%1", "%5", "%9") KG_MUL_("%2", "%6", "%10") KG_MUL_("%3", "%7", "%11")
            KG_CFMA2P_("%0", "%4", "%8") KG_CFMA2P_("%1", "%5", "%9") KG_CFMA2P_("%2", "%6", "%10") KG_CFMA2P_("%3", "%7", "%11")
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    else
        asm(KG_MUL_("%0", "%4", "%8") KG_MUL_("%1", "%5", "%9") KG_MUL_("%2", "%6", "%10") KG_MUL_("%3", "%7", "%11")
            KG_CFMA2C_("%0", "%4", "%8") KG_CFMA2C_("%1", "%5", "%9") KG_CFMA2C_("%2", "%6", "%10") KG_CFMA2C_("%3", "%7", "%11")
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}
// d_i = 2 c_i - t_i, four at a time (d_i may be c_i)
KG_DEV void kg_2cmt4(cf &d0, cf &d1, cf &d2, cf &d3, cf c0, cf c1, cf c2, cf c3, cf t0, cf t1, cf t2, cf t3)
{
    const cf two = cf{2.0f, 2.0f};
    asm(KG_2CMT_("%0", "%4", "%12", "%8") KG_2CMT_("%1", "%5", "%12", "%9") KG_2CMT_("%2", "%6", "%12", "%10") KG_2CMT_("%3", "%7", "%12", "%11")
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(t0), "v"(t1), "v"(t2), "v"(t3), "s"(two));
}

// Second stage of the radix-16 with its internal twiddles fused.  In: x[4c + d] = the first stage's u_d[c] (NOT yet multiplied
// by W16^(c d)); out: y[c + 4 d]; hook(4 + c) when row c is final.  43 packed instructions (16 + 32 as products + radix-4s).
template <int SIGN, class H> KG_DEV void kg_radix16_stage2f(cf (&x)[16], cf (&y)[16], H hook)
{
#define KG_W16C(K) cf{KG_W16[K][0], KG_W16[K][1]}
    const cf two = cf{2.0f, 2.0f};
    kg_radix4<SIGN>(x[0], x[1], x[2], x[3]);                        // row 0: no twiddles
    y[0] = x[0]; y[4] = x[1]; y[8] = x[2]; y[12] = x[3];
    hook(4);
    // rows c = 1, 2, 3 (z_d = x[4c + d]):  u1 = W^c z1,  s13 = u1 + W^(3c) z3,  d13 = 2 u1 - s13,
    //                                      s02 = z0 + W^(2c) z2,  d02 = 2 z0 - s02   (c = 2: W^4 = SIGN j: an add and a subtract)
    cf u1[3], s13[3], d13[3], s02a, s02b, d02a, d02b;
    if constexpr (SIGN > 0) {
        asm(KG_MUL_("%0", "%3", "%6") KG_MUL_("%1", "%4", "%7") KG_MUL_("%2", "%5", "%8")
            KG_CFMA2P_("%0", "%3", "%6") KG_CFMA2P_("%1", "%4", "%7") KG_CFMA2P_("%2", "%5", "%8")
            : "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]) : "v"(x[5]), "v"(x[9]), "v"(x[13]), "s"(KG_W16C(1)), "s"(KG_W16C(2)), "s"(KG_W16C(3)));
        asm(KG_CFMA1_("%0", "%6", "%9", "%3") KG_CFMA1_("%1", "%7", "%10", "%4") KG_CFMA1_("%2", "%8", "%11", "%5")
            KG_CFMA2P_("%0", "%6", "%9") KG_CFMA2P_("%1", "%7", "%10") KG_CFMA2P_("%2", "%8", "%11")
            : "=&v"(s13[0]), "=&v"(s13[1]), "=&v"(s13[2])
            : "v"(u1[0]), "v"(u1[1]), "v"(u1[2]), "v"(x[7]), "v"(x[11]), "v"(x[15]), "s"(KG_W16C(3)), "s"(KG_W16C(6)), "s"(KG_W16C(9)));
        asm(KG_CFMA1_("%0", "%4", "%6", "%2") KG_CFMA1_("%1", "%5", "%7", "%3")
            KG_CFMA2P_("%0", "%4", "%6") KG_CFMA2P_("%1", "%5", "%7")
            : "=&v"(s02a), "=&v"(s02b) : "v"(x[4]), "v"(x[12]), "v"(x[6]), "v"(x[14]), "s"(KG_W16C(2)), "s"(KG_W16C(6)));
    } else {
        asm(KG_MUL_("%0", "%3", "%6") KG_MUL_("%1", "%4", "%7") KG_MUL_("%2", "%5", "%8")
            KG_CFMA2C_("%0", "%3", "%6") KG_CFMA2C_("%1", "%4", "%7") KG_CFMA2C_("%2", "%5", "%8")
            : "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]) : "v"(x[5]), "v"(x[9]), "v"(x[13]), "s"(KG_W16C(1)), "s"(KG_W16C(2)), "s"(KG_W16C(3)));
        asm(KG_CFMA1_("%0", "%6", "%9", "%3") KG_CFMA1_("%1", "%7", "%10", "%4") KG_CFMA1_("%2", "%8", "%11", "%5")
            KG_CFMA2C_("%0", "%6", "%9") KG_CFMA2C_("%1", "%7", "%10") KG_CFMA2C_("%2", "%8", "%11")
            : "=&v"(s13[0]), "=&v"(s13[1]), "=&v"(s13[2])
            : "v"(u1[0]), "v"(u1[1]), "v"(u1[2]), "v"(x[7]), "v"(x[11]), "v"(x[15]), "s"(KG_W16C(3)), "s"(KG_W16C(6)), "s"(KG_W16C(9)));
        asm(KG_CFMA1_("%0", "%4", "%6", "%2") KG_CFMA1_("%1", "%5", "%7", "%3")
            KG_CFMA2C_("%0", "%4", "%6") KG_CFMA2C_("%1", "%5", "%7")
            : "=&v"(s02a), "=&v"(s02b) : "v"(x[4]), "v"(x[12]), "v"(x[6]), "v"(x[14]), "s"(KG_W16C(2)), "s"(KG_W16C(6)));
    }
    // the five differences d = 2 u - s
    asm(KG_2CMT_("%0", "%5", "%15", "%10") KG_2CMT_("%1", "%6", "%15", "%11") KG_2CMT_("%2", "%7", "%15", "%12")
        KG_2CMT_("%3", "%8", "%15", "%13") KG_2CMT_("%4", "%9", "%15", "%14")
        : "=&v"(d13[0]), "=&v"(d13[1]), "=&v"(d13[2]), "=&v"(d02a), "=&v"(d02b)
        : "v"(u1[0]), "v"(u1[1]), "v"(u1[2]), "v"(x[4]), "v"(x[12]),
          "v"(s13[0]), "v"(s13[1]), "v"(s13[2]), "v"(s02a), "v"(s02b), "s"(two));
    // row 1
    y[1] = s02a + s13[0]; y[9] = s02a - s13[0];
    kg_addsub_sj<SIGN>(y[5], y[13], d02a, d13[0]);
    hook(5);
    // row 2: s02, d02 = z0 +- (SIGN j) z2
    {
        cf s02, d02;
        kg_addsub_sj<SIGN>(s02, d02, x[8], x[10]);
        y[2] = s02 + s13[1]; y[10] = s02 - s13[1];
        kg_addsub_sj<SIGN>(y[6], y[14], d02, d13[1]);
    }
    hook(6);
    // row 3
    y[3] = s02b + s13[2]; y[11] = s02b - s13[2];
    kg_addsub_sj<SIGN>(y[7], y[15], d02b, d13[2]);
    hook(7);
#undef KG_W16C
}

// First stage behind inter-pass twiddles, fused:  in X[j] (j = 4a + b) and the fifteen twiddles w (w.w[j - 1] for X[j]);
// out x[4c + b] = u_b[c] = sum_a (w X)[4a + b] (SIGN j)^(a c), the input of kg_radix16_stage2f.  hook(0..3) between its blocks.
// 54 packed instructions (30 + 32 as products + radix-4s).
template <int SIGN, class H> KG_DEV void kg_radix16_stage1_tw(cf (&x)[16], const kg_tw15 &w, H hook)
{
    constexpr bool CJ = SIGN < 0;
    // u0_b = w_b X_b (b = 1..3; X_0 has no twiddle), u1_b = w_(4+b) X_(4+b)
    kg_cmul3v<CJ>(x[1], x[2], x[3], w.w[0], w.w[1], w.w[2]);
    hook(0);
    kg_cmul4v<CJ>(x[4], x[5], x[6], x[7], w.w[3], w.w[4], w.w[5], w.w[6]);
    hook(1);
    cf s02[4], s13[4];
    kg_cfma4v<CJ>(s02[0], s02[1], s02[2], s02[3], x[0], x[1], x[2], x[3], x[8], x[9], x[10], x[11], w.w[7], w.w[8], w.w[9], w.w[10]);
    hook(2);
    kg_cfma4v<CJ>(s13[0], s13[1], s13[2], s13[3], x[4], x[5], x[6], x[7], x[12], x[13], x[14], x[15], w.w[11], w.w[12], w.w[13], w.w[14]);
    hook(3);
    cf d02[4], d13[4];
    kg_2cmt4(d02[0], d02[1], d02[2], d02[3], x[0], x[1], x[2], x[3], s02[0], s02[1], s02[2], s02[3]);
    kg_2cmt4(d13[0], d13[1], d13[2], d13[3], x[4], x[5], x[6], x[7], s13[0], s13[1], s13[2], s13[3]);
#pragma unroll
    for (int b = 0; b < 4; b++) {
        x[b] = s02[b] + s13[b];
        x[8 + b] = s02[b] - s13[b];
        kg_addsub_sj<SIGN>(x[4 + b], x[12 + b], d02[b], d13[b]);
    }
}

// First stage of pass 0 of a correlator item, the conjugate products fused in:  p_j = c_j conj(d_j) (simd_multiply_conjugate_ccc,
// support/simd.cpp:39-67) never formed on their own for j >= 8; c and d are left intact.  hook(s), s = 0..3, is called when
// operands 4 s .. 4 s + 3 (rows 2 s, 2 s + 1 of the fetch) have been consumed.  56 packed instructions (32 + 32).
template <int SIGN, class H> KG_DEV void kg_radix16_stage1_cc(const cf (&c)[16], const cf (&d)[16], cf (&x)[16], H hook)
{
    cf u0[4], u1[4];
    kg_cmul4v_o<true>(u0[0], u0[1], u0[2], u0[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]);
    hook(0);
    kg_cmul4v_o<true>(u1[0], u1[1], u1[2], u1[3], c[4], c[5], c[6], c[7], d[4], d[5], d[6], d[7]);
    hook(1);
    cf s02[4], s13[4];
    kg_cfma4v<true>(s02[0], s02[1], s02[2], s02[3], u0[0], u0[1], u0[2], u0[3], c[8], c[9], c[10], c[11], d[8], d[9], d[10], d[11]);
    hook(2);
    kg_cfma4v<true>(s13[0], s13[1], s13[2], s13[3], u1[0], u1[1], u1[2], u1[3], c[12], c[13], c[14], c[15], d[12], d[13], d[14], d[15]);
    hook(3);
    cf d02[4], d13[4];
    kg_2cmt4(d02[0], d02[1], d02[2], d02[3], u0[0], u0[1], u0[2], u0[3], s02[0], s02[1], s02[2], s02[3]);
    kg_2cmt4(d13[0], d13[1], d13[2], d13[3], u1[0], u1[1], u1[2], u1[3], s13[0], s13[1], s13[2], s13[3]);
#pragma unroll
    for (int b = 0; b < 4; b++) {
        x[b] = s02[b] + s13[b];
        x[8 + b] = s02[b] - s13[b];
        kg_addsub_sj<SIGN>(x[4 + b], x[12 + b], d02[b], d13[b]);
    }
}

// twiddle16 + radix16 fused (97 packed instructions instead of 110), hooks as kg_radix16_h: s = 0..3 inside the first stage,
// s = 4..7 when row s - 4 of the outputs (y[s-4], y[s], y[s+4], y[s+8]) is final
template <int SIGN, class H> KG_DEV void kg_tw_radix16_h(cf (&x)[16], cf (&y)[16], const kg_tw15 &w, H hook)
{
    kg_radix16_stage1_tw<SIGN>(x, w, hook);
    kg_radix16_stage2f<SIGN>(x, y, hook);
}
// conj-products + radix16 fused (99 instead of 112; no copies of the operands)
template <int SIGN, class H> KG_DEV void kg_cc_radix16_h(const cf (&c)[16], const cf (&d)[16], cf (&y)[16], H hook)
{
    cf x[16];
    kg_radix16_stage1_cc<SIGN>(c, d, x, hook);
    kg_radix16_stage2f<SIGN>(x, y, hook);
}

// The same butterfly with a hook after each of its eight radix-4 groups (s = 0..3: the first stage,
// s = 4..7: the second, after which y[s-4], y[s], y[s+4], y[s+8] are final).  Callers use the hooks
// to issue memory instructions between the groups (operand loads of the next item, the LDS stores of
// finished outputs), fenced with kg_pin() so that they stay spread through the arithmetic instead of
// queueing as one burst behind it.
#define kg_pin() __builtin_amdgcn_sched_barrier(0)
template <int SIGN, class H> KG_DEV void kg_radix16_h(cf (&x)[16], cf (&y)[16], H hook)
{
#pragma unroll
    for (int b = 0; b < 4; b++) { kg_radix4<SIGN>(x[b], x[4 + b], x[8 + b], x[12 + b]); hook(b); }
#define KG_W16C(K) cf{KG_W16[K][0], KG_W16[K][1]}
    kg_radix4<SIGN>(x[0], x[1], x[2], x[3]);
    y[0] = x[0]; y[4] = x[1]; y[8] = x[2]; y[12] = x[3];
    hook(4);
    kg_cmul4s<(SIGN < 0)>(x[5], x[6], x[7], x[9], KG_W16C(1), KG_W16C(2), KG_W16C(3), KG_W16C(2));
    kg_radix4<SIGN>(x[4], x[5], x[6], x[7]);
    y[1] = x[4]; y[5] = x[5]; y[9] = x[6]; y[13] = x[7];
    hook(5);
    kg_cmul4s<(SIGN < 0)>(x[11], x[13], x[14], x[15], KG_W16C(6), KG_W16C(3), KG_W16C(6), KG_W16C(9));
    kg_radix4<SIGN, true>(x[8], x[9], x[10], x[11]);
    y[2] = x[8]; y[6] = x[9]; y[10] = x[10]; y[14] = x[11];
    hook(6);
    kg_radix4<SIGN>(x[12], x[13], x[14], x[15]);
    y[3] = x[12]; y[7] = x[13]; y[11] = x[14]; y[15] = x[15];
    hook(7);
#undef KG_W16C
}

// Inter-pass twiddles of the 4096-point transform for thread t of 256: pass 1
// needs W256^(j*(t&15)), pass 2 W4096^(j*t), j = 1..15, all held in registers
// (60 VGPRs) for the life of the kernel.  Values come from
// tab4096[k] = exp(+2*pi*i*k/4096) (fp32 roundings of double-precision values,
// host-built, L2 resident).
struct kg_tw4096 { kg_tw15 p1, p2; };

KG_DEV void kg_tw4096_load(kg_tw4096 &tw, const float2 *__restrict__ tab4096, int t)
{
#pragma unroll
    for (int j = 1; j < 16; j++) {
        tw.p1.w[j - 1] = kg_ld(&tab4096[(j * (t & 15)) << 4]);
        tw.p2.w[j - 1] = kg_ld(&tab4096[j * t]);
    }
}

template <int SIGN> KG_DEV void kg_twiddle16(cf (&x)[16], const kg_tw15 &w)
{
    kg_cmul4v<(SIGN < 0)>(x[1], x[2], x[3], x[4], w.w[0], w.w[1], w.w[2], w.w[3]);
    kg_cmul4v<(SIGN < 0)>(x[5], x[6], x[7], x[8], w.w[4], w.w[5], w.w[6], w.w[7]);
    kg_cmul4v<(SIGN < 0)>(x[9], x[10], x[11], x[12], w.w[8], w.w[9], w.w[10], w.w[11]);
    kg_cmul3v<(SIGN < 0)>(x[13], x[14], x[15], w.w[12], w.w[13], w.w[14]);
}

// 4096-point transform of x (thread t holds X[t + 256 j], j = 0..15) by a
// 256-thread group sharing two 4096-element LDS tiles.
// Out: y[m] = sum_k X[k] exp(SIGN*2*pi*i*k*n/4096) at n = t + 256 m.
// Two __syncthreads(): every thread of the workgroup must call it.  Back-to-back
// calls need no barrier in between (tileA is rewritten two barriers after its
// last read).
// Diagnostic stamp (s_memtime, shader cycles): exists only in STAMPS instantiations.
#define KG_STAMP(on, ptr, i)                                                          \
    do {                                                                              \
        if (on) {                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                        \
            if ((ptr) != nullptr) (ptr)[i] = __builtin_amdgcn_s_memtime();            \
            __builtin_amdgcn_sched_barrier(0);                                        \
        }                                                                             \
    } while (0)

// First half: passes 0 and 1, ends after the second barrier.
template <int SIGN, bool STAMPS = false>
KG_DEV void kg_subfft4096_a(cf (&x)[16], cf (&y)[16], float2 *tileA, float2 *tileB,
                            const kg_tw4096 &tw, int t, unsigned long long *st = nullptr)
{
    const int tl = t & 15, th = t >> 4;
    const int rd = t ^ (th & 15);          // P(t + 256 j) = 256 j + (t ^ ((t >> 4) & 15))
    // pass 0 (no twiddles): out index 16 t + m
    kg_radix16<SIGN>(x, y);
    KG_STAMP(STAMPS, st, 0);
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tileA[16 * t + (m ^ tl)], y[m]);
    KG_STAMP(STAMPS, st, 1);
    __syncthreads();
    KG_STAMP(STAMPS, st, 2);
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileA[rd + 256 * j]);
    if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KG_STAMP(STAMPS, st, 3);
    // pass 1: twiddle W256^(j*(t&15)), out index (t>>4)*256 + (t&15) + 16 m
    kg_twiddle16<SIGN>(x, tw.p1);
    kg_radix16<SIGN>(x, y);
    KG_STAMP(STAMPS, st, 4);
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tileB[th * 256 + 16 * m + (tl ^ m)], y[m]);
    __syncthreads();
    KG_STAMP(STAMPS, st, 5);
}

// Second half: pass 2; y[m] is the output at n = t + 256 m.
template <int SIGN, bool STAMPS = false>
KG_DEV void kg_subfft4096_b(cf (&x)[16], cf (&y)[16], const float2 *tileB,
                            const kg_tw4096 &tw, int t, unsigned long long *st = nullptr)
{
    const int rd = t ^ ((t >> 4) & 15);
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileB[rd + 256 * j]);
    if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KG_STAMP(STAMPS, st, 6);
    // pass 2: twiddle W4096^(j*t), out index t + 256 m (kept in registers)
    kg_twiddle16<SIGN>(x, tw.p2);
    kg_radix16<SIGN>(x, y);
    KG_STAMP(STAMPS, st, 7);
}

// kg_subfft4096_a with the LDS stores of each pass issued group by group as the outputs become
// final, and a caller hook h0(s), s = 0..3, between the first-stage groups of pass 0.
template <int SIGN, class H0>
KG_DEV void kg_subfft4096_a_spread(cf (&x)[16], cf (&y)[16], float2 *tileA, float2 *tileB,
                                   const kg_tw4096 &tw, int t, H0 h0)
{
    const int tl = t & 15, th = t >> 4;
    const int rd = t ^ (th & 15);
    kg_radix16_h<SIGN>(x, y, [&](int s) {
        if (s < 4) h0(s);
        else {
            kg_pin();
#pragma unroll
            for (int m = s - 4; m < 16; m += 4) kg_st_tile(&tileA[16 * t + (m ^ tl)], y[m]);
            kg_pin();
        }
    });
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileA[rd + 256 * j]);
    kg_twiddle16<SIGN>(x, tw.p1);
    kg_radix16_h<SIGN>(x, y, [&](int s) {
        if (s >= 4) {
            kg_pin();
#pragma unroll
            for (int m = s - 4; m < 16; m += 4) kg_st_tile(&tileB[th * 256 + 16 * m + (tl ^ m)], y[m]);
            kg_pin();
        }
    });
    __syncthreads();
}

template <int SIGN, bool STAMPS = false>
KG_DEV void kg_subfft4096(cf (&x)[16], cf (&y)[16], float2 *tileA, float2 *tileB,
                          const kg_tw4096 &tw, int t, unsigned long long *st = nullptr)
{
    kg_subfft4096_a<SIGN, STAMPS>(x, y, tileA, tileB, tw, t, st);
    kg_subfft4096_b<SIGN, STAMPS>(x, y, tileB, tw, t, st);
}

// kg_subfft4096 with the pass-1 twiddles W256^(j (t & 15)) read from a 240-entry LDS table
// (tw1[(j - 1) 16 + (t & 15)], filled by kg_tw1_fill) instead of 30 registers: for kernels that need
// those registers for something that saves global loads (kg_wf.hip).
KG_DEV void kg_tw1_fill(float2 *tw1, const float2 *__restrict__ tab4096, int t)
{
    if (t < 240) kg_st(&tw1[t], kg_ld(&tab4096[(((t >> 4) + 1) * (t & 15)) << 4]));
}
// Round 4: the SECOND exchange of these two forms is not swizzled.  Its writer holds (a, c) = (th, tl) in the lane and b = m
// in the register number of element 256 a + 16 b + c, so sixteen contiguous lanes of a ds_write_b64 already hit sixteen
// consecutive slots, and its reader's thirty-two lanes thirty-two consecutive ones: conflict-free as it stands, with ONE address
// register and immediate offsets.  (The first exchange's writer has b in the lane and c in the register number -- sixteen lanes
// on one slot column without the XOR, and an XOR of lane and register number is sixteen address registers.)  Those sixteen
// registers are what the waterfall kernel's persistent window values needed (tools/vgpr_live.py).
#ifndef KG_WF_FUSED_TW
#define KG_WF_FUSED_TW 1     // twiddles fused into the first butterflies (kg_tw_radix16_h), as in the correlators; 0: the A/B reference
#endif
template <int SIGN>
KG_DEV void kg_subfft4096_l(cf (&x)[16], cf (&y)[16], float2 *tileA, float2 *tileB, const float2 *tw1,
                            const kg_tw15 &p2, int t)
{
    const int tl = t & 15, th = t >> 4;
    const int rd = t ^ (th & 15);
    kg_radix16<SIGN>(x, y);
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tileA[16 * t + (m ^ tl)], y[m]);
    __syncthreads();
    kg_tw15 w1;
#pragma unroll
    for (int j = 1; j < 16; j++) w1.w[j - 1] = kg_ld_tile(&tw1[(j - 1) * 16 + tl]);
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileA[rd + 256 * j]);
#if KG_WF_FUSED_TW
    kg_tw_radix16_h<SIGN>(x, y, w1, [](int) {});
#else
    kg_twiddle16<SIGN>(x, w1);
    kg_radix16<SIGN>(x, y);
#endif
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tileB[t + 16 * (15 * th + m)], y[m]);    // 256 th + 16 m + tl: no swizzle needed (see above)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileB[t + 256 * j]);
#if KG_WF_FUSED_TW
    kg_tw_radix16_h<SIGN>(x, y, p2, [](int) {});
#else
    kg_twiddle16<SIGN>(x, p2);
    kg_radix16<SIGN>(x, y);
#endif
}

// kg_subfft4096_l with a caller hook h(k), k = 0..7, at eight points spread over the transform (after the pass-0
// butterfly, after its stores, behind the issued tile reads of each exchange, after each twiddle block, after the
// pass-1 butterfly and its stores): callers issue a few global loads at each (fenced with kg_pin()), so that the
// loads of a frame do not reach the texture addresser as one burst from four waves at once.
template <int SIGN, class H>
KG_DEV void kg_subfft4096_l_h(cf (&x)[16], cf (&y)[16], float2 *tileA, float2 *tileB, const float2 *tw1,
                              const kg_tw15 &p2, int t, H h)
{
    // (spreading the LDS stores of passes 0 and 1 through the butterflies as well, kg_radix16_h, measured the same)
    const int tl = t & 15, th = t >> 4;
    const int rd = t ^ (th & 15);
    kg_radix16<SIGN>(x, y);
    h(0);
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tileA[16 * t + (m ^ tl)], y[m]);
    h(1);
    __syncthreads();
    kg_tw15 w1;
#pragma unroll
    for (int j = 1; j < 16; j++) w1.w[j - 1] = kg_ld_tile(&tw1[(j - 1) * 16 + tl]);
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileA[rd + 256 * j]);
    h(2);
#if KG_WF_FUSED_TW
    kg_tw_radix16_h<SIGN>(x, y, w1, [&](int s) { if (s == 1) h(3); });
#else
    kg_twiddle16<SIGN>(x, w1);
    h(3);
    kg_radix16<SIGN>(x, y);
#endif
    h(4);
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tileB[t + 16 * (15 * th + m)], y[m]);    // 256 th + 16 m + tl: no swizzle needed (see above)
    h(5);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileB[t + 256 * j]);
    h(6);
#if KG_WF_FUSED_TW
    kg_tw_radix16_h<SIGN>(x, y, p2, [&](int s) { if (s == 1) h(7); });
#else
    kg_twiddle16<SIGN>(x, p2);
    h(7);
    kg_radix16<SIGN>(x, y);
#endif
}

// The same transform for kernels that run it once per workgroup (forward FFT of
// a sample block, code-table build): one tile, one rolled radix-16 body, the
// twiddles fetched per pass instead of being held (half the registers).
// Three __syncthreads(); several 256-thread groups of a larger workgroup may
// run it in lockstep, each on its own tile.
template <int SIGN>
KG_DEV void kg_subfft4096_once(cf (&x)[16], cf (&y)[16], float2 *tile,
                               const float2 *__restrict__ tab4096, int t)
{
    const int tl = t & 15, th = t >> 4;
    const int rd = t ^ (th & 15);
#pragma unroll 1
    for (int p = 0; p < 3; p++) {
        if (p > 0) {
            const int e = (p == 1) ? (t & 15) << 4 : t;
            kg_tw15 w;
#pragma unroll
            for (int j = 1; j < 16; j++) w.w[j - 1] = kg_ld(&tab4096[(j * e) & 4095]);
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = kg_ld(&tile[rd + 256 * j]);
            __syncthreads();               // everyone has read before anyone rewrites the tile
            kg_twiddle16<SIGN>(x, w);
        }
        kg_radix16<SIGN>(x, y);
        if (p == 0) {
#pragma unroll
            for (int m = 0; m < 16; m++) kg_st_tile(&tile[16 * t + (m ^ tl)], y[m]);
            __syncthreads();
        } else if (p == 1) {
#pragma unroll
            for (int m = 0; m < 16; m++) kg_st_tile(&tile[th * 256 + 16 * m + (tl ^ m)], y[m]);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// The same 4096-point transform by a 512-thread group: 8 points per thread, four radix-8 passes (Stockham
// autosort), three exchanges.  Half the registers per thread of the 256 x 16 form -- for the kernel that must keep
// four accumulators per output point (the 16368-lag window of E1B, kg_acq.hip) -- at the price of a third
// exchange.  Pass p (Ns = 8^p): thread i reads in[i + 512 j], multiplies by W_{8 Ns}^{j (i mod Ns)}, and writes
// out[(i div Ns) 8 Ns + (i mod Ns) + m Ns].  Each exchange has its own tile and its own swizzle, chosen so that every
// ds_write_b64 (16-lane groups, element index mod 16 distinct) and ds_read_b64 (32-lane groups, mod 32) is
// conflict-free and every read is base + immediate offset (tools/proto_fft8.py checks both exhaustively):
//   exchange 0: P(e) = e ^ ((e >> 4) & 7)      written at 8 i + (m ^ ((i >> 1) & 7)),  read at 512 j + (i ^ ((i >> 4) & 7))
//   exchange 1: P(e) = e ^ (((e >> 6) & 1) << 3)  written at (i >> 3) 64 + (i & 7) + 8 (m ^ ((i >> 3) & 1)),  read at 512 j + (i ^ (((i >> 6) & 1) << 3))
//   exchange 2: identity                        written at (i >> 6) 512 + (i & 63) + 64 m,  read at 512 j + i
// Three tiles (96 KiB): exchange k always uses tile k, rewritten two barriers after its last read.
// ---------------------------------------------------------------------------
template <bool CONJ> KG_DEV void kg_cmul2s(cf &a0, cf &a1, cf w0, cf w1)
{
    cf r0, r1;
    if constexpr (!CONJ)
        asm(KG_MUL_("%2", "%0", "%4") KG_MUL_("%3", "%1", "%5")
            KG_FMA_("%0", "%4", "%2", "neg_lo:[0,1,0]") KG_FMA_("%1", "%5", "%3", "neg_lo:[0,1,0]")
            : "+v"(a0), "+v"(a1), "=&v"(r0), "=&v"(r1) : "s"(w0), "s"(w1));
    else
        asm(KG_MUL_("%2", "%0", "%4") KG_MUL_("%3", "%1", "%5")
            KG_FMA_("%0", "%4", "%2", "neg_hi:[0,1,0]") KG_FMA_("%1", "%5", "%3", "neg_hi:[0,1,0]")
            : "+v"(a0), "+v"(a1), "=&v"(r0), "=&v"(r1) : "s"(w0), "s"(w1));
}

// In: x[j].  Out: y[m] = sum_j x[j] exp(SIGN*2*pi*i*j*m/8).  28 packed instructions.
// j = 2a + b: a radix-4 over a for b = 0, 1 (u_b[c] lands in x[2c + b]), u_1[c] *= W8^c (c = 2: SIGN*j, folded into
// the last adds), then y[c] = u_0[c] + u_1[c], y[c + 4] = u_0[c] - u_1[c].
template <int SIGN> KG_DEV void kg_radix8(cf (&x)[8], cf (&y)[8])
{
    kg_radix4<SIGN>(x[0], x[2], x[4], x[6]);
    kg_radix4<SIGN>(x[1], x[3], x[5], x[7]);
    kg_cmul2s<(SIGN < 0)>(x[3], x[7], cf{KG_W16[2][0], KG_W16[2][1]}, cf{KG_W16[6][0], KG_W16[6][1]});
    y[0] = x[0] + x[1]; y[4] = x[0] - x[1];
    y[1] = x[2] + x[3]; y[5] = x[2] - x[3];
    kg_addsub_sj<SIGN>(y[2], y[6], x[4], x[5]);
    y[3] = x[6] + x[7]; y[7] = x[6] - x[7];
}

struct kg_tw7 { cf w[7]; };
struct kg_tw4096_r8 { kg_tw7 p1, p2, p3; };     // W_64^(j (i & 7)), W_512^(j (i & 63)), W_4096^(j i), j = 1..7

KG_DEV void kg_tw4096_r8_load(kg_tw4096_r8 &tw, const float2 *__restrict__ tab4096, int i)
{
#pragma unroll
    for (int j = 1; j < 8; j++) {
        tw.p1.w[j - 1] = kg_ld(&tab4096[(j * (i & 7)) << 6]);
        tw.p2.w[j - 1] = kg_ld(&tab4096[(j * (i & 63)) << 3]);
        tw.p3.w[j - 1] = kg_ld(&tab4096[j * i]);
    }
}

template <int SIGN> KG_DEV void kg_twiddle8(cf (&x)[8], const kg_tw7 &w)
{
    kg_cmul4v<(SIGN < 0)>(x[1], x[2], x[3], x[4], w.w[0], w.w[1], w.w[2], w.w[3]);
    kg_cmul3v<(SIGN < 0)>(x[5], x[6], x[7], w.w[4], w.w[5], w.w[6]);
}

// Round 4: the radix-8 butterfly with its twiddles fused in (see kg_cfma4v above): 36 packed instructions for twiddle8 +
// radix8 (42), 38 for the eight conjugate products + radix8 (44).
// second stage: a_c = x[2c] (u_0[c]), b_c = x[2c + 1] (u_1[c]): y[c] = a_c + W8^c b_c, y[c + 4] = a_c - W8^c b_c
template <int SIGN> KG_DEV void kg_radix8_stage2f(cf (&x)[8], cf (&y)[8])
{
    const cf two = cf{2.0f, 2.0f};
    const cf w1 = cf{KG_W16[2][0], KG_W16[2][1]}, w3 = cf{KG_W16[6][0], KG_W16[6][1]};
    cf s1, s3, d1, d3;
    if constexpr (SIGN > 0)
        asm(KG_CFMA1_("%0", "%6", "%8", "%4") KG_CFMA1_("%1", "%7", "%9", "%5")
            KG_CFMA2P_("%0", "%6", "%8") KG_CFMA2P_("%1", "%7", "%9")
            KG_2CMT_("%2", "%4", "%10", "%0") KG_2CMT_("%3", "%5", "%10", "%1")
            : "=&v"(s1), "=&v"(s3), "=&v"(d1), "=&v"(d3) : "v"(x[2]), "v"(x[6]), "v"(x[3]), "v"(x[7]), "s"(w1), "s"(w3), "s"(two));
    else
        asm(KG_CFMA1_("%0", "%6", "%8", "%4") KG_CFMA1_("%1", "%7", "%9", "%5")
            KG_CFMA2C_("%0", "%6", "%8") KG_CFMA2C_("%1", "%7", "%9")
            KG_2CMT_("%2", "%4", "%10", "%0") KG_2CMT_("%3", "%5", "%10", "%1")
            : "=&v"(s1), "=&v"(s3), "=&v"(d1), "=&v"(d3) : "v"(x[2]), "v"(x[6]), "v"(x[3]), "v"(x[7]), "s"(w1), "s"(w3), "s"(two));
    y[0] = x[0] + x[1]; y[4] = x[0] - x[1];
    y[1] = s1; y[5] = d1;
    kg_addsub_sj<SIGN>(y[2], y[6], x[4], x[5]);
    y[3] = s3; y[7] = d3;
}
// the finals of the two first-stage radix-4s (b = 0, 1): u_b[c] lands in x[2c + b]
template <int SIGN> KG_DEV void kg_radix8_stage1_finals(cf (&x)[8], const cf (&s)[4], const cf (&dd)[4])
{
    // s = {s02_0, s02_1, s13_0, s13_1}, dd = {d02_0, d02_1, d13_0, d13_1}
#pragma unroll
    for (int b = 0; b < 2; b++) {
        x[b] = s[b] + s[2 + b];
        x[4 + b] = s[b] - s[2 + b];
        kg_addsub_sj<SIGN>(x[2 + b], x[6 + b], dd[b], dd[2 + b]);
    }
}
// twiddle8 + radix8 fused; hook() between the stages
template <int SIGN, class H> KG_DEV void kg_tw_radix8_h(cf (&x)[8], cf (&y)[8], const kg_tw7 &w, H hook)
{
    constexpr bool CJ = SIGN < 0;
    kg_cmul3v<CJ>(x[1], x[2], x[3], w.w[0], w.w[1], w.w[2]);       // u0_1 = w1 X1, u1_0 = w2 X2, u1_1 = w3 X3 (X0: no twiddle)
    cf s[4], dd[4];
    kg_cfma4v<CJ>(s[0], s[1], s[2], s[3], x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], w.w[3], w.w[4], w.w[5], w.w[6]);
    kg_2cmt4(dd[0], dd[1], dd[2], dd[3], x[0], x[1], x[2], x[3], s[0], s[1], s[2], s[3]);
    kg_radix8_stage1_finals<SIGN>(x, s, dd);
    hook();
    kg_radix8_stage2f<SIGN>(x, y);
}
template <int SIGN> KG_DEV void kg_tw_radix8(cf (&x)[8], cf (&y)[8], const kg_tw7 &w) { kg_tw_radix8_h<SIGN>(x, y, w, []() {}); }
// the eight conjugate products c_j conj(d_j) + radix8 fused (c, d intact); hook() when every operand has been consumed
template <int SIGN, class H> KG_DEV void kg_cc_radix8_h(const cf (&c)[8], const cf (&d)[8], cf (&y)[8], H hook)
{
    cf u[4], s[4], dd[4], x[8];
    kg_cmul4v_o<true>(u[0], u[1], u[2], u[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]);
    kg_cfma4v<true>(s[0], s[1], s[2], s[3], u[0], u[1], u[2], u[3], c[4], c[5], c[6], c[7], d[4], d[5], d[6], d[7]);
    hook();
    kg_2cmt4(dd[0], dd[1], dd[2], dd[3], u[0], u[1], u[2], u[3], s[0], s[1], s[2], s[3]);
    kg_radix8_stage1_finals<SIGN>(x, s, dd);
    kg_radix8_stage2f<SIGN>(x, y);
}

// Passes 0..2 with their exchanges: in x (thread i holds X[i + 512 j]); on return x holds the inputs of pass 3
// (tile 2 read back).  Three __syncthreads().  The caller finishes with kg_twiddle8(x, tw.p3); kg_radix8(x, y):
// y[m] is the output at n = i + 512 m.
template <int SIGN>
KG_DEV void kg_subfft4096_r8_a(cf (&x)[8], cf (&y)[8], float2 *tile0, float2 *tile1, float2 *tile2,
                               const kg_tw4096_r8 &tw, int i)
{
    const int c0 = (i >> 1) & 7, b1 = (i >> 3) & 1;
    kg_radix8<SIGN>(x, y);
#pragma unroll
    for (int m = 0; m < 8; m++) kg_st_tile(&tile0[8 * i + (m ^ c0)], y[m]);
    __syncthreads();
    const int r0 = i ^ ((i >> 4) & 7);
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = kg_ld_tile(&tile0[r0 + 512 * j]);
    kg_twiddle8<SIGN>(x, tw.p1);
    kg_radix8<SIGN>(x, y);
    const int w1 = (i >> 3) * 64 + (i & 7);
#pragma unroll
    for (int m = 0; m < 8; m++) kg_st_tile(&tile1[w1 + 8 * (m ^ b1)], y[m]);
    __syncthreads();
    const int r1 = i ^ (((i >> 6) & 1) << 3);
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = kg_ld_tile(&tile1[r1 + 512 * j]);
    kg_twiddle8<SIGN>(x, tw.p2);
    kg_radix8<SIGN>(x, y);
    const int w2 = (i >> 6) * 512 + (i & 63);
#pragma unroll
    for (int m = 0; m < 8; m++) kg_st_tile(&tile2[w2 + 64 * m], y[m]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = kg_ld_tile(&tile2[i + 512 * j]);
}

// ---------------------------------------------------------------------------
// Wave64 all-reduce steps without the LDS crossbar (ds_bpermute): quad_perm for
// lane^1 and lane^2, row_half_mirror / row_mirror for the 8- and 16-lane
// levels (the value is already uniform below that level), v_permlane16_swap /
// v_permlane32_swap (gfx950) across rows and halves.
// kg_xchg<L>(v): the partner's value at butterfly level L (0..5).
// ---------------------------------------------------------------------------
template <int L> KG_DEV unsigned kg_xchg_u(unsigned v)
{
    if constexpr (L == 0) return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0xB1, 0xF, 0xF, true);
    else if constexpr (L == 1) return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0x4E, 0xF, 0xF, true);
    else if constexpr (L == 2) return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0x141, 0xF, 0xF, true);
    else if constexpr (L == 3) return (unsigned) __builtin_amdgcn_mov_dpp((int) v, 0x140, 0xF, 0xF, true);
    else if constexpr (L == 4) {
        // rows (r0,r1,r2,r3) -> .x = (r0,r0,r2,r2), .y = (r1,r1,r3,r3): partner = the other one
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        const bool odd_row = (threadIdx.x >> 4) & 1;
        return odd_row ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        const bool hi = (threadIdx.x >> 5) & 1;
        return hi ? r[0] : r[1];
    }
}
template <int L> KG_DEV float kg_xchg(float v) { return __uint_as_float(kg_xchg_u<L>(__float_as_uint(v))); }
template <int L> KG_DEV int kg_xchg(int v) { return (int) kg_xchg_u<L>((unsigned) v); }

// ---------------------------------------------------------------------------
// Wave64 reductions to a wave-uniform value: four butterfly levels inside each row of 16, then
// row_bcast15 (rows 1, 3 take in the row below) and row_bcast31 (rows 2, 3 take in lanes 0..31): row 3
// holds the result, v_readlane 63 makes it scalar.  Every level is ONE instruction -- the combining
// operation with a DPP operand -- where hipcc emits copy + v_mov_dpp + canonicalise + operation (and a
// v_permlane*_swap butterfly needs a dozen instructions for the two cross-row levels).  A DPP operand
// must have been written at least two wait states earlier: the maximum and the sum are reduced
// together, interleaved, with one s_nop between levels.
// ---------------------------------------------------------------------------
#define KG_DPP_LEVELS_(OP2)                                                              \
    OP2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") OP2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
    OP2("row_half_mirror row_mask:0xf bank_mask:0xf") OP2("row_mirror row_mask:0xf bank_mask:0xf")             \
    OP2("row_bcast:15 row_mask:0xa bank_mask:0xf") OP2("row_bcast:31 row_mask:0xc bank_mask:0xf")
KG_DEV void kg_wave_max_sum(float &mx, float &sm)        // -> both wave-uniform
{
#define KG_OP2_(ctl) "v_max_f32_dpp %0, %0, %0 " ctl "\n\tv_add_f32_dpp %1, %1, %1 " ctl "\n\ts_nop 0\n\t"
    asm("s_nop 1\n\t" KG_DPP_LEVELS_(KG_OP2_) : "+v"(mx), "+v"(sm));
#undef KG_OP2_
    mx = __uint_as_float((unsigned) __builtin_amdgcn_readlane((int) __float_as_uint(mx), 63));
    sm = __uint_as_float((unsigned) __builtin_amdgcn_readlane((int) __float_as_uint(sm), 63));
}
KG_DEV int kg_wave_max(int x)
{
#define KG_OP2_(ctl) "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctl "\n\t"
    asm(KG_DPP_LEVELS_(KG_OP2_) : "+v"(x));
#undef KG_OP2_
    return __builtin_amdgcn_readlane(x, 63);
}
KG_DEV int kg_wave_min(int x)
{
#define KG_OP2_(ctl) "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 " ctl "\n\t"
    asm(KG_DPP_LEVELS_(KG_OP2_) : "+v"(x));
#undef KG_OP2_
    return __builtin_amdgcn_readlane(x, 63);
}
