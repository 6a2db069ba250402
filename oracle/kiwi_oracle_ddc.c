/*
 * kiwi_oracle_ddc.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 3: the waterfall DDC the reference runs in FPGA fabric
 * (verilog/rx/iq_mixer.v, cic_prune_var.v + cic_wf1.vh, waterfall_1cic.v).
 *
 * SELF-REFERENTIAL, FPGA PARITY UNPINNED: the NCO is a closed Xilinx DDS
 * Compiler IP with phase dithering (verilog/ipcore_properties/
 * ipcore_dds_sin_cos_13b_15b_48b.txt) whose table and dither sequence are not
 * in the tree, and no simulator (iverilog/verilator) is in this image.  What
 * IS restated bit for bit from the Verilog: the mixer's product/rounding
 * (iq_mixer.v:43-51), the pruned CIC's register widths, truncations, rounding
 * and R = 1 bypass (cic_wf1.vh, cic_prune_var.v).  Frozen here and documented
 * in DESIGN.md: the sine table (round(16383*cos/sin), 13-bit address = phase
 * bits 47:35, no dither), phase(n) = phase0 + n*inc, and pipeline register
 * delays dropped (pure latency): output k closes on input sample R*k + R-1.
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static int16_t nco_cos[8192], nco_sin[8192];
static int nco_ready;

static void nco_init(void)
{
    if (nco_ready) return;
    for (int a = 0; a < 8192; a++) {
        double ph = 2.0 * M_PI * a / 8192.0;
        nco_cos[a] = (int16_t) lrint(16383.0 * cos(ph));
        nco_sin[a] = (int16_t) lrint(16383.0 * sin(ph));
    }
    nco_ready = 1;
}

void ko_ddc_nco_table(int16_t *cos_tab, int16_t *sin_tab)
{
    nco_init();
    memcpy(cos_tab, nco_cos, sizeof nco_cos);
    memcpy(sin_tab, nco_sin, sizeof nco_sin);
}

/* iq_mixer.v:27-51 with IN_WIDTH 16, OUT_WIDTH 24:
 * mx = adc << 2 (18 b), my = {dds, 3'b0} (18 b), prod = mx * my (36 b),
 * out = {prod[35], prod[33 -: 23]} + prod[10].  |prod| < 2^34, so that is
 * (prod >>> 11) + prod[10] in 24-bit two's complement. */
static inline int32_t mix24(int16_t adc, int16_t dds)
{
    const int64_t prod = (int64_t) ((int32_t) adc * 4) * (int64_t) ((int32_t) dds * 8);
    return (int32_t) ((prod >> 11) + ((prod >> 10) & 1));
}

static inline int64_t sext(uint64_t v, int bits)      /* two's complement field -> int64 */
{
    const uint64_t m = 1ULL << (bits - 1);
    v &= (bits == 64) ? ~0ULL : ((1ULL << bits) - 1);
    return (int64_t) ((v ^ m) - m);
}

void ko_ddc_wf_reset(ko_ddc_wf_state *s)
{
    memset(s, 0, sizeof *s);
}

/* One waterfall DDC channel over n ADC samples.  Returns the number of IQ
 * pairs written to out (i, q interleaved, struct iq_t order).
 * log2r: decimation R = 1 << log2r, 0..13 (cic_prune_var.v decim case list). */
int ko_ddc_wf(ko_ddc_wf_state *s, const int16_t *adc, long n, uint64_t phase_inc, int log2r,
              int16_t *out)
{
    nco_init();
    const uint64_t M48 = (1ULL << 48) - 1;
    const long R = 1L << log2r;
    const int shift = 65 - 5 * log2r;                 /* ACC_WIDTH - ACC_R, cic_prune_var.v:224-247 */
    int nout = 0;
    for (long t = 0; t < n; t++) {
        const int a = (int) (s->phase >> 35);                           /* phase bits 47:35 */
        const int32_t m[2] = { mix24(adc[t], nco_cos[a]), mix24(adc[t], nco_sin[a]) };
        s->phase = (s->phase + phase_inc) & M48;
        s->n++;
        const int strobe = (s->sample_no == (uint32_t) (R - 1));       /* cic_prune_var.v:67-80 */
        s->sample_no = strobe ? 0 : s->sample_no + 1;
        for (int c = 0; c < 2; c++) {
            ko_ddc_cic_state *q = &s->cic[c];
            if (log2r == 0) {                                           /* R == 1 bypass, :289-297 */
                out[2 * nout + c] = (int16_t) (m[c] >> 8);              /* in[23 -: 16] */
                continue;
            }
            /* in <= in_data << (ACC_WIDTH - ACC_R), sign-extended to 89 bits; the
             * integrators wrap, so arithmetic modulo 2^128 keeps the low 89 bits exact */
            u128 in = (u128) (__int128) m[c] << shift;
            u128 I[4];
            for (int k = 0; k < 4; k++) I[k] = ((u128) q->integ[k][1] << 64) | q->integ[k][0];
            I[0] += in; I[1] += I[0]; I[2] += I[1]; I[3] += I[2];       /* cic_wf1.vh integrators 1-4 */
            for (int k = 0; k < 4; k++) { q->integ[k][0] = (uint64_t) I[k]; q->integ[k][1] = (uint64_t) (I[k] >> 64); }
            const uint32_t t28 = (uint32_t) (I[3] >> 61) & 0x0FFFFFFF;  /* [88 -: 28] */
            q->integ5 = (q->integ5 + t28) & 0x0FFFFFFF;                 /* 28-bit integrator 5 */
            if (strobe) {
                /* combs, cic_wf1.vh: widths 23,22,21,20,20; inputs drop 5,1,1,1,0 LSBs */
                static const int W[5] = {23, 22, 21, 20, 20};
                static const int D[5] = {5, 1, 1, 1, 0};
                int64_t v = sext(q->integ5, 28);
                for (int k = 0; k < 5; k++) {
                    const int64_t x = sext((uint64_t) (v >> D[k]), W[k]);   /* in_data of comb k */
                    const int64_t y = sext((uint64_t) (x - q->comb_prev[k]), W[k]);
                    q->comb_prev[k] = x;
                    v = y;
                }
                /* out = comb5[19 -: 16] + comb5[3] */
                out[2 * nout + c] = (int16_t) ((v >> 4) + ((v >> 3) & 1));
            }
        }
        if (strobe || log2r == 0) nout++;
    }
    return nout;
}
