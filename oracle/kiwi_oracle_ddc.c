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

/* comb widths / input truncations of cic_wf1.vh and cic_rx1_*.vh (see ko_ddc_shape below) */
static const int WF1_COMB_W[5] = {23, 22, 21, 20, 20}, WF1_COMB_D[5] = {5, 1, 1, 1, 0};
static const int RX1_COMB_W[3] = {22, 21, 20}, RX1_COMB_D[3] = {4, 1, 1};

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
                const int *W = WF1_COMB_W, *D = WF1_COMB_D;
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

/* The structural constants of the three pruned CICs, in one place: the code above and below
 * reads them, and tests/test_host_cpu.py compares them with the reference's generated
 * verilog/rx/cic_*.vh when the reference tree is present (that pins register widths,
 * truncations and the output rounding slice; the arithmetic around them stays a restatement).
 * Layout of ko_ddc_shape(): [0] N, [1] input bits, [2] output bits, then N integrator widths,
 * N comb widths, N comb input truncations, [..] out_msb, out_width, round_bit. */
static int hogenauer_acc_bits(int n, int r, int bin)          /* cic_gen.c: Bin + ceil(N log2 R) */
{
    int growth = 0;
    double v = 1.0;
    /* smallest g with 2^g >= R^N */
    double rn = 1.0;
    for (int i = 0; i < n; i++) rn *= (double) r;
    while (v < rn) { v *= 2.0; growth++; }
    return bin + growth;
}

int ko_ddc_shape(int which, int r, int *o)
{
    int k = 0;
    if (which == 0) {                          /* wf1: cic_wf1.vh, N = 5, R up to 8192, 24 -> 16 bits */
        const int acc = hogenauer_acc_bits(5, 8192, 24);
        o[k++] = 5; o[k++] = 24; o[k++] = 16;
        for (int i = 0; i < 4; i++) o[k++] = acc;
        o[k++] = 28;
        for (int i = 0; i < 5; i++) o[k++] = WF1_COMB_W[i];
        for (int i = 0; i < 5; i++) o[k++] = WF1_COMB_D[i];
        o[k++] = 19; o[k++] = 16; o[k++] = 3;
    } else if (which == 1) {                   /* rx1: cic_rx1_*.vh structure for decimation r, N = 3, 22 -> 18 bits */
        const int acc = hogenauer_acc_bits(3, r, 22);
        o[k++] = 3; o[k++] = 22; o[k++] = 18;
        o[k++] = acc; o[k++] = acc; o[k++] = 26;
        for (int i = 0; i < 3; i++) o[k++] = RX1_COMB_W[i];
        for (int i = 0; i < 3; i++) o[k++] = RX1_COMB_D[i];
        o[k++] = 19; o[k++] = 18; o[k++] = 1;
    } else if (which == 3) {                   /* rx2 wide: cic_rx2_20k.vh, N = 5, R = 2, 24 bits, no rounding */
        o[k++] = 5; o[k++] = 18; o[k++] = 24;
        for (int i = 0; i < 5; i++) o[k++] = 24;
        for (int i = 0; i < 5; i++) o[k++] = 24;
        for (int i = 0; i < 5; i++) o[k++] = 0;
        o[k++] = 23; o[k++] = 24; o[k++] = -1;
    } else {                                   /* rx2: cic_rx2_12k.vh, N = 5, R = 3, 18 -> 24 bits, unpruned */
        o[k++] = 5; o[k++] = 18; o[k++] = 24;
        for (int i = 0; i < 5; i++) o[k++] = 26;
        for (int i = 0; i < 5; i++) o[k++] = 26;
        for (int i = 0; i < 5; i++) o[k++] = 0;
        o[k++] = 25; o[k++] = 24; o[k++] = 1;
    }
    return k;
}

/* ======================================================================== */
/* Audio DDC: verilog/rx/rx.v:22-178                                         */
/*   IQ_MIXER (OUT_WIDTH = RX1_BITS = 22)                                    */
/*   -> cic_prune_var "rx1": N = 3, R = RX1_STD_DECIM = 1736 (kiwi.config:104)*/
/*   -> cic_prune_var "rx2": N = 5, R = RX2_STD_DECIM = 3, 18 -> 24 bits      */
/*   -> fir_iq (verilog/rx/fir_iq.sv): 65 taps, 18-bit coefficients, /2      */
/*   -> 3-word output mux {i[15:0]},{q[15:0]},{i[23:16],q[23:16]} (rx.v:172) */
/* rx1 widths for R = 1736 are those verilog/rx/cic_gen.c emits for that R    */
/* (acc 55 bits, integrators 55/55/26 with the third taking */
/* [54 -: 26], combs 22/21/20 dropping 4/1/1 LSBs, out = comb3[19 -: 18] +     */
/* comb3[1]); the checked-in cic_rx1_12k.vh is the stale R = 926 variant of    */
/* the same structure.  rx2: cic_rx2_12k.vh (26 bits throughout, out =        */
/* comb5[25 -: 24] + comb5[1]).  SELF-REFERENTIAL like the waterfall DDC.      */
/* ======================================================================== */

/* fir_iq.sv:91-123, the default (rx4/rx8) table: taps[0..32], symmetric */
const int32_t ko_cicf_taps65[33] = {
    0x00071, 0x3ffae, 0x3ff5b, 0x00029, 0x000f6, 0x0002a, 0x3fea6, 0x3ff32, 0x001aa, 0x001dc, 0x3fe5a,
    0x3fcae, 0x000fb, 0x00503, 0x000a7, 0x3f96f, 0x3fc85, 0x0076b, 0x00793, 0x3f927, 0x3f33f, 0x00401,
    0x01296, 0x00227, 0x3e7b0, 0x3f2dd, 0x01caf, 0x0200e, 0x3e310, 0x3bb4f, 0x00c0e, 0x0aeac, 0x1036e,
};

static inline int32_t mix22(int16_t adc, int16_t dds)     /* iq_mixer.v:43-51, OUT_WIDTH 22 */
{
    const int64_t prod = (int64_t) ((int32_t) adc * 4) * (int64_t) ((int32_t) dds * 8);
    return (int32_t) ((prod >> 13) + ((prod >> 12) & 1));
}

void ko_ddc_rx_reset(ko_ddc_rx_state *s)
{
    memset(s, 0, sizeof *s);
}

/* fir_iq.sv:45-77, the RX_CFG == 3 table (wide: "N=5,R=2,M=1, cutoff at 8 kHz") */
const int32_t ko_cicf_taps65_wide[33] = {
    0x0005f, 0x3ffa4, 0x3ff6c, 0x0003c, 0x000e6, 0x0000a, 0x3feb1, 0x3ff63, 0x001ad, 0x00199, 0x3fe3e,
    0x3fcfe, 0x0013d, 0x004b1, 0x00036, 0x3f9b3, 0x3fd2b, 0x0074d, 0x006b9, 0x3f904, 0x3f444, 0x00481,
    0x01177, 0x00126, 0x3e8ca, 0x3f490, 0x01bd5, 0x01d5a, 0x3e30e, 0x3bf82, 0x00f77, 0x0ac26, 0x0fd54,
};
/* fir_iq.sv:79-88, the RX_CFG == 14 table (17 taps) */
const int32_t ko_cicf_taps17[9] = {
    0x001dd, 0x001dd, 0x001dd, 0x3f290, 0x3ee98, 0x006e8, 0x04205, 0x084ab, 0x0a235,
};

/* The three RX instances kiwi.config / fir_iq.sv build (KO_RX_STD: rx4 / rx8, RX_CFG 4 and 8;
 * KO_RX_WIDE: rx3, RX_CFG 3, the 20.25 kHz mode; KO_RX_14: rx14, RX_CFG 14).  rx1 / rx2 widths are
 * what verilog/rx/cic_gen.c emits for the decimations of kiwi.config:101-105 -- pinned by
 * tests/golden/cic_ref.json, which holds cic_gen.c's own output (tests/test_ref_pins_cpu.py):
 *   std / rx14: rx1 R 1736, acc 55, third integrator [54 -: 26]; rx2 R 3, 26 bits, out [25 -: 24] + [1]
 *   wide:       rx1 R 1543, acc 54, third integrator [53 -: 26]; rx2 R 2, 24 bits, out [23 -: 24] */
typedef struct { int r1, r2, acc1, w2, round2, ntaps; const int32_t *taps; } rx_mode;
static const rx_mode RX_MODES[3] = {
    { 1736, 3, 55, 26, 1, 65, ko_cicf_taps65 },
    { 1543, 2, 54, 24, 0, 65, ko_cicf_taps65_wide },
    { 1736, 3, 55, 26, 1, 17, ko_cicf_taps17 },
};

int ko_ddc_rx_decim(int mode) { return RX_MODES[mode].r1 * RX_MODES[mode].r2 * 2; }

/* One audio channel over n ADC samples.  out: rx_iq_t records (6 bytes each).
 * Returns the number of records written. */
int ko_ddc_rx_mode(ko_ddc_rx_state *s, const int16_t *adc, long n, uint64_t phase_inc, uint8_t *out, int mode)
{
    nco_init();
    const rx_mode *md = &RX_MODES[mode];
    const uint64_t M48 = (1ULL << 48) - 1, MACC = (1ULL << md->acc1) - 1;
    const int R1 = md->r1, R2 = md->r2, SH3 = md->acc1 - 26, W2 = md->w2, NT = md->ntaps;
    int nout = 0;
    for (long t = 0; t < n; t++) {
        const int a = (int) (s->phase >> 35);
        const int32_t m[2] = { mix22(adc[t], nco_cos[a]), mix22(adc[t], nco_sin[a]) };
        s->phase = (s->phase + phase_inc) & M48;
        const int strobe1 = (s->cnt1 == (uint32_t) (R1 - 1));
        s->cnt1 = strobe1 ? 0 : s->cnt1 + 1;
        int32_t c1[2] = {0, 0};
        for (int c = 0; c < 2; c++) {
            /* rx1: integrators acc, acc, 26 bits */
            s->i1[c] = (s->i1[c] + (uint64_t) (int64_t) m[c]) & MACC;
            s->i2[c] = (s->i2[c] + s->i1[c]) & MACC;
            s->i3[c] = (s->i3[c] + (uint32_t) (s->i2[c] >> SH3)) & 0x03FFFFFF;     /* [acc-1 -: 26] */
            if (strobe1) {
                const int *W = RX1_COMB_W, *D = RX1_COMB_D;
                int64_t v = sext(s->i3[c], 26);
                for (int k = 0; k < 3; k++) {
                    const int64_t x = sext((uint64_t) (v >> D[k]), W[k]);
                    const int64_t y = sext((uint64_t) (x - s->comb1_prev[c][k]), W[k]);
                    s->comb1_prev[c][k] = x;
                    v = y;
                }
                c1[c] = (int32_t) sext((uint64_t) ((v >> 2) + ((v >> 1) & 1)), 18);   /* comb3[19 -: 18] + comb3[1] */
            }
        }
        if (!strobe1) continue;
        /* rx2: N = 5, R = R2, W2-bit registers, no pruning */
        const int strobe2 = (s->cnt2 == (uint32_t) (R2 - 1));
        s->cnt2 = strobe2 ? 0 : s->cnt2 + 1;
        int32_t c2[2] = {0, 0};
        for (int c = 0; c < 2; c++) {
            int64_t v = c1[c];
            for (int k = 0; k < 5; k++) { s->j[c][k] = sext((uint64_t) (s->j[c][k] + v), W2); v = s->j[c][k]; }
            if (strobe2) {
                for (int k = 0; k < 5; k++) {
                    const int64_t y = sext((uint64_t) (v - s->comb2_prev[c][k]), W2);
                    s->comb2_prev[c][k] = v;
                    v = y;
                }
                c2[c] = md->round2 ? (int32_t) sext((uint64_t) ((v >> 2) + ((v >> 1) & 1)), 24)   /* comb5[25 -: 24] + comb5[1] */
                                   : (int32_t) sext((uint64_t) v, 24);                             /* comb5[23 -: 24] */
            }
        }
        if (!strobe2) continue;
        /* fir_iq: shift register, NT symmetric taps, 42-bit accumulator, out = acc[41 -: 24],
         * an output on every second input (decim_by_2 starts at 0, fir_iq.sv:125-170) */
        int32_t y[2];
        for (int c = 0; c < 2; c++) {
            memmove(&s->fir_buf[c][1], &s->fir_buf[c][0], sizeof(int32_t) * (NT - 1));
            s->fir_buf[c][0] = c2[c];
            int64_t acc = 0;
            for (int k = 0; k < NT; k++) {
                const int32_t coef = (int32_t) sext((uint64_t) md->taps[k <= (NT - 1) / 2 ? k : NT - 1 - k], 18);
                acc = sext((uint64_t) (acc + (int64_t) s->fir_buf[c][k] * coef), 42);
            }
            y[c] = (int32_t) sext((uint64_t) (acc >> 18), 24);
        }
        const int emit = s->decim_by_2;
        s->decim_by_2 ^= 1;
        if (!emit) continue;
        /* rx.v:172 words -> rx_iq_t {u16 i, u16 q, u8 q3, u8 i3} (rx/data_pump.h:27-30) */
        uint8_t *o = out + 6 * (size_t) nout++;
        o[0] = (uint8_t) y[0]; o[1] = (uint8_t) (y[0] >> 8);
        o[2] = (uint8_t) y[1]; o[3] = (uint8_t) (y[1] >> 8);
        o[4] = (uint8_t) (y[1] >> 16); o[5] = (uint8_t) (y[0] >> 16);
    }
    return nout;
}

int ko_ddc_rx(ko_ddc_rx_state *s, const int16_t *adc, long n, uint64_t phase_inc, uint8_t *out)
{
    return ko_ddc_rx_mode(s, adc, n, phase_inc, out, KO_RX_STD);
}
