// kg_rxddc.hip -- the audio digital down-converter on gfx950.
//
// In the reference this is FPGA fabric, one RX instance per audio channel
// (verilog/rx/rx.v:22-178), clocked at the ADC rate:
//   IQ_MIXER, OUT_WIDTH = RX1_BITS = 22                       (iq_mixer.v)
//   cic_prune_var "rx1": N = 3, R = RX1_STD_DECIM = 1736      (kiwi.config:104)
//        55-bit accumulators, integrators 55/55/26 (the third takes [54 -: 26]),
//        combs 22/21/20 dropping 4/1/1 LSBs, out = comb3[19 -: 18] + comb3[1]
//   cic_prune_var "rx2": N = 5, R = RX2_STD_DECIM = 3, 26 bits, out = comb5[25 -: 24] + comb5[1]
//   fir_iq: 65 symmetric taps, 18-bit coefficients, 42-bit accumulator,
//        out = acc[41 -: 24], every second output kept          (fir_iq.sv)
//   output words {i[15:0]}, {q[15:0]}, {i[23:16], q[23:16]} (rx.v:172) = rx_iq_t
// Total decimation 1736 * 3 * 2 = 10416 (RX_DECIM_4CH, kiwi.config:141).  That is the rx4 / rx8
// instance (KG_RXDDC_STD); the rx3 "wide" instance (1543 * 2 * 2 = 6172, RX_DECIM_3CH) and the rx14
// instance (17-tap CICF) are the same structure with the parameters of rx_mode below.
// The host sets the NCO with CmdSetRXFreq (rx/rx_sound_cmd.cpp:41-51) and reads
// the records with CmdGetRX (rx/data_pump.cpp:101).
//
// Parallelisation: the rx1 integrators run at the ADC rate and its third one is
// pruned (it accumulates integrator2[54 -: 26]), so -- as in kg_ddc.hip -- the
// stream is cut into runs, run-local states are combined by an exact carry scan
// (c1' = c1 + e1, c2' = c2 + L*c1 + e2, modulo 2^64 which keeps the low 55 bits
// exact), the runs are integrated again from their exact states, and a prefix sum
// of run totals gives the pruned integrator.  Everything after rx1 runs at 72 kHz
// and is unpruned linear arithmetic in wrapping registers, i.e. exactly an FIR:
// rx2 is the 11-tap (1 + z + z^2)^5 at stride 3 modulo 2^26, fir_iq the 65-tap
// filter at stride 2 modulo 2^42 -- computed directly per output sample.
#include "kg_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;

#define RX_HIST 256          // rx1 outputs kept from earlier calls (a final output spans 203 of them)
#ifndef RX_LOG_THREADS
#define RX_LOG_THREADS 8     // threads (= runs) per workgroup of a run pass, log2
#endif
#define RX_THREADS (1 << RX_LOG_THREADS)
#define RX_TAB KG_NCO_TAB

// The three RX instances the reference builds (kiwi.config:101-105, fir_iq.sv:39-123); the widths
// are what verilog/rx/cic_gen.c emits for those decimations (tests/golden/cic_ref.json):
//   KG_RXDDC_STD   rx4 / rx8: rx1 R 1736 (acc 55: third integrator takes [54 -: 26]), rx2 R 3 (26 bits,
//                  out = comb5[25 -: 24] + comb5[1]), 65-tap CICF
//   KG_RXDDC_WIDE  rx3: rx1 R 1543 (acc 54: [53 -: 26]), rx2 R 2 (24 bits, out = comb5[23 -: 24], no
//                  rounding), the 65-tap RX_CFG == 3 CICF
//   KG_RXDDC_RX14  rx14: the std CICs, the 17-tap RX_CFG == 14 CICF
struct rx_mode {
    int r1, r2;
    int sh3;          // acc width - 26: what the third rx1 integrator drops
    int w2, round2;   // rx2 register width; 1: out = (v >> 2) + ((v >> 1) & 1), 0: out = v
    int ntaps;        // CICF taps (65 / 17)
    int nbox;         // taps of (1 + z + .. + z^(r2-1))^5 = 5 (r2 - 1) + 1
    int tapset;       // row of c_cicf_taps / c_box
};
static const rx_mode RX_MODES[3] = {
    {1736, 3, 29, 26, 1, 65, 11, 0},
    {1543, 2, 28, 24, 0, 65, 6, 1},
    {1736, 3, 29, 26, 1, 17, 11, 2},
};

// fir_iq.sv:91-123 (default: rx4 / rx8), :45-77 (RX_CFG == 3), :79-88 (RX_CFG == 14): taps[0..(NTAPS-1)/2]
// as 18-bit two's complement
__constant__ int c_cicf_taps[3][33] = {
    {0x00071, 0x3ffae, 0x3ff5b, 0x00029, 0x000f6, 0x0002a, 0x3fea6, 0x3ff32, 0x001aa, 0x001dc, 0x3fe5a,
     0x3fcae, 0x000fb, 0x00503, 0x000a7, 0x3f96f, 0x3fc85, 0x0076b, 0x00793, 0x3f927, 0x3f33f, 0x00401,
     0x01296, 0x00227, 0x3e7b0, 0x3f2dd, 0x01caf, 0x0200e, 0x3e310, 0x3bb4f, 0x00c0e, 0x0aeac, 0x1036e},
    {0x0005f, 0x3ffa4, 0x3ff6c, 0x0003c, 0x000e6, 0x0000a, 0x3feb1, 0x3ff63, 0x001ad, 0x00199, 0x3fe3e,
     0x3fcfe, 0x0013d, 0x004b1, 0x00036, 0x3f9b3, 0x3fd2b, 0x0074d, 0x006b9, 0x3f904, 0x3f444, 0x00481,
     0x01177, 0x00126, 0x3e8ca, 0x3f490, 0x01bd5, 0x01d5a, 0x3e30e, 0x3bf82, 0x00f77, 0x0ac26, 0x0fd54},
    {0x001dd, 0x001dd, 0x001dd, 0x3f290, 0x3ee98, 0x006e8, 0x04205, 0x084ab, 0x0a235},
};
// (1 + z + z^2)^5 and (1 + z)^5
__constant__ int c_box[3][11] = {{1, 5, 15, 30, 45, 51, 45, 30, 15, 5, 1}, {1, 5, 10, 10, 5, 1}, {1, 5, 15, 30, 45, 51, 45, 30, 15, 5, 1}};

struct rx_chan {
    u64 phase, phase_inc;
    u32 cnt1;                 // rx1 decimation counter
    u64 i1[2], i2[2];         // rx1 integrators 1, 2 (55 bits, kept modulo 2^64)
    u32 i3[2];                // rx1 integrator 3 (26 bits)
    u32 hist3[2][3];          // integrator-3 values at the last three strobes (rx1 comb registers)
    u64 n1;                   // rx1 outputs produced since reset
    int active;
};

#define RX_DEV __device__ __forceinline__

RX_DEV int mix22(int adc, int dds)            // iq_mixer.v:43-51 with OUT_WIDTH 22: (m >> 8) + bit 7
{
    const int m = adc * dds;
    return (m + 128) >> 8;                    // = (m >> 8) + bit 7 of m (|m| < 2^30)
}
RX_DEV int sx(int v, int bits) { return (v << (32 - bits)) >> (32 - bits); }
RX_DEV long long sx64(long long v, int bits) { return (v << (64 - bits)) >> (64 - bits); }

RX_DEV u64 rx_shfl_up64(u64 v, int d)
{
    const u32 lo = __shfl_up((u32) v, d), hi = __shfl_up((u32) (v >> 32), d);
    return ((u64) hi << 32) | lo;
}

// passes A and B of rx1.  grid = (ceil(nruns / 256), nlist)
template <bool PASS_B>
__global__ __launch_bounds__(RX_THREADS) void rx1_run_kernel(
    const short *__restrict__ adc, long n, int L, int nruns, const rx_chan *__restrict__ chans,
    const int *__restrict__ chan_list, const u32 *__restrict__ nco,
    u64 *__restrict__ st,                     // [nlist][2 comp][2 integ][nruns]: A out, B in (carried)
    u32 *__restrict__ c0rel, u32 *__restrict__ tau, long max_out, rx_mode md,
    int endref,                               // round 4: states referred to the END of the block, summed in levels (below)
    u64 *__restrict__ wg,                     // endref: [nlist][4][gridDim.x]: A out = the workgroup's totals, B in = what its first run adds
    u32 *__restrict__ wgt)                    // endref, pass B: [nlist][2][gridDim.x] out: the workgroup's integrator-3 total
{
    const u32 RX_R1 = (u32) md.r1;
    const int SH3 = md.sh3;
    __shared__ short tab[RX_TAB];             // sin(a) = tab[a], cos(a) = tab[a + 2048] (kg_rxddc_create; kg_ddc.hip)
    for (int i = threadIdx.x; i < RX_TAB / 2; i += RX_THREADS) ((u32 *) tab)[i] = nco[i];
    __syncthreads();
    const int li = blockIdx.y;
    const rx_chan ch = chans[chan_list[li]];
    const int r = blockIdx.x * RX_THREADS + threadIdx.x;
    const bool active = r < nruns;
    if (!active && !endref) return;               // (the workgroup-level sums of both passes need every thread at their barrier)
    const long s0 = active ? (long) r * L : n, s1 = (s0 + L < n) ? s0 + L : n;
    // the 48-bit accumulator sits in the TOP bits of a 64-bit register: it wraps by itself (no mask per sample)
    u64 ph = (ch.phase + (u64) s0 * ch.phase_inc) << 16;
    const u64 inc16 = ch.phase_inc << 16;
    u64 a1i = 0, a2i = 0, a1q = 0, a2q = 0;
    u64 *base = st + (long) li * 4 * nruns;
    if (PASS_B && active) {
        a1i = base[0 * nruns + r]; a2i = base[1 * nruns + r]; a1q = base[2 * nruns + r]; a2q = base[3 * nruns + r];
        if (endref) {
            // End-referred (kg_ddc.hip, sc_Tinv): what is stored is the sum over the earlier runs of their results advanced
            // to the block's end -- T(len): c1, c2 + len c1 -- plus the carried-in state advanced likewise; the state this run
            // starts from is that sum taken back by its distance to the end.
            const u64 *wb = wg + (long) li * 4 * gridDim.x + blockIdx.x;
            a1i += wb[0 * gridDim.x]; a2i += wb[1 * gridDim.x]; a1q += wb[2 * gridDim.x]; a2q += wb[3 * gridDim.x];
            const u64 back = (u64) (n - s0);
            a2i -= back * a1i; a2q -= back * a1q;
        }
    }
    u32 i3i = 0, i3q = 0;
    const u64 c0 = (u64) ch.cnt1 + (u64) s0;
    u32 k = (u32) (c0 % RX_R1);               // decimation counter at the run start
    long o = (long) (c0 / RX_R1);             // strobes of this call before the run
    u32 *c0i = c0rel + ((long) li * 2 + 0) * max_out, *c0q = c0rel + ((long) li * 2 + 1) * max_out;
    // Round 4 (as in kg_ddc.hip): pass A integrates BIASED mixer outputs, u = mix22 + 2^22 = (m + 128 + 2^30) >> 8 as an
    // unsigned value -- no sign extension into the 64-bit adds; a constant 2^22 leaves 2^22 len and 2^22 len (len + 1) / 2 in
    // the two integrators, taken off once per run (modulo 2^64).  Pass B lets integrator 3 run on in 32 bits and masks it to
    // 26 where it is stored (2^26 divides 2^32).
    auto step = [&](int a) {
        const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
        ph += inc16;
        if (!PASS_B) {
            const u32 ui = ((u32) (a * ec) + 0x40000080u) >> 8, uq = ((u32) (a * es) + 0x40000080u) >> 8;
            a1i += (u64) ui; a2i += a1i;
            a1q += (u64) uq; a2q += a1q;
        } else {
            const long long mi = mix22(a, ec), mq = mix22(a, es);
            a1i += (u64) mi; a2i += a1i;
            a1q += (u64) mq; a2q += a1q;
            i3i += (u32) (a2i >> SH3);                          // integrator2[acc-1 -: 26], masked where stored
            i3q += (u32) (a2q >> SH3);
            if (++k == RX_R1) {
                k = 0;
                c0i[o] = i3i & 0x03FFFFFFu;
                c0q[o] = i3q & 0x03FFFFFFu;
                o++;
            }
        }
    };
    // whole groups of eight samples from one 16-byte load, the ragged end sample by sample
    long t = s0;
    if ((((uintptr_t) (adc + s0)) & 15) == 0) {
        for (; t + 8 <= s1; t += 8) {
            const int4 v = *(const int4 *) (adc + t);
            step((short) v.x); step((short) (v.x >> 16)); step((short) v.y); step((short) (v.y >> 16));
            step((short) v.z); step((short) (v.z >> 16)); step((short) v.w); step((short) (v.w >> 16));
        }
    }
    for (; t < s1; t++) step(adc[t]);
    if (PASS_B) {
        if (!endref) {
            tau[((long) li * 2 + 0) * nruns + r] = i3i & 0x03FFFFFFu;
            tau[((long) li * 2 + 1) * nruns + r] = i3q & 0x03FFFFFFu;
            return;
        }
        // the prefix of the runs' integrator-3 totals in the same levels as the carry states: tau[r] = the sum of the
        // workgroup's runs before r, wgt = the workgroup's total (rx1_tau_wg_kernel: its start value; rx1_comb_kernel adds both)
        __shared__ u32 s_t[2][RX_THREADS / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u32 vi = active ? i3i : 0u, vq = active ? i3q : 0u;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 ai = __shfl_up(vi, d), aq = __shfl_up(vq, d);
            if (lane >= d) { vi += ai; vq += aq; }
        }
        if (lane == 63) { s_t[0][wave] = vi; s_t[1][wave] = vq; }
        u32 xi = __shfl_up(vi, 1), xq = __shfl_up(vq, 1);
        if (lane == 0) { xi = 0; xq = 0; }
        __syncthreads();
        u32 pi = 0, pq = 0;
        for (int w = 0; w < wave; w++) { pi += s_t[0][w]; pq += s_t[1][w]; }
        if (active) {
            tau[((long) li * 2 + 0) * nruns + r] = (pi + xi) & 0x03FFFFFFu;
            tau[((long) li * 2 + 1) * nruns + r] = (pq + xq) & 0x03FFFFFFu;
        }
        if (threadIdx.x == RX_THREADS - 1) {
            wgt[((long) li * 2 + 0) * gridDim.x + blockIdx.x] = (pi + vi) & 0x03FFFFFFu;
            wgt[((long) li * 2 + 1) * gridDim.x + blockIdx.x] = (pq + vq) & 0x03FFFFFFu;
        }
    } else {
        {   // the bias of pass A's inputs, out of the two integrators
            const u64 len = (u64) (s1 - s0), b1 = len << 22, b2 = (len * (len + 1) / 2) << 22;
            a1i -= b1; a2i -= b2; a1q -= b1; a2q -= b2;
        }
        if (!endref) {
            base[0 * nruns + r] = a1i; base[1 * nruns + r] = a2i; base[2 * nruns + r] = a1q; base[3 * nruns + r] = a2q;
            return;
        }
        // Round 4: the carry scan's first two levels ride on pass A (as in kg_ddc.hip).  The run's result advanced to the end of
        // the block, then prefix sums -- plain additions modulo 2^64 -- over the workgroup's 256 consecutive runs: st[r] = the
        // sum of the workgroup's runs before r, wg = the workgroup's total; rx1_scan_wg_kernel does the third level (one
        // wave per (channel, I/Q) over at most 64 totals) where rx1_scan_kernel was 52 us between the passes.
        {
            const u64 fwd = (u64) (n - s1);
            a2i += fwd * a1i; a2q += fwd * a1q;
        }
        if (!active) { a1i = a2i = a1q = a2q = 0; }
        __shared__ u64 s_w[4][RX_THREADS / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        u64 v[4] = {a1i, a2i, a1q, a2q}, x[4];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const u64 p = rx_shfl_up64(v[k], d); if (lane >= d) v[k] += p; }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { x[k] = rx_shfl_up64(v[k], 1); if (lane == 0) x[k] = 0; if (lane == 63) s_w[k][wave] = v[k]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u64 p = 0;
            for (int w = 0; w < wave; w++) p += s_w[k][w];
            if (active) base[(long) k * nruns + r] = p + x[k];
            if (threadIdx.x == RX_THREADS - 1) wg[((long) li * 4 + k) * gridDim.x + blockIdx.x] = p + v[k];
        }
    }
}

// carry scan of (i1, i2) per (channel, comp): one wave each
#define RX_SCAN_WAVES 8
#define RX_SCAN_PER 16                        // runs per lane held in registers (8192 runs per channel)
// Carry scan of the two exact integrators over the runs: one workgroup of eight waves per (channel, I/Q)
// (one wave took 171 us for 128 receivers x 16 384 runs: 256 sequential, uncoalesced steps per lane, twice).
__global__ __launch_bounds__(64 * RX_SCAN_WAVES) void rx1_scan_kernel(u64 *__restrict__ st, long n, int L, int nruns,
                                                                     rx_chan *__restrict__ chans, const int *__restrict__ chan_list)
{
    __shared__ u64 w1[RX_SCAN_WAVES], w2[RX_SCAN_WAVES], wl[RX_SCAN_WAVES];
    const int li = blockIdx.x >> 1, comp = blockIdx.x & 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    rx_chan *ch = chans + chan_list[li];
    u64 *e1 = st + ((long) li * 4 + 2 * comp) * nruns, *e2 = e1 + nruns;
    const int per = (nruns + 64 * RX_SCAN_WAVES - 1) / (64 * RX_SCAN_WAVES);
    const int r0 = gl * per < nruns ? gl * per : nruns, r1 = (r0 + per < nruns) ? r0 + per : nruns;
    auto run_len = [&](int r) -> u64 { const long s0 = (long) r * L; return (u64) ((s0 + L < n ? s0 + L : n) - s0); };
    // A lane's runs are consecutive (their order is the composition's), so its loads are its own lines: with up to
    // RX_SCAN_PER runs per lane they all go out at once into registers and both walks read those (the loop that
    // loaded a run, waited, composed, and loaded the same runs again for the second walk took 168 us of 16 + 16
    // dependent round trips for 128 receivers beside the waterfall's pass A).
    const bool regs = per <= RX_SCAN_PER;
    u64 f1[RX_SCAN_PER], f2[RX_SCAN_PER];
    if (regs) {
#pragma unroll
        for (int k = 0; k < RX_SCAN_PER; k++) {
            const int r = r0 + k;
            f1[k] = r < r1 ? e1[r] : 0ull;
            f2[k] = r < r1 ? e2[r] : 0ull;
        }
    }
    u64 a1 = 0, a2 = 0, len = 0;
    if (regs) {
#pragma unroll
        for (int k = 0; k < RX_SCAN_PER; k++) {
            if (r0 + k < r1) { const u64 l = run_len(r0 + k); a2 = a2 + l * a1 + f2[k]; a1 = a1 + f1[k]; len += l; }
        }
    } else {
        for (int r = r0; r < r1; r++) { const u64 l = run_len(r); a2 = a2 + l * a1 + e2[r]; a1 = a1 + e1[r]; len += l; }
    }
    u64 i1 = a1, i2 = a2, ilen = len;
    for (int d = 1; d < 64; d <<= 1) {
        const u64 p1 = rx_shfl_up64(i1, d), p2 = rx_shfl_up64(i2, d), pl = rx_shfl_up64(ilen, d);
        if (lane >= d) { i2 = p2 + ilen * p1 + i2; i1 = p1 + i1; ilen += pl; }
    }
    if (lane == 63) { w1[wave] = i1; w2[wave] = i2; wl[wave] = ilen; }
    u64 x1 = rx_shfl_up64(i1, 1), x2 = rx_shfl_up64(i2, 1), xl = rx_shfl_up64(ilen, 1);
    if (lane == 0) { x1 = 0; x2 = 0; xl = 0; }
    __syncthreads();
    // the saved state advanced through the earlier waves, then through the earlier lanes of this one
    u64 s1 = ch->i1[comp], s2 = ch->i2[comp];
    __syncthreads();                              // every wave has read the saved state
    for (int w = 0; w < wave; w++) { s2 = s2 + wl[w] * s1 + w2[w]; s1 = s1 + w1[w]; }
    u64 c1 = s1 + x1, c2 = s2 + xl * s1 + x2;
    if (regs) {
#pragma unroll
        for (int k = 0; k < RX_SCAN_PER; k++) {
            const int r = r0 + k;
            if (r < r1) {
                const u64 l = run_len(r);
                e1[r] = c1; e2[r] = c2;
                c2 = c2 + l * c1 + f2[k]; c1 = c1 + f1[k];
            }
        }
    } else {
        for (int r = r0; r < r1; r++) {
            const u64 l = run_len(r), g1 = e1[r], g2 = e2[r];
            e1[r] = c1; e2[r] = c2;
            c2 = c2 + l * c1 + g2; c1 = c1 + g1;
        }
    }
    if (r1 == nruns && r0 < nruns) { ch->i1[comp] = c1; ch->i2[comp] = c2; }
}

// The third level of the end-referred scan: one wave per (channel, I/Q) over its workgroup totals (at most 64: max_runs / 256).
// In: wg[li][2 comp + {0, 1}][w] = totals of workgroup w; out: the same slots = the carried-in state advanced to the end of the
// block + the totals of the workgroups before w; the pair's grand total is the state after the block.
__global__ __launch_bounds__(64) void rx1_scan_wg_kernel(u64 *__restrict__ wg, int gx, long n, rx_chan *__restrict__ chans,
                                                        const int *__restrict__ chan_list)
{
    const int li = blockIdx.x >> 1, comp = blockIdx.x & 1, lane = threadIdx.x;
    rx_chan *ch = chans + chan_list[li];
    u64 *t1 = wg + ((long) li * 4 + 2 * comp) * gx, *t2 = t1 + gx;
    u64 v1 = lane < gx ? t1[lane] : 0ull, v2 = lane < gx ? t2[lane] : 0ull;
    const u64 b1 = ch->i1[comp], b2 = ch->i2[comp] + (u64) n * b1;
    for (int d = 1; d < 64; d <<= 1) {
        const u64 p1 = rx_shfl_up64(v1, d), p2 = rx_shfl_up64(v2, d);
        if (lane >= d) { v1 += p1; v2 += p2; }
    }
    u64 x1 = rx_shfl_up64(v1, 1), x2 = rx_shfl_up64(v2, 1);
    if (lane == 0) { x1 = 0; x2 = 0; }
    if (lane < gx) { t1[lane] = b1 + x1; t2[lane] = b2 + x2; }
    if (lane == 63) { ch->i1[comp] = b1 + v1; ch->i2[comp] = b2 + v2; }
}

// Prefix sum of the runs' integrator-3 totals (mod 2^26): one workgroup per (channel, I/Q), run r = k * 512 + thread
// so that every tile of 512 runs is loaded and stored contiguously (as ddc_wf_scan_tau_kernel, kg_ddc.hip).
#define RX_TAU_TILES 32                       // max_runs = 16384 = 32 tiles of 512
__global__ __launch_bounds__(64 * RX_SCAN_WAVES) void rx1_scan_tau_kernel(u32 *__restrict__ tau, int nruns, rx_chan *__restrict__ chans,
                                                                         const int *__restrict__ chan_list)
{
    __shared__ u32 s_tot[RX_TAU_TILES * RX_SCAN_WAVES];
    __shared__ u32 s_w4[4];
    const int li = blockIdx.x >> 1, comp = blockIdx.x & 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    rx_chan *ch = chans + chan_list[li];
    u32 *tv = tau + ((long) li * 2 + comp) * nruns;
    const int ntile = (nruns + 511) >> 9;
    u32 inc[RX_TAU_TILES], own[RX_TAU_TILES];
#pragma unroll
    for (int k = 0; k < RX_TAU_TILES; k++) {
        const int r = (k << 9) + gl;
        own[k] = (k < ntile && r < nruns) ? tv[r] : 0u;
    }
#pragma unroll
    for (int k = 0; k < RX_TAU_TILES; k++) {
        u32 v = own[k];
        for (int d = 1; d < 64; d <<= 1) { const u32 a = __shfl_up(v, d); if (lane >= d) v += a; }
        inc[k] = v;
        if (lane == 63) s_tot[k * RX_SCAN_WAVES + wave] = v;
    }
    __syncthreads();
    u32 t = 0, tinc = 0;
    if (gl < 256) {
        t = s_tot[gl];
        tinc = t;
        for (int d = 1; d < 64; d <<= 1) { const u32 a = __shfl_up(tinc, d); if (lane >= d) tinc += a; }
        if (lane == 63) s_w4[wave] = tinc;
    }
    __syncthreads();
    if (gl < 256) {
        u32 base = 0;
        for (int w = 0; w < wave; w++) base += s_w4[w];
        s_tot[gl] = base + tinc - t;
    }
    const u32 i3 = ch->i3[comp];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RX_TAU_TILES; k++) {
        const int r = (k << 9) + gl;
        if (k < ntile && r < nruns) tv[r] = (i3 + s_tot[k * RX_SCAN_WAVES + wave] + inc[k] - own[k]) & 0x03FFFFFFu;
    }
    if (gl == 0) {
        u32 total = i3;
        for (int w = 0; w < 4; w++) total += s_w4[w];
        ch->i3[comp] = total & 0x03FFFFFFu;
    }
}

// ... and of the integrator-3 totals: in = the workgroups' totals, out = integrator 3 at each workgroup's first run.
__global__ __launch_bounds__(64) void rx1_tau_wg_kernel(u32 *__restrict__ wgt, int gx, rx_chan *__restrict__ chans, const int *__restrict__ chan_list)
{
    const int li = blockIdx.x >> 1, comp = blockIdx.x & 1, lane = threadIdx.x;
    rx_chan *ch = chans + chan_list[li];
    u32 v = lane < gx ? wgt[(long) blockIdx.x * gx + lane] : 0u;
    const u32 i3 = ch->i3[comp];
    for (int d = 1; d < 64; d <<= 1) { const u32 a = __shfl_up(v, d); if (lane >= d) v += a; }
    u32 x = __shfl_up(v, 1);
    if (lane == 0) x = 0;
    if (lane < gx) wgt[(long) blockIdx.x * gx + lane] = (i3 + x) & 0x03FFFFFFu;
    if (lane == 63) ch->i3[comp] = (i3 + v) & 0x03FFFFFFu;
}

// rx1 combs + rounding -> 18-bit samples appended behind the channel's history.
// one thread per (list entry, output)
__global__ __launch_bounds__(256) void rx1_comb_kernel(
    const u32 *__restrict__ c0rel, const u32 *__restrict__ i3start, int L, int nruns, long max_out,
    const rx_chan *__restrict__ chans, const int *__restrict__ chan_list, const long *__restrict__ nouts,
    const u32 *__restrict__ cnt_before, int *__restrict__ c1buf, long c1_stride, u32 *__restrict__ hist_out,
    int RX_R1,
    const u32 *__restrict__ wgt, int gx)      // end-referred levels: i3start[] is relative to wgt[li][comp][run / 256]; or null
{
    const int li = blockIdx.y;
    const rx_chan *ch = chans + chan_list[li];
    const long nout = nouts[li];
    const long o = (long) blockIdx.x * 256 + threadIdx.x;
    if (o >= nout) return;
    const u32 base = cnt_before[li];
    for (int comp = 0; comp < 2; comp++) {
        int c0[4];
        for (int d = 0; d < 4; d++) {
            const long oo = o - 3 + d;
            u32 v;
            if (oo < 0) v = ch->hist3[comp][3 + oo];
            else {
                const long g = (oo + 1) * RX_R1 - 1 - (long) base;      // sample index of the strobe
                const int run = (int) (g / L);
                const u32 wb = wgt ? wgt[((long) li * 2 + comp) * gx + (run >> RX_LOG_THREADS)] : 0u;
                v = (c0rel[((long) li * 2 + comp) * max_out + oo] + i3start[((long) li * 2 + comp) * nruns + run] + wb) & 0x03FFFFFFu;
            }
            c0[d] = sx((int) v, 26);
            if (o == nout - 1 && d >= 1) hist_out[((long) li * 2 + comp) * 3 + (d - 1)] = v;
        }
        const int W[3] = {22, 21, 20}, D[3] = {4, 1, 1};
        int v[4] = {c0[0], c0[1], c0[2], c0[3]};
        int cnt = 4;
        for (int k = 0; k < 3; k++) {
            int x[4];
            for (int d = 0; d < cnt; d++) x[d] = sx(v[d] >> D[k], W[k]);
            for (int d = 1; d < cnt; d++) v[d - 1] = sx(x[d] - x[d - 1], W[k]);
            cnt--;
        }
        c1buf[((long) chan_list[li] * 2 + comp) * c1_stride + RX_HIST + o] = sx((v[0] >> 2) + ((v[0] >> 1) & 1), 18);
    }
}

// rx2 + fir_iq + record packing: one thread per (list entry, final output), 128 outputs per
// workgroup.  The rx2 outputs the workgroup's final samples span (2 * 128 + ntaps - 1 of them per
// component) are computed once into LDS -- evaluated per final sample, each of the 65 taps recomputed
// its 11-tap rx2 sum: 715 loads where 65 LDS reads do (0.35 -> 0.03 ms per step at 128 receivers).
#define RX2_WG 128
__global__ __launch_bounds__(RX2_WG) void rx2_fir_kernel(
    const int *__restrict__ c1buf, long c1_stride, const int *__restrict__ chan_list,
    const long *__restrict__ n1_before,       // rx1 outputs produced before this call
    const long *__restrict__ q_first, const int *__restrict__ nfinal,
    unsigned short *__restrict__ out, long out_stride /* records */, rx_mode md, int by_chan /* rows of out by channel */)
{
    __shared__ int v2s[2][2 * RX2_WG + 65];
    const int li = blockIdx.y, ch = chan_list[li], t = threadIdx.x;
    const int q0 = blockIdx.x * RX2_WG;
    if (q0 >= nfinal[li]) return;             // the whole workgroup
    const long qabs0 = q_first[li] + q0;      // absolute final-output index of thread 0
    const long nb = n1_before[li];
    const long p_lo = 2 * qabs0 + 1 - (md.ntaps - 1);       // first rx2 output index needed
    const long p_hi = 2 * (q_first[li] + nfinal[li] - 1) + 1;   // last one any emitted sample of this call needs
    const int count = 2 * RX2_WG + md.ntaps - 1;
    for (int idx = t; idx < count; idx += RX2_WG) {
        const long p = p_lo + idx;            // rx2 output index
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {
            const int *c1 = c1buf + ((long) ch * 2 + comp) * c1_stride + RX_HIST;    // c1[j - nb] = absolute rx1 output j
            int v2 = 0;
            if (p >= 0 && p <= p_hi) {        // (beyond p_hi: the tail workgroup's unused slots; rx1 outputs not produced yet)
                int s = 0;                    // rx2: sum h[m] c1[r2 p + r2 - 1 - m] modulo 2^w2, then to 24 bits
                for (int m = 0; m < md.nbox; m++) {
                    const long j = md.r2 * p + md.r2 - 1 - m;
                    if (j >= 0) s += c_box[md.tapset][m] * c1[j - nb];
                }
                s = sx(s, md.w2);
                v2 = md.round2 ? sx((s >> 2) + ((s >> 1) & 1), 24) : sx(s, 24);
            }
            v2s[comp][idx] = v2;
        }
    }
    __syncthreads();
    const int qi = q0 + t;
    if (qi >= nfinal[li]) return;
    int y[2];
    const int half = (md.ntaps - 1) >> 1;
#pragma unroll
    for (int comp = 0; comp < 2; comp++) {
        long long acc = 0;
        for (int k = 0; k < md.ntaps; k++) {  // fir input index 2 q + 1 - k  ->  v2s index 2 t + ntaps - 1 - k
            const int coef = sx(c_cicf_taps[md.tapset][k <= half ? k : md.ntaps - 1 - k], 18);
            acc = sx64(acc + (long long) v2s[comp][2 * t + md.ntaps - 1 - k] * coef, 42);
        }
        y[comp] = sx((int) (acc >> 18), 24);
    }
    unsigned short *o = out + ((long) (by_chan ? ch : li) * out_stride + qi) * 3;
    o[0] = (unsigned short) y[0];
    o[1] = (unsigned short) y[1];
    o[2] = (unsigned short) (((y[1] >> 16) & 0xff) | (((y[0] >> 16) & 0xff) << 8));    // q3 | i3 << 8
}

// keep the last RX_HIST rx1 outputs, update counters / comb history / phase
__global__ __launch_bounds__(RX_HIST) void rx_finish_kernel(rx_chan *__restrict__ chans, const int *__restrict__ chan_list,
                                                           long n, const long *__restrict__ nouts,
                                                           const u32 *__restrict__ hist_new, int *__restrict__ c1buf,
                                                           long c1_stride, int RX_R1)
{
    const int li = blockIdx.x, t = threadIdx.x;
    rx_chan *ch = chans + chan_list[li];
    const long nout = nouts[li];
    for (int comp = 0; comp < 2; comp++) {
        int *b = c1buf + ((long) chan_list[li] * 2 + comp) * c1_stride;
        const int v = b[nout + t];            // the last RX_HIST entries of [history | new]
        __syncthreads();
        b[t] = v;
        __syncthreads();
    }
    if (t == 0) {
        const u64 M48 = (1ull << 48) - 1;
        ch->phase = (ch->phase + (u64) n * ch->phase_inc) & M48;
        ch->cnt1 = (u32) (((u64) ch->cnt1 + (u64) n) % RX_R1);
        ch->n1 += (u64) nout;
        if (nout > 0)
            for (int comp = 0; comp < 2; comp++)
                for (int d = 0; d < 3; d++) ch->hist3[comp][d] = hist_new[((long) li * 2 + comp) * 3 + d];
    }
}

// ---------------------------------------------------------------------------
struct kg_rxddc {
    kg_ctx *ctx;
    rx_mode md;
    int nchan; long max_samples;
    rx_chan *d_chans; std::vector<rx_chan> h;
    u32 *d_nco;
    u64 *d_st; u32 *d_c0rel, *d_tau, *d_hist;
    u64 *d_wg;                                 // [nchan][4][64]: workgroup totals / bases of the end-referred carry scan
    u32 *d_wgt;                                // [nchan][2][64]: likewise for the integrator-3 totals
    int *d_c1buf; long c1_stride;
    int max_runs; long max_out;
    kg_stage_cache pack_cache;                 // the per-call tables of the last push (a steady stream repeats them: no upload)
    std::vector<char> seen;                    // scratch of a push: channels listed so far
};

extern "C" {

int kg_rxddc_create(kg_ctx *ctx, int nchan, size_t max_samples, kg_rxddc **out)
{
    return kg_rxddc_create_mode(ctx, nchan, max_samples, KG_RXDDC_STD, out);
}

int kg_rxddc_decim(kg_rxddc *d) { return d ? d->md.r1 * d->md.r2 * 2 : KG_ERR_INVALID; }

int kg_rxddc_create_mode(kg_ctx *ctx, int nchan, size_t max_samples, int mode, kg_rxddc **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_rxddc_create: out is null");
    *out = nullptr;
    KG_REQUIRE(mode == KG_RXDDC_STD || mode == KG_RXDDC_WIDE || mode == KG_RXDDC_RX14, KG_ERR_INVALID,
               "kg_rxddc_create_mode: mode %d", mode);
    KG_REQUIRE(nchan >= 1 && nchan <= 4096, KG_ERR_INVALID, "kg_rxddc_create: nchan %d", nchan);
    KG_REQUIRE(max_samples >= 64 && max_samples <= ((size_t) 1 << 32), KG_ERR_INVALID, "kg_rxddc_create: max_samples %zu", max_samples);
    kg_rxddc *d = new (std::nothrow) kg_rxddc();
    KG_REQUIRE(d != nullptr, KG_ERR_NOMEM, "kg_rxddc_create: alloc");
    d->ctx = ctx; d->nchan = nchan; d->max_samples = (long) max_samples;
    d->md = RX_MODES[mode];
    const int RX_R1 = d->md.r1;
    d->h.assign(nchan, rx_chan());
    for (auto &c : d->h) memset(&c, 0, sizeof c);
    d->max_runs = (int) ((max_samples + 63) / 64);
    if (d->max_runs > 16384) d->max_runs = 16384;
    d->max_out = (long) (max_samples / RX_R1) + 2;
    d->c1_stride = RX_HIST + d->max_out + RX_HIST;
    KG_HIP(hipMalloc((void **) &d->d_chans, sizeof(rx_chan) * nchan));
    KG_HIP(hipMemset(d->d_chans, 0, sizeof(rx_chan) * nchan));
    KG_HIP(hipMalloc((void **) &d->d_nco, sizeof(short) * RX_TAB));
    KG_HIP(hipMalloc((void **) &d->d_st, sizeof(u64) * 4 * (size_t) nchan * d->max_runs));
    KG_HIP(hipMalloc((void **) &d->d_wg, sizeof(u64) * 4 * (size_t) nchan * 64));
    KG_HIP(hipMalloc((void **) &d->d_wgt, sizeof(u32) * 2 * (size_t) nchan * 64));
    KG_HIP(hipMalloc((void **) &d->d_tau, sizeof(u32) * 2 * (size_t) nchan * d->max_runs));
    KG_HIP(hipMalloc((void **) &d->d_c0rel, sizeof(u32) * 2 * (size_t) nchan * d->max_out));
    KG_HIP(hipMalloc((void **) &d->d_hist, sizeof(u32) * 6 * (size_t) nchan));
    KG_HIP(hipMalloc((void **) &d->d_c1buf, sizeof(int) * 2 * (size_t) nchan * d->c1_stride));
    KG_HIP(hipMemset(d->d_c1buf, 0, sizeof(int) * 2 * (size_t) nchan * d->c1_stride));
    // the NCO table as one 16-bit sine table of 10240 entries: cos(a) = T[a + 2048] (kg_common.h, kg_nco_table_build)
    std::vector<short> tab(RX_TAB);
    kg_nco_table_build(tab.data());
    KG_HIP(hipMemcpy(d->d_nco, tab.data(), sizeof(short) * RX_TAB, hipMemcpyHostToDevice));
    *out = d;
    return KG_OK;
}

void kg_rxddc_destroy(kg_rxddc *d)
{
    if (!d) return;
    (void) hipSetDevice(d->ctx->device);
    (void) hipStreamSynchronize(d->ctx->stream);
    (void) hipFree(d->d_chans); (void) hipFree(d->d_nco); 
    
    (void) hipFree(d->d_st); (void) hipFree(d->d_wg); (void) hipFree(d->d_wgt); (void) hipFree(d->d_tau); (void) hipFree(d->d_c0rel); (void) hipFree(d->d_hist);
    (void) hipFree(d->d_c1buf);
    kg_stage_cache_free(&d->pack_cache);
    delete d;
}

// CmdSetRXFreq (rx_sound_cmd.cpp:41-51: i_phase = round(f / adc_clk * 2^48)).  The FPGA keeps
// its filters running when the frequency changes; so does this: only the increment changes.
int kg_rxddc_set_freq(kg_rxddc *d, int ch, uint64_t phase_inc)
{
    KG_REQUIRE(d != nullptr, KG_ERR_INVALID, "kg_rxddc_set_freq: null argument");
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    KG_REQUIRE(ch >= 0 && ch < d->nchan, KG_ERR_INVALID, "kg_rxddc_set_freq: channel %d (0..%d)", ch, d->nchan - 1);
    hipStream_t st = d->ctx->stream;
    KG_HIP(hipStreamSynchronize(st));
    const u64 inc = phase_inc & ((1ull << 48) - 1);
    const int one = 1;
    d->h[ch].phase_inc = inc; d->h[ch].active = 1;
    KG_HIP(hipMemcpy(&d->d_chans[ch].phase_inc, &inc, sizeof inc, hipMemcpyHostToDevice));
    KG_HIP(hipMemcpy(&d->d_chans[ch].active, &one, sizeof one, hipMemcpyHostToDevice));
    return KG_OK;
}

// power-on state: all registers zero, phase zero
int kg_rxddc_reset(kg_rxddc *d, int ch)
{
    KG_REQUIRE(d != nullptr && ch >= 0 && ch < d->nchan, KG_ERR_INVALID, "kg_rxddc_reset: channel %d", ch);
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    hipStream_t st = d->ctx->stream;
    KG_HIP(hipStreamSynchronize(st));
    rx_chan &c = d->h[ch];
    const u64 inc = c.phase_inc; const int act = c.active;
    memset(&c, 0, sizeof c);
    c.phase_inc = inc; c.active = act;
    KG_HIP(hipMemcpy(d->d_chans + ch, &c, sizeof c, hipMemcpyHostToDevice));
    KG_HIP(hipMemset(d->d_c1buf + (size_t) ch * 2 * d->c1_stride, 0, sizeof(int) * 2 * d->c1_stride));
    return KG_OK;
}

// rx_iq_t records channel ch will produce for the next n ADC samples
long kg_rxddc_outputs(kg_rxddc *d, int ch, size_t n)
{
    if (!d || ch < 0 || ch >= d->nchan || !d->h[ch].active) return KG_ERR_INVALID;
    const rx_chan &c = d->h[ch];
    const u64 RX_R1 = (u64) d->md.r1, two_r2 = 2 * (u64) d->md.r2;
    const u64 n1_after = c.n1 + ((u64) c.cnt1 + (u64) n) / RX_R1;
    // final output q needs fir input 2q + 1, i.e. rx2 output 2q + 1, i.e. rx1 outputs up to r2 (2q + 1) + r2 - 1
    auto finals = [two_r2](u64 n1) -> u64 { return n1 < two_r2 ? 0 : (n1 - two_r2) / two_r2 + 1; };   // #q with 2 r2 q + 2 r2 - 1 < n1
    return (long) (finals(n1_after) - finals(c.n1));
}

int kg_rxddc_push_dev(kg_rxddc *d, const void *d_adc, size_t n, const int32_t *chan_list, int nlist,
                      void *d_out, size_t out_stride, int32_t *nouts)
{
    KG_REQUIRE(d && d_adc && chan_list && d_out, KG_ERR_INVALID, "kg_rxddc_push_dev: null argument");
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    KG_REQUIRE(n >= 1 && (long) n <= d->max_samples, KG_ERR_INVALID, "kg_rxddc_push_dev: n %zu (max %ld)", n, d->max_samples);
    KG_REQUIRE(nlist >= 1 && nlist <= d->nchan, KG_ERR_INVALID, "kg_rxddc_push_dev: nlist %d", nlist);
    KG_REQUIRE(((uintptr_t) d_adc & 1) == 0 && ((uintptr_t) d_out & 1) == 0, KG_ERR_INVALID, "kg_rxddc_push_dev: misaligned pointer");
    const u64 RX_R1 = (u64) d->md.r1, two_r2 = 2 * (u64) d->md.r2;
    auto finals = [two_r2](u64 n1) -> u64 { return n1 < two_r2 ? 0 : (n1 - two_r2) / two_r2 + 1; };
    std::vector<long> h_nouts(nlist), h_n1b(nlist), h_qfirst(nlist);
    std::vector<int> h_nfinal(nlist);
    std::vector<u32> h_cnt(nlist);
    long max_n1 = 0; int max_final = 0;
    d->seen.assign(d->nchan, 0);                  // (a flag per channel: no quadratic search of the list)
    for (int i = 0; i < nlist; i++) {
        const int ch = chan_list[i];
        KG_REQUIRE(ch >= 0 && ch < d->nchan && d->h[ch].active, KG_ERR_STATE, "kg_rxddc_push_dev: channel %d has no frequency set", ch);
        KG_REQUIRE(!d->seen[ch], KG_ERR_INVALID, "kg_rxddc_push_dev: channel %d listed twice", ch);
        d->seen[ch] = 1;
        const rx_chan &c = d->h[ch];
        h_cnt[i] = c.cnt1;
        h_nouts[i] = (long) (((u64) c.cnt1 + (u64) n) / RX_R1);
        h_n1b[i] = (long) c.n1;
        h_qfirst[i] = (long) finals(c.n1);
        h_nfinal[i] = (int) (finals(c.n1 + (u64) h_nouts[i]) - finals(c.n1));
        KG_REQUIRE((size_t) h_nfinal[i] <= out_stride, KG_ERR_INVALID, "kg_rxddc_push_dev: out_stride %zu < %d records", out_stride, h_nfinal[i]);
        if (h_nouts[i] > max_n1) max_n1 = h_nouts[i];
        if (h_nfinal[i] > max_final) max_final = h_nfinal[i];
        if (nouts) nouts[i] = h_nfinal[i];
    }
    int L = 64, target_runs = 8192;
    if (const char *e = kg_tuning_env("KIWIGPU_RXDDC_RUNS")) { const int v = atoi(e); if (v >= 256 && v <= d->max_runs) target_runs = v; }
    while (L < 8192 && (long) ((n + L - 1) / L) > target_runs) L <<= 1;
    const int nruns = (int) ((n + L - 1) / L);
    KG_REQUIRE(nruns <= d->max_runs, KG_ERR_INVALID, "kg_rxddc_push_dev: %d runs > %d", nruns, d->max_runs);
    hipStream_t st = d->ctx->stream;
    // per-call tables through the context's staging ring (no stream synchronisation)
    const int *s_list, *s_nfinal; const long *s_nouts, *s_n1b, *s_qfirst; const u32 *s_cnt;
    {
        std::vector<unsigned char> pack;
        auto put = [&](const void *src, size_t bytes) -> size_t {
            const size_t at = (pack.size() + 15) & ~(size_t) 15;
            pack.resize(at + bytes);
            memcpy(pack.data() + at, src, bytes);
            return at;
        };
        const size_t o_list = put(chan_list, sizeof(int) * nlist), o_nouts = put(h_nouts.data(), sizeof(long) * nlist);
        const size_t o_n1b = put(h_n1b.data(), sizeof(long) * nlist), o_qf = put(h_qfirst.data(), sizeof(long) * nlist);
        const size_t o_nf = put(h_nfinal.data(), sizeof(int) * nlist), o_cnt = put(h_cnt.data(), sizeof(u32) * nlist);
        void *base = nullptr;
        if ((rc = kg_ctx_stage_cached(d->ctx, &d->pack_cache, pack.data(), pack.size(), &base))) return rc;
        const unsigned char *b = (const unsigned char *) base;
        s_list = (const int *) (b + o_list); s_nouts = (const long *) (b + o_nouts); s_n1b = (const long *) (b + o_n1b);
        s_qfirst = (const long *) (b + o_qf); s_nfinal = (const int *) (b + o_nf); s_cnt = (const u32 *) (b + o_cnt);
    }
    KG_PLAN_ONLY(d->ctx);
    const dim3 grid((nruns + RX_THREADS - 1) / RX_THREADS, nlist);
    // end-referred carry states, summed inside pass A's workgroups and by one wave per (channel, I/Q) over the workgroup
    // totals (round 4); KIWIGPU_RXDDC_ENDREF=0: run-start states and rx1_scan_kernel, the A/B reference
    int endref = 1;
    if (const char *e = kg_tuning_env("KIWIGPU_RXDDC_ENDREF")) endref = atoi(e) != 0;
    static_assert(16384 / RX_THREADS <= 64, "rx1_scan_wg_kernel scans one workgroup total per lane");
    hipLaunchKernelGGL(rx1_run_kernel<false>, grid, dim3(RX_THREADS), 0, st, (const short *) d_adc, (long) n, L, nruns,
                       (const rx_chan *) d->d_chans, s_list, (const u32 *) d->d_nco, d->d_st,
                       d->d_c0rel, d->d_tau, d->max_out, d->md, endref, d->d_wg, d->d_wgt);
    KG_HIP(hipGetLastError());
    if (endref)
        hipLaunchKernelGGL(rx1_scan_wg_kernel, dim3(2 * nlist), dim3(64), 0, st, d->d_wg, (int) grid.x, (long) n, d->d_chans, s_list);
    else
        hipLaunchKernelGGL(rx1_scan_kernel, dim3(2 * nlist), dim3(64 * RX_SCAN_WAVES), 0, st, d->d_st, (long) n, L, nruns, d->d_chans,
                           s_list);
    KG_HIP(hipGetLastError());
    hipLaunchKernelGGL(rx1_run_kernel<true>, grid, dim3(RX_THREADS), 0, st, (const short *) d_adc, (long) n, L, nruns,
                       (const rx_chan *) d->d_chans, s_list, (const u32 *) d->d_nco, d->d_st,
                       d->d_c0rel, d->d_tau, d->max_out, d->md, endref, d->d_wg, d->d_wgt);
    KG_HIP(hipGetLastError());
    if (endref)
        hipLaunchKernelGGL(rx1_tau_wg_kernel, dim3(2 * nlist), dim3(64), 0, st, d->d_wgt, (int) grid.x, d->d_chans, s_list);
    else
        hipLaunchKernelGGL(rx1_scan_tau_kernel, dim3(2 * nlist), dim3(64 * RX_SCAN_WAVES), 0, st, d->d_tau, nruns, d->d_chans,
                           s_list);
    KG_HIP(hipGetLastError());
    if (max_n1 > 0) {
        hipLaunchKernelGGL(rx1_comb_kernel, dim3((unsigned) ((max_n1 + 255) / 256), nlist), dim3(256), 0, st,
                           (const u32 *) d->d_c0rel, (const u32 *) d->d_tau, L, nruns, d->max_out,
                           (const rx_chan *) d->d_chans, s_list, s_nouts,
                           s_cnt, d->d_c1buf, d->c1_stride, d->d_hist, d->md.r1,
                           endref ? (const u32 *) d->d_wgt : (const u32 *) nullptr, (int) grid.x);
        KG_HIP(hipGetLastError());
    }
    if (max_final > 0) {
        hipLaunchKernelGGL(rx2_fir_kernel, dim3((max_final + RX2_WG - 1) / RX2_WG, nlist), dim3(RX2_WG), 0, st,
                           (const int *) d->d_c1buf, d->c1_stride, s_list, s_n1b,
                           s_qfirst, s_nfinal, (unsigned short *) d_out, (long) out_stride, d->md, d->ctx->rows_by_chan);
        KG_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(rx_finish_kernel, dim3(nlist), dim3(RX_HIST), 0, st, d->d_chans, s_list, (long) n,
                       s_nouts, (const u32 *) d->d_hist, d->d_c1buf, d->c1_stride, d->md.r1);
    KG_HIP(hipGetLastError());
    for (int i = 0; i < nlist; i++) {
        rx_chan &c = d->h[chan_list[i]];
        c.cnt1 = (u32) (((u64) c.cnt1 + (u64) n) % RX_R1);
        c.n1 += (u64) h_nouts[i];
    }
    return KG_OK;
}

}  // extern "C"
