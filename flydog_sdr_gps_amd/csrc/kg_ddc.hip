// kg_ddc.hip -- the waterfall digital down-converter on gfx950.
//
// In the reference this is FPGA fabric, one instance per waterfall channel,
// clocked at the ADC rate:  IQ_MIXER (verilog/rx/iq_mixer.v:5-67: 48-bit phase
// accumulator, 15-bit sin/cos, 18x18 multiply, round to 24 bits)
// -> cic_prune_var "wf1" x2 (verilog/rx/cic_prune_var.v + cic_wf1.vh: 5
// integrators at the ADC rate, 89-bit wrapping accumulators, decimate by
// R = 2^k <= 8192, 5 combs, Hogenauer-pruned widths 89/89/89/89/28 and
// 23/22/21/20/20, rounded to 16 bits; R = 1 bypass) -> 16-bit IQ into the
// 8192-sample buffer the host reads as iq_t (verilog/rx/waterfall_1cic.v:58-142).
// The host programs it with CmdSetWFFreq / CmdSetWFDecim / CmdWFReset
// (rx/rx_waterfall.cpp:466,507,1005).  Here the ADC stream lives in HBM and all
// channels are computed from it.
//
// Bit-exactness target: a sequential restatement of the Verilog structure (the
// tests' CPU model).  Frozen by us because the Xilinx DDS IP is closed: the sine
// table, round(16383 cos/sin) addressed by phase bits 47:35, and no phase dither.
//
// Parallelisation.  The integrators are prefix sums over the whole stream and the
// pruning (the 5th integrator accumulates floor(I4 / 2^61)) makes the output depend
// on the exact wrapped value of I4 at every ADC sample, so time cannot simply be
// cut into independent pieces.  The stream is cut into runs of L samples:
//   A  every run integrates from a zero state  -> local end state e_r (stored 4 x 128 bit)
//   S  carry scan: c_{r+1} = T(L) c_r + e_r, where T(L) advances a state over L
//      zero-input samples (binomial coefficients; validated in tools/ and the
//      tests).  Chunks of runs per (channel, I/Q), one workgroup each: lane-local
//      sequential compose, a 6-step wave scan with the affine combine, the wave and
//      chunk totals folded in log steps, lane-local re-expansion.
//   B  every run integrates again from its exact carried state, feeds the pruned
//      5th integrator and records its value at every decimation strobe (relative
//      to the run start) plus the run total
//   S2 prefix sum of the run totals (mod 2^28)
//   C  per output: absolute I5, the five pruned combs (a 6-tap dependency on
//      earlier outputs only, so fully parallel), rounding, int16 store.
// Nothing above bit 88 of an integrator is ever read, so any modulus 2^k with k >= 89 keeps
// the result exact: the run passes work modulo 2^64 on (state >> shift) where 24 + 5 log2 R
// <= 64, modulo 2^96 (three 32-bit limbs) above that, and the scan modulo 2^96 throughout;
// states are stored 128 bits wide (upper limb zero).
#include <type_traits>
#include "kg_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;

struct u128 { u64 lo, hi; };

#define DDC_DEV __device__ __forceinline__

DDC_DEV u128 mk128(u64 lo, u64 hi) { u128 r; r.lo = lo; r.hi = hi; return r; }
DDC_DEV u128 add128(u128 a, u128 b)
{
    u128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}
DDC_DEV u128 mul128(u128 a, u128 b)          // modulo 2^128
{
    u128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
    return r;
}
// j(j+1)/2 and j(j+1)(j+2)/6 exactly (j < 2^40), as 128-bit values
DDC_DEV u128 binom2(u64 j)
{
    const u64 a = (j & 1) ? j : j / 2, b = (j & 1) ? (j + 1) / 2 : j + 1;
    return mk128(a * b, __umul64hi(a, b));
}
DDC_DEV u128 binom3(u64 j)
{
    // j (j+1) (j+2) < 2^120 is even and a multiple of 3: halve it, then divide by 3 exactly =
    // multiply by the inverse of 3 modulo 2^128 (no 64-bit division: that is a software loop here)
    const u128 ab = mk128(j * (j + 1), __umul64hi(j, j + 1));
    u128 p = mul128(ab, mk128(j + 2, 0));
    p.lo = (p.lo >> 1) | (p.hi << 63); p.hi >>= 1;
    return mul128(p, mk128(0xAAAAAAAAAAAAAAABull, 0xAAAAAAAAAAAAAAAAull));
}

struct ddc_state4 { u128 i[4]; };            // integrators 1..4

// Per-channel persistent state (HBM).
struct ddc_chan {
    // phase and sample_no are REFERENCE values: the NCO accumulator and the decimation counter as they stood when the host
    // last set them (set_wf / set_phase / reset).  A push passes the number of samples pushed since then (`pushed`, the
    // same for every channel of the call) and every kernel derives its own start values from it: no kernel writes these
    // two fields, so kernels of consecutive pushes on different streams (round 4: the deferred output stage) all read
    // consistent values.
    u64 phase;            // 48-bit NCO accumulator at the reference point
    u64 phase_inc;        // CmdSetWFFreq (rx_waterfall.cpp:507)
    int log2r;            // CmdSetWFDecim: R = 1 << log2r
    u32 sample_no;        // decimation counter (cic_prune_var.v:65-80) at the reference point
    ddc_state4 integ[2];  // I, Q integrators 1..4
    u32 integ5[2];        // 28-bit integrator 5
    u32 hist[2][5];       // integrator-5 value at the last five strobes (what the comb registers hold)
    int active;
};

// iq_mixer.v:27-51, IN_WIDTH 16, OUT_WIDTH 24: (prod >>> 11) + prod[10], prod = (adc<<2)*(dds<<3)
// prod = 32 * adc * dds and |adc * dds| < 2^30, so in 32-bit arithmetic: (m >> 6) + bit 5 of m
DDC_DEV int mix24(int adc, int dds)
{
    const int m = adc * dds;
    return (m + 32) >> 6;                     // = (m >> 6) + bit 5 of m, one instruction fewer (|m| < 2^30)
}

#define DDC_THREADS 256
#define DDC_TAB KG_NCO_TAB            // NCO table entries: sin(a) = T[a], cos(a) = T[a + 2048], a < 8192
#ifndef DDC_STAGE_STROBES
#define DDC_STAGE_STROBES 16      // strobes per run collected before a flush: 64 bytes per run and store (8 = half lines: ddc14 +6 %)
#endif
#define DDC_STAGE_ROW (DDC_STAGE_STROBES + 1)    // words per lane row of a staging tile (odd stride: conflict-free)
#define DDC_STAGE_BYTES ((DDC_THREADS / 64) * 2 * 64 * DDC_STAGE_ROW * 4)

// The scan's arithmetic.  Nothing above bit 88 of an integrator is ever read, so the scan keeps a state as 4 x 96
// bits in 32-bit limbs (the stored form stays 128 bits wide, upper limb zero): a product with a coefficient is
// 3 / 5 / 6 multiply-adds of 32 x 32 -> 64 bits (the run length and its binomials have 1-2 / 2 / 3 limbs) where
// the 128 x 128-bit product took some forty-five instructions, and a state crosses lanes in 12 shuffles, not 32.
struct u96 { u32 w[3]; };
struct sc4 { u96 i[4]; };                     // integrators 1..4, mod 2^96
struct sc_coef { u64 L, c2; u96 c3; };        // len, len (len+1) / 2 (exact: len < 2^32, asserted where the table is built), len (len+1) (len+2) / 6 mod 2^96
DDC_DEV u96 u96_zero() { u96 r; r.w[0] = r.w[1] = r.w[2] = 0; return r; }
DDC_DEV u96 u96_of(const u128 &v) { u96 r; r.w[0] = (u32) v.lo; r.w[1] = (u32) (v.lo >> 32); r.w[2] = (u32) v.hi; return r; }
DDC_DEV u128 u128_of(const u96 &v) { return mk128((u64) v.w[0] | ((u64) v.w[1] << 32), (u64) v.w[2]); }
DDC_DEV u96 add96(const u96 &a, const u96 &b)
{
    u96 r; unsigned c0, c1;
    r.w[0] = __builtin_addc(a.w[0], b.w[0], 0u, &c0);
    r.w[1] = __builtin_addc(a.w[1], b.w[1], c0, &c1);
    r.w[2] = a.w[2] + b.w[2] + c1;
    return r;
}
DDC_DEV u96 mul96_64(const u96 &a, u64 b)      // a * b mod 2^96
{
    const u32 b0 = (u32) b, b1 = (u32) (b >> 32);
    const u64 p0 = (u64) a.w[0] * b0;
    const u64 p1 = (u64) a.w[1] * b0 + (p0 >> 32);
    const u64 q0 = (u64) a.w[0] * b1 + (u32) p1;
    u96 r;
    r.w[0] = (u32) p0;
    r.w[1] = (u32) q0;
    r.w[2] = a.w[2] * b0 + a.w[1] * b1 + (u32) (p1 >> 32) + (u32) (q0 >> 32);
    return r;
}
DDC_DEV u96 mul96_96(const u96 &a, const u96 &b)
{
    u96 r = mul96_64(a, (u64) b.w[0] | ((u64) b.w[1] << 32));
    r.w[2] += a.w[0] * b.w[2];
    return r;
}
DDC_DEV sc_coef sc_coef_for(u64 len)
{
    sc_coef c;
    c.L = len; c.c2 = binom2(len).lo; c.c3 = u96_of(binom3(len));
    return c;
}
DDC_DEV sc4 sc_zero() { sc4 r; for (int k = 0; k < 4; k++) r.i[k] = u96_zero(); return r; }
DDC_DEV sc4 sc_of(const ddc_state4 &s) { sc4 r; for (int k = 0; k < 4; k++) r.i[k] = u96_of(s.i[k]); return r; }
DDC_DEV ddc_state4 state_of(const sc4 &s) { ddc_state4 r; for (int k = 0; k < 4; k++) r.i[k] = u128_of(s.i[k]); return r; }
// advance a state over len zero-input samples: the binomial matrix of the four cascaded integrators
DDC_DEV sc4 sc_Tc(const sc_coef &k, const sc4 &s)
{
    sc4 r;
    r.i[0] = s.i[0];
    r.i[1] = add96(s.i[1], mul96_64(s.i[0], k.L));
    r.i[2] = add96(add96(s.i[2], mul96_64(s.i[1], k.L)), mul96_64(s.i[0], k.c2));
    r.i[3] = add96(add96(add96(s.i[3], mul96_64(s.i[2], k.L)), mul96_64(s.i[1], k.c2)), mul96_96(s.i[0], k.c3));
    return r;
}
DDC_DEV sc4 sc_T(u64 len, const sc4 &s) { return sc_Tc(sc_coef_for(len), s); }
// The lengths the scan's log steps advance a state by are the same in every lane but a few: u << k (u = the samples
// of a full lane's runs: steps of the wave scan, then of the fold over the wave totals) and v << m (v = a full
// chunk).  Their coefficients come from the host in the kernel arguments (scalar registers); a lane whose length
// is another one (the ragged end of the last chunk) computes its own -- the binomials are 40 % of an advance.
#define DDC_SCAN_TAB 13
struct sc_tab { u64 len[DDC_SCAN_TAB]; sc_coef c[DDC_SCAN_TAB]; };
DDC_DEV sc4 sc_T_tab(const sc_tab &t, int j, u64 len, const sc4 &s)
{
    if (len == t.len[j]) return sc_Tc(t.c[j], s);
    return sc_T(len, s);
}
DDC_DEV sc4 sc_add(const sc4 &a, const sc4 &b) { sc4 r; for (int k = 0; k < 4; k++) r.i[k] = add96(a.i[k], b.i[k]); return r; }
DDC_DEV sc4 sc_shfl(const sc4 &s, int src)
{
    sc4 r;
    for (int k = 0; k < 4; k++) for (int w = 0; w < 3; w++) r.i[k].w[w] = __shfl(s.i[k].w[w], src);
    return r;
}
DDC_DEV sc4 sc_shfl_up(const sc4 &s, int d)
{
    sc4 r;
    for (int k = 0; k < 4; k++) for (int w = 0; w < 3; w++) r.i[k].w[w] = __shfl_up(s.i[k].w[w], d);
    return r;
}

// Round 4: END-REFERRED states.  T(a) T(b) = T(a + b) for every integer a, b (the binomial matrix of the cascaded
// integrators), so the state a run starts from, sum over the earlier runs j of T(s0 - s1_j) e_j + T(s0) S_0, is
// T(-(n - s0)) [ sum_j T(n - s1_j) e_j + T(n) S_0 ]: pass A advances its zero-state result to the END of the entry's share
// of the block (one advance per run, in the pass that has 17 instructions per sample to hide it in), the carry scan is a
// plain prefix SUM of 96-bit words -- no multiplications, no lengths -- and pass B takes its start state back from the end
// (one inverse advance per run); the state after the block is the sum itself.  A sum can be taken in levels wherever the
// values happen to be: inside pass A's workgroups, then over the workgroup totals (ddc_wf_run_kernel, ddc_wf_scan_wg_kernel)
// -- where the affine scan was a kernel of its own between the passes, 36 us alone and 57 us beside the bypass kernel (a
// chunked prefix-sum kernel in its place measured the same 36: waiting on other workgroups, not the multiplications, was
// its cost).
DDC_DEV u96 sub96(const u96 &a, const u96 &b)
{
    u96 r; unsigned c0, c1;
    r.w[0] = __builtin_subc(a.w[0], b.w[0], 0u, &c0);
    r.w[1] = __builtin_subc(a.w[1], b.w[1], c0, &c1);
    r.w[2] = a.w[2] - b.w[2] - c1;
    return r;
}
// T(-m): the coefficients of T(len) at len = -m are -m, m (m - 1) / 2, -m (m - 1) (m - 2) / 6  (m < 2^32)
DDC_DEV sc4 sc_Tinv(u64 m, const sc4 &s)
{
    if (m == 0) return s;
    const u64 c2 = binom2(m - 1).lo;
    const u96 c3 = m >= 2 ? u96_of(binom3(m - 2)) : u96_zero();
    sc4 r;
    r.i[0] = s.i[0];
    r.i[1] = sub96(s.i[1], mul96_64(s.i[0], m));
    r.i[2] = add96(sub96(s.i[2], mul96_64(s.i[1], m)), mul96_64(s.i[0], c2));
    r.i[3] = sub96(add96(sub96(s.i[3], mul96_64(s.i[2], m)), mul96_64(s.i[1], c2)), mul96_96(s.i[0], c3));
    return r;
}
DDC_DEV sc4 sc_Tinv_c(u64 m, u64 c2, const u96 &c3, const sc4 &s)
{
    sc4 r;
    r.i[0] = s.i[0];
    r.i[1] = sub96(s.i[1], mul96_64(s.i[0], m));
    r.i[2] = add96(sub96(s.i[2], mul96_64(s.i[1], m)), mul96_64(s.i[0], c2));
    r.i[3] = sub96(add96(sub96(s.i[3], mul96_64(s.i[2], m)), mul96_64(s.i[1], c2)), mul96_96(s.i[0], c3));
    return r;
}
// The distances to the end are the same for every channel that takes the whole block: n - k L, k = 0 .. nruns (run r ends
// at k = r + 1, starts at k = r).  Their coefficients -- 40 % of an advance, 128-bit products -- come from a table built
// once per (n, L) (ddc_endco_kernel; a steady stream of equal blocks builds it once); an entry with another share of the
// block (a capture) computes its own.
struct ddc_endco { u64 m, c2, c2i; u96 c3, c3i; u32 pad[2]; };        // T(m): m, c2, c3;  T(-m): -m, c2i, -c3i
DDC_DEV ddc_state4 ddc_to_end(const ddc_state4 &e, u64 len, const ddc_endco *__restrict__ co)
{
    if (co) {
        sc_coef k; k.L = co->m; k.c2 = co->c2; k.c3 = co->c3;
        return state_of(sc_Tc(k, sc_of(e)));
    }
    return state_of(sc_T(len, sc_of(e)));
}
DDC_DEV ddc_state4 ddc_from_end(const ddc_state4 &e, u64 len, const ddc_endco *__restrict__ co)
{
    if (co) return state_of(sc_Tinv_c(co->m, co->c2i, co->c3i, sc_of(e)));
    return state_of(sc_Tinv(len, sc_of(e)));
}
__global__ void ddc_endco_kernel(ddc_endco *__restrict__ tab, long n, int L, int nruns)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nruns) return;
    const long at = (long) k * L;
    const u64 m = at < n ? (u64) (n - at) : 0ull;
    ddc_endco c;
    c.m = m; c.c2 = binom2(m).lo; c.c3 = u96_of(binom3(m));
    c.c2i = m >= 1 ? binom2(m - 1).lo : 0ull;
    c.c3i = m >= 2 ? u96_of(binom3(m - 2)) : u96_zero();
    c.pad[0] = c.pad[1] = 0;
    tab[k] = c;
}
// inclusive prefix sum over lanes 0 .. 2^LOG - 1 (every lane of the group calls it)
template <int LOG> DDC_DEV void sc_scan_add(sc4 &e, int lane)
{
#pragma unroll
    for (int d = 1; d < (1 << LOG); d <<= 1) {
        const sc4 a = sc_shfl_up(e, d);
        if (lane >= d) e = sc_add(a, e);
    }
}

DDC_DEV ddc_state4 ddc_add_state(const ddc_state4 &a, const ddc_state4 &b) { return state_of(sc_add(sc_of(a), sc_of(b))); }
// Passes A and B.  grid = (ceil(nruns / 256), nchan).  The body of a thread's run; the kernel below adds pass A's
// workgroup-level prefix sum of the end-referred results (endref == 2).
// MODE (round 5): which of the body's forms an instantiation carries.  The register budget of a kernel is that of its
// widest form -- with everything in one function 119 / 124 registers, four waves per SIMD, for passes that stall on
// table reads and dependent adds and gain from every further wave -- so the launches are split by what their entries need:
//   DDC_ALL     every form (the staged and vector-store forms of R <= 8, the 128-bit pass A of runs longer than 1024)
//   DDC_NARROW  the 64-bit form alone: pass A of any decimation in runs of at most 1024 samples, pass B of 16 <= R <= 256
//   DDC_WIDE    pass B of R >= 512 alone (96-bit integrators)
enum { DDC_ALL = 0, DDC_NARROW = 1, DDC_WIDE = 2 };
#ifdef KG_DDC_ONE_FORM           // A/B build: every launch carries every form (round 4's shape)
static const bool DDC_ONE_FORM = true;
#else
static const bool DDC_ONE_FORM = false;
#endif
template <bool PASS_B, int MODE>
DDC_DEV void ddc_wf_run_body(
    const short *__restrict__ adc, long n, int L, int nruns,
    const ddc_chan *__restrict__ chans, const int *__restrict__ chan_list,
    const u32 *__restrict__ nco,              // the 16-bit table T[DDC_TAB] (kg_ddc_create), two entries per word
    ddc_state4 *__restrict__ local,           // A: out [nlist][2][nruns];  B: in = carried states
    u32 *__restrict__ c0rel,                  // B: relative I5 at strobes; entry li: I at c0off[li], Q right after
    u32 *__restrict__ tau,                    // B: [nlist][2][nruns]
    const long *__restrict__ c0off, const long *__restrict__ nouts,
    const int *__restrict__ sel,              // list entries this launch covers (null: all, in order)
    int stage_bytes,                          // pass B, R <= 8: dynamic LDS for the strobe staging tiles, else 0
    u64 pushed,                               // samples pushed since the call's reference point: the entry's own count is
    const u64 *__restrict__ pdelta,           //   pushed + pdelta[li] (channels retuned or joined at different times differ in age)
    const long *__restrict__ nlim,            // [nlist] samples of the block this entry consumes (capture: 8192 R; else n)
    const int *__restrict__ reset_tab,        // [nlist] capture: the decimation counter starts the block at zero (rst_wf_samp_wr)
    int endref,                               // states in `local` are referred to the END of the entry's share (sc_Tinv above)
    const ddc_endco *__restrict__ endco, long endco_n,    // [nruns + 1] coefficients of the distances n - k L for a share of endco_n samples
    const ddc_state4 *__restrict__ wgbase,    // endref == 2, pass B: [nlist][2][gridDim.x] what the workgroup's first run adds to local[]
    ddc_state4 &outI, ddc_state4 &outQ, bool &have,       // endref == 2, pass A: the run's end-referred result goes to the caller instead of local[]
    u32 &outTi, u32 &outTq)                                // endref == 2, pass B: the run's integrator-5 totals likewise (instead of tau[])
{
    __shared__ short tab[DDC_TAB];
    extern __shared__ u32 stage_lds[];        // [waves][2][64][DDC_STAGE_ROW] when stage_bytes != 0
    for (int i = threadIdx.x; i < DDC_TAB / 2; i += DDC_THREADS) ((u32 *) tab)[i] = nco[i];
    __syncthreads();
    const int li = sel ? sel[blockIdx.y] : (int) blockIdx.y;
    const ddc_chan ch = chans[chan_list[li]];
    const int reset_first = reset_tab[li];
    pushed += pdelta[li];
    const int r = blockIdx.x * DDC_THREADS + threadIdx.x;
    n = nlim[li];                             // this entry's share of the block (wave-uniform)
    if (r >= nruns || (long) r * L >= n) return;
    const long s0 = (long) r * L, s1 = (s0 + L < n) ? s0 + L : n;
    const ddc_endco *co = (endref && n == endco_n) ? endco + (PASS_B ? r : r + 1) : nullptr;
    const long wbI = ((long) li * 2 + 0) * gridDim.x + blockIdx.x, wbQ = ((long) li * 2 + 1) * gridDim.x + blockIdx.x;
    (void) wbI; (void) wbQ;
    // the 48-bit accumulator sits in the TOP bits of a 64-bit register: it wraps by itself (no mask per sample)
    u64 ph = (ch.phase + (pushed + (u64) s0) * ch.phase_inc) << 16;
    const u64 inc16 = ch.phase_inc << 16;
    const int log2r = ch.log2r;
    const u64 Rm1 = (1ull << log2r) - 1;
    const int shift = 65 - 5 * log2r;         // cic_prune_var.v:224-247

    if (log2r == 0) return;                   // R == 1 bypass: ddc_wf_bypass_kernel

    const u64 cnt_call = reset_first ? 0ull : (((u64) ch.sample_no + pushed) & Rm1);   // the decimation counter at the start of this call
    const u64 cnt0 = cnt_call + (u64) s0;                // samples since the counter was last zero
    const long lI = ((long) li * 2 + 0) * nruns + r, lQ = ((long) li * 2 + 1) * nruns + r;
    u32 i5i = 0, i5q = 0;
    u32 c = (u32) (cnt0 & Rm1);                           // decimation counter, sample_no
    long o = (long) (cnt0 >> log2r);                      // index of the next strobe's output
    u32 *c0i = c0rel + c0off[li], *c0q = c0i + ((nouts[li] + 3) & ~3l);      // planes start 16-byte aligned
    // eight samples per round from one 16-byte load (runs start 128-byte aligned relative to
    // the block; an unaligned block or the ragged end of the last run go sample by sample).
    // The load runs ONE GROUP AHEAD of the arithmetic: issued at the top of a group and consumed at the
    // top of the next, its L2 latency hides behind the ~170 instructions of the group in between (a
    // load consumed where it is issued stalled every group of every lane for the whole round trip).
    // The last whole group of a run re-reads itself: one load site, never skipped.
    const bool al = (((uintptr_t) (adc + s0)) & 15) == 0;          // L is a multiple of 8: every group of the run is aligned alike
    const long g1 = al ? s1 : s0;                                  // the group loops run to g1: a misaligned block goes sample by sample
    // Round 4: TWO groups ahead.  With the ADC stream coming from HBM (a block larger than the Infinity Cache, or one the
    // DMA engine has just written) the first touch of a 128-byte line -- one in eight of a lane's loads -- takes longer than
    // the one group of arithmetic the load used to run ahead of: the bench's rotation of nine 32 MiB blocks cost ddc14
    // 0.44 -> 0.47 ms.  KG_DDC_AHEAD=1 restores the single look-ahead.
#ifndef KG_DDC_AHEAD
#define KG_DDC_AHEAD 2
#endif
    const long last_grp = g1 - 8;                                  // start of the run's last whole group (callers check s0 + 8 <= g1)
    int4 ahead = make_int4(0, 0, 0, 0), ahead2 = make_int4(0, 0, 0, 0);
    if (s0 + 8 <= g1) {
        ahead = *(const int4 *) (adc + s0);
        if (KG_DDC_AHEAD == 2) ahead2 = *(const int4 *) (adc + (s0 + 8 <= last_grp ? s0 + 8 : last_grp));
    }
    auto samples8 = [&](long t, short (&buf)[8]) {                 // callers guarantee t + 8 <= g1 and walk t in steps of 8 from s0
        const int4 v = ahead;
        if (KG_DDC_AHEAD == 2) {
            ahead = ahead2;
            const long tn = t + 16 <= last_grp ? t + 16 : last_grp;     // (the last groups re-read the run's last one: one load site, never skipped)
            ahead2 = *(const int4 *) (adc + tn);
        } else {
            const long tn = (t + 16 <= g1) ? t + 8 : t;
            ahead = *(const int4 *) (adc + tn);
        }
        buf[0] = (short) v.x; buf[1] = (short) (v.x >> 16); buf[2] = (short) v.y; buf[3] = (short) (v.y >> 16);
        buf[4] = (short) v.z; buf[5] = (short) (v.z >> 16); buf[6] = (short) v.w; buf[7] = (short) (v.w >> 16);
    };

    // Pass A always starts from zero: over a run of at most 1024 samples the integrators hold plain
    // sums below 2^23 * 1024^4 / 24 < 2^59, so 64 bits are exact for every R (stored sign-extended).
    const bool short_zero_run = !PASS_B && L <= 1024;
    if (MODE == DDC_NARROW || (MODE == DDC_ALL && (log2r <= 8 || short_zero_run))) {
        // Narrow path.  The integrator inputs are m << shift, so bits [shift-1:0] of every
        // integrator stay zero, and nothing above bit 88 is ever read (integrator 5 takes
        // [88 -: 28]): for R <= 256 the live bits [88:shift] are 24 + 5 log2 R <= 64 bits, kept
        // here as (state >> shift) mod 2^64.  Same values as the 128-bit path, half the adds.
        const int sh5 = 5 * log2r - 4;                    // 61 - shift
        u64 I[4], Q[4];
        if (PASS_B) {
            ddc_state4 a = local[lI], b = local[lQ];
            if (endref == 2) { a = ddc_add_state(a, wgbase[wbI]); b = ddc_add_state(b, wgbase[wbQ]); }
            if (endref) { a = ddc_from_end(a, (u64) (n - s0), co); b = ddc_from_end(b, (u64) (n - s0), co); }
            for (int k = 0; k < 4; k++) {
                I[k] = (a.i[k].lo >> shift) | (a.i[k].hi << (64 - shift));
                Q[k] = (b.i[k].lo >> shift) | (b.i[k].hi << (64 - shift));
            }
        } else {
            for (int k = 0; k < 4; k++) I[k] = Q[k] = 0;
        }
        // Round 4, two instructions per sample and channel off each pass:
        //  pass A integrates BIASED mixer outputs, u = mix24 + 2^24 = (m + 32 + 2^30) >> 6 as an unsigned value (|m| < 2^30):
        //  no sign extension into the 64-bit adds; what a constant 2^24 leaves in the four integrators after len samples
        //  -- 2^24 x C(len + k - 1, k) -- is taken off once per run (everything is modulo 2^64 either way);
        //  pass B lets integrator 5 run on in 32 bits and masks it to 28 where it is stored (2^28 divides 2^32).
        auto step = [&](int a) {
            const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
            ph += inc16;
            if (!PASS_B) {
                const u32 ui = ((u32) (a * ec) + 0x40000020u) >> 6, uq = ((u32) (a * es) + 0x40000020u) >> 6;
                I[0] += (u64) ui; I[1] += I[0]; I[2] += I[1]; I[3] += I[2];
                Q[0] += (u64) uq; Q[1] += Q[0]; Q[2] += Q[1]; Q[3] += Q[2];
            } else {
                const long long mi = mix24(a, ec), mq = mix24(a, es);
                I[0] += (u64) mi; I[1] += I[0]; I[2] += I[1]; I[3] += I[2];
                Q[0] += (u64) mq; Q[1] += Q[0]; Q[2] += Q[1]; Q[3] += Q[2];
                i5i += (u32) (I[3] >> sh5);                           // integrator4[88 -: 28] (cic_wf1.vh), masked where stored
                i5q += (u32) (Q[3] >> sh5);
                c = (c + 1) & (u32) Rm1;
                if (c == 0) { c0i[o] = i5i & 0x0FFFFFFFu; c0q[o] = i5q & 0x0FFFFFFFu; o++; }      // strobe: sample_no was R - 1
            }
        };
        long t = s0;
        // R = 2 and R = 4 strobe four (two) times per group of eight samples: when the group is
        // aligned to the decimation counter and the output index to the store width, the strobes
        // of a group leave as ONE 16-byte (8-byte) store per component instead of scattered
        // 4-byte ones (every lane works on a different run, so each store is its own transaction:
        // at R = 2 the scalar stores made pass B 13 times slower than pass A).
        // R <= 8, pass B: every lane of a wave works on a different run, so a strobe store touches 64
        // different lines.  With staging tiles the wave collects 16 strobes per run in LDS
        // (lane rows of 17 words: conflict-free) and flushes them transposed: four lanes write
        // one run's 64 bytes, a store instruction writes 16 whole lines.  Needs the regular case:
        // a full wave of full runs aligned to the decimation, 16 | strobes per run, 16-byte
        // aligned staging rows; anything else takes the paths below.
        if (MODE == DDC_ALL && PASS_B && stage_bytes && log2r <= 3) {
            const int lane = threadIdx.x & 63, K = L >> log2r;
            const bool ok = al && c == 0 && (s1 - s0) == L && (K & (DDC_STAGE_STROBES - 1)) == 0 && (o & 3) == 0 &&
                            ((uintptr_t) c0i & 15) == 0 && ((uintptr_t) c0q & 15) == 0;
            if (__popcll(__ballot(ok)) == 64) {
                u32 *tI = stage_lds + (threadIdx.x >> 6) * (2 * 64 * DDC_STAGE_ROW), *tQ = tI + 64 * DDC_STAGE_ROW;
                // output index of the wave's first run (lane 0); lane l's run starts l * K later
                const long o_wave = ((long) __shfl((int) (o >> 32), 0) << 32) | (unsigned) __shfl((int) o, 0);
                int kcol = 0;
                long kdone = 0;                                           // strobes of this run already flushed
                auto flush = [&]() {                                      // 16 strobes of 64 runs -> 2 x 4 stores of whole lines
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    constexpr int LPR = DDC_STAGE_STROBES / 4, RPS = 64 / LPR;   // lanes per run, runs per store instruction
#pragma unroll
                    for (int q = 0; q < 64 / RPS; q++) {
                        const int rl = q * RPS + lane / LPR, col = (lane % LPR) * 4;
                        const u32 *ri = tI + rl * DDC_STAGE_ROW + col, *rq = tQ + rl * DDC_STAGE_ROW + col;
                        const uint4 vi = make_uint4(ri[0], ri[1], ri[2], ri[3]);
                        const uint4 vq = make_uint4(rq[0], rq[1], rq[2], rq[3]);
                        const long dst = o_wave + (long) rl * K + kdone + col;
                        *(uint4 *) (c0i + dst) = vi;
                        *(uint4 *) (c0q + dst) = vq;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    kdone += DDC_STAGE_STROBES;
                    kcol = 0;
                };
                // One copy of the loop per decimation: the strobe positions of a group are compile-time, so the group
                // is straight-line code -- the eight table reads go out together at its top (with a branch per sample
                // each read was waited for where it was issued: a lone wave per SIMD has nothing to hide that behind).
                // A full tile is flushed at the TOP of the next group, before that group's sample load is issued: the
                // load is then always the youngest vector-memory operation, and waiting for it does not wait for stores
                // issued a moment ago.
                auto staged_loop = [&](auto lr_tag) {
                    constexpr int LR = decltype(lr_tag)::value;
                    for (; t + 8 <= g1; t += 8) {
                        if (kcol == DDC_STAGE_STROBES) flush();           // wave-uniform
                        short buf[8];
                        samples8(t, buf);
                        int ec[8], es[8];
#pragma unroll
                        for (int w = 0; w < 8; w++) { ec[w] = KG_NCO_COS(tab, ph); es[w] = KG_NCO_SIN(tab, ph); ph += inc16; }
#pragma unroll
                        for (int w = 0; w < 8; w++) {
                            const long long mi = mix24(buf[w], ec[w]), mq = mix24(buf[w], es[w]);
                            I[0] += (u64) mi; I[1] += I[0]; I[2] += I[1]; I[3] += I[2];
                            Q[0] += (u64) mq; Q[1] += Q[0]; Q[2] += Q[1]; Q[3] += Q[2];
                            i5i += (u32) (I[3] >> sh5);
                            i5q += (u32) (Q[3] >> sh5);
                            if (((w + 1) & ((1 << LR) - 1)) == 0) {       // strobe (the group starts aligned)
                                tI[lane * DDC_STAGE_ROW + kcol + ((w + 1) >> LR) - 1] = i5i & 0x0FFFFFFFu;
                                tQ[lane * DDC_STAGE_ROW + kcol + ((w + 1) >> LR) - 1] = i5q & 0x0FFFFFFFu;
                            }
                        }
                        kcol += 8 >> LR;
                    }
                    if (kcol == DDC_STAGE_STROBES) flush();
                };
                if (log2r == 1) staged_loop(std::integral_constant<int, 1>());
                else if (log2r == 2) staged_loop(std::integral_constant<int, 2>());
                else staged_loop(std::integral_constant<int, 3>());
                o += K;                                                    // all K strobes of the run are out
            }
        }
        if (MODE == DDC_ALL && PASS_B && log2r <= 2 && c == 0 && (o & 3) == 0 && ((uintptr_t) c0i & 15) == 0 && ((uintptr_t) c0q & 15) == 0) {
            auto quiet = [&](int a) {                     // step() without the strobe store
                const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
                const long long mi = mix24(a, ec), mq = mix24(a, es);
                ph += inc16;
                I[0] += (u64) mi; I[1] += I[0]; I[2] += I[1]; I[3] += I[2];
                Q[0] += (u64) mq; Q[1] += Q[0]; Q[2] += Q[1]; Q[3] += Q[2];
                i5i += (u32) (I[3] >> sh5);
                i5q += (u32) (Q[3] >> sh5);
            };
            const u32 M28 = 0x0FFFFFFFu;
            if (log2r == 1) {
                for (; t + 8 <= g1; t += 8) {
                    short buf[8];
                    samples8(t, buf);
                    uint4 vi, vq;
                    quiet(buf[0]); quiet(buf[1]); vi.x = i5i & M28; vq.x = i5q & M28;
                    quiet(buf[2]); quiet(buf[3]); vi.y = i5i & M28; vq.y = i5q & M28;
                    quiet(buf[4]); quiet(buf[5]); vi.z = i5i & M28; vq.z = i5q & M28;
                    quiet(buf[6]); quiet(buf[7]); vi.w = i5i & M28; vq.w = i5q & M28;
                    *(uint4 *) (c0i + o) = vi; *(uint4 *) (c0q + o) = vq;
                    o += 4;
                }
            } else {
                for (; t + 8 <= g1; t += 8) {
                    short buf[8];
                    samples8(t, buf);
                    uint2 vi, vq;
                    quiet(buf[0]); quiet(buf[1]); quiet(buf[2]); quiet(buf[3]); vi.x = i5i & M28; vq.x = i5q & M28;
                    quiet(buf[4]); quiet(buf[5]); quiet(buf[6]); quiet(buf[7]); vi.y = i5i & M28; vq.y = i5q & M28;
                    *(uint2 *) (c0i + o) = vi; *(uint2 *) (c0q + o) = vq;
                    o += 2;
                }
            }
            // c is 0 again: eight is a multiple of R
        }
        for (; t + 8 <= g1; t += 8) {                     // whole groups, no per-sample masking
            short buf[8];
            samples8(t, buf);
            if (PASS_B && c + 8 <= (u32) Rm1) {
                // no strobe inside this group (most groups once R >= 16): straight-line code, the eight table
                // reads go out together (with the strobe test between them each was waited for where it was issued)
                int ec[8], es[8];
#pragma unroll
                for (int w = 0; w < 8; w++) { ec[w] = KG_NCO_COS(tab, ph); es[w] = KG_NCO_SIN(tab, ph); ph += inc16; }
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const long long mi = mix24(buf[w], ec[w]), mq = mix24(buf[w], es[w]);
                    I[0] += (u64) mi; I[1] += I[0]; I[2] += I[1]; I[3] += I[2];
                    Q[0] += (u64) mq; Q[1] += Q[0]; Q[2] += Q[1]; Q[3] += Q[2];
                    i5i += (u32) (I[3] >> sh5);
                    i5q += (u32) (Q[3] >> sh5);
                }
                c += 8;
                continue;
            }
#pragma unroll
            for (int w = 0; w < 8; w++) step(buf[w]);
        }
        for (; t < s1; t++) step(adc[t]);                 // ragged end of the block's last run
        if (PASS_B) {
            if (endref == 2) { outTi = i5i & 0x0FFFFFFFu; outTq = i5q & 0x0FFFFFFFu; have = true; }
            else { tau[lI] = i5i & 0x0FFFFFFFu; tau[lQ] = i5q & 0x0FFFFFFFu; }
        } else {
            {   // the bias of pass A's inputs, out of the four integrators: 2^24 x C(len + k - 1, k), k = 1 .. 4, modulo 2^64
                const u64 len = (u64) (s1 - s0);          // <= 8192: len (len+1) (len+2) (len+3) < 2^53
                const u64 b1 = len, b2 = len * (len + 1) / 2, b3 = len * (len + 1) * (len + 2) / 6,
                          b4 = len * (len + 1) * (len + 2) * (len + 3) / 24;
                I[0] -= b1 << 24; I[1] -= b2 << 24; I[2] -= b3 << 24; I[3] -= b4 << 24;
                Q[0] -= b1 << 24; Q[1] -= b2 << 24; Q[2] -= b3 << 24; Q[3] -= b4 << 24;
            }
            ddc_state4 a, b;
            for (int k = 0; k < 4; k++) {
                // R <= 256: bits above 88 are never read, a logical shift will do; wider R (zero-state
                // sums): the value is signed and must be extended through bit 88
                const u64 hi_i = log2r <= 8 ? I[k] >> (64 - shift) : (u64) ((long long) I[k] >> (shift ? 64 - shift : 63));
                const u64 hi_q = log2r <= 8 ? Q[k] >> (64 - shift) : (u64) ((long long) Q[k] >> (shift ? 64 - shift : 63));
                a.i[k] = mk128(I[k] << shift, hi_i);
                b.i[k] = mk128(Q[k] << shift, hi_q);
            }
            if (endref) { a = ddc_to_end(a, (u64) (n - s1), co); b = ddc_to_end(b, (u64) (n - s1), co); }
            if (endref == 2) { outI = a; outQ = b; have = true; }
            else { local[lI] = a; local[lQ] = b; }
        }
        return;
    }

    if (MODE == DDC_NARROW) return;               // (not reached: the narrow form returned above)
    if (PASS_B) {
        // Pass B, R >= 512.  Nothing above bit 88 of an integrator is ever read (integrator 5 takes [88 -: 28]), so
        // the states are kept mod 2^96 as three 32-bit limbs: an integrator step is one three-instruction carry
        // chain (v_add_co / v_addc_co / v_addc_co) where the 128-bit form took an add, a compare for the carry and
        // a second add with carry-in per half -- 89 -> 57 instructions per sample and channel, and these five of
        // the fourteen BASELINE channels were two thirds of pass B's work.
        struct u96 { u32 w[3]; };
        auto add96 = [](u96 &a, const u96 &b) {
            unsigned c0, c1;
            a.w[0] = __builtin_addc(a.w[0], b.w[0], 0u, &c0);
            a.w[1] = __builtin_addc(a.w[1], b.w[1], c0, &c1);
            a.w[2] = a.w[2] + b.w[2] + c1;
        };
        u96 I[4], Q[4];
        {
            ddc_state4 a = local[lI], b = local[lQ];
            if (endref == 2) { a = ddc_add_state(a, wgbase[wbI]); b = ddc_add_state(b, wgbase[wbQ]); }
            if (endref) { a = ddc_from_end(a, (u64) (n - s0), co); b = ddc_from_end(b, (u64) (n - s0), co); }
            for (int k = 0; k < 4; k++) {
                I[k].w[0] = (u32) a.i[k].lo; I[k].w[1] = (u32) (a.i[k].lo >> 32); I[k].w[2] = (u32) a.i[k].hi;
                Q[k].w[0] = (u32) b.i[k].lo; Q[k].w[1] = (u32) (b.i[k].lo >> 32); Q[k].w[2] = (u32) b.i[k].hi;
            }
        }
        auto step = [&](int a) {
            const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
            const long long mi = mix24(a, ec), mq = mix24(a, es);
            ph += inc16;
            // in = sign-extended m << shift (shift = 65 - 5 log2 R is 0 .. 20 here), 96 bits
            const u64 li = (u64) mi << shift, lq = (u64) mq << shift;
            u96 xi, xq;
            xi.w[0] = (u32) li; xi.w[1] = (u32) (li >> 32); xi.w[2] = (u32) (shift ? mi >> (64 - shift) : mi >> 63);
            xq.w[0] = (u32) lq; xq.w[1] = (u32) (lq >> 32); xq.w[2] = (u32) (shift ? mq >> (64 - shift) : mq >> 63);
            add96(I[0], xi); add96(I[1], I[0]); add96(I[2], I[1]); add96(I[3], I[2]);
            add96(Q[0], xq); add96(Q[1], Q[0]); add96(Q[2], Q[1]); add96(Q[3], Q[2]);
            // integrator 5 accumulates integrator4[88 -: 28] (cic_wf1.vh)
            i5i += (I[3].w[2] << 3) | (I[3].w[1] >> 29);             // (masked to 28 bits where it is stored)
            i5q += (Q[3].w[2] << 3) | (Q[3].w[1] >> 29);
            c = (c + 1) & (u32) Rm1;
            if (c == 0) { c0i[o] = i5i & 0x0FFFFFFFu; c0q[o] = i5q & 0x0FFFFFFFu; o++; }              // strobe: sample_no was R - 1
        };
        long t = s0;
        for (; t + 8 <= g1; t += 8) {
            short buf[8];
            samples8(t, buf);
            if (c + 8 <= (u32) Rm1) {                         // no strobe inside this group: see the narrow path
                int ec[8], es[8];
#pragma unroll
                for (int w = 0; w < 8; w++) { ec[w] = KG_NCO_COS(tab, ph); es[w] = KG_NCO_SIN(tab, ph); ph += inc16; }
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const long long mi = mix24(buf[w], ec[w]), mq = mix24(buf[w], es[w]);
                    const u64 li = (u64) mi << shift, lq = (u64) mq << shift;
                    u96 xi, xq;
                    xi.w[0] = (u32) li; xi.w[1] = (u32) (li >> 32); xi.w[2] = (u32) (shift ? mi >> (64 - shift) : mi >> 63);
                    xq.w[0] = (u32) lq; xq.w[1] = (u32) (lq >> 32); xq.w[2] = (u32) (shift ? mq >> (64 - shift) : mq >> 63);
                    add96(I[0], xi); add96(I[1], I[0]); add96(I[2], I[1]); add96(I[3], I[2]);
                    add96(Q[0], xq); add96(Q[1], Q[0]); add96(Q[2], Q[1]); add96(Q[3], Q[2]);
                    i5i += (I[3].w[2] << 3) | (I[3].w[1] >> 29);
                    i5q += (Q[3].w[2] << 3) | (Q[3].w[1] >> 29);
                }
                c += 8;
                continue;
            }
#pragma unroll
            for (int w = 0; w < 8; w++) step(buf[w]);
        }
        for (; t < s1; t++) step(adc[t]);
        if (endref == 2) { outTi = i5i & 0x0FFFFFFFu; outTq = i5q & 0x0FFFFFFFu; have = true; }
        else { tau[lI] = i5i & 0x0FFFFFFFu; tau[lQ] = i5q & 0x0FFFFFFFu; }
        return;
    }
    if (MODE != DDC_ALL) return;
    // pass A of runs longer than 1024 samples at R >= 512: the zero-state sums need the full width
    ddc_state4 SI, SQ;
    if (PASS_B) {
        SI = local[lI];
        SQ = local[lQ];
        if (endref == 2) { SI = ddc_add_state(SI, wgbase[wbI]); SQ = ddc_add_state(SQ, wgbase[wbQ]); }
        if (endref) { SI = ddc_from_end(SI, (u64) (n - s0), co); SQ = ddc_from_end(SQ, (u64) (n - s0), co); }
    } else {
        for (int k = 0; k < 4; k++) { SI.i[k] = mk128(0, 0); SQ.i[k] = mk128(0, 0); }
    }
    auto step = [&](int a) {
        const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
        const long long mi = mix24(a, ec), mq = mix24(a, es);
        ph += inc16;
        // in = sign-extended m << shift, 128 bits
        const u128 xi = mk128((u64) mi << shift, (u64) (shift ? mi >> (64 - shift) : mi >> 63));
        const u128 xq = mk128((u64) mq << shift, (u64) (shift ? mq >> (64 - shift) : mq >> 63));
        SI.i[0] = add128(SI.i[0], xi); SI.i[1] = add128(SI.i[1], SI.i[0]);
        SI.i[2] = add128(SI.i[2], SI.i[1]); SI.i[3] = add128(SI.i[3], SI.i[2]);
        SQ.i[0] = add128(SQ.i[0], xq); SQ.i[1] = add128(SQ.i[1], SQ.i[0]);
        SQ.i[2] = add128(SQ.i[2], SQ.i[1]); SQ.i[3] = add128(SQ.i[3], SQ.i[2]);
        if (PASS_B) {
            // integrator 5 accumulates integrator4[88 -: 28] (cic_wf1.vh)
            i5i = (i5i + (u32) ((SI.i[3].hi << 3) | (SI.i[3].lo >> 61))) & 0x0FFFFFFFu;
            i5q = (i5q + (u32) ((SQ.i[3].hi << 3) | (SQ.i[3].lo >> 61))) & 0x0FFFFFFFu;
            c = (c + 1) & (u32) Rm1;
            if (c == 0) { c0i[o] = i5i; c0q[o] = i5q; o++; }          // strobe: sample_no was R - 1
        }
    };
    long t = s0;
    for (; t + 8 <= g1; t += 8) {
        short buf[8];
        samples8(t, buf);
#pragma unroll
        for (int w = 0; w < 8; w++) step(buf[w]);
    }
    for (; t < s1; t++) step(adc[t]);
    if (PASS_B) {
        if (endref == 2) { outTi = i5i; outTq = i5q; have = true; }
        else { tau[lI] = i5i; tau[lQ] = i5q; }
    } else {
        if (endref) { SI = ddc_to_end(SI, (u64) (n - s1), co); SQ = ddc_to_end(SQ, (u64) (n - s1), co); }
        if (endref == 2) { outI = SI; outQ = SQ; have = true; }
        else { local[lI] = SI; local[lQ] = SQ; }
    }
}

// endref == 2 (round 4): the carry scan's first two levels ride on pass A.  A workgroup's 256 threads hold 256 consecutive runs of
// one channel: their end-referred results are prefix-summed right here (additions of 4 x 96 bits: a wave scan, the four wave
// totals through LDS) -- local[r] = the sum of the workgroup's runs before r, wgtot = the workgroup's total -- and what is
// left for a kernel of its own is the prefix over at most 64 workgroup totals per (channel, I/Q) (ddc_wf_scan_wg_kernel:
// one wave each, a few microseconds where the chunked scan took 36 alone and 57 .. 85 beside the bypass kernel).
template <bool PASS_B, int MODE>
__global__ __launch_bounds__(DDC_THREADS) void ddc_wf_run_kernel(
    const short *__restrict__ adc, long n, int L, int nruns,
    const ddc_chan *__restrict__ chans, const int *__restrict__ chan_list, const u32 *__restrict__ nco,
    ddc_state4 *__restrict__ local, u32 *__restrict__ c0rel, u32 *__restrict__ tau,
    const long *__restrict__ c0off, const long *__restrict__ nouts, const int *__restrict__ sel, int stage_bytes,
    u64 pushed, const u64 *__restrict__ pdelta, const long *__restrict__ nlim, const int *__restrict__ reset_tab, int endref,
    const ddc_endco *__restrict__ endco, long endco_n,
    ddc_state4 *__restrict__ wgtot,           // pass A, endref == 2: [nlist][2][gridDim.x] out
    const ddc_state4 *__restrict__ wgbase,    // pass B, endref == 2: [nlist][2][gridDim.x] in
    u32 *__restrict__ wgtau)                  // pass B, endref == 2: [nlist][2][gridDim.x] out: the workgroup's integrator-5 total
{
    ddc_state4 oI, oQ;
    bool have = false;
    u32 oTi = 0, oTq = 0;
    ddc_wf_run_body<PASS_B, MODE>(adc, n, L, nruns, chans, chan_list, nco, local, c0rel, tau, c0off, nouts, sel, stage_bytes, pushed,
                            pdelta, nlim, reset_tab, endref, endco, endco_n, wgbase, oI, oQ, have, oTi, oTq);
    if (PASS_B && endref == 2) {
        // The prefix of the runs' integrator-5 totals in the same levels: tau[r] = the sum of the workgroup's runs before r,
        // wgtau = the workgroup's total; ddc_wf_tau_wg_kernel turns the totals into each workgroup's absolute start value and
        // the combs add the two (ddc_wf_scan_tau_kernel was 16 us on the step's critical path, 100 in a busy GPU).
        __shared__ u32 s_t[2][DDC_THREADS / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int li = sel ? sel[blockIdx.y] : (int) blockIdx.y;
        const long r = (long) blockIdx.x * DDC_THREADS + threadIdx.x;
        u32 vi = have ? oTi : 0u, vq = have ? oTq : 0u;
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
        if (have) {
            tau[((long) li * 2 + 0) * nruns + r] = (pi + xi) & 0x0FFFFFFFu;
            tau[((long) li * 2 + 1) * nruns + r] = (pq + xq) & 0x0FFFFFFFu;
        }
        if (threadIdx.x == DDC_THREADS - 1) {
            wgtau[((long) li * 2 + 0) * gridDim.x + blockIdx.x] = (pi + vi) & 0x0FFFFFFFu;
            wgtau[((long) li * 2 + 1) * gridDim.x + blockIdx.x] = (pq + vq) & 0x0FFFFFFFu;
        }
    }
    if (!PASS_B && endref == 2) {
        __shared__ sc4 s_w[2][DDC_THREADS / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int li = sel ? sel[blockIdx.y] : (int) blockIdx.y;
        const long r = (long) blockIdx.x * DDC_THREADS + threadIdx.x;
        sc4 iI = have ? sc_of(oI) : sc_zero(), iQ = have ? sc_of(oQ) : sc_zero();
        sc_scan_add<6>(iI, lane);
        sc_scan_add<6>(iQ, lane);
        if (lane == 63) { s_w[0][wave] = iI; s_w[1][wave] = iQ; }
        __syncthreads();
        sc4 pI = sc_zero(), pQ = sc_zero();
        for (int w = 0; w < wave; w++) { pI = sc_add(pI, s_w[0][w]); pQ = sc_add(pQ, s_w[1][w]); }
        if (threadIdx.x == DDC_THREADS - 1) {                      // inclusive of the last thread: the workgroup's total
            wgtot[((long) li * 2 + 0) * gridDim.x + blockIdx.x] = state_of(sc_add(pI, iI));
            wgtot[((long) li * 2 + 1) * gridDim.x + blockIdx.x] = state_of(sc_add(pQ, iQ));
        }
        sc4 xI = sc_shfl_up(iI, 1), xQ = sc_shfl_up(iQ, 1);        // exclusive inside the wave
        if (lane == 0) { xI = sc_zero(); xQ = sc_zero(); }
        if (have) {
            local[((long) li * 2 + 0) * nruns + r] = state_of(sc_add(pI, xI));
            local[((long) li * 2 + 1) * nruns + r] = state_of(sc_add(pQ, xQ));
        }
    }
}

// The third level: one wave per (channel, I/Q) over its workgroup totals (at most 64: max_runs / 256).
__global__ __launch_bounds__(64) void ddc_wf_scan_wg_kernel(
    const ddc_state4 *__restrict__ wgtot, ddc_state4 *__restrict__ wgbase, int gx, ddc_chan *__restrict__ chans,
    const int *__restrict__ chan_list, const long *__restrict__ nlim, const int *__restrict__ reset_tab)
{
    const int pair = blockIdx.x, li = pair >> 1, comp = pair & 1, lane = threadIdx.x;
    ddc_chan *ch = chans + chan_list[li];
    if (ch->log2r == 0) return;
    const int reset_first = reset_tab[li];
    sc4 inc = lane < gx ? sc_of(wgtot[(long) pair * gx + lane]) : sc_zero();
    // the carried-in state, advanced to the end of the entry's share (lane 0; the loads above are in flight meanwhile)
    sc4 base = sc_zero();
    if (lane == 0 && !reset_first) base = sc_T((u64) nlim[li], sc_of(ch->integ[comp]));
    sc_scan_add<6>(inc, lane);
    base = sc_shfl(base, 0);
    sc4 exc = sc_shfl_up(inc, 1);
    if (lane == 0) exc = sc_zero();
    if (lane < gx) wgbase[(long) pair * gx + lane] = state_of(sc_add(base, exc));
    if (lane == 63) ch->integ[comp] = state_of(sc_add(base, inc));     // the state after the block
}

#define BYP_G 4                                // groups of four samples per thread and block: a block is BYP_G x 1024 samples
// A whole, aligned block of NB bypass channels, straight-line (the general form in the kernel spends most of its instructions
// on the ragged cases: a branch and an exec mask per group, channel and store).
template <int NB>
DDC_DEV void ddc_bypass_block(const int2 (&cur)[BYP_G], long blk, const short *tab, const u64 (&inc16)[4], const u64 (&ph0)[4],
                              short2 *const (&orow)[4])
{
#pragma unroll
    for (int g = 0; g < BYP_G; g++) {
        const long t0 = blk * (BYP_G * 1024) + g * 1024 + 4 * (long) threadIdx.x;
        const int2 v = cur[g];
        const short a[4] = {(short) v.x, (short) (v.x >> 16), (short) v.y, (short) (v.y >> 16)};
#pragma unroll
        for (int b = 0; b < NB; b++) {
            u64 ph = (ph0[b] + (u64) t0 * (inc16[b] >> 16)) << 16;    // top-aligned: wraps by itself
            int4 w0;
            int *w = (int *) &w0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
                const int mi = mix24(a[q], ec), mq = mix24(a[q], es);
                w[q] = (int) (((u32) (mi >> 8) & 0xffffu) | ((u32) (mq >> 8) << 16));      // short2 {i, q}
                ph += inc16[b];
            }
            *(int4 *) (orow[b] + t0) = w0;
        }
    }
}
// The whole blocks of a pass, in a loop of their own: the next block's loads, this block's arithmetic, its stores -- nothing
// conditional, so the wait in front of the arithmetic counts exactly the operations issued since this block's loads.
template <int NB>
DDC_DEV long ddc_bypass_whole(const short *__restrict__ adc, long blk, long nfb, const short *tab, const u64 (&inc16)[4],
                              const u64 (&ph0)[4], short2 *const (&orow)[4])
{
    auto fetch = [&](long bk, int2 (&v)[BYP_G]) {
#pragma unroll
        for (int g = 0; g < BYP_G; g++) v[g] = *(const int2 *) (adc + bk * (BYP_G * 1024) + g * 1024 + 4 * (long) threadIdx.x);
    };
    if (blk >= nfb) return blk;
    int2 cur[BYP_G], nxt[BYP_G];
    fetch(blk, cur);
    for (; blk < nfb; blk += gridDim.x) {
        fetch(blk + gridDim.x < nfb ? blk + gridDim.x : blk, nxt);
        // (opaque: a phase accumulator and a row address per group and channel, carried from block to block by the
        // compiler's strength reduction, were 216 registers and two waves per SIMD)
        long bo = blk;
        asm volatile("" : "+s"(bo));
        ddc_bypass_block<NB>(cur, bo, tab, inc16, ph0, orow);
#pragma unroll
        for (int g = 0; g < BYP_G; g++) cur[g] = nxt[g];
    }
    return blk;
}

// The same whole blocks when the caller's rows (or the sample stream) are NOT 16-byte (8-byte) aligned -- `out_stride = n + 1`
// is enough: four single-dword stores per lane and channel at a 16-byte pitch then took the place of one 16-byte store
// (86 us instead of 36 for 2^24 samples x 2 channels; tools/micro/bypass_rate.hip).  Transposed lane mapping instead: lane l of
// wave w takes samples 256 w + 64 q + l (q = 0 .. 3) of a 1024-sample group, so every load instruction reads 128 contiguous
// bytes and every store instruction writes 256, whatever the alignment.
template <int NB>
DDC_DEV long ddc_bypass_whole_t(const short *__restrict__ adc, long blk, long nfb, const short *tab, const u64 (&inc16)[4],
                                const u64 (&ph0)[4], short2 *const (&orow)[4])
{
    const long lo = 256 * (long) (threadIdx.x >> 6) + (threadIdx.x & 63);       // this lane's first sample inside a group
    auto fetch = [&](long bk, short (&v)[BYP_G][4]) {
#pragma unroll
        for (int g = 0; g < BYP_G; g++)
#pragma unroll
            for (int q = 0; q < 4; q++) v[g][q] = adc[bk * (BYP_G * 1024) + g * 1024 + lo + 64 * q];
    };
    if (blk >= nfb) return blk;
    short cur[BYP_G][4], nxt[BYP_G][4];
    fetch(blk, cur);
    for (; blk < nfb; blk += gridDim.x) {
        fetch(blk + gridDim.x < nfb ? blk + gridDim.x : blk, nxt);
        long bo = blk;
        asm volatile("" : "+s"(bo));
#pragma unroll
        for (int g = 0; g < BYP_G; g++) {
            const long t0 = bo * (BYP_G * 1024) + g * 1024 + lo;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                u64 ph = (ph0[b] + (u64) t0 * (inc16[b] >> 16)) << 16;    // top-aligned: wraps by itself
                const u64 step = inc16[b] << 6;                           // 64 samples on
                int *o = (int *) (orow[b] + t0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
                    const int mi = mix24(cur[g][q], ec), mq = mix24(cur[g][q], es);
                    o[64 * q] = (int) (((u32) (mi >> 8) & 0xffffu) | ((u32) (mq >> 8) << 16));      // short2 {i, q}
                    ph += step;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < BYP_G; g++)
#pragma unroll
            for (int q = 0; q < 4; q++) cur[g][q] = nxt[g][q];
    }
    return blk;
}

// R == 1 bypass (cic_prune_var.v:289-297): out = mixer output [23 -: 16], no filter state at all,
// so it is sample-parallel: lane-contiguous groups of four samples per thread.
__global__ __launch_bounds__(256, 4) void ddc_wf_bypass_kernel(
    const short *__restrict__ adc, long n, const ddc_chan *__restrict__ chans, const int *__restrict__ chan_list,
    const int *__restrict__ bypass_list, int nbypass,     // list entries with R == 1
    const u32 *__restrict__ nco, short2 *__restrict__ out, long out_stride, u64 pushed, const u64 *__restrict__ pdelta,
    const long *__restrict__ nlim,            // [nlist] samples this entry takes (capture: max_out; else n)
    const long *__restrict__ out_off,         // [nlist] entry li writes at out + li out_stride + out_off[li]
    int by_chan)                              // ... or at row chan_list[li] (kg_ctx::rows_by_chan)
{
    __shared__ short tab[DDC_TAB];
    for (int i = threadIdx.x; i < DDC_TAB / 2; i += 256) ((u32 *) tab)[i] = nco[i];
    __syncthreads();
    // Persistent over the blocks of the stream: the 20 KiB table is staged once per workgroup.  Round 4: ONE pass over the ADC
    // samples for ALL bypass channels (up to four at a time; the channels' parameters are read once, into scalar registers),
    // and the kernel was LATENCY-bound on its own reads: one 8-byte load per lane, waited for where it was issued, 8 KiB in
    // flight per CU -- 33 MB took 48 us of the kernel's 86 (knock-out builds: no stores 69 us, no table reads 83, loads alone
    // 8.6).  Now a thread has eight loads in flight (this block's four groups and the next block's) and the whole blocks
    // run in a straight-line loop (ddc_bypass_whole).
    const long bs = BYP_G * 1024;
    const long nblk = (n + bs - 1) / bs;
    const bool al8 = (((uintptr_t) adc) & 7) == 0;
    for (int b0 = 0; b0 < nbypass; b0 += 4) {
        const int nb4 = nbypass - b0 < 4 ? nbypass - b0 : 4;
        u64 inc16[4], ph0[4]; long nbs[4]; short2 *orow[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int li = bypass_list[b0 + (b < nb4 ? b : 0)];
            const ddc_chan *ch = chans + chan_list[li];
            inc16[b] = ch->phase_inc << 16;
            ph0[b] = ch->phase + (pushed + pdelta[li]) * ch->phase_inc;
            nbs[b] = b < nb4 ? nlim[li] : 0;
            orow[b] = out + (long) (by_chan ? chan_list[li] : li) * out_stride + out_off[li];
        }
        // samples up to which every channel of the four takes whole aligned blocks, rows that take 16-byte stores
        long nfull = n;
        bool rows16 = true;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (b < nb4) { nfull = nbs[b] < nfull ? nbs[b] : nfull; rows16 = rows16 && (((uintptr_t) orow[b]) & 15) == 0; }
        const long nany = nfull;                   // samples up to which every channel of the four takes whole blocks
        long blk = blockIdx.x;
        if (rows16 && al8) {
            const long nfb = nfull / bs;
            switch (nb4) {
            case 1: blk = ddc_bypass_whole<1>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            case 2: blk = ddc_bypass_whole<2>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            case 3: blk = ddc_bypass_whole<3>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            default: blk = ddc_bypass_whole<4>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            }
        } else {
            const long nfb = nany / bs;
            switch (nb4) {
            case 1: blk = ddc_bypass_whole_t<1>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            case 2: blk = ddc_bypass_whole_t<2>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            case 3: blk = ddc_bypass_whole_t<3>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            default: blk = ddc_bypass_whole_t<4>(adc, blk, nfb, tab, inc16, ph0, orow); break;
            }
        }
        // what is left: the ragged end of the stream, entries that take less than the block (a capture), unaligned rows
        for (; blk < nblk; blk += gridDim.x) {
            for (int g = 0; g < BYP_G; g++) {
                const long t0 = blk * bs + g * 1024 + 4 * (long) threadIdx.x;
                if (t0 >= n) continue;
                short a[4];
                const bool full = t0 + 4 <= n;
                if (full && al8) {
                    const int2 v = *(const int2 *) (adc + t0);
                    a[0] = (short) v.x; a[1] = (short) (v.x >> 16); a[2] = (short) v.y; a[3] = (short) (v.y >> 16);
                } else {
                    for (int q = 0; q < 4; q++) a[q] = (t0 + q < n) ? adc[t0 + q] : (short) 0;
                }
                for (int b = 0; b < nb4; b++) {
                    const long nb = nbs[b];
                    if (t0 >= nb) continue;
                    const bool fullb = t0 + 4 <= nb;
                    u64 ph = (ph0[b] + (u64) t0 * (inc16[b] >> 16)) << 16;    // top-aligned: wraps by itself
                    short2 *o = orow[b] + t0;
                    short2 r[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int ec = KG_NCO_COS(tab, ph), es = KG_NCO_SIN(tab, ph);
                        const int mi = mix24(a[q], ec), mq = mix24(a[q], es);
                        r[q] = make_short2((short) (mi >> 8), (short) (mq >> 8));
                        ph += inc16[b];
                    }
                    if (fullb && (((uintptr_t) o) & 15) == 0) {
                        int4 w0;
                        w0.x = *(int *) &r[0]; w0.y = *(int *) &r[1]; w0.z = *(int *) &r[2]; w0.w = *(int *) &r[3];
                        *(int4 *) o = w0;
                    } else {
                        for (int q = 0; q < 4; q++) if (t0 + q < nb) o[q] = r[q];
                    }
                }
            }
        }
    }
}

// Carry scan of the integrator states and prefix sum of the run totals.
// One wave per (channel, I/Q).  MODE 0: states (after pass A), MODE 1: tau (after pass B).
DDC_DEV u64 shfl_up64(u64 v, int d)
{
    const u32 lo = __shfl_up((u32) v, d), hi = __shfl_up((u32) (v >> 32), d);
    return ((u64) hi << 32) | lo;
}
DDC_DEV u64 shfl64(u64 v, int src)
{
    const u32 lo = __shfl((u32) v, src), hi = __shfl((u32) (v >> 32), src);
    return ((u64) hi << 32) | lo;
}
#define DDC_SCAN_WAVES 8
#define DDC_SCAN_MAX_CHUNKS 16
// Ordered fold of the affine maps held by lanes 0 .. 2^LOG - 1 (lane order = time order; a lane with
// len = 0 and e = 0 is the identity): log steps of the inclusive scan instead of a serial chain of
// wide multiply-adds.  -> the fold of lanes 0 .. lane (every lane of the group must call it)
template <int LOG> DDC_DEV void ddc_fold_lanes(sc4 &e, u64 &len, int lane, const sc_tab &t, int j0)
{
#pragma unroll
    for (int d = 1, j = j0; d < (1 << LOG); d <<= 1, j++) {
        const sc4 a = sc_shfl_up(e, d);
        const u64 alen = shfl_up64(len, d);
        if (lane >= d) { e = sc_add(sc_T_tab(t, j, len, a), e); len += alen; }
    }
}
// What a chunk of runs does to the integrator state: state_out = T(len) state_in + e.  Published by the
// workgroup that owns the chunk for the workgroups of the later chunks of the same (channel, I/Q).
struct ddc_chunk_agg { ddc_state4 e; u64 len; u32 epoch; u32 pad; };

// Carry scan of the integrator states.  A (channel, I/Q) pair's runs are cut into `nchunk` chunks, one
// workgroup of eight waves each: lane-local composition of a few runs, inclusive scan inside each wave,
// the eight wave totals folded through LDS -> the chunk's aggregate, published in global memory; the
// start state of the chunk = the saved state advanced through the aggregates of the chunks before it
// (waited for one by one; every workgroup publishes before it waits, and workgroups take their chunk
// numbers from a ticket counter in the order they start, so whoever is waited for is already running);
// then every lane walks its runs again from its exact start state.  (One workgroup per pair took 120 us
// for 16 384 runs -- 28 workgroups on 256 CUs, each lane composing 32 runs of 128-bit multiply-adds twice.)
__global__ __launch_bounds__(64 * DDC_SCAN_WAVES) void ddc_wf_scan_states_kernel(
    ddc_state4 *__restrict__ local, long n, int L, int nruns, ddc_chan *__restrict__ chans,
    const int *__restrict__ chan_list, int npairs, int nchunk, ddc_chunk_agg *__restrict__ aggs,
    u32 *__restrict__ ticket, u32 ticket_base, u32 epoch, const sc_tab tab,
    const long *__restrict__ nlim, const int *__restrict__ reset_tab)      // per entry: samples consumed; capture: the carried-in state is zero
{
    __shared__ sc4 w_state[DDC_SCAN_WAVES];
    __shared__ u64 w_len[DDC_SCAN_WAVES];
    __shared__ sc4 s_start;
    __shared__ u32 s_id;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    if (gl == 0) s_id = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base;
    __syncthreads();
    // (through readfirstlane: the ticket is the same in every lane, and with it everything derived from the entry -- its
    // share of the block, its run count, the chunk bounds -- stays in scalar registers)
    const int id = __builtin_amdgcn_readfirstlane((int) s_id), g = id / npairs, pair = id - g * npairs;     // chunk-major: chunk 0 of every pair first
    const int li = pair >> 1, comp = pair & 1;
    ddc_chan *ch = chans + chan_list[li];
    if (ch->log2r == 0) return;                   // (the whole pair: nobody waits for a bypass channel)
    const int reset_first = reset_tab[li];
    ddc_state4 *st = local + ((long) li * 2 + comp) * nruns;        // (rows of the whole launch's run count apart)
    n = nlim[li];                                 // this entry's share of the block and the runs that hold it
    { const int nr = (int) ((n + L - 1) >> (31 - __builtin_clz(L))); nruns = nr < nruns ? nr : nruns; }    // L is a power of two
    ddc_chunk_agg *agg = aggs + (long) pair * DDC_SCAN_MAX_CHUNKS;
    const int cper = (nruns + nchunk - 1) / nchunk;
    const int c0 = g * cper < nruns ? g * cper : nruns, c1 = c0 + cper < nruns ? c0 + cper : nruns;
    const int per = (c1 - c0 + 64 * DDC_SCAN_WAVES - 1) / (64 * DDC_SCAN_WAVES);
    const int r0 = c0 + gl * per < c1 ? c0 + gl * per : c1, r1 = (r0 + per < c1) ? r0 + per : c1;
    auto run_len = [&](int r) -> u64 { const long s0 = (long) r * L; return (u64) ((s0 + L < n ? s0 + L : n) - s0); };
    // the saved state is read before anything is published: the last chunk rewrites it at the end, after
    // it has seen every other chunk's aggregate
    sc4 saved = sc_zero();
    if (gl == 0 && !reset_first) saved = sc_of(ch->integ[comp]);
    // 1. lane-local composition
    sc4 acc = sc_zero(); u64 len = 0;
    const sc_coef kL = sc_coef_for((u64) L);      // every run but possibly the last has length L
    for (int r = r0; r < r1; r++) {
        const u64 l = run_len(r);
        acc = sc_add(l == (u64) L ? sc_Tc(kL, acc) : sc_T(l, acc), sc_of(st[r]));
        len += l;
    }
    // 2. inclusive wave scan: earlier lanes first
    sc4 inc = acc; u64 ilen = len;
#pragma unroll
    for (int d = 1, j = 0; d < 64; d <<= 1, j++) {
        const sc4 a = sc_shfl_up(inc, d);
        const u64 alen = shfl_up64(ilen, d);
        if (lane >= d) { inc = sc_add(sc_T_tab(tab, j, ilen, a), inc); ilen += alen; }
    }
    if (lane == 63) { w_state[wave] = inc; w_len[wave] = ilen; }
    __syncthreads();
    // 3. wave 0: the inclusive prefixes of the eight wave totals (back into w_state / w_len: wave w starts behind
    // prefix w - 1), the chunk's aggregate out, the aggregates of the earlier chunks in, folded in log steps
    if (wave == 0) {
        sc4 e = sc_zero(); u64 l = 0;
        if (lane < DDC_SCAN_WAVES) { e = w_state[lane]; l = w_len[lane]; }
        ddc_fold_lanes<3>(e, l, lane, tab, 6);
        if (lane < DDC_SCAN_WAVES) { w_state[lane] = e; w_len[lane] = l; }
        if (lane == DDC_SCAN_WAVES - 1 && g + 1 < nchunk) {
            agg[g].e = state_of(e); agg[g].len = l;
            __hip_atomic_store(&agg[g].epoch, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lanes past the earlier chunks carry a zero aggregate of a full chunk's length: nobody reads their fold, and
        // with it every lane advances by the table's lengths
        e = sc_zero();
        l = tab.len[9];
        if (lane < g) {
            while (__hip_atomic_load(&agg[lane].epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch)
                __builtin_amdgcn_s_sleep(2);
            ddc_state4 ge;
            for (int k = 0; k < 4; k++) {          // past the acquire: loads that do not come from a stale line
                ge.i[k].lo = __hip_atomic_load(&agg[lane].e.i[k].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ge.i[k].hi = __hip_atomic_load(&agg[lane].e.i[k].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            e = sc_of(ge);
            l = __hip_atomic_load(&agg[lane].len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ddc_fold_lanes<4>(e, l, lane, tab, 9);     // DDC_SCAN_MAX_CHUNKS = 16 lanes; lane g - 1: chunks 0 .. g - 1
        const int src = g > 0 ? g - 1 : 0;
        e = sc_shfl(e, src); l = shfl64(l, src);
        if (g == 0) { e = sc_zero(); l = 0; }
        const sc4 sv = sc_shfl(saved, 0);
        if (lane == 0) s_start = sc_add(sc_T(l, sv), e);
    }
    __syncthreads();
    // state at the start of this wave's first run: the chunk's start advanced through the earlier waves
    sc4 ws;
    {
        sc4 e = sc_zero(); u64 l = 0;
        if (wave > 0) { e = w_state[wave - 1]; l = w_len[wave - 1]; }
        ws = sc_add(sc_T(l, s_start), e);
    }
    // exclusive prefix of this lane inside its wave = inclusive of lane - 1
    sc4 exc = sc_shfl_up(inc, 1); u64 elen = shfl_up64(ilen, 1);
    if (lane == 0) { exc = sc_zero(); elen = 0; }
    sc4 c = sc_add(sc_T(elen, ws), exc);
    // 4. per-run carried states
    for (int r = r0; r < r1; r++) {
        const sc4 e = sc_of(st[r]);
        st[r] = state_of(c);
        const u64 l = run_len(r);
        c = sc_add(l == (u64) L ? sc_Tc(kL, c) : sc_T(l, c), e);
    }
    // the lane that owns the last run holds the end state
    if (r1 == nruns && r0 < nruns) ch->integ[comp] = state_of(c);
}

// ... and of the integrator-5 totals: in = the workgroups' totals, out = integrator 5 at each workgroup's first run.
__global__ __launch_bounds__(64) void ddc_wf_tau_wg_kernel(u32 *__restrict__ wgtau, int gx, ddc_chan *__restrict__ chans,
                                                          const int *__restrict__ chan_list, const int *__restrict__ reset_tab)
{
    const int pair = blockIdx.x, li = pair >> 1, comp = pair & 1, lane = threadIdx.x;
    ddc_chan *ch = chans + chan_list[li];
    if (ch->log2r == 0) return;
    const int reset_first = reset_tab[li];
    u32 v = lane < gx ? wgtau[(long) pair * gx + lane] : 0u;
    const u32 i5 = reset_first ? 0u : ch->integ5[comp];
    for (int d = 1; d < 64; d <<= 1) { const u32 a = __shfl_up(v, d); if (lane >= d) v += a; }
    u32 x = __shfl_up(v, 1);
    if (lane == 0) x = 0;
    if (lane < gx) wgtau[(long) pair * gx + lane] = (i5 + x) & 0x0FFFFFFFu;
    if (lane == 63) ch->integ5[comp] = (i5 + v) & 0x0FFFFFFFu;       // the state after the call
}

// Prefix sum of the runs' integrator-5 totals: one workgroup of eight waves per (channel, I/Q).
// Run r = k * 512 + thread: every load and store of a tile of 512 runs is contiguous across the workgroup
// (round 2 gave each thread 32 consecutive runs: every access its own line, 28 us for a 16 384-entry prefix).
// Per tile a wave scan (tile values stay in registers), the (tile, wave) totals -- at most 32 x 8 -- through LDS
// and one more scan by the first four waves, then the offsets are applied.
#define DDC_TAU_TILES 32                      // max_runs = 16384 = 32 tiles of 512
__global__ __launch_bounds__(64 * DDC_SCAN_WAVES) void ddc_wf_scan_tau_kernel(
    u32 *__restrict__ tau, int nruns, ddc_chan *__restrict__ chans, const int *__restrict__ chan_list,
    const long *__restrict__ nlim, int L, const int *__restrict__ reset_tab)
{
    __shared__ u32 s_tot[DDC_TAU_TILES * DDC_SCAN_WAVES];       // totals, then exclusive offsets
    __shared__ u32 s_w4[4];
    const int li = blockIdx.x >> 1, comp = blockIdx.x & 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = threadIdx.x;
    ddc_chan *ch = chans + chan_list[li];
    if (ch->log2r == 0) return;
    const int reset_first = reset_tab[li];
    u32 *tv = tau + ((long) li * 2 + comp) * nruns;
    {   // the runs this entry walked (li comes from blockIdx: a scalar load, a scalar ntile)
        const int nr = (int) ((nlim[li] + L - 1) >> (31 - __builtin_clz(L)));       // L is a power of two
        nruns = nr < nruns ? nr : nruns;
    }
    const int ntile = (nruns + 511) >> 9;         // <= DDC_TAU_TILES (kg_ddc_create caps max_runs at 16384)
    u32 inc[DDC_TAU_TILES], own[DDC_TAU_TILES];
#pragma unroll
    for (int k = 0; k < DDC_TAU_TILES; k++) {
        const int r = (k << 9) + gl;
        own[k] = (k < ntile && r < nruns) ? tv[r] : 0u;
    }
    // eight tiles at a time behind ONE scalar branch (ntile is wave-uniform): a capture's short entries skip the groups of
    // tiles they lack, and inside a group the eight scans stay independent straight-line chains (a branch per tile made
    // the continuous case -- all 32 tiles -- 13 -> 20 us: the chains then ran one after the other)
#pragma unroll
    for (int g8 = 0; g8 < DDC_TAU_TILES; g8 += 8) {
        if (g8 < ntile) {
#pragma unroll
            for (int k = g8; k < g8 + 8; k++) {
                u32 v = own[k];                       // (zero past ntile: loaded so above)
                for (int d = 1; d < 64; d <<= 1) { const u32 a = __shfl_up(v, d); if (lane >= d) v += a; }
                inc[k] = v;
                if (lane == 63) s_tot[k * DDC_SCAN_WAVES + wave] = v;
            }
        } else {
#pragma unroll
            for (int k = g8; k < g8 + 8; k++) {
                inc[k] = 0;
                if (lane == 63) s_tot[k * DDC_SCAN_WAVES + wave] = 0;
            }
        }
    }
    __syncthreads();
    // exclusive prefix over the 256 (tile, wave) totals, in time order: waves 0..3, one total per lane
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
        s_tot[gl] = base + tinc - t;              // exclusive
    }
    const u32 i5 = reset_first ? 0u : ch->integ5[comp];
    __syncthreads();                              // offsets written; every wave has read the saved value
#pragma unroll
    for (int k = 0; k < DDC_TAU_TILES; k++) {
        const int r = (k << 9) + gl;
        if (k < ntile && r < nruns)
            tv[r] = (i5 + s_tot[k * DDC_SCAN_WAVES + wave] + inc[k] - own[k]) & 0x0FFFFFFFu;     // I5 at the run's start
    }
    if (gl == 0) {                                // the sum of every run = state after the call
        u32 total = i5;
        for (int w = 0; w < 4; w++) total += s_w4[w];
        ch->integ5[comp] = total & 0x0FFFFFFFu;
    }
}

// sign-extend the low `bits` bits
DDC_DEV int sext32(int v, int bits) { return (v << (32 - bits)) >> (32 - bits); }

// Combs + rounding, four consecutive outputs per thread.  cic_wf1.vh: comb widths
// 23,22,21,20,20; their inputs drop 5,1,1,1,0 LSBs; out = comb5[19 -: 16] + comb5[3].
// Channels decimate differently, so the grid is flat: workgroup w belongs to the list entry
// li with wg_start[li] <= w < wg_start[li + 1] (bypass channels own none).  A workgroup first
// puts the absolute integrator-5 values of its 1024 outputs and the five before them in LDS
// (16-byte loads of the run-relative values: the I and Q planes of a channel start 16-byte aligned),
// then every thread runs the five combs over a nine-value window: 30 differences for four outputs
// where one output per thread took 15 each, and a quarter of the load and store instructions.
#define DDC_COMB_TILE 1024
__global__ __launch_bounds__(256) void ddc_wf_comb_kernel(
    const u32 *__restrict__ c0rel, const u32 *__restrict__ i5start, int log2L, int nruns,
    const long *__restrict__ c0off, const ddc_chan *__restrict__ chans, const int *__restrict__ chan_list, const long *__restrict__ nouts,
    u64 pushed, const u64 *__restrict__ pdelta,   // samples pushed since the entry's reference point: pushed + pdelta[li]
    const int *__restrict__ reset_tab,        // [nlist] capture: counter and comb registers start the block at zero
    const int *__restrict__ wg_start, int nlist,
    short2 *__restrict__ out, long out_stride, const long *__restrict__ out_off,   // entry li writes at out + li out_stride + out_off[li]
    u32 *__restrict__ hist_out,               // [nlist][2][5]
    const u32 *__restrict__ wgtau, int gx,    // end-referred levels: i5start[] is relative to its run pass workgroup's start value wgtau[li][comp][run / 256]; or null
    int by_chan)                              // rows of `out` by channel number (kg_ctx::rows_by_chan)
{
    __shared__ __attribute__((aligned(16))) int s_c0[2][DDC_COMB_TILE + 8];       // [d]: output o0 - 8 + d (three unused slots keep 16-byte rows)
    int lo = 0, hi = nlist;                   // wg_start[lo] <= blockIdx.x < wg_start[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int) blockIdx.x >= wg_start[mid]) lo = mid; else hi = mid; }
    const int li = lo, t = threadIdx.x;
    const ddc_chan *ch = chans + chan_list[li];
    const int log2r = ch->log2r;
    const int reset_first = reset_tab[li];
    const long nout = nouts[li], plane = (nout + 3) & ~3l;
    const long o0 = (long) (blockIdx.x - wg_start[li]) * DDC_COMB_TILE;
    const u32 base = reset_first ? 0u : (u32) (((u64) ch->sample_no + pushed + pdelta[li]) & ((1ull << log2r) - 1));     // sample_no before this call
    auto absolute = [&](int comp, long oo, u32 rel) -> u32 {
        const long g = ((oo + 1) << log2r) - 1 - (long) base;    // sample index of the strobe
        const int run = (int) (g >> log2L);
        const u32 wb = wgtau ? wgtau[((long) li * 2 + comp) * gx + (run >> 8)] : 0u;
        return (rel + i5start[((long) li * 2 + comp) * nruns + run] + wb) & 0x0FFFFFFFu;
    };
#pragma unroll
    for (int comp = 0; comp < 2; comp++) {
        const u32 *src = c0rel + c0off[li] + comp * plane;
        // the four outputs of this thread
        const long oo = o0 + 4 * t;
        u32 v[4] = {0, 0, 0, 0};
        if (oo + 4 <= nout) {
            const uint4 q = *(const uint4 *) (src + oo);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (int k = 0; k < 4; k++) if (oo + k < nout) v[k] = src[oo + k];
        }
        int4 w4;                              // one 16-byte LDS store per thread (four 4-byte ones: stride-4 words, 4-way conflicts)
        int *w4p = (int *) &w4;
        for (int k = 0; k < 4; k++) {
            u32 a = 0;
            if (oo + k < nout) {
                a = absolute(comp, oo + k, v[k]);
                if (oo + k >= nout - 5) hist_out[((long) li * 2 + comp) * 5 + (int) (oo + k - (nout - 5))] = a;
            }
            w4p[k] = sext32((int) a, 28);
        }
        *(int4 *) &s_c0[comp][8 + 4 * t] = w4;
        // the five outputs before the tile: strobes of earlier calls (zero after a reset) or earlier tiles
        if (t < 5) {
            const long pb = o0 - 5 + t;
            const u32 a = pb < 0 ? (reset_first ? 0u : ch->hist[comp][5 + pb]) : absolute(comp, pb, src[pb]);
            s_c0[comp][3 + t] = sext32((int) a, 28);
        }
    }
    __syncthreads();
    const long o = o0 + 4 * t;
    short res[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (o < nout) {
        for (int comp = 0; comp < 2; comp++) {
            // comb k output at position d needs its input at d and d-1
            const int W[5] = {23, 22, 21, 20, 20}, D[5] = {5, 1, 1, 1, 0};
            int v[9];
            {                                 // the nine-value window [3 + 4t, 12 + 4t) from three 16-byte reads of [4t, 4t + 12)
                const int4 q0 = *(const int4 *) &s_c0[comp][4 * t], q1 = *(const int4 *) &s_c0[comp][4 * t + 4],
                           q2 = *(const int4 *) &s_c0[comp][4 * t + 8];
                v[0] = q0.w; v[1] = q1.x; v[2] = q1.y; v[3] = q1.z; v[4] = q1.w; v[5] = q2.x; v[6] = q2.y; v[7] = q2.z; v[8] = q2.w;
            }
            int cnt = 9;
            for (int k = 0; k < 5; k++) {
                int x[9];
                for (int d = 0; d < cnt; d++) x[d] = sext32(v[d] >> D[k], W[k]);
                for (int d = 1; d < cnt; d++) v[d - 1] = sext32(x[d] - x[d - 1], W[k]);
                cnt--;
            }
            for (int k = 0; k < 4; k++) res[comp][k] = (short) ((v[k] >> 4) + ((v[k] >> 3) & 1));
        }
    }
    // out through LDS, so that the stores are lane-contiguous whatever the alignment of the caller's rows
    __syncthreads();                          // every thread has read its window
    int *s_out = &s_c0[0][0];
    {
        int4 o4;
        int *o4p = (int *) &o4;
        for (int k = 0; k < 4; k++) o4p[k] = (int) ((unsigned short) res[0][k] | ((unsigned) (unsigned short) res[1][k] << 16));
        *(int4 *) &s_out[4 * t] = o4;
    }
    __syncthreads();
    short2 *dst = out + (long) (by_chan ? chan_list[li] : li) * out_stride + out_off[li] + o0;
    for (int k = 0; k < 4; k++) {
        const int e = t + 256 * k;
        if (o0 + e < nout) { const int w = s_out[e]; dst[e] = make_short2((short) (w & 0xffff), (short) (w >> 16)); }
    }
}

// after a call: the comb history (phase and counter are reference values + the host's `pushed`, see ddc_chan)
__global__ void ddc_wf_finish_kernel(ddc_chan *__restrict__ chans, const int *__restrict__ chan_list, int nlist,
                                     long n, const long *__restrict__ nouts, const u32 *__restrict__ hist_new,
                                     const int *__restrict__ reset_tab)
{
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nlist) return;
    const int reset_first = reset_tab[li];
    ddc_chan *ch = chans + chan_list[li];
    if (ch->log2r == 0) return;
    const long nout = nouts[li];
    if (nout <= 0) return;
    for (int comp = 0; comp < 2; comp++) {
        u32 h[5];
        for (int d = 0; d < 5; d++) {
            // the last five strobes: fewer than five new ones keep part of the old history
            const long back = 4 - d;             // 0 = most recent
            if (back < nout) h[d] = hist_new[((long) li * 2 + comp) * 5 + d];
            else h[d] = reset_first ? 0u : ch->hist[comp][d + nout];
        }
        for (int d = 0; d < 5; d++) ch->hist[comp][d] = h[d];
    }
}

// ---------------------------------------------------------------------------
struct kg_ddc {
    kg_ctx *ctx;
    int nchan;
    long max_samples;
    ddc_chan *d_chans;
    std::vector<ddc_chan> h_chans;            // host mirror of the scalar fields
    u32 *d_nco;
    // Buffers a push's run passes fill and its output stage (run-total prefix, combs, finish) drains: TWO sets, used in
    // turn, so that push k + 1's run passes may overwrite nothing push k's output stage still reads (deferred mode).
    ddc_state4 *d_local[2]; u32 *d_c0rel[2], *d_tau[2], *d_hist;
    int max_runs; long c0_cap[2];
    ddc_chunk_agg *d_aggs; u32 *d_ticket; u32 ticket_base, epoch;     // chunked state scan
    hipStream_t side; hipEvent_t ev_fork, ev_join;                   // pass B of the small decimations beside the rest
    bool side_borrowed;                                              // `side` is the owner's (a receiver bank's): not destroyed here
    kg_stage_cache pack_cache;                 // the per-call tables of the last push (a steady stream repeats them: no upload)
    std::vector<u64> h_pushed;                 // per channel: samples pushed since its reference point (ddc_chan.phase / .sample_no)
    std::vector<char> h_stale;                 // per channel: a capture cut its filters short; the next continuous push resets it
    std::vector<char> seen;                    // scratch of a push: channels listed so far
    // Deferred output stage (kg_ddc_wf_set_deferred): the stage runs on `tail`, the context's stream carries only pass A,
    // the state scan and pass B, so the NEXT push's pass A starts while this push's combs are still writing.
    bool deferred;
    hipStream_t tail;
    hipEvent_t ev_runs, ev_tail[2];            // run passes of the push done (main stream) / output stage of the push of that parity done
    hipEvent_t ev_adc;                         // deferred mode: behind the output stream's last reader of the caller's samples (the bypass kernel)
    bool tail_rec[2];                          // ev_tail[p] has been recorded
    bool tail_unjoined;                        // the context's stream has not yet been made to wait for the last output stage
    int parity;                                // buffer set of the NEXT push
    void *after_ev;                            // kg_ddc_wf_tail_after: the next push's writers of the caller's rows wait for it
    ddc_endco *d_endco; long endco_n; int endco_L, endco_runs;      // end-distance coefficients of the last (n, L, nruns)
    ddc_state4 *d_wgtot[2], *d_wgbase[2];     // [nchan][2][DDC_MAX_GX] per buffer set: pass A's workgroup totals, pass B's workgroup bases
    u32 *d_wgtau[2];                           // likewise: the integrator-5 totals of pass B's workgroups, then their start values
};

static const int DDC_RUN_MIN = 64, DDC_RUN_MAX = 8192, DDC_TARGET_RUNS = 8192;
static const int DDC_MAX_GX = DDC_TARGET_RUNS * 2 / DDC_THREADS;      // workgroups of a run pass per channel: one lane each in ddc_wf_scan_wg_kernel
static_assert(DDC_MAX_GX <= 64, "ddc_wf_scan_wg_kernel scans one workgroup total per lane");
static_assert(DDC_THREADS == 256, "ddc_wf_comb_kernel finds a run's pass-B workgroup as run >> 8");

// The object's second stream, handed over by an owner that lays out its streams itself (kg_rxbank.hip: which hardware queue
// a stream lands on follows from the order streams are created in).  Before the first push.  Not part of the ABI.
int kg_ddc_use_side_stream(kg_ddc *d, hipStream_t s)
{
    KG_REQUIRE(d && s && !d->side, KG_ERR_STATE, "kg_ddc_use_side_stream: the object already has a second stream");
    KG_HIP(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
    KG_HIP(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
    d->side = s; d->side_borrowed = true;
    return KG_OK;
}

extern "C" {

int kg_ddc_create(kg_ctx *ctx, int nchan, size_t max_samples, kg_ddc **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_ddc_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nchan >= 1 && nchan <= 4096, KG_ERR_INVALID, "kg_ddc_create: nchan %d", nchan);
    KG_REQUIRE(max_samples >= 64 && max_samples <= ((size_t) 1 << 32), KG_ERR_INVALID,
               "kg_ddc_create: max_samples %zu", max_samples);
    kg_ddc *d = new (std::nothrow) kg_ddc();
    KG_REQUIRE(d != nullptr, KG_ERR_NOMEM, "kg_ddc_create: alloc");
    d->ctx = ctx; d->nchan = nchan; d->max_samples = (long) max_samples;
    d->h_chans.assign(nchan, ddc_chan());
    for (auto &c : d->h_chans) memset(&c, 0, sizeof c);
    d->max_runs = (int) ((max_samples + DDC_RUN_MIN - 1) / DDC_RUN_MIN);
    if (d->max_runs > DDC_TARGET_RUNS * 2) d->max_runs = DDC_TARGET_RUNS * 2;
    d->c0_cap[0] = d->c0_cap[1] = 0; d->d_c0rel[0] = d->d_c0rel[1] = nullptr;
    d->h_pushed.assign(nchan, 0);
    d->h_stale.assign(nchan, 0);
    d->deferred = false; d->tail = nullptr; d->tail_rec[0] = d->tail_rec[1] = false; d->tail_unjoined = false;
    d->parity = 0; d->after_ev = nullptr;
    KG_HIP(hipMalloc((void **) &d->d_chans, sizeof(ddc_chan) * nchan));
    KG_HIP(hipMemset(d->d_chans, 0, sizeof(ddc_chan) * nchan));
    KG_HIP(hipMalloc((void **) &d->d_nco, sizeof(short) * DDC_TAB));
    KG_HIP(hipFuncSetAttribute((const void *) ddc_wf_run_kernel<true, DDC_ALL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               DDC_STAGE_BYTES));
    for (int p = 0; p < 2; p++) {
        KG_HIP(hipMalloc((void **) &d->d_local[p], sizeof(ddc_state4) * 2 * (size_t) nchan * d->max_runs));
        KG_HIP(hipMalloc((void **) &d->d_tau[p], sizeof(u32) * 2 * (size_t) nchan * d->max_runs));
    }
    KG_HIP(hipMalloc((void **) &d->d_hist, sizeof(u32) * 10 * (size_t) nchan));
    KG_HIP(hipMalloc((void **) &d->d_endco, sizeof(ddc_endco) * ((size_t) d->max_runs + 1)));
    for (int p = 0; p < 2; p++) {
        KG_HIP(hipMalloc((void **) &d->d_wgtot[p], sizeof(ddc_state4) * 2 * (size_t) nchan * DDC_MAX_GX));
        KG_HIP(hipMalloc((void **) &d->d_wgbase[p], sizeof(ddc_state4) * 2 * (size_t) nchan * DDC_MAX_GX));
        KG_HIP(hipMalloc((void **) &d->d_wgtau[p], sizeof(u32) * 2 * (size_t) nchan * DDC_MAX_GX));
    }
    d->endco_n = -1; d->endco_L = 0; d->endco_runs = 0;
    KG_HIP(hipMalloc((void **) &d->d_aggs, sizeof(ddc_chunk_agg) * 2 * (size_t) nchan * DDC_SCAN_MAX_CHUNKS));
    KG_HIP(hipMemset(d->d_aggs, 0, sizeof(ddc_chunk_agg) * 2 * (size_t) nchan * DDC_SCAN_MAX_CHUNKS));
    KG_HIP(hipMalloc((void **) &d->d_ticket, sizeof(u32)));
    KG_HIP(hipMemset(d->d_ticket, 0, sizeof(u32)));
    d->ticket_base = 0; d->epoch = 0;
    // NCO table: ONE 16-bit sine table, sin(a) = T[a], cos(a) = T[a + 2048] (kg_common.h, kg_nco_table_build)
    std::vector<short> tab(DDC_TAB);
    kg_nco_table_build(tab.data());
    KG_HIP(hipMemcpy(d->d_nco, tab.data(), sizeof(short) * DDC_TAB, hipMemcpyHostToDevice));
    *out = d;
    return KG_OK;
}

void kg_ddc_destroy(kg_ddc *d)
{
    if (!d) return;
    (void) hipSetDevice(d->ctx->device);
    (void) hipStreamSynchronize(d->ctx->stream);
    if (d->side) (void) hipStreamSynchronize(d->side);
    if (d->tail) (void) hipStreamSynchronize(d->tail);
    (void) hipFree(d->d_chans); (void) hipFree(d->d_nco);
    for (int p = 0; p < 2; p++) { (void) hipFree(d->d_local[p]); (void) hipFree(d->d_tau[p]); (void) hipFree(d->d_c0rel[p]); }
    (void) hipFree(d->d_hist);
    (void) hipFree(d->d_aggs); (void) hipFree(d->d_ticket); (void) hipFree(d->d_endco);
    for (int p = 0; p < 2; p++) { (void) hipFree(d->d_wgtot[p]); (void) hipFree(d->d_wgbase[p]); (void) hipFree(d->d_wgtau[p]); }
    kg_stage_cache_free(&d->pack_cache);
    if (d->side) { (void) hipEventDestroy(d->ev_fork); (void) hipEventDestroy(d->ev_join); if (!d->side_borrowed) kg_stream_put(d->ctx->device, d->side); }
    if (d->tail) {
        (void) hipEventDestroy(d->ev_runs); (void) hipEventDestroy(d->ev_tail[0]); (void) hipEventDestroy(d->ev_tail[1]);
        (void) hipEventDestroy(d->ev_adc);
        kg_stream_put(d->ctx->device, d->tail);
    }
    delete d;
}

// Everything the object has in flight, on all of its streams.
static int ddc_sync_all(kg_ddc *d)
{
    KG_HIP(hipStreamSynchronize(d->ctx->stream));
    if (d->side) KG_HIP(hipStreamSynchronize(d->side));
    if (d->tail) KG_HIP(hipStreamSynchronize(d->tail));
    d->tail_unjoined = false;
    return KG_OK;
}

static inline u64 ddc_cur_phase(const kg_ddc *d, int ch)
{
    const ddc_chan &c = d->h_chans[ch];
    return (c.phase + d->h_pushed[ch] * c.phase_inc) & ((1ull << 48) - 1);
}
static inline u32 ddc_cur_cnt(const kg_ddc *d, int ch)
{
    const ddc_chan &c = d->h_chans[ch];
    return (u32) (((u64) c.sample_no + d->h_pushed[ch]) & ((1ull << c.log2r) - 1));
}

// The host copy of the scalar fields -> the device record (the filter state the host copy holds with it: zero after
// set_wf / reset).  Drains the object first.
static int ddc_upload(kg_ddc *d, int ch)
{
    int rc = ddc_sync_all(d);
    if (rc) return rc;
    KG_HIP(hipMemcpy(d->d_chans + ch, &d->h_chans[ch], sizeof(ddc_chan), hipMemcpyHostToDevice));
    return KG_OK;
}

// Move channel ch's reference point to "now" (phase and counter as they stand, pushed = 0) WITHOUT touching its filter
// state: only the two scalar fields go to the device.  Drains the object first.
static int ddc_rebase(kg_ddc *d, int ch)
{
    if (d->h_pushed[ch] == 0) return KG_OK;
    int rc = ddc_sync_all(d);
    if (rc) return rc;
    ddc_chan &c = d->h_chans[ch];
    c.phase = ddc_cur_phase(d, ch);
    c.sample_no = ddc_cur_cnt(d, ch);
    d->h_pushed[ch] = 0;
    KG_HIP(hipMemcpy(&d->d_chans[ch].phase, &c.phase, sizeof c.phase, hipMemcpyHostToDevice));
    KG_HIP(hipMemcpy(&d->d_chans[ch].sample_no, &c.sample_no, sizeof c.sample_no, hipMemcpyHostToDevice));
    return KG_OK;
}

int kg_ddc_set_wf(kg_ddc *d, int ch, uint64_t phase_inc, int decim)
{
    KG_REQUIRE(d != nullptr, KG_ERR_INVALID, "kg_ddc_set_wf: null argument");
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    KG_REQUIRE(ch >= 0 && ch < d->nchan, KG_ERR_INVALID, "kg_ddc_set_wf: channel %d (0..%d)", ch, d->nchan - 1);
    int log2r = -1;
    for (int k = 0; k <= 13; k++) if (decim == (1 << k)) log2r = k;
    KG_REQUIRE(log2r >= 0, KG_ERR_INVALID, "kg_ddc_set_wf: decimation %d is not 1, 2, 4 .. 8192", decim);
    // changing the rate or frequency resets the sampler, as sample_wf() does (CmdWFReset, :1005)
    ddc_chan &c = d->h_chans[ch];
    memset(&c, 0, sizeof c);
    c.phase_inc = phase_inc & ((1ull << 48) - 1);
    c.log2r = log2r;
    c.active = 1;
    d->h_pushed[ch] = 0;
    d->h_stale[ch] = 0;
    return ddc_upload(d, ch);
}

int kg_ddc_reset_wf(kg_ddc *d, int ch)
{
    KG_REQUIRE(d != nullptr, KG_ERR_INVALID, "kg_ddc_reset_wf: null argument");
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    KG_REQUIRE(ch >= 0 && ch < d->nchan && d->h_chans[ch].active, KG_ERR_INVALID,
               "kg_ddc_reset_wf: channel %d is not configured", ch);
    // rst_wf_samp_wr (waterfall_1cic.v:47): CIC registers and the decimation counter; the NCO
    // phase keeps running in the FPGA: reference phase + what has been pushed since
    ddc_chan &c = d->h_chans[ch];
    const u64 inc = c.phase_inc, ph = ddc_cur_phase(d, ch); const int l2 = c.log2r;
    memset(&c, 0, sizeof c);
    c.phase = ph; c.phase_inc = inc; c.log2r = l2; c.active = 1;
    d->h_pushed[ch] = 0;
    d->h_stale[ch] = 0;
    return ddc_upload(d, ch);
}

int kg_ddc_set_phase(kg_ddc *d, int ch, uint64_t phase)
{
    KG_REQUIRE(d != nullptr && ch >= 0 && ch < d->nchan && d->h_chans[ch].active, KG_ERR_INVALID,
               "kg_ddc_set_phase: channel %d is not configured", ch);
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    if ((rc = ddc_rebase(d, ch))) return rc;            // the counter's reference moves to "now" with the phase's
    if ((rc = ddc_sync_all(d))) return rc;
    ddc_chan &c = d->h_chans[ch];
    c.phase = phase & ((1ull << 48) - 1);
    KG_HIP(hipMemcpy(&d->d_chans[ch].phase, &c.phase, sizeof c.phase, hipMemcpyHostToDevice));
    return KG_OK;
}

// Number of IQ pairs channel ch will produce for the next n samples.
long kg_ddc_wf_outputs(kg_ddc *d, int ch, size_t n)
{
    if (!d || ch < 0 || ch >= d->nchan || !d->h_chans[ch].active) return KG_ERR_INVALID;
    // a channel a capture left behind is reset by its next continuous push: the counter starts that block at zero
    if (d->h_stale[ch]) return (long) ((u64) n >> d->h_chans[ch].log2r);
    return (long) (((u64) ddc_cur_cnt(d, ch) + (u64) n) >> d->h_chans[ch].log2r);
}

// Deferred output stage.  Off (the default): kg_ddc_wf_push_dev leaves everything it enqueues ordered on the context's
// stream -- the caller's next enqueue on that stream sees the outputs.  On: the push returns with its output stage
// (R = 1 bypass channels, run-total prefix, combs) on a stream of the object; the context's stream carries only pass A,
// the state scan and pass B, so that the NEXT push's run passes start while this push's combs are still writing.  Whoever
// consumes the outputs first makes ITS stream wait for them with kg_ddc_wf_join (null: the context's stream).
int kg_ddc_wf_set_deferred(kg_ddc *d, int on)
{
    KG_REQUIRE(d != nullptr, KG_ERR_INVALID, "kg_ddc_wf_set_deferred: null argument");
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    if ((rc = ddc_sync_all(d))) return rc;
    if (on && !d->tail) {
        if ((rc = kg_stream_get(d->ctx->device, &d->tail))) return rc;
        KG_HIP(hipEventCreateWithFlags(&d->ev_runs, hipEventDisableTiming));
        KG_HIP(hipEventCreateWithFlags(&d->ev_tail[0], hipEventDisableTiming));
        KG_HIP(hipEventCreateWithFlags(&d->ev_tail[1], hipEventDisableTiming));
        KG_HIP(hipEventCreateWithFlags(&d->ev_adc, hipEventDisableTiming));
    }
    d->deferred = on != 0;
    return KG_OK;
}

int kg_ddc_wf_join(kg_ddc *d, void *stream)
{
    KG_REQUIRE(d != nullptr, KG_ERR_INVALID, "kg_ddc_wf_join: null argument");
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    const int last = d->parity ^ 1;                     // buffer set of the last push
    if (!d->tail || !d->tail_rec[last]) return KG_OK;   // nothing deferred is outstanding
    hipStream_t st = stream ? (hipStream_t) stream : d->ctx->stream;
    KG_HIP(hipStreamWaitEvent(st, d->ev_tail[last], 0));
    if (st == d->ctx->stream) d->tail_unjoined = false;
    return KG_OK;
}

// The NEXT push's writers of the caller's output rows (bypass kernel, combs) will not start before `event` (a
// hipEvent_t recorded by the caller behind its last reader of those rows): the write-after-read edge of a caller that
// reads push k's rows on its own stream while push k + 1 is already running.  Consumed by that push.
int kg_ddc_wf_tail_after(kg_ddc *d, void *event)
{
    KG_REQUIRE(d != nullptr, KG_ERR_INVALID, "kg_ddc_wf_tail_after: null argument");
    d->after_ev = event;
    return KG_OK;
}

// kg_ddc_wf_push_dev (max_out = 0: the continuous sampler) and kg_ddc_wf_capture_dev (max_out >= 1: CmdWFReset with
// WF_SAMP_WR_RST at the block's first sample, then the one-shot sampler of verilog/rx/iq_sampler_8k_32b.v, which stops
// when it holds max_out pairs: only the first max_out << log2r samples of the block reach a channel's filters).
// kg_ddc_wf_step_dev: the sampler mode PER ENTRY (max_out_tab[i]: 0 continuous, >= 1 reset + one-shot of that many pairs) and
// an offset into each entry's row (out_off[i] pairs): a bank of receivers in both modes in one set of launches.
static int ddc_push_impl(kg_ddc *d, const void *d_adc, size_t n, const int32_t *chan_list, int nlist,
                         void *d_out, size_t out_stride, size_t max_out, int64_t *nouts,
                         const int64_t *max_out_tab = nullptr, const int64_t *out_off = nullptr)
{
    KG_REQUIRE(d && d_adc && chan_list && d_out, KG_ERR_INVALID, "kg_ddc_wf_push_dev: null argument");
    auto cap_of = [&](int i) -> u64 { return max_out_tab ? (u64) max_out_tab[i] : (u64) max_out; };     // 0: continuous sampler
    bool capture = false;                                           // some entry of the call captures
    for (int i = 0; i < nlist; i++) {
        KG_REQUIRE(!max_out_tab || (max_out_tab[i] >= 0 && max_out_tab[i] <= ((int64_t) 1 << 31)), KG_ERR_INVALID,
                   "kg_ddc_wf_step_dev: max_out[%d] = %lld", i, (long long) max_out_tab[i]);
        KG_REQUIRE(!out_off || (out_off[i] >= 0 && (size_t) out_off[i] <= out_stride), KG_ERR_INVALID,
                   "kg_ddc_wf_step_dev: out_off[%d] = %lld outside the row", i, out_off ? (long long) out_off[i] : 0ll);
        capture = capture || cap_of(i) != 0;
    }
    int rc = kg_ctx_use(d->ctx);
    if (rc) return rc;
    KG_REQUIRE(n >= 1 && (long) n <= d->max_samples, KG_ERR_INVALID, "kg_ddc_wf_push_dev: n %zu (max %ld)", n, d->max_samples);
    KG_REQUIRE(nlist >= 1 && nlist <= d->nchan, KG_ERR_INVALID, "kg_ddc_wf_push_dev: nlist %d", nlist);
    KG_REQUIRE(((uintptr_t) d_adc & 1) == 0 && ((uintptr_t) d_out & 3) == 0, KG_ERR_INVALID,
               "kg_ddc_wf_push_dev: misaligned pointer");
    {
        std::vector<char> &seen = d->seen;                          // (a flag per channel: no quadratic search of the list)
        seen.assign(d->nchan, 0);
        for (int i = 0; i < nlist; i++) {
            const int ch = chan_list[i];
            KG_REQUIRE(ch >= 0 && ch < d->nchan && d->h_chans[ch].active, KG_ERR_STATE,
                       "kg_ddc_wf_push_dev: channel %d is not configured", ch);
            KG_REQUIRE(!seen[ch], KG_ERR_INVALID, "kg_ddc_wf_push_dev: channel %d listed twice", ch);
            seen[ch] = 1;
        }
    }
    // A channel a capture left behind holds the filter state of a block cut short: the next CONTINUOUS push starts it
    // from the reset state, as the reference does when it changes sampler mode (CmdWFReset with WF_SAMP_CONTIN,
    // rx/rx_waterfall.cpp:971-978).  Synchronising, rare.  (Before the ages are compared: a reset moves the channel's
    // reference point.)
    // (A bank's PLAN pass changes nothing: it computes with the counts the reset WILL leave and the replay pass resets.)
    const bool plan_pass = d->ctx->arena && d->ctx->arena->mode == KG_ARENA_PLAN;
    std::vector<char> will_reset(nlist, 0);
    for (int i = 0; i < nlist; i++)
            if (cap_of(i) == 0 && d->h_stale[chan_list[i]]) {
                if (plan_pass) { will_reset[i] = 1; continue; }
                if ((rc = kg_ddc_reset_wf(d, chan_list[i]))) return rc;
                d->h_stale[chan_list[i]] = 0;
            }
    // Channels whose reference points differ in age (one was retuned -- the most common command of a connection -- or joined
    // later, or was left out of earlier calls): the call passes the YOUNGEST age as `pushed` and a table of what each entry has
    // on top of it.  The table is the same from push to push while no channel is touched (every age grows by n), so it rides in
    // the content-cached per-call tables; rounds 4-5 re-based every other channel of the call instead: two blocking copies per
    // channel, ~2000 for a bank of 1024 receivers after ONE `SET zoom=`.
    auto age_of = [&](int i) -> u64 { return will_reset[i] ? 0ull : d->h_pushed[chan_list[i]]; };
    u64 pushed = age_of(0), oldest = pushed;
    for (int i = 1; i < nlist; i++) {
        const u64 a = age_of(i);
        if (a < pushed) pushed = a;
        if (a > oldest) oldest = a;
    }
    // (the kernels form (age + sample index) x phase_inc in 64 bits mod 2^48: re-base long before anything wraps -- 2^62 samples)
    KG_REQUIRE(!(plan_pass && oldest + (u64) n >= (1ull << 62)), KG_ERR_STATE, "kg_ddc_wf_push_dev: 2^62 samples since a channel was last set");
    if (oldest + (u64) n >= (1ull << 62)) {
        for (int i = 0; i < nlist; i++) if ((rc = ddc_rebase(d, chan_list[i]))) return rc;
        pushed = 0;
    }
    std::vector<u64> h_pdelta(nlist);
    for (int i = 0; i < nlist; i++) h_pdelta[i] = age_of(i) - pushed;
    std::vector<long> h_nouts(nlist), h_off(nlist), h_nlim(nlist), h_outoff(nlist);
    std::vector<int> h_wg(nlist + 1), h_bypass, h_run, h_small, h_rest, h_big, h_reset(nlist);
    long max_nout = 0, c0_need = 0, comb_wgs = 0, n_run_max = 0, n_by_max = 0, n_run_sum = 0;
    for (int i = 0; i < nlist; i++) {
        const int ch = chan_list[i];
        const ddc_chan &c = d->h_chans[ch];
        // samples of the block this channel's filters see: all of them, or (capture) what fills the one-shot sampler
        const bool cap = cap_of(i) != 0;
        const u64 want = cap ? cap_of(i) << c.log2r : (u64) n;
        h_reset[i] = cap ? 1 : 0;
        h_outoff[i] = out_off ? (long) out_off[i] : 0l;
        h_nlim[i] = (long) (want < (u64) n ? want : (u64) n);
        h_nouts[i] = (long) (((cap || will_reset[i] ? 0ull : (u64) ddc_cur_cnt(d, ch)) + (u64) h_nlim[i]) >> c.log2r);
        if (c.log2r) { if (h_nlim[i] > n_run_max) n_run_max = h_nlim[i]; n_run_sum += h_nlim[i]; }
        else if (h_nlim[i] > n_by_max) n_by_max = h_nlim[i];
        KG_REQUIRE((size_t) (h_nouts[i] + h_outoff[i]) <= out_stride, KG_ERR_INVALID,
                   "kg_ddc_wf_push_dev: out_stride %zu < %ld outputs of channel %d (at offset %ld)", out_stride, h_nouts[i], ch, h_outoff[i]);
        if (h_nouts[i] > max_nout) max_nout = h_nouts[i];
        if (nouts) nouts[i] = h_nouts[i];
        h_off[i] = c0_need;
        if (c.log2r) c0_need += 2 * ((h_nouts[i] + 3) & ~3l);             // I and Q planes, each a multiple of 16 bytes
        h_wg[i] = (int) comb_wgs;
        if (c.log2r) comb_wgs += (h_nouts[i] + DDC_COMB_TILE - 1) / DDC_COMB_TILE;
        else h_bypass.push_back(i);
        if (c.log2r) { h_run.push_back(i); (c.log2r <= 3 ? h_small : (c.log2r <= 8 ? h_rest : h_big)).push_back(i); }
    }
    const size_t nmid = h_rest.size();            // h_rest = the 16 <= R <= 256 entries, then the R >= 512 ones
    h_rest.insert(h_rest.end(), h_big.begin(), h_big.end());
    h_wg[nlist] = (int) comb_wgs;
    KG_REQUIRE(comb_wgs < (1l << 31), KG_ERR_INVALID, "kg_ddc_wf_push_dev: too many outputs in one call");
    hipStream_t st = d->ctx->stream;
    const bool defer = d->deferred;
    const int par = d->parity;                     // this push's buffer set
    // The buffer set was last used two pushes ago: its output stage must have drained it (deferred mode; long done).
    if (d->tail && d->tail_rec[par]) {
        KG_HIP(hipStreamWaitEvent(st, d->ev_tail[par], 0));
        if (d->side) KG_HIP(hipStreamWaitEvent(d->side, d->ev_tail[par], 0));
    }
    if (c0_need > d->c0_cap[par]) {
        if ((rc = ddc_sync_all(d))) return rc;
        (void) hipFree(d->d_c0rel[par]);
        d->d_c0rel[par] = nullptr; d->c0_cap[par] = 0;  // nothing dangles if the allocation below fails
        KG_HIP(hipMalloc((void **) &d->d_c0rel[par], sizeof(u32) * (size_t) (c0_need + 16)));
        d->c0_cap[par] = c0_need;
    }
    ddc_state4 *const d_local = d->d_local[par];
    u32 *const d_c0rel = d->d_c0rel[par], *const d_tau = d->d_tau[par];
    // run length: a power of two between 64 and 8192, about 8192 runs per call
    // One thread per (channel, run): with few channels take more, shorter runs so that the two
    // run passes still put about four waves on every SIMD (14 channels on MI355X: 16384 runs
    // instead of 8192 = 2.18 -> 1.91 ms per 2^24 samples, although the state scan doubles).
    int target = DDC_TARGET_RUNS;
    {
        const long want_threads = (long) d->ctx->num_cus * 4 * 4 * 64;
        const long per_chan = (want_threads + nlist - 1) / nlist;
        if (per_chan > target) target = per_chan < d->max_runs ? (int) per_chan : d->max_runs;
    }
    if (const char *e = kg_tuning_env("KIWIGPU_DDC_RUNS")) { const int v = atoi(e); if (v >= 64 && v <= d->max_runs) target = v; }
    // the runs cover the longest share of the block any filtered channel takes (the whole block unless capturing)
    const long n_cover = n_run_max > 0 ? n_run_max : 1;
    int L = DDC_RUN_MIN;
    while (L < DDC_RUN_MAX && (long) ((n_cover + L - 1) / L) > target) L <<= 1;
    if (capture) {
        // a capture's channels take very different shares (8192 R samples each): size the runs by the TOTAL work, about
        // four waves per SIMD over all of them, inside what the longest share allows
        const long want_threads = (long) d->ctx->num_cus * 4 * 4 * 64;
        // (runs of at most 1024 samples: pass A then integrates every decimation in 64 bits -- short_zero_run in
        // ddc_wf_run_body; with SURVEY's receiver mix, 352 M sample-channels per step, the total-work rule alone chose 2048 and
        // pass A of the R >= 512 channels ran its 128-bit form: 1.13 ms per step against pass B's 0.71)
        int Lw = DDC_RUN_MIN;
        while (Lw < 1024 && n_run_sum / Lw > want_threads) Lw <<= 1;
        if (Lw > L) L = Lw;
        while (L < DDC_RUN_MAX && (n_cover + L - 1) / L > d->max_runs) L <<= 1;
    }
    const int nruns = (int) ((n_cover + L - 1) / L);
    int log2L = 0;
    while ((1 << log2L) < L) log2L++;
    KG_REQUIRE(nruns <= d->max_runs, KG_ERR_INVALID, "kg_ddc_wf_push_dev: %d runs > %d", nruns, d->max_runs);
    // The per-call tables go through the context's staging ring in one piece: no stream
    // synchronisation, and the previous call's kernels keep their own copy.
    const long *s_c0off, *s_nouts, *s_nlim, *s_outoff; const u64 *s_pdelta; const int *s_list, *s_wgoff, *s_bypass, *s_selrun, *s_selsmall, *s_selrest, *s_reset;
    {
        std::vector<unsigned char> pack;
        auto put = [&](const void *src, size_t bytes) -> size_t {
            const size_t at = (pack.size() + 15) & ~(size_t) 15;
            pack.resize(at + bytes);
            if (bytes) memcpy(pack.data() + at, src, bytes);
            return at;
        };
        const size_t o_off = put(h_off.data(), sizeof(long) * nlist), o_nouts = put(h_nouts.data(), sizeof(long) * nlist);
        const size_t o_nlim = put(h_nlim.data(), sizeof(long) * nlist), o_pdelta = put(h_pdelta.data(), sizeof(u64) * nlist);
        const size_t o_list = put(chan_list, sizeof(int) * nlist);
        const size_t o_wg = put(h_wg.data(), sizeof(int) * (nlist + 1));
        const size_t o_by = put(h_bypass.data(), sizeof(int) * h_bypass.size());
        const size_t o_run = put(h_run.data(), sizeof(int) * h_run.size());
        const size_t o_small = put(h_small.data(), sizeof(int) * h_small.size());
        const size_t o_rest = put(h_rest.data(), sizeof(int) * h_rest.size());
        const size_t o_reset = put(h_reset.data(), sizeof(int) * nlist), o_outoff = put(h_outoff.data(), sizeof(long) * nlist);
        // a CHANGED table is rewritten in the order of the context's stream: the previous push's output stage, on its own
        // stream in deferred mode, may still be reading the old one -- join it first (a steady stream never gets here)
        if (d->tail && d->tail_rec[par ^ 1] && d->tail_unjoined &&
            !(d->pack_cache.bytes == pack.size() && d->pack_cache.host && memcmp(d->pack_cache.host, pack.data(), pack.size()) == 0)) {
            KG_HIP(hipStreamWaitEvent(st, d->ev_tail[par ^ 1], 0));
            d->tail_unjoined = false;
        }
        void *base = nullptr;
        if ((rc = kg_ctx_stage_cached(d->ctx, &d->pack_cache, pack.data(), pack.size(), &base))) return rc;
        const unsigned char *b = (const unsigned char *) base;
        s_c0off = (const long *) (b + o_off); s_nouts = (const long *) (b + o_nouts); s_nlim = (const long *) (b + o_nlim);
        s_list = (const int *) (b + o_list); s_wgoff = (const int *) (b + o_wg);
        s_bypass = (const int *) (b + o_by); s_selrun = (const int *) (b + o_run);
        s_selsmall = (const int *) (b + o_small); s_selrest = (const int *) (b + o_rest);
        s_reset = (const int *) (b + o_reset); s_outoff = (const long *) (b + o_outoff); s_pdelta = (const u64 *) (b + o_pdelta);
    }
    KG_PLAN_ONLY(d->ctx);
    // The object's second stream: work that does not depend on the pass A -> scan -> pass B chain runs beside
    // it -- the R = 1 bypass channels (no filter state at all) from the start, pass B of the R <= 8 channels
    // (below) -- and joins before the call's last kernels.  KIWIGPU_DDC_SIDE=0: everything in line.
    bool side_on = true, side_used = false;
    if (const char *e = kg_tuning_env("KIWIGPU_DDC_SIDE")) side_on = atoi(e) != 0;
    auto side_ready = [&]() -> int {
        if (!d->side) {
            { const int rc_ = kg_stream_get(d->ctx->device, &d->side); if (rc_) return rc_; }
            KG_HIP(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
            KG_HIP(hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming));
        }
        return KG_OK;
    };
    // Deferred mode: the output stage's stream starts behind the tables and the caller's samples (fork event), behind the
    // previous push's output stage (stream order) and behind the caller's last reader of the rows (kg_ddc_wf_tail_after).
    hipStream_t ost = st;                          // where the writers of the caller's rows run
    if (defer) {
        if ((rc = side_ready())) return rc;
        ost = d->tail;
        KG_HIP(hipEventRecord(d->ev_fork, st));
        KG_HIP(hipStreamWaitEvent(ost, d->ev_fork, 0));
        if (d->after_ev) KG_HIP(hipStreamWaitEvent(ost, (hipEvent_t) d->after_ev, 0));
    } else if (d->after_ev) {
        KG_HIP(hipStreamWaitEvent(st, (hipEvent_t) d->after_ev, 0));
    }
    d->after_ev = nullptr;
    if (!h_bypass.empty()) {
        hipStream_t bst = ost;
        if (!defer && side_on && !h_run.empty()) {
            if ((rc = side_ready())) return rc;
            KG_HIP(hipEventRecord(d->ev_fork, st));                  // behind the staged tables
            KG_HIP(hipStreamWaitEvent(d->side, d->ev_fork, 0));
            bst = d->side; side_used = true;
        }
        const long nblk_by = (n_by_max + BYP_G * 1024 - 1) / (BYP_G * 1024), cap_by = (long) d->ctx->num_cus * 4;   // LDS: 20 KiB each, 4 per CU
        hipLaunchKernelGGL(ddc_wf_bypass_kernel, dim3((unsigned) (nblk_by < cap_by ? nblk_by : cap_by)), dim3(256),
                           0, bst, (const short *) d_adc, (long) n_by_max, (const ddc_chan *) d->d_chans, s_list,
                           s_bypass, (int) h_bypass.size(), (const u32 *) d->d_nco, (short2 *) d_out, (long) out_stride, pushed, s_pdelta, s_nlim, s_outoff, d->ctx->rows_by_chan);
        KG_HIP(hipGetLastError());
        if (defer) KG_HIP(hipEventRecord(d->ev_adc, bst));           // the output stream's only reader of d_adc
    }
    // end-referred carry states (round 4): prefix sums inside pass A's workgroups + one wave per (channel, I/Q) over the
    // workgroup totals; KIWIGPU_DDC_ENDREF=0: run-start states and the chunked affine scan -- the A/B reference
    int endref = 2;
    if (const char *e = kg_tuning_env("KIWIGPU_DDC_ENDREF")) endref = atoi(e) != 0 ? 2 : 0;
    if (endref && !h_run.empty() && (d->endco_n != n_cover || d->endco_L != L || d->endco_runs != nruns)) {
        // (a changed block length: rare.  The previous push's pass B may still be reading the old table on the object's
        // second stream)
        if ((rc = ddc_sync_all(d))) return rc;
        hipLaunchKernelGGL(ddc_endco_kernel, dim3((unsigned) ((nruns + 1 + 255) / 256)), dim3(256), 0, st, d->d_endco, n_cover, L, nruns);
        KG_HIP(hipGetLastError());
        d->endco_n = n_cover; d->endco_L = L; d->endco_runs = nruns;
    }
    const unsigned gx = (unsigned) ((nruns + DDC_THREADS - 1) / DDC_THREADS);
    if (!h_run.empty()) {
        // (runs of at most 1024 samples: every decimation integrates from zero in 64 bits -- the narrow form alone)
        auto pass_a = (L <= 1024 && !DDC_ONE_FORM) ? ddc_wf_run_kernel<false, DDC_NARROW> : ddc_wf_run_kernel<false, DDC_ALL>;
        hipLaunchKernelGGL(pass_a, dim3(gx, (unsigned) h_run.size()), dim3(DDC_THREADS), 0, st,
                           (const short *) d_adc, (long) n, L, nruns, (const ddc_chan *) d->d_chans, s_list,
                           (const u32 *) d->d_nco, d_local, d_c0rel, d_tau, s_c0off,
                           s_nouts, s_selrun, 0, pushed, s_pdelta, s_nlim, s_reset, endref, (const ddc_endco *) d->d_endco, d->endco_n,
                           d->d_wgtot[par], (const ddc_state4 *) nullptr, (u32 *) nullptr);
        KG_HIP(hipGetLastError());
    }
    {
        // chunks per (channel, I/Q): as many as keep every chunk at least a workgroup's worth of runs and
        // the whole grid resident at once
        const int npairs = 2 * nlist;
        int nchunk = (d->ctx->num_cus * 2) / npairs;
        if (nchunk > DDC_SCAN_MAX_CHUNKS) nchunk = DDC_SCAN_MAX_CHUNKS;
        while (nchunk > 1 && nruns / nchunk < 64 * DDC_SCAN_WAVES) nchunk--;
        if (nchunk < 1) nchunk = 1;
        // the advance coefficients of the scan's regular steps (sc_tab): a full lane's samples u << 0 .. 8, a full
        // chunk's v << 0 .. 3
        sc_tab tab;
        {
            const long cper = (nruns + nchunk - 1) / nchunk, per = (cper + 64 * DDC_SCAN_WAVES - 1) / (64 * DDC_SCAN_WAVES);
            const u64 u = (u64) per * (u64) L, v = (u64) cper * (u64) L;
            for (int j = 0; j < DDC_SCAN_TAB; j++) {
                const u64 len = j < 9 ? u << j : v << (j - 9);
                const unsigned __int128 l = len;
                // sc_coef keeps c2 = len (len + 1) / 2 in 64 bits (host table and sc_coef_for alike): exact while len < 2^32.
                // The longest advance is v << 3 <= 8 (n + nchunk L) with n <= max_runs x DDC_RUN_MAX = 2^27 samples per push.
                static_assert((unsigned long long) DDC_TARGET_RUNS * 2 * DDC_RUN_MAX * 16 <= (1ull << 32),
                              "sc_coef.c2 is 64 bits wide: an advance of 2^32 samples or more needs a u96 there");
                KG_REQUIRE(len < (1ull << 32), KG_ERR_INVALID, "kg_ddc_wf_push_dev: scan advance %llu too long for the 64-bit c2",
                           (unsigned long long) len);
                const unsigned __int128 c2 = l * (l + 1) / 2;                     // len < 2^32: below 2^63
                // len (len + 1) (len + 2) / 6 mod 2^96: c2 (len + 2) is 3 x the binomial; divide the exact product
                // (below 2^111) by 3
                const unsigned __int128 c3 = c2 * (l + 2) / 3;
                tab.len[j] = len;
                tab.c[j].L = len; tab.c[j].c2 = (u64) c2;
                tab.c[j].c3.w[0] = (u32) c3; tab.c[j].c3.w[1] = (u32) (c3 >> 32); tab.c[j].c3.w[2] = (u32) (c3 >> 64);
            }
        }
        if (endref == 2)
            hipLaunchKernelGGL(ddc_wf_scan_wg_kernel, dim3((unsigned) npairs), dim3(64), 0, st, (const ddc_state4 *) d->d_wgtot[par],
                               d->d_wgbase[par], (int) gx, d->d_chans, s_list, s_nlim, s_reset);
        else
            hipLaunchKernelGGL(ddc_wf_scan_states_kernel, dim3((unsigned) (npairs * nchunk)), dim3(64 * DDC_SCAN_WAVES), 0, st,
                               d_local, (long) n, L, nruns, d->d_chans, s_list, npairs, nchunk, d->d_aggs, d->d_ticket,
                               d->ticket_base, d->epoch + 1, tab, s_nlim, s_reset);
        KG_HIP(hipGetLastError());
        // only a launch that was accepted advances the ticket counter and publishes under the new epoch
        if (endref != 2) {
            d->epoch++;
            d->ticket_base += (u32) (npairs * nchunk);
        }
    }
    // Pass B.  The staged strobe flush (R <= 8) needs 34 KiB more LDS, hence its own launch.  Neither
    // launch fills the GPU by itself (a lane walks a whole run: 64 workgroups per channel), so when both
    // kinds of channel are present the staged launch runs beside the other one on a second stream of the
    // object (fork after the state scan, join before the run-total prefix); with only small decimations
    // it pays once those channels fill the GPU by themselves, otherwise the scattered stores are cheaper
    // than the smaller grid.
    bool staged = (long) h_small.size() * nruns >= (long) d->ctx->num_cus * 4 * 2 * 64;
    const bool beside = side_on && !h_small.empty() && !h_rest.empty();
    if (const char *e = kg_tuning_env("KIWIGPU_DDC_STAGED")) staged = atoi(e) != 0 && !h_small.empty();
    auto pass_b = [&](hipStream_t s, size_t nwhich, const int *sel, int stage, int mode) {
        auto k = mode == DDC_NARROW ? ddc_wf_run_kernel<true, DDC_NARROW> : (mode == DDC_WIDE ? ddc_wf_run_kernel<true, DDC_WIDE> : ddc_wf_run_kernel<true, DDC_ALL>);
        hipLaunchKernelGGL(k, dim3(gx, (unsigned) nwhich), dim3(DDC_THREADS), stage, s,
                           (const short *) d_adc, (long) n, L, nruns, (const ddc_chan *) d->d_chans, s_list,
                           (const u32 *) d->d_nco, d_local, d_c0rel, d_tau, s_c0off, s_nouts, sel, stage, pushed, s_pdelta, s_nlim, s_reset, endref, (const ddc_endco *) d->d_endco, d->endco_n,
                           (ddc_state4 *) nullptr, (const ddc_state4 *) d->d_wgbase[par], d->d_wgtau[par]);
    };
    // The entries above R = 8 by the form they need (DDC_NARROW / DDC_WIDE: fewer registers, more waves per SIMD) -- as two
    // launches when each of them fills the GPU by itself (a bank of receivers), as one launch of the general form otherwise
    // (configs[2]'s five and three such channels: two half-empty launches one after the other would lose more than the
    // waves gain).
    const long fill = (long) d->ctx->num_cus * 4 * 2 * 64;
    const size_t nbig = h_rest.size() - nmid;
    const bool split = !DDC_ONE_FORM && (nmid == 0 || (long) nmid * nruns >= fill) && (nbig == 0 || (long) nbig * nruns >= fill);
    auto pass_b_rest = [&](hipStream_t s) {
        if (h_rest.empty()) return;
        if (!split) { pass_b(s, h_rest.size(), s_selrest, 0, DDC_ALL); return; }
        if (nmid) pass_b(s, nmid, s_selrest, 0, DDC_NARROW);
        if (nbig) pass_b(s, nbig, s_selrest + nmid, 0, DDC_WIDE);
    };
    if (beside) {
        if ((rc = side_ready())) return rc;
        side_used = true;
        KG_HIP(hipEventRecord(d->ev_fork, st));
        KG_HIP(hipStreamWaitEvent(d->side, d->ev_fork, 0));
        pass_b(d->side, h_small.size(), s_selsmall, DDC_STAGE_BYTES, DDC_ALL);
        KG_HIP(hipGetLastError());
        pass_b_rest(st);
        KG_HIP(hipGetLastError());
    } else if (staged) {
        pass_b(st, h_small.size(), s_selsmall, DDC_STAGE_BYTES, DDC_ALL);
        KG_HIP(hipGetLastError());
        pass_b_rest(st);
        KG_HIP(hipGetLastError());
    } else if (!h_small.empty()) {
        pass_b(st, h_run.size(), s_selrun, 0, DDC_ALL);           // a few R <= 8 entries, not staged: everything in one launch
        KG_HIP(hipGetLastError());
    } else if (!h_run.empty()) {
        pass_b_rest(st);
        KG_HIP(hipGetLastError());
    }
    // The output stage: run-total prefix, combs, comb history -- in line on the context's stream, or (deferred) on the
    // object's output stream behind this push's run passes, with the context's stream free for the next push's.
    if (defer) {
        KG_HIP(hipEventRecord(d->ev_runs, st));
        KG_HIP(hipStreamWaitEvent(ost, d->ev_runs, 0));
    }
    if (side_used) {                              // everything the second stream did is behind this point of the output stage
        KG_HIP(hipEventRecord(d->ev_join, d->side));
        KG_HIP(hipStreamWaitEvent(ost, d->ev_join, 0));
    }
    if (endref == 2)
        hipLaunchKernelGGL(ddc_wf_tau_wg_kernel, dim3(2 * nlist), dim3(64), 0, ost, d->d_wgtau[par], (int) gx, d->d_chans, s_list,
                           s_reset);
    else
        hipLaunchKernelGGL(ddc_wf_scan_tau_kernel, dim3(2 * nlist), dim3(64 * DDC_SCAN_WAVES), 0, ost, d_tau, nruns, d->d_chans,
                           s_list, s_nlim, L, s_reset);
    KG_HIP(hipGetLastError());
    if (comb_wgs > 0) {
        hipLaunchKernelGGL(ddc_wf_comb_kernel, dim3((unsigned) comb_wgs), dim3(256), 0, ost,
                           (const u32 *) d_c0rel, (const u32 *) d_tau, log2L, nruns, s_c0off,
                           (const ddc_chan *) d->d_chans, s_list, s_nouts,
                           pushed, s_pdelta, s_reset, s_wgoff, nlist, (short2 *) d_out, (long) out_stride, s_outoff,
                           d->d_hist, endref == 2 ? (const u32 *) d->d_wgtau[par] : (const u32 *) nullptr, (int) gx, d->ctx->rows_by_chan);
        KG_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(ddc_wf_finish_kernel, dim3((nlist + 63) / 64), dim3(64), 0, ost, d->d_chans,
                       s_list, nlist, (long) n, s_nouts, (const u32 *) d->d_hist, s_reset);
    KG_HIP(hipGetLastError());
    if (d->tail) {                                // (also in line: a later deferred push finds its buffer set covered)
        KG_HIP(hipEventRecord(d->ev_tail[par], ost));
        d->tail_rec[par] = true;
        d->tail_unjoined = defer;
    }
    if (defer) {
        // Everything that READS the caller's samples is ordered in front of whatever the caller enqueues next on the
        // context's stream (a streaming caller refills or recycles d_adc there): the run passes are on that stream already;
        // the bypass kernel (output stream) and pass B of the small decimations (second stream) are joined here.  Only the
        // readers: the output stage proper (run-total prefix, combs) stays free to run beside the next push.
        if (!h_bypass.empty()) KG_HIP(hipStreamWaitEvent(st, d->ev_adc, 0));
        if (side_used) KG_HIP(hipStreamWaitEvent(st, d->ev_join, 0));
    }
    d->parity = par ^ 1;
    for (int i = 0; i < nlist; i++) {
        d->h_pushed[chan_list[i]] += (u64) n;                      // the NCO runs through the whole block either way
        if (cap_of(i) != 0) d->h_stale[chan_list[i]] = 1;          // filters stopped short of the block's end
    }
    return KG_OK;
}

int kg_ddc_wf_push_dev(kg_ddc *d, const void *d_adc, size_t n, const int32_t *chan_list, int nlist,
                       void *d_out, size_t out_stride, int64_t *nouts)
{
    return ddc_push_impl(d, d_adc, n, chan_list, nlist, d_out, out_stride, 0, nouts);
}

int kg_ddc_wf_capture_dev(kg_ddc *d, const void *d_adc, size_t n, const int32_t *chan_list, int nlist,
                          void *d_out, size_t out_stride, size_t max_out, int64_t *nouts)
{
    KG_REQUIRE(max_out >= 1 && max_out <= ((size_t) 1 << 31), KG_ERR_INVALID, "kg_ddc_wf_capture_dev: max_out %zu", max_out);
    return ddc_push_impl(d, d_adc, n, chan_list, nlist, d_out, out_stride, max_out, nouts);
}

int kg_ddc_wf_step_dev(kg_ddc *d, const void *d_adc, size_t n, const int32_t *chan_list, int nlist, void *d_out,
                       size_t out_stride, const int64_t *out_off, const int64_t *max_out, int64_t *nouts)
{
    KG_REQUIRE(max_out != nullptr, KG_ERR_INVALID, "kg_ddc_wf_step_dev: max_out is null");
    return ddc_push_impl(d, d_adc, n, chan_list, nlist, d_out, out_stride, 0, nouts, max_out, out_off);
}

}  // extern "C"
