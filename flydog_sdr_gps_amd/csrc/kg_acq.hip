// kg_acq.hip -- GPS C/A + Galileo E1B parallel-code-phase acquisition on gfx950.
//
// Replaces gps/search.cpp of the reference: the SearchInit() code-table build
// (:243-285, :309-346), Sample() (:382-449) and Correlate() (:453-499).
//
// Shape.  The reference is one fixed shape (gps/gps.h:62-73): NSAMPLES 65536 input
// samples, DECIM 4, FFT_LEN N = 16384.  Here the shape is a property of the kg_acq
// object: `nsamples` input samples carry signal (the rest of the 4*N-sample array is
// zero, as DecimateBy2float's zero tail, :145), N = P * 4096 with P = 4 (reference)
// or P = 16 (BASELINE configs[4]: 10 ms coherent = 163680 samples, N = 65536).  The
// text below is written for P = 4; for P = 16 read "16 planes / 16 sub-transforms".
//
// Data layout in HBM (private to this file): every N-bin spectrum is kept
// "residue-major": plane r (0..P-1) holds bins k = P*k1 + r.  Inside a plane,
// element k1 = t + 256 j (the value thread t feeds into leg j of its first
// radix-16) sits at row j>>1, column 2*(t + H) + (j&1): the two legs 2i, 2i+1
// of one thread are adjacent, so a thread fetches its 16 inputs with eight
// 16-byte loads that are contiguous across the wave.  Data planes have H = 0
// (8 rows x 512 = 4096 float2); code planes carry a halo of H columns on both
// sides of every row holding the wrapped neighbours, so the Doppler rotation
// code[(k - dop) mod N] (:471) is a plain column offset with no wrap logic.  The unnormalised backward transform of
// Correlate() only needs outputs n < 4092 (C/A) -- a quarter of the 16384 --
// so it is computed as
//      y[n] = sum_{k2=0..3} W_N^{n*k2} * IFFT_4096( X[4*k1 + k2] )[n],  n < 4096
// i.e. four 4096-point transforms (one 256-thread workgroup, 32 KiB of LDS)
// whose results are accumulated in registers; the 16384-point product array
// the reference materialises (rev_buf, :58) never exists.  The Doppler shift
// code[(k - dop) mod N] (:471) is, in this layout, a contiguous rotated read
// of plane (k2 - dop) & 3.  E1B (16368 outputs) keeps four accumulators per
// point, one per output quarter, in a 512-thread kernel at 8 points per thread
// (acq_correlate8_kernel).  Its operands use a second arrangement of the same
// planes, "layout B": element k1 = i + 512 j (the value thread i of 512 feeds
// into leg j of its first radix-8) at row j>>1 (of 4), column 2*(i + H) + (j&1),
// so that a thread fetches its 8 inputs with four 16-byte loads, contiguous
// across the wave.  Every data spectrum is written in both arrangements by the
// forward transform; a code spectrum in the one its kernel reads (B when the
// SV's window is longer than 4096 lags).
#include "kg_common.h"
#include "kg_fft.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

#define SUB      4096                 // points per sub-transform: N / P
#define NTAPS    31                   // gps/search.cpp:49
#define DECIM    4                    // gps/gps.h:62

// ---------------------------------------------------------------------------
// Front end: mix -> half-band /2 -> half-band /2   (Sample() / SearchInit())
// ---------------------------------------------------------------------------

enum { SRC_BITS = 0, SRC_IQ16 = 1, SRC_CHIPS = 2 };

// gps/search.cpp:101-136, column FT = 0.  Non-zero taps only: even j and the centre.
__constant__ float c_hb_even[16] = {
    -0.010233f,  0.010668f, -0.016324f,  0.024377f, -0.036482f,  0.056990f, -0.101993f,
     0.316926f,  0.316926f, -0.101993f,  0.056990f, -0.036482f,  0.024377f, -0.016324f,
     0.010668f, -0.010233f,
};
#define HB_CENTRE 0.500009f

KG_DEV float bipolar(int bit) { return bit ? -1.0f : 1.0f; }      // search.cpp:62-66

template <int SRC>
KG_DEV cf acq_source(const void *__restrict__ src, int i, int nvalid, int nchips, int boc)
{
    if (i >= nvalid) return cf{0.f, 0.f};            // DecimateBy2float zero tail, :145
    if constexpr (SRC == SRC_BITS) {
        // search.cpp:408-423: LSB-first bits, lo_sin = {1,1,0,0}, lo_cos = {1,0,0,1},
        // lo_phase advances by exactly 1.0 per sample.  :168-175: 1 -> -1.0, 0 -> +1.0.
        const uint8_t *p = (const uint8_t *) src;
        const int bit = (p[i >> 3] >> (i & 7)) & 1;
        const int ph = i & 3;
        const int ls = ph < 2, lc = (ph == 0) | (ph == 3);
        return cf{bipolar(bit ^ ls), bipolar(bit ^ lc)};
    } else if constexpr (SRC == SRC_IQ16) {
        const short2 v = ((const short2 *) src)[i];
        const float a = (float) v.x, b = (float) v.y;
        switch (i & 3) {                             // (a + jb) * (-j)^i
        case 0:  return cf{a, b};
        case 1:  return cf{b, -a};
        case 2:  return cf{-a, -b};
        default: return cf{-b, a};
        }
    } else {
        // search.cpp:250-267 / :315-329 with ca_rate = 1/16 exactly: chip index
        // i >> 4, BOC(1,1) half-chip flag = phase >= 0.5  <=>  (i & 15) >= 8.
        const uint8_t *p = (const uint8_t *) src;
        const int chip = p[(i >> 4) % nchips];
        const int b11 = boc ? ((i & 15) >= 8) : 0;
        return cf{bipolar(chip ^ b11), 0.f};
    }
}

// One half-band output, reference accumulation order (search.cpp:148-158):
// c0 term, then j = 2,4,..,30, then the centre tap.  Separate multiply and add.
KG_DEV cf hb_tap(const float2 *x)
{
    cf acc = kg_scale(kg_ld(&x[0]), c_hb_even[0]);
#pragma unroll
    for (int j = 1; j < 16; j++) acc = acc + kg_scale(kg_ld(&x[2 * j]), c_hb_even[j]);
    acc = acc + kg_scale(kg_ld(&x[(NTAPS - 1) / 2]), HB_CENTRE);
    return acc;
}

#define FE_TILE 256                          // stage-2 outputs per workgroup
#define FE_NY1  (2 * FE_TILE + NTAPS - 2)    // 541 stage-1 outputs needed
#define FE_NX   (2 * FE_NY1 + NTAPS - 2)     // 1111 input samples needed

template <int SRC>
__global__ __launch_bounds__(256) void acq_frontend_kernel(const uint8_t *__restrict__ src,
                                                          size_t src_stride, int nvalid, int nchips, int boc,
                                                          float2 *__restrict__ td, int fft_len)
{
    __shared__ float2 xs[FE_NX + 1];
    __shared__ float2 y1[FE_NY1 + 1];
    const int tid = threadIdx.x;
    const int o0 = blockIdx.x * FE_TILE;
    const void *s = src + (size_t) blockIdx.y * src_stride;
    for (int u = tid; u < FE_NX; u += 256) kg_st(&xs[u], acq_source<SRC>(s, 4 * o0 + u, nvalid, nchips, boc));
    __syncthreads();
    for (int u = tid; u < FE_NY1; u += 256) kg_st(&y1[u], hb_tap(&xs[2 * u]));
    __syncthreads();
    kg_st(&td[(size_t) blockIdx.y * fft_len + o0 + tid], hb_tap(&y1[2 * tid]));
}

// ---------------------------------------------------------------------------
// Forward N-point FFT (N = P * 4096), natural time order in -> residue-major
// spectrum out.  Two launches (both latency-bound, so the work is spread over
// many CUs):
//   acq_fft_sub_kernel      P workgroups per transform; workgroup g transforms the
//                           samples n = P*n1 + g (4096 points) -> F_g in scratch
//   acq_fft_combine_kernel  radix-P across g with W_N^{-k*g}, writes the planes
// ---------------------------------------------------------------------------
#define STAMP(i)                                                                      \
    do {                                                                              \
        if (STAMPS) {                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                        \
            if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)               \
                stamps[i] = __builtin_amdgcn_s_memrealtime();                         \
            __builtin_amdgcn_sched_barrier(0);                                        \
        }                                                                             \
    } while (0)

// STAMPS: diagnostic instantiation only (kg_acq_debug_fft_stamps); the product
// launches the STAMPS = false kernels, in which no stamp code exists.
template <bool STAMPS>
__global__ __launch_bounds__(256) void acq_fft_sub_kernel(const float2 *__restrict__ td,
                                                         float2 *__restrict__ scratch,   // [batch][P][4096]
                                                         const float2 *__restrict__ tab4096, int P,
                                                         unsigned long long *__restrict__ stamps)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];   // 2 x 4096
    const int t = threadIdx.x, g = blockIdx.x;
    const float2 *in = td + (size_t) blockIdx.y * P * SUB;
    float2 *out = scratch + ((size_t) blockIdx.y * P + g) * SUB;
    STAMP(0);
    cf x[16], y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld(&in[P * (t + 256 * j) + g]);
    kg_tw4096 tw;
    kg_tw4096_load(tw, tab4096, t);
    if (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    STAMP(1);
    kg_subfft4096<-1>(x, y, smem, smem + SUB, tw, t);
    STAMP(2);
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st(&out[t + 256 * m], y[m]);       // F_g[k'] at k'
    if (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    STAMP(3);
}

template <int P>
__global__ __launch_bounds__(256) void acq_fft_combine_kernel(const float2 *__restrict__ scratch,
                                                             float2 *__restrict__ planes,     // layout A, or null
                                                             float2 *__restrict__ planes_b,   // layout B, or null
                                                             size_t planes_stride,   // in float2
                                                             const float2 *__restrict__ tabN,
                                                             int halo)               // H: 0 data, >0 code
{
    static_assert(P == 4 || P == 16, "N = 16384 or 65536");
    const int k = blockIdx.x * 256 + threadIdx.x;      // k' < 4096
    const float2 *f = scratch + (size_t) blockIdx.y * (P * SUB);
    cf v[P];
    v[0] = kg_ld(&f[k]);
#pragma unroll
    for (int g = 1; g < P; g++) v[g] = kg_cmulc(kg_ld(&f[g * SUB + k]), kg_ld(&tabN[g * k]));   // W_N^{-k*g}
    // X[k + 4096 q], q = 0..P-1
    cf fq[P];
    if constexpr (P == 4) {
        kg_radix4<-1>(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int q = 0; q < 4; q++) fq[q] = v[q];
    } else {
        kg_radix16<-1>(v, fq);
    }
    // bin k + 4096 q -> plane k & (P-1), element k1 = k / P + (4096 / P) q
    if (planes) {                                      // layout A: k1 = tt + 256 j -> row j>>1, column 2 (tt + H) + (j&1)
        const int row = 2 * (256 + 2 * halo), plane = 8 * row;
        float2 *o = planes + (size_t) blockIdx.y * planes_stride + (size_t) (k & (P - 1)) * plane;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int k1 = k / P + (SUB / P) * q;
            const int tt = k1 & 255, j = k1 >> 8;
            kg_st(&o[(j >> 1) * row + 2 * (tt + halo) + (j & 1)], fq[q]);
            if (tt < halo) {                               // also the right halo of leg j-1
                const int jj = (j - 1) & 15;
                kg_st(&o[(jj >> 1) * row + 2 * (tt + 256 + halo) + (jj & 1)], fq[q]);
            }
            if (tt >= 256 - halo) {                        // and the left halo of leg j+1
                const int jj = (j + 1) & 15;
                kg_st(&o[(jj >> 1) * row + 2 * (tt - 256 + halo) + (jj & 1)], fq[q]);
            }
        }
    }
    if (planes_b) {                                    // layout B: k1 = ii + 512 j -> row j>>1, column 2 (ii + H) + (j&1)
        const int row = 2 * (512 + 2 * halo), plane = 4 * row;
        float2 *o = planes_b + (size_t) blockIdx.y * planes_stride + (size_t) (k & (P - 1)) * plane;
#pragma unroll
        for (int q = 0; q < P; q++) {
            const int k1 = k / P + (SUB / P) * q;
            const int ii = k1 & 511, j = k1 >> 9;
            kg_st(&o[(j >> 1) * row + 2 * (ii + halo) + (j & 1)], fq[q]);
            if (ii < halo) {
                const int jj = (j - 1) & 7;
                kg_st(&o[(jj >> 1) * row + 2 * (ii + 512 + halo) + (jj & 1)], fq[q]);
            }
            if (ii >= 512 - halo) {
                const int jj = (j + 1) & 7;
                kg_st(&o[(jj >> 1) * row + 2 * (ii - 512 + halo) + (jj & 1)], fq[q]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Correlate(): one workgroup per (block, SV, Doppler) cell, persistent over a
// cell list.  search.cpp:465-496.
// ---------------------------------------------------------------------------
struct acq_red { float p; int i; float s; int pad; };

// Round 4: each claim counter on its own 128-byte line.  Agent-scope atomics on ONE line are served one at a time, some 12.7 ns
// each chip-wide (the waterfall kernel's single counter was its floor, kg_wf.hip); the eight group counters sat in one line.
#define ACQ_CLAIM_STRIDE 32
#define ACQ_LDS_BYTES (2 * SUB * sizeof(float2) + 4 * sizeof(acq_red) + 16)     // + the claimed cell index

// One (block, SV) pair, prepared by the host: 16 bytes = one s_load_dwordx4.  The
// cells of a launch are the pairs x the Doppler bins; the kernel walks (pair, bin)
// with two additions per step (no division, no index lists).
struct acq_pair_desc {
    int data_off;     // float2 offset of the block's data spectrum
    int code_off;     // float2 offset of the SV's code spectrum
    int limit;        // peak-search window (search.cpp:486)
    int out;          // index of the pair's first cell in cells[]
};

struct acq_cell_desc { int data_off, code_off, dop, limit, out; };

// Launch constants (kernarg).  The cells of XCD group x are (pair x + 8 pg, bin di) in the order
// cell = pg * ndop + di.  A persistent workgroup takes cells `slot` and `slot + nslots` and then
// claims one more at a time from the group's counter (an agent-scope atomic): the two workgroups
// of a CU do not run at the same speed -- the older one wins the vector-issue arbitration (810 us
// against 1036 us for the same 82 cells, profiles/r02_acq_wg_life.txt) -- so a static split leaves
// every CU half empty for the last fifth of the launch.
struct acq_walk {
    int npairs;       // pairs in this launch's table
    int ndop, dop_lo;
};

// Work item = (cell, k2): one 4096-point sub-transform.  The operands of the
// NEXT item are loaded while the current one is transformed (PREFETCH).
//
// Wave-uniform constants of the combine twiddle W_N^{n*k2}, n = t + 256 m:
// W_N^{256 m k2} = W_R^{m k2} with R = N / 256 and, with m = 4a + b,
// = W_R^{4a k2} * W_R^{b k2}.  comb[k2] = { W_R^{k2}, W_R^{2 k2}, W_R^{3 k2},
// W_R^{4 k2}, W_R^{8 k2}, W_R^{12 k2}, -, - } (host-built, fp32 roundings of double
// values); quart[k2][q] = W_P^{q k2}: the factor of output quarter q (NQ = 4).

// Round 4: the second exchange without the swizzle (kg_fft.h, kg_subfft4096_l: its writer has the slot column in the lane, its
// stores and loads are conflict-free as they stand): one address register instead of sixteen.  -DACQ_X2_SWIZZLE=1: as before.
#ifndef ACQ_X2_SWIZZLE
#define ACQ_X2_SWIZZLE 0
#endif
#if ACQ_X2_SWIZZLE
#define ACQ_X2_WR(t, th, tl, m) ((th) * 256 + 16 * (m) + ((tl) ^ (m)))
#define ACQ_X2_RD(t, rd) (rd)
#else
#define ACQ_X2_WR(t, th, tl, m) ((t) + 16 * (15 * (th) + (m)))
#define ACQ_X2_RD(t, rd) (t)
#endif
template <int P, int NQ, bool PREFETCH, bool STAMPS = false>   // PREFETCH: always true (kept in the names)
__global__ __launch_bounds__(256, NQ == 1 ? 2 : 1) void acq_correlate_kernel(
    const float2 *__restrict__ data,  // [nblocks][P][4096]
    const float2 *__restrict__ code,  // [max_sats][P][8 rows][2 (256 + 2 H)]
    const float2 *__restrict__ tab4096, const float2 *__restrict__ tabN,
    const float2 *__restrict__ comb,           // [P][8]
    const float2 *__restrict__ quart,          // [P][4]
    const acq_pair_desc *__restrict__ pairs,   // pair p belongs to XCD group p & 7
    int *__restrict__ claim,                   // [8][ACQ_CLAIM_STRIDE] per-group cell counters, zero at launch
    acq_walk walk,
    int halo,                                  // H of the code planes
    kg_acq_cell *__restrict__ cells,           // [nblocks][nsats][ndop]
    unsigned long long *__restrict__ stamps = nullptr)
{
    static_assert(P == 4 || P == 16, "N = 16384 or 65536");
    constexpr int LOGP = P == 4 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tileA = smem, *tileB = smem + SUB;
    acq_red *red = (acq_red *) (smem + 2 * SUB);
    volatile int *red_claim = (volatile int *) (red + 4);
    const int t = threadIdx.x;
    // diagnostics: one workgroup, thread 0, 16 stamps per (cell, k2) item
    unsigned long long *st = (STAMPS && t == 0 && blockIdx.x == 8) ? stamps : nullptr;
    if (STAMPS && st) { st[0] = __builtin_amdgcn_s_memtime(); st[1] = __builtin_amdgcn_s_memrealtime(); }
    // diagnostics: every workgroup's start / end (100 MHz), XCC id and cell count at stamps[512 + 4 b ..]
    unsigned long long *wgt = (STAMPS && t == 0) ? stamps + 512 + 4 * blockIdx.x : nullptr;
    int wg_cells = 0;
    if (STAMPS && wgt) {
        wgt[0] = __builtin_amdgcn_s_memrealtime();
        wgt[2] = (unsigned long long) __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID, all 32 bits
    }

    kg_tw4096 tw;
    kg_tw4096_load(tw, tab4096, t);
    // W_N^{t*k2}, the lane half of the combine twiddle, from registers: W_N^{t*b} (b = 1..3) and,
    // for P = 16, W_N^{4*t*a} (a = 1..3); no per-item vector load, whose s_waitcnt would also drain
    // the operand prefetch of the next item
    cf wb[3], wa[3];
#pragma unroll
    for (int i = 1; i < 4; i++) { wb[i - 1] = kg_ld(&tabN[t * i]); wa[i - 1] = kg_ld(&tabN[(4 * t * i) & (P * SUB - 1)]); }
    (void) wa;
    auto sel3 = [](const cf (&w)[3], int i) { return i == 1 ? w[0] : (i == 2 ? w[1] : w[2]); };

    // XCD-aware: workgroups b and b+8 share an XCD (round-robin dispatch), so the
    // cells of one (block, SV) pair -- one code spectrum -- all belong to one
    // group and that spectrum stays in one L2.  Speed only, never correctness.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int gpairs = (walk.npairs - xcd + 7) >> 3;           // pairs xcd, xcd + 8, ...
    // Fewer pairs than groups -- the reference's own calling pattern, one SV per Correlate() call -- would
    // leave 8 - npairs XCDs idle: the CELLS are then dealt round-robin instead (cell c of the launch, pair-major,
    // belongs to group c & 7); the code spectrum lands in several L2s, which one or a few pairs can afford.
    const bool spread = walk.npairs < 8;
    const int ndop = walk.ndop, ncell = spread ? (walk.npairs * ndop - xcd + 7) >> 3 : gpairs * ndop;

    // Operand fetch: eight 16-byte buffer loads per spectrum.  The descriptor
    // (SGPRs) carries the plane base, soffset the row, voffset the lane column:
    // no per-load VALU address arithmetic.
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int rowb_c = 2 * (256 + 2 * halo) * (int) sizeof(float2);   // code row, bytes
    const int plane_c = 8 * 2 * (256 + 2 * halo);                     // code plane, float2
    cf d[16], c[16];
    // fetch() = prepare the two buffer descriptors (SALU) + rows 0..7; fetch_rows() issues the loads of
    // rows r0 .. r0+n-1 (one data and one code load per row) so that they can be spread
    struct acq_rsrc { __amdgpu_buffer_rsrc_t drs, crs; int dvo, cvo; };
    auto fetch_prepare = [&](int data_off, int code_off, int dop, int k2) {
        const int s = k2 - dop;
        const int q0 = s >> LOGP;                      // floor((k2 - dop) / P)
        acq_rsrc r;
        r.drs = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (data + data_off + k2 * SUB), 0, SUB * (int) sizeof(float2), 0x00020000);
        r.crs = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (code + code_off + (s & (P - 1)) * plane_c), 0, plane_c * (int) sizeof(float2), 0x00020000);
        r.dvo = t * 16; r.cvo = (t + q0 + halo) * 16;
        return r;
    };
    auto fetch_row = [&](const acq_rsrc &r, int i) {
        const u4 dv = __builtin_amdgcn_raw_buffer_load_b128(r.drs, r.dvo, i * 4096, 0);
        const u4 cv = __builtin_amdgcn_raw_buffer_load_b128(r.crs, r.cvo, i * rowb_c, 0);
        d[2 * i] = cf{__uint_as_float(dv[0]), __uint_as_float(dv[1])};
        d[2 * i + 1] = cf{__uint_as_float(dv[2]), __uint_as_float(dv[3])};
        c[2 * i] = cf{__uint_as_float(cv[0]), __uint_as_float(cv[1])};
        c[2 * i + 1] = cf{__uint_as_float(cv[2]), __uint_as_float(cv[3])};
    };
    auto fetch_drow = [&](const acq_rsrc &r, int i) {
        const u4 dv = __builtin_amdgcn_raw_buffer_load_b128(r.drs, r.dvo, i * 4096, 0);
        d[2 * i] = cf{__uint_as_float(dv[0]), __uint_as_float(dv[1])};
        d[2 * i + 1] = cf{__uint_as_float(dv[2]), __uint_as_float(dv[3])};
    };
    auto fetch_crow = [&](const acq_rsrc &r, int i) {
        const u4 cv = __builtin_amdgcn_raw_buffer_load_b128(r.crs, r.cvo, i * rowb_c, 0);
        c[2 * i] = cf{__uint_as_float(cv[0]), __uint_as_float(cv[1])};
        c[2 * i + 1] = cf{__uint_as_float(cv[2]), __uint_as_float(cv[3])};
    };
    auto fetch = [&](int data_off, int code_off, int dop, int k2) {
        const acq_rsrc r = fetch_prepare(data_off, code_off, dop, k2);
#pragma unroll
        for (int i = 0; i < 8; i++) fetch_row(r, i);
    };
    // cell index -> descriptor: one (uniform) division, one s_load_dwordx4
    auto describe = [&](int idx) {
        const int cell = spread ? (idx << 3) + xcd : idx;
        const int pg = cell / ndop, di = cell - pg * ndop;
        const acq_pair_desc pd = pairs[spread ? pg : (pg << 3) + xcd];
        return acq_cell_desc{pd.data_off, pd.code_off, walk.dop_lo + di, pd.limit, pd.out + di};
    };

    int cur_idx = slot, nxt_idx = slot + nslots;
    if (cur_idx >= ncell) return;
    acq_cell_desc cur = describe(cur_idx);
    fetch(cur.data_off, cur.code_off, cur.dop, 0);

    int st_item = 0;
    for (;;) {
        // one cell ahead; the cell after that is claimed now and its index crosses the workgroup
        // through LDS at this cell's last barrier
        const bool more = nxt_idx < ncell;
        const acq_cell_desc nxt = describe(more ? nxt_idx : cur_idx);
        int claimed = 0;
        // (the counter's value only, and the atomic optimizer off: see wf_frame_kernel -- otherwise wave 0 waits for it here)
        if (t == 0) claimed = __hip_atomic_fetch_add(&claim[xcd * ACQ_CLAIM_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cf acc[NQ][16];
        // The twiddle-accumulate of item k2 is DEFERRED into item k2+1's exchanges, where a wave otherwise
        // only waits (slots: behind the stores before a barrier, or after it while the tile reads are in
        // flight).  yprev holds the deferred item's transform; for NQ == 1 (256 registers, two workgroups per
        // CU) the code rows of the next item are fetched during pass 2 to make room, the data rows keep their
        // place in the conjugate product and pass 0.  NQ == 4 runs one workgroup per CU -- nothing else
        // fills its waits; slots {0,0,1,1} measured best for it (<4,4> 1.67 -> 1.53 ms on the acq59 workload).
        cf yprev[16], pbase = cf{0.f, 0.f}, pg[3], pG[3], pQ[3];
        (void) pQ;
        // acc[4a + b] (+)= yv[4a + b] * (bs * W_R^{4a kp}) * W_R^{b kp} for a in [a0, a1): C[4a + b] = B[a] * g[b]
        // is formed first (wave-uniform factors), the products accumulate through fused multiply-adds
        auto accumulate = [&](const cf (&yv)[16], int kp, cf bs, const cf (&g)[3], const cf (&G)[3],
                              const cf (&Q)[3], int a0, int a1) {
            (void) Q;
            if (kp == 0) {
#pragma unroll
                for (int q = 0; q < NQ; q++)
#pragma unroll
                    for (int a = 0; a < 4; a++)
                        if (a >= a0 && a < a1) {
#pragma unroll
                            for (int b4 = 0; b4 < 4; b4++) acc[q][4 * a + b4] = yv[4 * a + b4];
                        }
                return;
            }
            // B[a] = bs * W_R^{4a kp}: the three products in one block (kg_fft.h, batched products)
            cf B1, B2, B3;
            kg_cmul1x3s(B1, B2, B3, bs, G[0], G[1], G[2]);
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (a < a0 || a >= a1) continue;
                const cf Ba = a == 0 ? bs : (a == 1 ? B1 : (a == 2 ? B2 : B3));
                const cf C0 = Ba;
                cf C1, C2, C3;
                kg_cmul1x3s(C1, C2, C3, Ba, g[0], g[1], g[2]);
                if constexpr (NQ == 1) {
                    kg_cmac4v(acc[0][4 * a], acc[0][4 * a + 1], acc[0][4 * a + 2], acc[0][4 * a + 3],
                              yv[4 * a], yv[4 * a + 1], yv[4 * a + 2], yv[4 * a + 3], C0, C1, C2, C3);
                } else {
                    // four output quarters: z[m] = y[m] * C[m] once, then acc_q[m] += z[m] * W_P^(q*kp)
                    cf z0 = yv[4 * a], z1 = yv[4 * a + 1], z2 = yv[4 * a + 2], z3 = yv[4 * a + 3];
                    kg_cmul4v<false>(z0, z1, z2, z3, C0, C1, C2, C3);
                    const cf z[4] = {z0, z1, z2, z3};
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int m = 4 * a + b;
                        acc[0][m] = acc[0][m] + z[b];
                        if constexpr (P == 4) {            // quarters 1..3: times j^(q*kp), wave-uniform
                            if (kp == 1) {
                                acc[1][m] = kg_addj(acc[1][m], z[b]); acc[2][m] = acc[2][m] - z[b];
                                acc[3][m] = kg_subj(acc[3][m], z[b]);
                            } else if (kp == 2) {
                                acc[1][m] = acc[1][m] - z[b]; acc[2][m] = acc[2][m] + z[b];
                                acc[3][m] = acc[3][m] - z[b];
                            } else {
                                acc[1][m] = kg_subj(acc[1][m], z[b]); acc[2][m] = acc[2][m] - z[b];
                                acc[3][m] = kg_addj(acc[3][m], z[b]);
                            }
                        }
                    }
                    if constexpr (P != 4) {                // times W_16^(q*kp), wave-uniform
#pragma unroll
                        for (int q = 1; q < 4; q++)
                            kg_cmac4s(acc[q][4 * a], acc[q][4 * a + 1], acc[q][4 * a + 2], acc[q][4 * a + 3],
                                      z0, z1, z2, z3, Q[q - 1]);
                    }
                }
            }
        };
        // where the four quarters (a = 0..3) of the deferred accumulate go: slot 0 = behind the pass-0 stores,
        // before barrier 1; 1 = after barrier 1, pass-1 tile reads in flight; 2 = behind the pass-1 stores,
        // before barrier 2; 3 = after barrier 2, pass-2 tile reads in flight
#ifndef KG_DEFER_PLACE
#define KG_DEFER_PLACE {1, 1, 1, 1}
#endif
#ifndef KG_DEFER_PLACE4
#define KG_DEFER_PLACE4 {0, 0, 1, 1}
#endif
        auto deferred = [&](int slot, int k2) {
            constexpr int place1[4] = KG_DEFER_PLACE, place4[4] = KG_DEFER_PLACE4;
            int a0 = 4, a1 = 0;                                // the run of quarters placed in this slot
#pragma unroll
            for (int a = 0; a < 4; a++)
                if ((NQ == 1 ? place1[a] : place4[a]) == slot) { a0 = a < a0 ? a : a0; a1 = a + 1; }
            if (a0 < a1 && k2 > 0) {                           // (an empty slot leaves nothing behind, not even the test)
                kg_pin();
                accumulate(yprev, k2 - 1, pbase, pg, pG, pQ, a0, a1);
                kg_pin();
            }
        };
        const int tl = t & 15, th = t >> 4;
        const int rd = t ^ (th & 15);              // P(t + 256 j) = 256 j + (t ^ ((t >> 4) & 15)), kg_fft.h
        // rolled on purpose: unrolled (or with k2 a template constant) the
        // register allocator spills 80+ VGPRs
#pragma unroll 1
        for (int k2 = 0; k2 < P; k2++) {
            unsigned long long *sti = (STAMPS && st && st_item < 24) ? st + 16 + 16 * st_item : nullptr;
            KG_STAMP(STAMPS, sti, 8);
            cf base;                                           // W_N^{t*k2}, used after the transform
            if constexpr (P == 4) base = sel3(wb, k2);
            else {
                const int ka = k2 >> 2, kb = k2 & 3;
                const cf A = sel3(wa, ka), Bv = sel3(wb, kb);
                base = ka == 0 ? Bv : (kb == 0 ? A : kg_cmul(A, Bv));
            }
            cf x[16], y[16];
            // The operands of the next item -- (cur, k2+1) or (nxt, 0); the very last item re-reads its own
            // cell, harmless, so that there is ONE load site that is never skipped -- are fetched row by
            // row between the arithmetic groups of this item: a row's registers are free as soon as its
            // products are formed, and sixteen 1 KiB loads issued back to back queue behind each other in
            // the texture addresser for several hundred cycles.
            const bool same = k2 < P - 1;
            const acq_rsrc nr = fetch_prepare(same ? cur.data_off : nxt.data_off, same ? cur.code_off : nxt.code_off,
                                              same ? cur.dop : nxt.dop, (k2 + 1) & (P - 1));
#ifndef KG_FUSED_TW
#define KG_FUSED_TW 1
#endif
#if KG_FUSED_TW
            // Round 4: conj(data) * code (simd_multiply_conjugate_ccc, support/simd.cpp:39-67) FUSED into the first stage of
            // pass 0 (kg_cc_radix16_h, kg_fft.h): 99 packed instructions where products + butterfly took 112, and no copies
            // of the code operands (the in-place products needed sixteen).  The next item's data rows are requested two at
            // a time as the operands they overwrite are consumed; the LDS stores group by group as before.
            KG_STAMP(STAMPS, sti, 9);
            KG_STAMP(STAMPS, sti, 10);
            kg_cc_radix16_h<+1>(c, d, y, [&](int s) {
                kg_pin();
                if (s < 4) {
                    if constexpr (NQ == 1) { fetch_drow(nr, 2 * s); fetch_drow(nr, 2 * s + 1); }
                    else { fetch_row(nr, 2 * s); fetch_row(nr, 2 * s + 1); }
                } else {
#pragma unroll
                    for (int m = s - 4; m < 16; m += 4) kg_st_tile(&tileA[16 * t + (m ^ tl)], y[m]);
                }
                kg_pin();
            });
#else
            // conj(data) * code, simd_multiply_conjugate_ccc (support/simd.cpp:39-67)
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                x[j] = c[j]; x[j + 1] = c[j + 1]; x[j + 2] = c[j + 2]; x[j + 3] = c[j + 3];
                kg_cmul4v<true>(x[j], x[j + 1], x[j + 2], x[j + 3], d[j], d[j + 1], d[j + 2], d[j + 3]);
                if (STAMPS && j == 12) KG_STAMP(STAMPS, sti, 9);
                if (j >= 4) {                                   // rows 0..2 (registers of the batch before)
                    kg_pin();
                    if constexpr (NQ == 1) fetch_drow(nr, j / 4 - 1); else fetch_row(nr, j / 4 - 1);
                    kg_pin();
                }
            }
            KG_STAMP(STAMPS, sti, 10);
            // pass 0 (no twiddles), out index 16 t + m: rows 3..7 between its first-stage groups, the LDS
            // stores group by group as the outputs become final
            kg_radix16_h<+1>(x, y, [&](int s) {
                kg_pin();
                if (s < 4) {
                    if constexpr (NQ == 1) { fetch_drow(nr, 3 + s); if (s == 3) fetch_drow(nr, 7); }
                    else { fetch_row(nr, 3 + s); if (s == 3) fetch_row(nr, 7); }
                } else {
#pragma unroll
                    for (int m = s - 4; m < 16; m += 4) kg_st_tile(&tileA[16 * t + (m ^ tl)], y[m]);
                }
                kg_pin();
            });
#endif
            KG_STAMP(STAMPS, sti, 0);
            deferred(0, k2);
            KG_STAMP(STAMPS, sti, 1);
            __syncthreads();
            KG_STAMP(STAMPS, sti, 2);
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileA[rd + 256 * j]);
            deferred(1, k2);
            if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            KG_STAMP(STAMPS, sti, 3);
            // pass 1: twiddle W256^(j*(t&15)), out index (t>>4)*256 + (t&15) + 16 m
#if KG_FUSED_TW
            kg_tw_radix16_h<+1>(x, y, tw.p1, [&](int s) {
#else
            kg_twiddle16<+1>(x, tw.p1);
            kg_radix16_h<+1>(x, y, [&](int s) {
#endif
                if (s >= 4) {
                    kg_pin();
#pragma unroll
                    for (int m = s - 4; m < 16; m += 4) kg_st_tile(&tileB[ACQ_X2_WR(t, th, tl, m)], y[m]);
                    kg_pin();
                }
            });
            KG_STAMP(STAMPS, sti, 4);
            deferred(2, k2);
            __syncthreads();
            KG_STAMP(STAMPS, sti, 5);
            // wave-uniform constants (s_load), hidden behind pass 2
            cf g[3], G[3], Q[3];
            (void) Q;
#pragma unroll
            for (int i = 0; i < 3; i++) { g[i] = kg_ld(&comb[8 * k2 + i]); G[i] = kg_ld(&comb[8 * k2 + 3 + i]); }
            if constexpr (NQ == 4 && P != 4) {
#pragma unroll
                for (int q = 1; q < 4; q++) Q[q - 1] = kg_ld(&quart[4 * k2 + q]);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = kg_ld_tile(&tileB[ACQ_X2_RD(t, rd) + 256 * j]);
            deferred(3, k2);
            if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            KG_STAMP(STAMPS, sti, 6);
            // pass 2: twiddle W4096^(j*t), out index t + 256 m (kept in registers)
            // (NQ == 1: the next item's code rows, two per first-stage group)
            auto crows = [&](int s) {
                if constexpr (NQ == 1) {
                    if (s < 4) { kg_pin(); fetch_crow(nr, 2 * s); fetch_crow(nr, 2 * s + 1); kg_pin(); }
                }
            };
#if KG_FUSED_TW
            kg_tw_radix16_h<+1>(x, yprev, tw.p2, crows);
#else
            kg_twiddle16<+1>(x, tw.p2);
            kg_radix16_h<+1>(x, yprev, crows);
#endif
            pbase = base;
#pragma unroll
            for (int i = 0; i < 3; i++) { pg[i] = g[i]; pG[i] = G[i]; }
            if constexpr (NQ == 4 && P != 4) {
#pragma unroll
                for (int i = 0; i < 3; i++) pQ[i] = Q[i];
            }
            KG_STAMP(STAMPS, sti, 7);
            KG_STAMP(STAMPS, sti, 11);
            if (STAMPS) st_item++;
        }
        accumulate(yprev, P - 1, pbase, pg, pG, pQ, 0, 4);     // the cell's last item

        // search.cpp:486-490: power, first maximum (strict >), running total
        // Row r = m + 16 q holds n = t + 256 r.  `limit` is wave-uniform, so a row lies wholly inside the
        // window (256 (r + 1) <= limit: a scalar test; all but the last row or two) or needs the per-lane
        // test; the row number, not n, is tracked as the position of the maximum (an inline constant).
        const int limit = cur.limit, full_rows = limit >> 8;
        float bp = 0.f, sum = 0.f;
        int bi = 0;
        if constexpr (NQ == 1 && P == 4) {
            // the 16368-lag kernel's form of the scan (acq_correlate8_kernel): powers kept, total in packed pairs, maximum
            // by fmax, the lane's FIRST row holding it from a row mask built with a compare and an add-with-carry per row
            // (acq 0.796 -> 0.789 ms; P = 16 keeps the serial form below: its sixteen more registers of powers spilled there,
            // configs[4] 2.72 -> 2.76 ms; re-measured in round 4 with the sixteen registers the unswizzled exchange freed: no
            // spills, 2.66 -> 2.70 .. 2.73 ms)
            float pw[16];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const cf sq = acc[0][m] * acc[0][m];
                pw[m] = sq.x + sq.y;
                if (m >= full_rows) {
                    asm volatile("");
                    pw[m] = (t + 256 * m < limit) ? pw[m] : 0.f;
                }
            }
            cf s2 = cf{0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 16; m += 2) s2 = s2 + cf{pw[m], pw[m + 1]};
            sum = s2.x + s2.y;
#pragma unroll
            for (int m = 0; m < 16; m += 2) bp = __builtin_fmaxf(bp, __builtin_fmaxf(pw[m], pw[m + 1]));
            unsigned qm[2] = {0, 0};                               // two independent chains of eight rows
            // (gfx950 wants two wait states between a VALU that writes a carry / mask register and the VALU that reads it -- the
            // compiler pads its own pairs with s_nop 1, inside an asm block nobody does: two rows' compares first, into two
            // register pairs, then the four add-with-carry, every consumer three instructions behind its producer)
#pragma unroll
            for (int m = 7; m >= 0; m -= 2) {
                unsigned long long k0, k1, k2, k3;
                asm("v_cmp_eq_f32_e64 %2, %6, %10\n\tv_cmp_eq_f32_e64 %3, %7, %10\n\t"
                    "v_cmp_eq_f32_e64 %4, %8, %10\n\tv_cmp_eq_f32_e64 %5, %9, %10\n\t"
                    "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\tv_addc_co_u32_e64 %1, %3, %1, %1, %3\n\t"
                    "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\tv_addc_co_u32_e64 %1, %5, %1, %1, %5"
                    : "+v"(qm[0]), "+v"(qm[1]), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
                    : "v"(pw[m]), "v"(pw[m + 8]), "v"(pw[m - 1]), "v"(pw[m + 7]), "v"(bp));
            }
            const unsigned rowmask = qm[0] | (qm[1] << 8);
            bi = t + 256 * (int) __builtin_ctz(rowmask | 0x80000000u);
        } else if constexpr (NQ == 1) {
            int br = 0;
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const cf v = acc[0][m];
                const cf sq = v * v;
                const float pw = sq.x + sq.y;
                if (m < full_rows) {
                    const bool take = pw > bp;
                    bp = take ? pw : bp; br = take ? m : br;
                    sum += pw;
                } else {
                    const bool in = t + 256 * m < limit, take = in & (pw > bp);
                    bp = take ? pw : bp; br = take ? m : br;
                    sum += in ? pw : 0.f;
                }
            }
            bi = t + 256 * br;
        } else {
            // (one workgroup per CU: the straight masked scan measured faster here than a test per row or
            // per output quarter, 1.53 against 1.64 ms for <4,4> on the acq59 workload)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    const int n = t + 256 * m + SUB * q;
                    const cf v = acc[q][m];
                    const float pw = v.x * v.x + v.y * v.y;
                    const bool in = n < limit, take = in & (pw > bp);
                    bp = take ? pw : bp; bi = take ? n : bi;
                    sum += in ? pw : 0.f;
                }
            }
        }
#define ACQ_RED_STEP(L)                                                               \
        {                                                                                \
            const float op = kg_xchg<L>(bp), os = kg_xchg<L>(sum);                       \
            const int oi = kg_xchg<L>(bi);                                               \
            const bool take = (op > bp) | ((op == bp) & (oi < bi));   /* branch-free */  \
            bp = take ? op : bp; bi = take ? oi : bi;                                    \
            sum += os;                                                                   \
        }
        // the wave's maximum, the lowest n holding it (what one strict-> scan in ascending n finds), the total
        {
            float wmax = bp, wsum = sum;
            kg_wave_max_sum(wmax, wsum);
            const int wn = kg_wave_min(bp == wmax ? bi : 0x7fffffff);
            // red[] was last read before this cell's barriers
            if ((t & 63) == 0) { red[t >> 6].p = wmax; red[t >> 6].i = wn; red[t >> 6].s = wsum; }
        }
        if (t == 0) *red_claim = 2 * nslots + claimed;
        __syncthreads();
        // (rewritten only after eight more barriers; readfirstlane: the index must stay wave-uniform, or
        // every buffer load below turns into a waterfall loop over its descriptor)
        const int nn_idx = __builtin_amdgcn_readfirstlane(*red_claim);
        if (t < 64) {                                  // lanes 0..3 of wave 0 merge the four waves
            const acq_red r = red[t & 3];
            bp = r.p; bi = r.i; sum = r.s;
            ACQ_RED_STEP(0) ACQ_RED_STEP(1)
            if (t == 0) {
                const float ave = sum / (float) limit;     // :493
                kg_acq_cell cc;
                cc.snr = bp / ave;                         // :494
                cc.max_pwr = bp; cc.tot_pwr = sum; cc.idx = bi;
                cells[cur.out] = cc;
            }
        }
#undef ACQ_RED_STEP
        if (STAMPS) wg_cells++;
        if (!more) break;
        cur = nxt; cur_idx = nxt_idx; nxt_idx = nn_idx;
    }
    if (STAMPS && st) { st[2] = __builtin_amdgcn_s_memtime(); st[3] = __builtin_amdgcn_s_memrealtime(); }
    if (STAMPS && wgt) { wgt[1] = __builtin_amdgcn_s_memrealtime(); wgt[3] = (unsigned long long) wg_cells; }
}

// ---------------------------------------------------------------------------
// Correlate() for the 16368-lag window (E1B): the same cell walk with FOUR output quarters per point, by a
// 512-thread workgroup at 8 points per thread (kg_subfft4096_r8, kg_fft.h).  The 256 x 16 form above needs
// acc[4][16] = 128 registers of accumulators per thread on top of its transform: 438-453 registers, one wave per
// SIMD, nothing to issue while that wave waits (7 500 cycles per 4096-point item against 3 050 for the C/A
// kernel's pair of workgroups).  Here a thread owns outputs n = i + 512 m (m < 8) of each quarter: 64 registers of
// accumulators, two waves on every SIMD.  Thread i reads elements k1 = i + 512 j of a plane in layout B (head of
// this file): four 16-byte buffer loads per spectrum.
// Combine twiddle W_N^{n k2} = W_N^{i k2} (registers) x W_{N/512}^{m k2}, m = 4a + b: comb8[k2] = { W^{k2},
// W^{2 k2}, W^{3 k2}, W^{4 k2} } with W = W_{N/512}.
// ---------------------------------------------------------------------------
#define ACQ8_LDS_BYTES (3 * SUB * sizeof(float2) + 8 * sizeof(acq_red) + 16 + 16 * 8 * sizeof(float2) + 3 * 512 * 4)   // + the per-k2 constants + the cell-end hand-over


template <int P, bool STAMPS = false>        // STAMPS: diagnostic instantiation only (kg_acq_debug_corr_stamps)
__global__ __launch_bounds__(512, 1) void acq_correlate8_kernel(
    const float2 *__restrict__ data, const float2 *__restrict__ code,
    const float2 *__restrict__ tab4096, const float2 *__restrict__ tabN,
    const float2 *__restrict__ comb8,          // [P][4]
    const float2 *__restrict__ quart,          // [P][4]
    const acq_pair_desc *__restrict__ pairs, int *__restrict__ claim, acq_walk walk, int halo,
    kg_acq_cell *__restrict__ cells, unsigned long long *__restrict__ stamps = nullptr)
{
    static_assert(P == 4 || P == 16, "N = 16384 or 65536");
    constexpr int LOGP = P == 4 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tile0 = smem, *tile1 = smem + SUB, *tile2 = smem + 2 * SUB;
    acq_red *red = (acq_red *) (smem + 3 * SUB);
    volatile int *red_claim = (volatile int *) (red + 8);
    // per-item constants { W^{k2}, W^{2 k2}, W^{3 k2}, W^{4 k2}, W_P^{k2}, W_P^{2 k2}, W_P^{3 k2}, - } (W = W_{N/512}) in LDS,
    // read by broadcast ds_reads: as scalar loads they shared the lgkmcnt counter with the tile reads, which then
    // waited for a scalar-cache round trip in every item
    float2 *cst = (float2 *) ((char *) (red + 8) + 16);
    // cell-end hand-over (round 4): every lane's (maximum, its n, total) of the cell that just ended; waves 0..3 reduce
    // them -- their own and their SIMD partner's (lane i + 256) -- one item later, in barrier slack
    float *xch_p = (float *) (cst + 16 * 8);
    int *xch_i = (int *) (xch_p + 512);
    float *xch_s = (float *) (xch_i + 512);
    const int i = threadIdx.x;
    if (i < 8 * P) {
        const int k2 = i >> 3, k = i & 7;
        kg_st(&cst[i], k < 4 ? kg_ld(&comb8[4 * k2 + k]) : (k < 7 ? kg_ld(&quart[4 * k2 + (k - 3)]) : cf{0.f, 0.f}));
    }
    // diagnostics: workgroup 8, lane 0 of waves 0 and 4 (which share a SIMD), 16 stamps per item from slot 16 / 1040
    unsigned long long *st = (STAMPS && (i & 255) == 0 && blockIdx.x == 8) ? stamps + (i ? 1024 : 0) : nullptr;
    int st_item = 0;

    kg_tw4096_r8 tw;
    kg_tw4096_r8_load(tw, tab4096, i);
    cf wb[3], wa[3];                            // W_N^{i b}, W_N^{4 i a}: the lane half of the combine twiddle
#pragma unroll
    for (int k = 1; k < 4; k++) { wb[k - 1] = kg_ld(&tabN[i * k]); wa[k - 1] = kg_ld(&tabN[(4 * i * k) & (P * SUB - 1)]); }
    (void) wa;
    auto sel3 = [](const cf (&w)[3], int k) { return k == 1 ? w[0] : (k == 2 ? w[1] : w[2]); };

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int gpairs = (walk.npairs - xcd + 7) >> 3;
    const bool spread = walk.npairs < 8;       // as in acq_correlate_kernel
    const int ndop = walk.ndop, ncell = spread ? (walk.npairs * ndop - xcd + 7) >> 3 : gpairs * ndop;

    // Operand fetch (layout B, head of this file): four 16-byte buffer loads per spectrum -- legs 2p and 2p + 1 of
    // thread i sit side by side in row p, the wave's 64 lanes cover a contiguous KiB.  (First version: the 256-thread
    // layout read with eight 8-byte loads per spectrum at a 16-byte lane stride -- 128 load instructions per item and
    // CU at ~20 cycles of the one texture addresser each were 2 600 of an item's 4 300 cycles.)
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int rowb_c = 2 * (512 + 2 * halo) * (int) sizeof(float2);   // code row, bytes
    const int plane_c = 4 * 2 * (512 + 2 * halo);                     // code plane, float2
    cf d[8], c[8];
    struct acq_rsrc { __amdgpu_buffer_rsrc_t drs, crs; int dvo, cvo; };
    auto fetch_prepare = [&](int data_off, int code_off, int dop, int k2) {
        const int s = k2 - dop;
        const int q0 = s >> LOGP;
        acq_rsrc r;
        r.drs = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (data + data_off + k2 * SUB), 0, SUB * (int) sizeof(float2), 0x00020000);
        r.crs = __builtin_amdgcn_make_buffer_rsrc(
            (void *) (code + code_off + (s & (P - 1)) * plane_c), 0, plane_c * (int) sizeof(float2), 0x00020000);
        r.dvo = i * 16; r.cvo = (i + q0 + halo) * 16;
        return r;
    };
    auto fetch_row = [&](const acq_rsrc &r, int p) {       // row p: legs 2p, 2p + 1 of both spectra
        const u4 dv = __builtin_amdgcn_raw_buffer_load_b128(r.drs, r.dvo, p * 8192, 0);
        const u4 cv = __builtin_amdgcn_raw_buffer_load_b128(r.crs, r.cvo, p * rowb_c, 0);
        d[2 * p] = cf{__uint_as_float(dv[0]), __uint_as_float(dv[1])};
        d[2 * p + 1] = cf{__uint_as_float(dv[2]), __uint_as_float(dv[3])};
        c[2 * p] = cf{__uint_as_float(cv[0]), __uint_as_float(cv[1])};
        c[2 * p + 1] = cf{__uint_as_float(cv[2]), __uint_as_float(cv[3])};
    };
    // cell index -> descriptor.  The 16-byte pair record comes through the VECTOR memory path (every lane loads the same
    // address, readfirstlane makes it scalar again): as an s_load it shared lgkmcnt with the tile reads of the cell's
    // first item, which waited ~1 000 cycles for it (in-kernel stamps); a vector load is waited for where it is used.
    int vzero = 0;
    asm volatile("" : "+v"(vzero));                // opaque: keeps the compiler from scalarising the load below
    // Round 4: the record is REQUESTED at the top of a cell and turned into scalars only where the next cell's first
    // operands are addressed (item P - 2): resolved on the spot, every cell began with an exposed L2 round trip -- ~1 000
    // cycles of the ~4 800 between a cell's last item and the next cell's first (profiles/r03_e1b8_stamps_final.txt).
    struct acq_pend { int4 pv; int di; };
    auto describe_issue = [&](int idx) {
        const int cell = spread ? (idx << 3) + xcd : idx;
        const int pg = cell / ndop, di = cell - pg * ndop;
        return acq_pend{*(const int4 *) (pairs + (spread ? pg : (pg << 3) + xcd) + vzero), di};
    };
    auto describe_resolve = [&](const acq_pend &pd) {
        const int data_off = __builtin_amdgcn_readfirstlane(pd.pv.x), code_off = __builtin_amdgcn_readfirstlane(pd.pv.y);
        const int lim = __builtin_amdgcn_readfirstlane(pd.pv.z), out = __builtin_amdgcn_readfirstlane(pd.pv.w);
        return acq_cell_desc{data_off, code_off, walk.dop_lo + pd.di, lim, out + pd.di};
    };
    auto describe = [&](int idx) { return describe_resolve(describe_issue(idx)); };

    // The four passes of an item are software-pipelined over two items so that an item costs TWO workgroup
    // barriers, not three, and every interval between barriers holds two independent chains:
    //   phase A:  pass 2 of item n (tile 1 -> tile 2)   |  conj-multiply + pass 0 of item n+1 (-> tile 0)     barrier
    //   phase B:  pass 3 of item n (tile 2 -> registers, accumulate)  |  pass 1 of item n+1 (tile 0 -> tile 1)  barrier
    // (tile k is rewritten one barrier after its last read was consumed).  Items run on across cells: item n+1 of a
    // cell's last item is item 0 of the next cell, so there is no drain at a cell boundary and the barrier of the
    // cell-end reduction is the one that closes the last phase B.  d, c hold the operands of item n+1 at the top of
    // item n; the loads of item n+2 are issued as soon as the conjugate product has consumed them.
    const int c0 = (i >> 1) & 7, b1 = (i >> 3) & 1;
    const int r0 = i ^ ((i >> 4) & 7), r1 = i ^ (((i >> 6) & 1) << 3);
    const int w1 = (i >> 3) * 64 + (i & 7), w2 = (i >> 6) * 512 + (i & 63);
    auto fetch_item = [&](const acq_cell_desc &cd, int k2) {
        const acq_rsrc r = fetch_prepare(cd.data_off, cd.code_off, cd.dop, k2);
#pragma unroll
        for (int p = 0; p < 4; p++) fetch_row(r, p);
    };
    int cur_idx = slot, nxt_idx = slot + nslots;
    if (cur_idx >= ncell) return;
    acq_cell_desc cur = describe(cur_idx);
    fetch_item(cur, 0);
    {   // prologue: item 0 of the workgroup's first cell up to tile 1 (passes 0 and 1), operands of item 1 requested
        cf x[8], y[8];
        // (round 4: products and twiddles fused into the butterflies, kg_fft.h)
        kg_cc_radix8_h<+1>(c, d, y, [&]() { kg_pin(); fetch_item(cur, 1); kg_pin(); });
#pragma unroll
        for (int m = 0; m < 8; m++) kg_st_tile(&tile0[8 * i + (m ^ c0)], y[m]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = kg_ld_tile(&tile0[r0 + 512 * j]);
        kg_tw_radix8<+1>(x, y, tw.p1);
#pragma unroll
        for (int m = 0; m < 8; m++) kg_st_tile(&tile1[w1 + 8 * (m ^ b1)], y[m]);
        __syncthreads();
    }
#ifndef KG_E1B_HANDOVER
#define KG_E1B_HANDOVER(P) true
#endif
    constexpr bool HANDOVER = KG_E1B_HANDOVER(P);
    int prev_out = -1, prev_limit = 1;
    // waves 0..3: lane i reduces its own and lane i + 256's hand-over values, then the wave; red[wave] out
    auto reduce_pairs = [&]() {
        if (i < 256) {
            float bp = xch_p[i], sum = xch_s[i];
            int bi = xch_i[i];
            const float op = xch_p[i + 256], os = xch_s[i + 256];
            const int oi = xch_i[i + 256];
            const bool take = (op > bp) | ((op == bp) & (oi < bi));        // first maximum in ascending n
            bp = take ? op : bp; bi = take ? oi : bi;
            sum += os;
            float wmax = bp, wsum = sum;
            kg_wave_max_sum(wmax, wsum);
            const int wn = kg_wave_min(bp == wmax ? bi : 0x7fffffff);
            if ((i & 63) == 0) { red[i >> 6].p = wmax; red[i >> 6].i = wn; red[i >> 6].s = wsum; }
        }
    };
    auto merge_store = [&](int out, int limit) {
        if (i < 64) {                                  // lanes 0..3 (0..7 without the hand-over) of wave 0 merge the wave results
            const acq_red r = red[HANDOVER ? (i & 3) : (i & 7)];
            float bp = r.p, sum = r.s;
            int mi = r.i;
#define ACQ_RED_STEP(L)                                                               \
            {                                                                            \
                const float op = kg_xchg<L>(bp), os = kg_xchg<L>(sum);                   \
                const int oi = kg_xchg<L>(mi);                                           \
                const bool take = (op > bp) | ((op == bp) & (oi < mi));                  \
                bp = take ? op : bp; mi = take ? oi : mi;                                \
                sum += os;                                                               \
            }
            ACQ_RED_STEP(0) ACQ_RED_STEP(1)
            if (!HANDOVER) ACQ_RED_STEP(2)
#undef ACQ_RED_STEP
            if (i == 0) {
                const float ave = sum / (float) limit;     // :493
                kg_acq_cell cc;
                cc.snr = bp / ave;                         // :494
                cc.max_pwr = bp; cc.tot_pwr = sum; cc.idx = mi;
                cells[out] = cc;
            }
        }
    };
    for (;;) {
        const bool more = nxt_idx < ncell;
#ifndef KG_E1B_EAGER_DESCRIBE
        const acq_pend pend = describe_issue(more ? nxt_idx : cur_idx);
        acq_cell_desc nxt = cur;                       // (resolved at item P - 2, the first one that addresses the next cell)
#else
        const acq_cell_desc nxt = describe(more ? nxt_idx : cur_idx);
#endif
        int claimed = 0;
        // (lane 0 of wave 7 claims: wave 0 already carries the result merge and store of every cell)
        if (i == 448) claimed = __hip_atomic_fetch_add(&claim[xcd * ACQ_CLAIM_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cf acc[4][8];
#pragma unroll 1
        for (int k2 = 0; k2 < P; k2++) {
            cf base;                                           // W_N^{i k2}
            if constexpr (P == 4) base = sel3(wb, k2);
            else {
                const int ka = k2 >> 2, kb = k2 & 3;
                const cf A = sel3(wa, ka), Bv = sel3(wb, kb);
                base = ka == 0 ? Bv : (kb == 0 ? A : kg_cmul(A, Bv));
            }
            // (the workgroup's very last item has no item n+1: its phases still run pass 0 / pass 1 on re-read operands,
            // once per launch, because a conditional load site would make the compiler drain vmcnt at the join -- the
            // prefetch of every item would be waited for on the spot)
            unsigned long long *sti = (STAMPS && st && st_item < 60) ? st + 16 + 16 * st_item : nullptr;
            KG_STAMP(STAMPS, sti, 0);
#ifndef KG_E1B_EAGER_DESCRIBE
            if (k2 == P - 2) nxt = describe_resolve(pend);
#endif
            // The operands of item n+2 -- (cur, k2 + 2) or (nxt, k2 + 2 - P); ONE set of load sites, never skipped --
            // are requested one row (a data and a code load, two legs each) at a time BETWEEN the arithmetic blocks of both phases:
            // a buffer load costs the CU's one texture addresser about twenty cycles, all eight waves reach the same
            // point of an item together, and sixteen loads issued back to back held every wave in the issue queue for
            // 1 600 cycles of a 5 700-cycle item (in-kernel stamps, profiles/r03_e1b8_stamps_burst.txt).
            const acq_rsrc nr = fetch_prepare(k2 + 2 < P ? cur.data_off : nxt.data_off, k2 + 2 < P ? cur.code_off : nxt.code_off,
                                              k2 + 2 < P ? cur.dop : nxt.dop, (k2 + 2) & (P - 1));
            auto ld = [&](int j) { kg_pin(); fetch_row(nr, j); kg_pin(); };
            // ---- phase A: pass 2 of this item | conj-multiply + pass 0 of the next
            {
                cf xa[8], ya[8], xb[8], yb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) xa[j] = kg_ld_tile(&tile1[r1 + 512 * j]);
                // conj(data) * code, simd_multiply_conjugate_ccc (support/simd.cpp:39-67) -- round 4: fused into pass 0's first
                // stage, as the inter-pass twiddles are into theirs (kg_fft.h: 6 of 42 / 44 packed instructions per pass)
                (void) xb;
                KG_STAMP(STAMPS, sti, 1);
                kg_cc_radix8_h<+1>(c, d, yb, [&]() { ld(0); });
#pragma unroll
                for (int m = 0; m < 8; m++) kg_st_tile(&tile0[8 * i + (m ^ c0)], yb[m]);
                KG_STAMP(STAMPS, sti, 2);
                kg_tw_radix8_h<+1>(xa, ya, tw.p2, [&]() { ld(1); });
                KG_STAMP(STAMPS, sti, 3);
#pragma unroll
                for (int m = 0; m < 8; m++) kg_st_tile(&tile2[w2 + 64 * m], ya[m]);
                KG_STAMP(STAMPS, sti, 4);
                if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                KG_STAMP(STAMPS, sti, 5);
            }
            __syncthreads();
            KG_STAMP(STAMPS, sti, 6);
            // ---- phase B: pass 3 of this item + accumulate | pass 1 of the next
            cf x[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = kg_ld_tile(&tile2[i + 512 * j]);
            {
                cf xb[8], yb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) xb[j] = kg_ld_tile(&tile0[r0 + 512 * j]);
                kg_tw_radix8_h<+1>(xb, yb, tw.p1, [&]() { ld(2); });
#pragma unroll
                for (int m = 0; m < 8; m++) kg_st_tile(&tile1[w1 + 8 * (m ^ b1)], yb[m]);
            }
            KG_STAMP(STAMPS, sti, 7);
            // this item's constants (broadcast LDS reads, in order with the tile reads)
            cf g[3], G, Q[3];
            (void) Q;
#pragma unroll
            for (int k = 0; k < 3; k++) g[k] = kg_ld_tile(&cst[8 * k2 + k]);
            G = kg_ld_tile(&cst[8 * k2 + 3]);
            if constexpr (P != 4) {
#pragma unroll
                for (int q = 1; q < 4; q++) Q[q - 1] = kg_ld_tile(&cst[8 * k2 + 3 + q]);
            }
            kg_tw_radix8_h<+1>(x, y, tw.p3, [&]() { ld(3); });       // y[m]: the sub-transform at n = i + 512 m
            if (k2 == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int m = 0; m < 8; m++) acc[q][m] = y[m];
                if (!HANDOVER && prev_out >= 0) merge_store(prev_out, prev_limit);      // the cell before (wave 0; see the cell end)
            } else {
                // z[m] = y[m] * base * W^{4a k2} * W^{b k2}, m = 4a + b; then acc_q[m] += z[m] * W_P^{q k2}
                const cf B1 = kg_cmul(base, G);
                cf C1, C2, C3, C5, C6, C7;
                kg_cmul1x3v(C1, C2, C3, base, g[0], g[1], g[2]);
                kg_cmul1x3v(C5, C6, C7, B1, g[0], g[1], g[2]);
                kg_cmul4v<false>(y[0], y[1], y[2], y[3], base, C1, C2, C3);
                kg_cmul4v<false>(y[4], y[5], y[6], y[7], B1, C5, C6, C7);
#pragma unroll
                for (int m = 0; m < 8; m++) acc[0][m] = acc[0][m] + y[m];
                if constexpr (P == 4) {                        // quarters 1..3: times j^(q k2), wave-uniform
                    if (k2 == 1) {
#pragma unroll
                        for (int m = 0; m < 8; m++) {
                            acc[1][m] = kg_addj(acc[1][m], y[m]); acc[2][m] = acc[2][m] - y[m]; acc[3][m] = kg_subj(acc[3][m], y[m]);
                        }
                    } else if (k2 == 2) {
#pragma unroll
                        for (int m = 0; m < 8; m++) {
                            acc[1][m] = acc[1][m] - y[m]; acc[2][m] = acc[2][m] + y[m]; acc[3][m] = acc[3][m] - y[m];
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 8; m++) {
                            acc[1][m] = kg_subj(acc[1][m], y[m]); acc[2][m] = acc[2][m] - y[m]; acc[3][m] = kg_addj(acc[3][m], y[m]);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 1; q < 4; q++) {
                        kg_cmac4v1(acc[q][0], acc[q][1], acc[q][2], acc[q][3], y[0], y[1], y[2], y[3], Q[q - 1]);
                        kg_cmac4v1(acc[q][4], acc[q][5], acc[q][6], acc[q][7], y[4], y[5], y[6], y[7], Q[q - 1]);
                    }
                }
            }
            // The cell before: its wave reductions in item 1 (waves 0..3, which wait ~700 cycles at this barrier otherwise),
            // its merge and store in item 2 (wave 0) -- see the cell end.
            if (HANDOVER && k2 == 1 && prev_out >= 0) reduce_pairs();
            if (HANDOVER && k2 == 2 && prev_out >= 0) merge_store(prev_out, prev_limit);
            KG_STAMP(STAMPS, sti, 8);
            if (STAMPS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            KG_STAMP(STAMPS, sti, 9);
            if (k2 < P - 1) __syncthreads();                   // (the last item's phase B is closed by the cell-end barrier)
            KG_STAMP(STAMPS, sti, 10);
            if (STAMPS) st_item++;
        }
        // search.cpp:486-490: power, first maximum (strict >) over ascending n, running total; n = i + 512 m + 4096 q
        // Row r = m + 8 q holds n = i + 512 r.  `limit` is wave-uniform: a row lies wholly inside the window
        // (512 (r + 1) <= limit: a scalar test; all but the last row for 16368) or needs the per-lane test; the row
        // number, not n, is tracked as the position of the maximum.
        unsigned long long *stc = (STAMPS && st && st_item <= 60) ? st + 16 * st_item + 11 : nullptr;   // slots 11..15 of the cell's last item
        KG_STAMP(STAMPS, stc, 0);
        // Row r = m + 8 q holds n = i + 512 r; `limit` is wave-uniform, so only the rows from limit >> 9 on need the
        // per-lane window test.  The scan costs every lane 32 points and both waves of a SIMD run it together, so it is
        // written for instruction count, not as the serial loop (8.4 instructions per point as compare + two selects +
        // add): the powers (kept: 32 registers), their total in packed pairs, their maximum by v_max3; then the lane's
        // FIRST row holding that maximum -- what the strict-> scan of search.cpp:486-490 ends with -- from a 32-bit
        // row mask built with one compare and one add-with-carry per row (mask = 2 mask + (pw[r] == max), r
        // descending) and a find-first-bit.  All powers zero: row 0, as bi = 0 there.
        const int limit = cur.limit, full_rows = limit >> 9;
        float pw[32];
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const int r = m + 8 * q;
                const cf sq = acc[q][m] * acc[q][m];
                pw[r] = sq.x + sq.y;
                // (limit > 4096 in this kernel: rows 0..7 are always whole; for the others a REAL scalar branch per row --
                // the empty asm keeps the compiler from turning 24 wave-uniform tests into per-lane compares and selects)
                if (r >= 8 && r >= full_rows) {
                    asm volatile("");
                    pw[r] = (i + 512 * r < limit) ? pw[r] : 0.f;
                }
            }
        }
        cf s2 = cf{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 32; r += 2) s2 = s2 + cf{pw[r], pw[r + 1]};
        float sum = s2.x + s2.y, bp = 0.f;
#pragma unroll
        for (int r = 0; r < 32; r += 2) bp = __builtin_fmaxf(bp, __builtin_fmaxf(pw[r], pw[r + 1]));
        unsigned qm[4] = {0, 0, 0, 0};                             // one 8-bit mask per quarter: four independent chains
        // (two wait states between the compare that writes a mask register and the add that takes it as carry: the four quarters'
        // compares first, into four register pairs, then the four adds -- see the scan of acq_correlate_kernel)
#pragma unroll
        for (int m = 7; m >= 0; m--) {
            unsigned long long k0, k1, k2, k3;
            asm("v_cmp_eq_f32_e64 %4, %8, %12\n\tv_cmp_eq_f32_e64 %5, %9, %12\n\t"
                "v_cmp_eq_f32_e64 %6, %10, %12\n\tv_cmp_eq_f32_e64 %7, %11, %12\n\t"
                "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\tv_addc_co_u32_e64 %1, %5, %1, %1, %5\n\t"
                "v_addc_co_u32_e64 %2, %6, %2, %2, %6\n\tv_addc_co_u32_e64 %3, %7, %3, %3, %7"
                : "+v"(qm[0]), "+v"(qm[1]), "+v"(qm[2]), "+v"(qm[3]), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
                : "v"(pw[m]), "v"(pw[m + 8]), "v"(pw[m + 16]), "v"(pw[m + 24]), "v"(bp));
        }
        const unsigned rowmask = (qm[0] | (qm[1] << 8)) | ((qm[2] << 16) | (qm[3] << 24));
        int bi = i + 512 * (int) __builtin_ctz(rowmask | 0x80000000u);
        KG_STAMP(STAMPS, stc, 1);
        // Round 4: NO wave reduction here.  The three dependent DPP chains (maximum, lowest n among its holders, total) took
        // ~680 cycles in every wave, and at a cell end the waves that lose the issue arbitration (4..7) are the ones the
        // barrier waits for.  Every lane hands its three values over through LDS (three stores) instead; waves 0..3 reduce
        // their own and their partner's in item 1 of the next cell and wave 0 merges and stores in item 2, both inside the
        // ~700 cycles those waves wait at the phase-B barrier anyway (profiles/r03_e1b8_stamps_final.txt).
        if (HANDOVER) {
            xch_p[i] = bp; xch_i[i] = bi; xch_s[i] = sum;
        } else {                                       // in place: every wave reduces its own lanes here
            float wmax = bp, wsum = sum;
            kg_wave_max_sum(wmax, wsum);
            const int wn = kg_wave_min(bp == wmax ? bi : 0x7fffffff);
            if ((i & 63) == 0) { red[i >> 6].p = wmax; red[i >> 6].i = wn; red[i >> 6].s = wsum; }
        }
        KG_STAMP(STAMPS, stc, 2);
        if (i == 448) *red_claim = 2 * nslots + claimed;
        __syncthreads();
        KG_STAMP(STAMPS, stc, 3);
        const int nn_idx = __builtin_amdgcn_readfirstlane(*red_claim);
        prev_out = cur.out; prev_limit = limit;
        if (!more) {                                   // the workgroup's last cell: nothing to hide behind
            if (HANDOVER) {
                reduce_pairs();
                __syncthreads();
            }
            merge_store(prev_out, prev_limit);
        }
        KG_STAMP(STAMPS, stc, 4);
        if (!more) break;
        cur = nxt; cur_idx = nxt_idx; nxt_idx = nn_idx;
    }
}

// search.cpp:455,495: best Doppler bin per (block, SV): the serial scan keeps the
// first bin (ascending dop) holding the maximum snr, and only if it is > 0.
// One wave per pair; lane l scans bins l, l+64, ...
__global__ __launch_bounds__(64) void acq_select_kernel(const kg_acq_cell *__restrict__ cells,
                                                       int npairs, int dop_lo, int ndop,
                                                       kg_acq_result *__restrict__ out, int *__restrict__ claim)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p == 0 && lane < 16) claim[lane * ACQ_CLAIM_STRIDE] = 0;          // the correlators' cell counters, for the next launch
    if (p >= npairs) return;
    float bs = 0.f;
    int bd = 0x7fffffff, bidx = 0;
    for (int di = lane; di < ndop; di += 64) {
        const kg_acq_cell c = cells[(size_t) p * ndop + di];
        if (c.snr > bs) { bs = c.snr; bd = di; bidx = c.idx; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float os = __shfl_xor(bs, off);
        const int od = __shfl_xor(bd, off), oi = __shfl_xor(bidx, off);
        if (os > bs || (os == bs && od < bd)) { bs = os; bd = od; bidx = oi; }
    }
    if (lane == 0) {
        kg_acq_result r;
        r.valid = bs > 0.f ? 1 : 0;
        r.snr = r.valid ? bs : 0.f;
        r.dop = r.valid ? dop_lo + bd : 0;
        r.idx = r.valid ? bidx : 0;
        out[p] = r;
    }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct kg_acq {
    kg_ctx *ctx;
    int max_sats, dop_lo, dop_hi, ndop, max_blocks;
    // shape (see the head of this file)
    int nsamples;      // input samples per block that carry signal (<= 4 * fft_len)
    int fft_len;       // N
    int P;             // N / 4096: 4 or 16
    size_t in_stride;  // bytes of one block of int16 IQ staging: nsamples * 4
    int halo;          // H of the code planes: covers every floor((k2 - dop) / P)
    size_t code_len;   // float2 per code spectrum (P planes with halo)
    float2 *d_tabN;        // exp(+2 pi i k / N); the context's table for N = 16384
    bool own_tabN;
    float2 *d_comb;        // [P][8]  combine constants (acq_correlate_kernel)
    float2 *d_quart;       // [P][4]
    float2 *d_comb8;       // [P][4]  combine constants of the 512-thread four-quarter kernel
    int grid8;             // its persistent grid; 0: use acq_correlate_kernel<P, 4> (KIWIGPU_ACQ_E1B8=0)
    float2 *d_code;        // [max_sats][P planes with halo]
    float2 *d_data;        // [max_blocks][P][4096]  layout A
    float2 *d_data_b;      // [max_blocks][P][4096]  layout B (the 512-thread kernel's); null when that kernel is off
    std::vector<char> code_layout;   // per SV: 0 = A, 1 = B
    float2 *d_td;          // [max_blocks][N]  decimated time-domain samples per block
    float2 *d_td_code;     // [N]              same, for the code-table build
    float2 *d_fsub;        // [max_blocks][P][4096] sub-transforms awaiting the radix-P combine
    float2 *d_fsub_code;   // [P][4096]
    uint8_t *d_in;     // [max_blocks][in_stride]    host-input staging
    // Host-buffer Sample() calls return before the copy has run ("enqueue only"), so the
    // caller's samples are first copied into one of a few pinned slots; a slot is reused
    // once the event recorded behind its copy has fired.
    uint8_t *h_pin[4];
    hipEvent_t ev_pin[4];
    int pin_next;
    // batch calls: one pinned region for all blocks of the call, grown on demand
    uint8_t *h_batch; size_t batch_cap; hipEvent_t ev_batch;
    // its transfer runs on a copy-only stream so that it overlaps the previous batch's
    // correlation: ev_batch = transfer done, ev_in_free = the staging area has been consumed
    hipStream_t cstream; hipEvent_t ev_in_free; bool in_free_set;
    uint8_t *d_chips;  // [E1B_CODELEN max]
    kg_acq_cell *d_cells;
    kg_acq_result *d_results;
    int *d_claim;      // [16][ACQ_CLAIM_STRIDE] cell counters of the C/A ([0..8)) and E1B ([8..16)) launches; zero between launches
    std::vector<int> limits, code_set;
    // The (block, SV) pair tables of the last launch live in a slot of the context's staging
    // ring; they are reused while the SV list is unchanged and the slot has not come round.
    std::vector<int> last_sats;
    const acq_pair_desc *d_pairs1, *d_pairs4;   // 4092-window (C/A) and 16368-window (E1B) launches
    unsigned long pairs_seq;                    // ring position when they were staged
    int last_first, last_nblocks, last_nsats, np1, np4, table_nblocks;
    // Stream of the Sample() work: the context's stream, or (opt-in) a second one ordered
    // against it by events.
    hipStream_t fstream;
    bool own_fstream;
    std::vector<hipEvent_t> ev_ready, ev_done;   // data spectrum written / last reader done
    // Block b is covered by events ev_ready[ready_of[b]] / ev_done[done_of[b]] (-1: none yet).
    // A batch call records ONE event for all its blocks: every record and every cross-stream
    // wait is a packet the command processor handles between two kernels (8 blocks per
    // call: 358 -> 288 us per step when the per-block events went away).
    std::vector<int> ready_of, done_of;
    int grid1;
};

// Host mirror of the plane layouts (see the head of this file): A = 256 threads x 16 legs, B = 512 x 8.
static inline __host__ __device__ size_t plane_pos(int k1, int halo_col, int H, int layout)
{
    if (layout == 0) {
        const int t = k1 & 255, j = k1 >> 8, row = 2 * (256 + 2 * H);
        return (size_t) (j >> 1) * row + 2 * (t + halo_col + H) + (j & 1);
    }
    const int t = k1 & 511, j = k1 >> 9, row = 2 * (512 + 2 * H);
    return (size_t) (j >> 1) * row + 2 * (t + halo_col + H) + (j & 1);
}
// float2 per spectrum: layout A's size (the larger: 32 H against 16 H halo elements per plane) is the stride of both
static inline size_t spec_len(int P, int H) { return (size_t) P * 8 * 2 * (256 + 2 * H); }
static inline __host__ __device__ size_t plane_len(int H, int layout) { return layout == 0 ? (size_t) 8 * 2 * (256 + 2 * H) : (size_t) 4 * 2 * (512 + 2 * H); }

// What the reference reads PAST THE END of a code row.  Correlate() forms conj(data[i]) code[sat][N - dop + i] over a row that
// holds the spectrum twice (2 N entries, gps/search.cpp:54, :471): for a negative Doppler bin the last |dop| products index
// 2 N .. 2 N + |dop| - 1 -- the first |dop| bins of the NEXT satellite's row, code[sat + 1][0 .. |dop|) (rows are contiguous in
// the static array; a row nobody wrote is zero) -- not bins 0 .. |dop| - 1 of the satellite's own spectrum, which is what the
// index "(i - dop) mod N" would give.  Found in round 6 by running the reference's own search.cpp:
// 0.3 % of snr at dop = -13, and another noise peak wins for an absent SV.  The wrapped bins N, N + 1, ... of a row are the
// right halo of its last leg in the plane layout, and only a negative bin reads them: that halo is filled from row sat + 1.
__global__ void acq_code_overrun_kernel(float2 *__restrict__ dst, int layout_dst, const float2 *__restrict__ src, int layout_src, int P, int H)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;           // bin t of the next row = "bin N + t" of this one
    if (t >= P * H) return;
    const int k2 = t % P, k1 = t / P;
    const float2 v = src ? src[(size_t) k2 * plane_len(H, layout_src) + plane_pos(k1, 0, H, layout_src)] : make_float2(0.f, 0.f);
    const int T = layout_dst == 0 ? 256 : 512, J = SUB / T;
    dst[(size_t) k2 * plane_len(H, layout_dst) + plane_pos(k1 + T * (J - 1), T, H, layout_dst)] = v;
}

static void to_planes(const float *nat, std::vector<float2> &pl, int P, int H, int layout)
{
    const size_t plane = plane_len(H, layout);
    const int T = layout == 0 ? 256 : 512, J = SUB / T;
    pl.assign(spec_len(P, H), make_float2(0.f, 0.f));
    for (int k = 0; k < P * SUB; k++) {
        const float2 v = make_float2(nat[2 * k], nat[2 * k + 1]);
        const int k1 = k / P, t = k1 & (T - 1), j = k1 / T;
        float2 *p = pl.data() + (size_t) (k % P) * plane;
        p[plane_pos(k1, 0, H, layout)] = v;
        if (t < H) p[plane_pos(t + T * ((j - 1) & (J - 1)), T, H, layout)] = v;
        if (t >= T - H) p[plane_pos(t + T * ((j + 1) & (J - 1)), -T, H, layout)] = v;
    }
}
static void from_planes(const std::vector<float2> &pl, float *nat, int P, int H, int layout)
{
    const size_t plane = plane_len(H, layout);
    for (int k = 0; k < P * SUB; k++) {
        const float2 v = pl[(size_t) (k % P) * plane + plane_pos(k / P, 0, H, layout)];
        nat[2 * k] = v.x; nat[2 * k + 1] = v.y;
    }
}

// exp(+2 pi i k / n) in double, rounded to fp32, exact on the axes (as kg_ctx's tables).
static float2 unit_root(long k, long n)
{
    k %= n;
    if (k < 0) k += n;
    if (k == 0) return make_float2(1.f, 0.f);
    if (4 * k == n) return make_float2(0.f, 1.f);
    if (2 * k == n) return make_float2(-1.f, 0.f);
    if (4 * k == 3 * n) return make_float2(0.f, -1.f);
    const double a = 2.0 * M_PI * (double) k / (double) n;
    return make_float2((float) cos(a), (float) sin(a));
}

template <int SRC>
static int launch_frontend(kg_acq *a, hipStream_t st, const uint8_t *d_src, size_t stride, int nbatch,
                           int nvalid, int nchips, int boc, float2 *d_td, float2 *d_scratch,
                           float2 *d_planes, float2 *d_planes_b, size_t planes_stride, int halo)
{
    kg_ctx *c = a->ctx;
    hipLaunchKernelGGL(acq_frontend_kernel<SRC>, dim3(a->fft_len / FE_TILE, nbatch), dim3(256), 0,
                       st, d_src, stride, nvalid, nchips, boc, d_td, a->fft_len);
    KG_HIP(hipGetLastError());
    hipLaunchKernelGGL(acq_fft_sub_kernel<false>, dim3(a->P, nbatch), dim3(256), 2 * SUB * sizeof(float2),
                       st, (const float2 *) d_td, d_scratch, (const float2 *) c->d_tab4096, a->P,
                       (unsigned long long *) nullptr);
    KG_HIP(hipGetLastError());
    if (a->P == 4)
        hipLaunchKernelGGL(acq_fft_combine_kernel<4>, dim3(SUB / 256, nbatch), dim3(256), 0, st,
                           (const float2 *) d_scratch, d_planes, d_planes_b, planes_stride,
                           (const float2 *) a->d_tabN, halo);
    else
        hipLaunchKernelGGL(acq_fft_combine_kernel<16>, dim3(SUB / 256, nbatch), dim3(256), 0, st,
                           (const float2 *) d_scratch, d_planes, d_planes_b, planes_stride,
                           (const float2 *) a->d_tabN, halo);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

static int acq_init(kg_acq *a)
{
    kg_ctx *ctx = a->ctx;
    const int max_sats = a->max_sats, max_blocks = a->max_blocks, N = a->fft_len, P = a->P;
    {
        const int m = (a->dop_hi > -a->dop_lo ? a->dop_hi : -a->dop_lo);
        a->halo = (m + P - 1) / P + 2;
        a->code_len = spec_len(P, a->halo);
    }
    a->limits.assign(max_sats, 0); a->code_set.assign(max_sats, 0); a->code_layout.assign(max_sats, 0);
    a->last_first = a->last_nblocks = a->last_nsats = a->np1 = a->np4 = a->table_nblocks = 0;
    a->d_pairs1 = a->d_pairs4 = nullptr; a->pairs_seq = 0;
    a->in_stride = (size_t) a->nsamples * 4;
    const size_t spec = sizeof(float2) * N;
    if (N == 16384) { a->d_tabN = ctx->d_tab16384; a->own_tabN = false; }
    else {
        std::vector<float2> h(N);
        for (int k = 0; k < N; k++) h[k] = unit_root(k, N);
        KG_HIP(hipMalloc((void **) &a->d_tabN, spec));
        a->own_tabN = true;
        KG_HIP(hipMemcpy(a->d_tabN, h.data(), spec, hipMemcpyHostToDevice));
    }
    {
        const long R = N / 256;
        std::vector<float2> comb(P * 8, make_float2(1.f, 0.f)), quart(P * 4);
        static const int mult[6] = {1, 2, 3, 4, 8, 12};
        for (int k2 = 0; k2 < P; k2++) {
            for (int i = 0; i < 6; i++) comb[8 * k2 + i] = unit_root((long) mult[i] * k2, R);
            for (int q = 0; q < 4; q++) quart[4 * k2 + q] = unit_root((long) q * k2, P);
        }
        KG_HIP(hipMalloc((void **) &a->d_comb, sizeof(float2) * comb.size()));
        KG_HIP(hipMalloc((void **) &a->d_quart, sizeof(float2) * quart.size()));
        KG_HIP(hipMemcpy(a->d_comb, comb.data(), sizeof(float2) * comb.size(), hipMemcpyHostToDevice));
        KG_HIP(hipMemcpy(a->d_quart, quart.data(), sizeof(float2) * quart.size(), hipMemcpyHostToDevice));
        std::vector<float2> comb8(P * 4);
        for (int k2 = 0; k2 < P; k2++)
            for (int k = 0; k < 4; k++) comb8[4 * k2 + k] = unit_root((long) (k + 1) * k2, N / 512);
        KG_HIP(hipMalloc((void **) &a->d_comb8, sizeof(float2) * comb8.size()));
        KG_HIP(hipMemcpy(a->d_comb8, comb8.data(), sizeof(float2) * comb8.size(), hipMemcpyHostToDevice));
    }
    KG_HIP(hipMalloc((void **) &a->d_code, sizeof(float2) * a->code_len * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_data, spec * max_blocks));
    KG_HIP(hipMalloc((void **) &a->d_td, spec * max_blocks));
    KG_HIP(hipMalloc((void **) &a->d_td_code, spec));
    KG_HIP(hipMalloc((void **) &a->d_fsub, spec * max_blocks));
    KG_HIP(hipMalloc((void **) &a->d_fsub_code, spec));
    KG_HIP(hipMalloc((void **) &a->d_in, a->in_stride * max_blocks));
    for (int k = 0; k < 4; k++) {
        KG_HIP(hipHostMalloc((void **) &a->h_pin[k], a->in_stride, hipHostMallocDefault));
        KG_HIP(hipEventCreateWithFlags(&a->ev_pin[k], hipEventDisableTiming));
    }
    KG_HIP(hipEventCreateWithFlags(&a->ev_batch, hipEventDisableTiming));
    KG_HIP(hipEventCreateWithFlags(&a->ev_in_free, hipEventDisableTiming));
    { const int rc_ = kg_stream_get(ctx->device, &a->cstream); if (rc_) return rc_; }
    KG_HIP(hipMalloc((void **) &a->d_chips, 8192));
    KG_HIP(hipMalloc((void **) &a->d_cells, sizeof(kg_acq_cell) * (size_t) max_blocks * max_sats * a->ndop));
    KG_HIP(hipMalloc((void **) &a->d_results, sizeof(kg_acq_result) * (size_t) max_blocks * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_claim, sizeof(int) * 16 * ACQ_CLAIM_STRIDE));
    KG_HIP(hipMemset(a->d_claim, 0, sizeof(int) * 16 * ACQ_CLAIM_STRIDE));
    {
        // Sample() and Correlate() run in order on the context's stream.  A second stream
        // (KIWIGPU_ACQ_FRONT_STREAM=1) lets block b+1's front end overlap block b's
        // correlation: 60 instead of 65 us per 1-block step on MI355X, but when the front-end
        // kernels are dispatched together with the persistent correlator the latter can
        // spend its first pass at one workgroup per CU (16 blocks per step: 887 us against
        // 528 us in order; profiles/r01_streams.txt), so the overlap is opt-in.
        const char *e = kg_tuning_env("KIWIGPU_ACQ_FRONT_STREAM");
        if (e && e[0] == '1') {
            { const int rc_ = kg_stream_get(ctx->device, &a->fstream); if (rc_) return rc_; }
            a->own_fstream = true;
        }
    }
    a->ready_of.assign(max_blocks, -1); a->done_of.assign(max_blocks, -1);
    a->ev_ready.reserve(max_blocks); a->ev_done.reserve(max_blocks);
    for (int b = 0; b < max_blocks; b++) {
        hipEvent_t e1, e2;
        KG_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        a->ev_ready.push_back(e1);
        KG_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        a->ev_done.push_back(e2);
    }
    KG_HIP(hipMemset(a->d_data, 0, spec * max_blocks));
    a->d_data_b = nullptr;      // layout B: allocated (and from then on written by every Sample()) when the first code with a
                                // window longer than 4096 lags is set (ensure_data_b) -- a C/A-only searcher never pays for it
    KG_HIP(hipFuncSetAttribute((const void *) acq_fft_sub_kernel<false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SUB * sizeof(float2)));
    KG_HIP(hipFuncSetAttribute((const void *) acq_fft_sub_kernel<true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SUB * sizeof(float2)));
    // persistent grid: resident workgroups per CU x CUs, rounded to a multiple of 8 (XCDs)
    int occ1 = 0;
#define ACQ_SETUP(PP)                                                                                          \
    KG_HIP(hipFuncSetAttribute((const void *) acq_correlate_kernel<PP, 1, true>,                              \
                               hipFuncAttributeMaxDynamicSharedMemorySize, ACQ_LDS_BYTES));                   \
    KG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, acq_correlate_kernel<PP, 1, true>, 256,        \
                                                        ACQ_LDS_BYTES));
    if (P == 4) { ACQ_SETUP(4) } else { ACQ_SETUP(16) }
#undef ACQ_SETUP
    if (occ1 < 1) occ1 = 1;
    if (const char *e = kg_tuning_env("KIWIGPU_ACQ_WGS_PER_CU")) {      // experiments: fewer resident workgroups per CU
        const int v = atoi(e);
        if (v >= 1 && v < occ1) occ1 = v;
    }
    a->grid1 = (ctx->num_cus * occ1) & ~7;
    // the 16368-lag window runs on the 512-thread form (one workgroup = eight waves per CU, two per SIMD); round 3's
    // 256-thread four-accumulator form (acq_correlate_kernel<P, 4>) is no longer instantiated
    if (P == 4)
        KG_HIP(hipFuncSetAttribute((const void *) acq_correlate8_kernel<4>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, ACQ8_LDS_BYTES));
    else
        KG_HIP(hipFuncSetAttribute((const void *) acq_correlate8_kernel<16>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, ACQ8_LDS_BYTES));
    a->grid8 = ctx->num_cus & ~7;
    if (a->grid8 < 8) a->grid8 = 8;
    if (a->grid1 < 8) a->grid1 = 8;
    return KG_OK;
}

extern "C" {

int kg_acq_create_shape(kg_ctx *ctx, int max_sats, int dop_lo, int dop_hi, int max_blocks, int nsamples,
                        int fft_len, kg_acq **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_acq_create: out is null");
    *out = nullptr;
    KG_REQUIRE(max_sats >= 1 && max_sats <= 4096, KG_ERR_INVALID, "kg_acq_create: max_sats %d", max_sats);
    KG_REQUIRE(dop_lo <= dop_hi && dop_lo >= -1000 && dop_hi <= 1000, KG_ERR_INVALID,
               "kg_acq_create: Doppler range %d..%d (supported: within -1000..1000)", dop_lo, dop_hi);
    KG_REQUIRE(max_blocks >= 1 && max_blocks <= 65535, KG_ERR_INVALID, "kg_acq_create: max_blocks %d",
               max_blocks);
    KG_REQUIRE(fft_len == 16384 || fft_len == 65536, KG_ERR_INVALID,
               "kg_acq_create: fft_len %d (supported: 16384, 65536)", fft_len);
    KG_REQUIRE(nsamples >= 8 && nsamples <= DECIM * fft_len && nsamples % 8 == 0, KG_ERR_INVALID,
               "kg_acq_create: nsamples %d (a multiple of 8, at most %d)", nsamples, DECIM * fft_len);
    KG_REQUIRE((size_t) max_blocks * fft_len < ((size_t) 1 << 31), KG_ERR_INVALID,
               "kg_acq_create: max_blocks %d x fft_len %d exceeds the 2^31-element offset range", max_blocks, fft_len);
    kg_acq *a = new (std::nothrow) kg_acq();       // value-initialised: every pointer null
    KG_REQUIRE(a != nullptr, KG_ERR_NOMEM, "kg_acq_create: alloc");
    a->ctx = ctx; a->max_sats = max_sats; a->dop_lo = dop_lo; a->dop_hi = dop_hi;
    a->ndop = dop_hi - dop_lo + 1; a->max_blocks = max_blocks;
    a->nsamples = nsamples; a->fft_len = fft_len; a->P = fft_len / SUB;
    a->fstream = ctx->stream;
    rc = acq_init(a);
    if (rc != KG_OK) { kg_acq_destroy(a); return rc; }     // frees whatever was allocated
    *out = a;
    return KG_OK;
}

int kg_acq_create(kg_ctx *ctx, int max_sats, int dop_lo, int dop_hi, int max_blocks, kg_acq **out)
{
    return kg_acq_create_shape(ctx, max_sats, dop_lo, dop_hi, max_blocks, KG_ACQ_NSAMPLES, KG_ACQ_FFT_LEN, out);
}

int kg_acq_nsamples(kg_acq *a) { return a ? a->nsamples : KG_ERR_INVALID; }
int kg_acq_fft_len(kg_acq *a) { return a ? a->fft_len : KG_ERR_INVALID; }

void kg_acq_destroy(kg_acq *a)
{
    if (!a) return;
    (void) hipSetDevice(a->ctx->device);
    if (a->fstream) (void) hipStreamSynchronize(a->fstream);
    (void) hipStreamSynchronize(a->ctx->stream);
    for (size_t b = 0; b < a->ev_ready.size(); b++) (void) hipEventDestroy(a->ev_ready[b]);
    for (size_t b = 0; b < a->ev_done.size(); b++) (void) hipEventDestroy(a->ev_done[b]);
    if (a->own_fstream) kg_stream_put(a->ctx->device, a->fstream);
    if (a->own_tabN) (void) hipFree(a->d_tabN);
    (void) hipFree(a->d_comb); (void) hipFree(a->d_quart); (void) hipFree(a->d_comb8);   // hipFree(nullptr) is a no-op
    (void) hipFree(a->d_code); (void) hipFree(a->d_data); (void) hipFree(a->d_data_b); (void) hipFree(a->d_td);
    (void) hipFree(a->d_td_code);
    (void) hipFree(a->d_fsub); (void) hipFree(a->d_fsub_code);
    (void) hipFree(a->d_in); (void) hipFree(a->d_chips);
    for (int k = 0; k < 4; k++) {
        if (a->h_pin[k]) (void) hipHostFree(a->h_pin[k]);
        if (a->ev_pin[k]) (void) hipEventDestroy(a->ev_pin[k]);
    }
    if (a->h_batch) (void) hipHostFree(a->h_batch);
    if (a->ev_batch) (void) hipEventDestroy(a->ev_batch);
    if (a->ev_in_free) (void) hipEventDestroy(a->ev_in_free);
    if (a->cstream) { (void) hipStreamSynchronize(a->cstream); kg_stream_put(a->ctx->device, a->cstream); }
    (void) hipFree(a->d_cells); (void) hipFree(a->d_results); (void) hipFree(a->d_claim);
    delete a;
}

// The 512-thread correlator reads the data spectra in layout B.  The second copy exists only once an SV needs it: the
// blocks sampled before that moment (none in the reference's order, SearchInit before Sample) are converted through the
// host, once; afterwards every Sample() writes both layouts in its combine kernel.
static int ensure_data_b(kg_acq *a)
{
    if (a->d_data_b || a->grid8 == 0) return KG_OK;
    const size_t n = (size_t) a->fft_len * a->max_blocks;
    KG_HIP(hipStreamSynchronize(a->fstream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    float2 *d = nullptr;
    KG_HIP(hipMalloc((void **) &d, sizeof(float2) * n));
    std::vector<float2> pl(a->fft_len), plb, all(n);
    std::vector<float> nat(2 * (size_t) a->fft_len);
    if (hipMemcpy(all.data(), a->d_data, sizeof(float2) * n, hipMemcpyDeviceToHost) != hipSuccess) {
        (void) hipFree(d);
        KG_REQUIRE(false, KG_ERR_HIP, "ensure_data_b: download of the data spectra failed");
    }
    for (int b = 0; b < a->max_blocks; b++) {
        pl.assign(all.begin() + (size_t) b * a->fft_len, all.begin() + (size_t) (b + 1) * a->fft_len);
        from_planes(pl, nat.data(), a->P, 0, 0);
        to_planes(nat.data(), plb, a->P, 0, 1);
        memcpy(all.data() + (size_t) b * a->fft_len, plb.data(), sizeof(float2) * a->fft_len);
    }
    if (hipMemcpy(d, all.data(), sizeof(float2) * n, hipMemcpyHostToDevice) != hipSuccess) {
        (void) hipFree(d);
        KG_REQUIRE(false, KG_ERR_HIP, "ensure_data_b: upload of layout B failed");
    }
    a->d_data_b = d;
    return KG_OK;
}

static int set_limit(kg_acq *a, int sat, int limit)
{
    KG_REQUIRE(sat >= 0 && sat < a->max_sats, KG_ERR_INVALID, "sat %d out of range (0..%d)", sat,
               a->max_sats - 1);
    KG_REQUIRE(limit >= 1 && limit <= 4 * SUB, KG_ERR_INVALID, "limit %d out of range (1..%d)", limit,
               4 * SUB);
    const int layout = (limit > SUB && a->grid8 > 0) ? 1 : 0;        // which kernel will read this SV's code spectrum
    if (layout) {
        // layout B of the data spectra first: if it cannot be had the SV keeps what it had (an earlier code with its own
        // limit and layout stays searchable; nothing points the 512-thread kernel at a buffer that does not exist)
        int rc = ensure_data_b(a);
        if (rc) return rc;
    }
    a->limits[sat] = limit;
    a->code_layout[sat] = layout;
    a->code_set[sat] = 1;
    a->last_sats.clear();        // force the pair tables to be rebuilt
    return KG_OK;
}

// Row `sat` was just written (or row sat + 1 was): its beyond-the-end bins are row sat + 1's first ones (zero when that row was
// never set, as in the reference's static array).  On the context's stream.
static int acq_patch_overrun(kg_acq *a, int sat)
{
    if (sat < 0 || sat >= a->max_sats || !a->code_set[sat]) return KG_OK;
    const bool have_next = sat + 1 < a->max_sats && a->code_set[sat + 1];
    const int n = a->P * a->halo;
    hipLaunchKernelGGL(acq_code_overrun_kernel, dim3((n + 63) / 64), dim3(64), 0, a->ctx->stream,
                       a->d_code + (size_t) sat * a->code_len, a->code_layout[sat],
                       have_next ? (const float2 *) (a->d_code + (size_t) (sat + 1) * a->code_len) : (const float2 *) nullptr,
                       have_next ? a->code_layout[sat + 1] : 0, a->P, a->halo);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_acq_set_code(kg_acq *a, int sat, const uint8_t *chips, int nchips, int boc, int limit)
{
    KG_REQUIRE(a && chips, KG_ERR_INVALID, "kg_acq_set_code: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(nchips >= 1 && nchips <= 8192, KG_ERR_INVALID, "kg_acq_set_code: nchips %d", nchips);
    for (int i = 0; i < nchips; i++)
        KG_REQUIRE(chips[i] <= 1, KG_ERR_INVALID, "kg_acq_set_code: chips[%d] = %d is not 0/1", i, chips[i]);
    if ((rc = set_limit(a, sat, limit)) != KG_OK) return rc;
    KG_HIP(hipMemcpyAsync(a->d_chips, chips, nchips, hipMemcpyHostToDevice, a->ctx->stream));
    // the replica covers all DECIM * N samples, as the reference's covers NSAMPLES (:250)
    float2 *d_sv = a->d_code + (size_t) sat * a->code_len;
    rc = launch_frontend<SRC_CHIPS>(a, a->ctx->stream, a->d_chips, 0, 1, DECIM * a->fft_len, nchips, boc ? 1 : 0,
                                    a->d_td_code, a->d_fsub_code, a->code_layout[sat] ? nullptr : d_sv,
                                    a->code_layout[sat] ? d_sv : nullptr, a->code_len, a->halo);
    if (rc) return rc;
    if ((rc = acq_patch_overrun(a, sat)) || (rc = acq_patch_overrun(a, sat - 1))) return rc;
    KG_HIP(hipStreamSynchronize(a->ctx->stream));       // chips buffer is reused per call
    return KG_OK;
}

int kg_acq_set_code_fft(kg_acq *a, int sat, const float *code_fft, int limit)
{
    KG_REQUIRE(a && code_fft, KG_ERR_INVALID, "kg_acq_set_code_fft: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    if ((rc = set_limit(a, sat, limit)) != KG_OK) return rc;
    std::vector<float2> pl;
    to_planes(code_fft, pl, a->P, a->halo, a->code_layout[sat]);
    KG_HIP(hipMemcpyAsync(a->d_code + (size_t) sat * a->code_len, pl.data(), sizeof(float2) * a->code_len,
                          hipMemcpyHostToDevice, a->ctx->stream));
    if ((rc = acq_patch_overrun(a, sat)) || (rc = acq_patch_overrun(a, sat - 1))) return rc;
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

static int get_planes(kg_acq *a, const float2 *d, float *nat, int H, int layout)
{
    std::vector<float2> pl(spec_len(a->P, H));
    KG_HIP(hipStreamSynchronize(a->fstream));
    KG_HIP(hipMemcpyAsync(pl.data(), d, sizeof(float2) * pl.size(), hipMemcpyDeviceToHost, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    from_planes(pl, nat, a->P, H, layout);
    return KG_OK;
}

int kg_acq_get_code_fft(kg_acq *a, int sat, float *code_fft)
{
    KG_REQUIRE(a && code_fft, KG_ERR_INVALID, "kg_acq_get_code_fft: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(sat >= 0 && sat < a->max_sats, KG_ERR_INVALID, "kg_acq_get_code_fft: sat %d", sat);
    KG_REQUIRE(a->code_set[sat], KG_ERR_STATE, "kg_acq_get_code_fft: no code set for sat %d", sat);
    return get_planes(a, a->d_code + (size_t) sat * a->code_len, code_fft, a->halo, a->code_layout[sat]);
}

static int check_block(kg_acq *a, int block, const void *p, const char *who)
{
    KG_REQUIRE(a && p, KG_ERR_INVALID, "%s: null argument", who);
    KG_REQUIRE(block >= 0 && block < a->max_blocks, KG_ERR_INVALID, "%s: block %d out of range (0..%d)",
               who, block, a->max_blocks - 1);
    return kg_ctx_use(a->ctx);
}

// Caller's host samples -> the block's device staging area, through a pinned slot.
// One block's input, pinned host memory -> its staging slot in HBM: read by a kernel (the pinned buffer is
// mapped into the device's address space), 16 bytes per lane.  An enqueued hipMemcpyAsync of this size
// costs 250 us more under the HIP runtime PyTorch 2.10 bundles (ROCm 7.0) than under the system's 7.2
// (`tools/time_acq_host.py`, synchronous int16 IQ: 374 against 128 us per block); a kernel costs the
// same under both.
__global__ __launch_bounds__(256) void acq_stage_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

static int stage_host_block(kg_acq *a, uint8_t *d_stage, const void *src, size_t bytes)
{
    const int k = a->pin_next;
    a->pin_next = (k + 1) & 3;
    KG_HIP(hipEventSynchronize(a->ev_pin[k]));           // never-recorded events are complete
    memcpy(a->h_pin[k], src, bytes);
    if ((bytes & 15) == 0 && bytes >= 65536) {
        const size_t n16 = bytes / 16;
        hipLaunchKernelGGL(acq_stage_copy_kernel, dim3((unsigned) ((n16 + 255) / 256)), dim3(256), 0, a->fstream,
                           (const uint4 *) a->h_pin[k], (uint4 *) d_stage, n16);
        KG_HIP(hipGetLastError());
    } else {
        KG_HIP(hipMemcpyAsync(d_stage, a->h_pin[k], bytes, hipMemcpyHostToDevice, a->fstream));
    }
    KG_HIP(hipEventRecord(a->ev_pin[k], a->fstream));
    return KG_OK;
}
// Behind every front end that read d_in: the next batch copy (on cstream) waits for it.
static int staging_consumed(kg_acq *a)
{
    KG_HIP(hipEventRecord(a->ev_in_free, a->fstream));
    a->in_free_set = true;
    return KG_OK;
}

// Front-stream bracket for everything that (re)writes the data spectra of blocks
// b .. b+n-1.  Events are only ever re-recorded on the same stream, so waiting on a shared
// event that has since been recorded again waits for later work, never for less.
static int wait_once(hipStream_t st, const std::vector<hipEvent_t> &ev, const std::vector<int> &of, int b, int n)
{
    for (int i = b; i < b + n; i++) {
        const int e = of[i];
        if (e < 0) continue;
        bool seen = false;
        for (int k = b; k < i && !seen; k++) seen = of[k] == e;
        if (!seen) KG_HIP(hipStreamWaitEvent(st, ev[e], 0));
    }
    return KG_OK;
}
static int front_begin(kg_acq *a, int b, int n = 1)
{
    if (!a->own_fstream) return KG_OK;                                     // same stream: in order anyway
    return wait_once(a->fstream, a->ev_done, a->done_of, b, n);           // WAR
}
static int front_end(kg_acq *a, int b, int n = 1)
{
    if (!a->own_fstream) return KG_OK;
    KG_HIP(hipEventRecord(a->ev_ready[b], a->fstream));
    for (int i = b; i < b + n; i++) a->ready_of[i] = b;
    return KG_OK;
}

}  // extern "C"

template <int SRC>
static int sample_dev(kg_acq *a, int first, int nblocks, const void *d_src, size_t stride)
{
    int rc;
    if ((rc = front_begin(a, first, nblocks)) != KG_OK) return rc;
    const size_t off = (size_t) first * a->fft_len;
    rc = launch_frontend<SRC>(a, a->fstream, (const uint8_t *) d_src, stride, nblocks, a->nsamples, 0, 0,
                              a->d_td + off, a->d_fsub + off, a->d_data + off, a->d_data_b ? a->d_data_b + off : nullptr,
                              a->fft_len, 0);
    if (rc) return rc;
    return front_end(a, first, nblocks);
}

extern "C" {

int kg_acq_sample_bits_dev(kg_acq *a, int block, const void *d_packed)
{
    int rc = check_block(a, block, d_packed, "kg_acq_sample_bits_dev");
    if (rc) return rc;
    return sample_dev<SRC_BITS>(a, block, 1, d_packed, 0);
}

int kg_acq_sample_bits(kg_acq *a, int block, const uint8_t *packed)
{
    int rc = check_block(a, block, packed, "kg_acq_sample_bits");
    if (rc) return rc;
    uint8_t *stage = a->d_in + a->in_stride * block;
    if ((rc = stage_host_block(a, stage, packed, (size_t) a->nsamples / 8))) return rc;
    if ((rc = sample_dev<SRC_BITS>(a, block, 1, stage, 0)) != KG_OK) return rc;
    return staging_consumed(a);
}

int kg_acq_sample_iq16_dev(kg_acq *a, int block, const void *d_iq)
{
    int rc = check_block(a, block, d_iq, "kg_acq_sample_iq16_dev");
    if (rc) return rc;
    KG_REQUIRE(((uintptr_t) d_iq & 3) == 0, KG_ERR_INVALID, "kg_acq_sample_iq16_dev: pointer not 4-byte aligned");
    return sample_dev<SRC_IQ16>(a, block, 1, d_iq, 0);
}

int kg_acq_sample_iq16_batch_dev(kg_acq *a, int first, int nblocks, const void *d_iq, size_t stride_bytes)
{
    int rc = check_block(a, first, d_iq, "kg_acq_sample_iq16_batch_dev");
    if (rc) return rc;
    KG_REQUIRE(nblocks >= 1 && first + nblocks <= a->max_blocks, KG_ERR_INVALID,
               "kg_acq_sample_iq16_batch_dev: blocks %d..%d (max %d)", first, first + nblocks - 1, a->max_blocks);
    KG_REQUIRE(((uintptr_t) d_iq & 3) == 0 && (stride_bytes & 3) == 0, KG_ERR_INVALID,
               "kg_acq_sample_iq16_batch_dev: pointer/stride not 4-byte aligned");
    return sample_dev<SRC_IQ16>(a, first, nblocks, d_iq, stride_bytes);
}

int kg_acq_sample_iq16_batch(kg_acq *a, int first, int nblocks, const int16_t *iq, size_t stride_samples)
{
    int rc = check_block(a, first, iq, "kg_acq_sample_iq16_batch");
    if (rc) return rc;
    KG_REQUIRE(nblocks >= 1 && first + nblocks <= a->max_blocks, KG_ERR_INVALID,
               "kg_acq_sample_iq16_batch: blocks %d..%d (max %d)", first, first + nblocks - 1, a->max_blocks);
    KG_REQUIRE(stride_samples >= (size_t) a->nsamples, KG_ERR_INVALID, "kg_acq_sample_iq16_batch: stride %zu < %d samples",
               stride_samples, a->nsamples);
    const size_t bytes = a->in_stride * (size_t) nblocks;
    KG_HIP(hipEventSynchronize(a->ev_batch));            // the previous batch has left the pinned region
    if (bytes > a->batch_cap) {
        if (a->h_batch) KG_HIP(hipHostFree(a->h_batch));
        a->h_batch = nullptr; a->batch_cap = 0;
        KG_HIP(hipHostMalloc((void **) &a->h_batch, bytes, hipHostMallocDefault));
        a->batch_cap = bytes;
    }
    for (int b = 0; b < nblocks; b++)
        memcpy(a->h_batch + a->in_stride * b, iq + 2 * stride_samples * b, a->in_stride);
    uint8_t *stage = a->d_in + a->in_stride * first;
    // copy stream: after the front end that last read the staging area; front-end stream: after the copy
    if (a->in_free_set) KG_HIP(hipStreamWaitEvent(a->cstream, a->ev_in_free, 0));
    KG_HIP(hipMemcpyAsync(stage, a->h_batch, bytes, hipMemcpyHostToDevice, a->cstream));
    KG_HIP(hipEventRecord(a->ev_batch, a->cstream));
    KG_HIP(hipStreamWaitEvent(a->fstream, a->ev_batch, 0));
    rc = sample_dev<SRC_IQ16>(a, first, nblocks, stage, a->in_stride);
    if (rc) return rc;
    return staging_consumed(a);
}

int kg_acq_sample_iq16(kg_acq *a, int block, const int16_t *iq)
{
    int rc = check_block(a, block, iq, "kg_acq_sample_iq16");
    if (rc) return rc;
    uint8_t *stage = a->d_in + a->in_stride * block;
    if ((rc = stage_host_block(a, stage, iq, a->in_stride))) return rc;
    if ((rc = sample_dev<SRC_IQ16>(a, block, 1, stage, 0)) != KG_OK) return rc;
    return staging_consumed(a);
}

int kg_acq_set_data_fft(kg_acq *a, int block, const float *data_fft)
{
    int rc = check_block(a, block, data_fft, "kg_acq_set_data_fft");
    if (rc) return rc;
    std::vector<float2> pl, plb;
    to_planes(data_fft, pl, a->P, 0, 0);
    if ((rc = front_begin(a, block)) != KG_OK) return rc;
    KG_HIP(hipMemcpyAsync(a->d_data + (size_t) block * a->fft_len, pl.data(), sizeof(float2) * a->fft_len,
                          hipMemcpyHostToDevice, a->fstream));
    if (a->d_data_b) {
        to_planes(data_fft, plb, a->P, 0, 1);
        KG_HIP(hipMemcpyAsync(a->d_data_b + (size_t) block * a->fft_len, plb.data(), sizeof(float2) * a->fft_len,
                              hipMemcpyHostToDevice, a->fstream));
    }
    if ((rc = front_end(a, block)) != KG_OK) return rc;
    KG_HIP(hipStreamSynchronize(a->fstream));
    return KG_OK;
}

int kg_acq_get_data_fft(kg_acq *a, int block, float *data_fft)
{
    int rc = check_block(a, block, data_fft, "kg_acq_get_data_fft");
    if (rc) return rc;
    return get_planes(a, a->d_data + (size_t) block * a->fft_len, data_fft, 0, 0);
}

int kg_acq_get_data_td(kg_acq *a, int block, float *td)
{
    int rc = check_block(a, block, td, "kg_acq_get_data_td");
    if (rc) return rc;
    KG_HIP(hipStreamSynchronize(a->fstream));
    KG_HIP(hipMemcpyAsync(td, a->d_td + (size_t) block * a->fft_len, sizeof(float2) * a->fft_len,
                          hipMemcpyDeviceToHost, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

}  // extern "C"

template <int P, int NQ, bool STAMPS>
static void launch_correlate(kg_acq *a, hipStream_t st, int first, const acq_pair_desc *d_pairs, int npairs,
                             unsigned long long *d_stamps)
{
    static_assert(NQ == 1, "the four-accumulator form is not built any more (acq_correlate8_kernel took its place)");
    const int grid = a->grid1;
    const acq_walk w = {npairs, a->ndop, a->dop_lo};
    hipLaunchKernelGGL((acq_correlate_kernel<P, NQ, true, STAMPS>), dim3(grid), dim3(256), ACQ_LDS_BYTES, st,
                       (const float2 *) (a->d_data + (size_t) first * a->fft_len), (const float2 *) a->d_code,
                       (const float2 *) a->ctx->d_tab4096, (const float2 *) a->d_tabN,
                       (const float2 *) a->d_comb, (const float2 *) a->d_quart, d_pairs,
                       a->d_claim + (NQ == 1 ? 0 : 8 * ACQ_CLAIM_STRIDE), w, a->halo, a->d_cells, d_stamps);
}

extern "C" {

int kg_acq_correlate_blocks_async(kg_acq *a, int first, int nblocks, const int *sats, int nsats)
{
    KG_REQUIRE(a && sats, KG_ERR_INVALID, "kg_acq_correlate_async: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(nblocks >= 1 && first >= 0 && first + nblocks <= a->max_blocks, KG_ERR_INVALID,
               "kg_acq_correlate_async: blocks %d..%d (max %d)", first, first + nblocks - 1,
               a->max_blocks);
    KG_REQUIRE(nsats >= 1 && nsats <= a->max_sats, KG_ERR_INVALID,
               "kg_acq_correlate_async: nsats %d (max %d)", nsats, a->max_sats);
    hipStream_t st = a->ctx->stream;
    // The tables hold block indices relative to `first` (the kernel gets data + first).  They are
    // reused while the SV list is the same and the ring slot they were staged in is still theirs
    // (fewer than half a ring of uploads since); otherwise they are staged again -- a memcpy into
    // pinned memory and an enqueued transfer, never a stream synchronisation (the per-SV calling
    // pattern of the reference's SearchTask loop changes the list on every call).
    bool same = (int) a->last_sats.size() == nsats && a->table_nblocks == nblocks &&
                a->ctx->ring_next - a->pairs_seq < KG_RING_SLOTS / 2;
    for (int i = 0; same && i < nsats; i++) same = a->last_sats[i] == sats[i];
    if (!same) {
        // (block, SV) pairs are dealt to the 8 XCD groups round-robin (pair p -> group p & 7); a
        // pair's ndop cells stay together.  Separate tables for the 4092-window (C/A) and the
        // 16368-window (E1B) kernels, staged as one upload: [pairs1 | pairs4].
        for (int i = 0; i < nsats; i++) {
            KG_REQUIRE(sats[i] >= 0 && sats[i] < a->max_sats, KG_ERR_INVALID,
                       "kg_acq_correlate_async: sats[%d] = %d out of range", i, sats[i]);
            KG_REQUIRE(a->code_set[sats[i]], KG_ERR_STATE,
                       "kg_acq_correlate_async: no code set for sat %d", sats[i]);
        }
        std::vector<acq_pair_desc> tab[2];
        for (int blk = 0; blk < nblocks; blk++)
            for (int i = 0; i < nsats; i++) {
                const int sat = sats[i];
                acq_pair_desc pd;
                pd.data_off = blk * a->fft_len;
                pd.code_off = (int) (sat * a->code_len);
                pd.limit = a->limits[sat];
                pd.out = (blk * nsats + i) * a->ndop;
                tab[a->limits[sat] <= SUB ? 0 : 1].push_back(pd);
            }
        a->np1 = (int) tab[0].size(); a->np4 = (int) tab[1].size();
        tab[0].insert(tab[0].end(), tab[1].begin(), tab[1].end());
        void *d = nullptr;
        if ((rc = kg_ctx_stage(a->ctx, tab[0].data(), sizeof(acq_pair_desc) * tab[0].size(), &d)) != KG_OK) return rc;
        a->d_pairs1 = (const acq_pair_desc *) d;
        a->d_pairs4 = a->d_pairs1 + a->np1;
        a->pairs_seq = a->ctx->ring_next;
        a->last_sats.assign(sats, sats + nsats);
        a->table_nblocks = nblocks;
        // a table too large for a ring slot went through the context's scratch: never reused
        if (sizeof(acq_pair_desc) * tab[0].size() > KG_RING_SLOT_BYTES) a->last_sats.clear();
    }
    if (a->own_fstream && (rc = wait_once(st, a->ev_ready, a->ready_of, first, nblocks)) != KG_OK) return rc;   // RAW
    if (a->np1 > 0) {
        if (a->P == 4) launch_correlate<4, 1, false>(a, st, first, a->d_pairs1, a->np1, nullptr);
        else launch_correlate<16, 1, false>(a, st, first, a->d_pairs1, a->np1, nullptr);
        KG_HIP(hipGetLastError());
    }
    if (a->np4 > 0) {
        {
            KG_REQUIRE(a->d_data_b != nullptr, KG_ERR_STATE, "kg_acq_correlate_async: a long-window code without its layout of the data spectra");
            const acq_walk w = {a->np4, a->ndop, a->dop_lo};
            if (a->P == 4)
                hipLaunchKernelGGL(acq_correlate8_kernel<4>, dim3(a->grid8), dim3(512), ACQ8_LDS_BYTES, st,
                                   (const float2 *) (a->d_data_b + (size_t) first * a->fft_len), (const float2 *) a->d_code,
                                   (const float2 *) a->ctx->d_tab4096, (const float2 *) a->d_tabN, (const float2 *) a->d_comb8,
                                   (const float2 *) a->d_quart, a->d_pairs4, a->d_claim + 8 * ACQ_CLAIM_STRIDE, w, a->halo, a->d_cells);
            else
                hipLaunchKernelGGL(acq_correlate8_kernel<16>, dim3(a->grid8), dim3(512), ACQ8_LDS_BYTES, st,
                                   (const float2 *) (a->d_data_b + (size_t) first * a->fft_len), (const float2 *) a->d_code,
                                   (const float2 *) a->ctx->d_tab4096, (const float2 *) a->d_tabN, (const float2 *) a->d_comb8,
                                   (const float2 *) a->d_quart, a->d_pairs4, a->d_claim + 8 * ACQ_CLAIM_STRIDE, w, a->halo, a->d_cells);
        }
        KG_HIP(hipGetLastError());
    }
    const int npairs = nblocks * nsats;
    hipLaunchKernelGGL(acq_select_kernel, dim3(npairs), dim3(64), 0, st,
                       (const kg_acq_cell *) a->d_cells, npairs, a->dop_lo, a->ndop, a->d_results, a->d_claim);
    KG_HIP(hipGetLastError());
    if (a->own_fstream) {
        KG_HIP(hipEventRecord(a->ev_done[first], st));
        for (int b = first; b < first + nblocks; b++) a->done_of[b] = first;
    }
    a->last_first = first; a->last_nblocks = nblocks; a->last_nsats = nsats;
    return KG_OK;
}

int kg_acq_correlate_async(kg_acq *a, int nblocks, const int *sats, int nsats)
{
    return kg_acq_correlate_blocks_async(a, 0, nblocks, sats, nsats);
}

int kg_acq_fetch(kg_acq *a, kg_acq_result *results, kg_acq_cell *cells)
{
    KG_REQUIRE(a && results, KG_ERR_INVALID, "kg_acq_fetch: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(a->last_nsats > 0, KG_ERR_STATE, "kg_acq_fetch: nothing was launched");
    const size_t np = (size_t) a->last_nblocks * a->last_nsats;
    hipStream_t st = a->ctx->stream;
    KG_HIP(hipMemcpyAsync(results, a->d_results, sizeof(kg_acq_result) * np, hipMemcpyDeviceToHost, st));
    if (cells)
        KG_HIP(hipMemcpyAsync(cells, a->d_cells, sizeof(kg_acq_cell) * np * a->ndop, hipMemcpyDeviceToHost, st));
    KG_HIP(hipStreamSynchronize(st));
    return KG_OK;
}

int kg_acq_correlate(kg_acq *a, int nblocks, const int *sats, int nsats, kg_acq_result *results,
                     kg_acq_cell *cells)
{
    int rc = kg_acq_correlate_async(a, nblocks, sats, nsats);
    if (rc) return rc;
    return kg_acq_fetch(a, results, cells);
}

void *kg_acq_results_dev(kg_acq *a) { return a ? (void *) a->d_results : nullptr; }

int kg_acq_debug_corr_stamps(kg_acq *a, int nblocks, const int *sats, int nsats,
                             unsigned long long *stamps, int n)
{
    KG_REQUIRE(a && stamps && n >= 512 + 4 * 1024, KG_ERR_INVALID, "kg_acq_debug_corr_stamps: need 4608 slots");
    int rc = kg_acq_correlate_async(a, nblocks, sats, nsats);      // builds the lists, warms up
    if (rc) return rc;
    hipStream_t st = a->ctx->stream;
    if (a->np1 == 0 && a->np4 > 0 && a->grid8 > 0) {           // an all-E1B list: the 512-thread kernel's stamps
        unsigned long long *d = nullptr;
        const size_t bytes = sizeof(unsigned long long) * (512 + 4 * 1024);
        KG_HIP(hipMalloc((void **) &d, bytes));
        KG_HIP(hipMemset(d, 0, bytes));
        KG_HIP(hipMemsetAsync(a->d_claim, 0, sizeof(int) * 16 * ACQ_CLAIM_STRIDE, st));
        const acq_walk w = {a->np4, a->ndop, a->dop_lo};
        if (a->P == 4) {
            KG_HIP(hipFuncSetAttribute((const void *) acq_correlate8_kernel<4, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ACQ8_LDS_BYTES));
            hipLaunchKernelGGL((acq_correlate8_kernel<4, true>), dim3(a->grid8), dim3(512), ACQ8_LDS_BYTES, st,
                               (const float2 *) a->d_data_b, (const float2 *) a->d_code, (const float2 *) a->ctx->d_tab4096,
                               (const float2 *) a->d_tabN, (const float2 *) a->d_comb8, (const float2 *) a->d_quart,
                               a->d_pairs4, a->d_claim + 8 * ACQ_CLAIM_STRIDE, w, a->halo, a->d_cells, d);
        } else {
            KG_HIP(hipFuncSetAttribute((const void *) acq_correlate8_kernel<16, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ACQ8_LDS_BYTES));
            hipLaunchKernelGGL((acq_correlate8_kernel<16, true>), dim3(a->grid8), dim3(512), ACQ8_LDS_BYTES, st,
                               (const float2 *) a->d_data_b, (const float2 *) a->d_code, (const float2 *) a->ctx->d_tab4096,
                               (const float2 *) a->d_tabN, (const float2 *) a->d_comb8, (const float2 *) a->d_quart,
                               a->d_pairs4, a->d_claim + 8 * ACQ_CLAIM_STRIDE, w, a->halo, a->d_cells, d);
        }
        KG_HIP(hipGetLastError());
        KG_HIP(hipMemsetAsync(a->d_claim, 0, sizeof(int) * 16 * ACQ_CLAIM_STRIDE, st));
        KG_HIP(hipStreamSynchronize(st));
        KG_HIP(hipMemcpy(stamps, d, bytes, hipMemcpyDeviceToHost));
        KG_HIP(hipFree(d));
        return KG_OK;
    }
    KG_REQUIRE(a->np1 > 0, KG_ERR_STATE, "kg_acq_debug_corr_stamps: no C/A SV in the list");
    unsigned long long *d = nullptr;
    const size_t bytes = sizeof(unsigned long long) * (512 + 4 * 1024);
    KG_REQUIRE(a->grid1 <= 1024, KG_ERR_STATE, "kg_acq_debug_corr_stamps: grid %d", a->grid1);
    KG_HIP(hipMalloc((void **) &d, bytes));
    KG_HIP(hipMemset(d, 0, bytes));
    KG_HIP(hipMemsetAsync(a->d_claim, 0, sizeof(int) * 16 * ACQ_CLAIM_STRIDE, st));
    if (a->P == 4) {
        KG_HIP(hipFuncSetAttribute((const void *) acq_correlate_kernel<4, 1, true, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, ACQ_LDS_BYTES));
        launch_correlate<4, 1, true>(a, st, 0, a->d_pairs1, a->np1, d);
    } else {
        KG_HIP(hipFuncSetAttribute((const void *) acq_correlate_kernel<16, 1, true, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, ACQ_LDS_BYTES));
        launch_correlate<16, 1, true>(a, st, 0, a->d_pairs1, a->np1, d);
    }
    KG_HIP(hipGetLastError());
    KG_HIP(hipMemsetAsync(a->d_claim, 0, sizeof(int) * 16 * ACQ_CLAIM_STRIDE, st));
    KG_HIP(hipStreamSynchronize(st));
    KG_HIP(hipMemcpy(stamps, d, bytes, hipMemcpyDeviceToHost));
    KG_HIP(hipFree(d));
    return KG_OK;
}

int kg_acq_debug_fft_stamps(kg_acq *a, int block, unsigned long long *stamps, int n)
{
    int rc = check_block(a, block, stamps, "kg_acq_debug_fft_stamps");
    if (rc) return rc;
    KG_REQUIRE(n >= 4, KG_ERR_INVALID, "kg_acq_debug_fft_stamps: need room for 4 stamps");
    kg_ctx *c = a->ctx;
    unsigned long long *d = nullptr;
    KG_HIP(hipMalloc((void **) &d, 8 * sizeof(unsigned long long)));
    KG_HIP(hipStreamSynchronize(a->fstream));
    for (int rep = 0; rep < 3; rep++) {       // last repetition is the warm one
        hipLaunchKernelGGL(acq_fft_sub_kernel<true>, dim3(a->P, 1), dim3(256), 2 * SUB * sizeof(float2),
                           c->stream, (const float2 *) (a->d_td + (size_t) block * a->fft_len),
                           a->d_fsub + (size_t) block * a->fft_len, (const float2 *) c->d_tab4096, a->P, d);
        KG_HIP(hipGetLastError());
        KG_HIP(hipStreamSynchronize(c->stream));
    }
    KG_HIP(hipMemcpy(stamps, d, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    KG_HIP(hipFree(d));
    return KG_OK;
}

}  // extern "C"
