// kg_acq.hip -- GPS C/A + Galileo E1B parallel-code-phase acquisition on gfx950.
//
// Replaces gps/search.cpp of the reference: the SearchInit() code-table build
// (:243-285, :309-346), Sample() (:382-449) and Correlate() (:453-499).
//
// Data layout in HBM (private to this file): every 16384-bin spectrum is kept
// "residue-major": plane r (0..3) holds bins k = 4*k1 + r at position k1
// (4 planes x 4096 float2 = 128 KiB).  The unnormalised backward transform of
// Correlate() only needs outputs n < 4092 (C/A) -- a quarter of the 16384 --
// so it is computed as
//      y[n] = sum_{k2=0..3} W_N^{n*k2} * IFFT_4096( X[4*k1 + k2] )[n],  n < 4096
// i.e. four 4096-point transforms (one 256-thread workgroup, 32 KiB of LDS)
// whose results are accumulated in registers; the 16384-point product array
// the reference materialises (rev_buf, :58) never exists.  The Doppler shift
// code[(k - dop) mod N] (:471) is, in this layout, a contiguous rotated read
// of plane (k2 - dop) & 3.  E1B (16368 outputs) keeps four accumulators per
// point, one per output quarter.
#include "kg_common.h"
#include "kg_fft.h"

#include <stdlib.h>
#include <vector>

#define NSAMPLES KG_ACQ_NSAMPLES
#define FFT_LEN  KG_ACQ_FFT_LEN
#define SUB      4096                 // FFT_LEN / 4
#define NTAPS    31                   // gps/search.cpp:49

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
KG_DEV f2 acq_source(const void *__restrict__ src, int i, int nchips, int boc)
{
    if (i >= NSAMPLES) return f2{0.f, 0.f};          // DecimateBy2float zero tail, :145
    if constexpr (SRC == SRC_BITS) {
        // search.cpp:408-423: LSB-first bits, lo_sin = {1,1,0,0}, lo_cos = {1,0,0,1},
        // lo_phase advances by exactly 1.0 per sample.  :168-175: 1 -> -1.0, 0 -> +1.0.
        const uint8_t *p = (const uint8_t *) src;
        const int bit = (p[i >> 3] >> (i & 7)) & 1;
        const int ph = i & 3;
        const int ls = ph < 2, lc = (ph == 0) | (ph == 3);
        return f2{bipolar(bit ^ ls), bipolar(bit ^ lc)};
    } else if constexpr (SRC == SRC_IQ16) {
        const short2 v = ((const short2 *) src)[i];
        const float a = (float) v.x, b = (float) v.y;
        switch (i & 3) {                             // (a + jb) * (-j)^i
        case 0:  return f2{a, b};
        case 1:  return f2{b, -a};
        case 2:  return f2{-a, -b};
        default: return f2{-b, a};
        }
    } else {
        // search.cpp:250-267 / :315-329 with ca_rate = 1/16 exactly: chip index
        // i >> 4, BOC(1,1) half-chip flag = phase >= 0.5  <=>  (i & 15) >= 8.
        const uint8_t *p = (const uint8_t *) src;
        const int chip = p[(i >> 4) % nchips];
        const int b11 = boc ? ((i & 15) >= 8) : 0;
        return f2{bipolar(chip ^ b11), 0.f};
    }
}

// One half-band output, reference accumulation order (search.cpp:148-158):
// c0 term, then j = 2,4,..,30, then the centre tap.  Separate multiply and add.
KG_DEV f2 hb_tap(const f2 *x)
{
    f2 acc = x[0] * kg_splat(c_hb_even[0]);
#pragma unroll
    for (int j = 1; j < 16; j++) acc = acc + x[2 * j] * kg_splat(c_hb_even[j]);
    acc = acc + x[(NTAPS - 1) / 2] * kg_splat(HB_CENTRE);
    return acc;
}

#define FE_TILE 256                          // stage-2 outputs per workgroup
#define FE_NY1  (2 * FE_TILE + NTAPS - 2)    // 541 stage-1 outputs needed
#define FE_NX   (2 * FE_NY1 + NTAPS - 2)     // 1111 input samples needed

template <int SRC>
__global__ __launch_bounds__(256) void acq_frontend_kernel(const uint8_t *__restrict__ src,
                                                          size_t src_stride, int nchips, int boc,
                                                          f2 *__restrict__ td)
{
    __shared__ f2 xs[FE_NX + 1];
    __shared__ f2 y1[FE_NY1 + 1];
    const int tid = threadIdx.x;
    const int o0 = blockIdx.x * FE_TILE;
    const void *s = src + (size_t) blockIdx.y * src_stride;
    for (int u = tid; u < FE_NX; u += 256) xs[u] = acq_source<SRC>(s, 4 * o0 + u, nchips, boc);
    __syncthreads();
    for (int u = tid; u < FE_NY1; u += 256) y1[u] = hb_tap(&xs[2 * u]);
    __syncthreads();
    td[(size_t) blockIdx.y * FFT_LEN + o0 + tid] = hb_tap(&y1[2 * tid]);
}

// ---------------------------------------------------------------------------
// Forward 16384-point FFT, natural time order in -> residue-major spectrum out.
// One 1024-thread workgroup per transform: group g (256 threads) transforms the
// samples n = 4*n1 + g, then a radix-4 step with W_N^{-k*g} combines them.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void acq_fft_fwd_kernel(const f2 *__restrict__ td,
                                                          f2 *__restrict__ planes,
                                                          size_t planes_stride,   // in f2
                                                          const f2 *__restrict__ tab4096,
                                                          const f2 *__restrict__ tab16384)
{
    extern __shared__ __attribute__((aligned(16))) f2 smem[];   // 4 x 4096
    const int tid = threadIdx.x, g = tid >> 8, t = tid & 255;
    const f2 *in = td + (size_t) blockIdx.x * FFT_LEN;
    f2 *out = planes + (size_t) blockIdx.x * planes_stride;
    f2 *tile = smem + g * SUB;

    kg_tw4096 tw;
    kg_tw4096_load(tw, tab4096, t);
    f2 x[16], y[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = in[4 * (t + 256 * j) + g];
    kg_subfft4096<-1>(x, y, tile, tw, t);
    __syncthreads();                                   // all pass-2 reads done
#pragma unroll
    for (int m = 0; m < 16; m++) tile[t + 256 * m] = y[m];     // F_g[k'] at k'
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int k = tid + 1024 * u;                  // k' < 4096
        f2 f0 = smem[k], f1 = smem[SUB + k], f2_ = smem[2 * SUB + k], f3 = smem[3 * SUB + k];
        f1 = kg_cmulc(f1, tab16384[k]);                // W_N^{-k*n2}
        f2_ = kg_cmulc(f2_, tab16384[2 * k]);
        f3 = kg_cmulc(f3, tab16384[3 * k]);
        kg_radix4<-1>(f0, f1, f2_, f3);                // X[k + 4096 q], q = 0..3
        f2 *o = out + (size_t) (k & 3) * SUB + (k >> 2);
        o[0] = f0; o[1024] = f1; o[2048] = f2_; o[3072] = f3;
    }
}

// ---------------------------------------------------------------------------
// Correlate(): one workgroup per (block, SV, Doppler) cell, persistent over a
// cell list.  search.cpp:465-496.
// ---------------------------------------------------------------------------
struct acq_red { float p; int i; float s; };

// exp(+2 pi i k / 64): the wave-uniform part of the combine twiddle (scalar loads)
__constant__ f2 c_w64[64] = {
#define W64(k) {KG_W64[k][0], KG_W64[k][1]}
    W64(0), W64(1), W64(2), W64(3), W64(4), W64(5), W64(6), W64(7),
    W64(8), W64(9), W64(10), W64(11), W64(12), W64(13), W64(14), W64(15),
    W64(16), W64(17), W64(18), W64(19), W64(20), W64(21), W64(22), W64(23),
    W64(24), W64(25), W64(26), W64(27), W64(28), W64(29), W64(30), W64(31),
    W64(32), W64(33), W64(34), W64(35), W64(36), W64(37), W64(38), W64(39),
    W64(40), W64(41), W64(42), W64(43), W64(44), W64(45), W64(46), W64(47),
    W64(48), W64(49), W64(50), W64(51), W64(52), W64(53), W64(54), W64(55),
    W64(56), W64(57), W64(58), W64(59), W64(60), W64(61), W64(62), W64(63),
#undef W64
};

template <int NQ>
__global__ __launch_bounds__(256, 2) void acq_correlate_kernel(
    const f2 *__restrict__ data,      // [nblocks][4][4096]
    const f2 *__restrict__ code,      // [max_sats][4][4096]
    const f2 *__restrict__ tab4096, const f2 *__restrict__ tab16384,
    const int *__restrict__ sats,     // [nsats]  SV id at each list position
    const int *__restrict__ sel,      // [nsel]   list positions this launch handles
    const int *__restrict__ limits,   // [max_sats]
    int nsel, int nsats, int nblocks, int dop_lo, int ndop,
    kg_acq_cell *__restrict__ cells)  // [nblocks][nsats][ndop]
{
    __shared__ __attribute__((aligned(16))) f2 tile[SUB];
    __shared__ acq_red red[4];
    const int t = threadIdx.x;

    kg_tw4096 tw;
    kg_tw4096_load(tw, tab4096, t);

    // XCD-aware cell list: workgroups b and b+8 share an XCD (round-robin
    // dispatch), so (block, SV) pair p is served only by workgroups with
    // b % 8 == p % 8 and its code spectrum stays in one L2.  Speed only.
    const int npairs = nblocks * nsel;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int np_x = xcd < npairs ? (npairs - xcd + 7) >> 3 : 0;

    for (int lc = slot; lc < np_x * ndop; lc += nslots) {
        const int pi = lc / ndop, di = lc - pi * ndop;
        const int p = xcd + 8 * pi;
        const int blk = p / nsel, pos = sel[p - blk * nsel];
        const int sat = sats[pos];
        const int dop = dop_lo + di;
        const int limit = limits[sat];
        const f2 *dsp = data + (size_t) blk * FFT_LEN;
        const f2 *csp = code + (size_t) sat * FFT_LEN;

        f2 acc[NQ][16];
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int m = 0; m < 16; m++) acc[q][m] = f2{0.f, 0.f};
        // rolled on purpose: an unrolled loop lets the compiler hoist all 128
        // global loads to the top (>400 VGPRs) and quadruples the code size
#pragma unroll 1
        for (int k2 = 0; k2 < 4; k2++) {
            const int s = k2 - dop;
            const f2 *dp = dsp + k2 * SUB;
            const f2 *cp = csp + (s & 3) * SUB;
            const int q0 = s >> 2;                     // floor((k2 - dop) / 4)
            f2 x[16], y[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int k1 = t + 256 * j;
                const f2 d = dp[k1];
                const f2 c = cp[(k1 + q0) & (SUB - 1)];
                // conj(d) * c, simd_multiply_conjugate_ccc (support/simd.cpp:39-67)
                x[j] = kg_cmulc(c, d);
            }
            kg_subfft4096<+1>(x, y, tile, tw, t);
            // acc_q[n'] += y[n'] * W_N^{n'*k2} * j^{q*k2},  n' = t + 256 m, and
            // W_N^{n'*k2} = W_N^{t*k2} * W_64^{m*k2} (per-thread base x uniform table)
            const f2 base = tab16384[t * k2];
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const f2 w = kg_cmul(base, c_w64[(m * k2) & 63]);
                const f2 z = kg_cmul(y[m], w);
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const int r = (q * k2) & 3;        // j^(q*k2), wave-uniform
                    f2 zq = (r & 1) ? f2{-z.y, z.x} : z;
                    zq = (r & 2) ? -zq : zq;
                    acc[q][m] = acc[q][m] + zq;
                }
            }
        }

        // search.cpp:486-490: power, first maximum (strict >), running total
        float bp = 0.f, sum = 0.f;
        int bi = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const int n = t + 256 * m + SUB * q;
                const f2 v = acc[q][m];
                const float pw = v.x * v.x + v.y * v.y;
                if (n < limit) {
                    if (pw > bp) { bp = pw; bi = n; }
                    sum += pw;
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float op = __shfl_xor(bp, off);
            const int oi = __shfl_xor(bi, off);
            const float os = __shfl_xor(sum, off);
            if (op > bp || (op == bp && oi < bi)) { bp = op; bi = oi; }
            sum += os;
        }
        if ((t & 63) == 0) { red[t >> 6].p = bp; red[t >> 6].i = bi; red[t >> 6].s = sum; }
        __syncthreads();
        if (t == 0) {
            float mp = red[0].p, tot = red[0].s;
            int mi = red[0].i;
            for (int w = 1; w < 4; w++) {
                if (red[w].p > mp || (red[w].p == mp && red[w].i < mi)) { mp = red[w].p; mi = red[w].i; }
                tot += red[w].s;
            }
            const float ave = tot / (float) limit;     // :493
            kg_acq_cell c;
            c.snr = mp / ave;                          // :494
            c.max_pwr = mp; c.tot_pwr = tot; c.idx = mi;
            cells[((size_t) blk * nsats + pos) * ndop + di] = c;
        }
        // red[] is rewritten only after the next cell's barriers
    }
}

// search.cpp:455,495: best Doppler bin per (block, SV), strict > in ascending dop
__global__ void acq_select_kernel(const kg_acq_cell *__restrict__ cells, int npairs, int dop_lo,
                                  int ndop, kg_acq_result *__restrict__ out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    kg_acq_result r = {0.f, 0, 0, 0};
    float max_snr = 0.f;
    for (int di = 0; di < ndop; di++) {
        const kg_acq_cell c = cells[(size_t) p * ndop + di];
        if (c.snr > max_snr) { max_snr = c.snr; r.dop = dop_lo + di; r.idx = c.idx; r.valid = 1; }
    }
    r.snr = max_snr;
    out[p] = r;
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct kg_acq {
    kg_ctx *ctx;
    int max_sats, dop_lo, dop_hi, ndop, max_blocks;
    f2 *d_code;        // [max_sats][4][4096]
    f2 *d_data;        // [max_blocks][4][4096]
    f2 *d_td;          // [max_blocks][16384]  decimated time-domain samples per block
    f2 *d_td_code;     // [16384]              same, for the code-table build
    uint8_t *d_in;     // [max_blocks][NSAMPLES*4]    host-input staging
    uint8_t *d_chips;  // [E1B_CODELEN max]
    int *d_limits;     // [max_sats]
    int *d_sats, *d_sel1, *d_sel4;
    kg_acq_cell *d_cells;
    kg_acq_result *d_results;
    std::vector<int> limits, code_set;
    std::vector<int> last_sats;
    int last_nblocks, last_nsats, nsel1, nsel4;
    int grid1, grid4;
};

static const size_t IN_STRIDE = (size_t) NSAMPLES * 4;     // bytes per block of staging

static void to_planes(const float *nat, std::vector<float2> &pl)
{
    pl.resize(FFT_LEN);
    for (int k = 0; k < FFT_LEN; k++)
        pl[(size_t) (k & 3) * SUB + (k >> 2)] = make_float2(nat[2 * k], nat[2 * k + 1]);
}
static void from_planes(const std::vector<float2> &pl, float *nat)
{
    for (int k = 0; k < FFT_LEN; k++) {
        const float2 v = pl[(size_t) (k & 3) * SUB + (k >> 2)];
        nat[2 * k] = v.x; nat[2 * k + 1] = v.y;
    }
}

template <int SRC>
static int launch_frontend(kg_acq *a, const uint8_t *d_src, size_t stride, int nbatch, int nchips,
                           int boc, f2 *d_td, f2 *d_planes, size_t planes_stride)
{
    kg_ctx *c = a->ctx;
    hipLaunchKernelGGL(acq_frontend_kernel<SRC>, dim3(FFT_LEN / FE_TILE, nbatch), dim3(256), 0,
                       c->stream, d_src, stride, nchips, boc, d_td);
    KG_HIP(hipGetLastError());
    hipLaunchKernelGGL(acq_fft_fwd_kernel, dim3(nbatch), dim3(1024), 4 * SUB * sizeof(f2),
                       c->stream, (const f2 *) d_td, d_planes, planes_stride,
                       (const f2 *) c->d_tab4096, (const f2 *) c->d_tab16384);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

extern "C" {

int kg_acq_create(kg_ctx *ctx, int max_sats, int dop_lo, int dop_hi, int max_blocks, kg_acq **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_acq_create: out is null");
    *out = nullptr;
    KG_REQUIRE(max_sats >= 1 && max_sats <= 4096, KG_ERR_INVALID, "kg_acq_create: max_sats %d", max_sats);
    KG_REQUIRE(dop_lo <= dop_hi && dop_lo > -FFT_LEN / 2 && dop_hi < FFT_LEN / 2, KG_ERR_INVALID,
               "kg_acq_create: Doppler range %d..%d", dop_lo, dop_hi);
    KG_REQUIRE(max_blocks >= 1 && max_blocks <= 65535, KG_ERR_INVALID, "kg_acq_create: max_blocks %d",
               max_blocks);
    kg_acq *a = new (std::nothrow) kg_acq();
    KG_REQUIRE(a != nullptr, KG_ERR_NOMEM, "kg_acq_create: alloc");
    a->ctx = ctx; a->max_sats = max_sats; a->dop_lo = dop_lo; a->dop_hi = dop_hi;
    a->ndop = dop_hi - dop_lo + 1; a->max_blocks = max_blocks;
    a->limits.assign(max_sats, 0); a->code_set.assign(max_sats, 0);
    a->last_nblocks = a->last_nsats = a->nsel1 = a->nsel4 = 0;
    const size_t spec = sizeof(f2) * FFT_LEN;
    KG_HIP(hipMalloc((void **) &a->d_code, spec * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_data, spec * max_blocks));
    KG_HIP(hipMalloc((void **) &a->d_td, spec * max_blocks));
    KG_HIP(hipMalloc((void **) &a->d_td_code, spec));
    KG_HIP(hipMalloc((void **) &a->d_in, IN_STRIDE * max_blocks));
    KG_HIP(hipMalloc((void **) &a->d_chips, 8192));
    KG_HIP(hipMalloc((void **) &a->d_limits, sizeof(int) * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_sats, sizeof(int) * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_sel1, sizeof(int) * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_sel4, sizeof(int) * max_sats));
    KG_HIP(hipMalloc((void **) &a->d_cells, sizeof(kg_acq_cell) * (size_t) max_blocks * max_sats * a->ndop));
    KG_HIP(hipMalloc((void **) &a->d_results, sizeof(kg_acq_result) * (size_t) max_blocks * max_sats));
    KG_HIP(hipMemset(a->d_limits, 0, sizeof(int) * max_sats));
    KG_HIP(hipMemset(a->d_data, 0, spec * max_blocks));
    KG_HIP(hipFuncSetAttribute((const void *) acq_fft_fwd_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SUB * sizeof(f2)));
    // persistent grid: resident workgroups per CU x CUs, rounded to a multiple of 8 (XCDs)
    int occ1 = 0, occ4 = 0;
    KG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, acq_correlate_kernel<1>, 256, 0));
    KG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ4, acq_correlate_kernel<4>, 256, 0));
    if (occ1 < 1) occ1 = 1;
    if (occ4 < 1) occ4 = 1;
    a->grid1 = (ctx->num_cus * occ1) & ~7;
    a->grid4 = (ctx->num_cus * occ4) & ~7;
    if (a->grid1 < 8) a->grid1 = 8;
    if (a->grid4 < 8) a->grid4 = 8;
    *out = a;
    return KG_OK;
}

void kg_acq_destroy(kg_acq *a)
{
    if (!a) return;
    (void) hipSetDevice(a->ctx->device);
    (void) hipStreamSynchronize(a->ctx->stream);
    (void) hipFree(a->d_code); (void) hipFree(a->d_data); (void) hipFree(a->d_td);
    (void) hipFree(a->d_td_code);
    (void) hipFree(a->d_in); (void) hipFree(a->d_chips); (void) hipFree(a->d_limits);
    (void) hipFree(a->d_sats); (void) hipFree(a->d_sel1); (void) hipFree(a->d_sel4);
    (void) hipFree(a->d_cells); (void) hipFree(a->d_results);
    delete a;
}

static int set_limit(kg_acq *a, int sat, int limit)
{
    KG_REQUIRE(sat >= 0 && sat < a->max_sats, KG_ERR_INVALID, "sat %d out of range (0..%d)", sat,
               a->max_sats - 1);
    KG_REQUIRE(limit >= 1 && limit <= FFT_LEN, KG_ERR_INVALID, "limit %d out of range (1..%d)", limit,
               FFT_LEN);
    a->limits[sat] = limit;
    a->code_set[sat] = 1;
    a->last_sats.clear();        // force the selection lists to be rebuilt
    KG_HIP(hipMemcpyAsync(a->d_limits + sat, &a->limits[sat], sizeof(int), hipMemcpyHostToDevice,
                          a->ctx->stream));
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
    rc = launch_frontend<SRC_CHIPS>(a, a->d_chips, 0, 1, nchips, boc ? 1 : 0, a->d_td_code,
                                    a->d_code + (size_t) sat * FFT_LEN, FFT_LEN);
    if (rc) return rc;
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
    to_planes(code_fft, pl);
    KG_HIP(hipMemcpyAsync(a->d_code + (size_t) sat * FFT_LEN, pl.data(), sizeof(f2) * FFT_LEN,
                          hipMemcpyHostToDevice, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

static int get_planes(kg_acq *a, const f2 *d, float *nat)
{
    std::vector<float2> pl(FFT_LEN);
    KG_HIP(hipMemcpyAsync(pl.data(), d, sizeof(f2) * FFT_LEN, hipMemcpyDeviceToHost, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    from_planes(pl, nat);
    return KG_OK;
}

int kg_acq_get_code_fft(kg_acq *a, int sat, float *code_fft)
{
    KG_REQUIRE(a && code_fft, KG_ERR_INVALID, "kg_acq_get_code_fft: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(sat >= 0 && sat < a->max_sats, KG_ERR_INVALID, "kg_acq_get_code_fft: sat %d", sat);
    KG_REQUIRE(a->code_set[sat], KG_ERR_STATE, "kg_acq_get_code_fft: no code set for sat %d", sat);
    return get_planes(a, a->d_code + (size_t) sat * FFT_LEN, code_fft);
}

static int check_block(kg_acq *a, int block, const void *p, const char *who)
{
    KG_REQUIRE(a && p, KG_ERR_INVALID, "%s: null argument", who);
    KG_REQUIRE(block >= 0 && block < a->max_blocks, KG_ERR_INVALID, "%s: block %d out of range (0..%d)",
               who, block, a->max_blocks - 1);
    return kg_ctx_use(a->ctx);
}

int kg_acq_sample_bits_dev(kg_acq *a, int block, const void *d_packed)
{
    int rc = check_block(a, block, d_packed, "kg_acq_sample_bits_dev");
    if (rc) return rc;
    return launch_frontend<SRC_BITS>(a, (const uint8_t *) d_packed, 0, 1, 0, 0,
                                     a->d_td + (size_t) block * FFT_LEN, a->d_data + (size_t) block * FFT_LEN, FFT_LEN);
}

int kg_acq_sample_bits(kg_acq *a, int block, const uint8_t *packed)
{
    int rc = check_block(a, block, packed, "kg_acq_sample_bits");
    if (rc) return rc;
    uint8_t *stage = a->d_in + IN_STRIDE * block;
    KG_HIP(hipMemcpyAsync(stage, packed, NSAMPLES / 8, hipMemcpyHostToDevice, a->ctx->stream));
    return kg_acq_sample_bits_dev(a, block, stage);
}

int kg_acq_sample_iq16_dev(kg_acq *a, int block, const void *d_iq)
{
    int rc = check_block(a, block, d_iq, "kg_acq_sample_iq16_dev");
    if (rc) return rc;
    KG_REQUIRE(((uintptr_t) d_iq & 3) == 0, KG_ERR_INVALID, "kg_acq_sample_iq16_dev: pointer not 4-byte aligned");
    return launch_frontend<SRC_IQ16>(a, (const uint8_t *) d_iq, 0, 1, 0, 0,
                                     a->d_td + (size_t) block * FFT_LEN, a->d_data + (size_t) block * FFT_LEN, FFT_LEN);
}

int kg_acq_sample_iq16(kg_acq *a, int block, const int16_t *iq)
{
    int rc = check_block(a, block, iq, "kg_acq_sample_iq16");
    if (rc) return rc;
    uint8_t *stage = a->d_in + IN_STRIDE * block;
    KG_HIP(hipMemcpyAsync(stage, iq, (size_t) NSAMPLES * 4, hipMemcpyHostToDevice, a->ctx->stream));
    return kg_acq_sample_iq16_dev(a, block, stage);
}

int kg_acq_set_data_fft(kg_acq *a, int block, const float *data_fft)
{
    int rc = check_block(a, block, data_fft, "kg_acq_set_data_fft");
    if (rc) return rc;
    std::vector<float2> pl;
    to_planes(data_fft, pl);
    KG_HIP(hipMemcpyAsync(a->d_data + (size_t) block * FFT_LEN, pl.data(), sizeof(f2) * FFT_LEN,
                          hipMemcpyHostToDevice, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

int kg_acq_get_data_fft(kg_acq *a, int block, float *data_fft)
{
    int rc = check_block(a, block, data_fft, "kg_acq_get_data_fft");
    if (rc) return rc;
    return get_planes(a, a->d_data + (size_t) block * FFT_LEN, data_fft);
}

int kg_acq_get_data_td(kg_acq *a, int block, float *td)
{
    int rc = check_block(a, block, td, "kg_acq_get_data_td");
    if (rc) return rc;
    KG_HIP(hipMemcpyAsync(td, a->d_td + (size_t) block * FFT_LEN, sizeof(f2) * FFT_LEN, hipMemcpyDeviceToHost, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

int kg_acq_correlate_async(kg_acq *a, int nblocks, const int *sats, int nsats)
{
    KG_REQUIRE(a && sats, KG_ERR_INVALID, "kg_acq_correlate_async: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(nblocks >= 1 && nblocks <= a->max_blocks, KG_ERR_INVALID,
               "kg_acq_correlate_async: nblocks %d (max %d)", nblocks, a->max_blocks);
    KG_REQUIRE(nsats >= 1 && nsats <= a->max_sats, KG_ERR_INVALID,
               "kg_acq_correlate_async: nsats %d (max %d)", nsats, a->max_sats);
    hipStream_t st = a->ctx->stream;
    bool same = (int) a->last_sats.size() == nsats;
    for (int i = 0; same && i < nsats; i++) same = a->last_sats[i] == sats[i];
    if (!same) {
        std::vector<int> s1, s4;
        for (int i = 0; i < nsats; i++) {
            KG_REQUIRE(sats[i] >= 0 && sats[i] < a->max_sats, KG_ERR_INVALID,
                       "kg_acq_correlate_async: sats[%d] = %d out of range", i, sats[i]);
            KG_REQUIRE(a->code_set[sats[i]], KG_ERR_STATE,
                       "kg_acq_correlate_async: no code set for sat %d", sats[i]);
            (a->limits[sats[i]] <= SUB ? s1 : s4).push_back(i);
        }
        // the previous launch may still be reading the lists
        KG_HIP(hipStreamSynchronize(st));
        KG_HIP(hipMemcpy(a->d_sats, sats, sizeof(int) * nsats, hipMemcpyHostToDevice));
        if (!s1.empty()) KG_HIP(hipMemcpy(a->d_sel1, s1.data(), sizeof(int) * s1.size(), hipMemcpyHostToDevice));
        if (!s4.empty()) KG_HIP(hipMemcpy(a->d_sel4, s4.data(), sizeof(int) * s4.size(), hipMemcpyHostToDevice));
        a->nsel1 = (int) s1.size(); a->nsel4 = (int) s4.size();
        a->last_sats.assign(sats, sats + nsats);
    }
    const f2 *t4 = (const f2 *) a->ctx->d_tab4096, *t16 = (const f2 *) a->ctx->d_tab16384;
    if (a->nsel1 > 0) {
        hipLaunchKernelGGL(acq_correlate_kernel<1>, dim3(a->grid1), dim3(256), 0, st,
                           (const f2 *) a->d_data, (const f2 *) a->d_code, t4, t16,
                           (const int *) a->d_sats, (const int *) a->d_sel1, (const int *) a->d_limits,
                           a->nsel1, nsats, nblocks, a->dop_lo, a->ndop, a->d_cells);
        KG_HIP(hipGetLastError());
    }
    if (a->nsel4 > 0) {
        hipLaunchKernelGGL(acq_correlate_kernel<4>, dim3(a->grid4), dim3(256), 0, st,
                           (const f2 *) a->d_data, (const f2 *) a->d_code, t4, t16,
                           (const int *) a->d_sats, (const int *) a->d_sel4, (const int *) a->d_limits,
                           a->nsel4, nsats, nblocks, a->dop_lo, a->ndop, a->d_cells);
        KG_HIP(hipGetLastError());
    }
    const int npairs = nblocks * nsats;
    hipLaunchKernelGGL(acq_select_kernel, dim3((npairs + 63) / 64), dim3(64), 0, st,
                       (const kg_acq_cell *) a->d_cells, npairs, a->dop_lo, a->ndop, a->d_results);
    KG_HIP(hipGetLastError());
    a->last_nblocks = nblocks; a->last_nsats = nsats;
    return KG_OK;
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

}  // extern "C"
