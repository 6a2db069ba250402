// kg_snd.hip -- audio front on gfx950: data-pump unpack and the CFastFIR
// overlap-save passband filter.
//
// Replaces
//   snd_service() unpack loop      rx/data_pump.cpp:145-208   (rx_iq_t -> TYPECPX)
//   CFastFIR::SetupParameters      rx/CuteSDR/fastfir.cpp:171-232 (+ SetupCICFilter :148-158)
//   CFastFIR::ProcessData          rx/CuteSDR/fastfir.cpp:241-324
//   CFastFIR::FirPos               rx/CuteSDR/fastfir.h:33
//
// ProcessData is overlap-save with a 1024-point FFT and 512 new samples per
// block: block k transforms the channel's samples [512(k-1), 512(k+1)), multiplies
// by the 1024 frequency-domain coefficients (1/1024 folded in), transforms back
// and keeps the last 512 outputs.  Blocks only depend on input history, so every
// (channel, block) pair is independent: one wave each.
//
// 1024 = 16 * 16 * 4 by one wave (64 lanes x 16 points): two radix-16 passes and
// one radix-4 pass (four butterflies per lane), two LDS exchanges per transform
// in an 8 KiB XOR-swizzled tile that only this wave touches (no workgroup
// barrier).  The positions a lane ends the forward transform with are exactly
// the positions it starts the backward transform with (t + 64 j), so the
// spectrum is multiplied in registers and never goes through LDS.  Only outputs
// 512..1023 are needed: the last radix-4 of the backward transform computes two
// of its four outputs.
#include "kg_common.h"
#include "kg_fft.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define FIR_FFT 1024         // CONV_FFT_SIZE, rx/CuteSDR/cuteSDR.h:12
#define FIR_TAPS 513         // CONV_FIR_SIZE, rx/CuteSDR/fastfir.h:20
#define FIR_OUT 512          // FASTFIR_OUTBUF_SIZE, cuteSDR.h:14

// ---------------------------------------------------------------------------
// data pump unpack: rx_iq_t {u16 i, u16 q, u8 q3, u8 i3} (data_pump.h:27-30),
// sample-major, channel-minor -> out[ch][sample] TYPECPX
// ---------------------------------------------------------------------------
// raw_stride 0: the SPI layout, record (j, ch) at j * nchans + ch; otherwise one row of
// records per channel, raw_stride records apart (what kg_rxddc_push_dev writes)
__global__ void snd_unpack_kernel(const unsigned short *__restrict__ raw, long raw_stride, int nsamps, int nchans,
                                  const unsigned char *__restrict__ enabled, float rescale, float dc_i,
                                  float dc_q, int inversion, float2 *__restrict__ out, long out_stride)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, ch = blockIdx.y;
    if (j >= nsamps || !enabled[ch]) return;
    const unsigned short *p = raw + (raw_stride ? (long) ch * raw_stride + j : (long) j * nchans + ch) * 3;
    const unsigned lo_i = p[0], lo_q = p[1], hi = p[2];        // hi = q3 | i3 << 8
    const unsigned q3 = hi & 0xff, i3 = hi >> 8;
    // S24_8_16(h8, l16), types.h:44
    const int i = (int) ((i3 << 16) | lo_i | ((i3 & 0x80) ? 0xff000000u : 0u));
    const int q = (int) ((q3 << 16) | lo_q | ((q3 & 0x80) ? 0xff000000u : 0u));
    float2 o;
    if (inversion) { o.x = (float) i * rescale + dc_i; o.y = (float) q * rescale + dc_q; }   // :181-182
    else           { o.x = (float) q * rescale + dc_i; o.y = (float) i * rescale + dc_q; }   // :200-201
    out[(long) ch * out_stride + j] = o;
}

// ---------------------------------------------------------------------------
// 1024-point transform by one wave
// ---------------------------------------------------------------------------
struct fir_tw { cf p1[15]; cf p2[4][3]; };

KG_DEV void fir_tw_load(fir_tw &tw, const float2 *__restrict__ tab4096, int t)
{
#pragma unroll
    for (int j = 1; j < 16; j++) tw.p1[j - 1] = kg_ld(&tab4096[(j * (t & 15)) << 4]);      // W_256^(j*(t&15))
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int j = 1; j < 4; j++) tw.p2[u][j - 1] = kg_ld(&tab4096[(j * (t + 64 * u)) << 2]);   // W_1024^(j*b)
}

// wave-private LDS hand-over: within one wave DS operations complete in order, the
// compiler only has to be kept from reordering them
KG_DEV void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// x[j] = X[t + 64 j] in, x[u + 4 m] = Y[t + 64 u + 256 m] out (the same positions).
// LAST2: only m = 2, 3 of the final radix-4 are needed (outputs 512..1023).
template <int SIGN, bool LAST2>
KG_DEV void fir_fft1024(cf (&x)[16], float2 *tile, const fir_tw &tw, int t)
{
    const int tl = t & 15, th = t >> 4;
    cf y[16];
    kg_radix16<SIGN>(x, y);                                            // pass 0: out 16 t + m
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tile[16 * t + (m ^ tl)], y[m]);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 16; j++) {                                     // pass 1 in: t + 64 j
        const int e = t + 64 * j;
        x[j] = kg_ld(&tile[e ^ ((e >> 4) & 15)]);
    }
    wave_lds_fence();
#pragma unroll
    for (int j = 1; j < 16; j++) x[j] = kg_twmul<SIGN>(x[j], tw.p1[j - 1]);
    kg_radix16<SIGN>(x, y);                                            // out (t>>4)*256 + (t&15) + 16 m
#pragma unroll
    for (int m = 0; m < 16; m++) kg_st_tile(&tile[th * 256 + 16 * m + (tl ^ m)], y[m]);
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < 4; u++) {                                      // pass 2: radix 4, b = t + 64 u
        cf z[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = t + 64 * u + 256 * j;
            z[j] = kg_ld(&tile[e ^ ((e >> 4) & 15)]);
        }
#pragma unroll
        for (int j = 1; j < 4; j++) z[j] = kg_twmul<SIGN>(z[j], tw.p2[u][j - 1]);
        if constexpr (!LAST2) {
            kg_radix4<SIGN>(z[0], z[1], z[2], z[3]);
#pragma unroll
            for (int m = 0; m < 4; m++) x[u + 4 * m] = z[m];
        } else {
            const cf s02 = z[0] + z[2], s13 = z[1] + z[3], d02 = z[0] - z[2], d13 = z[1] - z[3];
            x[u + 8] = s02 - s13;                                      // m = 2
            x[u + 12] = kg_sub_sj<SIGN>(d02, d13);                     // m = 3
        }
    }
    wave_lds_fence();                                                  // tile is reused by the next transform
}

#define FIR_WAVES 4                          // (channel, block) pairs per workgroup

// One wave per (list entry, block).  in: per-channel history buffers
// [512 old | new samples...]; block b reads [512 b, 512 b + 1024).
// TAPS: also store the extension taps of fastfir.cpp:278-302 (either pointer may be null):
// pre = forward spectrum x m_CIC, post = the filtered spectrum, 1024 points per block.
template <bool TAPS>
__global__ __launch_bounds__(64 * FIR_WAVES) void fir_block_kernel(
    const float2 *__restrict__ hist, long hist_stride, const int *__restrict__ chan_list,
    const int *__restrict__ nblk, const float2 *__restrict__ coef,     // [nchan][1024]
    const float2 *__restrict__ tab4096, float2 *__restrict__ out, long out_stride, int max_blk,
    const float *__restrict__ cic, const int *__restrict__ cic_on,     // m_CIC = the table where do_CIC_comp, else 1.0 (:156)
    float2 *__restrict__ tap_pre, float2 *__restrict__ tap_post, long tap_stride, int by_chan /* rows of out by channel */)
{
    (void) max_blk;                           // the grid is sized from it; rows check their own nblk
    __shared__ __attribute__((aligned(16))) float2 tiles[FIR_WAVES][FIR_FFT];
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
    const int li = blockIdx.y, blk = blockIdx.x * FIR_WAVES + w;
    if (blk >= nblk[li]) return;                                       // whole wave leaves together
    const int ch = chan_list[li];
    float2 *tile = tiles[w];
    fir_tw tw;
    fir_tw_load(tw, tab4096, t);
    const float2 *src = hist + (long) ch * hist_stride + (long) FIR_OUT * blk;
    const float2 *cf_ = coef + (long) ch * FIR_FFT;
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld(&src[t + 64 * j]);
    fir_fft1024<-1, false>(x, tile, tw, t);                            // fastfir.cpp:274
    if (TAPS && tap_pre) {                                             // simd_multiply_cfc, :280-283
        float2 *p = tap_pre + (long) li * tap_stride + (long) FIR_FFT * blk;
        const bool on = cic_on[ch] != 0;
#pragma unroll
        for (int j = 0; j < 16; j++) { const float c = on ? cic[t + 64 * j] : 1.0f; kg_st(&p[t + 64 * j], cf{x[j].x * c, x[j].y * c}); }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_cmul(x[j], kg_ld(&cf_[t + 64 * j]));    // simd_multiply_ccc, :293
    if (TAPS && tap_post) {                                            // :299-302
        float2 *p = tap_post + (long) li * tap_stride + (long) FIR_FFT * blk;
#pragma unroll
        for (int j = 0; j < 16; j++) kg_st(&p[t + 64 * j], x[j]);
    }
    fir_fft1024<+1, true>(x, tile, tw, t);                             // :304
    float2 *dst = out + (long) (by_chan ? ch : li) * out_stride + (long) FIR_OUT * blk;
#pragma unroll
    for (int u = 0; u < 4; u++) {                                      // keep outputs 512..1023 (:307-310)
        kg_st(&dst[t + 64 * u], x[u + 8]);
        kg_st(&dst[t + 64 * u + 256], x[u + 12]);
    }
}

// The `buf_modified` path of ProcessData (fastfir.cpp:286-290): a PRE_FILTERED spectrum the caller has
// edited is multiplied by m_pFilterCoef (no CIC compensation: the edited buffer already carries it),
// transformed back, samples 512..1023 kept.  One wave per (list entry, block).
__global__ __launch_bounds__(64 * FIR_WAVES) void fir_refilter_kernel(
    const float2 *__restrict__ pre, long tap_stride, const int *__restrict__ chan_list, const int *__restrict__ nblk,
    const float2 *__restrict__ coef0,          // [nchan][1024] m_pFilterCoef
    const float2 *__restrict__ tab4096, float2 *__restrict__ out, long out_stride)
{
    __shared__ __attribute__((aligned(16))) float2 tiles[FIR_WAVES][FIR_FFT];
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
    const int li = blockIdx.y, blk = blockIdx.x * FIR_WAVES + w;
    if (blk >= nblk[li]) return;
    float2 *tile = tiles[w];
    fir_tw tw;
    fir_tw_load(tw, tab4096, t);
    const float2 *src = pre + (long) li * tap_stride + (long) FIR_FFT * blk;
    const float2 *cf_ = coef0 + (long) chan_list[li] * FIR_FFT;
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_cmul(kg_ld(&cf_[t + 64 * j]), kg_ld(&src[t + 64 * j]));   // simd_multiply_ccc, :287-290
    fir_fft1024<+1, true>(x, tile, tw, t);                             // :304
    float2 *dst = out + (long) li * out_stride + (long) FIR_OUT * blk;
#pragma unroll
    for (int u = 0; u < 4; u++) {                                      // outputs 512..1023 (:307-310)
        kg_st(&dst[t + 64 * u], x[u + 8]);
        kg_st(&dst[t + 64 * u + 256], x[u + 12]);
    }
}

// forward transform only: frequency-domain coefficients from the 1024 zero-padded taps
__global__ __launch_bounds__(64) void fir_coef_fft_kernel(const float2 *__restrict__ taps,
                                                         const float *__restrict__ cic,    // may be null
                                                         const float2 *__restrict__ tab4096,
                                                         float2 *__restrict__ coef)
{
    __shared__ __attribute__((aligned(16))) float2 tile[FIR_FFT];
    const int t = threadIdx.x;
    fir_tw tw;
    fir_tw_load(tw, tab4096, t);
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = kg_ld(&taps[t + 64 * j]);
    fir_fft1024<-1, false>(x, tile, tw, t);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        cf v = x[j];
        if (cic) { const float c = cic[t + 64 * j]; v = cf{v.x * c, v.y * c}; }          // SetupCICFilter :153-154
        kg_st(&coef[t + 64 * j], v);
    }
}

// append the new samples of every listed channel (n_each[li] of them, or n where n_each is null) behind its pending ones
__global__ void fir_append_kernel(const float2 *__restrict__ in, long in_stride, const int *__restrict__ chan_list,
                                  const int *__restrict__ fill, int n, const int *__restrict__ n_each, float2 *__restrict__ hist,
                                  long hist_stride, int by_chan /* rows of in by channel */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, li = blockIdx.y, ch = chan_list[li];
    if (i >= (n_each ? n_each[li] : n)) return;
    hist[(long) ch * hist_stride + FIR_OUT + fill[li] + i] = in[(long) (by_chan ? ch : li) * in_stride + i];
}

// after nblk blocks: the last 512 consumed samples plus the leftover move to the front.
// One workgroup per channel; everything is read before anything is written.
__global__ __launch_bounds__(1024) void fir_shift_kernel(float2 *__restrict__ hist, long hist_stride,
                                                        const int *__restrict__ chan_list,
                                                        const int *__restrict__ nblk, const int *__restrict__ rem)
{
    const int li = blockIdx.x, t = threadIdx.x;
    const int k = nblk[li];
    if (k == 0) return;
    float2 *h = hist + (long) chan_list[li] * hist_stride;
    const int cnt = FIR_OUT + rem[li];                                 // <= 1023
    float2 v = make_float2(0.f, 0.f);
    if (t < cnt) v = h[(long) FIR_OUT * k + t];
    __syncthreads();
    if (t < cnt) h[t] = v;
}

// ---------------------------------------------------------------------------
struct kg_fir {
    kg_ctx *ctx;
    int nchan, max_in;
    long hist_stride;
    float2 *d_hist, *d_coef, *d_coef0, *d_taps, *d_stage_in, *d_stage_out;   // d_coef0: m_pFilterCoef (no CIC compensation)
    float *d_cic;
    int *d_cic_on;                            // per channel: m_do_CIC_comp (SetupCICFilter, fastfir.cpp:148-158)
    std::vector<int> fill;                    // pending new samples per channel = FirPos()
    std::vector<char> coef_set;
    std::vector<char> seen;                   // scratch of a call: channels listed so far
    // host copies of the design tables (SetupWindowFunction / constructor)
    float window[FIR_TAPS], cic[FIR_FFT];
    int window_func, cic_3ch;
};

static void fir_window_table(int window_func, float *tbl)     // fastfir.cpp:102-146
{
    const double K_2PI = 2.0 * 3.14159265358979323846;
    if (window_func < 0) window_func = 0;
    const int D = FIR_TAPS - 1;
    for (int i = 0; i < FIR_TAPS; i++) {
        switch (window_func) {
        case 0: tbl[i] = (0.3635819 - 0.4891775 * cosf((K_2PI * i) / D) + 0.1365995 * cosf((2.0 * K_2PI * i) / D)
                          - 0.0106411 * cosf((3.0 * K_2PI * i) / D)); break;
        case 1: tbl[i] = (0.35875 - 0.48829 * cosf((K_2PI * i) / D) + 0.14128 * cosf((2.0 * K_2PI * i) / D)
                          - 0.01168 * cosf((3.0 * K_2PI * i) / D)); break;
        case 2: tbl[i] = (0.355768 - 0.487396 * cosf((K_2PI * i) / D) + 0.144232 * cosf((2.0 * K_2PI * i) / D)
                          - 0.012604 * cosf((3.0 * K_2PI * i) / D)); break;
        case 3: tbl[i] = (0.5 - 0.5 * cosf((K_2PI * i) / D)); break;
        default: tbl[i] = (0.54 - 0.46 * cosf((K_2PI * i) / D)); break;
        }
    }
}

static void fir_cic_table(int snd_rate_3ch, float *cic)       // fastfir.cpp:61-79
{
    const double K_PI = 3.14159265358979323846;
    for (int i = 0; i < FIR_FFT; i++) {
        const float f = fabs(fmod((float) i / FIR_FFT + 0.5f, 1.0f) - 0.5f);
        const float p1 = (snd_rate_3ch ? -3.107f : -2.969f);
        const float p2 = (snd_rate_3ch ? 32.04f : 36.26f);
        const float sincf_ = f ? sinf(f * K_PI) / (f * K_PI) : 1.0f;
        cic[i] = pow(sincf_, -5) + p1 * exp(p2 * (f - 0.5f));
    }
}

extern "C" {

int kg_fir_create(kg_ctx *ctx, int nchan, int max_in, kg_fir **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_fir_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nchan >= 1 && nchan <= 65536, KG_ERR_INVALID, "kg_fir_create: nchan %d", nchan);
    KG_REQUIRE(max_in >= 1 && max_in <= (1 << 22), KG_ERR_INVALID, "kg_fir_create: max_in %d", max_in);
    kg_fir *f = new (std::nothrow) kg_fir();
    KG_REQUIRE(f != nullptr, KG_ERR_NOMEM, "kg_fir_create: alloc");
    f->ctx = ctx; f->nchan = nchan; f->max_in = max_in;
    f->hist_stride = FIR_OUT + (FIR_OUT - 1) + max_in;      // history + leftover + one push
    f->fill.assign(nchan, 0); f->coef_set.assign(nchan, 0);
    f->window_func = -2; f->cic_3ch = -1;
    f->d_stage_in = f->d_stage_out = nullptr;
    KG_HIP(hipMalloc((void **) &f->d_hist, sizeof(float2) * f->hist_stride * nchan));
    KG_HIP(hipMemset(f->d_hist, 0, sizeof(float2) * f->hist_stride * nchan));   // m_pFFTBuf[] = 0, :61-66
    KG_HIP(hipMalloc((void **) &f->d_coef, sizeof(float2) * FIR_FFT * (size_t) nchan));
    KG_HIP(hipMalloc((void **) &f->d_coef0, sizeof(float2) * FIR_FFT * (size_t) nchan));
    KG_HIP(hipMalloc((void **) &f->d_taps, sizeof(float2) * FIR_FFT));
    KG_HIP(hipMalloc((void **) &f->d_cic, sizeof(float) * FIR_FFT));
    KG_HIP(hipMalloc((void **) &f->d_cic_on, sizeof(int) * nchan));
    KG_HIP(hipMemset(f->d_cic_on, 0, sizeof(int) * nchan));
    fir_cic_table(0, f->cic);                                          // the constructor's m_CIC, fastfir.cpp:61-79
    f->cic_3ch = 0;
    KG_HIP(hipMemcpy(f->d_cic, f->cic, sizeof f->cic, hipMemcpyHostToDevice));
    *out = f;
    return KG_OK;
}

void kg_fir_destroy(kg_fir *f)
{
    if (!f) return;
    (void) hipSetDevice(f->ctx->device);
    (void) hipStreamSynchronize(f->ctx->stream);
    (void) hipFree(f->d_hist); (void) hipFree(f->d_coef); (void) hipFree(f->d_coef0); (void) hipFree(f->d_taps); (void) hipFree(f->d_cic); (void) hipFree(f->d_cic_on);
    
    (void) hipFree(f->d_stage_in); (void) hipFree(f->d_stage_out);
    delete f;
}

static int fir_chan_ok(kg_fir *f, int ch, const char *who)
{
    KG_REQUIRE(f != nullptr, KG_ERR_INVALID, "%s: null argument", who);
    KG_REQUIRE(ch >= 0 && ch < f->nchan, KG_ERR_INVALID, "%s: channel %d (0..%d)", who, ch, f->nchan - 1);
    return kg_ctx_use(f->ctx);
}

int kg_fir_set_coef(kg_fir *f, int ch, const float *coef_fft)
{
    int rc = fir_chan_ok(f, ch, "kg_fir_set_coef");
    if (rc) return rc;
    KG_REQUIRE(coef_fft != nullptr, KG_ERR_INVALID, "kg_fir_set_coef: null argument");
    hipStream_t st = f->ctx->stream;
    KG_HIP(hipStreamSynchronize(st));
    KG_HIP(hipMemcpy(f->d_coef + (size_t) ch * FIR_FFT, coef_fft, sizeof(float2) * FIR_FFT, hipMemcpyHostToDevice));
    KG_HIP(hipMemcpy(f->d_coef0 + (size_t) ch * FIR_FFT, coef_fft, sizeof(float2) * FIR_FFT, hipMemcpyHostToDevice));
    const int off = 0;                        // m_CIC[] = 1.0: FlyDog builds with m_do_CIC_comp false (fastfir.cpp:94)
    KG_HIP(hipMemcpy(f->d_cic_on + ch, &off, sizeof off, hipMemcpyHostToDevice));
    f->coef_set[ch] = 1;
    return KG_OK;
}

int kg_fir_set_coef_plain(kg_fir *f, int ch, const float *coef_fft)
{
    int rc = fir_chan_ok(f, ch, "kg_fir_set_coef_plain");
    if (rc) return rc;
    KG_REQUIRE(coef_fft != nullptr, KG_ERR_INVALID, "kg_fir_set_coef_plain: null argument");
    KG_HIP(hipStreamSynchronize(f->ctx->stream));
    KG_HIP(hipMemcpy(f->d_coef0 + (size_t) ch * FIR_FFT, coef_fft, sizeof(float2) * FIR_FFT, hipMemcpyHostToDevice));
    return KG_OK;
}

int kg_fir_get_coef(kg_fir *f, int ch, float *coef_fft)
{
    int rc = fir_chan_ok(f, ch, "kg_fir_get_coef");
    if (rc) return rc;
    KG_REQUIRE(coef_fft != nullptr && f->coef_set[ch], KG_ERR_STATE, "kg_fir_get_coef: no coefficients for channel %d", ch);
    KG_HIP(hipStreamSynchronize(f->ctx->stream));
    KG_HIP(hipMemcpy(coef_fft, f->d_coef + (size_t) ch * FIR_FFT, sizeof(float2) * FIR_FFT, hipMemcpyDeviceToHost));
    return KG_OK;
}

// SetupParameters (fastfir.cpp:171-232).  Returns 1 when the sanity check (:193-200)
// rejects the parameters and the previous coefficients stay, as in the reference.
int kg_fir_setup(kg_fir *f, int ch, float FLoCut, float FHiCut, float Offset, float SampleRate,
                 int window_func, int do_cic_comp, int snd_rate_3ch)
{
    int rc = fir_chan_ok(f, ch, "kg_fir_setup");
    if (rc) return rc;
    const double K_2PI = 2.0 * 3.14159265358979323846, K_PI = 3.14159265358979323846;
    if (window_func < 0) window_func = 0;
    if (f->window_func != window_func) { fir_window_table(window_func, f->window); f->window_func = window_func; }
    hipStream_t st = f->ctx->stream;
    if (f->cic_3ch != (snd_rate_3ch ? 1 : 0)) {
        fir_cic_table(snd_rate_3ch, f->cic);
        f->cic_3ch = snd_rate_3ch ? 1 : 0;
        KG_HIP(hipStreamSynchronize(st));
        KG_HIP(hipMemcpy(f->d_cic, f->cic, sizeof f->cic, hipMemcpyHostToDevice));
    }
    FLoCut += Offset;
    FHiCut += Offset;
    if ((FLoCut >= FHiCut) || (FLoCut >= SampleRate / 2.0) || (FLoCut <= -SampleRate / 2.0) ||
        (FHiCut >= SampleRate / 2.0) || (FHiCut <= -SampleRate / 2.0))
        return 1;
    float nFL = FLoCut / SampleRate;
    float nFH = FHiCut / SampleRate;
    float nFc = (nFH - nFL) / 2.0;
    float nFs = K_2PI * (nFH + nFL) / 2.0;
    float fCenter = 0.5 * (float) (FIR_TAPS - 1);
    std::vector<float2> taps(FIR_FFT, make_float2(0.f, 0.f));
    for (int i = 0; i < FIR_TAPS; i++) {
        float x = (float) i - fCenter;
        float z;
        if ((float) i == fCenter) z = 2.0 * nFc;
        else z = (float) sinf(K_2PI * x * nFc) / (K_PI * x) * f->window[i];
        taps[i].x = z * cosf(nFs * x) / (float) FIR_FFT;
        taps[i].y = z * sinf(nFs * x) / (float) FIR_FFT;
    }
    KG_HIP(hipStreamSynchronize(st));
    KG_HIP(hipMemcpy(f->d_taps, taps.data(), sizeof(float2) * FIR_FFT, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fir_coef_fft_kernel, dim3(1), dim3(64), 0, st, (const float2 *) f->d_taps,
                       do_cic_comp ? (const float *) f->d_cic : (const float *) nullptr,
                       (const float2 *) f->ctx->d_tab4096, f->d_coef + (size_t) ch * FIR_FFT);
    KG_HIP(hipGetLastError());
    hipLaunchKernelGGL(fir_coef_fft_kernel, dim3(1), dim3(64), 0, st, (const float2 *) f->d_taps, (const float *) nullptr,
                       (const float2 *) f->ctx->d_tab4096, f->d_coef0 + (size_t) ch * FIR_FFT);      // m_pFilterCoef, :229
    KG_HIP(hipGetLastError());
    KG_HIP(hipStreamSynchronize(st));
    const int cic_on = do_cic_comp ? 1 : 0;
    KG_HIP(hipMemcpy(f->d_cic_on + ch, &cic_on, sizeof cic_on, hipMemcpyHostToDevice));
    f->coef_set[ch] = 1;
    return KG_OK;
}

int kg_fir_reset(kg_fir *f, int ch)
{
    int rc = fir_chan_ok(f, ch, "kg_fir_reset");
    if (rc) return rc;
    hipStream_t st = f->ctx->stream;
    KG_HIP(hipMemsetAsync(f->d_hist + (size_t) ch * f->hist_stride, 0, sizeof(float2) * f->hist_stride, st));
    f->fill[ch] = 0;
    return KG_OK;
}

int kg_fir_pos(kg_fir *f, int ch)                 // FirPos(), fastfir.h:33
{
    if (!f || ch < 0 || ch >= f->nchan) return KG_ERR_INVALID;
    return f->fill[ch];
}

static int fir_process_impl(kg_fir *f, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int n,
                            void *d_out, size_t out_stride, int32_t *nout, void *d_pre, void *d_post, size_t tap_stride,
                            const int32_t *n_each = nullptr);

int kg_fir_process_dev(kg_fir *f, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int n,
                       void *d_out, size_t out_stride, int32_t *nout)
{
    return fir_process_impl(f, chans, nch, d_in, in_stride, n, d_out, out_stride, nout, nullptr, nullptr, 0);
}

// Every listed channel with its own InLength (n_each[i] >= 0 samples at row i of d_in): the connections of a bank whose audio
// DDCs were started at different times deliver different record counts in one step (rx/rx_sound.cpp:503-601 runs per connection).
int kg_fir_process_each_dev(kg_fir *f, const int32_t *chans, int nch, const void *d_in, size_t in_stride, const int32_t *n_each,
                            void *d_out, size_t out_stride, int32_t *nout)
{
    KG_REQUIRE(n_each != nullptr, KG_ERR_INVALID, "kg_fir_process_each_dev: null argument");
    return fir_process_impl(f, chans, nch, d_in, in_stride, 0, d_out, out_stride, nout, nullptr, nullptr, 0, n_each);
}

int kg_fir_process_taps_dev(kg_fir *f, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int n,
                            void *d_out, size_t out_stride, int32_t *nout, void *d_pre, void *d_post, size_t tap_stride)
{
    return fir_process_impl(f, chans, nch, d_in, in_stride, n, d_out, out_stride, nout, d_pre, d_post, tap_stride);
}

int kg_fir_refilter_dev(kg_fir *f, const int32_t *chans, int nch, const int32_t *nblk, const void *d_pre,
                        size_t tap_stride, void *d_out, size_t out_stride)
{
    KG_REQUIRE(f && chans && nblk && d_pre && d_out, KG_ERR_INVALID, "kg_fir_refilter_dev: null argument");
    int rc = kg_ctx_use(f->ctx);
    if (rc) return rc;
    KG_REQUIRE(nch >= 1 && nch <= f->nchan, KG_ERR_INVALID, "kg_fir_refilter_dev: nch %d", nch);
    int max_blk = 0;
    for (int i = 0; i < nch; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < f->nchan && f->coef_set[chans[i]], KG_ERR_STATE,
                   "kg_fir_refilter_dev: channel %d has no filter", chans[i]);
        KG_REQUIRE(nblk[i] >= 0 && (size_t) nblk[i] * FIR_FFT <= tap_stride && (size_t) nblk[i] * FIR_OUT <= out_stride,
                   KG_ERR_INVALID, "kg_fir_refilter_dev: %d blocks do not fit the strides", nblk[i]);
        if (nblk[i] > max_blk) max_blk = nblk[i];
    }
    if (max_blk == 0) return KG_OK;
    std::vector<int> pack(2 * (size_t) nch);
    memcpy(pack.data(), chans, sizeof(int) * nch);
    memcpy(pack.data() + nch, nblk, sizeof(int) * nch);
    void *base = nullptr;
    if ((rc = kg_ctx_stage(f->ctx, pack.data(), sizeof(int) * pack.size(), &base))) return rc;
    const int *s_list = (const int *) base, *s_nblk = s_list + nch;
    hipLaunchKernelGGL(fir_refilter_kernel, dim3((max_blk + FIR_WAVES - 1) / FIR_WAVES, nch), dim3(64 * FIR_WAVES), 0,
                       f->ctx->stream, (const float2 *) d_pre, (long) tap_stride, s_list, s_nblk,
                       (const float2 *) f->d_coef0, (const float2 *) f->ctx->d_tab4096, (float2 *) d_out, (long) out_stride);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

}  // extern "C"

static int fir_process_impl(kg_fir *f, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int n,
                            void *d_out, size_t out_stride, int32_t *nout, void *d_pre, void *d_post, size_t tap_stride,
                            const int32_t *n_each)
{
    KG_REQUIRE(f && chans && d_in && d_out, KG_ERR_INVALID, "kg_fir_process_dev: null argument");
    int rc = kg_ctx_use(f->ctx);
    if (rc) return rc;
    KG_REQUIRE(nch >= 1 && nch <= f->nchan, KG_ERR_INVALID, "kg_fir_process_dev: nch %d", nch);
    if (n_each) {                                 // every entry its own InLength: n = the largest
        n = 0;
        for (int i = 0; i < nch; i++) {
            KG_REQUIRE(n_each[i] >= 0, KG_ERR_INVALID, "kg_fir_process_each_dev: n[%d] = %d", i, n_each[i]);
            if (n_each[i] > n) n = n_each[i];
        }
    }
    KG_REQUIRE(n >= 0 && n <= f->max_in, KG_ERR_INVALID, "kg_fir_process_dev: n %d (max %d)", n, f->max_in);
    std::vector<int> h_fill(nch), h_nblk(nch), h_rem(nch);
    int max_blk = 0;
    f->seen.assign(f->nchan, 0);                  // (a flag per channel: no quadratic search of the list)
    for (int i = 0; i < nch; i++) {
        const int ch = chans[i];
        KG_REQUIRE(ch >= 0 && ch < f->nchan, KG_ERR_INVALID, "kg_fir_process_dev: channel %d", ch);
        KG_REQUIRE(f->coef_set[ch], KG_ERR_STATE, "kg_fir_process_dev: no filter set for channel %d", ch);
        KG_REQUIRE(!f->seen[ch], KG_ERR_INVALID, "kg_fir_process_dev: channel %d listed twice", ch);
        f->seen[ch] = 1;
        h_fill[i] = f->fill[ch];
        const int tot = f->fill[ch] + (n_each ? n_each[i] : n);
        h_nblk[i] = tot / FIR_OUT;
        h_rem[i] = tot % FIR_OUT;
        KG_REQUIRE((size_t) h_nblk[i] * FIR_OUT <= out_stride || h_nblk[i] == 0, KG_ERR_INVALID,
                   "kg_fir_process_dev: out_stride %zu < %d outputs", out_stride, h_nblk[i] * FIR_OUT);
        KG_REQUIRE(!(d_pre || d_post) || (size_t) h_nblk[i] * FIR_FFT <= tap_stride, KG_ERR_INVALID,
                   "kg_fir_process_taps_dev: tap_stride %zu < %d blocks of 1024", tap_stride, h_nblk[i]);
        if (h_nblk[i] > max_blk) max_blk = h_nblk[i];
        if (nout) nout[i] = h_nblk[i] * FIR_OUT;
    }
    if (n == 0) return KG_OK;
    hipStream_t st = f->ctx->stream;
    // per-call tables through the context's staging ring (no stream synchronisation)
    const int *s_list, *s_fill, *s_nblk, *s_rem, *s_each = nullptr;
    {
        std::vector<int> pack((n_each ? 5 : 4) * (size_t) nch);
        memcpy(pack.data(), chans, sizeof(int) * nch);
        memcpy(pack.data() + nch, h_fill.data(), sizeof(int) * nch);
        memcpy(pack.data() + 2 * nch, h_nblk.data(), sizeof(int) * nch);
        memcpy(pack.data() + 3 * nch, h_rem.data(), sizeof(int) * nch);
        if (n_each) memcpy(pack.data() + 4 * nch, n_each, sizeof(int) * nch);
        void *base = nullptr;
        if ((rc = kg_ctx_stage(f->ctx, pack.data(), sizeof(int) * pack.size(), &base))) return rc;
        s_list = (const int *) base; s_fill = s_list + nch; s_nblk = s_list + 2 * nch; s_rem = s_list + 3 * nch;
        if (n_each) s_each = s_list + 4 * nch;
    }
    const int by_chan = f->ctx->rows_by_chan;
    KG_PLAN_ONLY(f->ctx);
    hipLaunchKernelGGL(fir_append_kernel, dim3((n + 255) / 256, nch), dim3(256), 0, st, (const float2 *) d_in,
                       (long) in_stride, s_list, s_fill, n, s_each, f->d_hist, f->hist_stride, by_chan);
    KG_HIP(hipGetLastError());
    if (max_blk > 0) {
        const dim3 grid((max_blk + FIR_WAVES - 1) / FIR_WAVES, nch);
        if (d_pre || d_post)
            hipLaunchKernelGGL(fir_block_kernel<true>, grid, dim3(64 * FIR_WAVES), 0, st,
                               (const float2 *) f->d_hist, f->hist_stride, s_list, s_nblk,
                               (const float2 *) f->d_coef, (const float2 *) f->ctx->d_tab4096, (float2 *) d_out,
                               (long) out_stride, max_blk, (const float *) f->d_cic, (const int *) f->d_cic_on, (float2 *) d_pre, (float2 *) d_post,
                               (long) tap_stride, by_chan);
        else
            hipLaunchKernelGGL(fir_block_kernel<false>, grid, dim3(64 * FIR_WAVES), 0, st,
                               (const float2 *) f->d_hist, f->hist_stride, s_list, s_nblk,
                               (const float2 *) f->d_coef, (const float2 *) f->ctx->d_tab4096, (float2 *) d_out,
                               (long) out_stride, max_blk, (const float *) nullptr, (const int *) nullptr, (float2 *) nullptr, (float2 *) nullptr, 0L, by_chan);
        KG_HIP(hipGetLastError());
        hipLaunchKernelGGL(fir_shift_kernel, dim3(nch), dim3(1024), 0, st, f->d_hist, f->hist_stride,
                           s_list, s_nblk, s_rem);
        KG_HIP(hipGetLastError());
    }
    for (int i = 0; i < nch; i++) f->fill[chans[i]] = h_rem[i];
    return KG_OK;
}

extern "C" {

// CFastFIR::ProcessData(rx_chan, InLength, In, Out) with host buffers: returns the
// number of samples written to out (0 or a multiple of 512), or a negative status.
int kg_fir_process(kg_fir *f, int ch, const float *in, int n, float *out)
{
    int rc = fir_chan_ok(f, ch, "kg_fir_process");
    if (rc) return rc;
    KG_REQUIRE(in && out, KG_ERR_INVALID, "kg_fir_process: null argument");
    KG_REQUIRE(!f->ctx->rows_by_chan, KG_ERR_STATE, "kg_fir_process: this object belongs to a receiver bank (its rows go by receiver "
               "number): step the bank");
    KG_REQUIRE(n >= 0 && n <= f->max_in, KG_ERR_INVALID, "kg_fir_process: n %d (max %d)", n, f->max_in);
    if (n == 0) return 0;
    hipStream_t st = f->ctx->stream;
    if (!f->d_stage_in) {
        KG_HIP(hipMalloc((void **) &f->d_stage_in, sizeof(float2) * f->max_in));
        KG_HIP(hipMalloc((void **) &f->d_stage_out, sizeof(float2) * (f->max_in + FIR_OUT)));
    }
    KG_HIP(hipMemcpyAsync(f->d_stage_in, in, sizeof(float2) * n, hipMemcpyHostToDevice, st));
    const int32_t c = ch;
    int32_t nout = 0;
    rc = kg_fir_process_dev(f, &c, 1, f->d_stage_in, f->max_in, n, f->d_stage_out, f->max_in + FIR_OUT, &nout);
    if (rc) return rc;
    if (nout > 0) KG_HIP(hipMemcpyAsync(out, f->d_stage_out, sizeof(float2) * nout, hipMemcpyDeviceToHost, st));
    KG_HIP(hipStreamSynchronize(st));
    return nout;
}

// snd_service() unpack (data_pump.cpp:145-208), device buffers.
static int unpack_impl(kg_ctx *ctx, const void *d_raw, size_t raw_stride, int nsamps, int nchans, const uint8_t *enabled,
                       float rescale, float dc_i, float dc_q, int spectral_inversion, void *d_out, size_t out_stride)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(d_raw && d_out && enabled, KG_ERR_INVALID, "kg_dpump_unpack_dev: null argument");
    KG_REQUIRE(nsamps >= 1 && nchans >= 1 && nchans <= 65535, KG_ERR_INVALID, "kg_dpump_unpack_dev: %d samples x %d channels", nsamps, nchans);
    KG_REQUIRE(((uintptr_t) d_raw & 1) == 0 && ((uintptr_t) d_out & 7) == 0, KG_ERR_INVALID, "kg_dpump_unpack_dev: misaligned pointer");
    KG_REQUIRE(out_stride >= (size_t) nsamps && (raw_stride == 0 || raw_stride >= (size_t) nsamps), KG_ERR_INVALID,
               "kg_dpump_unpack_dev: stride smaller than nsamps");
    void *d_en = nullptr;
    if ((rc = kg_ctx_stage(ctx, enabled, nchans, &d_en))) return rc;
    KG_PLAN_ONLY(ctx);
    hipLaunchKernelGGL(snd_unpack_kernel, dim3((nsamps + 255) / 256, nchans), dim3(256), 0, ctx->stream,
                       (const unsigned short *) d_raw, (long) raw_stride, nsamps, nchans, (const unsigned char *) d_en,
                       rescale, dc_i, dc_q, spectral_inversion ? 1 : 0, (float2 *) d_out, (long) out_stride);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_dpump_unpack_dev(kg_ctx *ctx, const void *d_raw, int nsamps, int nchans, const uint8_t *enabled,
                        float rescale, float dc_i, float dc_q, int spectral_inversion, void *d_out,
                        size_t out_stride)
{
    int rc = unpack_impl(ctx, d_raw, 0, nsamps, nchans, enabled, rescale, dc_i, dc_q, spectral_inversion, d_out, out_stride);
    if (rc) return rc;
    KG_HIP(hipStreamSynchronize(ctx->stream));           // documented as synchronous
    return KG_OK;
}

int kg_dpump_unpack_rows_dev(kg_ctx *ctx, const void *d_raw, size_t raw_stride, int nsamps, int nchans,
                             const uint8_t *enabled, float rescale, float dc_i, float dc_q, int spectral_inversion,
                             void *d_out, size_t out_stride)
{
    KG_REQUIRE(raw_stride >= 1, KG_ERR_INVALID, "kg_dpump_unpack_rows_dev: raw_stride 0");
    return unpack_impl(ctx, d_raw, raw_stride, nsamps, nchans, enabled, rescale, dc_i, dc_q, spectral_inversion, d_out,
                       out_stride);
}

}  // extern "C"
