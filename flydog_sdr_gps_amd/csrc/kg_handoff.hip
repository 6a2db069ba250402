// kg_handoff.hip -- CHANNEL::Start() arithmetic (host) and aperture_auto() (device).
//
// Reference: gps/channel.cpp:267-311; rx/rx_waterfall.cpp:1173-1273, rx/rx_util.cpp:905-912.
#include "kg_common.h"
#include "kg_libm.h"

#include <math.h>
#include <stdlib.h>
#include <new>
#include <vector>

#define APER_LEN 1024             // APER_PWR_LEN = WF_OUTPUT
#define APER_BINS 256             // 5 dB bands -185 .. 1090 dBm
#define APER_BIN0 (-37)           // band / 5 of the first bin: bands <= -190 are skipped

struct aper_row_cfg { int chan, algo, clear, start, stop; float param; };

__device__ __forceinline__ int aper_wire_to_dBm(int v, int cal)          // rx_util.cpp:905-912, v is a u1_t
{
    return -(255 - v) + cal;
}

__global__ __launch_bounds__(256) void aper_update_kernel(float *__restrict__ avg, const unsigned char *__restrict__ rows,
                                                          size_t row_stride, const aper_row_cfg *__restrict__ cfg, int cal)
{
    const aper_row_cfg c = cfg[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < c.start || i >= c.stop) return;
    float *a = avg + (size_t) c.chan * APER_LEN + i;
    const float pwr = aper_wire_to_dBm(rows[(size_t) blockIdx.y * row_stride + i], cal);
    const float param = c.param;
    float v = *a;
    if (c.clear) {
        v = pwr;                                                         // rx_waterfall.cpp:1184-1185
    } else if (c.algo == KG_APER_IIR) {                                  // :1199-1206
        float iir_gain = 1.0 - kg_libm::expf_glibc((float) (-param * pwr / 255.0));      // the host libm's expf, bit for bit (kg_libm.h)
        if (iir_gain <= 0.01) iir_gain = 0.01;
        v += (pwr - v) * iir_gain;
    } else if (c.algo == KG_APER_MMA) {                                  // :1208-1213
        v = ((v * (param - 1)) + pwr) / param;
    } else {                                                             // EMA :1215-1220
        v += (pwr - v) / param;
    }
    *a = v;
}

// One workgroup per channel: histogram of the 5 dB bands, then the serial scan of the sorted
// band list (:1246-1262) restated on the histogram: ascending bands, a strictly larger count
// wins (the lowest of equally populated bands), the last populated band is the maximum.
__global__ __launch_bounds__(256) void aper_report_kernel(const float *__restrict__ avg, const int *__restrict__ chans,
                                                          const int *__restrict__ audio_fft, int *__restrict__ out)
{
    __shared__ int hist[APER_BINS];
    __shared__ int bad;
    const int t = threadIdx.x, ch = chans[blockIdx.x];
    const int start = audio_fft[blockIdx.x] ? 256 : 0, stop = audio_fft[blockIdx.x] ? 768 : APER_LEN;
    hist[t] = 0;
    if (t == 0) bad = 0;
    __syncthreads();
    for (int i = start + t; i < stop; i += 256) {
        const float qf = floorf(avg[(size_t) ch * APER_LEN + i] / 5);      // b = (int) qf * 5, :1238
        if (!(qf > -38.f)) continue;                                        // b <= -190, :1239
        if (qf >= (float) (APER_BIN0 + APER_BINS)) { bad = 1; continue; }
        atomicAdd(&hist[(int) qf - APER_BIN0], 1);
    }
    __syncthreads();
    if (t == 0) {
        int max_count = 0, max_dBm = -999, min_dBm = 0, len = 0;
        for (int k = 0; k < APER_BINS; k++) {
            const int same = hist[k], last = (k + APER_BIN0) * 5;
            if (!same) continue;
            len += same;
            if (same > max_count) { max_count = same; min_dBm = last; }
            if (last > max_dBm) max_dBm = last;
        }
        if (!len) { max_dBm = -110; min_dBm = -120; }                       // :1264-1265
        if (max_dBm < -80) max_dBm = -80;                                   // :1271
        out[3 * blockIdx.x] = max_dBm; out[3 * blockIdx.x + 1] = min_dBm; out[3 * blockIdx.x + 2] = bad;
    }
}

struct kg_aper {
    kg_ctx *ctx;
    int nchan;
    float *d_avg;
};

extern "C" {

void kg_acq_chan_start(int is_e1b, int lo_shift, int ca_shift, double secs, kg_chan_start *o)
{
    if (!o) return;
    const float BIN_SIZE = 249.755859375;                                   // gps.h:69
    const double FC = 4.092e6, FS = 16.368e6, CPS = 1.023e6, L1_f = 1575.42e6;   // gps.h:42-49
    const int FS_I = 16368000;
    const double lo_dop = lo_shift * BIN_SIZE;                              // channel.cpp:281
    const double ca_dop = (lo_dop / L1_f) * CPS;                            // :282
    const uint32_t lo_rate = (FC + lo_dop) / FS * pow(2, 32);               // :285
    const uint32_t ca_rate = (CPS + ca_dop) / FS * pow(2, 32);              // :286
    const int code_creep = nearbyint((ca_dop * secs / CPS) * FS);           // :296
    const int code_period_ms = is_e1b ? 4 : 1;                              // :299 (E1B_/L1_CODE_PERIOD, gps.h:50,54)
    const int code_period_samples = FS_I / 1000 * code_period_ms;           // :300
    const uint32_t ca_pause = code_period_samples - ((ca_shift + code_creep) % code_period_samples);   // :302
    o->lo_dop = lo_dop; o->ca_dop = ca_dop;
    o->lo_rate = lo_rate; o->ca_rate = ca_rate;
    o->ca_pause = ca_pause; o->code_creep = code_creep;
}

int kg_aper_create(kg_ctx *ctx, int nchan, kg_aper **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_aper_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nchan >= 1 && nchan <= 65536, KG_ERR_INVALID, "kg_aper_create: nchan %d", nchan);
    kg_aper *a = new (std::nothrow) kg_aper();
    KG_REQUIRE(a != nullptr, KG_ERR_NOMEM, "kg_aper_create: alloc");
    a->ctx = ctx; a->nchan = nchan;
    KG_HIP(hipMalloc((void **) &a->d_avg, sizeof(float) * APER_LEN * (size_t) nchan));
    KG_HIP(hipMemset(a->d_avg, 0, sizeof(float) * APER_LEN * (size_t) nchan));
    *out = a;
    return KG_OK;
}

void kg_aper_destroy(kg_aper *a)
{
    if (!a) return;
    (void) hipSetDevice(a->ctx->device);
    (void) hipStreamSynchronize(a->ctx->stream);
    (void) hipFree(a->d_avg);
    delete a;
}

int kg_aper_update_dev(kg_aper *a, const int32_t *chans, int nrows, const void *d_rows, size_t row_stride,
                       const kg_aper_cfg *cfg, int waterfall_cal)
{
    KG_REQUIRE(a && chans && d_rows && cfg, KG_ERR_INVALID, "kg_aper_update_dev: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(nrows >= 1 && nrows <= a->nchan && row_stride >= APER_LEN, KG_ERR_INVALID,
               "kg_aper_update_dev: nrows %d, row_stride %zu", nrows, row_stride);
    std::vector<aper_row_cfg> h(nrows);
    std::vector<char> seen(a->nchan, 0);
    for (int i = 0; i < nrows; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < a->nchan && !seen[chans[i]], KG_ERR_INVALID,
                   "kg_aper_update_dev: chans[%d] = %d out of range or listed twice", i, chans[i]);
        seen[chans[i]] = 1;
        KG_REQUIRE(cfg[i].algo >= KG_APER_IIR && cfg[i].algo <= KG_APER_EMA, KG_ERR_INVALID,
                   "kg_aper_update_dev: cfg[%d].algo = %d", i, cfg[i].algo);
        h[i].chan = chans[i]; h[i].algo = cfg[i].algo; h[i].clear = cfg[i].clear != 0; h[i].param = cfg[i].param;
        h[i].start = cfg[i].audio_fft ? 256 : 0; h[i].stop = cfg[i].audio_fft ? 768 : APER_LEN;
    }
    void *d_cfg = nullptr;
    if ((rc = kg_ctx_stage(a->ctx, h.data(), sizeof(aper_row_cfg) * nrows, &d_cfg))) return rc;
    hipLaunchKernelGGL(aper_update_kernel, dim3(APER_LEN / 256, nrows), dim3(256), 0, a->ctx->stream, a->d_avg,
                       (const unsigned char *) d_rows, row_stride, (const aper_row_cfg *) d_cfg, waterfall_cal);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_aper_report(kg_aper *a, const int32_t *chans, int n, const int32_t *audio_fft, int32_t *signal, int32_t *noise)
{
    KG_REQUIRE(a && chans && audio_fft && signal && noise, KG_ERR_INVALID, "kg_aper_report: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(n >= 1 && n <= a->nchan, KG_ERR_INVALID, "kg_aper_report: n %d", n);
    std::vector<int> h(2 * (size_t) n);
    for (int i = 0; i < n; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < a->nchan, KG_ERR_INVALID, "kg_aper_report: chans[%d] = %d", i, chans[i]);
        h[i] = chans[i]; h[n + i] = audio_fft[i] != 0;
    }
    void *d_in = nullptr;
    if ((rc = kg_ctx_scratch_upload(a->ctx, h.data(), sizeof(int) * h.size(), &d_in))) return rc;
    int *d_out = nullptr;
    KG_HIP(hipMalloc((void **) &d_out, sizeof(int) * 3 * n));
    hipLaunchKernelGGL(aper_report_kernel, dim3(n), dim3(256), 0, a->ctx->stream, (const float *) a->d_avg,
                       (const int *) d_in, (const int *) d_in + n, d_out);
    hipError_t e = hipGetLastError();
    std::vector<int> r(3 * (size_t) n);
    if (e == hipSuccess) e = hipMemcpyAsync(r.data(), d_out, sizeof(int) * 3 * n, hipMemcpyDeviceToHost, a->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(a->ctx->stream);
    (void) hipFree(d_out);
    KG_HIP(e);
    for (int i = 0; i < n; i++) {
        KG_REQUIRE(!r[3 * i + 2], KG_ERR_INVALID, "kg_aper_report: channel %d holds an average above 1000 dBm", chans[i]);
        signal[i] = r[3 * i]; noise[i] = r[3 * i + 1];
    }
    return KG_OK;
}

int kg_aper_get(kg_aper *a, int chan, float *avg_pwr)
{
    KG_REQUIRE(a && avg_pwr, KG_ERR_INVALID, "kg_aper_get: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(chan >= 0 && chan < a->nchan, KG_ERR_INVALID, "kg_aper_get: channel %d", chan);
    KG_HIP(hipMemcpyAsync(avg_pwr, a->d_avg + (size_t) chan * APER_LEN, sizeof(float) * APER_LEN, hipMemcpyDeviceToHost,
                          a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

}  // extern "C"
