// kg_post.hip -- S-meter, CAgc and the AM / NBFM detectors for many receiver channels.
//
// Reference: rx/rx_sound.cpp:676-696 (S-meter), rx/CuteSDR/agc.cpp (CAgc),
// rx/rx_sound.cpp:766-783 (AM), :845-881 (NBFM).  TYPEREAL is float there and the
// literals are double, so the expressions below keep the reference's operand types
// (the library is built with -ffp-contract=off): the only operations that can differ
// from the CPU path are log10f and powf (device libm vs the host's).
//
// One wavefront per channel.  What the reference does with circular buffers is
// restated so that most of it runs in parallel over the samples of the call:
//   * the signal delay line is a pure delay of m_DelaySamples (agc.cpp:175-180);
//   * m_Peak is exactly the maximum of the last m_WindowSamples magnitudes: the
//     reference keeps the running maximum and rescans the window when the value that
//     leaves equals it (agc.cpp:193-210), and every real magnitude is >= -8, the floor of
//     the rescan; the window maximum is computed by log2(W) doubling passes in LDS;
//   * the two averagers with their data-dependent branches, the hang timer, the S-meter
//     recurrence and the AM DC-removal IIR are sequential: lane 0 walks the samples;
//   * gain (powf), scaling, the mono16 cast and the NBFM detector are parallel again.
#include "kg_common.h"

#include <math.h>
#include <stdlib.h>
#include <new>
#include <vector>

#define POST_CIRC 4096            // per-channel history ring: >= KG_POST_MAX_SAMPLES + 2047
#define POST_MAXW 2047            // MAX_DELAY_BUF - 1 (agc.h:16, agc.cpp:159-160)
#define AGC_OUTSCALE 0.7          // agc.cpp:64
#define MAX_AMPLITUDE 32767.0     // agc.cpp:66

struct post_chan {
    // CAgc parameters (agc.cpp:134-160)
    int agc_on, use_hang, delay_samples, window_samples, hang_time;
    float manual_agc_gain, knee, gain_slope, fixed_gain;
    float attack_rise_alpha, attack_fall_alpha, decay_rise_alpha, decay_fall_alpha;
    // CAgc state
    float decay_ave, attack_ave;
    int hang_timer;
    unsigned count;               // samples written to the rings so far (mod 2^32)
    // S-meter (rx_sound.cpp:249-250)
    float smeter_alpha, smeter_avg, smeter_tap0, smeter_tap1;
    // detectors
    double z1;                    // rx_sound.cpp:244
    float last_re, last_im;       // conn->last_sample
    int mode;
};

// (TYPEMONO16) v
__device__ __forceinline__ short post_mono16(float v)
{
    int w;
    if (!(v > -2147483648.0f && v < 2147483648.0f)) w = (int) 0x80000000u;
    else w = (int) v;
    return (short) (unsigned short) (unsigned) w;
}

__global__ __launch_bounds__(64) void post_kernel(
    post_chan *__restrict__ chan_tab, float2 *__restrict__ ring_in, float *__restrict__ ring_mag,
    const int *__restrict__ chans, const float2 *__restrict__ fir, size_t in_stride, int n,
    short *__restrict__ o_s16, float *__restrict__ o_demod, float2 *__restrict__ o_agc, size_t out_stride)
{
    __shared__ float bufA[POST_MAXW + KG_POST_MAX_SAMPLES];
    __shared__ float bufB[POST_MAXW + KG_POST_MAX_SAMPLES];
    __shared__ float s_db[KG_POST_MAX_SAMPLES];
    __shared__ float2 s_agc[KG_POST_MAX_SAMPLES];
    // one wave per channel walking sequential recursions (S-meter, CAgc): latency, among workgroups that fill the vector
    // units -- it takes the issue priority (beside the DDCs' run passes the kernel stretched from 77 to 450 .. 980 us)
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, row = blockIdx.x, ch = chans[row];
    post_chan *pc = &chan_tab[ch];
    const post_chan c = *pc;
    const float2 *in = fir + (size_t) row * in_stride;
    float2 *rin = ring_in + (size_t) ch * POST_CIRC;
    float *rmag = ring_mag + (size_t) ch * POST_CIRC;
    const unsigned cnt = c.count;
    const int W = c.window_samples, D = c.delay_samples;

    // ---- S-meter, per-sample part (rx_sound.cpp:683-687) ----
    const float snd_max_val = (float) ((1 << (15 - 2)) - 1);
    const float snd_max_pwr = snd_max_val * snd_max_val;
    for (int j = lane; j < n; j += 64) {
        const float2 x = in[j];
        const float pwr = x.x * x.x + x.y * x.y;
        s_db[j] = 10.0 * log10f((pwr / snd_max_pwr) + 1e-30);
    }

    float *P = bufA;                    // window maxima end up here, index W + j
    if (c.agc_on) {
        // ---- magnitudes (agc.cpp:189-191) into the rings and LDS ----
        for (int k = lane; k < W; k += 64) bufA[k] = rmag[(cnt - W + k) & (POST_CIRC - 1)];
        for (int j = lane; j < n; j += 64) {
            const float2 x = in[j];
            float mag = x.x * x.x + x.y * x.y;
            mag = 0.5 * log10f(mag / (MAX_AMPLITUDE * MAX_AMPLITUDE) + 1e-16);
            bufA[W + j] = mag;
            rmag[(cnt + j) & (POST_CIRC - 1)] = mag;
            rin[(cnt + j) & (POST_CIRC - 1)] = x;
        }
        __syncthreads();
        // ---- m_Peak = max of the last W magnitudes (agc.cpp:193-210) ----
        const int L = W + n;
        float *a = bufA, *b = bufB;
        int span = 1;                   // a[i] = max of the `span` entries ending at i
        while (2 * span <= W) {
            for (int i = lane; i < L; i += 64) b[i] = i >= span ? fmaxf(a[i], a[i - span]) : a[i];
            __syncthreads();
            float *t = a; a = b; b = t;
            span *= 2;
        }
        for (int j = lane; j < n; j += 64) b[W + j] = fmaxf(a[W + j], a[W + j - (W - span)]);
        __syncthreads();
        P = b;
    }

    // ---- the sequential part: lane 0 ----
    // One active lane: every instruction costs its full latency, so the loop is written
    // without branches.  Selecting alpha first and then evaluating the reference's
    // expression (1.0 - alpha) * ave + alpha * peak once is the same arithmetic as
    // evaluating it inside the taken branch; (1.0 - alpha) is an exact double per alpha.
    float *magsel = (P == bufA) ? bufB : bufA;
    if (lane == 0) {
        float attack = c.attack_ave, decay = c.decay_ave;
        int hang = c.hang_timer;
        float savg = c.smeter_avg, tap0 = c.smeter_tap0, tap1 = c.smeter_tap1;
        const double s1 = 1.0 - c.smeter_alpha;
        const double ar1 = 1.0 - c.attack_rise_alpha, af1 = 1.0 - c.attack_fall_alpha;
        const double dr1 = 1.0 - c.decay_rise_alpha, df1 = 1.0 - c.decay_fall_alpha;
        const int half = n / 2;
        if (!c.agc_on) {
            for (int j = 0; j < n; j++) {
                savg = s1 * savg + c.smeter_alpha * s_db[j];                           // rx_sound.cpp:688
                tap0 = j == 0 ? savg : tap0;
                tap1 = j == half ? savg : tap1;                                        // :693
            }
        } else {
            const bool use_hang = c.use_hang != 0;
            for (int j = 0; j < n; j++) {
                savg = s1 * savg + c.smeter_alpha * s_db[j];
                tap0 = j == 0 ? savg : tap0;
                tap1 = j == half ? savg : tap1;
                const float peak = P[W + j];
                const bool a_up = peak > attack;                                       // agc.cpp:215-218, 232-235
                const float aa = a_up ? c.attack_rise_alpha : c.attack_fall_alpha;
                attack = (a_up ? ar1 : af1) * attack + aa * peak;
                const bool d_up = peak > decay;                                        // :220-229 / :237-240
                const float da = d_up ? c.decay_rise_alpha : c.decay_fall_alpha;
                const float moved = (d_up ? dr1 : df1) * decay + da * peak;
                const bool hold = use_hang & !d_up & (hang < c.hang_time);             // hang timer running: keep
                decay = hold ? decay : moved;
                hang = use_hang ? (d_up ? 0 : (hold ? hang + 1 : hang)) : hang;
                magsel[j] = attack > decay ? attack : decay;                           // :244-247
            }
            pc->attack_ave = attack; pc->decay_ave = decay; pc->hang_timer = hang;
            pc->count = cnt + (unsigned) n;
        }
        pc->smeter_avg = savg; pc->smeter_tap0 = tap0; pc->smeter_tap1 = tap1;
    }
    __syncthreads();

    // ---- gain and output (agc.cpp:250-253, 259-292) ----
    short *ps16 = o_s16 ? o_s16 + (size_t) row * out_stride : nullptr;
    float2 *pagc = o_agc ? o_agc + (size_t) row * out_stride : nullptr;
    float *pdem = o_demod ? o_demod + (size_t) row * out_stride : nullptr;
    for (int j = lane; j < n; j += 64) {
        float2 y;
        float mono;
        if (c.agc_on) {
            const float mag = magsel[j];
            float gain;
            if (mag <= c.knee) gain = c.fixed_gain;
            else gain = AGC_OUTSCALE * powf(10.0, mag * (c.gain_slope - 1.0));
            // written in this launch for j >= D, by an earlier one otherwise
            const float2 d = j >= D ? in[j - D] : rin[(cnt + j - D) & (POST_CIRC - 1)];
            y.x = d.x * gain; y.y = d.y * gain;
            mono = y.x;
        } else {
            const float2 x = in[j];
            y.x = c.manual_agc_gain * x.x; y.y = c.manual_agc_gain * x.y;
            mono = y.x;
        }
        s_agc[j] = y;
        if (c.mode == KG_POST_AM) {                     // rx_sound.cpp:769-771, off the sequential loop
            const float pwr = y.x * y.x + y.y * y.y;
            s_db[j] = sqrtf(pwr);
        }
        if (c.mode == KG_POST_SSB && ps16) ps16[j] = post_mono16(mono);
        if (pagc && c.mode != KG_POST_SSB) pagc[j] = y;
    }
    __syncthreads();

    if (c.mode == KG_POST_AM) {
        // rx_sound.cpp:773-779: the DC-removal IIR is a recurrence -> lane 0; the envelope
        // is already in s_db (the S-meter is done with it), the differences go out in parallel
        float *s_dm = bufA;                             // free since the gain loop
        if (lane == 0) {
            double z1 = c.z1;
            for (int j = 0; j < n; j++) {
                const float z0 = s_db[j] + (z1 * 0.99f);
                s_dm[j] = z0 - z1;
                z1 = z0;
            }
            pc->z1 = z1;
        }
        __syncthreads();
        if (pdem)
            for (int j = lane; j < n; j += 64) pdem[j] = s_dm[j];
    } else if (c.mode == KG_POST_NBFM) {
        // rx_sound.cpp:845-881
        const float max_val = 32767, clipper_val = 8192;
        for (int j = lane; j < n; j += 64) {
            const float2 y = s_agc[j];
            const float i = y.x, q = y.y;
            const float iL = j ? s_agc[j - 1].x : c.last_re, qL = j ? s_agc[j - 1].y : c.last_im;
            const float pwr = i * i + q * q;
            float out = pwr ? (max_val * 0.340447550238101026565118445432744920253753662109375 *
                               (i * (q - qL) - q * (i - iL)) / pwr) : 0;
            out = out < -clipper_val ? -clipper_val : (out > clipper_val ? clipper_val : out);
            if (pdem) pdem[j] = out;
        }
        if (lane == 0 && n > 0) { pc->last_re = s_agc[n - 1].x; pc->last_im = s_agc[n - 1].y; }
    }
}

__global__ void post_reset_rings_kernel(float2 *ring_in, float *ring_mag, int ch0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, ch = ch0 + blockIdx.y;
    if (i < POST_CIRC) {
        ring_in[(size_t) ch * POST_CIRC + i] = make_float2(0.f, 0.f);      // agc.cpp:119-121
        ring_mag[(size_t) ch * POST_CIRC + i] = -16.0f;                    // :122
    }
}

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
struct post_host {               // the SetParameters() arguments last seen (agc.cpp:101-106)
    int agc_on, use_hang, threshold, manual_gain, decay;
    float slope_factor, sample_rate;
};

struct kg_post {
    kg_ctx *ctx;
    int nchan;
    post_chan *d_chan;
    float2 *d_ring_in;
    float *d_ring_mag;
    std::vector<post_chan> h_chan;       // parameters only; the state lives on the device
    std::vector<post_host> h_args;
    kg_stage_cache list_cache = {};      // the channel list of the last process call
};

static int post_check(kg_post *p, int ch, const char *who)
{
    KG_REQUIRE(p != nullptr, KG_ERR_INVALID, "%s: null object", who);
    KG_REQUIRE(ch >= 0 && ch < p->nchan, KG_ERR_INVALID, "%s: channel %d out of range (0..%d)", who, ch, p->nchan - 1);
    return kg_ctx_use(p->ctx);
}

// Parameter words of post_chan are rewritten from the host copy; the state words are
// patched individually so that a parameter change never rolls the device state back.
template <typename T> static int post_put(kg_post *p, int ch, T post_chan::*field, const T &v)
{
    p->h_chan[ch].*field = v;
    const size_t off = (size_t) ((char *) &(p->h_chan[ch].*field) - (char *) &p->h_chan[ch]);
    KG_HIP(hipMemcpyAsync((char *) (p->d_chan + ch) + off, &(p->h_chan[ch].*field), sizeof(T),
                          hipMemcpyHostToDevice, p->ctx->stream));
    return KG_OK;
}

static int post_reset_agc_state(kg_post *p, int ch)         // agc.cpp:117-131
{
    int rc;
    hipLaunchKernelGGL(post_reset_rings_kernel, dim3(POST_CIRC / 256), dim3(256), 0, p->ctx->stream,
                       p->d_ring_in, p->d_ring_mag, ch);
    KG_HIP(hipGetLastError());
    if ((rc = post_put(p, ch, &post_chan::hang_timer, 0))) return rc;
    if ((rc = post_put(p, ch, &post_chan::decay_ave, -5.0f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::attack_ave, -5.0f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::count, 0u))) return rc;
    return KG_OK;
}

extern "C" {

int kg_post_create(kg_ctx *ctx, int nchan, kg_post **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_post_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nchan >= 1 && nchan <= 65536, KG_ERR_INVALID, "kg_post_create: nchan %d", nchan);
    kg_post *p = new (std::nothrow) kg_post();
    KG_REQUIRE(p != nullptr, KG_ERR_NOMEM, "kg_post_create: alloc");
    p->ctx = ctx; p->nchan = nchan;
    KG_HIP(hipMalloc((void **) &p->d_chan, sizeof(post_chan) * nchan));
    KG_HIP(hipMalloc((void **) &p->d_ring_in, sizeof(float2) * POST_CIRC * (size_t) nchan));
    KG_HIP(hipMalloc((void **) &p->d_ring_mag, sizeof(float) * POST_CIRC * (size_t) nchan));
    post_chan z;
    memset(&z, 0, sizeof z);
    z.agc_on = 1;                                   // CAgc::CAgc(), agc.cpp:77-86
    z.delay_samples = 1; z.window_samples = 1;      // (int)(100.0 * .015), (int)(100.0 * .018)
    z.decay_ave = -5.0f; z.attack_ave = -5.0f;
    z.mode = KG_POST_SSB;
    p->h_chan.assign(nchan, z);
    post_host a = {1, 0, 0, 0, 0, 0.f, 100.0f};
    p->h_args.assign(nchan, a);
    KG_HIP(hipMemcpyAsync(p->d_chan, p->h_chan.data(), sizeof(post_chan) * nchan, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(post_reset_rings_kernel, dim3(POST_CIRC / 256, nchan), dim3(256), 0, ctx->stream,
                       p->d_ring_in, p->d_ring_mag, 0);
    KG_HIP(hipGetLastError());
    KG_HIP(hipStreamSynchronize(ctx->stream));
    *out = p;
    return KG_OK;
}

void kg_post_destroy(kg_post *p)
{
    if (!p) return;
    (void) hipSetDevice(p->ctx->device);
    (void) hipStreamSynchronize(p->ctx->stream);
    (void) hipFree(p->d_chan); (void) hipFree(p->d_ring_in); (void) hipFree(p->d_ring_mag);
    kg_stage_cache_free(&p->list_cache);
    delete p;
}

int kg_post_set_agc(kg_post *p, int ch, int agc_on, int use_hang, int threshold, int manual_gain,
                    int slope_factor, int decay, float sample_rate)
{
    int rc = post_check(p, ch, "kg_post_set_agc");
    if (rc) return rc;
    KG_REQUIRE(sample_rate >= 100.0f && (int) (sample_rate * .018) <= POST_MAXW, KG_ERR_INVALID,
               "kg_post_set_agc: sample rate %g (supported: 100 .. %d Hz)", (double) sample_rate, (int) (POST_MAXW / .018));
    post_host &a = p->h_args[ch];
    agc_on = agc_on != 0; use_hang = use_hang != 0;
    if (agc_on == a.agc_on && use_hang == a.use_hang && threshold == a.threshold && manual_gain == a.manual_gain &&
        slope_factor == a.slope_factor && decay == a.decay && sample_rate == a.sample_rate)
        return KG_OK;                                                       // agc.cpp:101-106
    a.agc_on = agc_on; a.use_hang = use_hang; a.threshold = threshold; a.manual_gain = manual_gain;
    a.slope_factor = slope_factor; a.decay = decay;
    if (a.sample_rate != sample_rate) {                                     // :115-131
        a.sample_rate = sample_rate;
        if ((rc = post_reset_agc_state(p, ch))) return rc;
    }
    const float rate = a.sample_rate;
    // agc.cpp:134-160, operand types as there (TYPEREAL members, double literals)
    const float manual_agc_gain = 32767.0 * powf(10.0, -(100 - (float) a.manual_gain) / 20.0);
    const float knee = (float) a.threshold / 20.0;
    const float gain_slope = a.slope_factor / 100.0;
    const float fixed_gain = AGC_OUTSCALE * powf(10.0, knee * (gain_slope - 1.0));
    const float ara = (1.0 - expf(-1.0 / (rate * .002)));
    const float afa = (1.0 - expf(-1.0 / (rate * .005)));
    const float dra = (1.0 - expf(-1.0 / (rate * (float) a.decay * .001 * .3)));
    const int hang_time = (int) (rate * (float) a.decay * .001);
    float dfa;
    if (a.use_hang) dfa = (1.0 - expf(-1.0 / (rate * .05)));
    else dfa = (1.0 - expf(-1.0 / (rate * (float) a.decay * .001)));
    int delay = (int) (rate * .015);
    const int window = (int) (rate * .018);
    if (delay >= 2048 - 1) delay = 2048 - 1;
    KG_REQUIRE(delay >= 1 && window >= 1, KG_ERR_INVALID, "kg_post_set_agc: delay %d window %d", delay, window);
    if ((rc = post_put(p, ch, &post_chan::agc_on, a.agc_on))) return rc;
    if ((rc = post_put(p, ch, &post_chan::use_hang, a.use_hang))) return rc;
    if ((rc = post_put(p, ch, &post_chan::delay_samples, delay))) return rc;
    if ((rc = post_put(p, ch, &post_chan::window_samples, window))) return rc;
    if ((rc = post_put(p, ch, &post_chan::hang_time, hang_time))) return rc;
    if ((rc = post_put(p, ch, &post_chan::manual_agc_gain, manual_agc_gain))) return rc;
    if ((rc = post_put(p, ch, &post_chan::knee, knee))) return rc;
    if ((rc = post_put(p, ch, &post_chan::gain_slope, gain_slope))) return rc;
    if ((rc = post_put(p, ch, &post_chan::fixed_gain, fixed_gain))) return rc;
    if ((rc = post_put(p, ch, &post_chan::attack_rise_alpha, ara))) return rc;
    if ((rc = post_put(p, ch, &post_chan::attack_fall_alpha, afa))) return rc;
    if ((rc = post_put(p, ch, &post_chan::decay_rise_alpha, dra))) return rc;
    if ((rc = post_put(p, ch, &post_chan::decay_fall_alpha, dfa))) return rc;
    KG_HIP(hipStreamSynchronize(p->ctx->stream));       // the host words just copied may change again
    return KG_OK;
}

int kg_post_agc_delay(kg_post *p, int ch)
{
    int rc = post_check(p, ch, "kg_post_agc_delay");
    if (rc) return rc;
    return p->h_chan[ch].delay_samples;
}

int kg_post_set_smeter(kg_post *p, int ch, float frate)
{
    int rc = post_check(p, ch, "kg_post_set_smeter");
    if (rc) return rc;
    KG_REQUIRE(frate > 0.f, KG_ERR_INVALID, "kg_post_set_smeter: frate %g", (double) frate);
    const float alpha = 1.0 - expf(-1.0 / ((float) frate * .01));          // rx_sound.cpp:248-249
    if ((rc = post_put(p, ch, &post_chan::smeter_alpha, alpha))) return rc;
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    return KG_OK;
}

int kg_post_set_mode(kg_post *p, int ch, int mode)
{
    int rc = post_check(p, ch, "kg_post_set_mode");
    if (rc) return rc;
    KG_REQUIRE(mode >= KG_POST_IQ && mode <= KG_POST_NBFM, KG_ERR_INVALID, "kg_post_set_mode: mode %d", mode);
    if ((rc = post_put(p, ch, &post_chan::mode, mode))) return rc;
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    return KG_OK;
}

int kg_post_reset(kg_post *p, int ch)
{
    int rc = post_check(p, ch, "kg_post_reset");
    if (rc) return rc;
    if ((rc = post_put(p, ch, &post_chan::smeter_avg, 0.f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::smeter_tap0, 0.f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::smeter_tap1, 0.f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::z1, 0.0))) return rc;
    if ((rc = post_put(p, ch, &post_chan::last_re, 0.f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::last_im, 0.f))) return rc;
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    return KG_OK;
}

int kg_post_process_dev(kg_post *p, const int32_t *chans, int nch, const void *d_fir, size_t in_stride,
                        int nsamps, void *d_s16, void *d_demod, void *d_agc, size_t out_stride)
{
    KG_REQUIRE(p && chans && d_fir, KG_ERR_INVALID, "kg_post_process_dev: null argument");
    int rc = kg_ctx_use(p->ctx);
    if (rc) return rc;
    KG_REQUIRE(nch >= 1 && nch <= p->nchan, KG_ERR_INVALID, "kg_post_process_dev: nch %d", nch);
    KG_REQUIRE(nsamps >= 1 && nsamps <= KG_POST_MAX_SAMPLES, KG_ERR_INVALID,
               "kg_post_process_dev: nsamps %d (1..%d)", nsamps, KG_POST_MAX_SAMPLES);
    KG_REQUIRE(in_stride >= (size_t) nsamps && out_stride >= (size_t) nsamps, KG_ERR_INVALID,
               "kg_post_process_dev: stride smaller than nsamps");
    std::vector<char> seen(p->nchan, 0);
    for (int i = 0; i < nch; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < p->nchan && !seen[chans[i]], KG_ERR_INVALID,
                   "kg_post_process_dev: chans[%d] = %d out of range or listed twice", i, chans[i]);
        seen[chans[i]] = 1;
    }
    hipStream_t st = p->ctx->stream;
    void *d_list = nullptr;
    if ((rc = kg_ctx_stage_cached(p->ctx, &p->list_cache, chans, sizeof(int) * nch, &d_list))) return rc;
    KG_PLAN_ONLY(p->ctx);
    hipLaunchKernelGGL(post_kernel, dim3(nch), dim3(64), 0, st, p->d_chan, p->d_ring_in, p->d_ring_mag,
                       (const int *) d_list, (const float2 *) d_fir, in_stride, nsamps,
                       (short *) d_s16, (float *) d_demod, (float2 *) d_agc, out_stride);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_post_smeter(kg_post *p, const int32_t *chans, int nch, float *avg_dB, float *taps)
{
    KG_REQUIRE(p && chans && avg_dB, KG_ERR_INVALID, "kg_post_smeter: null argument");
    int rc = kg_ctx_use(p->ctx);
    if (rc) return rc;
    std::vector<post_chan> h(p->nchan);
    KG_HIP(hipMemcpyAsync(h.data(), p->d_chan, sizeof(post_chan) * p->nchan, hipMemcpyDeviceToHost, p->ctx->stream));
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    for (int i = 0; i < nch; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < p->nchan, KG_ERR_INVALID, "kg_post_smeter: chans[%d] = %d", i, chans[i]);
        avg_dB[i] = h[chans[i]].smeter_avg;
        if (taps) { taps[2 * i] = h[chans[i]].smeter_tap0; taps[2 * i + 1] = h[chans[i]].smeter_tap1; }
    }
    return KG_OK;
}

}  // extern "C"
