// kg_post.hip -- S-meter, CAgc, the AM / NBFM detectors and what follows them up to out_samps_s2 (m_AM_FIR, the NBFM noise
// squelch, the de-emphasis filters) for many receiver channels.
//
// Reference: rx/rx_sound.cpp:676-696 (S-meter), rx/CuteSDR/agc.cpp (CAgc),
// rx/rx_sound.cpp:766-787 (AM + m_AM_FIR), :845-877 (NBFM + m_Squelch), :898-907 (de-emphasis); rx/CuteSDR/fir.cpp (CFir),
// rx/CuteSDR/squelch.cpp (CSquelch).  TYPEREAL is float there and the
// literals are double, so the expressions below keep the reference's operand types
// (the library is built with -ffp-contract=off), and log10f (which CAgc BRANCHES on) and powf are the
// host libm's algorithms restated on the device (kg_libm.h: bit-identical on every argument, round 6):
// nothing in this file computes differently from the reference built on this image.
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
//   * gain (powf), scaling, the mono16 cast and the NBFM detector are parallel again;
//   * CFir::ProcessFilter keeps its samples in a circular buffer and sums coefficient x sample over the BUFFER positions 0 ..
//     N-1 (fir.cpp:79-91), so the order of the float additions rotates with the write position: sample number g (since the
//     filter was initialised) starts its sum at the tap of age g mod N, runs up to age N-1 and wraps to age 0.  Restated per
//     output sample, that is a sum every lane can do for its own sample -- in that order, over a linear history in LDS;
//   * the squelch's noise average is one more sequential recursion (lane 0); its verdict applies to the whole block.
#include "kg_common.h"
#include "kg_libm.h"

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
    // CSquelch (squelch.cpp:106-107, 122-129) and its state (:67-77)
    float sq_alpha, sq_value, sq_threshold, sq_ave;
    int sq_state, sq_set;         // m_SquelchState, m_SetSquelch
    int sq_rc, squelched;         // the last nsq_nc_sq; s->squelched (rx_sound.cpp:877)
    int deemp, deemp_nfm;         // s->deemp, s->deemp_nfm (rx_sound_cmd.cpp:554)
};

#define POST_MAXTAPS 97           // MAX_NUMCOEF, fir.h:20
#define POST_HIST (POST_MAXTAPS - 1)
enum { POST_FIR_AM = 0, POST_FIR_SQ_HP = 1, POST_FIR_DEEMP_NFM = 2, POST_FIR_DEEMP_AM_SSB = 3, POST_NFIR = 4 };

struct post_cfir {                // one CFir, real-valued (fir.h:24-52)
    int ntaps, pos;               // m_NumTaps; samples since the filter was initialised, mod ntaps (m_State = (ntaps - pos) % ntaps)
    float taps[POST_MAXTAPS];
    float hist[POST_HIST];        // the last 96 inputs, newest last (m_rZBuf unrolled)
};

// (TYPEMONO16) v
__device__ __forceinline__ short post_mono16(float v)
{
    int w;
    if (!(v > -2147483648.0f && v < 2147483648.0f)) w = (int) 0x80000000u;
    else w = (int) v;
    return (short) (unsigned short) (unsigned) w;
}

// CFir::ProcessFilter for the n samples at src (fir.cpp:74-92 real -> real; :176-194 / :199-217 -> mono16 when `mono`).
// X: LDS, POST_HIST + n floats; T: LDS, POST_MAXTAPS floats; src, dst: LDS, dst may be src, neither may overlap X or T.
__device__ __forceinline__ void post_cfir_block(post_cfir *__restrict__ f, float *X, float *T, const float *src, float *dst, int n,
                                                int lane, bool mono)
{
    const int N = f->ntaps, pos = f->pos;
    for (int k = lane; k < POST_HIST; k += 64) X[k] = f->hist[k];
    for (int k = lane; k < N; k += 64) T[k] = f->taps[k];
    for (int j = lane; j < n; j += 64) X[POST_HIST + j] = src[j];
    __syncthreads();
    for (int j = lane; j < n; j += 64) {
        const float *x = X + POST_HIST + j;
        int a = (pos + j) % N;                                  // the age at buffer position 0
        float acc = T[a] * x[-a];                               // "do the 1st MAC"
        for (int t = 1; t < N; t++) {
            a = a + 1 == N ? 0 : a + 1;
            acc += T[a] * x[-a];
        }
        dst[j] = mono ? (float) post_mono16(acc) : acc;
    }
    __syncthreads();
    for (int k = lane; k < POST_HIST; k += 64) f->hist[k] = X[n + k];
    if (lane == 0) f->pos = (pos + n) % N;
    __syncthreads();
}

// CSquelch::PerformFMSquelch (squelch.cpp:151-231) for the n detector samples at demod (LDS; demod + KG_POST_MAX_SAMPLES .. + n is
// scratch): the mono16 output as floats at out (LDS), state and return value into *pc.  X, T as post_cfir_block.
__device__ __forceinline__ void post_squelch_block(post_chan *__restrict__ pc, const post_chan &c, post_cfir *__restrict__ hp, float *X,
                                                   float *T, float *demod, float *out, int n, int lane, int *s_sq)
{
    float *sqbuf = demod + KG_POST_MAX_SAMPLES;
    post_cfir_block(hp, X, T, demod, sqbuf, n, lane, false);                                     // :161
    if (lane == 0) {
        float ave = c.sq_ave;
        const double om = 1.0 - c.sq_alpha;
        for (int i = 0; i < n; i++) {
            const float mag = fabsf(sqbuf[i]);
            ave = om * ave + c.sq_alpha * mag;                                                   // :166
        }
        int state = c.sq_state, rc = 0;
        if (c.sq_value == 0) {                                                                   // :176-179
            if (state) rc = -1;
            state = 0;
        } else if (c.sq_threshold == 0) {                                                        // :182-185
            if (!state) rc = 1;
            state = 1;
        } else if (state) {                                                                      // :188-193
            if (ave < (c.sq_threshold - 50.0)) { rc = -1; state = 0; }
        } else {                                                                                 // :195-200
            if (ave >= (c.sq_threshold + 50.0)) { rc = 1; state = 1; }
        }
        if (c.sq_set) rc = state ? 1 : -1;                                                       // :218-221
        pc->sq_ave = ave; pc->sq_state = state; pc->sq_set = 0; pc->sq_rc = rc;
        if (rc != 0) pc->squelched = rc == 1;                                                    // rx_sound.cpp:877
        *s_sq = state;
    }
    __syncthreads();
    const int squelched = *s_sq;
    for (int j = lane; j < n; j += 64) out[j] = squelched ? 1.0f : (float) post_mono16(demod[j]);   // :205-207, :214-215
    __syncthreads();
}

// The two seams called on their own: m_*_FIR[ch].ProcessFilter(n, in, out) (kind 0: real -> real, 1: real -> mono16, 2: mono16 ->
// mono16) and m_Squelch[ch].PerformFMSquelch(n, in, out) -- the same device functions as the fused pass below.
__global__ __launch_bounds__(64) void post_cfir_kernel(post_cfir *__restrict__ cfir_tab, const int *__restrict__ chans, int slot, int kind,
                                                       const void *__restrict__ in, size_t in_stride, int n, void *__restrict__ out,
                                                       size_t out_stride)
{
    __shared__ float X[POST_HIST + KG_POST_MAX_SAMPLES];
    __shared__ float buf[KG_POST_MAX_SAMPLES];
    __shared__ float T[POST_MAXTAPS];
    const int lane = threadIdx.x, row = blockIdx.x, ch = chans[row];
    for (int j = lane; j < n; j += 64)
        buf[j] = kind == 2 ? (float) ((const short *) in)[(size_t) row * in_stride + j] : ((const float *) in)[(size_t) row * in_stride + j];
    __syncthreads();
    post_cfir_block(cfir_tab + (size_t) ch * POST_NFIR + slot, X, T, buf, buf, n, lane, kind != 0);
    for (int j = lane; j < n; j += 64) {
        if (kind == 0) ((float *) out)[(size_t) row * out_stride + j] = buf[j];
        else ((short *) out)[(size_t) row * out_stride + j] = (short) buf[j];
    }
}

__global__ __launch_bounds__(64) void post_squelch_kernel(post_chan *__restrict__ chan_tab, post_cfir *__restrict__ cfir_tab,
                                                          const int *__restrict__ chans, const float *__restrict__ in, size_t in_stride, int n,
                                                          short *__restrict__ out, size_t out_stride)
{
    __shared__ float X[POST_HIST + KG_POST_MAX_SAMPLES];
    __shared__ float demod[2 * KG_POST_MAX_SAMPLES];
    __shared__ float res[KG_POST_MAX_SAMPLES];
    __shared__ float T[POST_MAXTAPS];
    __shared__ int s_sq;
    const int lane = threadIdx.x, row = blockIdx.x, ch = chans[row];
    post_chan *pc = &chan_tab[ch];
    const post_chan c = *pc;
    for (int j = lane; j < n; j += 64) demod[j] = in[(size_t) row * in_stride + j];
    __syncthreads();
    post_squelch_block(pc, c, cfir_tab + (size_t) ch * POST_NFIR + POST_FIR_SQ_HP, X, T, demod, res, n, lane, &s_sq);
    for (int j = lane; j < n; j += 64) out[(size_t) row * out_stride + j] = (short) res[j];
}

__global__ __launch_bounds__(64) void post_kernel(
    post_chan *__restrict__ chan_tab, post_cfir *__restrict__ cfir_tab, float2 *__restrict__ ring_in, float *__restrict__ ring_mag,
    const int *__restrict__ chans, const float2 *__restrict__ fir, size_t in_stride, int n,
    short *__restrict__ o_s16, float *__restrict__ o_demod, float2 *__restrict__ o_agc, size_t out_stride, int by_chan)
{
    __shared__ float bufA[POST_MAXW + KG_POST_MAX_SAMPLES];
    __shared__ float bufB[POST_MAXW + KG_POST_MAX_SAMPLES];
    __shared__ float s_db[KG_POST_MAX_SAMPLES];
    __shared__ float2 s_agc[KG_POST_MAX_SAMPLES];
    __shared__ float s_taps[POST_MAXTAPS];
    __shared__ int s_sq;
    // one wave per channel walking sequential recursions (S-meter, CAgc): latency, among workgroups that fill the vector
    // units -- it takes the issue priority (beside the DDCs' run passes the kernel stretched from 77 to 450 .. 980 us)
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x, ch = chans[blockIdx.x], row = by_chan ? ch : (int) blockIdx.x;      // kg_ctx::rows_by_chan
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
        s_db[j] = 10.0 * kg_libm::log10f_glibc((float) ((pwr / snd_max_pwr) + 1e-30));      // the host libm's log10f, bit for bit (kg_libm.h)
    }

    float *P = bufA;                    // window maxima end up here, index W + j
    if (c.agc_on) {
        // ---- magnitudes (agc.cpp:189-191) into the rings and LDS ----
        for (int k = lane; k < W; k += 64) bufA[k] = rmag[(cnt - W + k) & (POST_CIRC - 1)];
        for (int j = lane; j < n; j += 64) {
            const float2 x = in[j];
            float mag = x.x * x.x + x.y * x.y;
            mag = 0.5 * kg_libm::log10f_glibc((float) (mag / (MAX_AMPLITUDE * MAX_AMPLITUDE) + 1e-16));
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
            else gain = AGC_OUTSCALE * kg_libm::powf_glibc_pos(10.0f, (float) (mag * (c.gain_slope - 1.0)));    // the host libm's powf (kg_libm.h)
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
        if (c.mode == KG_POST_SSB) {                    // rx_sound.cpp:893
            if (c.deemp) s_db[j] = (float) post_mono16(mono);
            else if (ps16) ps16[j] = post_mono16(mono);
        }
        if (pagc && c.mode != KG_POST_SSB) pagc[j] = y;
    }
    __syncthreads();
    post_cfir *fir4 = cfir_tab + (size_t) ch * POST_NFIR;

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
        // rx_sound.cpp:787: m_AM_FIR.ProcessFilter(ns_out, demod_samps_r, out_samps_s2)
        post_cfir_block(fir4 + POST_FIR_AM, bufB, s_taps, s_dm, s_db, n, lane, true);
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
            bufA[j] = out;
        }
        if (lane == 0 && n > 0) { pc->last_re = s_agc[n - 1].x; pc->last_im = s_agc[n - 1].y; }
        __syncthreads();
        // rx_sound.cpp:876: m_Squelch.PerformFMSquelch(ns_out, demod_samps_r, out_samps_s2)
        post_squelch_block(pc, c, fir4 + POST_FIR_SQ_HP, bufB, s_taps, bufA, s_db, n, lane, &s_sq);
    }
    if (c.mode == KG_POST_IQ) return;
    // rx_sound.cpp:898-907: de-emphasis, out_samps_s2 in place
    const bool nbfm = c.mode == KG_POST_NBFM;
    const bool de_emp = nbfm ? c.deemp_nfm != 0 : c.deemp != 0;
    if (c.mode == KG_POST_SSB && !de_emp) return;               // written by the gain loop
    if (de_emp)
        post_cfir_block(fir4 + (nbfm ? POST_FIR_DEEMP_NFM : POST_FIR_DEEMP_AM_SSB), bufB, s_taps, s_db, s_db, n, lane, true);
    if (ps16)
        for (int j = lane; j < n; j += 64) ps16[j] = (short) s_db[j];
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
    post_cfir *d_cfir;                   // [nchan][POST_NFIR]
    std::vector<post_chan> h_chan;       // parameters only; the state lives on the device
    std::vector<post_host> h_args;
    std::vector<post_cfir> h_cfir;       // taps as designed / handed over; pos and hist are the device's
    std::vector<char> h_fir_ready;       // [nchan][POST_NFIR]: initialised since create
    std::vector<char> h_sq_ready;        // kg_post_squelch_setup AND kg_post_squelch_set were called
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

// ---- CFir designs (fir.cpp:282-384 InitLPFilter, :403-486 InitHPFilter, :538-555 Izero): host arithmetic, operand types as there
namespace cfir_design {
static const double K_2PI = 2.0 * 3.14159265358979323846, K_PI = 3.14159265358979323846;      // datatypes.h:103-104

static float izero(float x)
{
    const float x2 = x / 2.0;
    float sum = 1.0, ds = 1.0, di = 1.0, tmp;
    const float errorlimit = 1e-9;
    do {
        tmp = x2 / di;
        tmp *= tmp;
        ds *= tmp;
        sum += ds;
        di += 1.0;
    } while (ds >= errorlimit * sum);
    return sum;
}

static float beta_of(float Astop)                             // :294-301 = :415-422
{
    if (Astop < 20.96) return 0;
    if (Astop >= 50.0) return .1102 * (Astop - 8.71);
    return .5842 * powf((Astop - 20.96), 0.4) + .07886 * (Astop - 20.96);
}

// (int) of the tap estimate; Fstop == Fpass makes it infinite, whose conversion C leaves undefined: as x86 converts it
static int to_int(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int) v : (int) 0x80000000u; }

static float kaiser(int n, int ntaps, float Beta, float izb, float Scale, float c)             // :327-328 = :451-452
{
    const float x = ((float) n - ((float) ntaps - 1.0) / 2.0) / (((float) ntaps - 1.0) / 2.0);
    return Scale * c * izero(Beta * sqrtf(1 - (x * x))) / izb;
}

static int lowpass(int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate, float *coef)
{
    const float normFpass = Fpass / Fsamprate, normFstop = Fstop / Fsamprate;
    const float normFcut = (normFstop + normFpass) / 2.0;
    const float Beta = beta_of(Astop);
    int ntaps = to_int((Astop - 8.0) / (2.285 * K_2PI * (normFstop - normFpass)) + 1);          // :304
    if (ntaps > POST_MAXTAPS) ntaps = POST_MAXTAPS;
    if (ntaps < 9) ntaps = 9;
    if (NumTaps) ntaps = NumTaps;
    const float fCenter = .5 * (float) (ntaps - 1);
    const float izb = izero(Beta);
    for (int n = 0; n < ntaps; n++) {
        const float x = (float) n - fCenter;
        float c;
        if ((float) n == fCenter) c = 2.0 * normFcut;                                           // :322-323
        else c = (float) sinf(K_2PI * x * normFcut) / (K_PI * x);                               // :325
        coef[n] = kaiser(n, ntaps, Beta, izb, Scale, c);
    }
    return ntaps;
}

static int highpass(int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate, float *coef)
{
    const float normFpass = Fpass / Fsamprate, normFstop = Fstop / Fsamprate;
    const float normFcut = (normFstop + normFpass) / 2.0;
    const float Beta = beta_of(Astop);
    int ntaps = to_int((Astop - 8.0) / (2.285 * K_2PI * (normFpass - normFstop)) + 1);          // :425
    if (ntaps > (POST_MAXTAPS - 1)) ntaps = POST_MAXTAPS - 1;
    if (ntaps < 3) ntaps = 3;
    ntaps |= 1;                                                                                 // :433
    if (NumTaps) ntaps = NumTaps;
    const float izb = izero(Beta);
    const float fCenter = .5 * (float) (ntaps - 1);
    for (int n = 0; n < ntaps; n++) {
        const float x = (float) n - (float) (ntaps - 1) / 2.0;                                  // :442
        float c;
        if ((float) n == fCenter) c = 1.0 - 2.0 * normFcut;                                     // :446
        else c = (float) (sinf(K_PI * x) / (K_PI * x) - sinf(K_2PI * x * normFcut) / (K_PI * x));   // :448
        coef[n] = kaiser(n, ntaps, Beta, izb, Scale, c);
    }
    return ntaps;
}
}  // namespace cfir_design

// A freshly initialised CFir on the device: the taps of h_cfir, zeroed buffer, m_State = 0 (fir.cpp:230-236, :344-350)
static int post_cfir_upload(kg_post *p, int ch, int which)
{
    post_cfir &f = p->h_cfir[(size_t) ch * POST_NFIR + which];
    f.pos = 0;
    memset(f.hist, 0, sizeof f.hist);
    KG_HIP(hipMemcpyAsync(p->d_cfir + (size_t) ch * POST_NFIR + which, &f, sizeof f, hipMemcpyHostToDevice, p->ctx->stream));
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    p->h_fir_ready[(size_t) ch * POST_NFIR + which] = 1;
    return KG_OK;
}

static int post_which(int which, bool init, const char *who)
{
    KG_REQUIRE(which == KG_CFIR_AM || which == KG_CFIR_DEEMP_NFM || which == KG_CFIR_DEEMP_AM_SSB || (!init && which == KG_CFIR_SQUELCH_HP),
               KG_ERR_INVALID, "%s: filter %d (KG_CFIR_AM, KG_CFIR_DEEMP_NFM, KG_CFIR_DEEMP_AM_SSB%s)", who, which,
               init ? "; the squelch's high-pass is designed by kg_post_squelch_setup" : ", KG_CFIR_SQUELCH_HP");
    return KG_OK;
}
static const int POST_WHICH_SLOT[4] = {POST_FIR_AM, POST_FIR_DEEMP_NFM, POST_FIR_DEEMP_AM_SSB, POST_FIR_SQ_HP};

static int post_list(kg_post *p, const int32_t *chans, int nch, const char *who, void **d_list)
{
    KG_REQUIRE(nch >= 1 && nch <= p->nchan, KG_ERR_INVALID, "%s: nch %d", who, nch);
    std::vector<char> seen(p->nchan, 0);
    for (int i = 0; i < nch; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < p->nchan && !seen[chans[i]], KG_ERR_INVALID,
                   "%s: chans[%d] = %d out of range or listed twice", who, i, chans[i]);
        seen[chans[i]] = 1;
    }
    return kg_ctx_stage_cached(p->ctx, &p->list_cache, chans, sizeof(int) * nch, d_list);
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
    KG_HIP(hipMalloc((void **) &p->d_cfir, sizeof(post_cfir) * POST_NFIR * (size_t) nchan));
    post_cfir f0;
    memset(&f0, 0, sizeof f0);
    f0.ntaps = 1;                                   // CFir::CFir(), fir.cpp:60-64 (its coefficient is indeterminate there: 0 here;
    p->h_cfir.assign((size_t) nchan * POST_NFIR, f0);      // a mode that needs an uninitialised filter is refused, kg_post_process_dev)
    p->h_fir_ready.assign((size_t) nchan * POST_NFIR, 0);
    p->h_sq_ready.assign(nchan, 0);
    KG_HIP(hipMemcpyAsync(p->d_cfir, p->h_cfir.data(), sizeof(post_cfir) * p->h_cfir.size(), hipMemcpyHostToDevice, ctx->stream));
    post_chan z;
    memset(&z, 0, sizeof z);
    z.agc_on = 1;                                   // CAgc::CAgc(), agc.cpp:77-86
    z.delay_samples = 1; z.window_samples = 1;      // (int)(100.0 * .015), (int)(100.0 * .018)
    z.decay_ave = -5.0f; z.attack_ave = -5.0f;
    z.mode = KG_POST_SSB;
    z.sq_state = 1;                                 // CSquelch::Reset(), squelch.cpp:67-77
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
    (void) hipFree(p->d_chan); (void) hipFree(p->d_ring_in); (void) hipFree(p->d_ring_mag); (void) hipFree(p->d_cfir);
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

// ---- the three libm functions the device code calls, over an array: what tests/test_libm_gpu.py compares with the image's libm
__global__ void math_kernel(int fn, float base, const float *__restrict__ x, unsigned first, size_t n, float *__restrict__ y)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const float v = x ? x[i] : __uint_as_float(first + (unsigned) i);
        y[i] = fn == KG_MATH_LOG10F ? kg_libm::log10f_glibc(v) : fn == KG_MATH_POWF ? kg_libm::powf_glibc_pos(base, v) : kg_libm::expf_glibc(v);
    }
}

int kg_math_dev(kg_ctx *ctx, int fn, float base, const void *d_x, uint32_t first_bits, size_t n, void *d_y)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(d_y && n >= 1 && ((uintptr_t) d_y & 3) == 0 && ((uintptr_t) d_x & 3) == 0, KG_ERR_INVALID, "kg_math_dev: bad argument");
    KG_REQUIRE(fn == KG_MATH_LOG10F || fn == KG_MATH_POWF || fn == KG_MATH_EXPF, KG_ERR_INVALID, "kg_math_dev: unknown function");
    KG_REQUIRE(fn != KG_MATH_POWF || (base >= 1.17549435e-38f && base < __builtin_huge_valf()), KG_ERR_INVALID,
               "kg_math_dev: powf's base must be positive, finite and normal (CAgc's is 10)");
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(math_kernel, dim3((unsigned) (blocks < 16384 ? blocks : 16384)), dim3(256), 0, ctx->stream, fn, base,
                       (const float *) d_x, first_bits, n, (float *) d_y);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_post_get_mode(kg_post *p, int ch)
{
    int rc = post_check(p, ch, "kg_post_get_mode");
    if (rc) return rc;
    return p->h_chan[ch].mode;
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
        const int ch = chans[i], mode = p->h_chan[ch].mode;
        const char *fr = &p->h_fir_ready[(size_t) ch * POST_NFIR];
        KG_REQUIRE(mode != KG_POST_AM || fr[POST_FIR_AM], KG_ERR_STATE,
                   "kg_post_process_dev: channel %d is in AM mode and its m_AM_FIR was never designed (kg_post_set_am_passband)", ch);
        KG_REQUIRE(mode != KG_POST_NBFM || p->h_sq_ready[ch], KG_ERR_STATE,
                   "kg_post_process_dev: channel %d is in NBFM mode without kg_post_squelch_setup + kg_post_squelch_set (rx_sound.cpp:261-262)", ch);
        KG_REQUIRE(!(mode == KG_POST_NBFM && p->h_chan[ch].deemp_nfm) || fr[POST_FIR_DEEMP_NFM], KG_ERR_STATE,
                   "kg_post_process_dev: channel %d has NBFM de-emphasis on and no m_nfm_deemp_FIR coefficients", ch);
        KG_REQUIRE(!((mode == KG_POST_AM || mode == KG_POST_SSB) && p->h_chan[ch].deemp) || fr[POST_FIR_DEEMP_AM_SSB], KG_ERR_STATE,
                   "kg_post_process_dev: channel %d has AM/SSB de-emphasis on and no m_am_ssb_deemp_FIR coefficients", ch);
    }
    hipStream_t st = p->ctx->stream;
    void *d_list = nullptr;
    if ((rc = kg_ctx_stage_cached(p->ctx, &p->list_cache, chans, sizeof(int) * nch, &d_list))) return rc;
    KG_PLAN_ONLY(p->ctx);
    hipLaunchKernelGGL(post_kernel, dim3(nch), dim3(64), 0, st, p->d_chan, p->d_cfir, p->d_ring_in, p->d_ring_mag,
                       (const int *) d_list, (const float2 *) d_fir, in_stride, nsamps,
                       (short *) d_s16, (float *) d_demod, (float2 *) d_agc, out_stride, p->ctx->rows_by_chan);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_post_cfir_init_lp(kg_post *p, int ch, int which, int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate)
{
    int rc = post_check(p, ch, "kg_post_cfir_init_lp");
    if (rc || (rc = post_which(which, true, "kg_post_cfir_init_lp"))) return rc;
    KG_REQUIRE(NumTaps >= 0 && NumTaps <= POST_MAXTAPS && Fsamprate > 0.f, KG_ERR_INVALID,
               "kg_post_cfir_init_lp: NumTaps %d (0..%d), sample rate %g", NumTaps, POST_MAXTAPS, (double) Fsamprate);
    post_cfir &f = p->h_cfir[(size_t) ch * POST_NFIR + POST_WHICH_SLOT[which]];
    f.ntaps = cfir_design::lowpass(NumTaps, Scale, Astop, Fpass, Fstop, Fsamprate, f.taps);
    if ((rc = post_cfir_upload(p, ch, POST_WHICH_SLOT[which]))) return rc;
    return f.ntaps;
}

int kg_post_cfir_init_const(kg_post *p, int ch, int which, int NumTaps, const float *coef, float Fsamprate)
{
    int rc = post_check(p, ch, "kg_post_cfir_init_const");
    if (rc || (rc = post_which(which, true, "kg_post_cfir_init_const"))) return rc;
    KG_REQUIRE(coef != nullptr && NumTaps >= 1, KG_ERR_INVALID, "kg_post_cfir_init_const: %d coefficients at %p", NumTaps, (const void *) coef);
    (void) Fsamprate;                                           // m_SampleRate is only used by GenerateHBFilter
    post_cfir &f = p->h_cfir[(size_t) ch * POST_NFIR + POST_WHICH_SLOT[which]];
    f.ntaps = NumTaps > POST_MAXTAPS ? POST_MAXTAPS : NumTaps;  // fir.cpp:223-226
    memcpy(f.taps, coef, sizeof(float) * (size_t) f.ntaps);
    if ((rc = post_cfir_upload(p, ch, POST_WHICH_SLOT[which]))) return rc;
    return f.ntaps;
}

int kg_post_cfir_get_taps(kg_post *p, int ch, int which, float *taps)
{
    int rc = post_check(p, ch, "kg_post_cfir_get_taps");
    if (rc || (rc = post_which(which, false, "kg_post_cfir_get_taps"))) return rc;
    const post_cfir &f = p->h_cfir[(size_t) ch * POST_NFIR + POST_WHICH_SLOT[which]];
    if (taps) memcpy(taps, f.taps, sizeof(float) * (size_t) f.ntaps);
    return f.ntaps;
}

int kg_post_cfir_process_dev(kg_post *p, const int32_t *chans, int nch, int which, int kind, const void *d_in, size_t in_stride, int nsamps,
                             void *d_out, size_t out_stride)
{
    KG_REQUIRE(p && chans && d_in && d_out, KG_ERR_INVALID, "kg_post_cfir_process_dev: null argument");
    int rc = kg_ctx_use(p->ctx);
    if (rc || (rc = post_which(which, false, "kg_post_cfir_process_dev"))) return rc;
    KG_REQUIRE(kind >= KG_CFIR_REAL_REAL && kind <= KG_CFIR_MONO16_MONO16, KG_ERR_INVALID, "kg_post_cfir_process_dev: kind %d", kind);
    KG_REQUIRE(nsamps >= 1 && nsamps <= KG_POST_MAX_SAMPLES && in_stride >= (size_t) nsamps && out_stride >= (size_t) nsamps, KG_ERR_INVALID,
               "kg_post_cfir_process_dev: nsamps %d (1..%d), strides %zu / %zu", nsamps, KG_POST_MAX_SAMPLES, in_stride, out_stride);
    void *d_list = nullptr;
    if ((rc = post_list(p, chans, nch, "kg_post_cfir_process_dev", &d_list))) return rc;
    for (int i = 0; i < nch; i++)
        KG_REQUIRE(p->h_fir_ready[(size_t) chans[i] * POST_NFIR + POST_WHICH_SLOT[which]], KG_ERR_STATE,
                   "kg_post_cfir_process_dev: filter %d of channel %d was never initialised", which, chans[i]);
    KG_PLAN_ONLY(p->ctx);
    hipLaunchKernelGGL(post_cfir_kernel, dim3(nch), dim3(64), 0, p->ctx->stream, p->d_cfir, (const int *) d_list, POST_WHICH_SLOT[which], kind,
                       d_in, in_stride, nsamps, d_out, out_stride);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_post_squelch_perform_dev(kg_post *p, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int nsamps, void *d_out,
                                size_t out_stride)
{
    KG_REQUIRE(p && chans && d_in && d_out, KG_ERR_INVALID, "kg_post_squelch_perform_dev: null argument");
    int rc = kg_ctx_use(p->ctx);
    if (rc) return rc;
    KG_REQUIRE(nsamps >= 1 && nsamps <= KG_POST_MAX_SAMPLES && in_stride >= (size_t) nsamps && out_stride >= (size_t) nsamps, KG_ERR_INVALID,
               "kg_post_squelch_perform_dev: nsamps %d (1..%d; squelch.cpp:155 returns at once past 1024), strides %zu / %zu", nsamps,
               KG_POST_MAX_SAMPLES, in_stride, out_stride);
    void *d_list = nullptr;
    if ((rc = post_list(p, chans, nch, "kg_post_squelch_perform_dev", &d_list))) return rc;
    for (int i = 0; i < nch; i++)
        KG_REQUIRE(p->h_sq_ready[chans[i]], KG_ERR_STATE, "kg_post_squelch_perform_dev: channel %d: kg_post_squelch_setup + kg_post_squelch_set first",
                   chans[i]);
    KG_PLAN_ONLY(p->ctx);
    hipLaunchKernelGGL(post_squelch_kernel, dim3(nch), dim3(64), 0, p->ctx->stream, p->d_chan, p->d_cfir, (const int *) d_list,
                       (const float *) d_in, in_stride, nsamps, (short *) d_out, out_stride);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_post_set_am_passband(kg_post *p, int ch, double locut, double hicut, double frate)
{
    // rx_sound_cmd.cpp:248-250: the handler first clamps the client's cuts to the Nyquist limit less one (idempotent for a caller that
    // passes the clamped s->locut / s->hicut); :268-282: "hbw for post AM det is max of hi/lo filter cuts"
    const int fmax = frate / 2 - 1;
    if (hicut > fmax) hicut = fmax;
    if (locut < -fmax) locut = -fmax;
    float hbw = fmaxf(fabs(hicut), fabs(locut));
    if (hbw > frate / 2) hbw = frate / 2;
    float stop = hbw * 1.8;
    if (stop > frate / 2) stop = frate / 2;
    return kg_post_cfir_init_lp(p, ch, KG_CFIR_AM, 0, 1.0, 50.0, hbw, stop, frate);
}

int kg_post_set_deemp(kg_post *p, int ch, int nfm, int de_emp)
{
    int rc = post_check(p, ch, "kg_post_set_deemp");
    if (rc) return rc;
    if ((rc = post_put(p, ch, nfm ? &post_chan::deemp_nfm : &post_chan::deemp, de_emp))) return rc;     // rx_sound_cmd.cpp:554
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    return KG_OK;
}

int kg_post_squelch_setup(kg_post *p, int ch, float samplerate)
{
    int rc = post_check(p, ch, "kg_post_squelch_setup");
    if (rc) return rc;
    KG_REQUIRE(samplerate > 0.f, KG_ERR_INVALID, "kg_post_squelch_setup: sample rate %g", (double) samplerate);
    // squelch.cpp:84-116: the noise average's time constant, the high-pass above the voice band (:135-139), Reset()
    const float squelch_hp_freq = 3000.0;                                                       // VOICE_BANDWIDTH, :46, :106
    const float alpha = (1.0 - expf(-1.0 / (samplerate * .02)));                                // :107
    post_cfir &f = p->h_cfir[(size_t) ch * POST_NFIR + POST_FIR_SQ_HP];
    f.ntaps = cfir_design::highpass(0, 1.0, 50.0, squelch_hp_freq * .8, squelch_hp_freq * .65, samplerate, f.taps);
    if ((rc = post_cfir_upload(p, ch, POST_FIR_SQ_HP))) return rc;
    if ((rc = post_put(p, ch, &post_chan::sq_alpha, alpha))) return rc;
    return kg_post_squelch_reset(p, ch);
}

int kg_post_squelch_reset(kg_post *p, int ch)                   // squelch.cpp:67-77
{
    int rc = post_check(p, ch, "kg_post_squelch_reset");
    if (rc) return rc;
    if ((rc = post_put(p, ch, &post_chan::sq_ave, 0.f))) return rc;
    if ((rc = post_put(p, ch, &post_chan::sq_state, 1))) return rc;
    if ((rc = post_put(p, ch, &post_chan::sq_set, 0))) return rc;
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    return KG_OK;
}

int kg_post_squelch_set(kg_post *p, int ch, int Value, int SquelchMax)
{
    int rc = post_check(p, ch, "kg_post_squelch_set");
    if (rc) return rc;
    KG_REQUIRE(p->h_fir_ready[(size_t) ch * POST_NFIR + POST_FIR_SQ_HP], KG_ERR_STATE,
               "kg_post_squelch_set: kg_post_squelch_setup first (rx_sound.cpp:261-262)");
    if (SquelchMax == 0) SquelchMax = 8192;                                                     // SQUELCH_MAX = CLIPPER_NBFM_VAL, squelch.cpp:57
    const float threshold = (float) (SquelchMax - ((SquelchMax * Value) / 99));                 // :126
    if ((rc = post_put(p, ch, &post_chan::sq_value, (float) Value))) return rc;
    if ((rc = post_put(p, ch, &post_chan::sq_threshold, threshold))) return rc;
    if ((rc = post_put(p, ch, &post_chan::sq_set, 1))) return rc;
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    p->h_sq_ready[ch] = 1;
    return KG_OK;
}

int kg_post_squelch_state(kg_post *p, const int32_t *chans, int nch, int32_t *nsq_nc_sq, int32_t *squelched, float *ave)
{
    KG_REQUIRE(p && chans, KG_ERR_INVALID, "kg_post_squelch_state: null argument");
    int rc = kg_ctx_use(p->ctx);
    if (rc) return rc;
    std::vector<post_chan> h(p->nchan);
    KG_HIP(hipMemcpyAsync(h.data(), p->d_chan, sizeof(post_chan) * p->nchan, hipMemcpyDeviceToHost, p->ctx->stream));
    KG_HIP(hipStreamSynchronize(p->ctx->stream));
    for (int i = 0; i < nch; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < p->nchan, KG_ERR_INVALID, "kg_post_squelch_state: chans[%d] = %d", i, chans[i]);
        if (nsq_nc_sq) nsq_nc_sq[i] = h[chans[i]].sq_rc;
        if (squelched) squelched[i] = h[chans[i]].squelched;
        if (ave) ave[i] = h[chans[i]].sq_ave;
    }
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
