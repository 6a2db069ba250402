// kg_wire.hip -- wire formats: IMA ADPCM (sound 4:1, waterfall 2:1), wf_pkt_t, sound payload.
//
// Reference: rx/csdr/ima_adpcm.cpp:89-214; rx/rx_waterfall.h:73-89 and
// rx/rx_waterfall.cpp:284, 1602-1639; rx/rx_sound.h:42-52 and rx/rx_sound.cpp:1122-1140,
// 1219-1254.  The coder is a recurrence over (index, previousValue): one lane per
// stream, the streams of a launch in parallel.  Integer arithmetic, bit-exact.
#include "kg_common.h"

#include <math.h>

#include <stdlib.h>
#include <new>
#include <vector>

// The 89 step sizes of the IMA ADPCM specification (ima_adpcm.cpp:96-108)
__constant__ int c_step_size[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97,
    107, 118, 130, 143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724,
    796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026,
    4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500,
    20350, 22385, 24623, 27086, 29794, 32767
};

struct adpcm_state { int index, previous; };

// One sample: ImaAdpcmEncode + the ImaAdpcmDecode it ends with (ima_adpcm.cpp:161-181, 110-135).
// `tab` is the step table in LDS (a lane-indexed lookup on the recurrence's critical path).
template <int POS, int NEG>
__device__ __forceinline__ unsigned adpcm_step(int sample, int &index, int &previous, const int *tab)
{
    const int step = tab[index];
    int diff = sample - previous;
    unsigned code = 0;
    if (diff < 0) { code = 8; diff = -diff; }
    int st = step, difference = step >> 3;
    if (diff >= st) { code |= 4; diff -= st; difference += step; }
    st >>= 1;
    if (diff >= st) { code |= 2; diff -= st; difference += step >> 1; }
    st >>= 1;
    if (diff >= st) { code |= 1; difference += step >> 2; }
    if (code & 8) difference = -difference;
    previous += difference;
    previous = previous > POS ? POS : (previous < NEG ? NEG : previous);
    // indexAdjustTable (:89-94): -1 for magnitudes 0..3, 2/4/6/8 for 4..7
    index += (code & 4) ? 2 * ((int) (code & 3) + 1) : -1;
    index = index < 0 ? 0 : (index > 88 ? 88 : index);
    return code;
}

// encode_ima_adpcm_i16_e8 for many channels: lane = channel.
__global__ __launch_bounds__(64) void adpcm_snd_kernel(adpcm_state *__restrict__ states, const int *__restrict__ chans,
                                                       int nch, const short *__restrict__ in, size_t in_stride,
                                                       int nsamps, unsigned char *__restrict__ out, size_t out_stride,
                                                       int by_chan /* rows of in / out by channel (kg_ctx::rows_by_chan) */)
{
    __shared__ int tab[89];
    // a single wave's recurrence among workgroups that fill the vector units: it takes the issue priority (the coder is
    // latency, the others are throughput).  Round 5 measured two rewrites of this recurrence and kept neither: the five
    // step sizes the next sample can meet requested ahead of the code (the table read off the critical path) and the rows
    // staged through LDS with lane-contiguous loads -- 86 us alone against this form's 66: the chain of dependent compares
    // and selects, not the memory, is what a sample costs (130 ns in wf_packet_kernel either way).
    __builtin_amdgcn_s_setprio(3);
    for (int i = threadIdx.x; i < 89; i += 64) tab[i] = c_step_size[i];
    __syncthreads();
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= nch) return;
    const int ch = chans[li], row = by_chan ? ch : li;
    int index = states[ch].index, previous = states[ch].previous;
    const short *p = in + (size_t) row * in_stride;
    unsigned char *q = out + (size_t) row * out_stride;
    int i = 0;
    // four samples -> two bytes per round; rows are only guaranteed 2-byte aligned
    for (; i + 4 <= nsamps; i += 4) {
        const int s0 = p[i], s1 = p[i + 1], s2 = p[i + 2], s3 = p[i + 3];
        unsigned b0 = adpcm_step<32767, -32768>(s0, index, previous, tab);
        b0 |= adpcm_step<32767, -32768>(s1, index, previous, tab) << 4;
        unsigned b1 = adpcm_step<32767, -32768>(s2, index, previous, tab);
        b1 |= adpcm_step<32767, -32768>(s3, index, previous, tab) << 4;
        q[i / 2] = (unsigned char) b0;
        q[i / 2 + 1] = (unsigned char) b1;
    }
    for (; i + 2 <= nsamps; i += 2) {
        unsigned b = adpcm_step<32767, -32768>(p[i], index, previous, tab);
        b |= adpcm_step<32767, -32768>(p[i + 1], index, previous, tab) << 4;
        q[i / 2] = (unsigned char) b;
    }
    states[ch].index = index;
    states[ch].previous = previous;
}

__global__ void snd_payload_kernel(const unsigned short *__restrict__ in, size_t in_stride, int nch, int nsamps,
                                   int little_endian, unsigned short *__restrict__ out, size_t out_stride_words)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (j >= nsamps || row >= nch) return;
    unsigned short v = in[(size_t) row * in_stride + j];
    if (!little_endian) v = (unsigned short) ((v >> 8) | (v << 8));      // (x >> 8) & 0xff first, rx_sound.cpp:1136-1137
    out[(size_t) row * out_stride_words + j] = v;
}

struct wf_pkt_dev_info { unsigned x_bin_server, flags_x_zoom_server, seq; int use_compression; };

// One workgroup (one wave) per row: the row is staged in LDS, lane 0 runs the coder, the
// packet bytes are stored by all lanes.
__global__ __launch_bounds__(64) void wf_packet_kernel(const unsigned char *__restrict__ rows, size_t row_stride,
                                                       const wf_pkt_dev_info *__restrict__ info,
                                                       unsigned char *__restrict__ pkts, size_t pkt_stride)
{
    __shared__ int tab[89];
    __shared__ unsigned char s_in[KG_WF_ADPCM_PAD + 1024 + 2];
    __shared__ unsigned char s_out[KG_WF_PKT_MAX];
    __builtin_amdgcn_s_setprio(3);                 // one lane's recurrence: latency, not throughput (see adpcm_snd_kernel)
    const int lane = threadIdx.x, r = blockIdx.x;
    const wf_pkt_dev_info f = info[r];
    const unsigned char *row = rows + (size_t) r * row_stride;
    for (int i = lane; i < 89; i += 64) tab[i] = c_step_size[i];
    for (int i = lane; i < 1024; i += 64) s_in[KG_WF_ADPCM_PAD + i] = row[i];
    __syncthreads();
    if (lane < KG_WF_ADPCM_PAD) s_in[lane] = s_in[KG_WF_ADPCM_PAD];      // adpcm_pad <- buf2[0], rx_waterfall.cpp:1625
    if (lane == 0) {                                                     // header, little-endian u4_t fields
        const unsigned w[4] = {0x20462f57u /* "W/F " */, f.x_bin_server, f.flags_x_zoom_server, f.seq};
        for (int k = 0; k < 4; k++)
            for (int b = 0; b < 4; b++) s_out[4 * k + b] = (unsigned char) (w[k] >> (8 * b));
    }
    __syncthreads();
    int nbytes;
    if (f.use_compression) {
        nbytes = (KG_WF_ADPCM_PAD + 1024) / 2;
        if (lane == 0) {
            int index = 0, previous = 0;                                 // memset(&adpcm_wf, 0, ...) :1626
            for (int i = 0; i < KG_WF_ADPCM_PAD + 1024; i += 2) {
                unsigned b = adpcm_step<255, 0>(s_in[i], index, previous, tab);
                b |= adpcm_step<255, 0>(s_in[i + 1], index, previous, tab) << 4;
                s_out[KG_WF_PKT_HDR + i / 2] = (unsigned char) b;
            }
        }
    } else {
        nbytes = 1024;
        for (int i = lane; i < 1024; i += 64) s_out[KG_WF_PKT_HDR + i] = s_in[KG_WF_ADPCM_PAD + i];
    }
    __syncthreads();
    unsigned char *pkt = pkts + (size_t) r * pkt_stride;
    for (int i = lane; i < KG_WF_PKT_HDR + nbytes; i += 64) pkt[i] = s_out[i];
}

struct kg_adpcm {
    kg_ctx *ctx;
    int nchan;
    adpcm_state *d_state;
    kg_stage_cache list_cache;           // the channel list of the last encode call
};

extern "C" {

int kg_adpcm_create(kg_ctx *ctx, int nchan, kg_adpcm **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_adpcm_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nchan >= 1 && nchan <= (1 << 20), KG_ERR_INVALID, "kg_adpcm_create: nchan %d", nchan);
    kg_adpcm *a = new (std::nothrow) kg_adpcm();
    KG_REQUIRE(a != nullptr, KG_ERR_NOMEM, "kg_adpcm_create: alloc");
    a->ctx = ctx; a->nchan = nchan;
    KG_HIP(hipMalloc((void **) &a->d_state, sizeof(adpcm_state) * nchan));
    KG_HIP(hipMemset(a->d_state, 0, sizeof(adpcm_state) * nchan));
    *out = a;
    return KG_OK;
}

void kg_adpcm_destroy(kg_adpcm *a)
{
    if (!a) return;
    (void) hipSetDevice(a->ctx->device);
    (void) hipStreamSynchronize(a->ctx->stream);
    (void) hipFree(a->d_state);
    kg_stage_cache_free(&a->list_cache);
    delete a;
}

int kg_adpcm_set_state(kg_adpcm *a, int chan, int index, int previous)
{
    KG_REQUIRE(a != nullptr, KG_ERR_INVALID, "kg_adpcm_set_state: null object");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(chan >= 0 && chan < a->nchan, KG_ERR_INVALID, "kg_adpcm_set_state: channel %d", chan);
    KG_REQUIRE(index >= 0 && index <= 88 && previous >= -32768 && previous <= 32767, KG_ERR_INVALID,
               "kg_adpcm_set_state: index %d previous %d", index, previous);
    const adpcm_state s = {index, previous};
    KG_HIP(hipMemcpyAsync(a->d_state + chan, &s, sizeof s, hipMemcpyHostToDevice, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    return KG_OK;
}

int kg_adpcm_get_state(kg_adpcm *a, int chan, int *index, int *previous)
{
    KG_REQUIRE(a && index && previous, KG_ERR_INVALID, "kg_adpcm_get_state: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(chan >= 0 && chan < a->nchan, KG_ERR_INVALID, "kg_adpcm_get_state: channel %d", chan);
    adpcm_state s;
    KG_HIP(hipMemcpyAsync(&s, a->d_state + chan, sizeof s, hipMemcpyDeviceToHost, a->ctx->stream));
    KG_HIP(hipStreamSynchronize(a->ctx->stream));
    *index = s.index; *previous = s.previous;
    return KG_OK;
}

int kg_adpcm_encode_dev(kg_adpcm *a, const int32_t *chans, int nch, const void *d_s16, size_t in_stride,
                        int nsamps, void *d_out, size_t out_stride)
{
    KG_REQUIRE(a && chans && d_s16 && d_out, KG_ERR_INVALID, "kg_adpcm_encode_dev: null argument");
    int rc = kg_ctx_use(a->ctx);
    if (rc) return rc;
    KG_REQUIRE(nch >= 1 && nch <= a->nchan, KG_ERR_INVALID, "kg_adpcm_encode_dev: nch %d", nch);
    KG_REQUIRE(nsamps >= 2 && (nsamps & 1) == 0, KG_ERR_INVALID, "kg_adpcm_encode_dev: nsamps %d must be even", nsamps);
    KG_REQUIRE(in_stride >= (size_t) nsamps && out_stride >= (size_t) nsamps / 2, KG_ERR_INVALID,
               "kg_adpcm_encode_dev: stride too small");
    KG_REQUIRE(((uintptr_t) d_s16 & 1) == 0, KG_ERR_INVALID, "kg_adpcm_encode_dev: input not 2-byte aligned");
    std::vector<char> seen(a->nchan, 0);
    for (int i = 0; i < nch; i++) {
        KG_REQUIRE(chans[i] >= 0 && chans[i] < a->nchan && !seen[chans[i]], KG_ERR_INVALID,
                   "kg_adpcm_encode_dev: chans[%d] = %d out of range or listed twice", i, chans[i]);
        seen[chans[i]] = 1;
    }
    hipStream_t st = a->ctx->stream;
    void *d_list = nullptr;
    if ((rc = kg_ctx_stage_cached(a->ctx, &a->list_cache, chans, sizeof(int) * nch, &d_list))) return rc;
    KG_PLAN_ONLY(a->ctx);
    hipLaunchKernelGGL(adpcm_snd_kernel, dim3((nch + 63) / 64), dim3(64), 0, st, a->d_state, (const int *) d_list,
                       nch, (const short *) d_s16, in_stride, nsamps, (unsigned char *) d_out, out_stride, a->ctx->rows_by_chan);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

// The IQ modes' payload (rx/rx_sound.cpp:1076-1096): out_samps_c -> (s2_t) re, (s2_t) im per sample, as they are (little_endian) or in
// network order.  (s2_t) of a float: truncation; outside the int16 range the low 16 bits of the int32 conversion (kg_post.hip).
__global__ void snd_iq_payload_kernel(const float2 *__restrict__ in, size_t in_stride, const int *__restrict__ chans, int nch, int nsamps,
                                      int little_endian, unsigned short *__restrict__ out, size_t out_stride_words, int by_chan)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, li = blockIdx.y;
    if (j >= nsamps || li >= nch) return;
    const int row = (by_chan && chans) ? chans[li] : li;
    const float2 v = in[(size_t) row * in_stride + j];
    auto s2 = [](float f) -> unsigned short {
        int w;
        if (!(f > -2147483648.0f && f < 2147483648.0f)) w = (int) 0x80000000u;
        else w = (int) f;
        return (unsigned short) (unsigned) w;
    };
    unsigned short re = s2(v.x), im = s2(v.y);
    if (!little_endian) { re = (unsigned short) ((re >> 8) | (re << 8)); im = (unsigned short) ((im >> 8) | (im << 8)); }
    out[(size_t) row * out_stride_words + 2 * j] = re;
    out[(size_t) row * out_stride_words + 2 * j + 1] = im;
}

int kg_snd_iq_payload_dev(kg_ctx *ctx, const int32_t *chans, int nch, const void *d_cpx, size_t in_stride, int nsamps, int little_endian,
                          void *d_out, size_t out_stride)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(d_cpx && d_out && nch >= 1 && nsamps >= 1, KG_ERR_INVALID, "kg_snd_iq_payload_dev: bad argument");
    KG_REQUIRE(in_stride >= (size_t) nsamps && out_stride >= 4 * (size_t) nsamps && (out_stride & 1) == 0 &&
               ((uintptr_t) d_out & 1) == 0 && ((uintptr_t) d_cpx & 7) == 0, KG_ERR_INVALID,
               "kg_snd_iq_payload_dev: strides/alignments (out_stride in bytes, >= 4 nsamps, even)");
    KG_REQUIRE(chans || !ctx->rows_by_chan, KG_ERR_INVALID, "kg_snd_iq_payload_dev: a receiver bank's rows go by channel: pass the list");
    void *d_list = nullptr;
    if (chans && (rc = kg_ctx_stage(ctx, chans, sizeof(int) * (size_t) nch, &d_list))) return rc;
    KG_PLAN_ONLY(ctx);
    hipLaunchKernelGGL(snd_iq_payload_kernel, dim3((nsamps + 255) / 256, nch), dim3(256), 0, ctx->stream,
                       (const float2 *) d_cpx, in_stride, (const int *) d_list, nch, nsamps, little_endian,
                       (unsigned short *) d_out, out_stride / 2, ctx->rows_by_chan);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_snd_payload_dev(kg_ctx *ctx, const void *d_s16, size_t in_stride, int nch, int nsamps, int little_endian,
                       void *d_out, size_t out_stride)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(d_s16 && d_out && nch >= 1 && nsamps >= 1, KG_ERR_INVALID, "kg_snd_payload_dev: bad argument");
    KG_REQUIRE(in_stride >= (size_t) nsamps && out_stride >= 2 * (size_t) nsamps && (out_stride & 1) == 0 &&
               ((uintptr_t) d_out & 1) == 0 && ((uintptr_t) d_s16 & 1) == 0, KG_ERR_INVALID,
               "kg_snd_payload_dev: strides/alignments (out_stride in bytes, even)");
    hipLaunchKernelGGL(snd_payload_kernel, dim3((nsamps + 255) / 256, nch), dim3(256), 0, ctx->stream,
                       (const unsigned short *) d_s16, in_stride, nch, nsamps, little_endian,
                       (unsigned short *) d_out, out_stride / 2);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

void kg_snd_header(uint8_t flags, uint32_t seq, float smeter_dBm, uint8_t *h)
{
    if (!h) return;
    if (smeter_dBm < -127.0) smeter_dBm = -127.0; else           // rx_sound.cpp:1223-1224
    if (smeter_dBm > 3.4) smeter_dBm = 3.4;
    const uint16_t sm = (uint16_t) ((smeter_dBm + 127.0) * 10);   // :1225
    h[0] = 'S'; h[1] = 'N'; h[2] = 'D';                          // :252
    h[3] = flags;
    h[4] = (uint8_t) seq; h[5] = (uint8_t) (seq >> 8); h[6] = (uint8_t) (seq >> 16); h[7] = (uint8_t) (seq >> 24);
    h[8] = (uint8_t) (sm >> 8); h[9] = (uint8_t) sm;             // SET_BE_U16
}

static const double GPS_WEEK_SEC = 7 * 24 * 3600.0;             // rx/rx_sound.cpp:93

void kg_snd_gps_begin(kg_gps_state *s, double clk_gps_secs, double dticks, double adc_clock_base,
                      double gps_delay, double gps_delay2)
{
    if (!s) return;                                             // rx/rx_sound.cpp:557
    s->gpssec = fmod(GPS_WEEK_SEC + clk_gps_secs + (dticks / adc_clock_base) - gps_delay + gps_delay2, GPS_WEEK_SEC);
}

void kg_snd_gps_stamp(kg_gps_state *s, int norm_nrx_samps, int fir_pos, int agc_on, int agc_delay,
                      int rx_decim, double adc_clock_base, double clk_gps_secs, uint64_t clk_ticks,
                      kg_iq_stamp *out)
{
    if (!s || !out) return;
    int sample_filter_delays = norm_nrx_samps - fir_pos;        // :638 (1) delay in the FIR filter
    if (agc_on) sample_filter_delays -= agc_delay;              // :640-641 (2) delay in the AGC
    s->gpssec = fmod(GPS_WEEK_SEC + s->gpssec + (rx_decim * sample_filter_delays / adc_clock_base), GPS_WEEK_SEC);   // :652
    memset(out, 0, sizeof *out);
    out->gpssec = (uint32_t) s->last_gpssec;                    // :654
    out->gpsnsec = s->gps_init ? (uint32_t) (1e9 * (s->last_gpssec - out->gpssec)) : 0;    // :655
    const double dt_to_pos_sol = s->last_gpssec - clk_gps_secs; // :656
    out->last_gps_solution = s->gps_init ? ((clk_ticks == 0) ? 255 : (uint8_t) (dt_to_pos_sol < 252.0 ? dt_to_pos_sol : 252.0)) : 0;   // :658
    if (!s->gps_init) s->gps_init = 1;                          // :659
    s->last_gpssec = s->gpssec;                                 // :661
}

int kg_wf_packets_dev(kg_ctx *ctx, const void *d_rows, size_t row_stride, int nrows, const kg_wf_pkt_info *info,
                      void *d_pkts, size_t pkt_stride, int32_t *pkt_bytes)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(d_rows && info && d_pkts && pkt_bytes && nrows >= 1, KG_ERR_INVALID, "kg_wf_packets_dev: bad argument");
    KG_REQUIRE(row_stride >= 1024 && pkt_stride >= KG_WF_PKT_MAX, KG_ERR_INVALID,
               "kg_wf_packets_dev: row_stride >= 1024 and pkt_stride >= %d", KG_WF_PKT_MAX);
    std::vector<wf_pkt_dev_info> h(nrows);
    for (int i = 0; i < nrows; i++) {
        h[i].x_bin_server = info[i].x_bin_server;
        // "don't use compression for zoom level zero because of bad interaction of narrow strong carriers with compression
        // algorithm" (rx_waterfall.cpp:1283-1285): use_compression = wf->compression && wf->zoom != 0
        const bool comp = info[i].use_compression != 0 && info[i].zoom != 0;
        h[i].flags_x_zoom_server = info[i].zoom | (comp ? 0x00010000u : 0u);   // WF_FLAGS_COMPRESSION
        h[i].seq = info[i].seq;
        h[i].use_compression = comp;
        pkt_bytes[i] = KG_WF_PKT_HDR + (comp ? (KG_WF_ADPCM_PAD + 1024) / 2 : 1024);
    }
    void *d_info = nullptr;
    if ((rc = kg_ctx_stage(ctx, h.data(), sizeof(wf_pkt_dev_info) * nrows, &d_info))) return rc;
    KG_PLAN_ONLY(ctx);
    hipLaunchKernelGGL(wf_packet_kernel, dim3(nrows), dim3(64), 0, ctx->stream, (const unsigned char *) d_rows,
                       row_stride, (const wf_pkt_dev_info *) d_info, (unsigned char *) d_pkts, pkt_stride);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

}  // extern "C"
