// kg_wf.hip -- waterfall frames on gfx950.
//
// Replaces, for a batch of (channel, frame) pairs, the reference's
//   sample_wf() unpack + window          rx/rx_waterfall.cpp:1049-1066
//   compute_frame(): 8192-point FFT      :1291
//                    power (+ CIC comp)  :1303-1351
//                    FFT-bin -> pixel    :1400-1473  (MAX / MIN / LAST / DROP / CMA)
//                    dB, clamp, u8       :1489-1554
// One 256-thread workgroup per frame, persistent over the frame list.
//
// FFT: 8192 = 2 x 4096 (samples n = 2*n1 + g); only bins k < fft_used <= 4096 are
// ever looked at (:756-763), so the radix-2 combine is pruned to
//     X[k] = F0[k] + W_8192^{-k} * F1[k],   k < 4096
// and the two 4096-point transforms run back to back in the same two LDS tiles
// (kg_fft.h).  The power spectrum goes through LDS once (16 KiB, reusing tile A)
// to turn the strided per-thread bins into the contiguous runs the pixel stage
// walks.
//
// Pixel stage: the reference walks bins in ascending order and restarts a
// pixel whenever the mapped bin changes (:1458-1478).  The host turns its
// fft2wf_map[] into (first bin, count) runs per pixel -- for the reference's
// monotone maps that is the same partition, and a pixel that is never reached
// keeps pwr_out = 0 exactly as after the reference's memset (:1385).  Each
// thread then reduces four adjacent pixels, walking each run in ascending bin
// order, so MAX/MIN/LAST/CMA reproduce the serial loop's result bit for bit.
#include "kg_common.h"
#include "kg_fft.h"
#include "kg_libm.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#define WF_NFFT  8192        // rx/rx_waterfall.h:61-62
#define WF_WIDTH 1024        // rx/rx_waterfall.h:65
#define SUB      4096

enum { WF_MAX = 0, WF_MIN, WF_LAST, WF_DROP, WF_CMA };    // rx/rx_waterfall.h:116
enum { WINF_BLACKMAN_HARRIS = 2 };                         // rx/rx_waterfall.h:162

// Per-channel state in HBM (what compute_frame() reads from wf_inst_t).
struct wf_chan_dev {
    int zoom, window_func, interp, comp_on, dc, fft_used, limit, pwc;
    float fft_offset;
    int pad[3];
    unsigned short first[WF_WIDTH];     // run of bins of each pixel (DROP: the sampled bin)
    unsigned short count[WF_WIDTH];
    float scale[WF_WIDTH];              // what the dB stage multiplies by (WF_CMA: already divided by the run length)
};

#ifndef WF_PERSIST
#define WF_PERSIST 16     // window rows (of 16) a thread keeps from frame to frame
#endif
#define WF_TABLE_WAYS 8
#define WF_CLAIM_STRIDE 32      // ints between the frame-claim counters (a 128-byte line each): eight groups + the exit count
#define WF_LDS_BYTES (2 * SUB * sizeof(float2) + 240 * sizeof(float2) + 16)       // + pass-1 twiddles + the claimed frame index

template <bool TAPS>
__global__ __launch_bounds__(256, 2) void wf_frame_kernel(
    const short2 *__restrict__ iq,            // iq_t {i, q}: frame f = 8192 of them from iq + 2 * frames[f].y
    const int2 *__restrict__ frames,          // [nframes] {channel, offset of the frame's first sample in units of two iq_t}
    const wf_chan_dev *__restrict__ chans,
    const float *__restrict__ windows,        // [4][8192]
    const float *__restrict__ cic_comp,       // [8192], kg_wf_set_tables' order: pairs of bins, then 4096 x 1.0f
    const float2 *__restrict__ tab4096, const float2 *__restrict__ tab8192,
    int nframes,
    unsigned char *__restrict__ out,          // [nframes][1024]
    int *__restrict__ claim,                  // [9][WF_CLAIM_STRIDE]: per group, frames handed out beyond the first two per workgroup; [8]: workgroups done
    float *__restrict__ tap_pwr, float *__restrict__ tap_pwr_out, float *__restrict__ tap_db)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2 *tileA = smem, *tileB = smem + SUB;
    float *pwr = (float *) smem;              // [4096], reuses tile A after the transforms
    float2 *tw1 = smem + 2 * SUB;
    volatile int *lds_claim = (volatile int *) (tw1 + 240);
    const int t = threadIdx.x;

    // pass-1 twiddles from an LDS table, pass-2 twiddles in registers: the 30 registers saved hold the
    // second parity of the frame and of the window (8-byte loads: 32 fewer load instructions per frame)
    kg_tw1_fill(tw1, tab4096, t);
    kg_tw15 p2;
#pragma unroll
    for (int j = 1; j < 16; j++) p2.w[j - 1] = kg_ld(&tab4096[j * t]);
    // combine twiddle W_8192^{k}, k = t + 256 m: W_8192^{t} (per thread) * W_32^{m} (immediate)
    const cf wbase = kg_ld(&tab8192[t]);

    // A frame is read as sixteen 8-byte loads per thread (samples 2 n1 and 2 n1 + 1, n1 = t + 256 j: the even ones feed the
    // first 4096-point transform, the odd ones the second) into registers the previous frame has just consumed, a frame
    // ahead: the frame fetch is the kernel's one HBM access, and at the top of the loop its whole latency was exposed.
    // All three streams -- frame, window, CIC factors -- are buffer loads: a scalar descriptor per stream, one lane offset
    // (8 t), the row as an immediate / scalar offset.
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    int2 raw[16];
    const int vo8 = t * 8;
    auto ld8 = [&](__amdgpu_buffer_rsrc_t rs, int row) {
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo8, row * 2048, 0);
        return v;
    };
    auto frame_rsrc = [&](int off2) {
        return __builtin_amdgcn_make_buffer_rsrc((void *) ((const int2 *) iq + (size_t) (unsigned) off2), 0, WF_NFFT * 4, 0x00020000);
    };
    auto window_rsrc = [&](int wfn) {
        return __builtin_amdgcn_make_buffer_rsrc((void *) (windows + (size_t) wfn * WF_NFFT), 0, WF_NFFT * 4, 0x00020000);
    };
    auto fetch = [&](int off2) {
        const __amdgpu_buffer_rsrc_t rs = frame_rsrc(off2);
#pragma unroll
        for (int j = 0; j < 16; j++) { const u2 v = ld8(rs, j); raw[j] = int2{(int) v[0], (int) v[1]}; }   // samples 2 n1 (even), 2 n1 + 1 (odd), n1 = t + 256 j
    };
    // window values of this thread's thirty-two samples (8-byte loads, L1 / L2 hits)
    float2 wv[16];
    auto fetch_window = [&](int wfn) {
        const __amdgpu_buffer_rsrc_t rs = window_rsrc(wfn);
#pragma unroll
        for (int j = 0; j < 16; j++) { const u2 v = ld8(rs, j); wv[j] = float2{__uint_as_float(v[0]), __uint_as_float(v[1])}; }
    };
    auto windowed = [&](cf (&x)[16], int g) {
        // sample_wf(): fi = (float)(s2_t)i * window[sn]  (:1054-1061)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int r = g ? raw[j].y : raw[j].x;
            const float w = g ? wv[j].y : wv[j].x;
            // two SDWA conversions (sign-extended halves), ONE packed multiply with the window value broadcast
            x[j] = cf{(float) (short) (r & 0xffff), (float) (short) (r >> 16)} * cf{w, w};
        }
    };

    // Everything a frame needs first -- its channel number, that channel's window, its even samples
    // and their window values -- is requested while the previous frame is still being computed, so a
    // frame starts with its first transform instead of a chain of three dependent memory round trips
    // (chan_of[f] -> the channel record -> window and samples).
    // Frames are handed out dynamically after the first two of a workgroup (frame b and b + grid): the
    // two workgroups of a CU do not run at the same speed (the older one wins the vector-issue
    // arbitration), so equal static shares leave CUs half empty at the end.  The frame after next is
    // claimed at the top of a frame and crosses the workgroup through LDS at the power-stage barrier.
    // The last workgroup to leave resets the two counters for the next launch.
    // Round 4: EIGHT claim counters, one per group of workgroups (blockIdx % 8: the workgroups of one XCD), each handing out the
    // frames f = 8 k + g of its group.  One counter for the whole chip was the kernel's floor: 28 672 agent-scope atomics on
    // one address take 0.36 ms whatever the frames cost (knock-out builds without butterflies, without tile traffic, without
    // the frame fetch all ran 0.363 ms; tools/ko_wf.sh).  A grid that is not a multiple of eight (a short list)
    // keeps one group.
    const int cng = (gridDim.x & 7) == 0 ? 8 : 1, cg = blockIdx.x & (cng - 1);
    int f = blockIdx.x, fn = blockIdx.x + gridDim.x;
    int cid;                                  // grid <= nframes: every workgroup has a first frame
    { const int2 fr = frames[f]; cid = fr.x; fetch(fr.y); }
    // the record of the frame after this one is fetched a frame ahead (at the bottom of the loop, as soon as its
    // index is known): at the top of a frame it would be a scalar-cache round trip with nothing to hide behind
    int2 fr_next = frames[fn < nframes ? fn : f];
    // Round 4: a thread's thirty-two window values are the same for every frame of a window function, and the channels of a
    // receiver nearly always share one (the client's default): they stay in their registers from frame to frame and are
    // re-read only when the next frame's channel uses another window (a wave-uniform branch around the loads; the
    // conservative wait counts the compiler merges at its join only cost the frames that do reload).
    int wfn_cur = chans[cid].window_func;
    fetch_window(wfn_cur);
    // ... and so are the sixteen CIC compensation factors of its bins (one table, applied or not per channel; a zoomed channel
    // reads the lower eight only): kept while the next frame wants the same ones
    float cicv[16];
    int cic_have = -1;                        // -1 nothing, else comp_on * 2 + (all sixteen loaded)
    for (;;) {
        const wf_chan_dev *ch = chans + cid;
        const int interp = ch->interp, dc = ch->dc, comp_on = ch->comp_on;
        // Round 4: a zoomed channel displays bins below fft_used = 2048 of the 8192 (rx/rx_waterfall.cpp:756-763; zoom 0:
        // 4096) -- 13 of BASELINE configs[2]'s 14 channels.  For those frames the upper half of the radix-2 combine, of the
        // CIC factors and of the power stage (k = t + 256 m, m >= 8) is never read: skipped behind ONE wave-uniform branch
        // (two copies of the frame body, with the last butterflies pruned as well, spilled 79 registers).
        const bool half = ch->fft_used <= 2048;
        const bool more = fn < nframes;
        // (the counter's value only: anything computed from it here would be waited for here.  Built with the
        // atomic optimizer off -- Makefile: its wave-aggregated form reads the result back with v_readfirstlane
        // right behind the atomic, an s_waitcnt vmcnt(0) at the top of the frame that made wave 0 reach the first
        // exchange barrier a memory round trip late, with the other three waiting there.)
        int claimed = 0;
        if (t == 0) claimed = __hip_atomic_fetch_add(&claim[cg * WF_CLAIM_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        cf x[16], y0[16], y1[16];
        windowed(x, 0);
        kg_subfft4096_l<-1>(x, y0, tileA, tileB, tw1, p2, t);
        // (the next frame's record was requested at the bottom of the previous frame; read here, not at the top of the
        // frame, where the dependent load of its channel's window number made every wave wait for it before its first
        // instruction of arithmetic)
        // (the compiler otherwise sinks half of the last butterflies of y0 to their use in the combine stage and keeps their
        // inputs AND the other half alive through the whole second transform: 46 registers instead of 32)
#pragma unroll
        for (int m = 0; m < 16; m++) asm volatile("" : "+v"(y0[m]));
        kg_pin();
        const int cid_next = fr_next.x;
        const int wfn_next = chans[cid_next].window_func;
        const bool rewin = wfn_next != wfn_cur;
        wfn_cur = wfn_next;
        kg_pin();
        windowed(x, 1);
        // Everything the second transform's duration can hide is requested DURING it, at eight points (kg_subfft4096_l_h):
        // in its first half the next frame's window values IF they are not the ones held, in its second half -- where the
        // pass-1 twiddles' registers are free -- the next frame (both parities, into the registers just consumed; the last
        // frame of a workgroup re-reads its own: one load site, never skipped -- a conditional one makes the compiler drain
        // vmcnt at the join) and this frame's CIC compensation factors (1.0f = the table's second half where the frame is not
        // compensated: x * 1.0f is exact) IF they are not the ones held.  As bursts from four waves at once the loads
        // queued behind each other in the CU's one texture addresser (the same finding as in acq_correlate8_kernel,
        // DESIGN.md 2.6).
        const int con = comp_on ? 1 : 0, cic_want = con * 2 + (half ? 0 : 1);
        const bool recic = (cic_have >> 1) != con || (!half && !(cic_have & 1));
        if (recic) cic_have = cic_want;
        {
            const __amdgpu_buffer_rsrc_t nsrc = frame_rsrc(fr_next.y);
            const __amdgpu_buffer_rsrc_t nwin = window_rsrc(wfn_next);
            const __amdgpu_buffer_rsrc_t cp = __builtin_amdgcn_make_buffer_rsrc(
                (void *) ((const float2 *) cic_comp + (comp_on ? 0 : 2048)), 0, 2048 * 8, 0x00020000);
            kg_subfft4096_l_h<-1>(x, y1, tileA, tileB, tw1, p2, t, [&](int k) {
                kg_pin();
                if (k < 4) {
                    if (k < WF_PERSIST / 4) {
                        if (rewin) {
#pragma unroll
                            for (int j = 4 * k; j < 4 * k + 4; j++) { const u2 v = ld8(nwin, j); wv[j] = float2{__uint_as_float(v[0]), __uint_as_float(v[1])}; }
                        }
                    } else {
#pragma unroll
                        for (int j = 4 * k; j < 4 * k + 4; j++) { const u2 v = ld8(nwin, j); wv[j] = float2{__uint_as_float(v[0]), __uint_as_float(v[1])}; }
                    }
                } else {
                    const int q = k - 4;
#pragma unroll
                    for (int j = 4 * q; j < 4 * q + 4; j++) { const u2 v = ld8(nsrc, j); raw[j] = int2{(int) v[0], (int) v[1]}; }
#pragma unroll
                    for (int j = 2 * q; j < 2 * q + 2; j++)
                        if (recic && (j < 4 || !half)) { const u2 v = ld8(cp, j); cicv[2 * j] = __uint_as_float(v[0]); cicv[2 * j + 1] = __uint_as_float(v[1]); }
                }
                kg_pin();
            });
        }
        // (tile A, about to become pwr[], was last read before the second transform's
        // second barrier)
        // this thread's four pixels: run starts, run lengths, scales -- four vector loads
        // issued here so that the combine / power stage hides them
        const ushort4 pf = ((const ushort4 *) ch->first)[t];
        const ushort4 pc = ((const ushort4 *) ch->count)[t];
        const float4 ps = ((const float4 *) ch->scale)[t];
        const int pfirst[4] = {pf.x, pf.y, pf.z, pf.w}, pcount[4] = {pc.x, pc.y, pc.z, pc.w};
        const float pscale[4] = {ps.x, ps.y, ps.z, ps.w};
        const int pwc = ch->pwc;
        const float fft_offset = ch->fft_offset;
        // X[k] = F0[k] + F1[k] * conj(W_8192^t) * conj(W_32^m), k = t + 256 m: the two factors applied to
        // all sixteen bins in blocks of four products (kg_fft.h), then the sum
#define WF_W32(m) cf{KG_W64[2 * (m)][0], KG_W64[2 * (m)][1]}
        auto combine_power = [&](auto m0_tag) {           // bins k = t + 256 m, m = m0 .. m0 + 7
            constexpr int m0 = decltype(m0_tag)::value;
            kg_cmul4v<true>(y1[m0], y1[m0 + 1], y1[m0 + 2], y1[m0 + 3], wbase, wbase, wbase, wbase);
            kg_cmul4v<true>(y1[m0 + 4], y1[m0 + 5], y1[m0 + 6], y1[m0 + 7], wbase, wbase, wbase, wbase);
            kg_cmul4s<true>(y1[m0], y1[m0 + 1], y1[m0 + 2], y1[m0 + 3], WF_W32(m0), WF_W32(m0 + 1), WF_W32(m0 + 2), WF_W32(m0 + 3));
            kg_cmul4s<true>(y1[m0 + 4], y1[m0 + 5], y1[m0 + 6], y1[m0 + 7], WF_W32(m0 + 4), WF_W32(m0 + 5), WF_W32(m0 + 6), WF_W32(m0 + 7));
#pragma unroll
            for (int m = m0; m < m0 + 8; m++) {
                const int k = t + 256 * m;
                cf X = y0[m] + y1[m];
                X = X * cf{cicv[m], cicv[m]};     // re *= CIC_comp[k], im *= CIC_comp[k] (:1342); 1.0f when off: exact
                const cf sq = X * X;
                float p = sq.x + sq.y;                                                      // re*re + im*im, :1345
                if (m == 0 && k < dc) p = 0.f;                                              // :1304 (dc <= 4: row 0 only)
                pwr[k] = p;
                if (TAPS && k < ch->fft_used) tap_pwr[(size_t) f * SUB + k] = p;
            }
        };
        combine_power(std::integral_constant<int, 0>());
        if (!half) combine_power(std::integral_constant<int, 8>());
#undef WF_W32
        if (t == 0) *lds_claim = 2 * gridDim.x + cng * claimed + cg;
        __syncthreads();
        const int fnn = __builtin_amdgcn_readfirstlane(*lds_claim);    // wave-uniform; rewritten after >= 6 barriers
        cid = cid_next;

        // pixels 4t .. 4t+3.  The interpolation mode is the frame's (wave-uniform); the run lengths
        // differ per pixel, so the walk goes to the longest run of the wave with the shorter ones
        // predicated: no divergent loops, the four pixels of a thread advance together, and every
        // pixel still sees its bins in ascending order (same result as the serial :1458-1478 loop).
        float pp[4] = {0.f, 0.f, 0.f, 0.f};   // memset(pwr_out, 0), :1385
        if (interp == WF_DROP) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (4 * t + u < pwc) pp[u] = pwr[pfirst[u]];                            // :1418
        } else {
            // (DPP reduce: six dependent ds_bpermute round trips per frame otherwise)
            const int cmax = kg_wave_max(max(max(pcount[0], pcount[1]), max(pcount[2], pcount[3])));
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (pcount[u] > 0) pp[u] = pwr[pfirst[u]];                             // :1468-1475
            int plast[4];                     // last bin of each run (the run's first bin when it is empty)
#pragma unroll
            for (int u = 0; u < 4; u++) plast[u] = pfirst[u] + max(pcount[u] - 1, 0);
            auto walk = [&](auto upd) {
                for (int i = 1; i < cmax; i++) {
                    float q[4];               // four unconditional LDS reads (index clamped into the run), one wait
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = pwr[min(pfirst[u] + i, plast[u])];
#pragma unroll
                    for (int u = 0; u < 4; u++) pp[u] = (i < pcount[u]) ? upd(pp[u], q[u]) : pp[u];
                }
            };
            switch (interp) {                                                           // :1461-1466
            case WF_CMA:  walk([](float p, float q) { return p + q; }); break;
            case WF_MAX:  walk([](float p, float q) { return q > p ? q : p; }); break;
            case WF_MIN:  walk([](float p, float q) { return q < p ? q : p; }); break;
            default:      walk([](float, float q) { return q; }); break;               // WF_LAST
            }
        }
        unsigned bytes = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int px = 4 * t + u;
            const float p = pp[u];
            // (WF_CMA: the divisor of :1499-1500 -- the pixel's run length, fixed per channel -- is already in
            // the scale the host stored: fft_scale, fft_scale_div2 or fft_scale / avgs, kg_wf_set_channel)
            const float scale = pscale[u];
            // dB = 10.0 * log10f(p*scale + 1e-30F) + fft_offset: the product and sum in
            // float, the 10.0* and + in double, one rounding to float (:1507).  log10f as v_log_f32 (log2, 1 ulp)
            // times log10(2): one instruction and a multiply instead of the library's twenty-odd; the argument
            // is never below 1e-30 (no denormal path), NaN stays NaN; the difference from log10f is a few 1e-6
            // in l, i.e. some 1e-5 dB, inside the bound the parity tests allow for the last ulp of log10f
            const float arg = p * scale + 1e-30f;
#ifdef KG_EXP_WF_LOG10F
            const float l = kg_libm::log10f_glibc(arg);
#else
            const float l = __builtin_amdgcn_logf(arg) * 0.30102999566398120f;
#endif
            float dB = (float) (10.0 * (double) l + (double) fft_offset);
            if (TAPS) { tap_pwr_out[(size_t) f * WF_WIDTH + px] = p; tap_db[(size_t) f * WF_WIDTH + px] = dB; }
            unsigned b;
            if (dB != dB) {
                b = 0;                        // (u1_t)(int)NaN: an untouched CMA pixel (0 * inf)
            } else {
                if (dB > 0.f) dB = 0.f;                                                  // :1543
                if (dB < -200.0f) dB = -200.0f;
                dB = dB - 1.0f;
                b = (unsigned) ((int) dB) & 0xffu;                                       // :1546
            }
            bytes |= b << (8 * u);
        }
        ((unsigned *) (out + (size_t) f * WF_WIDTH))[t] = bytes;
        __syncthreads();                      // pwr[] (tile A) is rewritten by the next frame
        if (!more) break;
        fr_next = frames[fnn < nframes ? fnn : fn];
        f = fn; fn = fnn;
    }
    if (t == 0) {
        const int done = __hip_atomic_fetch_add(&claim[8 * WF_CLAIM_STRIDE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int) gridDim.x - 1) {    // every other workgroup has made its last claim
            for (int g = 0; g <= 8; g++) __hip_atomic_store(&claim[g * WF_CLAIM_STRIDE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------
struct kg_wf {
    kg_ctx *ctx;
    int nchan;
    wf_chan_dev *d_chans;
    float *d_windows, *d_cic;
    short2 *d_iq;    unsigned char *d_out;  int stage_cap;      // staging for host-buffer calls
    kg_stage_cache chan_of_cache[WF_TABLE_WAYS];   // the {channel, offset} records of the last few distinct batches (a stream cycles through a few)
    int chan_of_victim;
    std::vector<int2> frame_tab;
    float *d_tap_pwr, *d_tap_pwr_out, *d_tap_db;
    int *d_claim;                             // the frame kernels' claim counters, zero between launches
    std::vector<char> chan_set;
    bool tables_set;
    int grid;
};

extern "C" {

int kg_wf_create(kg_ctx *ctx, int nchan, kg_wf **out)
{
    int rc = kg_ctx_use(ctx);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_wf_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nchan >= 1 && nchan <= 65536, KG_ERR_INVALID, "kg_wf_create: nchan %d", nchan);
    kg_wf *w = new (std::nothrow) kg_wf();
    KG_REQUIRE(w != nullptr, KG_ERR_NOMEM, "kg_wf_create: alloc");
    w->ctx = ctx; w->nchan = nchan; w->tables_set = false;
    w->chan_set.assign(nchan, 0);
    w->d_iq = nullptr; w->d_out = nullptr; w->stage_cap = 0;
    w->d_tap_pwr = w->d_tap_pwr_out = w->d_tap_db = nullptr;
    KG_HIP(hipMalloc((void **) &w->d_chans, sizeof(wf_chan_dev) * nchan));
    KG_HIP(hipMalloc((void **) &w->d_windows, sizeof(float) * 4 * WF_NFFT));
    KG_HIP(hipMalloc((void **) &w->d_cic, sizeof(float) * WF_NFFT));
    KG_HIP(hipMalloc((void **) &w->d_claim, sizeof(int) * 9 * WF_CLAIM_STRIDE));
    KG_HIP(hipMemset(w->d_claim, 0, sizeof(int) * 9 * WF_CLAIM_STRIDE));
    KG_HIP(hipFuncSetAttribute((const void *) wf_frame_kernel<false>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES));
    KG_HIP(hipFuncSetAttribute((const void *) wf_frame_kernel<true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES));
    int occ = 0;
    KG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wf_frame_kernel<false>, 256, WF_LDS_BYTES));
    if (occ < 1) occ = 1;
    if (const char *e = kg_tuning_env("KIWIGPU_WF_WGS_PER_CU")) {      // experiment: fewer workgroups per CU
        const int v = atoi(e);
        if (v >= 1 && v < occ) occ = v;
    }
    w->grid = ctx->num_cus * occ;
    *out = w;
    return KG_OK;
}

void kg_wf_destroy(kg_wf *w)
{
    if (!w) return;
    (void) hipSetDevice(w->ctx->device);
    (void) hipStreamSynchronize(w->ctx->stream);
    (void) hipFree(w->d_chans); (void) hipFree(w->d_windows); (void) hipFree(w->d_cic); (void) hipFree(w->d_claim);
    (void) hipFree(w->d_iq); (void) hipFree(w->d_out);
    (void) hipFree(w->d_tap_pwr); (void) hipFree(w->d_tap_pwr_out); (void) hipFree(w->d_tap_db);
    for (int k = 0; k < WF_TABLE_WAYS; k++) kg_stage_cache_free(&w->chan_of_cache[k]);
    delete w;
}

int kg_wf_set_tables(kg_wf *w, const float *windows, const float *cic_comp)
{
    KG_REQUIRE(w && windows && cic_comp, KG_ERR_INVALID, "kg_wf_set_tables: null argument");
    int rc = kg_ctx_use(w->ctx);
    if (rc) return rc;
    hipStream_t st = w->ctx->stream;
    KG_HIP(hipMemcpyAsync(w->d_windows, windows, sizeof(float) * 4 * WF_NFFT, hipMemcpyHostToDevice, st));
    // device copy in the frame kernel's own order: the factors of bins t + 256 m and t + 256 (m + 1), m
    // even, are adjacent at (m / 2) 512 + 2 t: eight lane-contiguous 8-byte loads per frame instead of
    // sixteen 4-byte ones.  Entries 4096.. hold 1.0f: an uncompensated frame reads those (x * 1.0f is exact).
    {
        std::vector<float> tr(WF_NFFT, 1.0f);
        for (int t = 0; t < 256; t++)
            for (int m = 0; m < 16; m++) tr[(m >> 1) * 512 + 2 * t + (m & 1)] = cic_comp[t + 256 * m];
        // in stream order like the window copy (frames still queued read the old table); the synchronise
        // below keeps `tr` alive until the transfer has run
        KG_HIP(hipMemcpyAsync(w->d_cic, tr.data(), sizeof(float) * WF_NFFT, hipMemcpyHostToDevice, st));
        KG_HIP(hipStreamSynchronize(st));
    }
    w->tables_set = true;
    return KG_OK;
}

int kg_wf_set_channel(kg_wf *w, int ch, const kg_wf_chan_cfg *cfg, const uint16_t *fft2wf_map,
                      const uint16_t *drop_sample, const float *fft_scale, const float *fft_scale_div2)
{
    KG_REQUIRE(w && cfg && fft2wf_map && drop_sample && fft_scale && fft_scale_div2, KG_ERR_INVALID,
               "kg_wf_set_channel: null argument");
    int rc = kg_ctx_use(w->ctx);
    if (rc) return rc;
    KG_REQUIRE(ch >= 0 && ch < w->nchan, KG_ERR_INVALID, "kg_wf_set_channel: channel %d (0..%d)", ch, w->nchan - 1);
    KG_REQUIRE(cfg->fft_used >= 1 && cfg->fft_used <= SUB, KG_ERR_INVALID,
               "kg_wf_set_channel: fft_used %d (1..%d)", cfg->fft_used, SUB);
    KG_REQUIRE(cfg->window_func >= 0 && cfg->window_func < 4, KG_ERR_INVALID,
               "kg_wf_set_channel: window_func %d", cfg->window_func);
    KG_REQUIRE(cfg->interp >= WF_MAX && cfg->interp <= WF_CMA, KG_ERR_INVALID,
               "kg_wf_set_channel: interp %d", cfg->interp);
    KG_REQUIRE(cfg->plot_width_clamped >= 0 && cfg->plot_width_clamped <= WF_WIDTH, KG_ERR_INVALID,
               "kg_wf_set_channel: plot_width_clamped %d", cfg->plot_width_clamped);
    // the only branch of compute_frame() this library implements (:1400); the other
    // one reads the map index as power (:1584-1592) and is unreachable for FlyDog
    KG_REQUIRE(cfg->fft_used >= cfg->plot_width, KG_ERR_INVALID,
               "kg_wf_set_channel: fft_used %d < plot_width %d is not supported", cfg->fft_used, cfg->plot_width);
    static thread_local wf_chan_dev h;
    memset(&h, 0, sizeof h);
    h.zoom = cfg->zoom; h.window_func = cfg->window_func; h.interp = cfg->interp;
    h.comp_on = (cfg->zoom > 1 && cfg->cic_comp && !cfg->overlapped) ? 1 : 0;        // :1324,:1335
    h.dc = (cfg->zoom == 0 && cfg->window_func == WINF_BLACKMAN_HARRIS) ? 4 : 2;     // :1303
    h.fft_used = cfg->fft_used; h.pwc = cfg->plot_width_clamped; h.fft_offset = cfg->fft_offset;
    // fft_used_limit: the first bin that maps outside the plot (:1433-1447)
    int limit = cfg->fft_used;
    for (int i = 0; i < cfg->fft_used; i++)
        if (fft2wf_map[i] >= WF_WIDTH) { limit = i; break; }
    h.limit = limit;
    if (cfg->interp == WF_DROP) {
        for (int i = 0; i < cfg->plot_width_clamped; i++) {
            // (bins at and above fft_used are never displayed -- rx_waterfall.cpp:756-763 -- and, for fft_used <= 2048, no longer
            // computed: an entry up there would read what an earlier frame left in the tile)
            KG_REQUIRE(drop_sample[i] < cfg->fft_used, KG_ERR_INVALID, "kg_wf_set_channel: drop_sample[%d] = %d is not below fft_used %d", i,
                       drop_sample[i], cfg->fft_used);
            h.first[i] = drop_sample[i]; h.count[i] = 1;
        }
    } else {
        int i = 0;
        while (i < limit) {                   // maximal runs of equal mapped bin
            const int bin = fft2wf_map[i];
            int j = i + 1;
            while (j < limit && fft2wf_map[j] == bin) j++;
            h.first[bin] = (unsigned short) i;      // a later run of the same bin restarts it (:1468)
            h.count[bin] = (unsigned short) (j - i);
            i = j;
        }
    }
    // The scale the dB stage multiplies by: for WF_CMA the reference picks fft_scale, fft_scale_div2 or
    // fft_scale / avgs by the pixel's bin count (:1499-1500), a property of the map, so it is resolved here
    // (the same float division); a CMA pixel no bin maps to keeps avgs = 0: scale / 0 = inf, 0 * inf = NaN,
    // (u1_t)(int)NaN = byte 0, as in the reference.
    for (int i = 0; i < WF_WIDTH; i++) {
        float sc = fft_scale[i];
        if (cfg->interp == WF_CMA) {
            const int avgs = h.count[i];
            sc = (avgs == 1) ? fft_scale[i] : ((avgs == 2) ? fft_scale_div2[i] : fft_scale[i] / (float) avgs);
        }
        h.scale[i] = sc;
    }
    hipStream_t st = w->ctx->stream;
    KG_HIP(hipStreamSynchronize(st));         // frames in flight may read the old record
    KG_HIP(hipMemcpy(w->d_chans + ch, &h, sizeof h, hipMemcpyHostToDevice));
    w->chan_set[ch] = 1;
    return KG_OK;
}

static int wf_launch(kg_wf *w, int nframes, const int32_t *chan_of, const uint64_t *frame_off, uint64_t iq_len,
                     const void *d_iq, void *d_out, bool taps)
{
    KG_REQUIRE(w->tables_set, KG_ERR_STATE, "kg_wf_frames: kg_wf_set_tables was not called");
    KG_REQUIRE(nframes >= 1, KG_ERR_INVALID, "kg_wf_frames: nframes %d", nframes);
    for (int f = 0; f < nframes; f++) {
        KG_REQUIRE(chan_of[f] >= 0 && chan_of[f] < w->nchan, KG_ERR_INVALID,
                   "kg_wf_frames: chan_of[%d] = %d out of range", f, chan_of[f]);
        KG_REQUIRE(w->chan_set[chan_of[f]], KG_ERR_STATE, "kg_wf_frames: channel %d is not configured", chan_of[f]);
    }
    hipStream_t st = w->ctx->stream;
    // per-frame records {channel, offset / 2}: frames back to back unless the caller says where each one starts
    w->frame_tab.resize(nframes);
    for (int f = 0; f < nframes; f++) {
        const uint64_t off = frame_off ? frame_off[f] : (uint64_t) f * WF_NFFT;
        KG_REQUIRE((off & 1) == 0 && off < ((uint64_t) 1 << 32), KG_ERR_INVALID,
                   "kg_wf_frames: frame_off[%d] = %llu (even, below 2^32 samples; back-to-back frames: nframes <= 524288)",
                   f, (unsigned long long) off);
        // the caller states the extent of what d_iq points at: a stale offset must not become an out-of-bounds read
        KG_REQUIRE(!frame_off || off + WF_NFFT <= iq_len, KG_ERR_INVALID,
                   "kg_wf_frames_at_dev: frame %d at %llu + 8192 runs past iq_len %llu", f, (unsigned long long) off,
                   (unsigned long long) iq_len);
        w->frame_tab[f] = make_int2(chan_of[f], (int) (unsigned) (off >> 1));
    }
    void *d_chan_of = nullptr;                 // the arrays are the caller's: staged copy, no stream synchronisation
    {
        int rc = kg_ctx_stage_cached_ways(w->ctx, w->chan_of_cache, WF_TABLE_WAYS, &w->chan_of_victim, w->frame_tab.data(),
                                          sizeof(int2) * nframes, &d_chan_of);
        if (rc) return rc;
    }
    KG_PLAN_ONLY(w->ctx);
    const int grid = nframes < w->grid ? nframes : w->grid;
    if (!taps) {
        hipLaunchKernelGGL(wf_frame_kernel<false>, dim3(grid), dim3(256), WF_LDS_BYTES, st,
                           (const short2 *) d_iq, (const int2 *) d_chan_of, (const wf_chan_dev *) w->d_chans,
                           (const float *) w->d_windows, (const float *) w->d_cic,
                           (const float2 *) w->ctx->d_tab4096, (const float2 *) w->ctx->d_tab8192, nframes,
                           (unsigned char *) d_out, w->d_claim, (float *) nullptr, (float *) nullptr, (float *) nullptr);
    } else {
        hipLaunchKernelGGL(wf_frame_kernel<true>, dim3(grid), dim3(256), WF_LDS_BYTES, st,
                           (const short2 *) d_iq, (const int2 *) d_chan_of, (const wf_chan_dev *) w->d_chans,
                           (const float *) w->d_windows, (const float *) w->d_cic,
                           (const float2 *) w->ctx->d_tab4096, (const float2 *) w->ctx->d_tab8192, nframes,
                           (unsigned char *) d_out, w->d_claim, w->d_tap_pwr, w->d_tap_pwr_out, w->d_tap_db);
    }
    KG_HIP(hipGetLastError());
    return KG_OK;
}

int kg_wf_frames_dev(kg_wf *w, int nframes, const int32_t *chan_of, const void *d_iq, void *d_out)
{
    KG_REQUIRE(w && chan_of && d_iq && d_out, KG_ERR_INVALID, "kg_wf_frames_dev: null argument");
    int rc = kg_ctx_use(w->ctx);
    if (rc) return rc;
    KG_REQUIRE(((uintptr_t) d_iq & 7) == 0 && ((uintptr_t) d_out & 3) == 0, KG_ERR_INVALID,
               "kg_wf_frames_dev: d_iq must be 8-byte and d_out 4-byte aligned");
    return wf_launch(w, nframes, chan_of, nullptr, 0, d_iq, d_out, false);
}

int kg_wf_frames_at_dev(kg_wf *w, int nframes, const int32_t *chan_of, const uint64_t *frame_off, uint64_t iq_len,
                        const void *d_iq, void *d_out)
{
    KG_REQUIRE(w && chan_of && frame_off && d_iq && d_out, KG_ERR_INVALID, "kg_wf_frames_at_dev: null argument");
    int rc = kg_ctx_use(w->ctx);
    if (rc) return rc;
    KG_REQUIRE(((uintptr_t) d_iq & 7) == 0 && ((uintptr_t) d_out & 3) == 0, KG_ERR_INVALID,
               "kg_wf_frames_at_dev: d_iq must be 8-byte and d_out 4-byte aligned");
    return wf_launch(w, nframes, chan_of, frame_off, iq_len, d_iq, d_out, false);
}

static int wf_stage(kg_wf *w, int nframes)
{
    if (nframes > w->stage_cap) {
        KG_HIP(hipStreamSynchronize(w->ctx->stream));
        (void) hipFree(w->d_iq); (void) hipFree(w->d_out);
        (void) hipFree(w->d_tap_pwr); (void) hipFree(w->d_tap_pwr_out); (void) hipFree(w->d_tap_db);
        w->d_tap_pwr = w->d_tap_pwr_out = w->d_tap_db = nullptr;
        KG_HIP(hipMalloc((void **) &w->d_iq, sizeof(short2) * WF_NFFT * (size_t) nframes));
        KG_HIP(hipMalloc((void **) &w->d_out, (size_t) WF_WIDTH * nframes));
        w->stage_cap = nframes;
    }
    return KG_OK;
}

int kg_wf_frames(kg_wf *w, int nframes, const int32_t *chan_of, const int16_t *iq, uint8_t *out)
{
    KG_REQUIRE(w && chan_of && iq && out, KG_ERR_INVALID, "kg_wf_frames: null argument");
    int rc = kg_ctx_use(w->ctx);
    if (rc) return rc;
    KG_REQUIRE(nframes >= 1, KG_ERR_INVALID, "kg_wf_frames: nframes %d", nframes);
    if ((rc = wf_stage(w, nframes)) != KG_OK) return rc;
    hipStream_t st = w->ctx->stream;
    KG_HIP(hipMemcpyAsync(w->d_iq, iq, sizeof(short2) * WF_NFFT * (size_t) nframes, hipMemcpyHostToDevice, st));
    if ((rc = wf_launch(w, nframes, chan_of, nullptr, 0, w->d_iq, w->d_out, false)) != KG_OK) return rc;
    KG_HIP(hipMemcpyAsync(out, w->d_out, (size_t) WF_WIDTH * nframes, hipMemcpyDeviceToHost, st));
    KG_HIP(hipStreamSynchronize(st));
    return KG_OK;
}

int kg_wf_debug_frame(kg_wf *w, int ch, const int16_t *iq, uint8_t *out, float *pwr, float *pwr_out, float *dB)
{
    KG_REQUIRE(w && iq && out && pwr && pwr_out && dB, KG_ERR_INVALID, "kg_wf_debug_frame: null argument");
    int rc = kg_ctx_use(w->ctx);
    if (rc) return rc;
    if ((rc = wf_stage(w, 1)) != KG_OK) return rc;
    hipStream_t st = w->ctx->stream;
    if (!w->d_tap_pwr) {
        KG_HIP(hipMalloc((void **) &w->d_tap_pwr, sizeof(float) * SUB * (size_t) w->stage_cap));
        KG_HIP(hipMalloc((void **) &w->d_tap_pwr_out, sizeof(float) * WF_WIDTH * (size_t) w->stage_cap));
        KG_HIP(hipMalloc((void **) &w->d_tap_db, sizeof(float) * WF_WIDTH * (size_t) w->stage_cap));
    }
    KG_HIP(hipMemsetAsync(w->d_tap_pwr, 0, sizeof(float) * SUB, st));
    KG_HIP(hipMemcpyAsync(w->d_iq, iq, sizeof(short2) * WF_NFFT, hipMemcpyHostToDevice, st));
    const int32_t c = ch;
    if ((rc = wf_launch(w, 1, &c, nullptr, 0, w->d_iq, w->d_out, true)) != KG_OK) return rc;
    KG_HIP(hipMemcpyAsync(out, w->d_out, WF_WIDTH, hipMemcpyDeviceToHost, st));
    KG_HIP(hipMemcpyAsync(pwr, w->d_tap_pwr, sizeof(float) * SUB, hipMemcpyDeviceToHost, st));
    KG_HIP(hipMemcpyAsync(pwr_out, w->d_tap_pwr_out, sizeof(float) * WF_WIDTH, hipMemcpyDeviceToHost, st));
    KG_HIP(hipMemcpyAsync(dB, w->d_tap_db, sizeof(float) * WF_WIDTH, hipMemcpyDeviceToHost, st));
    KG_HIP(hipStreamSynchronize(st));
    return KG_OK;
}

}  // extern "C"
