// kg_rxbank.hip -- a bank of virtual receivers stepped with ONE call (BASELINE configs[3]).
//
// In the reference every connection has a waterfall coroutine and a sound coroutine that loop over what the data pump and
// the FPGA hand them (rx/data_pump.cpp:292-341 data_pump(); rx/rx_sound.cpp:333-601 the c2s_sound() loop: in_samps ->
// CFastFIR -> S-meter / AGC / demod -> compression -> snd_pkt; rx/rx_waterfall.cpp:930-1170 c2s_waterfall() -> sample_wf()
// -> compute_frame() -> wf_pkt_t).  A bank is nrx such connections on one GPU, fed from one block of ADC samples per step:
//
//   waterfall of receiver k   IQ_MIXER + wf1 CIC (kg_ddc) in the sampler mode the reference chose for it
//                               one-shot    CmdWFReset + 8192 outputs, the non-overlapped frame (rx_waterfall.cpp:1005-1041)
//                               overlapped  continuous sampler, the frame = the ring's newest 8192 outputs read at the
//                                           synchronised write address (:967-983, CmdGetWFContSamps) -- the mode sample_wf()
//                                           switches to when a frame takes longer to fill than two display periods
//                             -> window + FFT + power + pixels + dB + u8 row (kg_wf) -> wf_pkt_t, ADPCM (kg_wf_packets_dev)
//   audio of receiver k       IQ_MIXER + rx1 / rx2 CIC + CICF (kg_rxddc) -> rx_iq_t records -> snd_service() unpack
//                             -> CFastFIR (kg_fir) -> S-meter + CAgc + detector (kg_post) -> IMA ADPCM (kg_adpcm)
//
// The per-seam objects are the bank's and are configured with their own entry points (kg_rxbank_ddc() ... kg_rxbank_adpcm());
// what the bank adds is the step: which kernel goes on which of its three streams (waterfall chain, audio chain, the two
// sequential coders), the events between them, and ONE upload per step for the small tables of all stages: every entry point
// below is called twice per step, first in the context's PLAN mode (kg_common.h, kg_arena: tables collected, nothing
// launched, no state advanced), then -- behind the one transfer -- for real.
#include "kg_common.h"

#include <chrono>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <vector>

#define BANK_SLOTS KG_RXBANK_SLOTS         // step-table slots in flight (a slot is reused BANK_SLOTS steps later)
#define BANK_RING_PAIRS (8 * KG_WF_NFFT)   // an overlapped receiver's sample ring (iq_t pairs); >= 2 frames
#define BANK_PKT_STRIDE 1056               // >= KG_WF_PKT_MAX, a multiple of 16

struct bank_move { int row, src, dst, cnt; };         // iq_t pairs inside one receiver's ring

// The ring of an overlapped receiver is linear: when the next step's outputs would run past its end, the newest
// 8192 - m pairs (m = outputs per step) move to the front and writing goes on behind them, so that a frame is always
// 8192 consecutive pairs.  One workgroup per move; source and destination never overlap (ring >= 2 frames).
__global__ __launch_bounds__(256) void rxbank_move_kernel(short2 *__restrict__ iq, long stride, const bank_move *__restrict__ tab)
{
    const bank_move m = tab[blockIdx.x];
    short2 *row = iq + (long) m.row * stride;
    for (int i = threadIdx.x; i < m.cnt; i += 256) row[m.dst + i] = row[m.src + i];
}

struct kg_rxbank {
    int device, nrx, mode, decim_rx;

    size_t n;                                   // ADC samples per step
    hipStream_t s_main, s_side, s_tail, s_up;   // waterfall chain / audio chain / both sequential coders / the table upload
    hipStream_t s_ddc2;                         // the waterfall DDC's second stream (R = 1 bypass, pass B of R <= 8 beside the rest)
    kg_ctx *c_main, *c_side, *c_tail;
    kg_ddc *ddc; kg_wf *wf; kg_rxddc *rx; kg_fir *fir; kg_post *post; kg_adpcm *adpcm;
    // device buffers
    short2 *d_wfiq; size_t wf_stride;           // [nrx][wf_stride] iq_t: a one-shot receiver uses pairs 0 .. 8191 of its row, an overlapped one all of it
    unsigned char *d_rows, *d_pkts;             // [frame][1024], [frame][BANK_PKT_STRIDE]
    unsigned char *d_raw; size_t nrec_max;      // [nrx][nrec_max] rx_iq_t (6 bytes)
    float2 *d_xin, *d_firo; size_t firo_stride; // [nrx][nrec_max], [nrx][firo_stride]
    short *d_s16; unsigned char *d_pay;         // [nrx][firo_stride], [nrx][firo_stride / 2]
    float2 *d_agc; unsigned char *d_iqpay;      // [nrx][firo_stride] the AGC's complex output, [nrx][4 firo_stride] the IQ mode's payload
    std::vector<char> little_endian;            // per receiver: s->little_endian
    std::vector<int32_t> real_list, iq_le_list, iq_be_list;
    // step tables: BANK_SLOTS pinned host / device slots
    kg_arena arena; unsigned char *h_slots, *d_slots; size_t slot_bytes;
    hipEvent_t ev_end[BANK_SLOTS][3];           // behind a step's last enqueue on main / side / tail
    bool ev_end_rec[BANK_SLOTS];
    hipEvent_t ev_tab, ev_fir, ev_tail, ev_frames, ev_pk;
    bool tail_pending, pk_pending;
    uint64_t step;
    float rescale, dc_i, dc_q; int spectral_inversion;      // snd_service() unpack, rx/data_pump.cpp:73-74,145-208
    // per receiver
    std::vector<int> decim; std::vector<char> wf_set, overlapped;
    // Connections come and go one at a time (rx/rx_sound.cpp:264-269: every c2s_sound() has its own CFastFIR position, sequence
    // number and loop): a receiver is stepped while `active`; its audio DDC / CFastFIR were reset when IT joined, so what a step
    // yields -- records, sound blocks -- is per receiver.
    std::vector<char> active;
    std::vector<uint32_t> snd_seq;              // per receiver: sound blocks emitted since it joined (the seq of its packets)
    std::vector<int32_t> last_nrec, last_nfir;  // per receiver: the last step's counts (kg_rxbank_audio_map)
    std::vector<int32_t> act, blk_list, nfir_act;   // scratch: this step's active receivers; one sound block's receivers
    std::vector<long> ring_w, ring_total;       // overlapped: next write position, outputs written since the sampler was set
    std::vector<kg_wf_pkt_info> pkt;
    // scratch of a step
    std::vector<int32_t> all, h_nrec, h_nfir, chan_of, pkt_bytes, rx_of_frame;
    std::vector<int64_t> out_off, max_out, h_nw; std::vector<uint64_t> frame_off;
    std::vector<uint8_t> enabled; std::vector<bank_move> moves; std::vector<kg_wf_pkt_info> pkt_step;
    std::vector<long> ring_w2, ring_total2;
    kg_rxbank_step_info last;
    // where the host's share of a step goes (kg_rxbank_host_profile): seconds per phase, summed over `prof_steps` steps
    double prof[16]; uint64_t prof_steps;
};

enum { PF_PLAN = 0, PF_UPLOAD, PF_RXDDC, PF_UNPACK, PF_FIR, PF_POST, PF_ADPCM, PF_DDC, PF_FRAMES, PF_PKT, PF_EVENTS, PF_END, PF_N };
static const char *const PF_NAME[PF_N] = {"plan pass", "slot wait + upload", "rxddc", "unpack", "fir", "post", "adpcm", "wf ddc", "frames", "packets",
                                          "events between the chains", "end-of-step events"};
struct bank_tick {
    std::chrono::steady_clock::time_point t;
    bank_tick() : t(std::chrono::steady_clock::now()) {}
    void lap(kg_rxbank *b, int k, bool on) {
        const auto n = std::chrono::steady_clock::now();
        if (on) b->prof[k] += std::chrono::duration<double>(n - t).count();
        t = n;
    }
};

// The streams of closed banks are kept for the next one (per device, never destroyed).  Which hardware queue a NEW stream lands
// on is the HIP runtime's choice at that moment, and a process that had created and destroyed a few banks got streams whose
// kernels took turns instead of running side by side: every later bank of the default bench run stepped in 1.37 ... 1.50 ms
// where the first one took 1.24 (and the same bank 0.98 alone).  A bank that inherits the streams of the first keeps its
// placement.
struct bank_stream_set { int device; hipStream_t s[5]; };
static std::vector<bank_stream_set> g_free_sets;
static std::mutex g_free_mutex;

static void bank_set_arena(kg_rxbank *b, int mode)
{
    b->arena.mode = mode;
    kg_arena *a = mode == KG_ARENA_OFF ? nullptr : &b->arena;
    b->c_main->arena = a; b->c_side->arena = a; b->c_tail->arena = a;
}

// One pass over the step's entry points, in the order their tables sit in the slot.
static int bank_pass(kg_rxbank *b, const void *d_adc, bool plan)
{
    const int NR = b->nrx;
    int rc;
    bank_tick tk;
    const bool pf = !plan;
    b->act.clear();
    for (int k = 0; k < NR; k++) {
        b->enabled[k] = b->active[k] ? 1 : 0;
        if (b->active[k]) b->act.push_back(k);
    }
    const int NA = (int) b->act.size();
    const int32_t *act = b->act.data();
    int nrec_max = 0, nfir_max = 0;
    if (!plan) for (int k = 0; k < NR; k++) { b->last_nrec[k] = 0; b->last_nfir[k] = 0; }
    // ---- audio chain (side stream): rx.v -> rx_iq_t -> snd_service() unpack -> CFastFIR; the tail: S-meter, CAgc, detector, ADPCM
    if (NA > 0) {
        if ((rc = kg_rxddc_push_dev(b->rx, d_adc, b->n, act, NA, b->d_raw, b->nrec_max, b->h_nrec.data()))) return rc;
        tk.lap(b, PF_RXDDC, pf);
        for (int i = 0; i < NA; i++) if (b->h_nrec[i] > nrec_max) nrec_max = b->h_nrec[i];
        // (rows are unpacked up to the largest count of the step; a receiver's CFastFIR takes its own count of them)
        if (nrec_max > 0 &&
            (rc = kg_dpump_unpack_rows_dev(b->c_side, b->d_raw, b->nrec_max, nrec_max, NR, b->enabled.data(), b->rescale, b->dc_i, b->dc_q,
                                           b->spectral_inversion, b->d_xin, b->nrec_max)))
            return rc;
        tk.lap(b, PF_UNPACK, pf);
        if (!plan && b->tail_pending) {                       // the coders of the step before have read fir_out
            KG_HIP(hipStreamWaitEvent(b->s_side, b->ev_tail, 0));
            b->tail_pending = false;
        }
        tk.lap(b, PF_EVENTS, pf);
        if ((rc = kg_fir_process_each_dev(b->fir, act, NA, b->d_xin, b->nrec_max, b->h_nrec.data(), b->d_firo, b->firo_stride, b->h_nfir.data())))
            return rc;
        tk.lap(b, PF_FIR, pf);
        for (int i = 0; i < NA; i++) if (b->h_nfir[i] > nfir_max) nfir_max = b->h_nfir[i];
    }
    if (nfir_max > 0) {
        if (!plan) {
            KG_HIP(hipEventRecord(b->ev_fir, b->s_side));
            KG_HIP(hipStreamWaitEvent(b->s_tail, b->ev_fir, 0));
        }
        tk.lap(b, PF_EVENTS, pf);
        for (int blk = 0; blk < nfir_max / KG_FIR_OUT; blk++) {         // one sound packet per 512 samples (rx_sound.cpp:601-1170)
            const size_t o = (size_t) blk * KG_FIR_OUT;
            b->blk_list.clear();                                          // the receivers that completed this block in this step
            for (int i = 0; i < NA; i++) if (b->h_nfir[i] >= (int) (o + KG_FIR_OUT)) b->blk_list.push_back(act[i]);
            const int NB = (int) b->blk_list.size();
            if ((rc = kg_post_process_dev(b->post, b->blk_list.data(), NB, b->d_firo + o, b->firo_stride, KG_FIR_OUT, b->d_s16 + o, nullptr,
                                          b->d_agc + o, b->firo_stride)))
                return rc;
            tk.lap(b, PF_POST, pf);
            // the real modes' blocks go through the ADPCM coder, the IQ mode's out as (s2_t) pairs (rx_sound.cpp:1076-1140)
            b->real_list.clear(); b->iq_le_list.clear(); b->iq_be_list.clear();
            for (int i = 0; i < NB; i++) {
                const int k = b->blk_list[i];
                if (kg_post_get_mode(b->post, k) != KG_POST_IQ) b->real_list.push_back(k);
                else (b->little_endian[k] ? b->iq_le_list : b->iq_be_list).push_back(k);
            }
            if (!b->real_list.empty() &&
                (rc = kg_adpcm_encode_dev(b->adpcm, b->real_list.data(), (int) b->real_list.size(), b->d_s16 + o, b->firo_stride, KG_FIR_OUT,
                                          b->d_pay + o / 2, b->firo_stride / 2)))
                return rc;
            for (int le = 0; le < 2; le++) {
                std::vector<int32_t> &l = le ? b->iq_le_list : b->iq_be_list;
                if (!l.empty() &&
                    (rc = kg_snd_iq_payload_dev(b->c_tail, l.data(), (int) l.size(), b->d_agc + o, b->firo_stride, KG_FIR_OUT, le,
                                                b->d_iqpay + 4 * o, 4 * b->firo_stride)))
                    return rc;
            }
            tk.lap(b, PF_ADPCM, pf);
        }
        if (!plan) {
            KG_HIP(hipEventRecord(b->ev_tail, b->s_tail));
            b->tail_pending = true;
        }
        tk.lap(b, PF_EVENTS, pf);
    }
    // ---- waterfall chain (main stream)
    int nframes = 0;
    b->moves.clear();
    for (int i = 0; i < NA; i++) {
        const int k = act[i];
        KG_REQUIRE(b->wf_set[k], KG_ERR_STATE, "kg_rxbank_step: kg_rxbank_set_wf was not called for receiver %d (since it joined)", k);
        if (b->overlapped[k]) {
            const long m = (long) (b->n / (size_t) b->decim[k]);
            long w = b->ring_w[k];
            if (w + m > BANK_RING_PAIRS) {
                const long keep = KG_WF_NFFT - m;
                if (keep > 0) b->moves.push_back(bank_move{k, (int) (w - keep), 0, (int) keep});
                w = keep;
            }
            b->out_off[i] = w; b->max_out[i] = 0;
            b->ring_w2[k] = w + m; b->ring_total2[k] = b->ring_total[k] + m;
            if (b->ring_total2[k] >= KG_WF_NFFT) {                       // (before that: the reference sleeps, "fill pipe", :978)
                b->chan_of[nframes] = k;
                b->frame_off[nframes] = (uint64_t) k * b->wf_stride + (uint64_t) (w + m - KG_WF_NFFT);
                nframes++;
            }
        } else {
            b->out_off[i] = 0; b->max_out[i] = KG_WF_NFFT;
            b->chan_of[nframes] = k;
            b->frame_off[nframes] = (uint64_t) k * b->wf_stride;
            nframes++;
        }
    }
    if (!b->moves.empty()) {
        void *d_mv = nullptr;
        if ((rc = kg_ctx_stage(b->c_main, b->moves.data(), sizeof(bank_move) * b->moves.size(), &d_mv))) return rc;
        if (!plan) {
            hipLaunchKernelGGL(rxbank_move_kernel, dim3((unsigned) b->moves.size()), dim3(256), 0, b->s_main, b->d_wfiq, (long) b->wf_stride,
                               (const bank_move *) d_mv);
            KG_HIP(hipGetLastError());
        }
    }
    if (NA > 0 &&
        (rc = kg_ddc_wf_step_dev(b->ddc, d_adc, b->n, act, NA, b->d_wfiq, b->wf_stride, b->out_off.data(), b->max_out.data(),
                                 b->h_nw.data())))
        return rc;
    tk.lap(b, PF_DDC, pf);
    if (!plan && b->pk_pending) {                         // the packets of the step before have read the rows
        KG_HIP(hipStreamWaitEvent(b->s_main, b->ev_pk, 0));
        b->pk_pending = false;
    }
    tk.lap(b, PF_EVENTS, pf);
    if (nframes > 0) {
        if ((rc = kg_wf_frames_at_dev(b->wf, nframes, b->chan_of.data(), b->frame_off.data(), (uint64_t) NR * b->wf_stride, b->d_wfiq,
                                      b->d_rows)))
            return rc;
        tk.lap(b, PF_FRAMES, pf);
        if (!plan) {
            KG_HIP(hipEventRecord(b->ev_frames, b->s_main));
            KG_HIP(hipStreamWaitEvent(b->s_tail, b->ev_frames, 0));
        }
        tk.lap(b, PF_EVENTS, pf);
        for (int f = 0; f < nframes; f++) {
            b->pkt_step[f] = b->pkt[b->chan_of[f]];
            b->pkt_step[f].seq = b->snd_seq[b->chan_of[f]];              // out->seq = wf->snd_seq, rx_waterfall.cpp:1635
        }
        if ((rc = kg_wf_packets_dev(b->c_tail, b->d_rows, KG_WF_WIDTH, nframes, b->pkt_step.data(), b->d_pkts, BANK_PKT_STRIDE,
                                    b->pkt_bytes.data())))
            return rc;
        tk.lap(b, PF_PKT, pf);
        if (!plan) {
            KG_HIP(hipEventRecord(b->ev_pk, b->s_tail));
            b->pk_pending = true;
        }
        tk.lap(b, PF_EVENTS, pf);
    }
    if (!plan) {
        for (int i = 0; i < NA; i++) {
            const int k = act[i];
            if (b->overlapped[k]) { b->ring_w[k] = b->ring_w2[k]; b->ring_total[k] = b->ring_total2[k]; }
        }
        for (int f = 0; f < nframes; f++) b->rx_of_frame[f] = b->chan_of[f];
        kg_rxbank_step_info &s = b->last;
        const int first = NA > 0 ? act[0] : 0;             // the step's scalar audio fields: the lowest-numbered active receiver's
        s.step = b->step; s.nframes = nframes; s.nrec = NA > 0 ? b->h_nrec[0] : 0; s.nfir = NA > 0 ? b->h_nfir[0] : 0;
        s.fir_pos = kg_fir_pos(b->fir, first);
        s.snd_seq = b->snd_seq[first]; s.table_bytes = (int32_t) b->arena.used; s.nmoves = (int32_t) b->moves.size();
        for (int i = 0; i < NA; i++) {
            b->last_nrec[act[i]] = b->h_nrec[i]; b->last_nfir[act[i]] = b->h_nfir[i];
            b->snd_seq[act[i]] += (uint32_t) (b->h_nfir[i] / KG_FIR_OUT);
        }
    }
    return KG_OK;
}

extern "C" {

void kg_rxbank_destroy(kg_rxbank *b)
{
    if (!b) return;
    (void) hipSetDevice(b->device);
    for (hipStream_t s : {b->s_main, b->s_side, b->s_tail, b->s_ddc2, b->s_up}) if (s) (void) hipStreamSynchronize(s);
    if (b->c_main && b->c_side && b->c_tail) bank_set_arena(b, KG_ARENA_OFF);
    kg_adpcm_destroy(b->adpcm); kg_post_destroy(b->post); kg_fir_destroy(b->fir); kg_rxddc_destroy(b->rx);
    kg_wf_destroy(b->wf); kg_ddc_destroy(b->ddc);
    kg_ctx_destroy(b->c_tail); kg_ctx_destroy(b->c_side); kg_ctx_destroy(b->c_main);
    (void) hipFree(b->d_wfiq); (void) hipFree(b->d_rows); (void) hipFree(b->d_pkts); (void) hipFree(b->d_raw);
    (void) hipFree(b->d_xin); (void) hipFree(b->d_firo); (void) hipFree(b->d_s16); (void) hipFree(b->d_pay);
    (void) hipFree(b->d_agc); (void) hipFree(b->d_iqpay);
    if (b->h_slots) (void) hipHostFree(b->h_slots);
    (void) hipFree(b->d_slots);
    for (int i = 0; i < BANK_SLOTS; i++) for (int j = 0; j < 3; j++) if (b->ev_end[i][j]) (void) hipEventDestroy(b->ev_end[i][j]);
    for (hipEvent_t e : {b->ev_tab, b->ev_fir, b->ev_tail, b->ev_frames, b->ev_pk}) if (e) (void) hipEventDestroy(e);
    if (b->s_main && b->s_side && b->s_tail && b->s_ddc2 && b->s_up) {          // a complete set: kept for the next bank
        std::lock_guard<std::mutex> lk(g_free_mutex);
        g_free_sets.push_back(bank_stream_set{b->device, {b->s_main, b->s_side, b->s_tail, b->s_ddc2, b->s_up}});
    } else {
        for (hipStream_t s : {b->s_main, b->s_side, b->s_tail, b->s_ddc2, b->s_up}) if (s) (void) hipStreamDestroy(s);
    }
    delete b;
}

int kg_rxbank_create(int device, int nrx, size_t adc_samples_per_step, int rx_mode, kg_rxbank **out)
{
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_rxbank_create: out is null");
    *out = nullptr;
    KG_REQUIRE(nrx >= 1 && nrx <= 4096, KG_ERR_INVALID, "kg_rxbank_create: nrx %d (1..4096)", nrx);
    KG_REQUIRE(adc_samples_per_step >= 8192 && adc_samples_per_step <= ((size_t) 1 << 27), KG_ERR_INVALID,
               "kg_rxbank_create: %zu ADC samples per step (8192 .. 2^27)", adc_samples_per_step);
    kg_rxbank *b = new (std::nothrow) kg_rxbank();
    KG_REQUIRE(b != nullptr, KG_ERR_NOMEM, "kg_rxbank_create: alloc");
    b->device = device; b->nrx = nrx; b->mode = rx_mode; b->n = adc_samples_per_step;
    int rc = KG_OK;
    {   // (what says "no gfx950 device": there is no CPU fallback)
        int ndev = 0;
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) {
            kg_set_error("kg_rxbank_create: no HIP device (%s); libkiwigpu has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
            delete b;
            return KG_ERR_NO_DEVICE;
        }
        if (device < 0 || device >= ndev) { kg_set_error("kg_rxbank_create: device %d out of range (0..%d)", device, ndev - 1); delete b; return KG_ERR_INVALID; }
    }
#define BANK_TRY(call) do { if ((rc = (call)) != KG_OK) { kg_rxbank_destroy(b); return rc; } } while (0)
#define BANK_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { kg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
                            hipGetErrorString(e_)); kg_rxbank_destroy(b); return KG_ERR_HIP; } } while (0)
    // The four streams that carry kernels are created back to back, the upload's fifth.  Which hardware queue (and, behind
    // it, which of the four dispatch pipes) a stream lands on is the HIP runtime's choice -- the least-shared queue of its pool
    // at that moment -- and depends on every stream the process created before: measured on one bank, 0.98 ms per step or
    // 1.25 ... 1.38 by what had run earlier in the process (tools/micro/stream_pairs.hip measures which pairs of
    // streams run side by side).  Creating them together at least keeps them off one another's queues while the pool has
    // free ones.
    BANK_HIP(hipSetDevice(device));
    {
        hipDeviceProp_t prop;
        BANK_HIP(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            kg_set_error("kg_rxbank_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
            delete b;
            return KG_ERR_NO_DEVICE;
        }
    }
    bool reused = false;
    {
        std::lock_guard<std::mutex> lk(g_free_mutex);
        for (size_t i = 0; i < g_free_sets.size(); i++)
            if (g_free_sets[i].device == device) {
                const bank_stream_set st = g_free_sets[i];
                g_free_sets.erase(g_free_sets.begin() + (long) i);
                b->s_main = st.s[0]; b->s_side = st.s[1]; b->s_tail = st.s[2]; b->s_ddc2 = st.s[3]; b->s_up = st.s[4];
                reused = true;
                break;
            }
    }
    if (!reused) {
    BANK_HIP(hipStreamCreateWithFlags(&b->s_main, hipStreamNonBlocking));
    BANK_HIP(hipStreamCreateWithFlags(&b->s_side, hipStreamNonBlocking));
    BANK_HIP(hipStreamCreateWithFlags(&b->s_tail, hipStreamNonBlocking));
    BANK_HIP(hipStreamCreateWithFlags(&b->s_ddc2, hipStreamNonBlocking));
    BANK_HIP(hipStreamCreateWithFlags(&b->s_up, hipStreamNonBlocking));
    }
    BANK_TRY(kg_ctx_create_on_stream(device, b->s_main, &b->c_main));
    BANK_TRY(kg_ctx_create_on_stream(device, b->s_side, &b->c_side));
    BANK_TRY(kg_ctx_create_on_stream(device, b->s_tail, &b->c_tail));
    BANK_TRY(kg_ddc_create(b->c_main, nrx, b->n, &b->ddc));
    BANK_TRY(kg_ddc_use_side_stream(b->ddc, b->s_ddc2));
    BANK_TRY(kg_wf_create(b->c_main, nrx, &b->wf));
    BANK_TRY(kg_rxddc_create_mode(b->c_side, nrx, b->n, rx_mode, &b->rx));
    b->decim_rx = kg_rxddc_decim(b->rx);
    b->nrec_max = b->n / (size_t) b->decim_rx + 2;
    BANK_TRY(kg_fir_create(b->c_side, nrx, (int) b->nrec_max, &b->fir));
    BANK_TRY(kg_post_create(b->c_tail, nrx, &b->post));
    BANK_TRY(kg_adpcm_create(b->c_tail, nrx, &b->adpcm));
    b->wf_stride = BANK_RING_PAIRS;
    b->firo_stride = ((b->nrec_max + KG_FIR_OUT - 1) / KG_FIR_OUT + 1) * KG_FIR_OUT;
    BANK_HIP(hipMalloc((void **) &b->d_wfiq, sizeof(short2) * b->wf_stride * nrx));
    BANK_HIP(hipMemset(b->d_wfiq, 0, sizeof(short2) * b->wf_stride * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_rows, (size_t) KG_WF_WIDTH * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_pkts, (size_t) BANK_PKT_STRIDE * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_raw, 6 * b->nrec_max * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_xin, sizeof(float2) * b->nrec_max * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_firo, sizeof(float2) * b->firo_stride * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_s16, sizeof(short) * b->firo_stride * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_pay, b->firo_stride / 2 * nrx));
    BANK_HIP(hipMemset(b->d_rows, 0, (size_t) KG_WF_WIDTH * nrx));
    BANK_HIP(hipMemset(b->d_pkts, 0, (size_t) BANK_PKT_STRIDE * nrx));
    BANK_HIP(hipMemset(b->d_s16, 0, sizeof(short) * b->firo_stride * nrx));
    BANK_HIP(hipMemset(b->d_pay, 0, b->firo_stride / 2 * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_agc, sizeof(float2) * b->firo_stride * nrx));
    BANK_HIP(hipMalloc((void **) &b->d_iqpay, 4 * b->firo_stride * nrx));
    BANK_HIP(hipMemset(b->d_agc, 0, sizeof(float2) * b->firo_stride * nrx));
    BANK_HIP(hipMemset(b->d_iqpay, 0, 4 * b->firo_stride * nrx));
    b->little_endian.assign(nrx, 0);
    {   // what a step's tables can take: <= 12 of them + a channel list for S-meter / AGC / detector and one for the coder per sound
        // block (a receiver can complete nrec_max / 512 + 1 blocks in a step) -- checked HERE, not found out by every later step
        const size_t blocks_max = b->nrec_max / KG_FIR_OUT + 1;
        if (12 + 4 * blocks_max > KG_ARENA_MAX_ENTRIES) {
            kg_set_error("kg_rxbank_create: %zu ADC samples per step are up to %zu sound blocks per receiver and step; the step table holds %d "
                         "(a step of at most %zu samples)", b->n, blocks_max, (KG_ARENA_MAX_ENTRIES - 12) / 4,
                         (size_t) ((KG_ARENA_MAX_ENTRIES - 12) / 4 - 1) * KG_FIR_OUT * (size_t) b->decim_rx);
            kg_rxbank_destroy(b);
            return KG_ERR_INVALID;
        }
        b->slot_bytes = (8192 + (size_t) (384 + 12 * blocks_max) * nrx + 63) & ~(size_t) 63;
    }
    BANK_HIP(hipHostMalloc((void **) &b->h_slots, b->slot_bytes * BANK_SLOTS, hipHostMallocDefault));
    BANK_HIP(hipMalloc((void **) &b->d_slots, b->slot_bytes * BANK_SLOTS));
    for (int i = 0; i < BANK_SLOTS; i++) {
        // hipEventBlockingSync: a host that has run BANK_SLOTS steps ahead SLEEPS in kg_rxbank_step's slot wait instead of spinning
        // (the reference's server is one cooperative thread: it asks kg_rxbank_ready first and yields instead)
        for (int j = 0; j < 3; j++) BANK_HIP(hipEventCreateWithFlags(&b->ev_end[i][j], hipEventDisableTiming | hipEventBlockingSync));
        b->ev_end_rec[i] = false;
    }
    for (hipEvent_t *e : {&b->ev_tab, &b->ev_fir, &b->ev_tail, &b->ev_frames, &b->ev_pk})
        BANK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    // rescale = MPOW(2, -RXOUT_SCALE + CUTESDR_SCALE) * MPOW(10, CICF_GAIN_dB / 20), rx/data_pump.cpp:73-74 (kg_rxbank_set_unpack overrides)
    b->rescale = powf(2.f, -23 + 15) * powf(10.f, 4.5f / 20.f); b->dc_i = b->dc_q = 0.f; b->spectral_inversion = 0;
    b->tail_pending = b->pk_pending = false; b->step = 0;
    b->active.assign(nrx, 1); b->snd_seq.assign(nrx, 0); b->last_nrec.assign(nrx, 0); b->last_nfir.assign(nrx, 0);
    b->c_main->rows_by_chan = b->c_side->rows_by_chan = b->c_tail->rows_by_chan = 1;      // buffers hold one row per RECEIVER
    b->decim.assign(nrx, 0); b->wf_set.assign(nrx, 0); b->overlapped.assign(nrx, 0);
    b->ring_w.assign(nrx, 0); b->ring_total.assign(nrx, 0); b->ring_w2.assign(nrx, 0); b->ring_total2.assign(nrx, 0);
    b->pkt.assign(nrx, kg_wf_pkt_info{0, 0, 0, 1});
    b->all.resize(nrx); for (int i = 0; i < nrx; i++) b->all[i] = i;
    b->h_nrec.assign(nrx, 0); b->h_nfir.assign(nrx, 0); b->chan_of.assign(nrx, 0); b->pkt_bytes.assign(nrx, 0); b->rx_of_frame.assign(nrx, -1);
    b->out_off.assign(nrx, 0); b->max_out.assign(nrx, 0); b->h_nw.assign(nrx, 0); b->frame_off.assign(nrx, 0);
    b->enabled.assign(nrx, 1); b->pkt_step.resize(nrx);
    memset(&b->arena, 0, sizeof b->arena);
    memset(&b->last, 0, sizeof b->last);
    BANK_HIP(hipDeviceSynchronize());
    *out = b;
    return KG_OK;
#undef BANK_TRY
#undef BANK_HIP
}

kg_ddc *kg_rxbank_ddc(kg_rxbank *b) { return b ? b->ddc : nullptr; }
kg_wf *kg_rxbank_wf(kg_rxbank *b) { return b ? b->wf : nullptr; }
kg_rxddc *kg_rxbank_rxddc(kg_rxbank *b) { return b ? b->rx : nullptr; }
kg_fir *kg_rxbank_fir(kg_rxbank *b) { return b ? b->fir : nullptr; }
kg_post *kg_rxbank_post(kg_rxbank *b) { return b ? b->post : nullptr; }
kg_adpcm *kg_rxbank_adpcm(kg_rxbank *b) { return b ? b->adpcm : nullptr; }
kg_ctx *kg_rxbank_ctx(kg_rxbank *b) { return b ? b->c_main : nullptr; }

int kg_rxbank_set_wf(kg_rxbank *b, int rx, uint64_t phase_inc, int decim, int overlapped)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_set_wf: null argument");
    KG_REQUIRE(rx >= 0 && rx < b->nrx, KG_ERR_INVALID, "kg_rxbank_set_wf: receiver %d (0..%d)", rx, b->nrx - 1);
    KG_REQUIRE(decim >= 1 && (decim & (decim - 1)) == 0 && decim <= 8192, KG_ERR_INVALID, "kg_rxbank_set_wf: decimation %d", decim);
    if (overlapped) {
        const size_t m = b->n / (size_t) decim;
        // (m even: a frame is read at ring offset w + m - 8192, and kg_wf_frames_at_dev takes 8-byte aligned frames)
        KG_REQUIRE(b->n % (size_t) decim == 0 && m >= 2 && m <= KG_WF_NFFT && KG_WF_NFFT % m == 0, KG_ERR_INVALID,
                   "kg_rxbank_set_wf: overlapped sampling needs a step (%zu samples) that yields an even divisor of 8192 outputs at R = %d "
                   "(a faster sampler fills a frame per step: use the one-shot mode)", b->n, decim);
    } else {
        KG_REQUIRE((size_t) KG_WF_NFFT * (size_t) decim <= b->n, KG_ERR_INVALID,
                   "kg_rxbank_set_wf: a one-shot frame at R = %d takes %zu samples, a step has %zu: use the overlapped mode "
                   "(rx_waterfall.cpp:967-983)", decim, (size_t) KG_WF_NFFT * (size_t) decim, b->n);
    }
    for (hipStream_t s : {b->s_main, b->s_side, b->s_tail}) KG_HIP(hipStreamSynchronize(s));
    int rc = kg_ddc_set_wf(b->ddc, rx, phase_inc, decim);
    if (rc) return rc;
    b->decim[rx] = decim; b->overlapped[rx] = overlapped ? 1 : 0; b->wf_set[rx] = 1;
    b->ring_w[rx] = 0; b->ring_total[rx] = 0;             // CmdWFReset: the sampler starts empty ("fill pipe")
    return KG_OK;
}

// A connection ends: the receiver is left out of every stage from the next step on (its state stays where it is).
int kg_rxbank_leave(kg_rxbank *b, int rx)
{
    KG_REQUIRE(b && rx >= 0 && rx < b->nrx, KG_ERR_INVALID, "kg_rxbank_leave: receiver %d", rx);
    b->active[rx] = 0;
    return KG_OK;
}

// A connection starts on receiver rx (rx/rx_sound.cpp:236-269, rx/rx_waterfall.cpp:205-330): its audio DDC, CFastFIR position,
// S-meter / detector state, ADPCM state and sound sequence number start from zero, its waterfall sampler is empty and waits for
// kg_rxbank_set_wf; no other receiver is touched (their DDC counters, FIR positions and sequence numbers run on -- from here
// on this receiver's sound blocks complete on its OWN steps).  The host then configures it through the per-seam objects
// (kg_rxddc_set_freq, kg_fir_setup, kg_post_*) as for a fresh bank.  Drains the bank's streams.
int kg_rxbank_join(kg_rxbank *b, int rx)
{
    KG_REQUIRE(b && rx >= 0 && rx < b->nrx, KG_ERR_INVALID, "kg_rxbank_join: receiver %d", rx);
    KG_HIP(hipSetDevice(b->device));
    for (hipStream_t s : {b->s_up, b->s_main, b->s_side, b->s_tail}) KG_HIP(hipStreamSynchronize(s));
    int rc;
    if ((rc = kg_rxddc_reset(b->rx, rx))) return rc;
    if ((rc = kg_fir_reset(b->fir, rx))) return rc;
    if ((rc = kg_post_reset(b->post, rx))) return rc;
    if ((rc = kg_adpcm_set_state(b->adpcm, rx, 0, 0))) return rc;
    for (hipStream_t s : {b->s_side, b->s_tail}) KG_HIP(hipStreamSynchronize(s));
    b->snd_seq[rx] = 0; b->last_nrec[rx] = 0; b->last_nfir[rx] = 0;
    b->wf_set[rx] = 0; b->ring_w[rx] = 0; b->ring_total[rx] = 0;
    b->active[rx] = 1;
    return KG_OK;
}

int kg_rxbank_is_active(kg_rxbank *b, int rx)
{
    KG_REQUIRE(b && rx >= 0 && rx < b->nrx, KG_ERR_INVALID, "kg_rxbank_is_active: receiver %d", rx);
    return b->active[rx] ? 1 : 0;
}

// Per receiver, after the last step (arrays of nrx entries, any may be NULL): rx_iq_t records and CFastFIR outputs the step gave
// it (0 for a receiver that is not active), FirPos() now, sound blocks it has emitted since it joined (= the seq its next
// packets carry).
int kg_rxbank_audio_map(kg_rxbank *b, int32_t *nrec, int32_t *nfir, int32_t *fir_pos, uint32_t *snd_seq)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_audio_map: null argument");
    for (int k = 0; k < b->nrx; k++) {
        if (nrec) nrec[k] = b->last_nrec[k];
        if (nfir) nfir[k] = b->last_nfir[k];
        if (fir_pos) fir_pos[k] = kg_fir_pos(b->fir, k);
        if (snd_seq) snd_seq[k] = b->snd_seq[k];
    }
    return KG_OK;
}

int kg_rxbank_set_little_endian(kg_rxbank *b, int rx, int little_endian)
{
    KG_REQUIRE(b && rx >= 0 && rx < b->nrx, KG_ERR_INVALID, "kg_rxbank_set_little_endian: receiver %d", rx);
    b->little_endian[rx] = little_endian ? 1 : 0;
    return KG_OK;
}

int kg_rxbank_set_unpack(kg_rxbank *b, float rescale, float dc_i, float dc_q, int spectral_inversion)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_set_unpack: null argument");
    b->rescale = rescale; b->dc_i = dc_i; b->dc_q = dc_q; b->spectral_inversion = spectral_inversion ? 1 : 0;
    return KG_OK;
}

int kg_rxbank_set_wf_pkt(kg_rxbank *b, int rx, uint32_t x_bin_server, uint32_t zoom, int use_compression)
{
    KG_REQUIRE(b && rx >= 0 && rx < b->nrx, KG_ERR_INVALID, "kg_rxbank_set_wf_pkt: receiver %d", rx);
    b->pkt[rx].x_bin_server = x_bin_server; b->pkt[rx].zoom = zoom; b->pkt[rx].use_compression = use_compression ? 1 : 0;
    return KG_OK;
}

int kg_rxbank_step(kg_rxbank *b, const void *d_adc, void *adc_ready_event, kg_rxbank_step_info *info)
{
    KG_REQUIRE(b && d_adc, KG_ERR_INVALID, "kg_rxbank_step: null argument");
    KG_HIP(hipSetDevice(b->device));
    bank_tick tk;
    const int slot = (int) (b->step % BANK_SLOTS);
    if (b->ev_end_rec[slot])                               // the step that last used this slot (BANK_SLOTS steps ago) has run
        for (int j = 0; j < 3; j++) KG_HIP(hipEventSynchronize(b->ev_end[slot][j]));
    kg_arena &a = b->arena;
    a.h_base = b->h_slots + (size_t) slot * b->slot_bytes; a.d_base = b->d_slots + (size_t) slot * b->slot_bytes;
    a.cap = b->slot_bytes; a.used = 0; a.nent = 0; a.cursor = 0;
    tk.lap(b, PF_UPLOAD, true);
    bank_set_arena(b, KG_ARENA_PLAN);
    int rc = bank_pass(b, d_adc, true);
    tk.lap(b, PF_PLAN, true);
    if (rc == KG_OK) {
        // the ONE transfer of the step, on a stream of its own (ordered behind nothing but earlier uploads: the slot is free)
        hipStream_t us = b->s_up;
        if (const char *ev = kg_tuning_env("KIWIGPU_BANK_UPLOAD")) us = ev[0] == 'm' ? b->s_main : (ev[0] == 's' ? b->s_side : b->s_up);
        hipError_t e = hipMemcpyAsync(a.d_base, a.h_base, a.used, hipMemcpyHostToDevice, us);
        if (e == hipSuccess) e = hipEventRecord(b->ev_tab, us);
        if (e == hipSuccess && us != b->s_main) e = hipStreamWaitEvent(b->s_main, b->ev_tab, 0);
        if (e == hipSuccess && us != b->s_side) e = hipStreamWaitEvent(b->s_side, b->ev_tab, 0);
        if (e == hipSuccess && adc_ready_event) {
            e = hipStreamWaitEvent(b->s_main, (hipEvent_t) adc_ready_event, 0);
            if (e == hipSuccess) e = hipStreamWaitEvent(b->s_side, (hipEvent_t) adc_ready_event, 0);
        }
        if (e != hipSuccess) { kg_set_error("kg_rxbank_step: %s", hipGetErrorString(e)); rc = KG_ERR_HIP; }
    }
    tk.lap(b, PF_UPLOAD, true);
    if (rc == KG_OK) {
        a.mode = KG_ARENA_REPLAY;
        rc = bank_pass(b, d_adc, false);
        if (rc == KG_OK && a.cursor != a.nent) { kg_set_error("kg_rxbank_step: %d of %d planned tables replayed", a.cursor, a.nent); rc = KG_ERR_STATE; }
    }
    bank_set_arena(b, KG_ARENA_OFF);
    if (rc) return rc;
    tk = bank_tick();
    KG_HIP(hipEventRecord(b->ev_end[slot][0], b->s_main));
    KG_HIP(hipEventRecord(b->ev_end[slot][1], b->s_side));
    KG_HIP(hipEventRecord(b->ev_end[slot][2], b->s_tail));
    b->ev_end_rec[slot] = true;
    tk.lap(b, PF_END, true);
    b->prof_steps++;
    b->step++;
    if (info) *info = b->last;
    return KG_OK;
}

// `stream` waits until the readers (both DDCs) of the ADC block of the step `steps_back` steps ago are done -- 1: the last
// step, 2: the one before it (the buffer a double-buffered ADC ring is about to refill), ... up to BANK_SLOTS.  The caller's
// writer of that buffer goes behind this.
int kg_rxbank_adc_done(kg_rxbank *b, void *stream, int steps_back)
{
    KG_REQUIRE(b && stream, KG_ERR_INVALID, "kg_rxbank_adc_done: null argument");
    KG_REQUIRE(steps_back >= 1 && steps_back <= BANK_SLOTS, KG_ERR_INVALID, "kg_rxbank_adc_done: steps_back %d (1..%d)", steps_back, BANK_SLOTS);
    if (b->step < (uint64_t) steps_back) return KG_OK;      // no such step yet: nothing has read the buffer
    const int slot = (int) ((b->step - (uint64_t) steps_back) % BANK_SLOTS);
    KG_HIP(hipStreamWaitEvent((hipStream_t) stream, b->ev_end[slot][0], 0));
    KG_HIP(hipStreamWaitEvent((hipStream_t) stream, b->ev_end[slot][1], 0));
    return KG_OK;
}

// 1: the next kg_rxbank_step will not wait (the step that last used its table slot, BANK_SLOTS steps ago, has run); 0: it would
// block until then -- a cooperative host (the reference's coroutine server) yields and asks again (NextTask in the data pump).
int kg_rxbank_ready(kg_rxbank *b)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_ready: null argument");
    KG_HIP(hipSetDevice(b->device));
    const int slot = (int) (b->step % BANK_SLOTS);
    if (!b->ev_end_rec[slot]) return 1;
    for (int j = 0; j < 3; j++) {
        const hipError_t e = hipEventQuery(b->ev_end[slot][j]);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) { kg_set_error("kg_rxbank_ready: %s", hipGetErrorString(e)); return KG_ERR_HIP; }
    }
    return 1;
}

int kg_rxbank_poll(kg_rxbank *b)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_poll: null argument");
    KG_HIP(hipSetDevice(b->device));
    for (hipStream_t s : {b->s_main, b->s_side, b->s_tail}) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) { kg_set_error("kg_rxbank_poll: %s", hipGetErrorString(e)); return KG_ERR_HIP; }
    }
    return 1;
}

int kg_rxbank_sync(kg_rxbank *b)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_sync: null argument");
    KG_HIP(hipSetDevice(b->device));
    for (hipStream_t s : {b->s_up, b->s_main, b->s_side, b->s_tail}) KG_HIP(hipStreamSynchronize(s));
    return KG_OK;
}

// The host's share of the steps since the last call (or since create), by phase: text into buf.  The slot wait is where a
// host that runs ahead of the GPU blocks (BANK_SLOTS steps deep), so measure with the bank drained between steps.
int kg_rxbank_host_profile(kg_rxbank *b, char *buf, size_t len)
{
    KG_REQUIRE(b && buf && len > 0, KG_ERR_INVALID, "kg_rxbank_host_profile: null argument");
    size_t at = 0;
    double tot = 0;
    const double n = b->prof_steps ? (double) b->prof_steps : 1.0;
    for (int k = 0; k < PF_N; k++) tot += b->prof[k];
    at += (size_t) snprintf(buf + at, len - at, "%llu steps, %.1f us of host time per step:", (unsigned long long) b->prof_steps, tot / n * 1e6);
    for (int k = 0; k < PF_N && at < len; k++) at += (size_t) snprintf(buf + at, len - at, " %s %.1f;", PF_NAME[k], b->prof[k] / n * 1e6);
    for (int k = 0; k < PF_N; k++) b->prof[k] = 0;
    b->prof_steps = 0;
    return KG_OK;
}

int kg_rxbank_buffers(kg_rxbank *b, kg_rxbank_bufs *out)
{
    KG_REQUIRE(b && out, KG_ERR_INVALID, "kg_rxbank_buffers: null argument");
    out->wf_iq = b->d_wfiq; out->wf_iq_stride = b->wf_stride;
    out->wf_rows = b->d_rows; out->wf_pkts = b->d_pkts; out->wf_pkt_stride = BANK_PKT_STRIDE;
    out->rx_raw = b->d_raw; out->rx_stride = b->nrec_max;
    out->rx_in = b->d_xin;
    out->fir_out = b->d_firo; out->fir_stride = b->firo_stride;
    out->s16 = b->d_s16; out->adpcm = b->d_pay; out->agc = b->d_agc; out->iq_pay = b->d_iqpay;
    return KG_OK;
}

int kg_rxbank_frame_map(kg_rxbank *b, int32_t *rx_of_frame, uint64_t *frame_off, int32_t *pkt_bytes)
{
    KG_REQUIRE(b != nullptr, KG_ERR_INVALID, "kg_rxbank_frame_map: null argument");
    for (int f = 0; f < b->last.nframes; f++) {
        if (rx_of_frame) rx_of_frame[f] = b->rx_of_frame[f];
        if (frame_off) frame_off[f] = b->frame_off[f];
        if (pkt_bytes) pkt_bytes[f] = b->pkt_bytes[f];
    }
    return b->last.nframes;
}

}  // extern "C"
