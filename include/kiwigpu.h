/*
 * kiwigpu.h -- C ABI of libkiwigpu.so: the MI355X (gfx950) implementation of
 * the FlyDog_SDR_GPS DSP hot path.
 *
 * The reference has no FFI seam for this path; the seam is function-level
 * inside the kiwid process.  Every entry point below
 * names the reference function(s) it replaces (file:line under the reference
 * tree).  INTEGRATION.md shows the binding a maintainer adds on the reference
 * side.
 *
 * Conventions
 *   - plain C types only; complex data is interleaved float (re, im), the
 *     layout of fftwf_complex / TYPECPX in the reference;
 *   - every function returns KG_OK (0) or a negative kg_status; nothing throws;
 *     kg_last_error() gives the text of the last failure on this thread;
 *   - "_dev" variants take pointers to device (HBM) memory and only enqueue
 *     work on the context's stream; the others take host memory, copy, and
 *     (where they return results) synchronise;
 *   - the library owns all device memory it allocates; callers keep ownership
 *     of every buffer they pass in;
 *   - there is NO CPU fallback: without a usable gfx950 device kg_ctx_create
 *     fails with KG_ERR_NO_DEVICE.
 *   - the reference runs this path from cooperative coroutines on one host
 *     thread (support/coroutines.cpp); the async entry points plus kg_ctx_poll()
 *     are meant to be called around its NextTask() yield points.
 */
#ifndef KIWIGPU_H
#define KIWIGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KG_ABI_VERSION 4     /* 4 (round 6): kg_post_cfir_* / _squelch_* / _set_deemp (AM, NBFM reach d_s16), kg_rxbank_join / _leave /
                              * _audio_map / _ready, kg_fir_process_each_dev; 3 (round 5): kg_rxbank_*, kg_ddc_wf_step_dev; 2 (round 4): kg_wf_frames_at_dev takes the extent of d_iq; kg_ctx_mark */

typedef enum {
    KG_OK = 0,
    KG_ERR_NO_DEVICE = -1,   /* no HIP device / not gfx950 / HIP runtime failure at init */
    KG_ERR_INVALID = -2,     /* bad argument */
    KG_ERR_HIP = -3,         /* a HIP call failed; see kg_last_error() */
    KG_ERR_NOMEM = -4,
    KG_ERR_STATE = -5        /* call out of order (e.g. correlate before a code was set) */
} kg_status;

const char *kg_strerror(int status);
const char *kg_last_error(void);
int kg_abi_version(void);

/* ------------------------------------------------------------------------ */
/* Context: one per GPU per process.                                         */
/* ------------------------------------------------------------------------ */
typedef struct kg_ctx kg_ctx;

/* stream: a hipStream_t to enqueue on (e.g. the caller's or PyTorch's current
 * stream), or NULL to let the library create its own non-blocking stream. */
int kg_ctx_create(int device, void *stream, kg_ctx **out);
/* The same, but `stream` is always taken as given: NULL here means HIP's legacy default
 * ("null") stream, which is what torch.cuda.current_stream().cuda_stream is (0) when no
 * other stream was selected.  The library never destroys a caller's stream. */
int kg_ctx_create_on_stream(int device, void *stream, kg_ctx **out);
void kg_ctx_destroy(kg_ctx *ctx);
int kg_ctx_sync(kg_ctx *ctx);                 /* hipStreamSynchronize */
int kg_ctx_poll(kg_ctx *ctx);                 /* 1 = stream idle, 0 = busy, <0 error */
void *kg_ctx_stream(kg_ctx *ctx);             /* the hipStream_t in use */
int kg_ctx_device_name(kg_ctx *ctx, char *buf, size_t len);
int kg_ctx_num_cus(kg_ctx *ctx);

/* Device-memory helpers for callers that do not bring their own HIP runtime
 * (the reference's host is plain C++): allocate / free HBM, blocking copies on
 * the context's stream. */
int kg_dev_alloc(kg_ctx *ctx, size_t bytes, void **out);
int kg_dev_free(kg_ctx *ctx, void *ptr);
int kg_dev_upload(kg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int kg_dev_download(kg_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* hipMemGetInfo of the context's device (lifecycle tests, capacity planning). */
int kg_dev_mem_info(kg_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* HIP-event stopwatch on the context's stream (bench.py and C++ callers). */
int kg_timer_start(kg_ctx *ctx);
int kg_timer_stop(kg_ctx *ctx, float *elapsed_ms);    /* synchronises on the stop event */
/* Profiling aid: enqueues an empty kernel (`kg_mark_kernel`) of `tag` workgroups of 64 threads on the
 * context's stream (1 <= tag <= 65535).  Counter rows of `rocprofv3 --pmc`, which carry no timestamps
 * or user markers, can then be attributed to the phase between two tags by dispatch order
 * (bench.py's live HBM-traffic passes). */
int kg_ctx_mark(kg_ctx *ctx, int tag);

/* ------------------------------------------------------------------------ */
/* GPS C/A + E1B parallel-code-phase acquisition.                            */
/* Replaces gps/search.cpp: SearchInit() code tables (:183-350), Sample()    */
/* (:382-449), Correlate() (:453-499).                                       */
/* The reference's shape (gps/gps.h:62-73): NSAMPLES 65536, DECIM 4,          */
/* FFT_LEN 16384 -- what kg_acq_create() builds.  kg_acq_create_shape() builds */
/* the same algorithm for a longer coherent interval (BASELINE.json           */
/* configs[4]); every size below then reads kg_acq_nsamples() / kg_acq_fft_len() */
/* in place of the two constants.                                             */
/* ------------------------------------------------------------------------ */
#define KG_ACQ_NSAMPLES 65536
#define KG_ACQ_FFT_LEN  16384
#define KG_ACQ_DECIM    4         /* gps/gps.h:62 */
/* BASELINE.json configs[4]: 10 ms coherent at FS = 16.368 MHz, zero-padded to a
 * 65536-point transform at SAMPLE_RATE = FS/DECIM = 4.092 MHz.  One Doppler bin is
 * SAMPLE_RATE / FFT_LEN = 62.44 Hz there (249.76 Hz in the reference shape). */
#define KG_ACQ10_NSAMPLES 163680
#define KG_ACQ10_FFT_LEN  65536
#define KG_ACQ_L1_LIMIT 4092      /* SAMPLE_RATE/1000*L1_CODE_PERIOD,  search.cpp:486 */
#define KG_ACQ_E1B_LIMIT 16368    /* SAMPLE_RATE/1000*E1B_CODE_PERIOD, search.cpp:486 */

typedef struct kg_acq kg_acq;

/* Result of Correlate() for one SV.  valid == 0 means no Doppler bin had
 * snr > 0 (all-zero or NaN input): the reference then leaves its out-pointers
 * untouched and returns 0 (search.cpp:455,495). */
typedef struct { float snr; int32_t dop; int32_t idx; int32_t valid; } kg_acq_result;

/* One (SV, Doppler) cell of the search.cpp:465-496 loop. */
typedef struct { float snr; float max_pwr; float tot_pwr; int32_t idx; } kg_acq_cell;

/* max_sats: size of the code table (reference MAX_SATS = 64, gps.h:123).
 * dop_lo..dop_hi: Doppler bins searched (reference -20..20, search.cpp:465).
 * max_blocks: how many independent 65536-sample blocks ("receivers") can be
 * resident and searched in one launch. */
int kg_acq_create(kg_ctx *ctx, int max_sats, int dop_lo, int dop_hi, int max_blocks,
                  kg_acq **out);
/* The same engine for another shape of the same algorithm: a sample block holds
 * `nsamples` input samples at FS (a multiple of 8, at most DECIM * fft_len; the rest of
 * the DECIM * fft_len array Sample() decimates is zero, as DecimateBy2float's tail is,
 * search.cpp:145), the transforms are fft_len points (16384 or 65536), the code replica
 * covers all DECIM * fft_len samples as the reference's covers NSAMPLES (:250), a Doppler
 * bin is one bin of that transform, the peak-search windows stay 4092 / 16368.
 * kg_acq_create() == kg_acq_create_shape(.., KG_ACQ_NSAMPLES, KG_ACQ_FFT_LEN, ..). */
int kg_acq_create_shape(kg_ctx *ctx, int max_sats, int dop_lo, int dop_hi, int max_blocks,
                        int nsamples, int fft_len, kg_acq **out);
int kg_acq_nsamples(kg_acq *acq);
int kg_acq_fft_len(kg_acq *acq);
void kg_acq_destroy(kg_acq *acq);

/* SearchInit() per-SV body, run on the device: resample chips {0,1} at 16
 * samples/chip over NSAMPLES, Bipolar, (boc: XOR BOC(1,1), search.cpp:317),
 * 2x DecimateBy2float, forward FFT.  limit = peak-search window
 * (KG_ACQ_L1_LIMIT / KG_ACQ_E1B_LIMIT). */
int kg_acq_set_code(kg_acq *acq, int sat, const uint8_t *chips, int nchips, int boc,
                    int limit);
/* Upload a ready-made code spectrum (FFT_LEN complex, natural bin order), i.e.
 * the first half of the reference's code[sat][] (search.cpp:54,283). */
int kg_acq_set_code_fft(kg_acq *acq, int sat, const float *code_fft, int limit);
int kg_acq_get_code_fft(kg_acq *acq, int sat, float *code_fft);      /* natural order */

/* Sample(): packed 1-bit IF (8192 bytes, LSB first; verilog/gps/sampler.v) ->
 * data spectrum of block `block`. */
int kg_acq_sample_bits(kg_acq *acq, int block, const uint8_t *packed);
int kg_acq_sample_bits_dev(kg_acq *acq, int block, const void *d_packed);
/* Extension (BASELINE.json configs[1]): NSAMPLES complex int16 samples
 * (i,q interleaved) at the FS/4 IF; mix by (-j)^n, then as Sample(). */
int kg_acq_sample_iq16(kg_acq *acq, int block, const int16_t *iq);
int kg_acq_sample_iq16_dev(kg_acq *acq, int block, const void *d_iq);
/* nblocks consecutive blocks from one device array, block b at d_iq + b*stride_bytes:
 * one launch set for all of them. */
int kg_acq_sample_iq16_batch_dev(kg_acq *acq, int first_block, int nblocks, const void *d_iq,
                                 size_t stride_bytes);
/* The same for nblocks blocks in host memory, stride_samples complex samples apart: copied
 * into the library's pinned staging region before the call returns, one transfer and one
 * front-end launch for the batch. */
int kg_acq_sample_iq16_batch(kg_acq *acq, int first, int nblocks, const int16_t *iq, size_t stride_samples);
/* Inject / read back Correlate()'s `data` argument (fwd_buf after Sample()),
 * FFT_LEN complex in natural bin order. */
int kg_acq_set_data_fft(kg_acq *acq, int block, const float *data_fft);
int kg_acq_get_data_fft(kg_acq *acq, int block, float *data_fft);
/* The FFT_LEN decimated time-domain samples Sample() feeds its FFT. */
int kg_acq_get_data_td(kg_acq *acq, int block, float *td);

/* Correlate() for nsats SVs x (dop_hi-dop_lo+1) bins x nblocks blocks
 * (blocks 0..nblocks-1), one launch.  Enqueue only: no stream synchronisation in steady
 * state, also when the SV list differs from the previous call's (the reference's
 * SearchTask loop asks for one SV at a time); the pair table goes through the context's
 * staging ring, whose slot reuse can wait on an event once 32 uploads are outstanding.
 * With fewer than 8 (block, SV) pairs in a launch -- that calling pattern -- the cells,
 * not the pairs, are dealt over the 8 XCDs, so one SV's bins use the whole GPU.
 * Stream model: Sample() and Correlate() run in order on the context's stream
 * (a second stream for Sample() is an opt-in experiment). */
int kg_acq_correlate_async(kg_acq *acq, int nblocks, const int *sats, int nsats);
int kg_acq_correlate_blocks_async(kg_acq *acq, int first_block, int nblocks, const int *sats,
                                  int nsats);
/* Wait and copy out.  results[nblocks*nsats] (block-major), cells may be NULL
 * or [nblocks*nsats*ndop]. */
int kg_acq_fetch(kg_acq *acq, kg_acq_result *results, kg_acq_cell *cells);
/* Convenience: async + fetch for one block. */
int kg_acq_correlate(kg_acq *acq, int block_count, const int *sats, int nsats,
                     kg_acq_result *results, kg_acq_cell *cells);
/* Device pointer to the last launch's kg_acq_result array (for RCCL gathers). */
void *kg_acq_results_dev(kg_acq *acq);

/* ------------------------------------------------------------------------ */
/* Waterfall frames.  Replaces, per (channel, frame): the unpack + window of   */
/* sample_wf() (rx/rx_waterfall.cpp:1049-1066) and compute_frame()             */
/* (:1275-1575: 8192-point FFT, power, CIC compensation, FFT-bin -> pixel      */
/* reduce, dB, clamp, u8).  The tables and maps stay the caller's: they are    */
/* the arrays c2s_waterfall_init() (:122-203) and the c2s_waterfall() loop     */
/* (:775-928) already build, passed as they are.                               */
/* ------------------------------------------------------------------------ */
#define KG_WF_NFFT  8192      /* WF_C_NFFT = WF_C_NSAMPS, rx/rx_waterfall.h:61-62 */
#define KG_WF_WIDTH 1024      /* WF_WIDTH, rx/rx_waterfall.h:65 */

typedef struct kg_wf kg_wf;

/* The scalar members of wf_inst_t (rx/rx_waterfall.h:118-157) compute_frame() reads. */
typedef struct {
    int32_t zoom;                 /* wf->zoom */
    int32_t window_func;          /* WINF_WF_* 0..3, rx_waterfall.h:160-163 */
    int32_t interp;               /* wf_interp_t: 0 MAX, 1 MIN, 2 LAST, 3 DROP, 4 CMA (rx_waterfall.h:116) */
    int32_t cic_comp;             /* wf->cic_comp */
    int32_t overlapped;           /* wf->overlapped_sampling */
    int32_t fft_used;             /* 4096 at zoom 0, 2048 otherwise (:756-763) */
    int32_t plot_width;           /* :772 */
    int32_t plot_width_clamped;   /* :773 */
    float fft_offset;             /* :898 */
} kg_wf_chan_cfg;

int kg_wf_create(kg_ctx *ctx, int nchan, kg_wf **out);
void kg_wf_destroy(kg_wf *wf);
/* wf_shmem_t.window_function[4][8192] and .CIC_comp[8192] (rx_waterfall.h:166-172) */
int kg_wf_set_tables(kg_wf *wf, const float *window_function, const float *cic_comp);
/* wf_inst_t.fft2wf_map[fft_used], .drop_sample[1024], .fft_scale[1024],
 * .fft_scale_div2[1024] of channel ch.  Only the "FFT >= plot" case (:1400) is
 * supported (fft_used >= plot_width), the only one reachable for FlyDog. */
int kg_wf_set_channel(kg_wf *wf, int ch, const kg_wf_chan_cfg *cfg, const uint16_t *fft2wf_map,
                      const uint16_t *drop_sample, const float *fft_scale,
                      const float *fft_scale_div2);
/* nframes frames; frame f belongs to channel chan_of[f] (host array), its input is
 * iq[f][8192] {int16 i, int16 q} (struct iq_t, :95-97) and its output out[f][1024]
 * bytes (wf_pkt_t.un.buf).  _dev: device pointers, enqueue only.  A frame's start is
 * kept as a 32-bit sample offset: nframes <= 524288 (2^32 / 8192) per call. */
int kg_wf_frames_dev(kg_wf *wf, int nframes, const int32_t *chan_of, const void *d_iq, void *d_out);
int kg_wf_frames(kg_wf *wf, int nframes, const int32_t *chan_of, const int16_t *iq, uint8_t *out);
/* The same for frames that are NOT back to back: frame f starts frame_off[f] samples (iq_t pairs; even,
 * below 2^32) after d_iq -- frames taken where the DDC left them (kg_ddc_wf_push_dev's per-channel rows),
 * the way sample_wf() reads the FPGA's sample ring in place (rx/rx_waterfall.cpp:1036-1066).  iq_len = how
 * many iq_t pairs d_iq points at: every frame must satisfy frame_off[f] + 8192 <= iq_len (KG_ERR_INVALID
 * otherwise -- an offset is never turned into a device read outside the caller's buffer). */
int kg_wf_frames_at_dev(kg_wf *wf, int nframes, const int32_t *chan_of, const uint64_t *frame_off,
                        uint64_t iq_len, const void *d_iq, void *d_out);
/* One frame with the intermediate arrays of compute_frame(): pwr[4096] (entries
 * below fft_used are written), pwr_out[1024], dB[1024] (before the clamp). */
int kg_wf_debug_frame(kg_wf *wf, int ch, const int16_t *iq, uint8_t *out, float *pwr,
                      float *pwr_out, float *dB);

/* ------------------------------------------------------------------------ */
/* Waterfall DDC.  In the reference this is FPGA fabric behind SPI commands:   */
/* WATERFALL_1CIC (verilog/rx/waterfall_1cic.v:20-144) = IQ_MIXER (iq_mixer.v)  */
/* + cic_prune_var "wf1" (cic_prune_var.v, cic_wf1.vh) + IQ_SAMPLER_8K_32B,     */
/* programmed with CmdSetWFFreq / CmdSetWFDecim / CmdWFReset and read with      */
/* CmdGetWFSamples (rx/rx_waterfall.cpp:466,507,1005,1036).  Here the ADC        */
/* stream is an int16 array in HBM and every listed channel is computed from it. */
/* ------------------------------------------------------------------------ */
typedef struct kg_ddc kg_ddc;

int kg_ddc_create(kg_ctx *ctx, int nchan, size_t max_samples, kg_ddc **out);
void kg_ddc_destroy(kg_ddc *ddc);
/* CmdSetWFFreq (48-bit phase increment, i_offset of rx_waterfall.cpp:498-507) and
 * CmdSetWFDecim (R = 1, 2, 4 .. 8192).  Also resets the channel (phase = 0). */
int kg_ddc_set_wf(kg_ddc *ddc, int ch, uint64_t phase_inc, int decim);
/* CmdWFReset with WF_SAMP_WR_RST: zero the CIC registers and the decimation
 * counter; the NCO phase keeps running. */
int kg_ddc_reset_wf(kg_ddc *ddc, int ch);
int kg_ddc_set_phase(kg_ddc *ddc, int ch, uint64_t phase);
/* The NCO sine / cosine tables both DDCs use (8192 int16 entries each, addressed by phase bits 47:35):
 * round(16383 cos / sin(2 pi a / 8192)) -- the frozen stand-in for the closed Xilinx DDS core of
 * verilog/rx/iq_mixer.v:60-65 (no dither).  Host function, needs no GPU; the device keeps the two as one
 * 10240-entry sine table (cos a = T[a + 2048]). */
int kg_ddc_nco_table(int16_t *cos_tab, int16_t *sin_tab);
/* IQ pairs channel ch will produce for the next n ADC samples. */
long kg_ddc_wf_outputs(kg_ddc *ddc, int ch, size_t n);
/* Run n ADC samples (device int16 array) through the listed channels.  Channel
 * chan_list[i] writes its IQ pairs {int16 i, int16 q} (struct iq_t) to
 * d_out + i*out_stride (in pairs); nouts[i] (may be NULL) receives the count.
 * State (NCO phase, CIC registers, decimation phase) carries over to the next
 * call, so a stream may be pushed in pieces of any length.  Enqueue only.
 * d_adc and the rows may sit at any 2- / 4-byte alignment; R = 1 channels are fastest (16-byte stores) when d_adc is
 * 8-byte aligned and every row 16-byte aligned, i.e. d_out 16-byte aligned and out_stride a multiple of 4. */
int kg_ddc_wf_push_dev(kg_ddc *ddc, const void *d_adc, size_t n, const int32_t *chan_list,
                       int nlist, void *d_out, size_t out_stride, int64_t *nouts);
/* The reference's NON-OVERLAPPED waterfall frame (sample_wf(), rx/rx_waterfall.cpp:1005-1041): CmdWFReset with
 * WF_SAMP_RD_RST | WF_SAMP_WR_RST -- verilog/rx/waterfall_1cic.v:45-47,106-128: both CICs (registers, decimation
 * counter) and the sampler's write pointer are cleared, the NCO keeps running -- then the ONE-SHOT sampler
 * (IQ_SAMPLER_8K_32B, wr_continuous = 0) fills with the next 8192 outputs and stops.  Here: the reset falls on the
 * block's first sample; channel chan_list[i] writes its first min(max_out, n >> log2 R) pairs to d_out + i*out_stride
 * and only the max_out * R samples that produce them reach its filters (a zoom-1 channel reads 8192 samples of a
 * 4 Mi-sample block, not all of it); every NCO advances by the whole block.  A channel captured from is left with its
 * filters stopped short: the next kg_ddc_wf_push_dev on it starts from the reset state, as the reference does when it
 * switches the sampler to continuous mode (CmdWFReset with WF_SAMP_CONTIN, :971-978).  Enqueue only. */
int kg_ddc_wf_capture_dev(kg_ddc *ddc, const void *d_adc, size_t n, const int32_t *chan_list,
                          int nlist, void *d_out, size_t out_stride, size_t max_out, int64_t *nouts);
/* Both sampler modes in ONE call (round 5; a bank of receivers, kg_rxbank below): entry i is pushed through the continuous
 * sampler when max_out[i] == 0 (as kg_ddc_wf_push_dev) and captured -- CmdWFReset at the block's first sample, then the
 * one-shot sampler of max_out[i] pairs -- when max_out[i] >= 1 (as kg_ddc_wf_capture_dev); its pairs go to
 * d_out + i*out_stride + out_off[i] (out_off may be NULL: all zero; out_off[i] + the entry's outputs <= out_stride), e.g. the
 * write position of a sample ring.  Enqueue only. */
int kg_ddc_wf_step_dev(kg_ddc *ddc, const void *d_adc, size_t n, const int32_t *chan_list, int nlist, void *d_out,
                       size_t out_stride, const int64_t *out_off, const int64_t *max_out, int64_t *nouts);
/* Deferred output stage (round 4) -- the non-blocking submit / poll form SURVEY 8(b) asks of the DDC seam (today every
 * CmdGetWFSamples is a blocking SPI transaction, platform/common/spi.cpp:487-507).  Off (default): everything a push
 * enqueues is ordered on the context's stream.  On: the push returns with its output stage (R = 1 bypass channels,
 * run-total prefix, combs: everything that WRITES d_out) on a stream of the object, and the context's stream carries
 * only the run passes -- the next push's run passes start while this push's outputs are still being written.
 *   kg_ddc_wf_join(ddc, stream)      `stream` (a hipStream_t; NULL: the context's) waits for the last push's outputs:
 *                                    call it on whichever stream reads d_out first.
 *   kg_ddc_wf_tail_after(ddc, event) the NEXT push's writers of d_out start only after `event` (a hipEvent_t the
 *                                    caller recorded behind its last reader of the rows): the write-after-read edge of
 *                                    a caller that still reads push k's rows while push k + 1 runs.  One-shot.
 * d_adc: whatever reads the caller's samples is ordered on the context's stream before the push returns (in both modes), so
 * the caller may refill or recycle d_adc on that stream right behind the push; only the rows need kg_ddc_wf_join.
 * Every other entry point of the object drains the deferred work first. */
int kg_ddc_wf_set_deferred(kg_ddc *ddc, int on);
int kg_ddc_wf_join(kg_ddc *ddc, void *stream);
int kg_ddc_wf_tail_after(kg_ddc *ddc, void *event);

/* ------------------------------------------------------------------------ */
/* Audio DDC.  In the reference: one RX instance per audio channel in FPGA      */
/* fabric (verilog/rx/rx.v:22-178): IQ_MIXER (22 bits) -> CIC N=3 R=1736 ->      */
/* CIC N=5 R=3 -> 65-tap CICF /2 (fir_iq.sv) -> 24-bit IQ at ADC/10416 (the rx4 / */
/* rx8 instance; rx3 and rx14: kg_rxddc_create_mode), read as                     */
/* rx_iq_t records with CmdGetRX (rx/data_pump.cpp:101) after the NCO was set    */
/* with CmdSetRXFreq (rx/rx_sound_cmd.cpp:41-51).                                */
/* ------------------------------------------------------------------------ */
typedef struct kg_rxddc kg_rxddc;

int kg_rxddc_create(kg_ctx *ctx, int nchan, size_t max_samples, kg_rxddc **out);   /* KG_RXDDC_STD */
/* The RX instances the reference builds, selected by its RX_CFG (kiwi.config:101-105, 140-143;
 * verilog/rx/fir_iq.sv:39-123; register widths as verilog/rx/cic_gen.c emits them):
 *   KG_RXDDC_STD   rx4 / rx8: CIC 1736 -> CIC 3 -> 65-tap CICF / 2 = ADC / 10416 (12 kHz class)
 *   KG_RXDDC_WIDE  rx3:       CIC 1543 -> CIC 2 -> 65-tap CICF (RX_CFG == 3 taps) / 2 = ADC / 6172 (20.25 kHz)
 *   KG_RXDDC_RX14  rx14:      CIC 1736 -> CIC 3 -> 17-tap CICF (RX_CFG == 14 taps) / 2 = ADC / 10416 */
enum { KG_RXDDC_STD = 0, KG_RXDDC_WIDE = 1, KG_RXDDC_RX14 = 2 };
int kg_rxddc_create_mode(kg_ctx *ctx, int nchan, size_t max_samples, int mode, kg_rxddc **out);
int kg_rxddc_decim(kg_rxddc *ddc);                          /* ADC samples per output record */
void kg_rxddc_destroy(kg_rxddc *ddc);
/* CmdSetRXFreq: 48-bit phase increment i_phase = round(f / adc_clk * 2^48).  The
 * filters keep running across a retune, as in the FPGA. */
int kg_rxddc_set_freq(kg_rxddc *ddc, int ch, uint64_t phase_inc);
int kg_rxddc_reset(kg_rxddc *ddc, int ch);                 /* power-on state */
long kg_rxddc_outputs(kg_rxddc *ddc, int ch, size_t n);     /* records the next n samples yield */
/* n ADC samples (device int16 array) through the listed channels: channel
 * chan_list[i] writes nouts[i] rx_iq_t records {u16 i, u16 q, u8 q3, u8 i3}
 * (rx/data_pump.h:27-30) to d_out + i*out_stride records.  State carries over
 * between calls.  Enqueue only. */
int kg_rxddc_push_dev(kg_rxddc *ddc, const void *d_adc, size_t n, const int32_t *chan_list,
                      int nlist, void *d_out, size_t out_stride, int32_t *nouts);

/* ------------------------------------------------------------------------ */
/* Audio front: data-pump unpack and the CFastFIR passband filter.             */
/* ------------------------------------------------------------------------ */
/* snd_service() unpack (rx/data_pump.cpp:145-208): nsamps * nchans rx_iq_t records
 * {u16 i, u16 q, u8 q3, u8 i3} (rx/data_pump.h:27-30), sample-major / channel-minor,
 * -> out[ch][nsamps] TYPECPX: re = q*rescale + DC_offset_I, im = i*rescale +
 * DC_offset_Q (I and Q as given when spectral_inversion).  enabled[ch] == 0 leaves
 * that channel's output untouched (rx_channels[ch].data_enabled).  Device buffers;
 * enabled is a host array.  Synchronous. */
int kg_dpump_unpack_dev(kg_ctx *ctx, const void *d_raw, int nsamps, int nchans,
                        const uint8_t *enabled, float rescale, float dc_i, float dc_q,
                        int spectral_inversion, void *d_out, size_t out_stride);

/* The same unpack for records stored one row per channel, raw_stride records apart: the
 * layout kg_rxddc_push_dev writes (many receivers, no SPI interleave).  Enqueue only. */
int kg_dpump_unpack_rows_dev(kg_ctx *ctx, const void *d_raw, size_t raw_stride, int nsamps, int nchans,
                             const uint8_t *enabled, float rescale, float dc_i, float dc_q,
                             int spectral_inversion, void *d_out, size_t out_stride);

#define KG_FIR_FFT_SIZE 1024      /* CONV_FFT_SIZE, rx/CuteSDR/cuteSDR.h:12 */
#define KG_FIR_OUT      512       /* FASTFIR_OUTBUF_SIZE, rx/CuteSDR/cuteSDR.h:14 */

typedef struct kg_fir kg_fir;      /* the m_PassbandFIR[MAX_RX_CHANS] array, rx/rx_sound.cpp:150 */

/* max_in: the largest InLength of one ProcessData call. */
int kg_fir_create(kg_ctx *ctx, int nchan, int max_in, kg_fir **out);
void kg_fir_destroy(kg_fir *fir);
/* CFastFIR::SetupParameters(instance, FLoCut, FHiCut, Offset, SampleRate)
 * (rx/CuteSDR/fastfir.cpp:171-232) with the window of SetupWindowFunction
 * (:102-146; window_func < 0 = Blackman-Nuttall) and SetupCICFilter (:148-158;
 * snd_rate_3ch selects the :70-71 constants).  Taps are designed on the host in
 * the reference's float arithmetic, transformed on the device.  Returns 1 when the
 * sanity check (:193-200) rejects the parameters and the old filter stays. */
int kg_fir_setup(kg_fir *fir, int ch, float FLoCut, float FHiCut, float Offset, float SampleRate,
                 int window_func, int do_cic_comp, int snd_rate_3ch);
/* Or hand over the reference's own m_pFilterCoef_CIC[1024] (complex float); the channel's m_CIC[] is
 * then 1.0 (m_do_CIC_comp false, what FlyDog builds: fastfir.cpp:94). */
int kg_fir_set_coef(kg_fir *fir, int ch, const float *coef_fft);
int kg_fir_get_coef(kg_fir *fir, int ch, float *coef_fft);
int kg_fir_reset(kg_fir *fir, int ch);
int kg_fir_pos(kg_fir *fir, int ch);                        /* CFastFIR::FirPos(), fastfir.h:33 */
/* CFastFIR::ProcessData(rx_chan, InLength, In, Out) (fastfir.cpp:241-324), host
 * buffers: returns the number of complex samples written to out (0 or a multiple
 * of 512), or a negative status. */
int kg_fir_process(kg_fir *fir, int ch, const float *in, int n, float *out);
/* The same for a list of channels at once, device buffers: channel chans[i] takes
 * n samples from d_in + i*in_stride and writes nout[i] samples to d_out + i*out_stride
 * (strides in complex samples).  Enqueue only. */
int kg_fir_process_dev(kg_fir *fir, const int32_t *chans, int nch, const void *d_in,
                       size_t in_stride, int n, void *d_out, size_t out_stride, int32_t *nout);
/* The same with an InLength of its own for every listed channel (n_each[i] >= 0 samples at row i of d_in): connections whose
 * audio DDCs were started at different times deliver different record counts in one data-pump interval. */
int kg_fir_process_each_dev(kg_fir *fir, const int32_t *chans, int nch, const void *d_in, size_t in_stride,
                            const int32_t *n_each, void *d_out, size_t out_stride, int32_t *nout);

/* ---------------------------------------------------------------------------
 * What consumes the CFastFIR output in c2s_sound(), per receiver channel
 * (SURVEY.md 8(f) rank 1): S-meter (rx/rx_sound.cpp:248-250, 676-696), the m_Agc[]
 * array (rx/rx_sound.cpp:152; rx/CuteSDR/agc.cpp) and the AM / NBFM detectors
 * (rx/rx_sound.cpp:766-783, 845-881).  One object holds these for nchan receivers;
 * kg_post_process_dev() runs a batch of channels in one launch (one wavefront per
 * channel: the recursions are sequential per channel, parallel across channels).
 * ------------------------------------------------------------------------- */
typedef struct kg_post kg_post;

enum {                      /* what follows the S-meter for a channel (the `switch (s->mode)` of :763-900) */
    KG_POST_IQ   = 0,       /* MODE_IQ/DRM: CAgc complex -> complex (rx_sound.cpp:1096-1100)                      -> d_agc */
    KG_POST_SSB  = 1,       /* MODE_USB/USN/LSB/LSN/CW/CWN: CAgc complex -> mono16 (:893)                         -> d_s16 */
    KG_POST_AM   = 2,       /* MODE_AM/AMN: CAgc, envelope, DC-removal IIR (:766-783) -> d_demod; m_AM_FIR (:787) -> d_s16 */
    KG_POST_NBFM = 3        /* MODE_NBFM/NNFM: CAgc, fmdemod_quadri + clipper (:845-875) -> d_demod;
                             * m_Squelch.PerformFMSquelch (:876, rx/CuteSDR/squelch.cpp:151-231)                  -> d_s16 */
};                          /* SSB, AM, NBFM: then the de-emphasis filter over d_s16 in place when it is on (:898-907)   */

#define KG_POST_MAX_SAMPLES 1024  /* per call and channel; c2s_sound() hands over ns_out = 512 */

int kg_post_create(kg_ctx *ctx, int nchan, kg_post **out);
void kg_post_destroy(kg_post *post);
/* CAgc::SetParameters(AgcOn, UseHang, Threshold, ManualGain, SlopeFactor, Decay, SampleRate)
 * (agc.cpp:98-163): returns at once when nothing changed; a new sample rate clears the
 * delay line, the magnitude window and the averagers.  Constants are computed on the
 * host with the reference's float/double expressions.  A fresh object is in the state
 * the reference reaches after its first call with a new sample rate. */
int kg_post_set_agc(kg_post *post, int chan, int agc_on, int use_hang, int threshold, int manual_gain,
                    int slope_factor, int decay, float sample_rate);
int kg_post_agc_delay(kg_post *post, int chan);           /* CAgc::GetDelaySamples(), agc.h:27 */
/* sMeterAlpha = 1 - expf(-1 / (frate * ATTACK_TIMECONST)) (rx_sound.cpp:248-249) */
int kg_post_set_smeter(kg_post *post, int chan, float frate);
int kg_post_set_mode(kg_post *post, int chan, int mode);
int kg_post_get_mode(kg_post *post, int chan);              /* -> KG_POST_*, or < 0 */
/* The three libm functions the device code of this path calls, over an array: y[i] = f(x[i]), or, with d_x NULL, f of the float
 * whose bit pattern is first_bits + i.  KG_MATH_LOG10F: the S-meter's and CAgc's log10f (rx/rx_sound.cpp:687, rx/CuteSDR/agc.cpp:191;
 * CAgc BRANCHES on it, :215-240); KG_MATH_POWF: powf(base, x), CAgc's gain with base 10 (agc.cpp:250-253; base positive, finite,
 * normal); KG_MATH_EXPF: aperture_auto()'s IIR gain (rx/rx_waterfall.cpp:1199).  The reference calls the platform's libm; the device
 * functions restate the GNU C Library 2.35 algorithms of this image (csrc/kg_libm.h) and equal them bit for bit on every argument
 * (tests/test_libm_gpu.py, tools/check_libm.py --exhaustive), which is what makes the audio chain's outputs the reference's own bits.
 * Enqueue only. */
enum { KG_MATH_LOG10F = 0, KG_MATH_POWF = 1, KG_MATH_EXPF = 2 };
int kg_math_dev(kg_ctx *ctx, int fn, float base, const void *d_x, uint32_t first_bits, size_t n, void *d_y);
/* A new connection on the channel: sMeterAvg_dB = 0, z1 = 0 (rx_sound.cpp:244,250),
 * conn->last_sample = 0.  The AGC object persists across connections, as m_Agc[] does. */
int kg_post_reset(kg_post *post, int chan);
/* One pass over nsamps FIR output samples of each listed channel (d_fir + i*in_stride,
 * complex float).  Outputs (any may be NULL) at row i*out_stride of d_s16 (int16: out_samps_s2, every
 * mode but IQ), d_demod (float: the detector's output, AM / NBFM), d_agc (complex float: the AGC's output,
 * every mode but SSB).  Float -> mono16 is
 * the reference's (TYPEMONO16) cast: truncation; outside the int16 range (undefined in
 * C) the low 16 bits of the int32 conversion, as x86 does.  Enqueue only. */
int kg_post_process_dev(kg_post *post, const int32_t *chans, int nch, const void *d_fir, size_t in_stride,
                        int nsamps, void *d_s16, void *d_demod, void *d_agc, size_t out_stride);
/* The CFir objects of a channel (rx/rx_sound.cpp:153-156; rx/CuteSDR/fir.cpp): the filter behind the AM detector and the
 * two de-emphasis filters.  (CSquelch owns a fourth, its noise high-pass: kg_post_squelch_setup.) */
enum { KG_CFIR_AM = 0, KG_CFIR_DEEMP_NFM = 1, KG_CFIR_DEEMP_AM_SSB = 2,
       KG_CFIR_SQUELCH_HP = 3 };     /* CSquelch::m_HpFir: readable and runnable, designed only by kg_post_squelch_setup */
enum { KG_CFIR_REAL_REAL = 0, KG_CFIR_REAL_MONO16 = 1, KG_CFIR_MONO16_MONO16 = 2 };   /* the ProcessFilter overloads, fir.cpp:74, :176, :199 */
/* CFir::InitLPFilter(NumTaps, Scale, Astop, Fpass, Fstop, Fsamprate) (fir.cpp:282-384): Kaiser-windowed sinc designed on the
 * host in the reference's float arithmetic; clears the filter's sample buffer.  Returns the tap count (9..97), or < 0. */
int kg_post_cfir_init_lp(kg_post *post, int chan, int which, int NumTaps, float Scale, float Astop, float Fpass,
                         float Fstop, float Fsamprate);
/* CFir::InitConstFir(NumTaps, pCoef, Fsamprate) (fir.cpp:220-240): the caller's coefficients (the reference hands over a row
 * of rx/rx_filter.h's de-emphasis tables, rx/rx_sound_cmd.cpp:556-585); more than 97 are cut to 97.  Returns the tap count. */
int kg_post_cfir_init_const(kg_post *post, int chan, int which, int NumTaps, const float *coef, float Fsamprate);
int kg_post_cfir_get_taps(kg_post *post, int chan, int which, float *taps);    /* -> tap count; taps may be NULL */
/* m_*_FIR[chan].ProcessFilter(nsamps, in, out) on its own, for a list of channels (rows in_stride / out_stride elements apart;
 * float or int16 by `kind`; in == out allowed): the same device code as inside kg_post_process_dev.  The sums run in the
 * reference's order (fir.cpp:79-91: over the circular buffer's positions, so the first tap of the sum rotates with the
 * write position): bit-exact.  Enqueue only. */
int kg_post_cfir_process_dev(kg_post *post, const int32_t *chans, int nch, int which, int kind, const void *d_in, size_t in_stride,
                             int nsamps, void *d_out, size_t out_stride);
/* m_Squelch[chan].PerformFMSquelch(nsamps, in, out) on its own (squelch.cpp:151-231): float detector samples -> mono16 (1 when
 * squelched); the return values through kg_post_squelch_state.  Enqueue only. */
int kg_post_squelch_perform_dev(kg_post *post, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int nsamps,
                                void *d_out, size_t out_stride);
/* The post-AM-detector filter as a passband change designs it (rx/rx_sound_cmd.cpp:248-250, 268-282): the cuts clamped to
 * +-(frate / 2 - 1) as the handler clamps s->locut / s->hicut (a no-op for a caller that passes those), hbw = max(|hicut|, |locut|)
 * capped at frate / 2, stop = 1.8 hbw capped at frate / 2, m_AM_FIR.InitLPFilter(0, 1.0, 50.0, hbw, stop, frate).  Returns the
 * tap count.  A channel in KG_POST_AM mode without it is refused (the reference's undesigned CFir holds garbage). */
int kg_post_set_am_passband(kg_post *post, int chan, double locut, double hicut, double frate);
/* "SET de_emp=%d nfm=%d" (rx/rx_sound_cmd.cpp:543-554): s->deemp_nfm (nfm != 0) or s->deemp = de_emp; non-zero turns the
 * corresponding filter on for the channel's NBFM resp. AM / SSB modes (do_de_emp, rx/rx_sound.cpp:482).  The coefficients
 * come through kg_post_cfir_init_const. */
int kg_post_set_deemp(kg_post *post, int chan, int nfm, int de_emp);
/* CSquelch (rx/CuteSDR/squelch.cpp) of a channel: SetupParameters(rx_chan, samplerate) (:84-116; designs the noise high-pass,
 * InitHPFilter(0, 1.0, 50.0, 2400, 1950, samplerate), and resets), SetSquelch(Value, SquelchMax) (:122-129; SquelchMax 0 =
 * 8192), Reset() (:67-77).  rx/rx_sound.cpp:261-262 calls the first two for every new connection; a channel in KG_POST_NBFM
 * mode without them is refused.  (The reference clears conn->last_sample with Reset(), rx/rx_sound_cmd.cpp:238-239: that
 * is kg_post_reset's.) */
int kg_post_squelch_setup(kg_post *post, int chan, float samplerate);
int kg_post_squelch_set(kg_post *post, int chan, int Value, int SquelchMax);
int kg_post_squelch_reset(kg_post *post, int chan);
/* After the last pass: nsq_nc_sq[i] = what PerformFMSquelch returned for channel chans[i] (-1 opened, 0 no change, +1 closed),
 * squelched[i] = s->squelched as rx/rx_sound.cpp:877 keeps it (SND_FLAG_SQUELCH_UI, :1232), ave[i] = m_SquelchAve.  Any
 * output may be NULL.  Synchronises the stream. */
int kg_post_squelch_state(kg_post *post, const int32_t *chans, int nch, int32_t *nsq_nc_sq, int32_t *squelched, float *ave);
/* S-meter state after the last pass: avg_dB[i] = sMeterAvg_dB, and (taps != NULL)
 * taps[2i], taps[2i+1] = the values receive_S_meter() is handed at j == 0 and j == ns_out/2
 * (rx_sound.cpp:693), all before S_meter_cal is added.  Synchronises the stream. */
int kg_post_smeter(kg_post *post, const int32_t *chans, int nch, float *avg_dB, float *taps);

/* ---------------------------------------------------------------------------
 * Wire formats: the IMA ADPCM coder of rx/csdr/ima_adpcm.cpp
 * as c2s_sound() and compute_frame() use it, the waterfall packet and the sound packet
 * header, so that what leaves the GPU is byte-compatible with the web client.
 * Integer work: bit-exact.
 * ------------------------------------------------------------------------- */
typedef struct kg_adpcm kg_adpcm;   /* the per-connection `ima_adpcm_state_t adpcm_snd` (rx/rx_sound.h) of nchan channels */

int kg_adpcm_create(kg_ctx *ctx, int nchan, kg_adpcm **out);
void kg_adpcm_destroy(kg_adpcm *a);
/* memset(&s->adpcm_snd, 0, ...) is set_state(chan, 0, 0); get_state() is what the
 * "MSG audio_adpcm_state=%d,%d" message carries (rx/rx_sound.cpp:1314): index, previousValue. */
int kg_adpcm_set_state(kg_adpcm *a, int chan, int index, int previous);
int kg_adpcm_get_state(kg_adpcm *a, int chan, int *index, int *previous);
/* encode_ima_adpcm_i16_e8(out_samps_s2, bp_real_u1, ns_out, &s->adpcm_snd) (ima_adpcm.cpp:185-197,
 * rx/rx_sound.cpp:1122) for a list of channels: row i of d_s16 (int16, in_stride samples apart)
 * -> nsamps/2 bytes at d_out + i*out_stride (bytes).  nsamps even.  Enqueue only. */
int kg_adpcm_encode_dev(kg_adpcm *a, const int32_t *chans, int nch, const void *d_s16, size_t in_stride,
                        int nsamps, void *d_out, size_t out_stride);
/* The uncompressed payload (rx/rx_sound.cpp:1126-1140): int16 rows copied as they are
 * (little_endian != 0) or byte-swapped to network order.  Enqueue only. */
int kg_snd_payload_dev(kg_ctx *ctx, const void *d_s16, size_t in_stride, int nch, int nsamps,
                       int little_endian, void *d_out, size_t out_stride);
/* The IQ modes' payload (rx/rx_sound.cpp:1076-1096; MODE_IQ, and SAS / QAM / DRM monitor with their own sources): row i of d_cpx
 * (complex float: the AGC's output, d_agc of kg_post_process_dev) -> nsamps x {(s2_t) re, (s2_t) im}, 4 nsamps bytes at
 * d_out + i*out_stride, little-endian as they are or in network order.  chans: NULL (rows 0 .. nch-1), or the channel list
 * (a receiver bank's rows go by channel).  Enqueue only. */
int kg_snd_iq_payload_dev(kg_ctx *ctx, const int32_t *chans, int nch, const void *d_cpx, size_t in_stride, int nsamps,
                          int little_endian, void *d_out, size_t out_stride);
/* The 10 header bytes of snd_pkt_real_t (rx/rx_sound.h:42-48; rx/rx_sound.cpp:252,
 * 1219-1254): "SND", flags, seq little-endian, S-meter clamped to -127 .. 3.4 dBm and sent
 * big-endian in 0.1 dB steps above -127.  Host only. */
void kg_snd_header(uint8_t flags, uint32_t seq, float smeter_dBm, uint8_t *h);

/* The GPS time stamp of snd_pkt_iq_t (rx/rx_sound.h:61-64), host arithmetic as c2s_sound() does it.
 * State per sound connection (snd_t::gpssec, last_gpssec, gps_init, rx/rx_sound.h:120-122). */
typedef struct { double gpssec, last_gpssec; int32_t gps_init; int32_t pad; } kg_gps_state;
typedef struct { uint32_t gpssec, gpsnsec; uint8_t last_gps_solution; uint8_t pad[3]; } kg_iq_stamp;
/* rx/rx_sound.cpp:557, once per data-pump buffer: gpssec = fmod(week + clk.gps_secs + dticks /
 * clk.adc_clock_base - gps_delay + gps_delay2, week); dticks = the buffer's 48-bit tick count minus
 * clk.ticks. */
void kg_snd_gps_begin(kg_gps_state *s, double clk_gps_secs, double dticks, double adc_clock_base,
                      double gps_delay, double gps_delay2);
/* rx/rx_sound.cpp:636-661, once per 512-sample FIR output block: the FIR delay (norm_nrx_samps -
 * fir_pos, fir_pos = kg_fir_pos() before the block) and the AGC delay (kg_post_agc_delay(), when the
 * AGC is on) are taken off, the header carries the PREVIOUS block's time (last_gpssec), and
 * last_gps_solution = 255 without a clock solution (clk_ticks == 0), else min(252, seconds since it),
 * 0 on the first block of a connection. */
void kg_snd_gps_stamp(kg_gps_state *s, int norm_nrx_samps, int fir_pos, int agc_on, int agc_delay,
                      int rx_decim, double adc_clock_base, double clk_gps_secs, uint64_t clk_ticks,
                      kg_iq_stamp *out);

#define KG_WF_ADPCM_PAD 10                          /* ADPCM_PAD, rx/rx_waterfall.h:83 */
#define KG_WF_PKT_HDR   16                          /* id4, x_bin_server, flags_x_zoom_server, seq */
#define KG_WF_PKT_MAX   (KG_WF_PKT_HDR + KG_WF_ADPCM_PAD + 1024)   /* sizeof(wf_pkt_t) */
typedef struct {
    uint32_t x_bin_server;          /* wf->start or wf->prev_start (rx_waterfall.cpp:1603-1616) */
    uint32_t zoom;                  /* wf->zoom or wf->prev_zoom; WF_FLAGS_COMPRESSION is added here */
    uint32_t seq;                   /* wf->snd_seq (:1635) */
    int32_t use_compression;        /* wf->compression, the connection's setting; a row at zoom 0 is never compressed (:1283-1285) */
} kg_wf_pkt_info;
/* wf_pkt_t for nrows waterfall rows (1024 u8 each, row_stride bytes apart, as
 * kg_wf_frames_dev leaves them): header + either the row or, compressed, the 10 pad bytes
 * (copies of the first pixel) and the row through encode_ima_adpcm_u8_e8 with a fresh state
 * (rx_waterfall.cpp:1622-1631).  Packet i starts at d_pkts + i*pkt_stride (>= KG_WF_PKT_MAX);
 * pkt_bytes[i] (host) = what goes on the wire: 16 + wf->out_bytes.  Enqueue only. */
int kg_wf_packets_dev(kg_ctx *ctx, const void *d_rows, size_t row_stride, int nrows,
                      const kg_wf_pkt_info *info, void *d_pkts, size_t pkt_stride, int32_t *pkt_bytes);

/* ---------------------------------------------------------------------------
 * Hand-off: what turns the path's results into the reference's
 * own programming.
 * ------------------------------------------------------------------------- */
typedef struct {
    double lo_dop, ca_dop;          /* Hz: Doppler from the FFT bin shift, and of the code rate */
    uint32_t lo_rate, ca_rate;      /* CmdSetRateLO / CmdSetRateCG words */
    uint32_t ca_pause;              /* CmdPause takes ca_pause - 1 when ca_pause != 0 */
    int32_t code_creep;             /* samples */
} kg_chan_start;
/* The arithmetic of CHANNEL::Start(sat, t_sample, lo_shift, ca_shift, snr)
 * (gps/channel.cpp:281-311) for a kg_acq_result: lo_shift = result.dop, ca_shift =
 * result.idx * DECIM (gps/search.cpp:575), secs = (timer_us() - t_sample) / 1e6.
 * Host only, double arithmetic as in the reference. */
void kg_acq_chan_start(int is_e1b, int lo_shift, int ca_shift, double secs, kg_chan_start *out);

typedef struct kg_aper kg_aper;     /* wf_inst_t::avg_pwr[APER_PWR_LEN] of nchan waterfalls (rx/rx_waterfall.h:154) */
enum { KG_APER_IIR = 0, KG_APER_MMA = 1, KG_APER_EMA = 2 };      /* aper_algo_t, rx/rx_waterfall.h:113 */
typedef struct {
    int32_t algo;                   /* KG_APER_*; a single-shot request is MMA with param 8 (:1192-1195) */
    float param;                    /* wf->aper_param */
    int32_t clear;                  /* wf->avg_clear: load the averages from this row (:1183-1188) */
    int32_t audio_fft;              /* rx_chan >= wf_chans: pixels 256..767 only (:1180-1181) */
} kg_aper_cfg;
int kg_aper_create(kg_ctx *ctx, int nchan, kg_aper **out);
void kg_aper_destroy(kg_aper *a);
/* The averaging half of aperture_auto() (rx/rx_waterfall.cpp:1183-1222) for row i of d_rows
 * (1024 u8 pixels, row_stride apart) and channel chans[i]; pixels go through dB_wire_to_dBm()
 * with waterfall_cal (rx/rx_util.cpp:905-912).  Enqueue only. */
int kg_aper_update_dev(kg_aper *a, const int32_t *chans, int nrows, const void *d_rows, size_t row_stride,
                       const kg_aper_cfg *cfg, int waterfall_cal);
/* The reporting half (:1233-1272): signal = highest 5 dB band present (at least -80), noise =
 * the most populated band (the lowest of equals); bands <= -190 are masked areas; no band
 * at all gives -110 / -120.  Averages must lie within -190 .. 1000 dBm.  Synchronises. */
int kg_aper_report(kg_aper *a, const int32_t *chans, int n, const int32_t *audio_fft, int32_t *signal,
                   int32_t *noise);
int kg_aper_get(kg_aper *a, int chan, float *avg_pwr);            /* 1024 floats */

/* kg_fir_process_dev plus the extension taps of ProcessData (SURVEY.md 8(f) rank 4;
 * rx/CuteSDR/fastfir.cpp:278-302): for block b of list entry i, 1024 complex floats at
 * d_pre / d_post + i*tap_stride + b*1024 (either may be NULL): pre = the forward spectrum
 * times the CIC compensation table (what receive_FFT(PRE_FILTERED) is handed), post = the
 * filtered spectrum (receive_FFT(POST_FILTERED), specAF_FFT).  A PRE_FILTERED extension that
 * edits the buffer (the `buf_modified` path, :286-290) is not supported on this path.  The
 * other taps of c2s_sound() are plain buffers of this API: receive_iq_pre_fir = the unpack
 * output, receive_iq_pre_agc = the FIR output (an iq_buf_t ring when d_out walks
 * [N_DPBUF][512]), receive_iq_post_agc / receive_real / receive_S_meter = kg_post outputs. */
int kg_fir_process_taps_dev(kg_fir *fir, const int32_t *chans, int nch, const void *d_in, size_t in_stride, int n,
                            void *d_out, size_t out_stride, int32_t *nout, void *d_pre, void *d_post,
                            size_t tap_stride);

/* A PRE_FILTERED extension that EDITS the spectrum it is handed (`buf_modified`, fastfir.cpp:286-290):
 * after kg_fir_process_taps_dev() delivered d_pre, the caller rewrites those 1024-point blocks on the
 * device and calls this.  Block b (b < nblk[i]) of list entry i is filtered again as the reference
 * filters a modified buffer -- m_pFilterCoef (the coefficients WITHOUT the CIC compensation) times the
 * edited block, backward transform, samples 512..1023 -- into d_out + i*out_stride + 512 b, replacing
 * what kg_fir_process_taps_dev wrote there.  The un-compensated coefficients are those kg_fir_setup
 * designed, or what kg_fir_set_coef_plain handed over.  Enqueue only. */
int kg_fir_refilter_dev(kg_fir *fir, const int32_t *chans, int nch, const int32_t *nblk, const void *d_pre,
                        size_t tap_stride, void *d_out, size_t out_stride);
/* m_pFilterCoef[1024] (complex float) for kg_fir_refilter_dev when kg_fir_set_coef supplied
 * m_pFilterCoef_CIC; kg_fir_set_coef alone uses the same array for both. */
int kg_fir_set_coef_plain(kg_fir *fir, int ch, const float *coef_fft);

/* Diagnostics: re-runs the 4096-point stage of the forward FFT of `block` in a
 * stamped build of the kernel and returns 4 s_memrealtime readings (100 MHz):
 * start, inputs + twiddles loaded, transform done, results stored. */
int kg_acq_debug_fft_stamps(kg_acq *acq, int block, unsigned long long *stamps, int n);
/* Diagnostics: one Correlate() launch in a stamped build; stamps[0..3] = kernel
 * start/end (s_memtime cycles, s_memrealtime 100 MHz) of one workgroup, then 16
 * s_memtime readings per 4096-point work item (24 items) from stamps[16]; from stamps[512],
 * four values per workgroup b: start, end (s_memrealtime), XCC id register, cells done.
 * n >= 4608. */
int kg_acq_debug_corr_stamps(kg_acq *acq, int nblocks, const int *sats, int nsats,
                             unsigned long long *stamps, int n);

/* ------------------------------------------------------------------------ */
/* A bank of virtual receivers stepped with ONE call (round 5; BASELINE           */
/* configs[3]).  In the reference every connection runs a waterfall and a sound   */
/* coroutine over what the data pump hands them: data_pump()                      */
/* (rx/data_pump.cpp:292-341), the c2s_sound() loop (rx/rx_sound.cpp:333-601:     */
/* in_samps -> CFastFIR -> S-meter / AGC / demod -> compression) and               */
/* c2s_waterfall() -> sample_wf() -> compute_frame() -> wf_pkt_t                    */
/* (rx/rx_waterfall.cpp:930-1170).  A bank is nrx such connections fed from one     */
/* block of ADC samples per step; kg_rxbank_step() enqueues everything a step        */
/* needs -- both DDCs, frames, packets, unpack, CFastFIR, S-meter / CAgc, ADPCM --   */
/* on the bank's own streams, with ONE host-to-device transfer (the small tables     */
/* of all stages) per step.  The per-seam objects are the bank's and are configured  */
/* with their own entry points above (tables, maps, frequencies, filters, AGC).      */
/* ------------------------------------------------------------------------ */
typedef struct kg_rxbank kg_rxbank;

/* nrx receivers on `device`, adc_samples_per_step int16 ADC samples per step, audio DDCs of rx_mode (KG_RXDDC_*). */
int kg_rxbank_create(int device, int nrx, size_t adc_samples_per_step, int rx_mode, kg_rxbank **out);
void kg_rxbank_destroy(kg_rxbank *bank);
/* The owned objects, channel k = receiver k (never destroy them; reconfigure only between steps -- the setters of the
 * objects synchronise their own stream). */
kg_ctx *kg_rxbank_ctx(kg_rxbank *bank);       /* the waterfall chain's context: kg_dev_download etc. */
kg_ddc *kg_rxbank_ddc(kg_rxbank *bank);       /* do NOT call kg_ddc_set_wf on it: kg_rxbank_set_wf */
kg_wf *kg_rxbank_wf(kg_rxbank *bank);         /* kg_wf_set_tables, kg_wf_set_channel */
kg_rxddc *kg_rxbank_rxddc(kg_rxbank *bank);   /* kg_rxddc_set_freq */
kg_fir *kg_rxbank_fir(kg_rxbank *bank);       /* kg_fir_setup */
kg_post *kg_rxbank_post(kg_rxbank *bank);     /* kg_post_set_agc / _set_smeter / _set_mode / _reset */
kg_adpcm *kg_rxbank_adpcm(kg_rxbank *bank);
/* CmdSetWFFreq + CmdSetWFDecim + the sampler mode sample_wf() decides on (rx/rx_waterfall.cpp:962-1008):
 *   overlapped == 0   CmdWFReset + the one-shot sampler every step: the non-overlapped frame (:1005-1041); needs
 *                     8192 * decim <= adc_samples_per_step
 *   overlapped == 1   the continuous sampler (CmdWFReset with WF_SAMP_CONTIN, :971-978): every step adds
 *                     adc_samples_per_step / decim outputs (a divisor of 8192) to the receiver's ring and the frame is the
 *                     ring's newest 8192 outputs (CmdGetWFContSamps, :980-991); no frame until the ring holds 8192
 *                     ("fill pipe", :978).  Set kg_wf_chan_cfg.overlapped accordingly (it switches the CIC compensation off).
 * Resets the receiver's sampler.  Synchronises the bank. */
int kg_rxbank_set_wf(kg_rxbank *bank, int rx, uint64_t phase_inc, int decim, int overlapped);
/* Connections come and go one at a time (every c2s_sound() / c2s_waterfall() of the reference is its own loop with its own
 * CFastFIR position and sequence numbers, rx/rx_sound.cpp:264-269, 503-613).  A fresh bank has every receiver active.
 * kg_rxbank_leave: the receiver is skipped by every stage from the next step on.  kg_rxbank_join: its audio DDC, CFastFIR,
 * S-meter / detector, ADPCM state and sound sequence number start from zero and its waterfall sampler waits for
 * kg_rxbank_set_wf; the OTHER receivers are not touched -- from here on this receiver's records per step and the steps on
 * which its 512-sample sound blocks complete are its own (kg_rxbank_audio_map).  Between steps; join drains the bank. */
int kg_rxbank_join(kg_rxbank *bank, int rx);
int kg_rxbank_leave(kg_rxbank *bank, int rx);
int kg_rxbank_is_active(kg_rxbank *bank, int rx);
/* Per receiver after the last step (arrays of nrx entries, any may be NULL): records and CFastFIR outputs (0 or k * 512) the
 * step gave it, FirPos() now, sound blocks emitted since it joined (the seq of its next wf_pkt_t, rx_waterfall.cpp:1635). */
int kg_rxbank_audio_map(kg_rxbank *bank, int32_t *nrec, int32_t *nfir, int32_t *fir_pos, uint32_t *snd_seq);
/* wf_pkt_t header fields of receiver rx (x_bin_server, zoom, compression); seq is the bank's sound sequence number. */
int kg_rxbank_set_wf_pkt(kg_rxbank *bank, int rx, uint32_t x_bin_server, uint32_t zoom, int use_compression);
/* snd_service() unpack parameters (default: rescale of rx/data_pump.cpp:73-74, no DC offset, no inversion) */
/* "SET little-endian" of a connection (rx/rx_sound.cpp:1076-1096): the byte order of receiver rx's IQ-mode payload (default:
 * network order).  A receiver whose kg_post mode is KG_POST_IQ gets its sound blocks as IQ payload rows (bufs.iq_pay), the
 * others as ADPCM rows (bufs.adpcm); the mode is read at every step. */
int kg_rxbank_set_little_endian(kg_rxbank *bank, int rx, int little_endian);
int kg_rxbank_set_unpack(kg_rxbank *bank, float rescale, float dc_i, float dc_q, int spectral_inversion);

typedef struct {
    uint64_t step;         /* steps taken before this one */
    int32_t nframes;       /* waterfall frames (= rows = packets) of this step: kg_rxbank_frame_map says whose */
    /* the next four: of the lowest-numbered ACTIVE receiver (all receivers of a bank that was never joined into agree);
     * per receiver: kg_rxbank_audio_map */
    int32_t nrec;          /* rx_iq_t records */
    int32_t nfir;          /* CFastFIR outputs: 0 or k * 512 (k sound blocks: s16 / adpcm rows hold k * 512 / k * 256) */
    int32_t fir_pos;       /* FirPos() after the step */
    uint32_t snd_seq;      /* sound blocks emitted before this step = the seq of this step's wf_pkt_t (rx_waterfall.cpp:1635) */
    int32_t table_bytes;   /* what the step's one upload carried */
    int32_t nmoves;        /* overlapped rings wrapped this step */
} kg_rxbank_step_info;
/* One step: n = adc_samples_per_step samples at d_adc (device) through every receiver.  Enqueue only (the host returns
 * after some thirty launches; kg_rxbank_poll / _sync say when the work is done).  adc_ready_event: a hipEvent_t recorded
 * behind the writer of d_adc, or NULL.  info may be NULL.  A negative return from the PLAN half of the call (bad state, a
 * receiver without a waterfall setting) leaves the bank as it was; an error of the HIP runtime while the step is being
 * enqueued leaves it half-advanced: destroy the bank.  One host thread per bank. */
int kg_rxbank_step(kg_rxbank *bank, const void *d_adc, void *adc_ready_event, kg_rxbank_step_info *info);
/* The host runs at most KG_RXBANK_SLOTS steps ahead of the GPU: kg_rxbank_step(k) returns only when step k - 8 has
 * completed on the device.  An ADC ring of KG_RXBANK_SLOTS + 1 buffers therefore needs NO device-side ordering of its writer:
 * when step k has been enqueued, the buffer step k + 1 will use was last read by step k - 8, which is done (measured: the
 * resident step time with every block copied from pinned host memory, 1.25 ms).  A shorter ring orders its writer with
 * kg_rxbank_adc_done: `stream` (hipStream_t) waits until the readers of the ADC block of the step `steps_back` steps ago are
 * done (1: the last step; 2: the one before it = the buffer a double-buffered ring refills next; at most 8) -- correct, but a
 * wait enqueued on a stream that shares a hardware queue with one of the bank's holds that queue until the old step has
 * completed (measured with two buffers: 1.47 ms per step instead of 1.25). */
#define KG_RXBANK_SLOTS 8
int kg_rxbank_adc_done(kg_rxbank *bank, void *stream, int steps_back);
int kg_rxbank_poll(kg_rxbank *bank);          /* 1 = all streams idle, 0 = busy, <0 error */
/* 1: the next kg_rxbank_step will not wait for its table slot; 0: it would sleep until the step KG_RXBANK_SLOTS back has run --
 * a cooperative host (the reference's coroutine server, NextTask) yields and asks again. */
int kg_rxbank_ready(kg_rxbank *bank);
int kg_rxbank_sync(kg_rxbank *bank);
/* Frame f of the last step belongs to receiver rx_of_frame[f], was read at wf_iq + frame_off[f] pairs, and its packet has
 * pkt_bytes[f] bytes on the wire (arrays of nrx entries, any may be NULL).  Returns nframes. */
int kg_rxbank_frame_map(kg_rxbank *bank, int32_t *rx_of_frame, uint64_t *frame_off, int32_t *pkt_bytes);
/* The bank's device buffers (valid until kg_rxbank_destroy; read them after kg_rxbank_sync or behind the bank's streams). */
typedef struct {
    void *wf_iq;   size_t wf_iq_stride;   /* [nrx][wf_iq_stride] iq_t: the samplers' rows (one-shot: pairs 0..8191) */
    void *wf_rows;                        /* [frame][1024] u8 */
    void *wf_pkts; size_t wf_pkt_stride;  /* [frame][wf_pkt_stride] wf_pkt_t bytes */
    void *rx_raw;  size_t rx_stride;      /* [nrx][rx_stride] rx_iq_t (6 bytes each) */
    void *rx_in;                          /* [nrx][rx_stride] TYPECPX: what CFastFIR was fed */
    void *fir_out; size_t fir_stride;     /* [nrx][fir_stride] TYPECPX */
    void *s16;                            /* [nrx][fir_stride] int16: CAgc / detector output */
    void *adpcm;                          /* [nrx][fir_stride / 2] bytes: the real modes' ADPCM payload */
    void *agc;                            /* [nrx][fir_stride] TYPECPX: the AGC's output (rx->agc_samples_c; every mode but SSB) */
    void *iq_pay;                         /* [nrx][4 fir_stride] bytes: the IQ mode's payload, (s2_t) re, (s2_t) im per sample */
} kg_rxbank_bufs;
int kg_rxbank_buffers(kg_rxbank *bank, kg_rxbank_bufs *out);
/* Where the host's share of the steps since the last call went, by phase (microseconds per step, text). */
int kg_rxbank_host_profile(kg_rxbank *bank, char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* KIWIGPU_H */
