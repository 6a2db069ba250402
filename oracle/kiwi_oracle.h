/*
 * kiwi_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's algorithms for the hot path
 * (flydog-sdr/FlyDog_SDR_GPS: gps/search.cpp, gps/cacode.h, gps/e1bcode.h,
 * rx/rx_waterfall.cpp, rx/CuteSDR/fastfir.cpp, rx/data_pump.cpp).  Each
 * function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call anything in this directory.  The product (libkiwigpu.so) never
 * does, and has no CPU fallback.
 *
 * PINNING STATUS
 *   - PRN generators (ko_cacode): PINNED.  Checked against the reference's own
 *     gps/cacode.h compiled in place (oracle/_ref, see oracle/Makefile) and
 *     against the IS-GPS-200 first-10-chip known answers.
 *   - E1B memory-code unpack (ko_e1b_from_hex): PINNED against the reference's
 *     known answers at gps/search.cpp:295,302 (E01 0xf5d71, E02 0x96b85).
 *   - Everything that goes through an FFT (SearchInit code tables, Sample, the decimators, Correlate; the waterfall's
 *     tables and compute_frame; CFastFIR): PINNED since round 6 by the reference's own gps/search.cpp, rx/rx_waterfall.cpp and
 *     rx/CuteSDR/fastfir.cpp, compiled in place (oracle/build_ref.sh) against the FFTW3 API the image ships -- hipFFTW, in
 *     the seat libfftw3f has in the reference's own build (Makefile:365-366) -- and run on the GPU box
 *     (tools/make_ref_fft_golden.py -> tests/golden/acq_fftref.npz, wf_fftref.npz, fastfir_fftref.npz;
 *     tests/test_ref_pins_cpu.py).  hipFFTW's transform is not FFTW's: spectra are held to 1e-5 of their largest bin (achieved
 *     2e-7), every index / count / byte to equality.  The oracle FFT is a double-precision radix-2 rounded to fp32 on store.
 *   - CAgc, IMA ADPCM, CFir, CSquelch, the CIC shapes, constants: PINNED by the reference's sources built in place.
 *   - Restatement only: the data-pump unpack, the S-meter / AM / NBFM loops of c2s_sound(), CHANNEL::Start, the c2s_waterfall()
 *     parameter formulas; the DDC (FPGA fabric, closed DDS core).
 */
#ifndef KIWI_ORACLE_H
#define KIWI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } ko_cpx;

/* ---- constants pinned from the reference -------------------------------- */
#define KO_NSAMPLES   65536      /* gps/gps.h:69-73  NSAMPLES = FS_I/BIN_SIZE          */
#define KO_DECIM      4          /* gps/gps.h:62                                       */
#define KO_FFT_LEN    16384      /* gps/gps.h:72     FFT_LEN = NSAMPLES/DECIM          */
#define KO_NTAPS      31         /* gps/search.cpp:49                                  */
#define KO_L1_CODELEN 1023       /* kiwi.config:266                                    */
#define KO_E1B_CODELEN 4092      /* kiwi.config:270                                    */
#define KO_SAMPLE_RATE 4092000   /* gps/gps.h:64     SAMPLE_RATE = FS_I/DECIM          */
#define KO_DOP_LO     (-20)      /* gps/search.cpp:465  (int)(-5000/249.755859375)     */
#define KO_DOP_HI     20

/* ---- PRN generators ------------------------------------------------------ */
/* gps/cacode.h:23-53.  t0,t1 <= 10: G2 tap pair; otherwise t1 is the 10-bit G2
 * initial state (QZSS rows of gps/sats.cpp:74-84).  Writes 1023 chips {0,1}. */
void ko_cacode(int t0, int t1, uint8_t *chips);

/* gps/e1bcode.h:66-79: hex memory code -> 4092 chips, MSB-first per nibble.
 * Returns 0, or -1 on a bad hex digit. */
int ko_e1b_from_hex(const char *hex, uint8_t *chips);

/* ---- FFT ------------------------------------------------------------------ */
/* Unnormalised complex DFT, sign=-1 forward (FFTW_FORWARD), +1 backward.
 * n is a power of two.  prec=1: double-precision arithmetic, fp32 on store
 * (the parity oracle).  prec=0: fp32 arithmetic radix-4 Stockham (the "port"
 * used for the CPU-baseline timing; what FFTW3f would do in spirit).
 * Replaces fftwf_execute at gps/search.cpp:280,342,447,481,
 * rx/rx_waterfall.cpp:1291, rx/CuteSDR/fastfir.cpp:274,304. */
void ko_fft(int n, int sign, const ko_cpx *in, ko_cpx *out, int prec);
/* prec: 1 = double radix-2 rounded to fp32 on store, 0 = fp32 radix-4 Stockham, 2 = the registered hook (an
 * independent FFT supplied by the test: unnormalised, sign -1 forward / +1 backward, in may equal out) */
typedef void (*ko_fft_hook)(int n, int sign, const ko_cpx *in, ko_cpx *out);
void ko_set_fft_hook(ko_fft_hook fn);

/* ---- GPS acquisition ------------------------------------------------------ */
/* gps/search.cpp:140-166 DecimateBy2float, in place.  buf must have room for
 * size+KO_NTAPS elements (the tail is zero-filled first, :145).  Returns size/2. */
int ko_decimate_by2_float(int size, ko_cpx *buf);

/* gps/search.cpp:243-285 (boc=0) / :309-346 (boc=1): resample the chip
 * sequence at CPS/FS = 1/16 chip per sample over NSAMPLES samples, Bipolar
 * (0 -> +1, 1 -> -1), 2x DecimateBy2float, forward FFT.  *phase is the
 * ca_phase / e1b_phase accumulator that persists across SVs (:206,:307).
 * out receives FFT_LEN bins (the reference stores them twice, :283-284). */
void ko_code_fft(const uint8_t *chips, int nchips, int boc, float *phase,
                 ko_cpx *out, int prec);

/* Same, stopping before the FFT: the FFT_LEN decimated replica samples. */
void ko_code_replica(const uint8_t *chips, int nchips, int boc, float *phase,
                     ko_cpx *out);

/* gps/search.cpp:382-449 Sample(): 8192 bytes of 1-bit IF, LSB first ->
 * XOR quadrature mix -> DecimateBy2binary -> DecimateBy2float -> forward FFT.
 * out receives FFT_LEN bins.  If td != NULL it receives the FFT_LEN
 * decimated time-domain samples (the FFT input). */
void ko_sample_bits(const uint8_t *packed, ko_cpx *out, ko_cpx *td, int prec);

/* Extension (BASELINE.json configs[1], "synthetic int16 IQ"): NSAMPLES complex
 * int16 samples at the FC = FS/4 IF.  Mix = multiply by (-j)^n (exact), then
 * the same two half-band stages and FFT as Sample(). */
void ko_sample_iq16(const int16_t *iq, ko_cpx *out, ko_cpx *td, int prec);

typedef struct { float snr; int dop; int idx; int valid; } ko_acq_result;
typedef struct { float snr; float max_pwr; float tot_pwr; int idx; } ko_acq_cell;

/* gps/search.cpp:453-499 Correlate().  code: FFT_LEN bins (natural order),
 * data: FFT_LEN bins, limit = SAMPLE_RATE/1000*code_period_ms (4092 L1/QZSS,
 * 16368 E1B; :486).  cells (may be NULL) receives one entry per Doppler bin.
 * valid=0 reproduces "no snr > 0 seen, out-pointers untouched" (:495). */
ko_acq_result ko_correlate(const ko_cpx *code, const ko_cpx *data, int limit,
                           int dop_lo, int dop_hi, ko_acq_cell *cells, int prec);

/* Many SVs: codes[nsv][FFT_LEN]; OpenMP-free pthread shard over SVs when
 * nthreads > 1 (CPU baseline "T_all"). */
void ko_correlate_many(const ko_cpx *codes, int nsv, const ko_cpx *data,
                       const int *limits, int dop_lo, int dop_hi,
                       ko_acq_result *out, ko_acq_cell *cells, int prec,
                       int nthreads);

/* ---- the same acquisition functions for another shape (BASELINE.json configs[4]) ----
 * The reference is one shape: NSAMPLES = 65536 = DECIM * FFT_LEN (gps/gps.h:62-73).  The _n
 * forms restate the same loops with FFT_LEN = fft_len (a power of two) and a sample block of
 * `nsamples` <= DECIM * fft_len input samples followed by zeros (what DecimateBy2float's
 * zero tail, search.cpp:145, already is for the last taps); the code replica covers all
 * DECIM * fft_len samples as the reference's covers NSAMPLES (:250).  configs[4]: nsamples =
 * 163680 (10 ms at FS), fft_len = 65536.  An extension beyond the reference: PARITY UNPINNED
 * by nature; the un-suffixed functions are these with (65536, 16384). */
void ko_code_replica_n(const uint8_t *chips, int nchips, int boc, float *phase, ko_cpx *out, int fft_len);
void ko_code_fft_n(const uint8_t *chips, int nchips, int boc, float *phase, ko_cpx *out, int prec, int fft_len);
void ko_sample_bits_n(const uint8_t *packed, ko_cpx *out, ko_cpx *td, int prec, int nsamples, int fft_len);
void ko_sample_iq16_n(const int16_t *iq, ko_cpx *out, ko_cpx *td, int prec, int nsamples, int fft_len);
ko_acq_result ko_correlate_n(const ko_cpx *code, const ko_cpx *data, int limit, int dop_lo, int dop_hi,
                             ko_acq_cell *cells, int prec, int fft_len);
/* With the row the reference reads BEHIND the satellite's own for a negative Doppler bin (gps/search.cpp:471 over the doubled
 * row of :54: entries 2 N .. 2 N + |dop| - 1 are code[sat + 1][0 .. |dop|)): `next` = that row's spectrum, NULL = a row never
 * written (zeros).  The forms without it are the NULL case.  nexts: [nsv][fft_len] with has_next[nsv] flags, or NULL. */
ko_acq_result ko_correlate_next_n(const ko_cpx *code, const ko_cpx *next, const ko_cpx *data, int limit, int dop_lo, int dop_hi,
                                  ko_acq_cell *cells, int prec, int fft_len);
void ko_correlate_many_next_n(const ko_cpx *codes, const ko_cpx *nexts, const unsigned char *has_next, int nsv, const ko_cpx *data,
                              const int *limits, int dop_lo, int dop_hi, ko_acq_result *out, ko_acq_cell *cells, int prec,
                              int nthreads, int fft_len);
void ko_correlate_many_n(const ko_cpx *codes, int nsv, const ko_cpx *data, const int *limits, int dop_lo,
                         int dop_hi, ko_acq_result *out, ko_acq_cell *cells, int prec, int nthreads,
                         int fft_len);

/* ---- waterfall (rx/rx_waterfall.cpp) ---------------------------------------- */
#define KO_WF_NFFT  8192         /* rx/rx_waterfall.h:61-62 WF_C_NFFT = WF_C_NSAMPS */
#define KO_WF_WIDTH 1024         /* rx/rx_waterfall.h:65 */
enum { KO_WINF_HANNING = 0, KO_WINF_HAMMING = 1, KO_WINF_BLACKMAN_HARRIS = 2, KO_WINF_NONE = 3 };  /* rx_waterfall.h:160-163 */
enum { KO_WF_MAX = 0, KO_WF_MIN, KO_WF_LAST, KO_WF_DROP, KO_WF_CMA };                             /* rx_waterfall.h:116 */

typedef struct {
    int zoom, decim, fft_used, plot_width, plot_width_clamped;
    float start, fft_scale, fft_offset;
    uint64_t i_offset;           /* 48-bit NCO phase increment */
} ko_wf_params;

typedef struct {
    int zoom, window_func, interp, cic_comp, overlapped;
    int fft_used, plot_width, plot_width_clamped;
    const uint16_t *fft2wf_map;  /* [fft_used]  wf_inst_t.fft2wf_map  */
    const uint16_t *drop_sample; /* [1024]      wf_inst_t.drop_sample */
    const float *fft_scale, *fft_scale_div2;   /* [1024] */
    float fft_offset;
    const float *CIC_comp;       /* [8192] */
} ko_wf_cfg;

void ko_wf_window(int winf, float *window);                  /* c2s_waterfall_init :136-171 */
void ko_wf_cic_comp(float *comp);                            /* c2s_waterfall_init :175-185 */
void ko_wf_params_for(int zoom, float start, double adc_clock, double ui_srate,
                      int spectral_inversion, ko_wf_params *out);
void ko_wf_build_maps(int fft_used, int plot_width, int plot_width_clamped,
                      int spectral_inversion, uint16_t *fft2wf_map, uint16_t *drop_sample);
void ko_wf_window_iq(const int16_t *iq, const float *window, ko_cpx *out);   /* sample_wf :1049-1066 */
/* compute_frame :1275-1575.  samps: 8192 windowed samples.  out: 1024 bytes.
 * Optional taps: pwr[fft_used], pwr_out[1024], dB[1024] (before clamp). */
void ko_wf_compute_frame(const ko_wf_cfg *wf, const ko_cpx *samps, uint8_t *out,
                         float *pwr_dbg, float *pwr_out_dbg, float *dB_dbg, int prec);

/* ---- audio front: data_pump unpack, CFastFIR ---------------------------------- */
#define KO_FIR_FFT_SIZE 1024     /* CONV_FFT_SIZE, rx/CuteSDR/cuteSDR.h:12 */
#define KO_FIR_SIZE     513      /* CONV_FIR_SIZE, rx/CuteSDR/fastfir.h:20 */

typedef struct {
    int in_pos;                  /* m_InBufInPos                       */
    ko_cpx buf[KO_FIR_FFT_SIZE]; /* m_pFFTBuf                          */
    ko_cpx overlap[KO_FIR_SIZE]; /* m_pFFTOverlapBuf                   */
} ko_fir_state;

float ko_dpump_rescale(int use_cicf);
void ko_dpump_unpack(const uint8_t *raw, int nsamps, int nchans, const uint8_t *enabled,
                     float rescale, float dc_i, float dc_q, int spectral_inversion,
                     ko_cpx *out, int out_stride);
void ko_fir_cic_coeffs(int snd_rate_3ch, float *cic);
void ko_fir_window(int window_func, float *tbl);
int ko_fir_design(float FLoCut, float FHiCut, float Offset, float SampleRate, const float *window,
                  int do_cic_comp, const float *cic_coeffs, ko_cpx *coef, ko_cpx *coef_cic,
                  ko_cpx *time_coef, int prec);
void ko_fir_reset(ko_fir_state *s);
int ko_fir_process(ko_fir_state *s, const ko_cpx *coef_cic, const ko_cpx *in, int n, ko_cpx *out, int prec);
int ko_fir_process_taps(ko_fir_state *s, const ko_cpx *coef_cic, const float *cic, const ko_cpx *in, int n,
                        ko_cpx *out, int prec, ko_cpx *pre, ko_cpx *post);

/* ---- waterfall DDC (verilog/rx: iq_mixer.v, cic_prune_var.v, cic_wf1.vh) ---- */
typedef struct {
    uint64_t integ[4][2];        /* integrators 1-4, 89 bits kept in 128 (lo, hi)  */
    uint32_t integ5;             /* integrator 5, 28 bits                          */
    int64_t comb_prev[5];        /* cic_comb.v prev_data of combs 1-5              */
} ko_ddc_cic_state;

typedef struct {
    uint64_t phase;              /* 48-bit NCO accumulator (phase of the next sample) */
    uint64_t n;                  /* samples consumed                                   */
    uint32_t sample_no;          /* decimation counter, cic_prune_var.v:65-80          */
    ko_ddc_cic_state cic[2];     /* I and Q                                            */
} ko_ddc_wf_state;

void ko_ddc_nco_table(int16_t *cos_tab, int16_t *sin_tab);     /* 8192 entries each */
void ko_ddc_wf_reset(ko_ddc_wf_state *s);
int ko_ddc_wf(ko_ddc_wf_state *s, const int16_t *adc, long n, uint64_t phase_inc, int log2r,
              int16_t *out);

/* ---- audio DDC (verilog/rx/rx.v, cic_rx1/rx2, fir_iq.sv) ------------------------ */
typedef struct {
    uint64_t phase;              /* 48-bit NCO accumulator                            */
    uint32_t cnt1, cnt2;         /* decimation counters of rx1 (R 1736) and rx2 (R 3) */
    int decim_by_2;              /* fir_iq.sv decim_by_2 toggle                       */
    uint64_t i1[2], i2[2];       /* rx1 integrators 1, 2 (55 bits)                    */
    uint32_t i3[2];              /* rx1 integrator 3 (26 bits)                        */
    int64_t comb1_prev[2][3];    /* rx1 comb registers                                */
    int64_t j[2][5];             /* rx2 integrators (26 bits)                         */
    int64_t comb2_prev[2][5];    /* rx2 comb registers                                */
    int32_t fir_buf[2][65];      /* fir_iq bufI / bufQ, [0] newest                    */
} ko_ddc_rx_state;

void ko_ddc_rx_reset(ko_ddc_rx_state *s);
int ko_ddc_rx(ko_ddc_rx_state *s, const int16_t *adc, long n, uint64_t phase_inc, uint8_t *out);
/* The same for each RX instance the reference builds (kiwi.config:101-105, fir_iq.sv:39-123):
 * KO_RX_STD rx4 / rx8 (1736 x 3 x 2, 65 taps), KO_RX_WIDE rx3 (1543 x 2 x 2, the RX_CFG == 3 taps),
 * KO_RX_14 rx14 (1736 x 3 x 2, the 17-tap RX_CFG == 14 filter). */
enum { KO_RX_STD = 0, KO_RX_WIDE = 1, KO_RX_14 = 2 };
int ko_ddc_rx_mode(ko_ddc_rx_state *s, const int16_t *adc, long n, uint64_t phase_inc, uint8_t *out, int mode);
int ko_ddc_rx_decim(int mode);                   /* total decimation: 10416 / 6172 / 10416 */
extern const int32_t ko_cicf_taps65[33], ko_cicf_taps65_wide[33], ko_cicf_taps17[9];
int ko_ddc_shape(int which, int r, int *o);      /* 0 wf1, 1 rx1 (decimation r), 2 rx2, 3 rx2 wide */

/* ---- part 6: S-meter, CAgc, AM / NBFM detectors (kiwi_oracle_post.c) ---- */
#define KO_AGC_MAX_DELAY_BUF 2048        /* agc.h:16 */
typedef struct {                          /* agc.h:30-64 */
    int agc_on, use_hang, threshold, manual_gain, decay;
    float slope_factor, sample_rate;
    float manual_agc_gain, decay_ave, attack_ave;
    float attack_rise_alpha, attack_fall_alpha, decay_rise_alpha, decay_fall_alpha;
    float fixed_gain, knee, gain_slope, peak;
    int sig_delay_ptr, mag_buf_pos, delay_samples, window_samples, hang_time, hang_timer;
    ko_cpx sig_delay_buf[KO_AGC_MAX_DELAY_BUF];
    float mag_buf[KO_AGC_MAX_DELAY_BUF];
} ko_agc_state;
size_t ko_agc_state_size(void);
void ko_agc_init(ko_agc_state *s);
int ko_agc_delay(const ko_agc_state *s);                  /* CAgc::GetDelaySamples(), agc.h:27 */
void ko_agc_set_parameters(ko_agc_state *s, int agc_on, int use_hang, int threshold, int manual_gain,
                           int slope_factor, int decay, float sample_rate);
void ko_agc_process_cpx(ko_agc_state *s, int n, const ko_cpx *in, ko_cpx *out);
void ko_agc_process_s16(ko_agc_state *s, int n, const ko_cpx *in, int16_t *out);
float ko_smeter_alpha(float frate);
float ko_smeter_process(float avg_dB, float alpha, int n, const ko_cpx *in, float *tap);
void ko_am_detect(double *z1, int n, const ko_cpx *agc, float *demod);
void ko_nbfm_detect(ko_cpx *last, int n, const ko_cpx *agc, float *demod);

/* ---- part 6b: CFir (m_AM_FIR, the de-emphasis filters) and CSquelch (kiwi_oracle_cfir.c) ---- */
#define KO_CFIR_MAX_NUMCOEF 97           /* fir.h:20 */
#define KO_SQ_MAX_SQBUF_SIZE 1024        /* squelch.h:14 */
typedef struct {                          /* fir.h:24-52, the real-valued members */
    int num_taps, state;
    float sample_rate;
    float coef[KO_CFIR_MAX_NUMCOEF * 2];
    float zbuf[KO_CFIR_MAX_NUMCOEF];
} ko_cfir_state;
size_t ko_cfir_state_size(void);
void ko_cfir_init(ko_cfir_state *f);
int ko_cfir_num_taps(const ko_cfir_state *f);
void ko_cfir_get_taps(const ko_cfir_state *f, float *taps);
void ko_cfir_init_const(ko_cfir_state *f, int num_taps, const float *coef, float fs);
int ko_cfir_init_lp(ko_cfir_state *f, int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate);
int ko_cfir_init_hp(ko_cfir_state *f, int NumTaps, float Scale, float Astop, float Fpass, float Fstop, float Fsamprate);
void ko_cfir_process_rr(ko_cfir_state *f, int n, const float *in, float *out);
void ko_cfir_process_rm(ko_cfir_state *f, int n, const float *in, int16_t *out);
void ko_cfir_process_mm(ko_cfir_state *f, int n, const int16_t *in, int16_t *out);
typedef struct {                          /* squelch.h:30-56, what PerformFMSquelch uses */
    int squelch_state, set_squelch;
    float sample_rate, squelch_hp_freq;
    float squelch_value, squelch_threshold, squelch_ave, squelch_alpha;
    ko_cfir_state hp;
} ko_squelch_state;
size_t ko_squelch_state_size(void);
void ko_squelch_init(ko_squelch_state *s);
void ko_squelch_reset(ko_squelch_state *s);
void ko_squelch_setup(ko_squelch_state *s, float samplerate);
void ko_squelch_set(ko_squelch_state *s, int Value, int SquelchMax);
int ko_squelch_is_squelched(const ko_squelch_state *s);
float ko_squelch_ave(const ko_squelch_state *s);
int ko_squelch_perform_fm(ko_squelch_state *s, int n, const float *in, int16_t *out);

/* ---- part 7: wire formats (kiwi_oracle_wire.c) ---- */
void ko_snd_iq_payload(const ko_cpx *in, int n, int little_endian, uint8_t *out);      /* rx_sound.cpp:1076-1096 */
#define KO_WF_WIDTH 1024                 /* rx_waterfall.h:64 */
#define KO_WF_ADPCM_PAD 10               /* rx_waterfall.h:83 */
typedef struct { int index, previous; } ko_adpcm_state;      /* ima_adpcm.h: index, previousValue */
const int *ko_adpcm_step_table(void);
void ko_adpcm_encode_i16(const int16_t *in, uint8_t *out, int n, ko_adpcm_state *s);
void ko_adpcm_encode_u8(const uint8_t *in, uint8_t *out, int n, ko_adpcm_state *s);
void ko_adpcm_decode_i16(const uint8_t *in, int16_t *out, int nbytes, ko_adpcm_state *s);
void ko_adpcm_decode_u8(const uint8_t *in, uint8_t *out, int nbytes, ko_adpcm_state *s);
int ko_wf_packet(const uint8_t *row, uint32_t x_bin_server, uint32_t zoom, uint32_t seq, int use_compression,
                 uint8_t *pkt);
void ko_snd_header(uint8_t flags, uint32_t seq, float smeter_dBm, uint8_t *h);

/* ---- part 8: acquisition hand-off arithmetic, waterfall autoscale (kiwi_oracle_handoff.c) ---- */
typedef struct {
    double lo_dop, ca_dop;
    uint32_t lo_rate, ca_rate, ca_pause;
    int32_t code_creep;
} ko_chan_start_out;
void ko_chan_start(int is_e1b, int lo_shift, int ca_shift, double secs, ko_chan_start_out *o);
void ko_aper_update(float *avg_pwr, const uint8_t *bp, int algo, float param, int clear, int start, int stop,
                    int waterfall_cal);
void ko_aper_report(const float *avg_pwr, int start, int stop, int *signal, int *noise);

/* GPS time stamp of the IQ sound packet (rx/rx_sound.cpp:557, :636-661) */
typedef struct { double gpssec, last_gpssec; int gps_init; int pad; } ko_gps_state;
void ko_snd_gps_begin(ko_gps_state *s, double clk_gps_secs, double dticks, double adc_clock_base,
                      double gps_delay, double gps_delay2);
void ko_snd_gps_stamp(ko_gps_state *s, int norm_nrx_samps, int fir_pos, int agc_on, int agc_delay, int rx_decim,
                      double adc_clock_base, double clk_gps_secs, uint64_t clk_ticks, uint32_t *gpssec,
                      uint32_t *gpsnsec, uint8_t *last_gps_solution);

/* ---- part 11: the platform's log10f restated, and the check of the restatement against libm (kiwi_oracle_libm.c) ---- */
float ko_logf_restated(float x, int fused);
float ko_log10f_restated(float x, int fused);
void ko_libm_log10f(const float *x, float *y, size_t n);
void ko_libm_log10f_bits(uint32_t first, size_t n, float *y);
uint64_t ko_libm_check_range(uint32_t first, uint64_t n, uint64_t step, int fused, int threads, uint64_t *bad_logf,
                             uint64_t *bad_log10f, uint32_t *first_bad);
float ko_powf_restated(float x, float y, int fused);
float ko_expf_restated(float x, int fused, int fused_residual);
void ko_libm_powf_bits(float base, uint32_t first, size_t n, float *y);
void ko_libm_expf_bits(uint32_t first, size_t n, float *y);
void ko_libm_powf(float base, const float *x, float *y, size_t n);
void ko_libm_expf(const float *x, float *y, size_t n);
uint64_t ko_libm_check_pow_exp(uint32_t first, uint64_t n, uint64_t step, int fused, int fused_residual, int threads,
                               uint64_t *bad_pow10, uint64_t *bad_exp, uint64_t *bad_rand, uint64_t *nrand);

#ifdef __cplusplus
}
#endif
#endif
