// Harness around the reference's OWN STATEMENTS for the signal path of c2s_sound() between CFastFIR and the bytes of the sound packet:
// /root/reference/rx/rx_sound.cpp -- the S-meter loop, the out_samps_s2 bookkeeping, `switch (s->mode)` with the AM detector + m_AM_FIR,
// the NBFM detector + clipper + m_Squelch, the SSB / CW AGC, the de-emphasis filters (lines 676-908); the packet section: the IQ
// modes' AGC and (s2_t) pairs in either byte order, the mono modes' ADPCM / raw payload in either byte order (1035-1140); the
// S-meter field, the flags and the sequence number of the header (1222-1253).  Those statements are the body of a server coroutine
// (no function to call), so oracle/build_ref.sh cuts the line ranges out of the file WHERE IT LIES at build time (sed into a
// temporary directory that is deleted; nothing of the text enters the repository) and this harness #includes the cuts inside a
// function, in the coroutine's own order, with the locals c2s_sound() declares -- by cuts of its own declaration lines:
//     SND_CUT_GPSCONST rx_sound.cpp:92-93      const double gps_delay, gps_week_sec                      (file scope)
//     SND_CUT_DECLS    rx_sound.cpp:244-250    double z1; double frate = ext_update_get_sample_rateHz(); sMeterAlpha; sMeterAvg_dB, sMeter_dBm
//     SND_CUT_PKTINIT  rx_sound.cpp:252-255    "SND" into both headers, s->seq = 0
//     SND_CUT_MASKED   rx_sound.cpp:285        bool masked = false, ...
//     SND_CUT_OVERLOAD rx_sound.cpp:295        bool squelched_overload = false
//     SND_CUT_NORM     rx_sound.cpp:306-319    ref_nrx_samps, norm_nrx_samps, gps_delay2 by firmware mode
//     SND_CUT_FLAGS    rx_sound.cpp:461-482    the SND_FLAG_* bits; per packet: isNBFM, isDRM, IQ_or_DRM_or_stereo, the packet write
//                                              pointers, flags / seq / smeter pointers, do_de_emp
//     SND_CUT_HOOKS    rx_sound.cpp:488-497    u2_t bc = 0; the extension hooks of ext_users[rx_chan]
//     SND_CUT_TICKS    rx_sound.cpp:536-537    the data-pump buffer's 48-bit tick count, its distance to the last GPS solution
//     SND_CUT_GPSSEC   rx_sound.cpp:557        s->gpssec of the buffer
//     SND_CUT_GPSSTAMP rx_sound.cpp:638-661    the FIR / AGC delay correction and the IQ header's GPS stamp (per CFastFIR block)
//     SND_CUT_PATH     rx_sound.cpp:676-908    the path (per CFastFIR block)
//     SND_CUT_PACKET   rx_sound.cpp:1035-1140  the payload (per CFastFIR block)
//     SND_CUT_HEADER   rx_sound.cpp:1222-1253  the header (per packet)
// CAgc, CFir, CSquelch, the ADPCM coder are the reference's own (rx/CuteSDR/agc.cpp, fir.cpp, squelch.cpp, rx/csdr/ima_adpcm.cpp linked
// in place), time_diff48() its support/timing.cpp, the de-emphasis tables its rx_filter.h, mode_flags[] its mode.h, the packet structs its rx_sound.h.  Test infrastructure.
//
// What the harness supplies (no arithmetic): the objects rx_sound.cpp:148-160 defines (m_Agc, m_Squelch, m_AM_FIR, m_nfm_deemp_FIR,
// m_am_ssb_deemp_FIR; the two CFastFIR arrays are only DECLARED -- their constructors plan transforms, and the SAM case that names
// one is never selected -- and m_RsId is a declaration of the one member the path names behind `if (kiwi.dbgUs)`, never true);
// snd_inst[] / ext_users[] / dpump / kiwi and one rx_dpump_t, iq_buf_t, wf_inst_t, conn_t as zeroed storage; the loop around the
// path (`do { ... } while (bc < LOOP_BC)`: the script says how many blocks a packet takes, as the FIR's output cadence does in
// the server); ext_update_get_sample_rateHz() returning the script's rate; S_meter_cal (kiwi.json, default -13); a receive_S_meter
// hook that records the two taps; the configuration calls rx_sound_cmd.cpp makes on the client's SET commands as script lines.
//
//   sndpath_ref script.txt in.bin out.bin
// script lines (the first must be R):
//   R rate fw_sel nrx_samps rx_decim adc_clock_base -> configuration (the firmware mode, its buffer size, the audio decimation, the ADC clock);
//                                                     the cut declarations run; appends norm_nrx_samps and gps_delay2 (three floats that sum to it) (z1 = 0, sMeterAlpha from rate, sMeterAvg_dB = 0); squelch SetupParameters
//   A on hang thresh manGain slope decay           -> m_Agc[0].SetParameters(..., rate)                 (rx_sound_cmd.cpp:351)
//   L hbw stop                                     -> m_AM_FIR[0].InitLPFilter(0, 1.0, 50.0, hbw, stop, rate)   (rx_sound_cmd.cpp:282)
//   Q value max                                    -> m_Squelch[0].SetSquelch(value, max)                (rx_sound_cmd.cpp:430)
//   E deemp deemp_nfm                              -> s->deemp / s->deemp_nfm and the InitConstFir calls of rx_sound_cmd.cpp:586-640
//   M mode                                         -> s->mode (mode.h numbering)
//   W compression little_endian                    -> s->compression, s->little_endian (`SET compression=`, `SET little-endian`)
//   V overflow                                     -> dpump.rx_adc_ovfl (the ADC overflow the data pump saw)
//   C clk_ticks clk_gps_secs                       -> clk.ticks, clk.gps_secs: the last GPS solution (clk_ticks 0: none yet)
//   T ticks fir_pos  (one per block of the next P) -> the data-pump buffer's tick count and CFastFIR's FirPos() for that block
//   P n1 [n2 ...]                                  -> ONE PACKET of CFastFIR blocks of n1, n2, ... complex floats from in.bin.  Appends, per
//                                                     block: sMeterAvg_dB, sMeter_dBm, tap0, tap1, s->squelched, then n outputs as floats:
//                                                     out_samps_s2 (mono modes) or re, im pairs after the IQ modes' AGC (2 n floats);
//                                                     then per packet: header size, bc, and the header + bc payload bytes as floats
#include "types.h"           // rx_sound.cpp:20-64 in its own order (rsid.h, the RSID decoder's DRM resampler headers, left out)
#include "options.h"
#include "config.h"
#include "kiwi.h"
#include "mode.h"
#include "printf.h"
#include "rx.h"
#include "rx_util.h"
#include "clk.h"
#include "mem.h"
#include "misc.h"
#include "str.h"
#include "timer.h"
#include "nbuf.h"
#include "web.h"
#include "spi.h"
#include "gps.h"
#include "coroutines.h"
#include "cuteSDR.h"
#include "rx_noise.h"
#include "teensy.h"
#include "agc.h"
#include "fir.h"
#include "iir.h"
#include "squelch.h"
#include "debug.h"
#include "data_pump.h"
#include "cfg.h"
#include "mongoose.h"
#include "ima_adpcm.h"
#include "ext_int.h"
#include "fastfir.h"
#include "noiseproc.h"
#include "lms.h"
#include "dx.h"
#include "noise_blank.h"
#include "rx_sound.h"
#include "rx_sound_cmd.h"
#include "rx_waterfall.h"
#include "rx_filter.h"
#include "wdsp.h"
#include "fpga.h"
#include "rf_attn.h"
#include "timing.h"
#undef printf
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include SND_CUT_GPSCONST
snd_t snd_inst[MAX_RX_CHANS];                    // rx_sound.cpp:87
clk_t clk;                                       // init/clk.cpp:40
int fw_sel, nrx_samps, rx_decim;                 // main.cpp:65, config.h:51-52
CAgc m_Agc[MAX_RX_CHANS];                        // :148-160
CSquelch m_Squelch[MAX_RX_CHANS];
extern CFastFIR m_chan_null_FIR[MAX_RX_CHANS];   // (declared only: see the head of this file)
CFir m_AM_FIR[MAX_RX_CHANS];
CFir m_nfm_deemp_FIR[MAX_RX_CHANS];
CFir m_am_ssb_deemp_FIR[MAX_RX_CHANS];
int S_meter_cal = -13;                           // rx/rx_init.cpp:127, :140, :314
ext_users_t ext_users[MAX_RX_CHANS];
dpump_t dpump;
kiwi_t kiwi;
struct rsid_never { void receive(int, TYPEMONO16 *) {} };
static rsid_never m_RsId[MAX_RX_CHANS];          // (see the head of this file)
extern "C" void _TaskWakeup(int, u4_t, void *) {}
static double g_rate;
double ext_update_get_sample_rateHz(int) { return g_rate; }
static float g_tap[2]; static int g_ntap;
static void smeter_hook(int, float v) { if (g_ntap < 2) g_tap[g_ntap] = v; g_ntap++; }

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *inf = fopen(argv[2], "rb"), *outf = fopen(argv[3], "wb");
    if (!sf || !inf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    const int rx_chan = 0;
    snd_t *s = &snd_inst[rx_chan];
    rx_dpump_t *rx = (rx_dpump_t *) calloc(1, sizeof(rx_dpump_t));
    iq_buf_t *iq = (iq_buf_t *) calloc(1, sizeof(iq_buf_t));
    wf_inst_t *wf = (wf_inst_t *) calloc(1, sizeof(wf_inst_t));
    conn_t *conn = (conn_t *) calloc(1, sizeof(conn_t));
    int j;
    static TYPECPX fir_buf[FASTFIR_OUTBUF_SIZE];
    char line[1024];
    double adc_base;
    if (!fgets(line, sizeof line, sf) || sscanf(line, "R %lf %d %d %d %lf", &g_rate, &fw_sel, &nrx_samps, &rx_decim, &adc_base) != 5) return 3;
    clk.adc_clock_base = adc_base;
    std::vector<std::pair<unsigned long long, int> > tq;        // the T lines waiting for their blocks
    ext_users[rx_chan].receive_S_meter = smeter_hook;
    s->compression = 1; s->agc = 1;                              // rx_sound.cpp:237-238
#include SND_CUT_DECLS
#include SND_CUT_PKTINIT
#include SND_CUT_MASKED
#include SND_CUT_OVERLOAD
#include SND_CUT_NORM
    (void) ref_nrx_samps;
    {   // what the firmware-mode switch gave: norm_nrx_samps, and gps_delay2 (a double) as three floats that sum to it
        const float a = (float) gps_delay2, b = (float) (gps_delay2 - (double) a), c = (float) (gps_delay2 - (double) a - (double) b);
        const float cfg[4] = {(float) norm_nrx_samps, a, b, c};
        fwrite(cfg, sizeof(float), 4, outf);
    }
    (void) masked_area; (void) check_masked;
    m_Squelch[rx_chan].SetupParameters(rx_chan, frate);          // rx_sound.cpp:261-262
    m_Squelch[rx_chan].SetSquelch(0, 0);
    s->mode = MODE_USB;
    while (fgets(line, sizeof line, sf)) {
        const char op = line[0];
        if (op == 'A') {
            int on, hang, thr, man, slope, decay;
            if (sscanf(line + 1, "%d %d %d %d %d %d", &on, &hang, &thr, &man, &slope, &decay) != 6) return 3;
            s->agc = on;                                         // rx_sound_cmd.cpp:343
            m_Agc[rx_chan].SetParameters(on, hang, thr, man, slope, decay, frate);
        } else if (op == 'L') {
            float hbw, stop;
            if (sscanf(line + 1, "%f %f", &hbw, &stop) != 2) return 3;
            m_AM_FIR[rx_chan].InitLPFilter(0, 1.0, 50.0, hbw, stop, frate);
        } else if (op == 'Q') {
            int v, mx;
            if (sscanf(line + 1, "%d %d", &v, &mx) != 2) return 3;
            m_Squelch[rx_chan].SetSquelch(v, mx);
        } else if (op == 'E') {
            int de, de_nfm;
            if (sscanf(line + 1, "%d %d", &de, &de_nfm) != 2) return 3;
            s->deemp = de; s->deemp_nfm = de_nfm;
            const bool r12k = fabs(frate - 12000.0) < fabs(frate - 20250.0);         // snd_rate == SND_RATE_4CH
            if (de)     m_am_ssb_deemp_FIR[rx_chan].InitConstFir(N_DEEMP_TAPS, r12k ? am_ssb_deemp_12000[de - 1] : am_ssb_deemp_20250[de - 1], frate);
            if (de_nfm) m_nfm_deemp_FIR[rx_chan].InitConstFir(N_DEEMP_TAPS, r12k ? nfm_deemp_12000[de_nfm - 1] : nfm_deemp_20250[de_nfm - 1], frate);
        } else if (op == 'M') {
            int mode;
            if (sscanf(line + 1, "%d", &mode) != 1) return 3;
            s->mode = mode;
        } else if (op == 'W') {
            int comp, le;
            if (sscanf(line + 1, "%d %d", &comp, &le) != 2) return 3;
            s->compression = comp; s->little_endian = le != 0;
        } else if (op == 'V') {
            int ov;
            if (sscanf(line + 1, "%d", &ov) != 1) return 3;
            dpump.rx_adc_ovfl = ov;
        } else if (op == 'C') {
            unsigned long long t; double secs;
            if (sscanf(line + 1, "%llu %lf", &t, &secs) != 2) return 3;
            clk.ticks = t; clk.gps_secs = secs;
        } else if (op == 'T') {
            unsigned long long t; int fp;
            if (sscanf(line + 1, "%llu %d", &t, &fp) != 2) return 3;
            tq.push_back(std::make_pair(t, fp));
        } else if (op == 'P') {
            // one pass of the `while (TRUE)` loop of c2s_sound() from :459 on: one packet
#include SND_CUT_FLAGS
#include SND_CUT_HOOKS
            (void) isDRM; (void) receive_iq_pre_fir; (void) receive_iq_pre_agc; (void) receive_iq_pre_agc_tid; (void) bp_real_s2; (void) bp_iq_s2;
            int ns_out;
            char *q = line + 1;
            for (;;) {                                           // do { ... } while (bc < LOOP_BC): the script's block list
                char *e;
                ns_out = (int) strtol(q, &e, 10);
                if (e == q) break;
                q = e;
                if (ns_out < 1 || ns_out > FASTFIR_OUTBUF_SIZE) return 3;
                if (fread(fir_buf, sizeof(TYPECPX), ns_out, inf) != (size_t) ns_out) return 4;
                TYPECPX *fir_samps_c = fir_buf;
                g_ntap = 0; g_tap[0] = g_tap[1] = 0;
                int fir_pos = 0;
                rx->rd_pos = 0; rx->ticks[0] = 0;
                if (!tq.empty()) { rx->ticks[0] = tq.front().first; fir_pos = tq.front().second; tq.erase(tq.begin()); }
                {
#include SND_CUT_TICKS
#include SND_CUT_GPSSEC
#include SND_CUT_GPSSTAMP
                    (void) dt_to_pos_sol;
#include SND_CUT_PATH
                    const float hdr[5] = {sMeterAvg_dB, sMeter_dBm, g_tap[0], g_tap[1], (float) s->squelched};
                    fwrite(hdr, sizeof(float), 5, outf);
                    if (!IQ_or_DRM_or_stereo)
                        for (int i = 0; i < ns_out; i++) { const float v = (float) out_samps_s2[i]; fwrite(&v, sizeof v, 1, outf); }
#include SND_CUT_PACKET
                    if (IQ_or_DRM_or_stereo) fwrite(fir_buf, sizeof(TYPECPX), ns_out, outf);      // after m_Agc (:1052)
                }
            }
#include SND_CUT_HEADER
            const u1_t *pkt = IQ_or_DRM_or_stereo ? (const u1_t *) &s->out_pkt_iq : (const u1_t *) &s->out_pkt_real;
            const int hsize = IQ_or_DRM_or_stereo ? (int) sizeof(s->out_pkt_iq.h) : (int) sizeof(s->out_pkt_real.h);
            const float sz[2] = {(float) hsize, (float) bc};
            fwrite(sz, sizeof(float), 2, outf);
            for (int i = 0; i < hsize + bc; i++) { const float v = (float) pkt[i]; fwrite(&v, sizeof v, 1, outf); }
        } else if (op != '\n' && op != '#') return 3;
    }
    fclose(outf);
    return 0;
}
