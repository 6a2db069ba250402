// Driver for the reference's own waterfall frame path, built IN PLACE from /root/reference/rx/rx_waterfall.cpp
// (c2s_waterfall_init(): the four window functions, CIC_comp[], the transform plans; compute_frame(): transform, power, CIC
// compensation, FFT bin -> pixel reduction, dB, quantisation, ADPCM) + rx/csdr/ima_adpcm.cpp, against the FFTW3 API the image
// ships (hipFFTW; oracle/build_ref.sh).  Built here, RUN ON THE GPU BOX (tools/make_ref_fft_golden.py).  Test infrastructure.
//
// What the driver supplies:
//   * str_hash_init(): the command-hash setup c2s_waterfall_init() starts with -- a no-op, no arithmetic;
//   * per frame, the fields of wf_inst_t that the c2s_waterfall() coroutine sets from `SET zoom= start=` before it calls
//     sample_wf() / compute_frame() (rx_waterfall.cpp:756-928: fft_used, plot_width, the fft2wf_map / drop_sample tables,
//     fft_scale[], fft_offset, interp, window_func, cic_comp, overlapped_sampling): they are INPUTS here, read from in.bin
//     as the test wrote them (the product's host mirror builds them; that mirror is compared with the oracle's elsewhere);
//   * for the aperture scenarios (compute_frame() calls the static aperture_auto() when wf->aper == AUTO, :1619-1620): the
//     clock timer_sec() -- the reference's platform/common/timer.cpp reads CLOCK_MONOTONIC and stat()s the installed server
//     binary; here the script says what time it is (an input) -- and the configuration value waterfall_cal that rx_init.cpp reads
//     from kiwi.json (default -13, rx_init.cpp:139, :315; the script gives it).  dB_wire_to_dBm() and qsort_intcomp() are the
//     reference's own (rx/rx_util.cpp, support/misc.cpp, linked in place);
//   * sample_wf()'s unpack + window (row W3) is the reference's own text too: its statements for the samples of one SPI chunk
//     (rx_waterfall.cpp:1046-1066: iqp = miso's words; fi = (float)(s4_t)(s2_t) iqp->i * window[sn]; hw_c_samps[sn] = ...) are cut out of
//     the file at build time (oracle/build_ref.sh, WF_CUT_WINDOW; declarations :1011-1012) and #included below once per chunk of
//     NWF_SAMPS pairs, each chunk copied fresh from the test's frame into a SPI_MISO -- without sample_wf()'s SPI pulls and
//     coroutine sleeps around them, whose ninth pass re-windows a stale buffer.
//
//   wf_ref script.txt in.bin out.bin
// script lines:
//   T                      -> appends window_function[4][8192], CIC_comp[8192] (floats) and n_chunks
//   F zoom window_func interp cic_comp overlapped fft_used plot_width plot_width_clamped fft_offset compression start seq
//                          -> reads fft2wf_map[fft_used] (u16), drop_sample[1024] (u16), fft_scale[1024], fft_scale_div2[1024]
//                             (float), iq[8192][2] (s16) from in.bin; runs compute_frame(0); appends: out_bytes, fft_used_limit,
//                             x_bin_server, flags_x_zoom_server, seq (5 floats), hw_fft[0 .. fft_used) (complex floats) and
//                             the out_bytes packet payload bytes (as floats)
//   P on algo param clear need now waterfall_cal wf_chans
//                          -> the aperture fields of the following F frames (wf->aper = AUTO when on; aper_algo, aper_param;
//                             avg_clear and need_autoscale are SET when clear / need are non-negative; the clock reads `now`);
//                             the averages and counters then persist from frame to frame as they do in the server, and every
//                             F appends after its payload: signal, noise, done_autoscale, report_sec, avg_clear (5 floats) and
//                             avg_pwr[1024]
#include "types.h"           // rx_waterfall.cpp:20-44, in its own order
#include "config.h"
#include "kiwi.h"
#include "clk.h"
#include "misc.h"
#include "nbuf.h"
#include "web.h"
#include "spi.h"
#include "gps.h"
#include "coroutines.h"
#include "debug.h"
#include "data_pump.h"
#include "cfg.h"
#include "datatypes.h"
#include "ext_int.h"
#include "rx_noise.h"
#include "noiseproc.h"
#include "dx.h"
#include "non_block.h"
#include "noise_blank.h"
#include "str.h"
#include "mem.h"
#include "rx_waterfall.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// (wf_shmem_p and its storage are rx_waterfall.cpp's own, :77-78)
void str_hash_init(const char *, str_hash_t *, str_hashes_t *, bool) {}
static u4_t g_now;
u4_t timer_sec() { return g_now; }
int S_meter_cal = -13, waterfall_cal = -13;     // rx/rx_init.cpp:127; the kiwi.json defaults of :139-140, :314-315; P sets waterfall_cal
int wf_chans = 4;                                // main.cpp:65: the firmware mode's waterfall count (rx4.wf4 = 4, rx14.wf0 = 0); P sets it
struct aper_keep { int on, algo, need, done, sent, clear, signal, noise, last_noise, last_signal; float param; u4_t report_sec;
                   float avg_pwr[APER_PWR_LEN]; };
static aper_keep ap;
void c2s_waterfall_init();
void compute_frame(int rx_chan);

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *inf = fopen(argv[2], "rb"), *outf = fopen(argv[3], "wb");
    if (!sf || !inf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    c2s_waterfall_init();
    wf_shmem_t *S = wf_shmem_p;
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'T') {
            fwrite(S->window_function, sizeof(float), N_WF_WINF * WF_C_NSAMPS, outf);
            fwrite(S->CIC_comp, sizeof(float), WF_C_NSAMPS, outf);
            const float nc = (float) S->n_chunks;
            fwrite(&nc, sizeof nc, 1, outf);
        } else if (op == 'P') {
            int on, algo, clear, need, cal, nwf; float param; unsigned now;
            if (fscanf(sf, "%d %d %f %d %d %u %d %d", &on, &algo, &param, &clear, &need, &now, &cal, &nwf) != 8) return 3;
            waterfall_cal = cal; wf_chans = nwf;
            ap.on = on; ap.algo = algo; ap.param = param; g_now = now;
            if (clear >= 0) ap.clear = clear;
            if (need >= 0) ap.need = need;
        } else if (op == 'F') {
            int zoom, winf, interp, cic, ovl, fft_used, pw, pwc, comp, start, seq; float off;
            if (fscanf(sf, "%d %d %d %d %d %d %d %d %f %d %d %d", &zoom, &winf, &interp, &cic, &ovl, &fft_used, &pw, &pwc, &off, &comp,
                       &start, &seq) != 12) return 3;
            wf_inst_t *wf = &S->wf_inst[0];
            fft_t *fft = &S->fft_inst[0];
            memset((void *) wf, 0, sizeof *wf);
            wf->rx_chan = 0; wf->zoom = zoom; wf->window_func = winf; wf->interp = (wf_interp_t) interp; wf->cic_comp = cic != 0;
            wf->overlapped_sampling = ovl != 0; wf->fft_used = fft_used; wf->plot_width = pw; wf->plot_width_clamped = pwc;
            wf->fft_offset = off; wf->compression = comp != 0; wf->start = start; wf->prev_start = -1; wf->prev_zoom = -1;
            wf->snd_seq = (u4_t) seq; wf->aper = MAN; wf->fft_used_limit = 0;
            if (ap.on) {                                                     // what persists in wf_inst_t between frames
                wf->aper = AUTO; wf->aper_algo = ap.algo; wf->aper_param = ap.param; wf->need_autoscale = ap.need;
                wf->done_autoscale = ap.done; wf->sent_autoscale = ap.sent; wf->avg_clear = ap.clear; wf->signal = ap.signal;
                wf->noise = ap.noise; wf->last_noise = ap.last_noise; wf->last_signal = ap.last_signal; wf->report_sec = ap.report_sec;
                memcpy(wf->avg_pwr, ap.avg_pwr, sizeof ap.avg_pwr);
            }
            if (fft_used < 1 || fft_used > (int) (sizeof wf->fft2wf_map / sizeof wf->fft2wf_map[0])) return 3;
            if (fread(wf->fft2wf_map, sizeof(u2_t), fft_used, inf) != (size_t) fft_used) return 4;
            if (fread(wf->drop_sample, sizeof(u2_t), WF_WIDTH, inf) != WF_WIDTH) return 4;
            if (fread(wf->fft_scale, sizeof(float), WF_WIDTH, inf) != WF_WIDTH) return 4;
            if (fread(wf->fft_scale_div2, sizeof(float), WF_WIDTH, inf) != WF_WIDTH) return 4;
            static s2_t iq[WF_C_NSAMPS][2];
            if (fread(iq, sizeof iq, 1, inf) != 1) return 4;
            {   // sample_wf()'s own statements, one SPI chunk at a time
                static SPI_MISO miso_storage;
                SPI_MISO *miso = &miso_storage;
                struct iq_t { u2_t i, q; } __attribute__((packed));       // rx_waterfall.cpp:95-97 (file-local there)
                int k, sn = 0;
#include WF_CUT_WINDOW_DECLS
                for (int chunk = 0; sn < WF_C_NSAMPS; chunk++) {
                    const int left = WF_C_NSAMPS - chunk * NWF_SAMPS, m = left < NWF_SAMPS ? left : NWF_SAMPS;
                    memcpy(&miso->word[0], &iq[chunk * NWF_SAMPS][0], (size_t) m * 4);
#include WF_CUT_WINDOW
                }
            }
            compute_frame(0);
            const float hdr[5] = {(float) wf->out_bytes, (float) wf->fft_used_limit, (float) wf->out.x_bin_server,
                                  (float) wf->out.flags_x_zoom_server, (float) wf->out.seq};
            fwrite(hdr, sizeof(float), 5, outf);
            fwrite(fft->hw_fft, sizeof(fftwf_complex), fft_used, outf);
            for (int i = 0; i < wf->out_bytes; i++) { const float b = (float) wf->out.un.buf[i]; fwrite(&b, sizeof b, 1, outf); }
            if (ap.on) {
                ap.done = wf->done_autoscale; ap.sent = wf->sent_autoscale; ap.clear = wf->avg_clear; ap.signal = wf->signal;
                ap.noise = wf->noise; ap.last_noise = wf->last_noise; ap.last_signal = wf->last_signal; ap.report_sec = wf->report_sec;
                memcpy(ap.avg_pwr, wf->avg_pwr, sizeof ap.avg_pwr);
                const float st[5] = {(float) wf->signal, (float) wf->noise, (float) wf->done_autoscale, (float) wf->report_sec,
                                     (float) wf->avg_clear};
                fwrite(st, sizeof(float), 5, outf);
                fwrite(wf->avg_pwr, sizeof(float), APER_PWR_LEN, outf);
            }
        } else return 3;
    }
    fclose(outf);
    return 0;
}
