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
//   * sample_wf()'s unpack + window of ONE frame (rx_waterfall.cpp:1046-1061: fi = (float)(s4_t)(s2_t) iq.i * window[sn]) with
//     the reference's own window table, without the SPI chunk loop around it (whose 9th pass re-windows a stale buffer: SURVEY
//     8(a) W3) -- two lines, restated here.
//
//   wf_ref script.txt in.bin out.bin
// script lines:
//   T                      -> appends window_function[4][8192], CIC_comp[8192] (floats) and n_chunks
//   F zoom window_func interp cic_comp overlapped fft_used plot_width plot_width_clamped fft_offset compression start seq
//                          -> reads fft2wf_map[fft_used] (u16), drop_sample[1024] (u16), fft_scale[1024], fft_scale_div2[1024]
//                             (float), iq[8192][2] (s16) from in.bin; runs compute_frame(0); appends: out_bytes, fft_used_limit,
//                             x_bin_server, flags_x_zoom_server, seq (5 floats), hw_fft[0 .. fft_used) (complex floats) and
//                             the out_bytes packet payload bytes (as floats)
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
            if (fft_used < 1 || fft_used > (int) (sizeof wf->fft2wf_map / sizeof wf->fft2wf_map[0])) return 3;
            if (fread(wf->fft2wf_map, sizeof(u2_t), fft_used, inf) != (size_t) fft_used) return 4;
            if (fread(wf->drop_sample, sizeof(u2_t), WF_WIDTH, inf) != WF_WIDTH) return 4;
            if (fread(wf->fft_scale, sizeof(float), WF_WIDTH, inf) != WF_WIDTH) return 4;
            if (fread(wf->fft_scale_div2, sizeof(float), WF_WIDTH, inf) != WF_WIDTH) return 4;
            static s2_t iq[WF_C_NSAMPS][2];
            if (fread(iq, sizeof iq, 1, inf) != 1) return 4;
            const float *window = S->window_function[winf];
            for (int sn = 0; sn < WF_C_NSAMPS; sn++) {                       // rx_waterfall.cpp:1049-1060
                const s4_t ii = (s4_t) (s2_t) iq[sn][0], qq = (s4_t) (s2_t) iq[sn][1];
                fft->hw_c_samps[sn][0] = ((float) ii) * window[sn];
                fft->hw_c_samps[sn][1] = ((float) qq) * window[sn];
            }
            compute_frame(0);
            const float hdr[5] = {(float) wf->out_bytes, (float) wf->fft_used_limit, (float) wf->out.x_bin_server,
                                  (float) wf->out.flags_x_zoom_server, (float) wf->out.seq};
            fwrite(hdr, sizeof(float), 5, outf);
            fwrite(fft->hw_fft, sizeof(fftwf_complex), fft_used, outf);
            for (int i = 0; i < wf->out_bytes; i++) { const float b = (float) wf->out.un.buf[i]; fwrite(&b, sizeof b, 1, outf); }
        } else return 3;
    }
    fclose(outf);
    return 0;
}
