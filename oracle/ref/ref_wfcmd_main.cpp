// Harness around the reference's OWN STATEMENTS for what the waterfall connection derives from `SET zoom= start=` / `SET zoom= cf=`:
// the decimation and NCO words it sends to the FPGA and the tables compute_frame() reads -- fft_used, plot_width,
// fft2wf_map[], drop_sample[], wf2fft_map[], fft_scale[], fft_scale_div2[], fft_offset (row W6's map).  Those statements are inside
// c2s_waterfall(), a server coroutine (no function to call), so oracle/build_ref.sh cuts the line ranges out of
// /root/reference/rx/rx_waterfall.cpp WHERE IT LIES at build time (sed into a temporary directory that is deleted; nothing of the text
// enters the repository) and this harness #includes the cuts inside a function, in the coroutine's own order:
//     WF_CUT_MACROS   rx_waterfall.cpp:67-69     MAX_FFT_USED, MAX_START(z)
//     WF_CUT_BITS     rx_waterfall.cpp:217-221   the CMD_* bits of cmd_recv
//     WF_CUT_LOCALS   rx_waterfall.cpp:253-269   c2s_waterfall()'s locals (HZperStart, spectral_inversion, zoom = -1, start = -1, ...)
//     WF_CUT_INIT     rx_waterfall.cpp:271-283   wf = &WF_SHMEM->wf_inst[rx_chan], its initial fields, n_chunks
//     WF_CUT_ZOOM     rx_waterfall.cpp:365-529   `case CMD_SET_ZOOM: { ... }`: the command parser, decim, samp_wait, i_offset, the SPI calls
//     WF_CUT_MAPS     rx_waterfall.cpp:756-928   fft_used, plot_width, `if (new_map) { ... }`, `if (new_scale_mask) { ... }`
// Test infrastructure only.
//
// What the harness supplies (no arithmetic): zeroed conn_t / wf_shmem_t storage; the configuration globals (ui_srate, kiwi, wf_chans,
// waterfall_cal, dx's masked list, cfg_cfg); spi_set / spi_set3 that RECORD what they are given (the reference's are the SPI driver);
// empty send_msg / conn_other (no second connection) / rx_server_send_config; a clock that stands still; kiwi_str_begins_with is the
// reference's own (support/str.cpp, linked in place).
//
//   wfcmd_ref script.txt out.bin
// script lines (the first must be C):
//   C adc_clock_corrected ui_srate spectral_inversion n_chunks     -> configuration; the LOCALS and INIT cuts run
//   X lo hi                                                       -> one more masked range (dx.masked_list), Hz
//   K <the client's command, e.g. SET zoom=3 start=12345.0>       -> the ZOOM cut under `switch (key)`, then the MAPS cut; appends (doubles):
//        nspi, then per recorded SPI call: 1 for CmdSetWFDecim / 2 for CmdSetWFFreq (else 1000 + the command), wparam, lparam, w2param;
//        zoom, start, samp_wait_ms, chunk_wait_us, fft_used, plot_width, plot_width_clamped, fft_offset, fft_used_limit, new_map (was the
//        map rebuilt: the zoom changed) and, if it was, fft2wf_map[fft_used] and drop_sample[plot_width_clamped] (fft_used >= plot_width)
//        or wf2fft_map[plot_width_clamped];
//        fft_scale[plot_width_clamped], fft_scale_div2[plot_width_clamped]
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
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include WF_CUT_MACROS
#include WF_CUT_BITS
static wf_shmem_t shmem_storage;
wf_shmem_t *wf_shmem_p = &shmem_storage;        // rx_waterfall.cpp:77-78
snd_t snd_inst[MAX_RX_CHANS];
kiwi_t kiwi;
double ui_srate;
int wf_chans = 4, waterfall_cal = -13;
dxlist_t dx;
cfg_t cfg_cfg;
struct spi_rec { double cmd, w, l, w2; };
static std::vector<spi_rec> g_spi;
static double tag(SPI_CMD c) { return c == CmdSetWFDecim ? 1.0 : c == CmdSetWFFreq ? 2.0 : 1000.0 + (double) c; }   // by the reference's own enum (kiwi.gen.h)
void _spi_set(SPI_CMD cmd, uint16_t wparam, uint32_t lparam) { g_spi.push_back({tag(cmd), (double) wparam, (double) lparam, -1.0}); }
void spi_set3(SPI_CMD cmd, uint16_t wparam, uint32_t lparam, uint16_t w2param) { g_spi.push_back({tag(cmd), (double) wparam, (double) lparam, (double) w2param}); }
void send_msg(conn_t *, bool, const char *, ...) {}
conn_t *conn_other(conn_t *, int) { return NULL; }
void rx_server_send_config(conn_t *) {}
u4_t timer_ms() { return 0; }
u4_t timer_sec() { return 0; }

static void put(FILE *f, double v) { fwrite(&v, sizeof v, 1, f); }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s script out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *outf = fopen(argv[2], "wb");
    if (!sf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    char line[512];
    double adc; int inv, nch;
    if (!fgets(line, sizeof line, sf) || sscanf(line, "C %lf %lf %d %d", &adc, &ui_srate, &inv, &nch) != 4) return 3;
    kiwi.spectral_inversion = inv != 0;
    shmem_storage.n_chunks = nch;
    conn_t *conn = (conn_t *) calloc(1, sizeof(conn_t));
    conn->adc_clock_corrected = adc; conn->isWF_conn = true; conn->rx_channel = 0;
    static dx_mask_t masks[64];
    dx.masked_list = masks; dx.masked_len = 0;
    int rx_chan = conn->rx_channel;
    wf_inst_t *wf;
#include WF_CUT_LOCALS
#include WF_CUT_INIT
    (void) k; (void) n; (void) wband; (void) _wband; (void) scale; (void) _scale; (void) _speed; (void) cmap; (void) aper; (void) algo;
    (void) _dvar; (void) _pipe; (void) aper_param; (void) tr_cmds; (void) adc_clock_corrected; (void) n_chunks;
    while (fgets(line, sizeof line, sf)) {
        if (line[0] == 'X') {
            int lo, hi;
            if (sscanf(line, "X %d %d", &lo, &hi) != 2 || dx.masked_len >= 64) return 3;
            masks[dx.masked_len].masked_lo = lo; masks[dx.masked_len].masked_hi = hi; dx.masked_len++;
            continue;
        }
        if (line[0] != 'K' || line[1] != ' ') return 3;
        char *cmd = line + 2;
        cmd[strcspn(cmd, "\r\n")] = 0;
        g_spi.clear();
        bool did_cmd = false;
        u2_t key = CMD_SET_ZOOM;
        switch (key) {
#include WF_CUT_ZOOM
        default: break;
        }
        (void) did_cmd;
        if (zoom == -1 || start < 0) return 5;          // (the coroutine waits for CMD_ALL before it goes on, :736-754)
        const bool had_new_map = new_map;
#include WF_CUT_MAPS
        put(outf, (double) g_spi.size());
        for (auto &r : g_spi) { put(outf, r.cmd); put(outf, r.w); put(outf, r.l); put(outf, r.w2); }
        put(outf, wf->zoom); put(outf, wf->start); put(outf, wf->samp_wait_ms); put(outf, wf->chunk_wait_us); put(outf, wf->fft_used);
        put(outf, wf->plot_width); put(outf, wf->plot_width_clamped); put(outf, wf->fft_offset); put(outf, wf->fft_used_limit);
        put(outf, had_new_map);
        if (had_new_map && wf->fft_used >= wf->plot_width) {
            for (i = 0; i < wf->fft_used; i++) put(outf, wf->fft2wf_map[i]);
            for (i = 0; i < wf->plot_width_clamped; i++) put(outf, wf->drop_sample[i]);
        } else if (had_new_map)
            for (i = 0; i < wf->plot_width_clamped; i++) put(outf, wf->wf2fft_map[i]);
        for (i = 0; i < wf->plot_width_clamped; i++) put(outf, wf->fft_scale[i]);
        for (i = 0; i < wf->plot_width_clamped; i++) put(outf, wf->fft_scale_div2[i]);
    }
    fclose(outf);
    return 0;
}
