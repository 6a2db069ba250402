// Driver for the reference's own rx_sound_set_freq() (the audio NCO's 48-bit phase increment), built IN PLACE from
// /root/reference/rx/rx_sound_cmd.cpp (oracle/build_ref.sh), and for the passband part of its `SET mod= low_cut= high_cut=` handler:
// the clamp to the Nyquist limit, the normalised passband, the half bandwidth and stop frequency of the post-AM-detector filter and
// its design (rx_sound_cmd.cpp:243-272, 276-286 -- statements inside rx_sound_cmd(), cut out of the file at build time into a
// temporary directory and #included below; the three lines between the cuts are the two CFastFIR::SetupParameters calls, which take
// the clamped s->locut / s->hicut this driver reports; m_AM_FIR is the reference's own CFir, rx/CuteSDR/fir.cpp in place).
// Test infrastructure only.
//
// What the driver supplies (no arithmetic): the configuration globals the function reads (ui_srate, do_sdr) and an spi_set3() that
// RECORDS the words it is given (the reference's is the SPI driver); every other symbol of rx_sound_cmd.cpp -- the `SET` command
// handler and what it calls -- is never reached and stays unresolved at link time.
//
//   sndcmd_ref script.txt out.bin
// script lines:
//   F freq_kHz adc_clock_corrected ui_srate spectral_inversion    -> s->freq, conn->adc_clock_corrected, ui_srate, s->spectral_inversion;
//                                                                    rx_sound_set_freq(conn, s); appends (doubles) the SPI command
//                                                                    tag (1 = CmdSetRXFreq), wparam, lparam (i_phase >> 16), w2param
//                                                                    (i_phase & 0xffff)
//   B low_cut high_cut frate                                       -> the handler's passband statements for a client that sent these cuts;
//                                                                    appends (doubles) s->locut, s->hicut, s->norm_locut, s->norm_hicut,
//                                                                    s->norm_pbc, conn->half_bw and the 128 outputs of an impulse through
//                                                                    m_AM_FIR (= its taps, then zeros)
#include "types.h"
#include "config.h"
#include "kiwi.h"
#include "mode.h"
#include "rx.h"
#include "clk.h"
#include "misc.h"
#include "spi.h"
#include "cuteSDR.h"
#include "agc.h"
#include "fir.h"
#include "iir.h"
#include "squelch.h"
#include "data_pump.h"
#include "ext_int.h"
#include "fastfir.h"
#include "noiseproc.h"
#include "rx_sound.h"
#include "rx_sound_cmd.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

CFir m_AM_FIR[MAX_RX_CHANS];                      // rx_sound.cpp:157
double ui_srate;
int do_sdr = 1;
static double g_rec[4];
void spi_set3(SPI_CMD cmd, uint16_t wparam, uint32_t lparam, uint16_t w2param)
{
    g_rec[0] = cmd == CmdSetRXFreq ? 1.0 : 1000.0 + (double) cmd; g_rec[1] = wparam; g_rec[2] = lparam; g_rec[3] = w2param;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s script out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *outf = fopen(argv[2], "wb");
    if (!sf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    conn_t *conn = (conn_t *) calloc(1, sizeof(conn_t));
    snd_t *s = (snd_t *) calloc(1, sizeof(snd_t));
    double f, adc, sr; int inv;
    char op;
    conn->rx_channel = 3;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'F') {
            if (fscanf(sf, "%lf %lf %lf %d", &f, &adc, &sr, &inv) != 4) return 3;
            s->freq = f; conn->adc_clock_corrected = adc; ui_srate = sr; s->spectral_inversion = inv != 0;
            g_rec[0] = g_rec[1] = g_rec[2] = g_rec[3] = -1;
            rx_sound_set_freq(conn, s);
            fwrite(g_rec, sizeof(double), 4, outf);
        } else if (op == 'B') {
            double _locut, _hicut, frate;                     // rx_sound_cmd.cpp:159 (double _freq, _locut, _hicut), :101 (double frate)
            if (fscanf(sf, "%lf %lf %lf", &_locut, &_hicut, &frate) != 3) return 3;
            const int rx_chan = conn->rx_channel;
            s->locut = s->hicut = 0;                          // a fresh connection (memset at rx_sound.cpp:236)
#include SNDCMD_CUT_PB_A
#include SNDCMD_CUT_PB_B
            const double o[6] = {s->locut, s->hicut, s->norm_locut, s->norm_hicut, s->norm_pbc, conn->half_bw};
            fwrite(o, sizeof(double), 6, outf);
            TYPEREAL in[128] = {0}, out[128];
            in[0] = 1.0f;
            if (!no_pb_change) m_AM_FIR[rx_chan].ProcessFilter(128, in, out); else memset(out, 0, sizeof out);
            for (int i = 0; i < 128; i++) { const double v = out[i]; fwrite(&v, sizeof v, 1, outf); }
        } else return 3;
    }
    fclose(outf);
    return 0;
}
