// Driver for the reference's own rx_sound_set_freq() (the audio NCO's 48-bit phase increment), built IN PLACE from
// /root/reference/rx/rx_sound_cmd.cpp (oracle/build_ref.sh).  Test infrastructure only.
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
    while (fscanf(sf, " F %lf %lf %lf %d", &f, &adc, &sr, &inv) == 4) {
        s->freq = f; conn->adc_clock_corrected = adc; ui_srate = sr; s->spectral_inversion = inv != 0;
        conn->rx_channel = 3;
        g_rec[0] = g_rec[1] = g_rec[2] = g_rec[3] = -1;
        rx_sound_set_freq(conn, s);
        fwrite(g_rec, sizeof(double), 4, outf);
    }
    fclose(outf);
    return 0;
}
