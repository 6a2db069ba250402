// Driver for the reference's own CHANNEL::Start (the hand-off from Correlate() to a tracking channel), built
// IN PLACE from /root/reference/gps/channel.cpp + ephemeris.cpp + sats.cpp (oracle/build_ref.sh).  No transform is involved; runs
// in the build container (tools/make_ref_golden.py -> tests/golden/chan_ref.npz).  Test infrastructure only.
//
// ChanStart() programs the FPGA: its RESULTS are the SPI commands it sends.  The driver's _spi_set() records them (command,
// 16-bit parameter, 32-bit parameter) -- that is the output -- and the other entry points of the server runtime this path calls
// do nothing: the sleep, the wake-up, the status display, the log printers.  timer_us() answers the time the test asks for, so
// that `secs`, the time since the samples were taken (channel.cpp:293), is an input.
//
//   chan_ref script.txt out.bin
// script lines:
//   S ch sat t_sample_us now_us lo_shift ca_shift snr   -> ChanStart(ch, sat, t_sample, lo_shift, ca_shift, snr) with timer_us() == now:
//                                                          appends the number of SPI commands, then (cmd, wparam, lparam) each
#include "types.h"
#include "config.h"
#include "kiwi.h"
#include "gps.h"
#include "spi.h"
#include "coroutines.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static std::vector<double> g_cmds;
static u4_t g_now;
void _spi_set(SPI_CMD cmd, uint16_t wparam, uint32_t lparam) { g_cmds.push_back((double) cmd); g_cmds.push_back((double) wparam); g_cmds.push_back((double) lparam); }
u4_t timer_us() { return g_now; }
void GPSstat(STAT, double, int, int, int, int, double) {}
void lprintf(const char *, ...) {}
void alt_printf(const char *, ...) {}
extern "C" {
void *_TaskSleep(const char *, u64_t, u4_t *) { return NULL; }
void _TaskWakeup(int, u4_t, void *) {}
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s script out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *outf = fopen(argv[2], "wb");
    if (!sf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op != 'S') return 3;
        int ch, sat, t_sample, lo_shift, ca_shift, snr; unsigned now;
        if (fscanf(sf, "%d %d %d %u %d %d %d", &ch, &sat, &t_sample, &now, &lo_shift, &ca_shift, &snr) != 7) return 3;
        g_now = now;
        g_cmds.clear();
        ChanStart(ch, sat, t_sample, lo_shift, ca_shift, snr);
        const double n = (double) (g_cmds.size() / 3);
        fwrite(&n, sizeof n, 1, outf);
        fwrite(g_cmds.data(), sizeof(double), g_cmds.size(), outf);
    }
    fclose(outf);
    return 0;
}
