// Driver for the reference's own data pump unpack, built IN PLACE: this translation unit includes
// /root/reference/rx/data_pump.cpp itself (snd_service() is a static function of that file).  No FFT is involved -- the file only
// needs the FFTW3 API header on the include path because support/shmem.h pulls rx_waterfall.h in (oracle/build_ref.sh) -- so the
// binary runs in the build container (tools/make_ref_golden.py -> tests/golden/dpump_ref.npz).  Test infrastructure only.
//
// data_pump.cpp is the server's interrupt task: it fetches one SPI buffer of rx_iq_t records + the trailer from the FPGA, yields,
// sleeps and wakes the sound tasks.  The driver defines the entry points of the SERVER RUNTIME and hardware that file calls -- the
// SPI read hands over the test's own buffer (records + ticks + the two write counters, as rx_audio_mem.v lays them out), the SPI
// write, the control-register write, the clock and the scheduler calls do nothing -- and the server's configuration globals it
// reads (rx_chans, nrx_samps, nrx_bufs, which channels are enabled, kiwi.spectral_inversion, the DC offsets): this test's inputs.
// The arithmetic -- S24_8_16, `rescale` (its MPOW expression is evaluated by data_pump.cpp's own static initialiser), the
// re / im swap, the ring position and the 48-bit tick assembly -- is the reference's compiled code.
//
//   dpump_ref script.txt in.bin out.bin
// script lines:
//   G rx_chans nrx_samps inversion dc_i dc_q enabled_mask   -> the configuration (data_pump_init() runs)
//   S                       -> one snd_service() over the next buffer of in.bin: nrx_samps * rx_chans records of 6 bytes, then
//                              the 10-byte trailer; appends for every ENABLED channel: wr_pos after, the 48-bit ticks (as two
//                              floats: high 24 bits, low 24 bits) and the nrx_samps complex floats of in_samps[wr_pos before]
//   R                       -> appends `rescale`
#include REF_DATA_PUMP_CPP
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// ---- the server runtime / hardware entry points data_pump.cpp calls: no arithmetic in any of them
static spi_shmem_t the_spi_shmem;
spi_shmem_t *spi_shmem_p = &the_spi_shmem;
static FILE *g_in;
static int g_bytes;
void spi_get3_noduplex(SPI_CMD, SPI_MISO *rx, int bytes, uint16_t, uint16_t, uint16_t)
{
    if (bytes != g_bytes || fread(&rx->word[0], 1, bytes, g_in) != (size_t) bytes) { fprintf(stderr, "dpump_ref: short SPI buffer (%d)\n", bytes); exit(4); }
    rx->status = 0;
}
void _spi_set(SPI_CMD, uint16_t, uint32_t) {}
void ctrl_clr_set(u2_t, u2_t) {}
u4_t timer_us() { return 0; }
extern "C" {
int _CreateTask(funcP_t, const char *, void *, int, u4_t, int) { return 0; }
void *_TaskSleep(const char *, u64_t, u4_t *) { return NULL; }
void _TaskWakeup(int, u4_t, void *) {}
}
// ---- the server's configuration globals it reads: the test's inputs
kiwi_t kiwi;
TYPEREAL DC_offset_I, DC_offset_Q;
int rx_chans, nrx_bufs, nrx_samps, nrx_samps_loop, nrx_samps_rem;
rx_chan_t rx_channels[MAX_RX_CHANS];
bool itask_run = true;

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *outf = fopen(argv[3], "wb");
    g_in = fopen(argv[2], "rb");
    if (!sf || !g_in || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'G') {
            int inv, mask; float dci, dcq;
            if (fscanf(sf, "%d %d %d %f %f %d", &rx_chans, &nrx_samps, &inv, &dci, &dcq, &mask) != 6) return 3;
            kiwi.spectral_inversion = inv != 0; DC_offset_I = dci; DC_offset_Q = dcq;
            nrx_bufs = 8; nrx_samps_loop = 0; nrx_samps_rem = 0;
            for (int ch = 0; ch < MAX_RX_CHANS; ch++) { rx_channels[ch].data_enabled = (mask >> ch) & 1; rx_dpump[ch].wr_pos = rx_dpump[ch].rd_pos = 0; }
            data_pump_init();                              // rx_xfer_size, rxd, rxt (data_pump.cpp:398-419)
            g_bytes = rx_xfer_size;
        } else if (op == 'S') {
            int before[MAX_RX_CHANS];
            for (int ch = 0; ch < rx_chans; ch++) before[ch] = rx_dpump[ch].wr_pos;
            snd_service();
            for (int ch = 0; ch < rx_chans; ch++) {
                if (!rx_channels[ch].data_enabled) continue;
                const u64_t t = rx_dpump[ch].ticks[before[ch]];
                const float hdr[3] = {(float) rx_dpump[ch].wr_pos, (float) (u4_t) ((t >> 24) & 0xffffff), (float) (u4_t) (t & 0xffffff)};
                fwrite(hdr, sizeof(float), 3, outf);
                fwrite(rx_dpump[ch].in_samps[before[ch]], sizeof(TYPECPX), nrx_samps, outf);
            }
        } else if (op == 'R') {
            const float r = rescale;
            fwrite(&r, sizeof r, 1, outf);
        } else return 3;
    }
    fclose(outf);
    return 0;
}
