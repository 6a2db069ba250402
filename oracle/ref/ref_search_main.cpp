// Driver for the reference's own acquisition path, built IN PLACE: this translation unit includes
// /root/reference/gps/search.cpp itself (its Sample(), Correlate() and decimators are static functions of that file), and links
// the reference's gps/sats.cpp and support/simd.cpp, against the FFTW3 API the image ships (hipFFTW; oracle/build_ref.sh).
// Built here, RUN ON THE GPU BOX (hipFFTW executes on the GPU; tools/make_ref_fft_golden.py).  Test infrastructure only.
//
// search.cpp is a coroutine of a server: it yields to the scheduler, sleeps, prints to the server's log, fetches its samples
// over SPI from the FPGA and starts a task.  The driver defines those entry points of the SERVER RUNTIME -- none of them
// computes anything: the scheduler calls return at once, the printer is silent, the SPI read hands over the test's own sample
// bytes in the reference's 512-byte packets (gps/search.cpp:389-404), the task is not started.  Everything numerical --
// COEF[][], the C/A and E1B code generators, the code tables of SearchInit(), the bit mixing and decimation of Sample(), the
// conjugate product, the transforms' call sequence, the power scan and its comparisons -- is the reference's compiled code.
//
//   search_ref init                         -> SearchInit(): prints MAX_SATS-independent facts (sat count) to stderr
//   search_ref script.txt in.bin out.bin
// script lines (SearchInit() has run):
//   T sat            -> appends code[sat][0 .. FFT_LEN) (complex floats)                      (row G3)
//   S                -> Sample() over the next 8192 bytes of in.bin (65536 one-bit samples, LSB first): appends the
//                       decimated time-domain block is NOT observable (fwd_buf is transformed in place): appends fwd_buf =
//                       the data spectrum, FFT_LEN complex floats                             (rows G4-G6)
//   D n              -> DecimateBy2float over the next n complex floats of in.bin: appends n / 2 complex floats (row G5)
//   B n              -> DecimateBy2binary over the next 2 n bytes (the bits[][2] array): appends n / 2 complex floats (row G6)
//   C sat            -> Correlate(sat, fwd_buf, &dop, &idx): appends snr, dop, idx            (rows G7, G9)
#include REF_SEARCH_CPP                 // -DREF_SEARCH_CPP='"<reference>/gps/search.cpp"' (oracle/build_ref.sh)
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

// ---- the server runtime's entry points this file calls: no arithmetic in any of them
gps_t gps;
static spi_shmem_t the_spi_shmem;
spi_shmem_t *spi_shmem_p = &the_spi_shmem;
static FILE *g_in;                                         // Sample()'s packets come from here
void GPSstat_init() {}
void alt_printf(const char *, ...) {}
extern "C" {                                               // (C_LINKAGE in support/coroutines.h)
int _CreateTask(funcP_t, const char *, void *, int, u4_t, int) { return 0; }
void *_TaskSleep(const char *, u64_t, u4_t *) { return NULL; }
void _NextTask(const char *, u4_t, u_int64_t) {}
}
void _spi_set(SPI_CMD, uint16_t, uint32_t) {}
void _spi_get(SPI_CMD cmd, SPI_MISO *rx, int bytes, uint16_t, uint32_t)
{
    if (cmd != CmdGetGPSSamples || fread(rx->byte, 1, bytes, g_in) != (size_t) bytes) { fprintf(stderr, "search_ref: short sample packet\n"); exit(4); }
}

int main(int argc, char **argv)
{
    if (argc == 2 && argv[1][0] == 'i') {
        SearchInit();
        fprintf(stderr, "SearchInit ok: %d Navstar %d QZSS %d E1B\n", gps.n_Navstar, gps.n_QZSS, gps.n_E1B);
        return 0;
    }
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin | init\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *outf = fopen(argv[3], "wb");
    g_in = fopen(argv[2], "rb");
    if (!sf || !g_in || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    SearchInit();
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'T') {
            int sat;
            if (fscanf(sf, "%d", &sat) != 1) return 3;
            fwrite(code[sat], sizeof(fftwf_complex), FFT_LEN, outf);
        } else if (op == 'S') {
            Sample();
            fwrite(fwd_buf, sizeof(fftwf_complex), FFT_LEN, outf);
        } else if (op == 'D' || op == 'B') {
            int n;
            if (fscanf(sf, "%d", &n) != 1 || n < 2 || n > NSAMPLES) return 3;
            static fftwf_complex buf[NSAMPLES + 2 * NTAPS];
            static char bb[NSAMPLES][2];
            int got;
            if (op == 'D') {
                if (fread(buf, sizeof(fftwf_complex), n, g_in) != (size_t) n) return 4;
                got = DecimateBy2float(n, buf, buf, false);
            } else {
                if (fread(bb, 2, n, g_in) != (size_t) n) return 4;
                got = DecimateBy2binary(n, bb, buf, false);
            }
            fwrite(buf, sizeof(fftwf_complex), got, outf);
        } else if (op == 'C') {
            int sat, dop = 0, idx = 0;
            if (fscanf(sf, "%d", &sat) != 1) return 3;
            const float snr = Correlate(sat, fwd_buf, &dop, &idx);
            const float r[3] = {snr, (float) dop, (float) idx};
            fwrite(r, sizeof(float), 3, outf);
        } else return 3;
    }
    fclose(outf);
    return 0;
}
