// Driver for the reference's own CFastFIR, built IN PLACE from /root/reference/rx/CuteSDR/fastfir.cpp + support/simd.cpp
// (oracle/build_ref.sh) against the FFTW3 API the image ships: hipFFTW (/opt/rocm/lib/libhipfftw.so, header
// /opt/rocm/include/hipfft/hipfftw.h -- AMD's implementation of the fftw3.h interface over hipFFT; the reference's Makefile
// takes FFTW3 from the distribution the same way, Makefile:365-366).  hipFFTW executes its transforms on the GPU: the binary is
// built here and RUN ON THE GPU BOX (tools/make_ref_fft_golden.py), which needs nothing of /root/reference at run time.
// Test infrastructure only.
//
// What the driver supplies besides main(): the three globals of the server that fastfir.cpp reads -- they are this test's
// configuration, not arithmetic: snd_rate (SND_RATE_4CH: the 12 kHz mode FlyDog builds), ext_users[] (zero: no extension has
// registered an FFT hook, until an H line registers this driver's), snd_inst[] (zero: no audio-spectrum hook).  The H hook is an
// EXTENSION, i.e. a caller of the reference's interface (extensions/ext.h:33, :62-63): it records the 1024-bin buffer ProcessData
// hands it (PRE_FILTERED: the forward spectrum times the CIC compensation; POST_FILTERED: the filtered spectrum) and, when asked to
// edit, zeroes bins 256 .. 767 of the PRE buffer and answers true -- which makes ProcessData filter the edited buffer
// (fastfir.cpp:286-290), the `buf_modified` path.
//
//   fastfir_ref script.txt in.bin out.bin
// script lines:
//   W window_func                  -> SetupWindowFunction(window_func)   (only -1 is used: >= 0 prints through the server's printf)
//   C do_cic_comp                  -> SetupCICFilter(do_cic_comp)
//   P inst lo hi offset fs         -> SetupParameters(inst, lo, hi, offset, fs)
//   H flags edit                   -> ext_users[0].receive_FFT = the hook, FFT_flags = flags (1 PRE_FILTERED, 2 POST_FILTERED, 0: unregister)
//   D n                            -> ProcessData(0, n, <n complex floats of in.bin>, out): appends the returned count, FirPos(),
//                                     the number of hook calls, per call its flag and the 1024 complex bins it was handed (PRE:
//                                     before the edit), and the count complex outputs
#include "cuteSDR.h"         // fastfir.cpp:49-54, in its own order
#include "fastfir.h"
#include "ext_int.h"
#include "rx_sound.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int snd_rate = SND_RATE_4CH;
ext_users_t ext_users[MAX_RX_CHANS];
snd_t snd_inst[MAX_RX_CHANS];

struct tap_rec { float flag; TYPECPX bins[CONV_FFT_SIZE]; };
static std::vector<tap_rec> g_taps;
static int g_edit;
static bool tap_hook(int, int, int flags, int, int ns_out, TYPECPX *samps)
{
    tap_rec r;
    r.flag = (float) flags;
    for (int k = 0; k < CONV_FFT_SIZE; k++) r.bins[k] = k < ns_out ? samps[k] : TYPECPX();
    g_taps.push_back(r);
    if (flags == PRE_FILTERED && g_edit) {
        for (int k = 256; k < 768; k++) samps[k].re = samps[k].im = 0;
        return true;
    }
    return false;
}

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *inf = fopen(argv[2], "rb"), *outf = fopen(argv[3], "wb");
    if (!sf || !inf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    static CFastFIR fir;
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'W') {
            int w;
            if (fscanf(sf, "%d", &w) != 1) return 3;
            fir.SetupWindowFunction(w);
        } else if (op == 'C') {
            int c;
            if (fscanf(sf, "%d", &c) != 1) return 3;
            fir.SetupCICFilter(c != 0);
        } else if (op == 'P') {
            int inst; float lo, hi, off, fs;
            if (fscanf(sf, "%d %f %f %f %f", &inst, &lo, &hi, &off, &fs) != 5) return 3;
            fir.SetupParameters(inst, lo, hi, off, fs);
        } else if (op == 'H') {
            int flags;
            if (fscanf(sf, "%d %d", &flags, &g_edit) != 2) return 3;
            ext_users[0].receive_FFT = flags ? tap_hook : NULL;
            ext_users[0].FFT_flags = flags;
        } else if (op == 'D') {
            int n;
            if (fscanf(sf, "%d", &n) != 1) return 3;
            std::vector<TYPECPX> in(n), out(n + 1024);
            if (n && fread(in.data(), sizeof(TYPECPX), n, inf) != (size_t) n) return 4;
            g_taps.clear();
            const int got = fir.ProcessData(0, n, in.data(), out.data());
            const float hdr[2] = {(float) got, (float) fir.FirPos()};
            fwrite(hdr, sizeof(float), 2, outf);
            if (ext_users[0].receive_FFT) {
                const float nt = (float) g_taps.size();
                fwrite(&nt, sizeof nt, 1, outf);
                for (size_t t = 0; t < g_taps.size(); t++) { fwrite(&g_taps[t].flag, sizeof(float), 1, outf); fwrite(g_taps[t].bins, sizeof(TYPECPX), CONV_FFT_SIZE, outf); }
            }
            fwrite(out.data(), sizeof(TYPECPX), got, outf);
        } else return 3;
    }
    fclose(outf);
    return 0;
}
