// Driver for the reference's own CAgc, built IN PLACE from /root/reference/rx/CuteSDR/agc.cpp
// (+ agc.h, datatypes.h, kiwi.h and the generated kiwi.gen.h; oracle/Makefile).  Test
// infrastructure only.
//
//   agc_ref script.txt in.bin out.bin
// script lines:
//   P agc_on use_hang threshold manual_gain slope decay sample_rate   -> SetParameters(...)
//   C n     -> ProcessData(n, complex in, complex out): consumes n complex floats of in.bin,
//              appends n complex floats to out.bin
//   M n     -> ProcessData(n, complex in, mono16 out): appends n int16 (as float) to out.bin
//   D       -> appends GetDelaySamples() (one float) to out.bin
#include "agc.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *inf = fopen(argv[2], "rb"), *outf = fopen(argv[3], "wb");
    if (!sf || !inf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    static CAgc agc;
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'P') {
            int on, hang, thr, man, slope, decay; float sr;
            if (fscanf(sf, "%d %d %d %d %d %d %f", &on, &hang, &thr, &man, &slope, &decay, &sr) != 7) return 3;
            agc.SetParameters(on != 0, hang != 0, thr, man, slope, decay, sr);
        } else if (op == 'C' || op == 'M') {
            int n;
            if (fscanf(sf, "%d", &n) != 1) return 3;
            std::vector<TYPECPX> in(n), out(n);
            if (fread(in.data(), sizeof(TYPECPX), n, inf) != (size_t) n) return 4;
            if (op == 'C') {
                agc.ProcessData(n, in.data(), out.data());
                fwrite(out.data(), sizeof(TYPECPX), n, outf);
            } else {
                std::vector<TYPEMONO16> m(n);
                agc.ProcessData(n, in.data(), m.data());
                for (int i = 0; i < n; i++) { float f = (float) m[i]; fwrite(&f, sizeof f, 1, outf); }
            }
        } else if (op == 'D') {
            float f = (float) agc.GetDelaySamples();
            fwrite(&f, sizeof f, 1, outf);
        } else return 3;
    }
    fclose(outf);
    return 0;
}
