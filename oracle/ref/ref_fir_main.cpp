// Driver for the reference's own CFir, built IN PLACE from /root/reference/rx/CuteSDR/fir.cpp
// (+ fir.h, datatypes.h, kiwi.h and the generated kiwi.gen.h; oracle/build_ref.sh).  Test
// infrastructure only.
//
//   fir_ref script.txt in.bin out.bin
// script lines (floats of in.bin are consumed in order, floats are appended to out.bin):
//   L numtaps scale astop fpass fstop fs  -> InitLPFilter(...): appends the returned tap count
//   H numtaps scale astop fpass fstop fs  -> InitHPFilter(...): appends the returned tap count
//   K n fs   -> InitConstFir(n, <n floats of in.bin>, fs)
//   R n      -> ProcessFilter(n, TYPEREAL in, TYPEREAL out)
//   M n      -> ProcessFilter(n, TYPEREAL in, TYPEMONO16 out)     (mono16 written as float)
//   S n      -> ProcessFilter(n, TYPEMONO16 in, TYPEMONO16 out)   (the input floats hold int16 values)
// The designed taps are private members; an impulse through `R` reads them back exactly
// (1.0f * c plus zeros).
#include "fir.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *inf = fopen(argv[2], "rb"), *outf = fopen(argv[3], "wb");
    if (!sf || !inf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    static CFir fir;
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'L' || op == 'H') {
            int nt; float scale, astop, fpass, fstop, fs;
            if (fscanf(sf, "%d %f %f %f %f %f", &nt, &scale, &astop, &fpass, &fstop, &fs) != 6) return 3;
            float got = (float) (op == 'L' ? fir.InitLPFilter(nt, scale, astop, fpass, fstop, fs)
                                           : fir.InitHPFilter(nt, scale, astop, fpass, fstop, fs));
            fwrite(&got, sizeof got, 1, outf);
        } else if (op == 'K') {
            int n; float fs;
            if (fscanf(sf, "%d %f", &n, &fs) != 2) return 3;
            std::vector<TYPEREAL> c(n);
            if (fread(c.data(), sizeof(TYPEREAL), n, inf) != (size_t) n) return 4;
            fir.InitConstFir(n, c.data(), fs);
        } else if (op == 'R' || op == 'M' || op == 'S') {
            int n;
            if (fscanf(sf, "%d", &n) != 1) return 3;
            std::vector<TYPEREAL> in(n), out(n);
            if (fread(in.data(), sizeof(TYPEREAL), n, inf) != (size_t) n) return 4;
            if (op == 'R') {
                fir.ProcessFilter(n, in.data(), out.data());
            } else {
                std::vector<TYPEMONO16> m(n), mi(n);
                if (op == 'M') {
                    fir.ProcessFilter(n, in.data(), m.data());
                } else {
                    for (int i = 0; i < n; i++) mi[i] = (TYPEMONO16) in[i];
                    fir.ProcessFilter(n, mi.data(), m.data());
                }
                for (int i = 0; i < n; i++) out[i] = (float) m[i];
            }
            fwrite(out.data(), sizeof(float), n, outf);
        } else return 3;
    }
    fclose(outf);
    return 0;
}
