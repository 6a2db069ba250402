// Driver for the reference's own CSquelch, built IN PLACE from /root/reference/rx/CuteSDR/squelch.cpp
// and fir.cpp (its m_HpFir) (+ their headers and the generated kiwi.gen.h; oracle/build_ref.sh).
// Test infrastructure only.
//
//   squelch_ref script.txt in.bin out.bin
// script lines:
//   P rate        -> SetupParameters(0, rate)            (rx/rx_sound.cpp:261)
//   Q value max   -> SetSquelch(value, max)              (rx/rx_sound.cpp:262, rx/rx_sound_cmd.cpp:430)
//   Z             -> Reset()                             (rx/rx_sound_cmd.cpp:238)
//   F n           -> PerformFMSquelch(n, <n floats of in.bin>, mono16 out): appends the n outputs
//                    (as floats) and then the return value nsq_nc_sq (-1, 0, +1)
#include "squelch.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s script in.bin out.bin\n", argv[0]); return 2; }
    FILE *sf = fopen(argv[1], "r"), *inf = fopen(argv[2], "rb"), *outf = fopen(argv[3], "wb");
    if (!sf || !inf || !outf) { fprintf(stderr, "cannot open files\n"); return 2; }
    static CSquelch sq;
    char op;
    while (fscanf(sf, " %c", &op) == 1) {
        if (op == 'P') {
            float rate;
            if (fscanf(sf, "%f", &rate) != 1) return 3;
            sq.SetupParameters(0, rate);
        } else if (op == 'Q') {
            int v, mx;
            if (fscanf(sf, "%d %d", &v, &mx) != 2) return 3;
            sq.SetSquelch(v, mx);
        } else if (op == 'Z') {
            sq.Reset();
        } else if (op == 'F') {
            int n;
            if (fscanf(sf, "%d", &n) != 1) return 3;
            std::vector<TYPEREAL> in(n), out(n + 1);
            std::vector<TYPEMONO16> m(n);
            if (fread(in.data(), sizeof(TYPEREAL), n, inf) != (size_t) n) return 4;
            int rc = sq.PerformFMSquelch(n, in.data(), m.data());
            for (int i = 0; i < n; i++) out[i] = (float) m[i];
            out[n] = (float) rc;
            fwrite(out.data(), sizeof(float), n + 1, outf);
        } else return 3;
    }
    fclose(outf);
    return 0;
}
