// Prints, as one JSON object, the constants of the reference's gps/gps.h, kiwi.h and of the
// kiwi.gen.h its own assembler generates from kiwi.config -- the numbers the oracle and the
// product hard-pin (SURVEY.md 8a "Constants").  Compiled IN PLACE against the reference's
// headers (oracle/Makefile).  Test infrastructure only.
#include "types.h"
#include "kiwi.h"
#include "gps.h"
#undef printf
#include <stdio.h>

#define D(name) fprintf(stdout, "  \"%s\": %.17g,\n", #name, (double) (name))

int main()
{
    fprintf(stdout, "{\n");
    // gps/gps.h
    D(FC); D(FS); D(FS_I); D(CPS); D(L1_f); D(BIN_SIZE); D(DECIM); D(FFT_LEN); D(NSAMPLES);
    D(SAMPLE_RATE); D(MIN_SIG); D(L1_CODELEN); D(E1B_CODELEN); D(L1_CODE_PERIOD); D(E1B_CODE_PERIOD);
    D(MAX_SATS); D(NUM_E1B_SATS);
    // kiwi.gen.h (generated from kiwi.config by e_cpu/asm)
    D(GPS_SAMPS); D(NWF_SAMPS); D(RX1_STD_DECIM); D(RX2_STD_DECIM); D(RX1_WIDE_DECIM); D(RX2_WIDE_DECIM);
    D(RX1_BITS); D(RX2_BITS); D(RXO_BITS); D(WF1_BITS); D(WFO_BITS); D(RX1_STAGES); D(RX2_STAGES);
    D(WF1_STAGES); D(VAL_CICF_DECIM_BY_2); D(MAX_ZOOM);
    // kiwi.h
    D(RXOUT_SCALE); D(CUTESDR_SCALE);
    fprintf(stdout, "  \"_\": 0\n}\n");
    return 0;
}
