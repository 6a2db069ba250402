// Driver for the reference's own Galileo E1-B memory codes, compiled IN PLACE from
// /root/reference/gps/e1bcode.h (with the reference's types.h / gps.h and the kiwi.gen.h its
// own e_cpu assembler generated; see oracle/Makefile).  Test infrastructure only; the binary
// goes to oracle/_ref/ (git-ignored).
//
//   e1b_ref   -> NUM_E1B_SATS lines: "<prn> <4092 chips as '0'/'1'>"
#include "e1bcode.h"      // -I$(REFERENCE)/gps
#undef printf             // the reference's printf.h redirects printf into its logger
#include <stdio.h>

int main()
{
    for (int prn = 1; prn <= NUM_E1B_SATS; prn++) {
        E1BCODE e(prn);
        char line[E1B_CODELEN + 1];
        for (int i = 0; i < E1B_CODELEN; i++) { line[i] = (char) ('0' + e.Chip()); e.Clock(); }
        line[E1B_CODELEN] = 0;
        fprintf(stdout, "%d %s\n", prn, line);
    }
    return 0;
}
