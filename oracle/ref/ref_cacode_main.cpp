// Driver for the reference's own C/A generator, compiled IN PLACE from
// /root/reference/gps/cacode.h (a standalone header: it needs only <memory.h>).
// Test infrastructure only; output goes to oracle/_ref/ (git-ignored).
//
//   cacode_ref T0 T1   -> prints the 1023 chips as a string of '0'/'1'
#include <stdio.h>
#include <stdlib.h>
#include "cacode.h"      // -I$(REFERENCE)/gps

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s t0 t1\n", argv[0]); return 2; }
    CACODE ca(atoi(argv[1]), atoi(argv[2]));
    for (int i = 0; i < 1023; i++) { putchar('0' + ca.Chip()); ca.Clock(); }
    putchar('\n');
    return 0;
}
