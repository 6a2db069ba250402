// Driver for the reference's own IMA ADPCM coder, built IN PLACE from
// /root/reference/rx/csdr/ima_adpcm.cpp (oracle/Makefile).  Test infrastructure only.
//
//   adpcm_ref enc_i16|enc_u8|dec_i16|dec_u8 index previous blocklen < in > out
// The input is processed in blocks of `blocklen` samples (enc) / bytes (dec) with the state
// carried across blocks, as c2s_sound() carries adpcm_snd; the final "index previous" goes to
// stderr.
#include "types.h"
#include "ima_adpcm.h"
#undef printf
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: %s mode index previous blocklen\n", argv[0]); return 2; }
    ima_adpcm_state_t st;
    memset(&st, 0, sizeof st);
    st.index = atoi(argv[2]); st.previousValue = atoi(argv[3]);
    const int bl = atoi(argv[4]);
    const bool enc = !strncmp(argv[1], "enc", 3), wide = strstr(argv[1], "i16") != NULL;
    const size_t in_unit = enc ? (wide ? 2 : 1) : 1;
    std::vector<unsigned char> in(bl * in_unit), out(bl * 4);
    for (;;) {
        const size_t got = fread(in.data(), in_unit, bl, stdin);
        if (got == 0) break;
        const int n = (int) got;
        if (enc && wide) { encode_ima_adpcm_i16_e8((short *) in.data(), out.data(), n, &st); fwrite(out.data(), 1, n / 2, stdout); }
        else if (enc) { encode_ima_adpcm_u8_e8(in.data(), out.data(), n, &st); fwrite(out.data(), 1, n / 2, stdout); }
        else if (wide) { decode_ima_adpcm_e8_i16(in.data(), (short *) out.data(), n, &st); fwrite(out.data(), 2, 2 * n, stdout); }
        else { decode_ima_adpcm_e8_u8(in.data(), out.data(), n, &st); fwrite(out.data(), 1, 2 * n, stdout); }
    }
    fprintf(stderr, "%d %d\n", st.index, st.previousValue);
    return 0;
}
