/*
 * kiwi_oracle_wire.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 7: wire formats.
 *   IMA ADPCM coder        rx/csdr/ima_adpcm.cpp:89-214 (tables :89-108, decode step :110-135,
 *                          encode step :161-181, i16 -> e8 :185-197, u8 -> e8 :201-214)
 *   waterfall packet       rx/rx_waterfall.h:73-89 (wf_pkt_t), rx/rx_waterfall.cpp:284, 1602-1639
 *   sound packet header    rx/rx_sound.h:42-52, rx/rx_sound.cpp:252, 1219-1254
 * Integer arithmetic throughout: parity bar = bit-exact.
 * PINNING: ima_adpcm.cpp does not compile from its own files (printf.h -> conn.h -> kiwi.h ->
 * the generated kiwi.gen.h), so no reference object.  The coder is the published IMA/DVI ADPCM
 * of the CWI audio library (the file's own provenance, :60-84); the i16 path is pinned in
 * tests/test_host_cpu.py against CPython's audioop.lin2adpcm / adpcm2lin, an independent
 * implementation of that codec (nibble order swapped: the reference puts the first sample of a
 * pair in the LOW nibble, :192-193).  The u8 path differs only by its clamps (:205-206).
 */
#include "kiwi_oracle.h"

#include <math.h>

#include <string.h>

static const int index_adjust[16] = { -1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8 };   /* :89-94 */

/* The 89 step sizes of the IMA ADPCM specification (:96-108) */
static const int step_size[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97,
    107, 118, 130, 143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724,
    796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026,
    4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500,
    20350, 22385, 24623, 27086, 29794, 32767
};

const int *ko_adpcm_step_table(void) { return step_size; }

/* :110-135 */
static inline int adpcm_decode_step(unsigned code, ko_adpcm_state *s, int pos_clamp, int neg_clamp)
{
    const int step = step_size[s->index];
    int difference = step >> 3;
    if (code & 1) difference += step >> 2;
    if (code & 2) difference += step >> 1;
    if (code & 4) difference += step;
    if (code & 8) difference = -difference;
    s->previous += difference;
    if (s->previous > pos_clamp) s->previous = pos_clamp;
    else if (s->previous < neg_clamp) s->previous = neg_clamp;
    s->index += index_adjust[code];
    if (s->index < 0) s->index = 0;
    else if (s->index > 88) s->index = 88;
    return s->previous;
}

/* :161-181 */
static inline unsigned adpcm_encode_step(int sample, ko_adpcm_state *s, int pos_clamp, int neg_clamp)
{
    int diff = sample - s->previous;
    int step = step_size[s->index];
    unsigned code = 0;
    if (diff < 0) { code = 8; diff = -diff; }
    if (diff >= step) { code |= 4; diff -= step; }
    step >>= 1;
    if (diff >= step) { code |= 2; diff -= step; }
    step >>= 1;
    if (diff >= step) code |= 1;
    adpcm_decode_step(code, s, pos_clamp, neg_clamp);
    return code;
}

/* :185-197.  n samples -> n/2 bytes */
void ko_adpcm_encode_i16(const int16_t *in, uint8_t *out, int n, ko_adpcm_state *s)
{
    for (int i = 0; i < n / 2; i++) {
        unsigned b = adpcm_encode_step(in[2 * i], s, 32767, -32768);
        b |= adpcm_encode_step(in[2 * i + 1], s, 32767, -32768) << 4;
        out[i] = (uint8_t) b;
    }
}

/* :201-214; in and out may be the same buffer */
void ko_adpcm_encode_u8(const uint8_t *in, uint8_t *out, int n, ko_adpcm_state *s)
{
    for (int i = 0; i < n / 2; i++) {
        const uint8_t i0 = in[2 * i], i1 = in[2 * i + 1];
        unsigned b = adpcm_encode_step(i0, s, 255, 0);
        b |= adpcm_encode_step(i1, s, 255, 0) << 4;
        out[i] = (uint8_t) b;
    }
}

/* :137-147 */
void ko_adpcm_decode_i16(const uint8_t *in, int16_t *out, int nbytes, ko_adpcm_state *s)
{
    for (int i = 0, k = 0; i < nbytes; i++) {
        out[k++] = (int16_t) adpcm_decode_step(in[i] & 0xf, s, 32767, -32768);
        out[k++] = (int16_t) adpcm_decode_step((in[i] >> 4) & 0xf, s, 32767, -32768);
    }
}

/* :149-159 */
void ko_adpcm_decode_u8(const uint8_t *in, uint8_t *out, int nbytes, ko_adpcm_state *s)
{
    for (int i = 0, k = 0; i < nbytes; i++) {
        out[k++] = (uint8_t) adpcm_decode_step(in[i] & 0xf, s, 255, 0);
        out[k++] = (uint8_t) adpcm_decode_step((in[i] >> 4) & 0xf, s, 255, 0);
    }
}

static void put_le32(uint8_t *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }

/* wf_pkt_t (rx_waterfall.h:73-89, packed, little-endian host) filled as compute_frame() does
 * (rx_waterfall.cpp:284, 1602-1639): returns the byte count that goes on the wire
 * (16-byte header + wf->out_bytes). */
int ko_wf_packet(const uint8_t *row, uint32_t x_bin_server, uint32_t zoom, uint32_t seq, int use_compression,
                 uint8_t *pkt)
{
    use_compression = use_compression && zoom != 0;           /* :1283-1285: wf->compression && wf->zoom != 0 */
    memcpy(pkt, "W/F ", 4);                                   /* :284 */
    put_le32(pkt + 4, x_bin_server);                          /* :1615 */
    uint32_t fz = zoom;                                       /* :1616 */
    uint8_t *un = pkt + 16;
    int out_bytes;
    if (use_compression) {
        uint8_t tmp[KO_WF_ADPCM_PAD + KO_WF_WIDTH];
        memset(tmp, row[0], KO_WF_ADPCM_PAD);                 /* :1625 adpcm_pad <- buf2[0] */
        memcpy(tmp + KO_WF_ADPCM_PAD, row, KO_WF_WIDTH);      /* the pixels were written to buf2 */
        ko_adpcm_state st = {0, 0};                           /* :1626 */
        ko_adpcm_encode_u8(tmp, un, KO_WF_ADPCM_PAD + KO_WF_WIDTH, &st);   /* :1627 */
        out_bytes = (KO_WF_ADPCM_PAD + KO_WF_WIDTH) / 2;      /* :1628 */
        fz |= 0x00010000u;                                    /* :1629 WF_FLAGS_COMPRESSION */
    } else {
        memcpy(un, row, KO_WF_WIDTH);
        out_bytes = KO_WF_WIDTH;                              /* :1631 */
    }
    put_le32(pkt + 8, fz);
    put_le32(pkt + 12, seq);                                  /* :1635 */
    return 16 + out_bytes;
}

/* The 10 header bytes of snd_pkt_real_t (rx_sound.h:42-48) as rx_sound.cpp:252, 1219-1254
 * fills them: id "SND", flags, seq little-endian, S-meter big-endian in 0.1 dB above -127 dBm. */
/* rx/rx_sound.cpp:1076-1096: the IQ modes' payload.  (s2_t) of a TYPEREAL: truncation; out of range as x86 converts (low 16 bits
 * of the int32 conversion), the same definition as kiwi_oracle_post.c's to_mono16. */
void ko_snd_iq_payload(const ko_cpx *in, int n, int little_endian, uint8_t *out)
{
    for (int j = 0; j < n; j++) {
        const float f[2] = {in[j].re, in[j].im};
        for (int c = 0; c < 2; c++) {
            int32_t w;
            if (!(f[c] > -2147483648.0f && f[c] < 2147483648.0f)) w = (int32_t) 0x80000000u;
            else w = (int32_t) f[c];
            const uint16_t v = (uint16_t) (uint32_t) w;
            if (little_endian) { *out++ = (uint8_t) v; *out++ = (uint8_t) (v >> 8); }      /* :1079-1081 */
            else { *out++ = (uint8_t) (v >> 8); *out++ = (uint8_t) v; }                     /* :1088-1091 */
        }
    }
}

void ko_snd_header(uint8_t flags, uint32_t seq, float smeter_dBm, uint8_t *h)
{
    if (smeter_dBm < -127.0) smeter_dBm = -127.0; else         /* :1223-1224 */
    if (smeter_dBm > 3.4) smeter_dBm = 3.4;
    const uint16_t sm = (uint16_t) ((smeter_dBm + 127.0) * 10);   /* :1225 */
    memcpy(h, "SND", 3);
    h[3] = flags;
    put_le32(h + 4, seq);                                      /* :1252 SET_LE_U32 */
    h[8] = sm >> 8; h[9] = sm & 0xff;                          /* :1226 SET_BE_U16 */
}

/* The GPS time stamp of snd_pkt_iq_t: rx/rx_sound.cpp:557 (per data-pump buffer) and :636-661 (per FIR
 * output block).  Pure double arithmetic on host values. */
static const double KO_GPS_WEEK_SEC = 7 * 24 * 3600.0;         /* :93 */
void ko_snd_gps_begin(ko_gps_state *s, double clk_gps_secs, double dticks, double adc_clock_base,
                      double gps_delay, double gps_delay2)
{
    s->gpssec = fmod(KO_GPS_WEEK_SEC + clk_gps_secs + (dticks / adc_clock_base) - gps_delay + gps_delay2, KO_GPS_WEEK_SEC);
}
void ko_snd_gps_stamp(ko_gps_state *s, int norm_nrx_samps, int fir_pos, int agc_on, int agc_delay, int rx_decim,
                      double adc_clock_base, double clk_gps_secs, uint64_t clk_ticks, uint32_t *gpssec,
                      uint32_t *gpsnsec, uint8_t *last_gps_solution)
{
    int sample_filter_delays = norm_nrx_samps - fir_pos;       /* :638 */
    if (agc_on) sample_filter_delays -= agc_delay;             /* :640-641 */
    s->gpssec = fmod(KO_GPS_WEEK_SEC + s->gpssec + (rx_decim * sample_filter_delays / adc_clock_base), KO_GPS_WEEK_SEC);
    *gpssec = (uint32_t) s->last_gpssec;                       /* :654 */
    *gpsnsec = s->gps_init ? (uint32_t) (1e9 * (s->last_gpssec - *gpssec)) : 0;
    const double dt_to_pos_sol = s->last_gpssec - clk_gps_secs;
    *last_gps_solution = s->gps_init ? ((clk_ticks == 0) ? 255 : (uint8_t) (dt_to_pos_sol < 252.0 ? dt_to_pos_sol : 252.0)) : 0;
    if (!s->gps_init) s->gps_init = 1;
    s->last_gpssec = s->gpssec;
}
