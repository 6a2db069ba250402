/*
 * kiwi_oracle_handoff.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Part 8: what turns results into the reference's own programming.
 *   CHANNEL::Start() NCO rates, code creep, ca_pause       gps/channel.cpp:267-311
 *   aperture_auto() waterfall autoscale                    rx/rx_waterfall.cpp:1173-1273,
 *                                                          dB_wire_to_dBm rx/rx_util.cpp:905-912
 * PINNED BY THE REFERENCE ITSELF (round 6): CHANNEL::Start by gps/channel.cpp built in place (tests/golden/chan_ref.npz, 205
 * calls: the SPI words equal); aperture_auto() by the reference's compute_frame() -> aperture_auto() run on the GPU box with
 * rx_util.cpp's dB_wire_to_dBm and misc.cpp's qsort_intcomp linked in place (tests/golden/aper_fftref.npz, 36 frames: avg_pwr[]
 * after every frame bit-exact -- the IIR's expf included, same libm -- signal / noise / counters equal).
 * tests/test_ref_pins_cpu.py holds both.
 */
#include "kiwi_oracle.h"

#include <math.h>
#include <stdlib.h>

#define FC 4.092e6            /* gps.h:42 */
#define FS 16.368e6           /* gps.h:43 */
#define FS_I 16368000         /* gps.h:44 */
#define CPS 1.023e6           /* gps.h:46 */
#define L1_f 1575.42e6        /* gps.h:49 */
static const float BIN_SIZE = 249.755859375;      /* gps.h:69 */

/* gps/channel.cpp:281-311.  secs = (timer_us() - t_sample) / 1e6 is the caller's. */
void ko_chan_start(int is_e1b, int lo_shift, int ca_shift, double secs, ko_chan_start_out *o)
{
    const double lo_dop = lo_shift * BIN_SIZE;                          /* :281 */
    const double ca_dop = (lo_dop / L1_f) * CPS;                        /* :282 */
    const uint32_t lo_rate = (FC + lo_dop) / FS * pow(2, 32);           /* :285 */
    const uint32_t ca_rate = (CPS + ca_dop) / FS * pow(2, 32);          /* :286 */
    const int code_creep = nearbyint((ca_dop * secs / CPS) * FS);       /* :296 */
    const int code_period_ms = is_e1b ? 4 : 1;                          /* :299, gps.h:50,54 */
    const int code_period_samples = FS_I / 1000 * code_period_ms;       /* :300 */
    const uint32_t ca_pause = code_period_samples - ((ca_shift + code_creep) % code_period_samples);   /* :302 */
    o->lo_dop = lo_dop; o->ca_dop = ca_dop;
    o->lo_rate = lo_rate; o->ca_rate = ca_rate;
    o->code_creep = code_creep; o->ca_pause = ca_pause;
}

static inline int wire_to_dBm(int db_value, int waterfall_cal)          /* rx_util.cpp:905-912 */
{
    if (db_value < 0) db_value = 0;
    if (db_value > 255) db_value = 255;
    return -(255 - db_value) + waterfall_cal;
}

/* rx_waterfall.cpp:1183-1222: the averaging step over pixels [start, stop) */
void ko_aper_update(float *avg_pwr, const uint8_t *bp, int algo, float param, int clear, int start, int stop,
                    int waterfall_cal)
{
    if (clear) {                                                        /* :1183-1185 */
        for (int i = start; i < stop; i++) avg_pwr[i] = wire_to_dBm(bp[i], waterfall_cal);
        return;
    }
    switch (algo) {
    case 0:                                                             /* IIR :1199-1206 */
        for (int i = start; i < stop; i++) {
            float pwr = wire_to_dBm(bp[i], waterfall_cal);
            float iir_gain = 1.0 - expf(-param * pwr / 255.0);
            if (iir_gain <= 0.01) iir_gain = 0.01;
            avg_pwr[i] += (pwr - avg_pwr[i]) * iir_gain;
        }
        break;
    case 1:                                                             /* MMA :1208-1213 */
        for (int i = start; i < stop; i++) {
            float pwr = wire_to_dBm(bp[i], waterfall_cal);
            avg_pwr[i] = ((avg_pwr[i] * (param - 1)) + pwr) / param;
        }
        break;
    case 2:                                                             /* EMA :1215-1220 */
        for (int i = start; i < stop; i++) {
            float pwr = wire_to_dBm(bp[i], waterfall_cal);
            avg_pwr[i] += (pwr - avg_pwr[i]) / param;
        }
        break;
    }
}

static int intcomp(const void *a, const void *b) { return *(const int *) a - *(const int *) b; }   /* support/misc.cpp:76-80 */

/* rx_waterfall.cpp:1233-1272 */
void ko_aper_report(const float *avg_pwr, int start, int stop, int *signal, int *noise)
{
    int band[1024], len = 0;
    for (int i = start; i < stop; i++) {
        const int b = ((int) floorf(avg_pwr[i] / 5)) * 5;               /* :1238, RESOLUTION_dB 5 */
        if (b <= -190) continue;                                        /* :1239 */
        band[len++] = b;
    }
    int max_count = 0, max_dBm = -999, min_dBm = 0;
    if (len) {
        qsort(band, len, sizeof(int), intcomp);
        int last = band[0], same = 0;
        for (int i = 0; i <= len; i++) {                                /* :1249-1262 */
            if (i == len || band[i] != last) {
                if (same > max_count) { max_count = same; min_dBm = last; }
                if (last > max_dBm) max_dBm = last;
                if (i == len) break;
                same = 1;
                last = band[i];
            } else {
                same++;
            }
        }
    } else {
        max_dBm = -110;                                                 /* :1264-1265 */
        min_dBm = -120;
    }
    if (max_dBm < -80) max_dBm = -80;                                   /* :1271 */
    *signal = max_dBm;
    *noise = min_dBm;
}
