// search_dropin.cpp -- a plain C++11 host caller of libkiwigpu through include/kiwigpu.h only (no HIP,
// no torch): the sequence INTEGRATION.md section 1 patches into the reference's gps/search.cpp.
//
//   SearchInit()   code tables for the SVs searched              (gps/search.cpp:183-350)
//   SearchTask()   per SV: Sample(); snr = Correlate(sat, ..., &lo_shift, &ca_shift);
//                  ca_shift *= DECIM; if (snr >= min_sig) ChanStart(...)        (:548-601)
//
// The search loop asks for ONE SV per Correlate() call, like the reference, and between the enqueue
// and the fetch it polls kg_ctx_poll() -- the place where the reference's coroutine yields
// (NextTaskP, :479).  `--batch` searches all SVs in one launch instead (the throughput mode).
//
//   search_dropin <packed_bits_file> [--batch] [--sats a,b,c] [--repeat n]
// input: NSAMPLES/8 = 8192 bytes of 1-bit IF, LSB first (what CmdGetGPSSamples returns, :398-406).
// output, one line per SV: "sat <i> prn <p> snr <f> lo_shift <d> ca_shift <d> lo_rate 0x<x> ca_rate 0x<x> ca_pause <d>"
// and a last line "per-SV latency us: median <f> min <f>".
#include "kiwigpu.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define DECIM    KG_ACQ_DECIM         // gps/gps.h:62
#define MIN_SIG  16                   // gps/gps.h:60

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ < 0) { fprintf(stderr, "%s -> %s\n", #call, kg_last_error()); return 1; } \
    } while (0)

// gps/cacode.h:23-53 restated: G1 = x^10+x^3+1, G2 = x^10+x^9+x^8+x^6+x^3+x^2+1, taps (t0, t1)
static void cacode(int t0, int t1, uint8_t *chips)
{
    int g1[11], g2[11];
    for (int i = 1; i <= 10; i++) g1[i] = g2[i] = 1;
    for (int n = 0; n < 1023; n++) {
        chips[n] = (uint8_t) (g1[10] ^ g2[t0] ^ g2[t1]);
        g1[0] = g1[3] ^ g1[10];
        g2[0] = g2[2] ^ g2[3] ^ g2[6] ^ g2[8] ^ g2[9] ^ g2[10];
        for (int i = 10; i >= 1; i--) { g1[i] = g1[i - 1]; g2[i] = g2[i - 1]; }
    }
}

// Navstar PRN 1..32 G2 tap pairs (IS-GPS-200; the first 32 rows of gps/sats.cpp's Sats[])
static const int TAPS[32][2] = {
    {2, 6}, {3, 7}, {4, 8}, {5, 9}, {1, 9}, {2, 10}, {1, 8}, {2, 9}, {3, 10}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {7, 8},
    {8, 9}, {9, 10}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {6, 9}, {1, 3}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {8, 10},
    {1, 6}, {2, 7}, {3, 8}, {4, 9},
};

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s packed_bits_file [--batch] [--sats a,b,c] [--repeat n]\n", argv[0]); return 2; }
    bool batch = false;
    int repeat = 1;
    std::vector<int> sats;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--batch")) batch = true;
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--sats") && i + 1 < argc) {
            for (char *t = strtok(argv[++i], ","); t; t = strtok(NULL, ",")) sats.push_back(atoi(t));
        }
    }
    if (sats.empty()) sats.push_back(0);
    std::vector<uint8_t> packed(KG_ACQ_NSAMPLES / 8);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(packed.data(), 1, packed.size(), f) != packed.size()) { fprintf(stderr, "cannot read %zu bytes from %s\n", packed.size(), argv[1]); return 2; }
    fclose(f);

    // ---- SearchInit()
    kg_ctx *kg = NULL;
    kg_acq *kacq = NULL;
    CHECK(kg_ctx_create(0, NULL, &kg));
    CHECK(kg_acq_create(kg, 64 /* MAX_SATS */, -20, 20 /* +-5000 Hz / BIN_SIZE, :465 */, 1, &kacq));
    for (size_t k = 0; k < sats.size(); k++) {
        if (sats[k] < 0 || sats[k] >= 32) { fprintf(stderr, "this example knows the 32 Navstar rows only\n"); return 2; }
        uint8_t chips[1023];
        cacode(TAPS[sats[k]][0], TAPS[sats[k]][1], chips);
        CHECK(kg_acq_set_code(kacq, sats[k], chips, 1023, 0, KG_ACQ_L1_LIMIT));
    }

    // ---- SearchTask() loop body
    std::vector<double> lat;
    std::vector<kg_acq_result> res(sats.size());
    int lo_shift = 0, ca_shift = 0;                           // :513: carried over when nothing is found
    for (int rep = 0; rep < repeat; rep++) {
        const double t_sample = now_us();
        CHECK(kg_acq_sample_bits(kacq, 0, packed.data()));    // Sample(): enqueue only
        if (batch) {
            CHECK(kg_acq_correlate_async(kacq, 1, sats.data(), (int) sats.size()));
            while (kg_ctx_poll(kg) == 0) { /* NextTaskP("corr GPU", NT_LONG_RUN) */ }
            CHECK(kg_acq_fetch(kacq, res.data(), NULL));
            lat.push_back((now_us() - t_sample) / sats.size());
        } else {
            for (size_t k = 0; k < sats.size(); k++) {
                const double t0 = now_us();
                CHECK(kg_acq_correlate_async(kacq, 1, &sats[k], 1));          // Correlate(sat, ...)
                while (kg_ctx_poll(kg) == 0) { /* the coroutine would yield here */ }
                CHECK(kg_acq_fetch(kacq, &res[k], NULL));
                lat.push_back(now_us() - t0);
            }
        }
        if (rep + 1 < repeat) continue;
        for (size_t k = 0; k < sats.size(); k++) {
            const kg_acq_result &r = res[k];
            if (r.valid) { lo_shift = r.dop; ca_shift = r.idx * DECIM; }      // :495, :575
            kg_chan_start cs;
            memset(&cs, 0, sizeof cs);
            if (r.snr >= MIN_SIG)                                             // ChanStart(), :601
                kg_acq_chan_start(0, lo_shift, ca_shift, (now_us() - t_sample) / 1e6, &cs);
            printf("sat %d prn %d snr %.4f lo_shift %d ca_shift %d lo_rate 0x%08x ca_rate 0x%08x ca_pause %u\n",
                   sats[k], sats[k] + 1, r.snr, lo_shift, ca_shift, cs.lo_rate, cs.ca_rate, cs.ca_pause);
        }
    }
    std::sort(lat.begin(), lat.end());
    printf("per-SV latency us: median %.1f min %.1f (%s, %zu calls)\n", lat[lat.size() / 2], lat[0],
           batch ? "one launch for all SVs" : "one Correlate() call per SV", lat.size());
    kg_acq_destroy(kacq);                                     // SearchFree()
    kg_ctx_destroy(kg);
    return 0;
}
