// rxbank_dropin.cpp -- a plain C++11 host of a BANK of receivers through include/kiwigpu.h only (no HIP, no torch): what
// INTEGRATION.md section 6 puts in the place of the reference's per-connection loops
//
//   data_pump()       rx/data_pump.cpp:292-341      one ADC block per step instead of one SPI transaction per 8 ms
//   c2s_sound()       rx/rx_sound.cpp:333-601       CFastFIR -> S-meter / CAgc / detector -> ADPCM, every receiver
//   c2s_waterfall()   rx/rx_waterfall.cpp:930-1170  sample_wf() (one-shot or overlapped sampler) -> compute_frame() -> wf_pkt_t
//
// with ONE library call per step (kg_rxbank_step) and kg_rxbank_poll() where the reference's coroutines yield.  The
// per-seam objects are configured once through the same entry points the single-seam examples use.
//
//   rxbank_dropin <tables.bin> <adc.bin> <out.bin> <steps>
// tables.bin: int32 nrx; int32 samples_per_step; float windows[4][8192]; float cic_comp[8192]; then per receiver
//             { kg_wf_chan_cfg cfg; uint64 wf_phase_inc; int32 decim; int32 overlapped; uint32 x_bin; uint16 fft2wf_map[4096];
//               uint16 drop_sample[1024]; float fft_scale[1024]; float fft_scale_div2[1024];
//               uint64 rx_phase_inc; float lo, hi, fs }
// adc.bin:    int16 ADC samples, `steps` blocks of samples_per_step back to back
// out.bin:    per step: kg_rxbank_step_info; per frame { int32 rx; uint8 row[1024]; int32 pkt_bytes; uint8 pkt[KG_WF_PKT_MAX] };
//             per receiver { int16 s16[nfir]; uint8 adpcm[nfir / 2] }
#include "kiwigpu.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ < 0) { fprintf(stderr, "%s -> %s\n", #call, kg_last_error()); return 1; } \
    } while (0)

static bool read_all(const char *path, std::vector<unsigned char> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t) n);
    const bool ok = fread(out.data(), 1, (size_t) n, f) == (size_t) n;
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: rxbank_dropin <tables.bin> <adc.bin> <out.bin> <steps>\n"); return 2; }
    const int steps = atoi(argv[4]);
    std::vector<unsigned char> tb, adcb;
    if (!read_all(argv[1], tb) || !read_all(argv[2], adcb) || steps < 1) { fprintf(stderr, "cannot read the inputs\n"); return 2; }
    const unsigned char *p = tb.data();
    int32_t nrx, n32;
    memcpy(&nrx, p, 4); p += 4;
    memcpy(&n32, p, 4); p += 4;
    const size_t n = (size_t) n32;
    if (adcb.size() < 2 * n * (size_t) steps) { fprintf(stderr, "adc.bin holds fewer than %d blocks of %zu samples\n", steps, n); return 2; }
    const float *windows = (const float *) p; p += sizeof(float) * 4 * 8192;
    const float *cic = (const float *) p; p += sizeof(float) * 8192;

    kg_rxbank *bank = nullptr;
    if (kg_rxbank_create(0, nrx, n, KG_RXDDC_STD, &bank) < 0) { fprintf(stderr, "kg_rxbank_create: %s\n", kg_last_error()); return 3; }
    // c2s_waterfall_init() / the connections' SET commands: every seam through its own entry points, on the bank's objects
    CHECK(kg_wf_set_tables(kg_rxbank_wf(bank), windows, cic));
    for (int k = 0; k < nrx; k++) {
        kg_wf_chan_cfg cfg;
        uint64_t wf_inc, rx_inc; int32_t decim, overlapped; uint32_t x_bin; float lo, hi, fs;
        memcpy(&cfg, p, sizeof cfg); p += sizeof cfg;
        memcpy(&wf_inc, p, 8); p += 8;
        memcpy(&decim, p, 4); p += 4;
        memcpy(&overlapped, p, 4); p += 4;
        memcpy(&x_bin, p, 4); p += 4;
        const uint16_t *map = (const uint16_t *) p; p += 2 * 4096;
        const uint16_t *drop = (const uint16_t *) p; p += 2 * 1024;
        const float *scale = (const float *) p; p += 4 * 1024;
        const float *scale2 = (const float *) p; p += 4 * 1024;
        memcpy(&rx_inc, p, 8); p += 8;
        memcpy(&lo, p, 4); p += 4; memcpy(&hi, p, 4); p += 4; memcpy(&fs, p, 4); p += 4;
        CHECK(kg_rxbank_set_wf(bank, k, wf_inc, decim, overlapped));                        // CmdSetWFFreq / CmdSetWFDecim / sampler mode
        CHECK(kg_wf_set_channel(kg_rxbank_wf(bank), k, &cfg, map, drop, scale, scale2));     // new map / new scale
        CHECK(kg_rxbank_set_wf_pkt(bank, k, x_bin, (uint32_t) cfg.zoom, 1));
        CHECK(kg_rxddc_set_freq(kg_rxbank_rxddc(bank), k, rx_inc));                          // CmdSetRXFreq
        CHECK(kg_fir_setup(kg_rxbank_fir(bank), k, lo, hi, 0.f, fs, -1, 0, 0));              // m_PassbandFIR[k].SetupParameters
        CHECK(kg_post_set_agc(kg_rxbank_post(bank), k, 1, 0, -100, 50, 6, 1000, fs));        // m_Agc[k].SetParameters
        CHECK(kg_post_set_smeter(kg_rxbank_post(bank), k, fs));
        CHECK(kg_post_set_mode(kg_rxbank_post(bank), k, KG_POST_SSB));
        CHECK(kg_post_reset(kg_rxbank_post(bank), k));
    }
    if ((size_t) (p - tb.data()) != tb.size()) { fprintf(stderr, "tables.bin: %zu bytes, expected %zu\n", tb.size(), (size_t) (p - tb.data())); return 2; }

    kg_ctx *kg = kg_rxbank_ctx(bank);
    kg_rxbank_bufs bufs;
    CHECK(kg_rxbank_buffers(bank, &bufs));
    void *d_adc = nullptr;
    CHECK(kg_dev_alloc(kg, 2 * n * (size_t) steps, &d_adc));
    CHECK(kg_dev_upload(kg, d_adc, adcb.data(), 2 * n * (size_t) steps));
    FILE *fo = fopen(argv[3], "wb");
    if (!fo) { fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
    std::vector<unsigned char> rows((size_t) nrx * 1024), pkts((size_t) nrx * bufs.wf_pkt_stride);
    std::vector<int16_t> s16(bufs.fir_stride);
    std::vector<unsigned char> pay(bufs.fir_stride / 2);
    std::vector<int32_t> rx_of((size_t) nrx), pkt_bytes((size_t) nrx);
    long polls = 0;
    double enq_us = 0.0;
    for (int s = 0; s < steps; s++) {
        kg_rxbank_step_info info;
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(kg_rxbank_step(bank, (const char *) d_adc + 2 * n * (size_t) s, nullptr, &info));       // the whole step: enqueue only
        enq_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        int idle;
        while ((idle = kg_rxbank_poll(bank)) == 0) polls++;            // NextTask() in the reference's coroutines
        CHECK(idle);
        const int nf = kg_rxbank_frame_map(bank, rx_of.data(), nullptr, pkt_bytes.data());
        fwrite(&info, sizeof info, 1, fo);
        if (nf > 0) {
            CHECK(kg_dev_download(kg, rows.data(), bufs.wf_rows, (size_t) nf * 1024));
            CHECK(kg_dev_download(kg, pkts.data(), bufs.wf_pkts, (size_t) nf * bufs.wf_pkt_stride));
        }
        for (int f = 0; f < nf; f++) {
            fwrite(&rx_of[f], 4, 1, fo);
            fwrite(rows.data() + (size_t) f * 1024, 1, 1024, fo);
            fwrite(&pkt_bytes[f], 4, 1, fo);
            fwrite(pkts.data() + (size_t) f * bufs.wf_pkt_stride, 1, KG_WF_PKT_MAX, fo);
        }
        for (int k = 0; k < nrx && info.nfir > 0; k++) {
            CHECK(kg_dev_download(kg, s16.data(), (const char *) bufs.s16 + 2 * bufs.fir_stride * (size_t) k, 2 * (size_t) info.nfir));
            CHECK(kg_dev_download(kg, pay.data(), (const char *) bufs.adpcm + bufs.fir_stride / 2 * (size_t) k, (size_t) info.nfir / 2));
            fwrite(s16.data(), 2, (size_t) info.nfir, fo);
            fwrite(pay.data(), 1, (size_t) info.nfir / 2, fo);
        }
    }
    fclose(fo);
    printf("%d receivers x %d steps of %zu ADC samples: kg_rxbank_step %.1f us of host time per step, %ld polls\n", nrx, steps, n,
           enq_us / steps, polls);
    kg_dev_free(kg, d_adc);
    kg_rxbank_destroy(bank);
    return 0;
}
