// waterfall_dropin.cpp -- a plain C++11 host caller of libkiwigpu through include/kiwigpu.h only (no HIP, no torch): the
// sequence INTEGRATION.md section 3 patches into the reference's rx/rx_waterfall.cpp, for all waterfall channels at once.
//
//   c2s_waterfall_init()   window functions + CIC compensation table -> kg_wf_set_tables        (:122-203)
//   c2s_waterfall()        CmdSetWFFreq / CmdSetWFDecim -> kg_ddc_set_wf;  new map / new scale -> kg_wf_set_channel
//                                                                                               (:466,:507,:775-928)
//   sample_wf()            non-overlapped: CmdWFReset + the one-shot sampler's 8192 outputs -> kg_ddc_wf_capture_dev
//                                                                                               (:1005-1067)
//   compute_frame()        window .. FFT .. power .. pixels .. dB .. u8 row -> kg_wf_frames_dev  (:1275-1575)
//                          "W/F " header + ADPCM                            -> kg_wf_packets_dev (:1602-1639)
// with kg_ctx_poll() where the reference's coroutine sleeps (WFSleepReasonUsec, :1024-1031).
//
// The tables are the CALLER's, exactly as in the reference: this program reads them from a file the way the reference
// would hand over WF_SHMEM->window_function, WF_SHMEM->CIC_comp and each wf_inst_t's arrays -- it computes none of them.
//
//   waterfall_dropin <tables.bin> <adc.bin> <out.bin> [steps]
// tables.bin: int32 nchan; float windows[4][8192]; float cic_comp[8192]; then per channel
//             { kg_wf_chan_cfg cfg; uint64 phase_inc; int32 decim; uint32 x_bin; uint16 fft2wf_map[4096]; uint16 drop_sample[1024];
//               float fft_scale[1024]; float fft_scale_div2[1024] }
// adc.bin:    int16 ADC samples: `steps` (default 1) blocks back to back, block length = (file length / 2) / steps
// out.bin:    per step and channel: uint8 row[1024]; int32 pkt_bytes; uint8 pkt[KG_WF_PKT_MAX]
#include "kiwigpu.h"

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

struct chan_tables {
    kg_wf_chan_cfg cfg;
    uint64_t phase_inc;
    int32_t decim;
    uint32_t x_bin;
    const uint16_t *map, *drop;
    const float *scale, *scale2;
};

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: waterfall_dropin <tables.bin> <adc.bin> <out.bin> [steps]\n"); return 2; }
    const int steps = argc > 4 ? atoi(argv[4]) : 1;
    std::vector<unsigned char> tb, adcb;
    if (!read_all(argv[1], tb) || !read_all(argv[2], adcb) || steps < 1) { fprintf(stderr, "cannot read the inputs\n"); return 2; }
    const unsigned char *p = tb.data();
    int32_t nchan;
    memcpy(&nchan, p, 4); p += 4;
    const float *windows = (const float *) p; p += sizeof(float) * 4 * 8192;
    const float *cic = (const float *) p; p += sizeof(float) * 8192;
    std::vector<chan_tables> ch((size_t) nchan);
    for (int c = 0; c < nchan; c++) {
        memcpy(&ch[c].cfg, p, sizeof(kg_wf_chan_cfg)); p += sizeof(kg_wf_chan_cfg);
        memcpy(&ch[c].phase_inc, p, 8); p += 8;
        memcpy(&ch[c].decim, p, 4); p += 4;
        memcpy(&ch[c].x_bin, p, 4); p += 4;
        ch[c].map = (const uint16_t *) p; p += 2 * 4096;
        ch[c].drop = (const uint16_t *) p; p += 2 * 1024;
        ch[c].scale = (const float *) p; p += 4 * 1024;
        ch[c].scale2 = (const float *) p; p += 4 * 1024;
    }
    if ((size_t) (p - tb.data()) != tb.size()) { fprintf(stderr, "tables.bin: %zu bytes, expected %zu\n", tb.size(), (size_t) (p - tb.data())); return 2; }
    const size_t n = adcb.size() / 2 / (size_t) steps;          // samples per block

    kg_ctx *kg = nullptr;
    kg_wf *kwf = nullptr;
    kg_ddc *kddc = nullptr;
    if (kg_ctx_create(0, nullptr, &kg) < 0) { fprintf(stderr, "kg_ctx_create: %s\n", kg_last_error()); return 3; }
    // c2s_waterfall_init()
    CHECK(kg_wf_create(kg, nchan, &kwf));
    CHECK(kg_wf_set_tables(kwf, windows, cic));
    CHECK(kg_ddc_create(kg, nchan, n, &kddc));
    // c2s_waterfall(): per channel the two SPI commands and the arrays "new_map" / "new_scale_mask" rebuilt
    std::vector<int32_t> chans((size_t) nchan);
    std::vector<kg_wf_pkt_info> info((size_t) nchan);
    for (int c = 0; c < nchan; c++) {
        chans[c] = c;
        CHECK(kg_ddc_set_wf(kddc, c, ch[c].phase_inc, ch[c].decim));
        CHECK(kg_wf_set_channel(kwf, c, &ch[c].cfg, ch[c].map, ch[c].drop, ch[c].scale, ch[c].scale2));
        info[c].x_bin_server = ch[c].x_bin; info[c].zoom = (uint32_t) ch[c].cfg.zoom; info[c].use_compression = 1;
    }
    void *d_adc = nullptr, *d_iq = nullptr, *d_rows = nullptr, *d_pkts = nullptr;
    CHECK(kg_dev_alloc(kg, 2 * n, &d_adc));
    CHECK(kg_dev_alloc(kg, (size_t) nchan * 8192 * 4, &d_iq));
    CHECK(kg_dev_alloc(kg, (size_t) nchan * 1024, &d_rows));
    CHECK(kg_dev_alloc(kg, (size_t) nchan * KG_WF_PKT_MAX, &d_pkts));
    FILE *fo = fopen(argv[3], "wb");
    if (!fo) { fprintf(stderr, "cannot write %s\n", argv[3]); return 2; }
    std::vector<unsigned char> rows((size_t) nchan * 1024), pkts((size_t) nchan * KG_WF_PKT_MAX);
    std::vector<int32_t> pkt_bytes((size_t) nchan);
    std::vector<int64_t> nouts((size_t) nchan);
    long polls = 0;
    for (int s = 0; s < steps; s++) {
        CHECK(kg_dev_upload(kg, d_adc, adcb.data() + 2 * n * (size_t) s, 2 * n));
        for (int c = 0; c < nchan; c++) info[c].seq = (uint32_t) s;
        // sample_wf(), every channel: reset + one-shot sampler; compute_frame(); the packet -- three enqueues, no wait
        CHECK(kg_ddc_wf_capture_dev(kddc, d_adc, n, chans.data(), nchan, d_iq, 8192, 8192, nouts.data()));
        for (int c = 0; c < nchan; c++)
            if (nouts[c] != 8192) { fprintf(stderr, "channel %d: the block fills only %lld of the sampler's 8192\n", c, (long long) nouts[c]); return 4; }
        CHECK(kg_wf_frames_dev(kwf, nchan, chans.data(), d_iq, d_rows));
        CHECK(kg_wf_packets_dev(kg, d_rows, 1024, nchan, info.data(), d_pkts, KG_WF_PKT_MAX, pkt_bytes.data()));
        int idle;
        while ((idle = kg_ctx_poll(kg)) == 0) polls++;          // NextTask("wf GPU") in the reference's coroutine
        CHECK(idle);
        CHECK(kg_dev_download(kg, rows.data(), d_rows, rows.size()));
        CHECK(kg_dev_download(kg, pkts.data(), d_pkts, pkts.size()));
        for (int c = 0; c < nchan; c++) {
            fwrite(rows.data() + (size_t) c * 1024, 1, 1024, fo);
            fwrite(&pkt_bytes[c], 4, 1, fo);
            fwrite(pkts.data() + (size_t) c * KG_WF_PKT_MAX, 1, KG_WF_PKT_MAX, fo);
        }
    }
    fclose(fo);
    printf("%d channels x %d frames, %zu ADC samples per block, %ld polls\n", nchan, steps, n, polls);
    kg_dev_free(kg, d_adc); kg_dev_free(kg, d_iq); kg_dev_free(kg, d_rows); kg_dev_free(kg, d_pkts);
    kg_ddc_destroy(kddc);
    kg_wf_destroy(kwf);
    kg_ctx_destroy(kg);
    return 0;
}
