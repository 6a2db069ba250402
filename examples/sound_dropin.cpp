// sound_dropin.cpp -- a plain C++11 host caller of libkiwigpu through include/kiwigpu.h only (no HIP, no torch): the sequence
// INTEGRATION.md section 4 patches into the reference's rx/data_pump.cpp and rx/rx_sound.cpp, for all audio channels at once.
//
//   snd_service()    one SPI buffer of rx_iq_t records -> kg_dpump_unpack_dev                              (data_pump.cpp:145-208)
//   c2s_sound()      m_PassbandFIR[ch].ProcessData      -> kg_fir_process_dev   (0 or 512 samples per channel)  (rx_sound.cpp:601)
//                    S-meter, m_Agc, detector, m_AM_FIR / m_Squelch, de-emphasis -> kg_post_process_dev         (:676-908)
//                    payload: ADPCM / raw mono / (s2_t) IQ pairs -> kg_adpcm_encode_dev / kg_snd_payload_dev / kg_snd_iq_payload_dev
//                                                                                                               (:1035-1140)
//                    header: flags, sequence number, S-meter     -> kg_snd_header                               (:1222-1253)
// and, on the client's SET commands, what rx_sound_cmd.cpp does: kg_fir_setup, kg_post_set_agc / _set_mode / _set_am_passband /
// _squelch_setup / _squelch_set / _set_deemp (the de-emphasis coefficients are the caller's, rx/rx_filter.h: read from cfg.bin).
// Compressed mono audio goes out as one packet per four 512-sample blocks (`while (bc < LOOP_BC)`, :1216), everything else per block.
//
//   sound_dropin <cfg.bin> <raw.bin> <out.bin>
// cfg.bin: int32 nch, nsamps (records per channel and SPI buffer), nbuf; float rate, rescale (rx/data_pump.cpp:73-74); then per channel
//          { int32 mode (KG_POST_*), agc[6] (on, hang, thresh, manGain, slope, decay), de_emp, nfm, squelch, compression, little_endian;
//            float lo, hi; float deemp_taps[79] }       (taps: the rx_filter.h row the channel's de-emphasis setting selects, or zeros)
// raw.bin: nbuf buffers of nsamps * nch rx_iq_t records (6 bytes each), sample-major / channel-minor
// out.bin: packets in the order they complete: int32 channel, int32 bytes, then the header (10) + payload bytes
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

#pragma pack(push, 1)
struct chan_cfg { int32_t mode, agc[6], de_emp, nfm, squelch, compression, little_endian; float lo, hi; float deemp[79]; };
#pragma pack(pop)
enum { SND_FLAG_ADC_OVFL = 0x02, SND_FLAG_MODE_IQ = 0x08, SND_FLAG_COMPRESSED = 0x10, SND_FLAG_SQUELCH_UI = 0x40,
       SND_FLAG_LITTLE_ENDIAN = 0x80 };                  // rx/rx_sound.cpp:461-468

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: sound_dropin <cfg.bin> <raw.bin> <out.bin>\n"); return 2; }
    std::vector<unsigned char> cfgb, raw;
    if (!read_all(argv[1], cfgb) || !read_all(argv[2], raw)) { fprintf(stderr, "cannot read inputs\n"); return 2; }
    int32_t nch, nsamps, nbuf; float rate, rescale;
    memcpy(&nch, &cfgb[0], 4); memcpy(&nsamps, &cfgb[4], 4); memcpy(&nbuf, &cfgb[8], 4); memcpy(&rate, &cfgb[12], 4); memcpy(&rescale, &cfgb[16], 4);
    const chan_cfg *cc = (const chan_cfg *) &cfgb[20];
    if (cfgb.size() != 20 + (size_t) nch * sizeof(chan_cfg) || raw.size() != (size_t) nbuf * nsamps * nch * 6) { fprintf(stderr, "bad sizes\n"); return 2; }
    FILE *outf = fopen(argv[3], "wb");
    if (!outf) return 2;

    kg_ctx *ctx; kg_fir *fir; kg_post *post; kg_adpcm *ad;
    CHECK(kg_ctx_create(0, NULL, &ctx));
    CHECK(kg_fir_create(ctx, nch, nsamps, &fir));
    CHECK(kg_post_create(ctx, nch, &post));
    CHECK(kg_adpcm_create(ctx, nch, &ad));
    const bool r12k = rate < 16000.f;
    for (int ch = 0; ch < nch; ch++) {                   // the connection's first commands (rx_sound_cmd.cpp)
        const chan_cfg &c = cc[ch];
        const int fmax = (int) (rate / 2 - 1);           // the handler clamps the client's cuts first (rx_sound_cmd.cpp:248-250)
        const float lo = c.lo < -fmax ? (float) -fmax : c.lo, hi = c.hi > fmax ? (float) fmax : c.hi;
        if (kg_fir_setup(fir, ch, lo, hi, 0.f, rate, -1, 0, r12k ? 0 : 1) != 0) { fprintf(stderr, "kg_fir_setup(%d): %s\n", ch, kg_last_error()); return 1; }
        CHECK(kg_post_set_smeter(post, ch, rate));
        CHECK(kg_post_set_agc(post, ch, c.agc[0], c.agc[1], c.agc[2], c.agc[3], c.agc[4], c.agc[5], rate));
        CHECK(kg_post_set_am_passband(post, ch, lo, hi, rate));
        CHECK(kg_post_squelch_setup(post, ch, rate));
        CHECK(kg_post_squelch_set(post, ch, c.squelch, 0));
        if (c.de_emp) CHECK(kg_post_cfir_init_const(post, ch, c.nfm ? KG_CFIR_DEEMP_NFM : KG_CFIR_DEEMP_AM_SSB, 79, c.deemp, rate));
        CHECK(kg_post_set_deemp(post, ch, c.nfm, c.de_emp));
        CHECK(kg_post_set_mode(post, ch, c.mode));
        CHECK(kg_post_reset(post, ch));
    }

    void *d_raw, *d_in, *d_fir, *d_s16, *d_agc, *d_pay;
    const size_t in_stride = (size_t) nsamps, blk = KG_FIR_OUT;
    CHECK(kg_dev_alloc(ctx, (size_t) nsamps * nch * 6, &d_raw));
    CHECK(kg_dev_alloc(ctx, (size_t) nch * in_stride * 8, &d_in));
    CHECK(kg_dev_alloc(ctx, (size_t) nch * blk * 8, &d_fir));
    CHECK(kg_dev_alloc(ctx, (size_t) nch * blk * 2, &d_s16));
    CHECK(kg_dev_alloc(ctx, (size_t) nch * blk * 8, &d_agc));
    CHECK(kg_dev_alloc(ctx, (size_t) nch * blk * 4, &d_pay));
    std::vector<int32_t> chans(nch), nout(nch);
    std::vector<uint8_t> enabled(nch, 1);
    for (int ch = 0; ch < nch; ch++) chans[ch] = ch;
    std::vector<std::vector<uint8_t> > pend(nch);        // a compressed connection's packet under construction
    std::vector<int> pend_blocks(nch, 0);
    std::vector<uint32_t> seq(nch, 0);
    std::vector<uint8_t> pay((size_t) nch * blk * 4);
    std::vector<float> avg(nch), taps(2 * (size_t) nch);
    std::vector<int32_t> sq_rc(nch), sq(nch);
    std::vector<float> sq_ave(nch);

    for (int b = 0; b < nbuf; b++) {                     // data_pump(): one SPI buffer per interrupt
        CHECK(kg_dev_upload(ctx, d_raw, &raw[(size_t) b * nsamps * nch * 6], (size_t) nsamps * nch * 6));
        CHECK(kg_dpump_unpack_dev(ctx, d_raw, nsamps, nch, enabled.data(), rescale, 0.f, 0.f, 0, d_in, in_stride));
        CHECK(kg_fir_process_dev(fir, chans.data(), nch, d_in, in_stride, nsamps, d_fir, blk, nout.data()));
        while (kg_ctx_poll(ctx) == 0) { }                // where c2s_sound() yields (NextTask)
        std::vector<int32_t> ready;
        for (int ch = 0; ch < nch; ch++) if (nout[ch] == (int32_t) blk) ready.push_back(ch);
        if (ready.empty()) continue;
        const int nr = (int) ready.size();
        // rows of d_fir are by list position of `chans` (= channel here); the post stage takes the ready channels' rows in place:
        // every stage below is called once per ready channel set whose rows are contiguous -- here one call per channel keeps the
        // example short (the bank, examples/rxbank_dropin.cpp, batches them)
        for (int i = 0; i < nr; i++) {
            const int ch = ready[i];
            const chan_cfg &c = cc[ch];
            const char *fir_row = (const char *) d_fir + (size_t) ch * blk * 8;
            char *s16_row = (char *) d_s16 + (size_t) ch * blk * 2, *agc_row = (char *) d_agc + (size_t) ch * blk * 8;
            char *pay_row = (char *) d_pay + (size_t) ch * blk * 4;
            const int32_t one = ch;
            CHECK(kg_post_process_dev(post, &one, 1, fir_row, blk, (int) blk, s16_row, NULL, agc_row, blk));
            const bool iq = c.mode == KG_POST_IQ;
            int bytes;
            if (iq) { CHECK(kg_snd_iq_payload_dev(ctx, NULL, 1, agc_row, blk, (int) blk, c.little_endian, pay_row, blk * 4)); bytes = (int) blk * 4; }
            else if (c.compression) { CHECK(kg_adpcm_encode_dev(ad, &one, 1, s16_row, blk, (int) blk, pay_row, blk / 2)); bytes = (int) blk / 2; }
            else { CHECK(kg_snd_payload_dev(ctx, s16_row, blk, 1, (int) blk, c.little_endian, pay_row, blk * 2)); bytes = (int) blk * 2; }
            CHECK(kg_ctx_sync(ctx));
            CHECK(kg_dev_download(ctx, &pay[0], pay_row, (size_t) bytes));
            pend[ch].insert(pend[ch].end(), pay.begin(), pay.begin() + bytes);
            pend_blocks[ch]++;
            if (pend[ch].size() < 1024) continue;        // while (bc < LOOP_BC)
            CHECK(kg_post_smeter(post, &one, 1, &avg[ch], &taps[2 * (size_t) ch]));
            CHECK(kg_post_squelch_state(post, &one, 1, &sq_rc[ch], &sq[ch], &sq_ave[ch]));
            uint8_t hdr[10];
            const uint8_t flags = (uint8_t) ((iq ? SND_FLAG_MODE_IQ : 0) | (c.compression && !iq ? SND_FLAG_COMPRESSED : 0)
                                             | (sq[ch] ? SND_FLAG_SQUELCH_UI : 0) | (c.little_endian ? SND_FLAG_LITTLE_ENDIAN : 0));
            kg_snd_header(flags, ++seq[ch], avg[ch] + (float) -13 /* S_meter_cal */, hdr);
            const int32_t rec[2] = {ch, (int32_t) (10 + pend[ch].size())};
            fwrite(rec, 4, 2, outf);
            fwrite(hdr, 1, 10, outf);
            fwrite(pend[ch].data(), 1, pend[ch].size(), outf);
            pend[ch].clear(); pend_blocks[ch] = 0;
        }
    }
    fclose(outf);
    kg_dev_free(ctx, d_raw); kg_dev_free(ctx, d_in); kg_dev_free(ctx, d_fir); kg_dev_free(ctx, d_s16); kg_dev_free(ctx, d_agc); kg_dev_free(ctx, d_pay);
    kg_adpcm_destroy(ad); kg_post_destroy(post); kg_fir_destroy(fir); kg_ctx_destroy(ctx);
    return 0;
}
