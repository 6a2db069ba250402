#!/bin/bash
# Builds oracle/_ref/: the pieces of the reference that compile from their own sources, each
# compiled where it lies under $REFERENCE (nothing is copied into the repo; _ref/ is git-ignored
# and holds binaries and generated files only).  Test infrastructure, run in the build container;
# the GPU box uses the prebuilt files.
#
#   1. the reference's own eCPU assembler (e_cpu/asm/*.cpp) is compiled and run, in a temporary
#      directory, on the reference's kiwi.config / *.asm  ->  _ref/gen/kiwi.gen.h (the generated
#      header most reference translation units include);
#   2. against the reference's headers + that kiwi.gen.h:
#        cacode_ref    gps/cacode.h                      (C/A generator)
#        e1b_ref       gps/e1bcode.h                     (the 50 Galileo E1-B memory codes)
#        gpsconst_ref  gps/gps.h, kiwi.h, kiwi.gen.h     (the constants the oracle hard-pins)
#        agc_ref       rx/CuteSDR/agc.cpp                (CAgc)
#        adpcm_ref     rx/csdr/ima_adpcm.cpp             (IMA ADPCM coder/decoder)
#        fir_ref       rx/CuteSDR/fir.cpp                (CFir: m_AM_FIR, the de-emphasis filters)
#        squelch_ref   rx/CuteSDR/squelch.cpp + fir.cpp  (CSquelch: the NBFM noise squelch)
#        cic_gen_ref   verilog/rx/cic_gen.c              (Hogenauer pruning generator; its .vh
#                                                         output for every CIC instance -> _ref/cic/)
#
#   3. (round 6) the FFT-dependent files, against the FFTW3 API the image ships (hipFFTW) -- built here, run on the GPU box:
#        fastfir_ref   rx/CuteSDR/fastfir.cpp + support/simd.cpp                (CFastFIR)
#        search_ref    gps/search.cpp (included by the driver: its functions are static) + sats.cpp + simd.cpp
#                                                                                (SearchInit, Sample, Correlate, decimators)
#        wf_ref        rx/rx_waterfall.cpp + ima_adpcm.cpp + CuteSDR/noiseproc.cpp + rx_util.cpp + support/misc.cpp
#                                                         (c2s_waterfall_init, compute_frame, aperture_auto)
#        dpump_ref     rx/data_pump.cpp (included by the driver)                  (snd_service: runs here, no transform)
#        chan_ref      gps/channel.cpp + ephemeris.cpp + sats.cpp                  (CHANNEL::Start: runs here)
#
#        sndpath_ref   rx/rx_sound.cpp:676-908, 1035-1140, 1222-1253 (LINE RANGES of c2s_sound(): S-meter, detectors, SSB AGC, de-emphasis,
#                                                         payload, header; cut at build time, see below) + agc.cpp, fir.cpp, squelch.cpp,
#                                                         ima_adpcm.cpp      (runs here)
#        wfcmd_ref     rx/rx_waterfall.cpp:365-529, 756-928 (LINE RANGES of c2s_waterfall(): the `SET zoom=` case, the map / scale /
#                                                         mask construction; cut at build time) + support/str.cpp   (runs here)
#        sndcmd_ref    rx/rx_sound_cmd.cpp                (rx_sound_set_freq: the audio NCO's phase increment; runs here)
# NOT built: the rest of the two coroutines (connection handling, packet assembly, the noise blankers: outside SURVEY 8).
set -e
REFERENCE=${REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
CXX=${CXX:-g++}
CC=${CC:-gcc}
if [ ! -f "$REFERENCE/gps/cacode.h" ]; then
    echo "reference tree absent: using prebuilt oracle/_ref/ if any"
    exit 0
fi
mkdir -p "$OUT/gen" "$OUT/cic"

# ---- 1. kiwi.gen.h from the reference's own assembler
W=$(mktemp -d /tmp/kiwiref.XXXXXX)
trap 'rm -rf "$W"' EXIT
mkdir -p "$W/e_cpu" "$W/verilog" "$W/gen"
# the assembler reads ./kiwi.asm, ./*.config, ../kiwi.config and writes ../verilog/*.vh: give it a
# scratch directory of that shape (temporary, deleted on exit)
cp "$REFERENCE"/e_cpu/*.asm "$REFERENCE"/e_cpu/*.config "$W/e_cpu/"
cp "$REFERENCE/kiwi.config" "$W/"
$CXX -I"$REFERENCE/e_cpu" -I"$REFERENCE/e_cpu/asm" -I"$REFERENCE" -D_GNU_SOURCE -w \
    "$REFERENCE"/e_cpu/asm/*.cpp -o "$W/asm"
(cd "$W/e_cpu" && "$W/asm" -o "$W/gen" > "$W/asm.log" 2>&1) || { cat "$W/asm.log"; exit 1; }
cp "$W/gen/kiwi.gen.h" "$W/gen/other.gen.h" "$OUT/gen/"

# ---- 2. the buildable pieces, in place
R=$REFERENCE
INC="-I$R -I$R/gps -I$R/rx -I$R/rx/CuteSDR -I$R/rx/csdr -I$R/rx/kiwi -I$R/rx/wdsp -I$R/rx/Teensy \
 -I$R/support -I$R/platform/common -I$R/platform/beaglebone -I$R/arch/sitara -I$R/init -I$R/net \
 -I$R/web -I$R/dev -I$R/ui -I$R/extensions -I$R/pkgs -I$R/pkgs/mongoose -I$R/pkgs/jsmn \
 -I$R/pkgs/sha256 -I$OUT/gen"
# the defines the reference's Makefile passes for a BeagleBone build (Makefile, Makefile.comp.inc)
DEF="-std=gnu++11 -DKIWI -DKIWISDR -DHOST -DDEBIAN_VERSION=11 -DVERSION_MAJ=1 -DVERSION_MIN=663 \
 -DARCH_CPU=x86 -DCPU_AM3359 -DPLATFORM_beaglebone_black"
# -ffp-contract=off: x86-64 gcc does not contract by default at -O2 without -march flags; stated
# so that the AGC's float/double expressions are evaluated operation by operation
OPT="-O2 -ffp-contract=off -w"
$CXX -O2 -I$R/gps -o "$OUT/cacode_ref" "$HERE/ref/ref_cacode_main.cpp"
$CXX $OPT $DEF $INC -o "$OUT/e1b_ref" "$HERE/ref/ref_e1b_main.cpp"
$CXX $OPT $DEF $INC -o "$OUT/gpsconst_ref" "$HERE/ref/ref_gpsconst_main.cpp"
$CXX $OPT $DEF $INC -o "$OUT/agc_ref" "$HERE/ref/ref_agc_main.cpp" "$R/rx/CuteSDR/agc.cpp" -lm
$CXX $OPT $DEF $INC -o "$OUT/adpcm_ref" "$HERE/ref/ref_adpcm_main.cpp" "$R/rx/csdr/ima_adpcm.cpp"
# fir.cpp's InitLPFilter(..., dump = false) references the server's debug printer real_printf
# (support/printf.cpp, which does not link without the web-server runtime) inside `if (dump)`.  No
# stand-in is written for it: the symbol is left unresolved at link time (it resolves to address 0);
# the drivers never pass dump = true, so it is never called.
UNRES="-Wl,--unresolved-symbols=ignore-all"
$CXX $OPT $DEF $INC -o "$OUT/fir_ref" "$HERE/ref/ref_fir_main.cpp" "$R/rx/CuteSDR/fir.cpp" -lm $UNRES
$CXX $OPT $DEF $INC -o "$OUT/squelch_ref" "$HERE/ref/ref_squelch_main.cpp" "$R/rx/CuteSDR/squelch.cpp" \
    "$R/rx/CuteSDR/fir.cpp" -lm $UNRES
# ---- 3. the FFT-dependent files, against the FFTW3 API the image ships.
# The reference takes FFTW3 from the distribution (`apt-get install libfftw3-dev`, Makefile:365-366); this image has no
# libfftw3f, but ROCm ships hipFFTW: AMD's implementation of the fftw3.h interface (fftwf_plan_dft_1d, fftwf_execute, ...) over
# hipFFT -- /opt/rocm/include/hipfft/hipfftw.h, /opt/rocm/lib/libhipfftw.so.  The reference includes the interface as <fftw3.h>:
# a directory holding ONE symlink of that name to the image's header is put on the include path (under _ref/, not in the
# repository; no header text is written).  hipFFTW runs its transforms on the GPU, so these binaries are built HERE and RUN ON
# THE GPU BOX (tools/make_ref_fft_golden.py; they need nothing of $REFERENCE at run time).
HIPFFTW_H=${HIPFFTW_H:-/opt/rocm/include/hipfft/hipfftw.h}
if [ -f "$HIPFFTW_H" ] && [ -f /opt/rocm/lib/libhipfftw.so ]; then
    mkdir -p "$OUT/fftw3_api"
    ln -sf "$HIPFFTW_H" "$OUT/fftw3_api/fftw3.h"
    EXT=$(for d in "$R"/extensions/*/; do printf -- "-I%s " "$d"; done)
    PKG=$(for d in "$R"/pkgs/*/; do printf -- "-I%s " "$d"; done)
    FINC="-I$OUT/fftw3_api -I$(dirname "$HIPFFTW_H") -I/opt/rocm/include $INC $EXT $PKG"
    FLIB="-L/opt/rocm/lib -lhipfftw -Wl,-rpath,/opt/rocm/lib -lm $UNRES"
    # CFastFIR (rows A3, A4): fastfir.cpp + the reference's simd.cpp; the driver supplies the three globals fastfir.cpp reads
    $CXX $OPT $DEF $FINC -o "$OUT/fastfir_ref" "$HERE/ref/ref_fastfir_main.cpp" "$R/rx/CuteSDR/fastfir.cpp" "$R/support/simd.cpp" $FLIB
    # acquisition (rows G3-G7, G9): the driver TU includes gps/search.cpp itself (Sample(), Correlate() and the decimators are
    # static there) and defines the server-runtime entry points that file calls -- scheduler yields, log printer, SPI packet
    # read, task start: no arithmetic (oracle/ref/ref_search_main.cpp says which and why); gps/sats.cpp and support/simd.cpp
    # are the reference's
    $CXX $OPT $DEF $FINC -no-pie -DREF_SEARCH_CPP="\"$R/gps/search.cpp\"" -o "$OUT/search_ref" "$HERE/ref/ref_search_main.cpp" \
        "$R/gps/sats.cpp" "$R/support/simd.cpp" $FLIB
    # waterfall frames (rows W1, W4-W8): rx_waterfall.cpp's c2s_waterfall_init() and compute_frame() + the reference's ima_adpcm.cpp;
    # the driver supplies a no-op for the command-hash setup and the per-frame inputs the c2s_waterfall() coroutine would have set
    # (rx_waterfall.cpp holds a global CNoiseProc array: the reference's own rx/CuteSDR/noiseproc.cpp supplies its constructor)
    # aperture_auto() (static, called by compute_frame() when wf->aper == AUTO) uses dB_wire_to_dBm() of rx/rx_util.cpp and
    # qsort_intcomp() of support/misc.cpp: both files linked in place; misc.cpp names DIR_CFG, which the reference's Makefile
    # passes (Makefile:278, :496)
    cut_lines() {   # file first last out 'text the first line must contain' 'text the last line must contain'
        sed -n "${2}p" "$1" | grep -qF -- "$5" || { echo "build_ref.sh: $1:$2 is not '$5'"; exit 1; }
        sed -n "${3}p" "$1" | grep -qF -- "$6" || { echo "build_ref.sh: $1:$3 is not '$6'"; exit 1; }
        sed -n "${2},${3}p" "$1" > "$4"
    }
    # sample_wf()'s unpack + window of the samples of one SPI chunk (row W3: rx_waterfall.cpp:1046-1066, with the declarations of
    # :1011-1012) is a line range of a function that pulls its chunks over SPI between coroutine sleeps: cut at build time into the
    # temporary directory and #included by the driver once per chunk of the test's frame (the ninth pass of the reference's own chunk
    # loop re-windows a stale buffer, SURVEY 8 W3: the driver hands over every chunk fresh)
    mkdir -p "$W/wfcut"
    cut_lines "$R/rx/rx_waterfall.cpp" 1011 1012 "$W/wfcut/wf_window_decls.inc" 's4_t ii, qq;' 'iq_t *iqp;'
    cut_lines "$R/rx/rx_waterfall.cpp" 1046 1066 "$W/wfcut/wf_window.inc" 'iqp = (iq_t*) &(miso->word[0]);' '}'
    sed -n '1065p' "$R/rx/rx_waterfall.cpp" | grep -qF 'sn++;' || { echo "build_ref.sh: the window loop does not end at rx_waterfall.cpp:1066"; exit 1; }
    $CXX $OPT $DEF $FINC -I"$W/wfcut" '-DWF_CUT_WINDOW_DECLS="wf_window_decls.inc"' '-DWF_CUT_WINDOW="wf_window.inc"' \
        '-DDIR_CFG=STRINGIFY(/root/kiwi.config)' -no-pie -o "$OUT/wf_ref" "$HERE/ref/ref_wf_main.cpp" \
        "$R/rx/rx_waterfall.cpp" "$R/rx/csdr/ima_adpcm.cpp" "$R/rx/CuteSDR/noiseproc.cpp" "$R/rx/rx_util.cpp" "$R/support/misc.cpp" $FLIB
    # the data pump's unpack (rows A1, A2): no FFT is called, but data_pump.cpp's headers need the FFTW3 API header; the driver TU
    # includes rx/data_pump.cpp itself (snd_service() is static) and defines the SPI / scheduler entry points it calls.  Runs HERE.
    $CXX $OPT $DEF $FINC -no-pie -DREF_DATA_PUMP_CPP="\"$R/rx/data_pump.cpp\"" -o "$OUT/dpump_ref" "$HERE/ref/ref_dpump_main.cpp" -lm $UNRES
    # the hand-off (8(f) rank 3): CHANNEL::Start of gps/channel.cpp + ephemeris.cpp + sats.cpp; its results are the SPI commands
    # it sends, which the driver's _spi_set records.  Runs HERE.
    GPSD=$(for d in $(find "$R/gps" -type d); do printf -- "-I%s " "$d"; done)
    $CXX $OPT $DEF $GPSD $FINC -no-pie -o "$OUT/chan_ref" "$HERE/ref/ref_chan_main.cpp" "$R/gps/channel.cpp" "$R/gps/ephemeris.cpp" \
        "$R/gps/sats.cpp" -lm $UNRES
    # c2s_sound()'s signal path between CFastFIR and the bytes of the sound packet -- the S-meter loop, the AM and NBFM detectors with what
    # follows them, the SSB AGC, the de-emphasis filters (rx/rx_sound.cpp:676-908), the payload section (IQ AGC + (s2_t) pairs, ADPCM / raw,
    # either byte order: 1035-1140), the header's S-meter field, flags and sequence number (1222-1253), the IQ header's GPS stamp (536-537, 557, 638-661) -- is the body of a server coroutine: no function to call,
    # and the file as a whole needs the web server.  The STATEMENTS are compiled instead: the line ranges are cut out of the file where
    # it lies into the temporary directory (deleted on exit; nothing of the text enters the repository or oracle/_ref/) and
    # oracle/ref/ref_sndpath_main.cpp #includes them inside a function that declares c2s_sound()'s locals (by cuts of its own
    # declaration lines where they are declarations).  Each cut is checked to begin and end where this recipe expects.  Linked with
    # the reference's agc.cpp, fir.cpp, squelch.cpp in place.  No transform: runs HERE.
    SND="$R/rx/rx_sound.cpp"
    mkdir -p "$W/sndcut"
    cut_lines "$SND" 92 93 "$W/sndcut/snd_gpsconst.inc" 'const double gps_delay    = ' 'const double gps_week_sec = '
    cut_lines "$SND" 244 250 "$W/sndcut/snd_decls.inc" 'double z1 = 0;' 'float sMeterAvg_dB = 0, sMeter_dBm;'
    cut_lines "$SND" 306 319 "$W/sndcut/snd_norm.inc" 'int ref_nrx_samps = NRX_SAMPS_CHANS(8);' '}'
    cut_lines "$SND" 536 537 "$W/sndcut/snd_ticks.inc" 'const u64_t ticks   = rx->ticks[rx->rd_pos];' 'const u64_t dticks  = time_diff48(ticks, clk.ticks);'
    cut_lines "$SND" 557 557 "$W/sndcut/snd_gpssec.inc" 's->gpssec = fmod(gps_week_sec + clk.gps_secs + (dticks/clk.adc_clock_base) - gps_delay + gps_delay2, gps_week_sec);' 's->gpssec = fmod('
    cut_lines "$SND" 638 661 "$W/sndcut/snd_gpsstamp.inc" 'int sample_filter_delays = norm_nrx_samps - fir_pos;' 's->last_gpssec = s->gpssec;'
    cut_lines "$SND" 252 255 "$W/sndcut/snd_pktinit.inc" 'strncpy(s->out_pkt_real.h.id, "SND", 3);' 's->seq = 0;'
    cut_lines "$SND" 285 285 "$W/sndcut/snd_masked.inc" 'bool masked = false, masked_area = false, check_masked = false;' 'bool masked = false'
    cut_lines "$SND" 295 295 "$W/sndcut/snd_overload.inc" 'bool squelched_overload = false;' 'bool squelched_overload = false;'
    cut_lines "$SND" 461 482 "$W/sndcut/snd_flags.inc" '#define	SND_FLAG_LPF' 'bool do_de_emp = '
    cut_lines "$SND" 488 497 "$W/sndcut/snd_hooks.inc" 'u2_t bc = 0;' 'tid_t receive_real_tid'
    cut_lines "$SND" 676 908 "$W/sndcut/snd_path.inc" 'TYPECPX *s_samps_c = fir_samps_c;' '}'
    sed -n '907p' "$SND" | grep -qF '}' && sed -n '898p' "$SND" | grep -qF 'if (do_de_emp) {' || { echo "build_ref.sh: the de-emphasis block is not at rx_sound.cpp:898-908"; exit 1; }
    cut_lines "$SND" 1035 1140 "$W/sndcut/snd_packet.inc" '#define SILENCE_VALUE 1' '}'
    sed -n '1100p' "$SND" | grep -qF 'if (!isDRM) {' && sed -n '1142p' "$SND" | grep -qF '#ifdef DRM' || { echo "build_ref.sh: the packet section is not at rx_sound.cpp:1035-1140"; exit 1; }
    cut_lines "$SND" 1222 1253 "$W/sndcut/snd_header.inc" '#define SMETER_BIAS 127.0' 'wf->snd_seq = s->seq;'
    ALLD=$(find "$R/rx" "$R/extensions" "$R/pkgs" -maxdepth 2 -type d | sed 's/^/-I/' | tr '\n' ' ')
    $CXX $OPT $DEF $FINC $ALLD -I"$W/sndcut" '-DSND_CUT_GPSCONST="snd_gpsconst.inc"' '-DSND_CUT_NORM="snd_norm.inc"' '-DSND_CUT_TICKS="snd_ticks.inc"' \
        '-DSND_CUT_GPSSEC="snd_gpssec.inc"' '-DSND_CUT_GPSSTAMP="snd_gpsstamp.inc"' '-DSND_CUT_DECLS="snd_decls.inc"' '-DSND_CUT_PKTINIT="snd_pktinit.inc"' '-DSND_CUT_MASKED="snd_masked.inc"' \
        '-DSND_CUT_OVERLOAD="snd_overload.inc"' '-DSND_CUT_FLAGS="snd_flags.inc"' '-DSND_CUT_HOOKS="snd_hooks.inc"' '-DSND_CUT_PATH="snd_path.inc"' \
        '-DSND_CUT_PACKET="snd_packet.inc"' '-DSND_CUT_HEADER="snd_header.inc"' -no-pie -o "$OUT/sndpath_ref" "$HERE/ref/ref_sndpath_main.cpp" \
        "$R/rx/CuteSDR/agc.cpp" "$R/rx/CuteSDR/fir.cpp" "$R/rx/CuteSDR/squelch.cpp" "$R/rx/csdr/ima_adpcm.cpp" "$R/support/timing.cpp" -lm $UNRES
    # the same for what c2s_waterfall() derives from `SET zoom= start=` / `cf=` (rows W2, W6's map): the command's case block with the
    # decimation / NCO words it hands to spi_set / spi_set3, and the fft_used / plot_width / map / scale / mask construction -- line ranges of
    # rx/rx_waterfall.cpp cut into the temporary directory, #included by oracle/ref/ref_wfcmd_main.cpp in the coroutine's own order;
    # kiwi_str_begins_with is the reference's (support/str.cpp in place).  Runs HERE.
    WFC="$R/rx/rx_waterfall.cpp"
    mkdir -p "$W/wfcut"
    cut_lines "$WFC" 67 69 "$W/wfcut/wf_macros.inc" '#define MAX_FFT_USED' '#define	MAX_START(z)'
    cut_lines "$WFC" 217 221 "$W/wfcut/wf_bits.inc" '#define	CMD_ZOOM	0x01' '#define	CMD_ALL'
    cut_lines "$WFC" 253 269 "$W/wfcut/wf_locals.inc" 'int i, j, k, n;' 'int wf_cal = waterfall_cal;'
    cut_lines "$WFC" 271 283 "$W/wfcut/wf_init.inc" 'wf = &WF_SHMEM->wf_inst[rx_chan];' 'int n_chunks = WF_SHMEM->n_chunks;'
    cut_lines "$WFC" 365 529 "$W/wfcut/wf_zoom.inc" 'case CMD_SET_ZOOM: {' '}'
    sed -n '528p' "$WFC" | grep -qF 'break;' && sed -n '531p' "$WFC" | grep -qF 'case CMD_SET_MAX_MIN_DB:' || { echo "build_ref.sh: the CMD_SET_ZOOM case does not end at rx_waterfall.cpp:529"; exit 1; }
    cut_lines "$WFC" 756 928 "$W/wfcut/wf_maps.inc" 'wf->fft_used = WF_C_NFFT / WF_USING_HALF_FFT;' '}'
    sed -n '927p' "$WFC" | grep -qF 'new_scale_mask = false;' || { echo "build_ref.sh: the scale / mask block does not end at rx_waterfall.cpp:928"; exit 1; }
    $CXX $OPT $DEF $FINC $ALLD -I"$W/wfcut" '-DWF_CUT_MACROS="wf_macros.inc"' '-DWF_CUT_BITS="wf_bits.inc"' '-DWF_CUT_LOCALS="wf_locals.inc"' \
        '-DWF_CUT_INIT="wf_init.inc"' '-DWF_CUT_ZOOM="wf_zoom.inc"' '-DWF_CUT_MAPS="wf_maps.inc"' -no-pie -o "$OUT/wfcmd_ref" \
        "$HERE/ref/ref_wfcmd_main.cpp" "$R/support/str.cpp" -lm $UNRES
    # the audio NCO's phase increment (row D6): rx_sound_set_freq() of rx/rx_sound_cmd.cpp, the file linked in place; the driver records
    # the words it hands to spi_set3.  Runs HERE.
    # ... and the passband statements of its `SET mod= low_cut= high_cut=` handler (:243-272, 276-286: clamp, normalised passband, hbw / stop,
    # m_AM_FIR's design), cut at build time like the ranges above; the lines between the cuts are the two CFastFIR::SetupParameters calls
    SCMD="$R/rx/rx_sound_cmd.cpp"
    mkdir -p "$W/sndcmdcut"
    cut_lines "$SCMD" 243 272 "$W/sndcmdcut/pb_a.inc" 'bool no_pb_change = (_hicut == 0 && _locut == 0);' ''
    sed -n '269p' "$SCMD" | grep -qF 'float hbw = fmaxf(fabs(s->hicut), fabs(s->locut));' && sed -n '274p' "$SCMD" | grep -qF 'm_PassbandFIR[rx_chan].SetupParameters' \
        && sed -n '275p' "$SCMD" | grep -qF 'm_chan_null_FIR[rx_chan].SetupParameters' || { echo "build_ref.sh: the passband statements are not at rx_sound_cmd.cpp:243-286"; exit 1; }
    cut_lines "$SCMD" 276 286 "$W/sndcmdcut/pb_b.inc" 'conn->half_bw = hbw;' '}'
    sed -n '282p' "$SCMD" | grep -qF 'm_AM_FIR[rx_chan].InitLPFilter(0, 1.0, 50.0, hbw, stop, frate);' || { echo "build_ref.sh: m_AM_FIR's design is not at rx_sound_cmd.cpp:282"; exit 1; }
    $CXX $OPT $DEF $FINC $ALLD -I"$W/sndcmdcut" '-DSNDCMD_CUT_PB_A="pb_a.inc"' '-DSNDCMD_CUT_PB_B="pb_b.inc"' -no-pie -o "$OUT/sndcmd_ref" \
        "$HERE/ref/ref_sndcmd_main.cpp" "$R/rx/rx_sound_cmd.cpp" "$R/rx/CuteSDR/fir.cpp" -lm $UNRES
    FFT_BUILT=" fastfir_ref search_ref wf_ref dpump_ref chan_ref sndpath_ref wfcmd_ref sndcmd_ref"
else
    echo "hipFFTW absent: the FFT-dependent reference files are not built"
    FFT_BUILT=""
fi
$CC -O1 -w -DKIWISDR -I"$OUT/gen" "$R/verilog/rx/cic_gen.c" -lm -o "$OUT/cic_gen_ref"
(cd "$OUT/cic" && "$OUT/cic_gen_ref" > cic_gen.log 2>&1)
echo "built oracle/_ref: cacode_ref e1b_ref gpsconst_ref agc_ref adpcm_ref fir_ref squelch_ref cic_gen_ref$FFT_BUILT (+ gen/kiwi.gen.h, cic/*.vh) from $REFERENCE"
