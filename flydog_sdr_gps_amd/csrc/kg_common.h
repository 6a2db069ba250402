// kg_common.h -- host-side plumbing shared by the libkiwigpu translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/kiwigpu.h"

void kg_set_error(const char *fmt, ...);

#define KG_HIP(call)                                                               \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            kg_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,            \
                         hipGetErrorString(e_));                                   \
            return KG_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

#define KG_REQUIRE(cond, status, ...)                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            kg_set_error(__VA_ARGS__);                                             \
            return (status);                                                       \
        }                                                                          \
    } while (0)

// One step's small tables of EVERY stage of a receiver bank in one pinned host block, uploaded with ONE transfer
// (kg_rxbank.hip).  While a context's `arena` is set and active, kg_ctx_stage / kg_ctx_stage_cached[_ways] neither copy
// nor cache: in the PLAN pass they append the table to the host block and answer the address it will have on the
// device (the entry point then returns before its first launch and before it touches its state: KG_PLAN_ONLY); in the
// REPLAY pass -- the same calls with the same arguments, after the block's one upload was enqueued -- they answer the same
// addresses again, checking that the table is the one planned.
enum { KG_ARENA_OFF = 0, KG_ARENA_PLAN = 1, KG_ARENA_REPLAY = 2 };
#define KG_ARENA_MAX_ENTRIES 192     /* <= 12 tables per step + 4 per sound block of the step (kg_rxbank_create checks its step against it) */
struct kg_arena {
    int mode;
    unsigned char *h_base, *d_base;           // the current slot of the owner's ring (pinned host / device)
    size_t cap, used;
    int nent, cursor;
    size_t off[KG_ARENA_MAX_ENTRIES], len[KG_ARENA_MAX_ENTRIES];
};

struct kg_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int num_cus;
    char name[256];
    hipEvent_t ev_start, ev_stop;
    // twiddle tables shared by every transform, built in double on the host
    float2 *d_tab4096;    // exp(+2 pi i k / 4096),  k < 4096
    float2 *d_tab16384;   // exp(+2 pi i k / 16384), k < 16384
    float2 *d_tab8192;    // exp(+2 pi i k / 8192),  k < 8192
    // small per-call tables (descriptor lists) of entry points that own no object
    void *d_scratch;
    size_t scratch_bytes;
    // Staging ring for the small per-call tables of the enqueue-only entry points (channel lists,
    // per-channel counts): KG_RING_SLOTS pinned host slots and as many device slots.  A call copies
    // its table into the next host slot and enqueues the transfer to the matching device slot; the
    // kernels it then enqueues read the device slot.  A slot comes round again KG_RING_SLOTS uploads
    // later and waits (normally not at all) for the event recorded half a ring after its last use.
    unsigned char *h_ring, *d_ring;
    hipEvent_t ring_ev[32];
    unsigned long ring_next;
    kg_arena *arena;                          // a receiver bank's step tables (null: the ring / the caches above)
    // A receiver bank's buffers hold one row per RECEIVER while its calls list only the receivers that are active (or, for
    // the coders, that completed a sound block this step): on the bank's contexts the row of list entry li, in every
    // caller-visible buffer of the enqueue-only entry points, is chans[li] instead of li.
    int rows_by_chan;
};

// An enqueue-only entry point that stages its tables through kg_ctx_stage* puts this right behind the (last) staging
// call: in a bank's PLAN pass the call ends here -- nothing launched, no state advanced.
#define KG_PLAN_ONLY(ctx_) do { if ((ctx_)->arena && (ctx_)->arena->mode == KG_ARENA_PLAN) return KG_OK; } while (0)

#define KG_RING_SLOTS 32
#define KG_RING_SLOT_BYTES ((size_t) 512 * 1024)

// A device copy of src[0..bytes) for kernels enqueued on ctx->stream after this call (and before
// KG_RING_SLOTS / 2 further kg_ctx_stage calls).  Does not synchronise the stream in steady state.
// Tables larger than a slot take the synchronous scratch buffer.
int kg_ctx_stage(kg_ctx *ctx, const void *src, size_t bytes, void **d_out);

// The same for a table that is usually the SAME from call to call (a channel list, the frame ->
// channel map of a batch): the owner keeps a kg_stage_cache; an unchanged table (memcmp with the
// host copy) costs no transfer at all, a changed one goes through the ring and, in stream order, into
// the cache's own device buffer (kernels still reading the old contents were enqueued before it).
struct kg_stage_cache { unsigned char *host; void *dev; size_t cap, bytes; };
int kg_ctx_stage_cached(kg_ctx *ctx, kg_stage_cache *sc, const void *src, size_t bytes, void **d_out);
void kg_stage_cache_free(kg_stage_cache *sc);       // the caller has drained the stream
// The same with `ways` caches looked up in turn and replaced round robin: a caller whose table cycles through a few
// variants (the frame tables of a streaming waterfall: the slow channel's frame completes every fourth push, so four
// tables alternate) uploads each of them once.  With one way every push of bench.py's cfg2_chain re-uploaded 64 KiB:
// a pinned-ring copy + two enqueued transfers, ~34 us of copy engines in line with the kernels of a 0.58 ms step.
int kg_ctx_stage_cached_ways(kg_ctx *ctx, kg_stage_cache *sc, int ways, int *victim, const void *src, size_t bytes, void **d_out);

// Device scratch of at least `bytes`, filled from `src` before returning (synchronous: the
// previous user of the scratch is drained first).  Valid until the next call on this context.
int kg_ctx_scratch_upload(kg_ctx *ctx, const void *src, size_t bytes, void **d_out);

// The library's tuning switches (KIWIGPU_DDC_RUNS, KIWIGPU_DDC_SIDE, KIWIGPU_DDC_ENDREF, KIWIGPU_DDC_STAGED,
// KIWIGPU_RXDDC_ENDREF, KIWIGPU_WF_WGS_PER_CU, KIWIGPU_ACQ_WGS_PER_CU, KIWIGPU_ACQ_FRONT_STREAM) are A/B aids: they are read
// ONLY when KIWIGPU_TUNING=1 is set too -- a stray variable in a host's environment does not change what the library does.
static inline const char *kg_tuning_env(const char *name)
{
    const char *on = getenv("KIWIGPU_TUNING");
    return (on && on[0] == '1' && on[1] == 0) ? getenv(name) : nullptr;
}

// Make ctx's device current on the calling thread.
static inline int kg_ctx_use(kg_ctx *ctx)
{
    KG_REQUIRE(ctx != nullptr, KG_ERR_INVALID, "null context");
    KG_HIP(hipSetDevice(ctx->device));
    return KG_OK;
}

// The DDCs' NCO reads: ph holds the 48-bit accumulator in its top bits, the table index is its top 13 (KG_NCO_* below).
// KG_EXP_NCO selects a timing experiment (WRONG values; bench.py marks such builds): 1 = no table reads at all (what a
// conflict-free NCO could gain at most), 2 = the hardware's v_sin_f32 / v_cos_f32 + scale + round (what replacing the table costs).
#if !defined(KG_EXP_NCO)
#define KG_NCO_COS(tab_, ph_) ((int) (tab_)[((ph_) >> 51) + 2048])
#define KG_NCO_SIN(tab_, ph_) ((int) (tab_)[(ph_) >> 51])
#elif KG_EXP_NCO == 1
#define KG_NCO_COS(tab_, ph_) ((int) ((ph_) >> 50) - 8192)
#define KG_NCO_SIN(tab_, ph_) ((int) (((ph_) >> 49) & 0x3fff) - 8192)
#else
#define KG_NCO_COS(tab_, ph_) ((int) __builtin_rintf(16383.0f * __builtin_amdgcn_cosf((float) (unsigned) ((ph_) >> 51) * (1.0f / 8192.0f))))
#define KG_NCO_SIN(tab_, ph_) ((int) __builtin_rintf(16383.0f * __builtin_amdgcn_sinf((float) (unsigned) ((ph_) >> 51) * (1.0f / 8192.0f))))
#endif

// The DDCs' NCO table (frozen by us: the Xilinx DDS core of verilog/rx/iq_mixer.v is closed IP), as ONE 16-bit sine
// table of KG_NCO_TAB = 10240 entries: T[j] = round(16383 sin(2 pi j / 8192)) for j < 8192 and, BY CONSTRUCTION,
// T[j] = T[j - 8192] beyond.  sin(a) = T[a], cos(a) = T[a + 2048] for a < 8192: two sign-extending 16-bit LDS reads
// from one address.  That this equals round(16383 cos(2 pi a / 8192)) -- the oracle's definition -- is checked once in
// the CPU suite through kg_ddc_nco_table (tests/test_host_cpu.py), not at run time.
#define KG_NCO_TAB 10240
extern "C" __attribute__((visibility("hidden"))) void kg_nco_table_build(short *tab);        // kg_ctx.hip (not part of the ABI: kg_ddc_nco_table is)

// kg_ddc.hip, for kg_rxbank.hip (not part of the ABI): the DDC's second stream supplied by the owner.
struct kg_ddc;
__attribute__((visibility("hidden"))) int kg_ddc_use_side_stream(kg_ddc *ddc, hipStream_t stream);

// The library's own streams come from a per-device pool and go back to it; they are never destroyed.  Which hardware queue a
// NEW stream lands on is the runtime's choice at that moment, and streams created after others had been destroyed were seen
// to get queues on which kernels take turns with other streams' instead of running beside them: an object
// that inherits a stream inherits its placement.  Synchronise a stream before giving it back.
int kg_stream_get(int device, hipStream_t *out);          // the device is current (kg_ctx_use) when this is called
void kg_stream_put(int device, hipStream_t s);
