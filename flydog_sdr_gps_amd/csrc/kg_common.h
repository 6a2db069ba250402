// kg_common.h -- host-side plumbing shared by the libkiwigpu translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
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
};

// Device scratch of at least `bytes`, filled from `src` before returning (synchronous: the
// previous user of the scratch is drained first).  Valid until the next call on this context.
int kg_ctx_scratch_upload(kg_ctx *ctx, const void *src, size_t bytes, void **d_out);

// Make ctx's device current on the calling thread.
static inline int kg_ctx_use(kg_ctx *ctx)
{
    KG_REQUIRE(ctx != nullptr, KG_ERR_INVALID, "null context");
    KG_HIP(hipSetDevice(ctx->device));
    return KG_OK;
}
