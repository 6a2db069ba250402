// kg_ctx.hip -- context, error text, stopwatch.
#include "kg_common.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

static thread_local char g_err[512] = "";

void kg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int kg_ctx_scratch_upload(kg_ctx *c, const void *src, size_t bytes, void **d_out)
{
    KG_REQUIRE(c && src && d_out && bytes > 0, KG_ERR_INVALID, "kg_ctx_scratch_upload: bad argument");
    KG_HIP(hipStreamSynchronize(c->stream));
    if (bytes > c->scratch_bytes) {
        if (c->d_scratch) KG_HIP(hipFree(c->d_scratch));
        c->d_scratch = nullptr; c->scratch_bytes = 0;
        const size_t cap = bytes < 4096 ? 4096 : bytes;
        KG_HIP(hipMalloc(&c->d_scratch, cap));
        c->scratch_bytes = cap;
    }
    KG_HIP(hipMemcpy(c->d_scratch, src, bytes, hipMemcpyHostToDevice));
    *d_out = c->d_scratch;
    return KG_OK;
}

// A bank's step tables (kg_arena, kg_common.h): plan = append, replay = the same address again.
static int arena_stage(kg_arena *a, const void *src, size_t bytes, void **d_out)
{
    if (a->mode == KG_ARENA_PLAN) {
        const size_t at = (a->used + 63) & ~(size_t) 63;
        KG_REQUIRE(a->nent < KG_ARENA_MAX_ENTRIES && at + bytes <= a->cap, KG_ERR_NOMEM,
                   "step tables: entry %d of %zu bytes does not fit (%zu of %zu used)", a->nent, bytes, a->used, a->cap);
        memcpy(a->h_base + at, src, bytes);
        a->off[a->nent] = at; a->len[a->nent] = bytes; a->nent++;
        a->used = at + bytes;
        *d_out = a->d_base + at;
        return KG_OK;
    }
    KG_REQUIRE(a->cursor < a->nent && a->len[a->cursor] == bytes && memcmp(a->h_base + a->off[a->cursor], src, bytes) == 0,
               KG_ERR_STATE, "step tables: replayed table %d (%zu bytes) is not the planned one", a->cursor, bytes);
    *d_out = a->d_base + a->off[a->cursor++];
    return KG_OK;
}

int kg_ctx_stage(kg_ctx *c, const void *src, size_t bytes, void **d_out)
{
    KG_REQUIRE(c && src && d_out && bytes > 0, KG_ERR_INVALID, "kg_ctx_stage: bad argument");
    if (c->arena && c->arena->mode != KG_ARENA_OFF) return arena_stage(c->arena, src, bytes, d_out);
    if (bytes > KG_RING_SLOT_BYTES) return kg_ctx_scratch_upload(c, src, bytes, d_out);
    if (!c->h_ring) {
        KG_HIP(hipHostMalloc((void **) &c->h_ring, KG_RING_SLOTS * KG_RING_SLOT_BYTES, hipHostMallocDefault));
        KG_HIP(hipMalloc((void **) &c->d_ring, KG_RING_SLOTS * KG_RING_SLOT_BYTES));
        for (int i = 0; i < KG_RING_SLOTS; i++) KG_HIP(hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming));
        c->ring_next = 0;
    }
    const unsigned long j = c->ring_next++;
    const int slot = (int) (j % KG_RING_SLOTS);
    // The slot was last used by upload j - SLOTS; its consumers were enqueued before upload
    // j - SLOTS/2 was made (no call makes that many uploads before launching), and the event of
    // that upload was recorded in front of it.
    if (j >= KG_RING_SLOTS) KG_HIP(hipEventSynchronize(c->ring_ev[(j - KG_RING_SLOTS / 2) % KG_RING_SLOTS]));
    KG_HIP(hipEventRecord(c->ring_ev[slot], c->stream));
    unsigned char *h = c->h_ring + (size_t) slot * KG_RING_SLOT_BYTES, *d = c->d_ring + (size_t) slot * KG_RING_SLOT_BYTES;
    memcpy(h, src, bytes);
    KG_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    *d_out = d;
    return KG_OK;
}

int kg_ctx_stage_cached(kg_ctx *c, kg_stage_cache *sc, const void *src, size_t bytes, void **d_out)
{
    KG_REQUIRE(c && sc && src && d_out && bytes > 0, KG_ERR_INVALID, "kg_ctx_stage_cached: bad argument");
    if (c->arena && c->arena->mode != KG_ARENA_OFF) return arena_stage(c->arena, src, bytes, d_out);
    if (sc->dev && sc->bytes == bytes && memcmp(sc->host, src, bytes) == 0) { *d_out = sc->dev; return KG_OK; }
    if (bytes > sc->cap) {                      // grows rarely: drain, then replace both copies
        KG_HIP(hipStreamSynchronize(c->stream));
        if (sc->dev) KG_HIP(hipFree(sc->dev));
        free(sc->host);
        sc->dev = nullptr; sc->host = nullptr; sc->cap = sc->bytes = 0;
        const size_t cap = bytes < 4096 ? 4096 : bytes + bytes / 2;
        sc->host = (unsigned char *) malloc(cap);
        KG_REQUIRE(sc->host != nullptr, KG_ERR_NOMEM, "kg_ctx_stage_cached: %zu host bytes", cap);
        KG_HIP(hipMalloc(&sc->dev, cap));
        sc->cap = cap;
    }
    void *d = nullptr;
    const int rc = kg_ctx_stage(c, src, bytes, &d);
    if (rc) { sc->bytes = 0; return rc; }
    KG_HIP(hipMemcpyAsync(sc->dev, d, bytes, hipMemcpyDeviceToDevice, c->stream));
    memcpy(sc->host, src, bytes);
    sc->bytes = bytes;
    *d_out = d;                                 // this call reads the ring copy; later calls the cache
    return KG_OK;
}

int kg_ctx_stage_cached_ways(kg_ctx *c, kg_stage_cache *sc, int ways, int *victim, const void *src, size_t bytes, void **d_out)
{
    KG_REQUIRE(c && sc && victim && src && d_out && bytes > 0 && ways >= 1, KG_ERR_INVALID, "kg_ctx_stage_cached_ways: bad argument");
    if (c->arena && c->arena->mode != KG_ARENA_OFF) return arena_stage(c->arena, src, bytes, d_out);
    for (int w = 0; w < ways; w++)
        if (sc[w].dev && sc[w].bytes == bytes && memcmp(sc[w].host, src, bytes) == 0) { *d_out = sc[w].dev; return KG_OK; }
    const int v = *victim % ways;
    *victim = (v + 1) % ways;
    return kg_ctx_stage_cached(c, &sc[v], src, bytes, d_out);
}

void kg_stage_cache_free(kg_stage_cache *sc)
{
    if (!sc) return;
    if (sc->dev) (void) hipFree(sc->dev);
    free(sc->host);
    sc->dev = nullptr; sc->host = nullptr; sc->cap = sc->bytes = 0;
}

#include <mutex>
static std::mutex g_stream_mutex;
static std::vector<std::pair<int, hipStream_t>> g_stream_pool;

int kg_stream_get(int device, hipStream_t *out)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_mutex);
        for (size_t i = 0; i < g_stream_pool.size(); i++)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (long) i);
                return KG_OK;
            }
    }
    KG_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return KG_OK;
}

void kg_stream_put(int device, hipStream_t s)
{
    if (!s) return;
    std::lock_guard<std::mutex> lk(g_stream_mutex);
    g_stream_pool.push_back(std::make_pair(device, s));
}

__global__ void kg_mark_kernel() {}

// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; streams
// beyond that SHARE a queue and run in order with whoever they share it with.  A receiver bank uses five streams: a HOST that
// runs banks should export GPU_MAX_HW_QUEUES=8 before it initialises HIP (bench.py does).  The library
// does not touch the process environment (rounds 4-5 set the variable from a constructor: a side effect on every other HIP
// user of the process, and a race with getenv in a threaded host).

extern "C" {

int kg_ctx_mark(kg_ctx *c, int tag)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_REQUIRE(tag >= 1 && tag <= 65535, KG_ERR_INVALID, "kg_ctx_mark: tag %d (1..65535)", tag);
    hipLaunchKernelGGL(kg_mark_kernel, dim3(tag), dim3(64), 0, c->stream);
    KG_HIP(hipGetLastError());
    return KG_OK;
}

const char *kg_last_error(void) { return g_err; }

int kg_abi_version(void) { return KG_ABI_VERSION; }

void kg_nco_table_build(short *tab)
{
    for (int j = 0; j < 8192; j++) tab[j] = (short) lrint(16383.0 * sin(2.0 * M_PI * j / 8192.0));
    for (int j = 8192; j < KG_NCO_TAB; j++) tab[j] = tab[j - 8192];
}

int kg_ddc_nco_table(int16_t *cos_tab, int16_t *sin_tab)
{
    KG_REQUIRE(cos_tab && sin_tab, KG_ERR_INVALID, "kg_ddc_nco_table: null argument");
    short tab[KG_NCO_TAB];
    kg_nco_table_build(tab);
    for (int a = 0; a < 8192; a++) { sin_tab[a] = tab[a]; cos_tab[a] = tab[a + 2048]; }
    return KG_OK;
}

const char *kg_strerror(int s)
{
    switch (s) {
    case KG_OK: return "ok";
    case KG_ERR_NO_DEVICE: return "no usable gfx950 device";
    case KG_ERR_INVALID: return "invalid argument";
    case KG_ERR_HIP: return "HIP runtime error";
    case KG_ERR_NOMEM: return "out of memory";
    case KG_ERR_STATE: return "call out of order";
    default: return "unknown status";
    }
}

static int make_table(float2 **d, int n)
{
    std::vector<float2> h(n);
    for (int k = 0; k < n; k++) {
        double a = 2.0 * M_PI * (double) k / (double) n;
        h[k].x = (float) cos(a);
        h[k].y = (float) sin(a);
    }
    // exact values on the axes
    h[0] = make_float2(1.f, 0.f);
    h[n / 4] = make_float2(0.f, 1.f);
    h[n / 2] = make_float2(-1.f, 0.f);
    h[3 * n / 4] = make_float2(0.f, -1.f);
    KG_HIP(hipMalloc((void **) d, sizeof(float2) * n));
    KG_HIP(hipMemcpy(*d, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    return KG_OK;
}

static int ctx_create(int device, void *stream, bool use_given, kg_ctx **out)
{
    KG_REQUIRE(out != nullptr, KG_ERR_INVALID, "kg_ctx_create: out is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        kg_set_error("kg_ctx_create: no HIP device (%s); libkiwigpu has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return KG_ERR_NO_DEVICE;
    }
    KG_REQUIRE(device >= 0 && device < ndev, KG_ERR_INVALID,
               "kg_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
    hipDeviceProp_t prop;
    KG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        kg_set_error("kg_ctx_create: device %d is %s; this library is built for gfx950 only",
                     device, prop.gcnArchName);
        return KG_ERR_NO_DEVICE;
    }
    KG_HIP(hipSetDevice(device));
    kg_ctx *c = (kg_ctx *) calloc(1, sizeof(kg_ctx));
    KG_REQUIRE(c != nullptr, KG_ERR_NOMEM, "kg_ctx_create: calloc");
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    if (use_given) {
        c->stream = (hipStream_t) stream;
        c->own_stream = false;
    } else {
        { const int rc_ = kg_stream_get(device, &c->stream); if (rc_) { free(c); return rc_; } }
        c->own_stream = true;
    }
    KG_HIP(hipEventCreate(&c->ev_start));
    KG_HIP(hipEventCreate(&c->ev_stop));
    int rc;
    if ((rc = make_table(&c->d_tab4096, 4096)) != KG_OK) { kg_ctx_destroy(c); return rc; }
    if ((rc = make_table(&c->d_tab16384, 16384)) != KG_OK) { kg_ctx_destroy(c); return rc; }
    if ((rc = make_table(&c->d_tab8192, 8192)) != KG_OK) { kg_ctx_destroy(c); return rc; }
    *out = c;
    return KG_OK;
}

int kg_ctx_create(int device, void *stream, kg_ctx **out)
{
    return ctx_create(device, stream, stream != nullptr, out);
}

int kg_ctx_create_on_stream(int device, void *stream, kg_ctx **out)
{
    return ctx_create(device, stream, true, out);
}

void kg_ctx_destroy(kg_ctx *c)
{
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipStreamSynchronize(c->stream);
    (void) hipFree(c->d_tab4096);
    (void) hipFree(c->d_tab16384);
    (void) hipFree(c->d_tab8192);
    if (c->d_scratch) (void) hipFree(c->d_scratch);
    if (c->h_ring) {
        (void) hipHostFree(c->h_ring); (void) hipFree(c->d_ring);
        for (int i = 0; i < KG_RING_SLOTS; i++) (void) hipEventDestroy(c->ring_ev[i]);
    }
    (void) hipEventDestroy(c->ev_start);
    (void) hipEventDestroy(c->ev_stop);
    if (c->own_stream) kg_stream_put(c->device, c->stream);
    free(c);
}

int kg_ctx_sync(kg_ctx *c)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_HIP(hipStreamSynchronize(c->stream));
    return KG_OK;
}

int kg_ctx_poll(kg_ctx *c)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    kg_set_error("kg_ctx_poll: %s", hipGetErrorString(e));
    return KG_ERR_HIP;
}

void *kg_ctx_stream(kg_ctx *c) { return c ? (void *) c->stream : nullptr; }

int kg_ctx_device_name(kg_ctx *c, char *buf, size_t len)
{
    KG_REQUIRE(c && buf && len > 0, KG_ERR_INVALID, "kg_ctx_device_name: bad argument");
    snprintf(buf, len, "%s", c->name);
    return KG_OK;
}

int kg_ctx_num_cus(kg_ctx *c) { return c ? c->num_cus : KG_ERR_INVALID; }

int kg_dev_alloc(kg_ctx *c, size_t bytes, void **out)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_REQUIRE(out != nullptr && bytes > 0, KG_ERR_INVALID, "kg_dev_alloc: bad argument");
    *out = nullptr;
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) { kg_set_error("kg_dev_alloc: out of device memory (%zu bytes)", bytes); return KG_ERR_NOMEM; }
    KG_HIP(e);
    return KG_OK;
}

int kg_dev_free(kg_ctx *c, void *p)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_HIP(hipStreamSynchronize(c->stream));
    KG_HIP(hipFree(p));
    return KG_OK;
}

int kg_dev_upload(kg_ctx *c, void *dst, const void *src, size_t bytes)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_REQUIRE(dst && src, KG_ERR_INVALID, "kg_dev_upload: null argument");
    KG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    KG_HIP(hipStreamSynchronize(c->stream));
    return KG_OK;
}

int kg_dev_download(kg_ctx *c, void *dst, const void *src, size_t bytes)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_REQUIRE(dst && src, KG_ERR_INVALID, "kg_dev_download: null argument");
    KG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    KG_HIP(hipStreamSynchronize(c->stream));
    return KG_OK;
}

int kg_dev_mem_info(kg_ctx *c, size_t *free_bytes, size_t *total_bytes)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_REQUIRE(free_bytes && total_bytes, KG_ERR_INVALID, "kg_dev_mem_info: null argument");
    KG_HIP(hipStreamSynchronize(c->stream));
    KG_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return KG_OK;
}

int kg_timer_start(kg_ctx *c)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_HIP(hipEventRecord(c->ev_start, c->stream));
    return KG_OK;
}

int kg_timer_stop(kg_ctx *c, float *ms)
{
    int rc = kg_ctx_use(c);
    if (rc) return rc;
    KG_REQUIRE(ms != nullptr, KG_ERR_INVALID, "kg_timer_stop: ms is null");
    KG_HIP(hipEventRecord(c->ev_stop, c->stream));
    KG_HIP(hipEventSynchronize(c->ev_stop));
    KG_HIP(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return KG_OK;
}

}  // extern "C"
