// Context, staging helpers, rollout state machine and the encode/decode drivers of
// libtezip_hip.so.  Reference control flow: /root/reference/src/compress.py:183-373 and
// /root/reference/src/decompress.py:105-256 (cited per function).
#include <dlfcn.h>
#include <stdarg.h>

#include <algorithm>
#include <chrono>
#include <thread>

#include "tz_internal.h"

// ------------------------------------------------------------------------------ errors
int tz_fail(tz_ctx* ctx, int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return status;
}

// Kernels whose waits are hand-built (k_scan2p polls status words of other workgroups) bound those waits and report an
// expiry here instead of hanging the GPU: one pinned host word the device can write.
int tz_fault_word(tz_ctx* ctx) {
    if (ctx->h_fault) return TZ_OK;
    void* h = nullptr;
    TZ_HIP(ctx, hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped));
    *(volatile unsigned*)h = 0;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipHostFree(h);
        return tz_fail(ctx, TZ_ERR_HIP, "no device pointer for the fault word");
    }
    ctx->h_fault = (volatile unsigned*)h;
    ctx->d_fault = (unsigned*)d;
    return TZ_OK;
}

int tz_stream_sync(tz_ctx* ctx) {
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_fault && *ctx->h_fault) {
        const unsigned f = *ctx->h_fault;
        *ctx->h_fault = 0;
        if (f & TZ_FAULT_SCAN_POLL)
            return tz_fail(ctx, TZ_ERR_HIP,
                           "inverse scan (k_scan2p): a workgroup gave up waiting for the block sum of a workgroup in front of it "
                           "(workgroups not dispatched in index order?); the scanned output of that launch is invalid");
        return tz_fail(ctx, TZ_ERR_HIP, "a kernel reported fault 0x%x", f);
    }
    return TZ_OK;
}

// diagnostic: makes the next inverse scans poll for the epoch `epoch_skew` launches ahead (never published when != 0) and
// give up after `poll_limit` polls (0 = the built-in 2^22); (0, 0) restores normal operation.  tests/test_gpu_parity.py
// uses it to see the bounded wait fail loudly.
extern "C" int tz_scan_fault_inject(tz_ctx* ctx, unsigned epoch_skew, unsigned poll_limit) {
    if (!ctx) return TZ_ERR_INVALID;
    ctx->scan_dbg_skew = epoch_skew & 0xFFFFu;
    ctx->scan_dbg_limit = poll_limit;
    return TZ_OK;
}

extern "C" int tz_version(void) { return 101; }

// What this library was compiled with: "tezip_hip <version> gfx950" and, after "defines:", every diagnostic switch of
// csrc/ that was on (TZW_ABL produces WRONG results by design; TZW_STAMPS / TZW_LEAD / TZW_ISSUE_AT / TZW_PK change the
// kernels that are measured).  A library whose string names any of them is a measurement build: bench.py and the test
// suite refuse it (tezip_amd/_lib.py diagnostic_defines).
extern "C" const char* tz_build_info(void) {
    return "tezip_hip 101 gfx950 defines:"
#ifdef TZW_ABL
           " TZW_ABL"
#endif
#ifdef TZW_STAMPS
           " TZW_STAMPS"
#endif
#ifdef TZW_LEAD
           " TZW_LEAD"
#endif
#ifdef TZW_ISSUE_AT
           " TZW_ISSUE_AT"
#endif
#ifdef TZW_PK
           " TZW_PK"
#endif
        ;
}

int tz_check_pred_contract(tz_ctx* ctx, const char* who) {
    const int now = tz_get_contract(ctx);
    if (ctx->pred_contract && now != ctx->pred_contract)
        return tz_fail(ctx, TZ_ERR_STATE,
                       "%s: the resident predictions were made under TZ-PA%d, the contract in force is now TZ-PA%d "
                       "(tz_set_contract between the rollout and its encode/decode): roll out again",
                       who, ctx->pred_contract, now);
    return TZ_OK;
}

extern "C" int tz_rollout_contract(tz_ctx* ctx) {
    if (!ctx) return TZ_ERR_INVALID;
    if (!ctx->have_rollout) return tz_fail(ctx, TZ_ERR_STATE, "no rollout in this context");
    return ctx->pred_contract;
}

extern "C" const char* tz_strerror(int s) {
    switch (s) {
        case TZ_OK: return "ok";
        case TZ_ERR_INVALID: return "invalid argument";
        case TZ_ERR_NO_DEVICE: return "no HIP device";
        case TZ_ERR_HIP: return "HIP runtime error";
        case TZ_ERR_STATE: return "call order / missing state";
        case TZ_ERR_NOMEM: return "out of device memory";
        case TZ_ERR_UNSUPPORTED: return "unsupported model shape";
    }
    return "unknown status";
}

extern "C" const char* tz_last_error(const tz_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

// ----------------------------------------------------------------------------- context
extern "C" int tz_ctx_create(int device, void* hip_stream, tz_ctx** out) {
    if (!out) return TZ_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TZ_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return TZ_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return TZ_ERR_NO_DEVICE;
    tz_ctx* ctx = new tz_ctx();
    {
        const char* e = getenv("TEZIP_CONV16");  // diagnostic default of tz_set_conv_impl
        if (e && e[0] == '0') ctx->conv_impl = 0;
        e = getenv("TEZIP_LAT");                 // k_convlat: 0 never, 1 cost model (default), 2 wherever eligible
        if (e) ctx->lat_mode = atoi(e);
        e = getenv("TEZIP_PA");                  // arithmetic contract a context starts with (tz_set_contract): 0 (default), 1 or 2
        if (e && e[0] >= '0' && e[0] <= '2' && !e[1]) ctx->contract = e[0] - '0';
        e = getenv("TEZIP_WINO_IPW");            // measurements: column blocks per k_wino workgroup (0 = per launch)
        if (e) ctx->wino_ipw = atoi(e);
    }
    ctx->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cus = prop.multiProcessorCount;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        // a caller's stream: the split gate launches of "E-part ahead" only pay when the compute stream outranks stream2
        // (measured: at equal priority the side workgroups sit on the CUs the critical path wants, +5 % per step)
        int prio = 0, prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamGetPriority(ctx->stream, &prio) != hipSuccess || prio != prio_greatest || prio_greatest == prio_least) {
            (void)hipGetLastError();
            if (ctx->epart_mode < 0) ctx->epart_mode = 0;
        }
    } else {
        // (highest dispatch priority: the side launches on stream2 -- lowest -- are there to fill what this stream leaves idle)
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest) != hipSuccess) {
            delete ctx;
            return TZ_ERR_HIP;
        }
        ctx->own_stream = true;
        if (prio_greatest == prio_least && ctx->epart_mode < 0) ctx->epart_mode = 0;   // no priorities on this device: no side launches by default
    }
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_keys, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_frames, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_compute, hipEventDisableTiming) != hipSuccess) {
        tz_ctx_destroy(ctx);
        return TZ_ERR_HIP;
    }
    {
        const char* e = getenv("TEZIP_SPLIT");
        if (e) ctx->split_rollout = atoi(e);
        e = getenv("TEZIP_DECODE_UNFUSED");
        if (e) ctx->decode_unfused = atoi(e);
        e = getenv("TEZIP_EPART");
        if (e) ctx->epart_mode = atoi(e);
        // stream2 takes work that must not delay the compute stream's (the side launches of "E-part ahead" fill the CUs the
        // critical path leaves idle): lowest dispatch priority the device offers
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_least) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
            tz_ctx_destroy(ctx);
            return TZ_ERR_HIP;
        }
    }
    ctx->ring_size = 1 << 20;
    if (hipHostMalloc((void**)&ctx->ring, ctx->ring_size, hipHostMallocDefault) != hipSuccess) {
        ctx->ring = nullptr;
        ctx->ring_size = 0;
    }
    *out = ctx;
    return TZ_OK;
}

extern "C" int tz_ctx_destroy(tz_ctx* ctx) {
    if (!ctx) return TZ_OK;
    (void)hipSetDevice(ctx->device);
    (void)tz_payload_settle(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    tz_model_free(ctx);
    tz_pool_release_all(ctx);
    for (auto& p : ctx->pool) (void)hipFree(p.first);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    if (ctx->d_pred) (void)hipFree(ctx->d_pred);
    if (ctx->d_sched) (void)hipFree(ctx->d_sched);
    if (ctx->d_payload) (void)hipFree(ctx->d_payload);
    if (ctx->d_payload_stage) (void)hipFree(ctx->d_payload_stage);
    if (ctx->ev_payload) (void)hipEventDestroy(ctx->ev_payload);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->d_scan_status) (void)hipFree(ctx->d_scan_status);
    if (ctx->h_fault) (void)hipHostFree((void*)ctx->h_fault);
    for (auto& s : ctx->prof)
        for (auto& e : s.pending) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    if (ctx->ring) (void)hipHostFree(ctx->ring);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->down_stream) {
        (void)hipStreamSynchronize(ctx->down_stream);
        (void)hipStreamDestroy(ctx->down_stream);
    }
    for (int i = 0; i < tz_ctx::kStages; ++i) {
        if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    }
    for (auto e : ctx->chunk_ev) (void)hipEventDestroy(e);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    for (int l = 0; l < TZ_MAX_LEVELS; ++l) {
        if (ctx->ev_epart_src[l]) (void)hipEventDestroy(ctx->ev_epart_src[l]);
        if (ctx->ev_epart_done[l]) (void)hipEventDestroy(ctx->ev_epart_done[l]);
        if (l < 2 && ctx->ev_cal[l]) (void)hipEventDestroy(ctx->ev_cal[l]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_keys) (void)hipEventDestroy(ctx->ev_keys);
    if (ctx->ev_frames) (void)hipEventDestroy(ctx->ev_frames);
    if (ctx->ev_compute) (void)hipEventDestroy(ctx->ev_compute);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TZ_OK;
}

extern "C" int tz_ctx_synchronize(tz_ctx* ctx) {
    if (!ctx) return TZ_ERR_INVALID;
    TZ_TRY(tz_payload_settle(ctx));
    return tz_stream_sync(ctx);
}

int tz_payload_settle(tz_ctx* ctx) {
    if (!ctx->payload_inflight) return TZ_OK;
    ctx->payload_inflight = false;
    TZ_HIP(ctx, hipEventSynchronize(ctx->ev_payload));
    return TZ_OK;
}

extern "C" int tz_set_payload_deferred(tz_ctx* ctx, int on) {
    if (!ctx) return TZ_ERR_INVALID;
    if (!on) TZ_TRY(tz_payload_settle(ctx));
    ctx->defer_payload = on ? 1 : 0;
    return TZ_OK;
}

extern "C" int tz_payload_wait(tz_ctx* ctx) {
    if (!ctx) return TZ_ERR_INVALID;
    return tz_payload_settle(ctx);
}

extern "C" void* tz_ctx_stream(tz_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---------------------------------------------------------------------- memory helpers
int tz_ptr_kind(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory: clear the sticky error
        return 0;
    }
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return 2;
    return attr.type == hipMemoryTypeHost ? 1 : 0;
}

bool tz_is_device_ptr(const void* p) { return tz_ptr_kind(p) == 2; }

extern "C" int tz_host_alloc(size_t bytes, void** out) {
    if (!out) return TZ_ERR_INVALID;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? TZ_ERR_NO_DEVICE : TZ_ERR_NOMEM;
    }
    return TZ_OK;
}

extern "C" int tz_host_free(void* p) {
    if (!p) return TZ_OK;
    return hipHostFree(p) == hipSuccess ? TZ_OK : TZ_ERR_HIP;
}

// one pinned staging buffer, free to be overwritten (its previous DMA has completed)
static int stage_acquire(tz_ctx* ctx, int* idx) {
    const int i = ctx->stage_next;
    ctx->stage_next = (i + 1) % tz_ctx::kStages;
    if (!ctx->stage[i] || !ctx->stage_ev[i]) {
        hipError_t e = hipSuccess;
        if (!ctx->stage[i]) e = hipHostMalloc((void**)&ctx->stage[i], tz_ctx::kStageBytes, hipHostMallocDefault);
        if (e == hipSuccess && !ctx->stage_ev[i]) e = hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {   // a buffer without its event is no use: the next call starts over
            if (ctx->stage[i]) (void)hipHostFree(ctx->stage[i]);
            ctx->stage[i] = nullptr;
            ctx->stage_ev[i] = nullptr;
            (void)hipGetLastError();
            return tz_fail(ctx, TZ_ERR_NOMEM, "pinned staging buffer: %s", hipGetErrorString(e));
        }
    }
    if (ctx->stage_busy[i]) {
        TZ_HIP(ctx, hipEventSynchronize(ctx->stage_ev[i]));
        ctx->stage_busy[i] = false;
    }
    *idx = i;
    return TZ_OK;
}

// memcpy between pageable memory and a pinned staging buffer on a few threads: one core moves
// ~8-10 GB/s, a PCIe 5 x16 link four to five times that
static void copy_mt(void* dst, const void* src, size_t n) {
    constexpr size_t kMin = (size_t)2 << 20;
    constexpr int kThreads = 4;
    if (n < 2 * kMin) {
        memcpy(dst, src, n);
        return;
    }
    const size_t per = ((n + kThreads - 1) / kThreads + 4095) & ~(size_t)4095;
    std::thread th[kThreads - 1];
    int started = 0;
    for (int i = 1; i < kThreads; ++i) {
        const size_t off = per * i;
        if (off >= n) break;
        th[started++] = std::thread([=] { memcpy((uint8_t*)dst + off, (const uint8_t*)src + off, std::min(per, n - off)); });
    }
    memcpy(dst, src, std::min(per, n));
    for (int i = 0; i < started; ++i) th[i].join();
}

int tz_h2d(tz_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return TZ_OK;
    if (tz_ptr_kind(src) != 0) {
        TZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
        return TZ_OK;
    }
    // pageable: memcpy of chunk k+1 into a pinned buffer overlaps the DMA of chunk k
    for (size_t off = 0; off < bytes; off += tz_ctx::kStageBytes) {
        const size_t n = std::min(tz_ctx::kStageBytes, bytes - off);
        int i;
        TZ_TRY(stage_acquire(ctx, &i));
        copy_mt(ctx->stage[i], (const uint8_t*)src + off, n);
        TZ_HIP(ctx, hipMemcpyAsync((uint8_t*)dst + off, ctx->stage[i], n, hipMemcpyHostToDevice, s));
        TZ_HIP(ctx, hipEventRecord(ctx->stage_ev[i], s));
        ctx->stage_busy[i] = true;
    }
    return TZ_OK;
}

int tz_d2h(tz_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return TZ_OK;
    if (tz_ptr_kind(dst) != 0) {
        TZ_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
        return TZ_OK;
    }
    // pageable: the DMA of chunk k+1 into a pinned buffer overlaps the memcpy of chunk k out of one
    int prev = -1;
    size_t prev_off = 0, prev_n = 0;
    for (size_t off = 0; off < bytes; off += tz_ctx::kStageBytes) {
        const size_t n = std::min(tz_ctx::kStageBytes, bytes - off);
        int i;
        TZ_TRY(stage_acquire(ctx, &i));
        TZ_HIP(ctx, hipMemcpyAsync(ctx->stage[i], (const uint8_t*)src + off, n, hipMemcpyDeviceToHost, s));
        TZ_HIP(ctx, hipEventRecord(ctx->stage_ev[i], s));
        ctx->stage_busy[i] = true;
        if (prev >= 0) {
            TZ_HIP(ctx, hipEventSynchronize(ctx->stage_ev[prev]));
            ctx->stage_busy[prev] = false;
            copy_mt((uint8_t*)dst + prev_off, ctx->stage[prev], prev_n);
        }
        prev = i;
        prev_off = off;
        prev_n = n;
    }
    if (prev >= 0) {
        TZ_HIP(ctx, hipEventSynchronize(ctx->stage_ev[prev]));
        ctx->stage_busy[prev] = false;
        copy_mt((uint8_t*)dst + prev_off, ctx->stage[prev], prev_n);
    }
    return TZ_OK;
}

static constexpr size_t kPoolUsed = (size_t)1 << 63;  // top bit of the size marks "handed out"

// TEZIP_POISON=<byte> (diagnostic): every device buffer handed out -- fresh or recycled -- is first filled with that byte,
// so that a kernel reading something nobody wrote shows up as a parity failure instead of depending on what the memory
// held before (tests/test_gpu_poison.py runs the parity cases this way).
int tz_poison_byte() {
    static const int v = [] {
        const char* e = getenv("TEZIP_POISON");
        return e && *e ? (atoi(e) & 0xff) | 0x100 : 0;
    }();
    return v;
}
int tz_poison(tz_ctx* ctx, void* p, size_t bytes) {
    const int v = tz_poison_byte();
    if (v && p && bytes) {   // finished before the caller goes on: some initialisations are synchronous copies
        TZ_HIP(ctx, hipMemsetAsync(p, v & 0xff, bytes, ctx->stream));
        TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return TZ_OK;
}

int tz_pool_alloc(tz_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i) {
        size_t sz = ctx->pool[i].second;
        if ((sz & kPoolUsed) || sz < bytes) continue;
        if (best < 0 || sz < ctx->pool[best].second) best = i;
    }
    if (best >= 0 && ctx->pool[best].second <= 2 * bytes + (1 << 20)) {
        ctx->pool[best].second |= kPoolUsed;
        *out = ctx->pool[best].first;
        return tz_poison(ctx, *out, ctx->pool[best].second & ~kPoolUsed);
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return tz_fail(ctx, TZ_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ctx->pool.push_back({p, bytes | kPoolUsed});
    *out = p;
    return tz_poison(ctx, p, bytes);
}

void tz_pool_release_all(tz_ctx* ctx) {
    for (auto& e : ctx->pool) e.second &= ~kPoolUsed;
}

int tz_ensure(tz_ctx* ctx, void** buf, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *buf) return TZ_OK;
    if (*buf) {
        TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(*buf);
        *buf = nullptr;
        *cap = 0;
    }
    hipError_t e = hipMalloc(buf, bytes ? bytes : 16);
    if (e != hipSuccess) return tz_fail(ctx, TZ_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    *cap = bytes;
    return tz_poison(ctx, *buf, bytes);
}

int tz_upload(tz_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return TZ_OK;
    size_t need = (bytes + 63) & ~(size_t)63;
    if (!ctx->ring || need > ctx->ring_size / 4) {  // large or no ring: the caller's block is free on return
        TZ_TRY(tz_h2d(ctx, dst, src, bytes, ctx->stream));
        if (tz_ptr_kind(src) != 0) TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return TZ_OK;
    }
    if (ctx->ring_pos + need > ctx->ring_size) {  // wrap: everything queued so far must have left the ring
        TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->ring_pos = 0;
    }
    uint8_t* slot = ctx->ring + ctx->ring_pos;
    ctx->ring_pos += need;
    memcpy(slot, src, bytes);
    TZ_HIP(ctx, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return TZ_OK;
}

int tz_dev_in(tz_ctx* ctx, const void* p, size_t bytes, const void** dev) {
    if (bytes == 0 || tz_is_device_ptr(p)) {
        *dev = p;
        return TZ_OK;
    }
    void* d;
    TZ_TRY(tz_pool_alloc(ctx, bytes, &d));
    TZ_TRY(tz_h2d(ctx, d, p, bytes, ctx->stream));
    if (tz_ptr_kind(p) != 0) TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffer is free again on return
    *dev = d;
    return TZ_OK;
}

int tz_dev_out(tz_ctx* ctx, void* p, size_t bytes, tz_out* o) {
    o->bytes = bytes;
    if (bytes == 0 || tz_is_device_ptr(p)) {
        o->host = nullptr;
        o->dev = p;
        return TZ_OK;
    }
    o->host = p;
    return tz_pool_alloc(ctx, bytes, &o->dev);
}

int tz_dev_out_finish(tz_ctx* ctx, std::vector<tz_out>& outs) {
    bool any = false;
    for (auto& o : outs)
        if (o.host && o.bytes && !o.done) {
            TZ_TRY(tz_d2h(ctx, o.host, o.dev, o.bytes, ctx->stream));
            any = true;
        }
    if (any) TZ_TRY(tz_stream_sync(ctx));
    return TZ_OK;
}

// --------------------------------------------------------------------------- profiling
static const char* kProfNames[TZP_COUNT] = {"conv3x3_mfma", "err0", "delta", "quant", "spatial_delta_hist",
                                            "lut_remap", "undelta_scan", "reconstruct", "sse",
                                            "conv16_lds_dma", "conv16b_level0", "conv_small_valu", "conv3x3_general",
                                            "convlat_small_grid", "wino_pa2", "table_create", "quant_serial_chains"};

namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char* e = getenv("TEZIP_ROCTX");
        if (!e || atoi(e) == 0) return;
        void* h = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (h) {
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
        }
        if (!push || !pop) {
            push = nullptr;
            pop = nullptr;
            fprintf(stderr, "[tezip] TEZIP_ROCTX is set but no ROCTx library could be opened (%s): no ranges\n", dlerror() ? dlerror() : "symbols missing");
        }
    }
};
const RoctxApi& roctx_api() {
    static RoctxApi api;
    return api;
}
}  // namespace

bool tz_roctx_push(const char* name) {
    const RoctxApi& r = roctx_api();
    if (!r.push) return false;
    r.push(name);
    return true;
}
void tz_roctx_pop() {
    const RoctxApi& r = roctx_api();
    if (r.pop) r.pop();
}

tz_prof_scope::tz_prof_scope(tz_ctx* c, int k) : ctx(c), cls(k) {
    rx = tz_roctx_push(k >= 0 && k < TZP_COUNT ? kProfNames[k] : "tz_stage");
    if (!ctx->prof_on) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        a = b = nullptr;
        return;
    }
    (void)hipEventRecord(a, ctx->stream);
}
tz_prof_scope::~tz_prof_scope() {
    if (rx) tz_roctx_pop();
    if (!a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof[cls].pending.push_back({a, b});
    ctx->prof[cls].pending_sub.push_back(sub);
}

extern "C" int tz_prof_enable(tz_ctx* ctx, int on) {
    if (!ctx) return TZ_ERR_INVALID;
    ctx->prof_on = on != 0;
    return TZ_OK;
}
extern "C" int tz_prof_count(void) { return TZP_COUNT; }
extern "C" const char* tz_prof_name(int i) { return (i >= 0 && i < TZP_COUNT) ? kProfNames[i] : ""; }

static int prof_drain(tz_ctx* ctx) {
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& s : ctx->prof) {
        for (size_t k = 0; k < s.pending.size(); ++k) {
            auto& e = s.pending[k];
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
                s.total_ms += ms;
                s.launches += 1;
                const int sub = s.pending_sub[k];
                if (sub >= 0 && sub < TZP_COUNT) {
                    ctx->prof[sub].total_ms += ms;
                    ctx->prof[sub].launches += 1;
                }
            }
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        s.pending.clear();
        s.pending_sub.clear();
    }
    return TZ_OK;
}

extern "C" int tz_prof_get(tz_ctx* ctx, int i, double* total_ms, long long* launches) {
    if (!ctx || i < 0 || i >= TZP_COUNT) return TZ_ERR_INVALID;
    TZ_TRY(prof_drain(ctx));
    if (total_ms) *total_ms = ctx->prof[i].total_ms;
    if (launches) *launches = ctx->prof[i].launches;
    return TZ_OK;
}

extern "C" int tz_prof_reset(tz_ctx* ctx) {
    if (!ctx) return TZ_ERR_INVALID;
    TZ_TRY(prof_drain(ctx));
    for (auto& s : ctx->prof) {
        s.total_ms = 0;
        s.launches = 0;
    }
    return TZ_OK;
}

extern "C" int tz_timer_start(tz_ctx* ctx) {
    if (!ctx) return TZ_ERR_INVALID;
    TZ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return TZ_OK;
}
extern "C" int tz_timer_stop(tz_ctx* ctx, float* ms) {
    if (!ctx || !ms) return TZ_ERR_INVALID;
    TZ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    TZ_HIP(ctx, hipEventSynchronize(ctx->ev1));
    TZ_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return TZ_OK;
}

// ------------------------------------------------------------------------------ rollout
static int pad8(int v) { return (v + 7) / 8 * 8; }  // data_utils.py:103-107

// decompress.py:123-129: is there a non-zero sample in frame f?  16 bytes per lane where the frame allows it.
// Key frames of a PINNED host stack, fetched by the compute stream itself (zero-copy reads over PCIe: page-locked host
// memory is device-addressable): the first predictor step then never waits for a DMA engine that may still be busy with
// the previous sequence's deferred payload (tz_set_payload_deferred) -- HIP hands streams to SDMA engines as it likes, and
// a host -> device copy queued behind a 126 MB device -> host transfer starts 2.3 ms late.
__global__ __launch_bounds__(256) void k_fetch_frames(const uint8_t* __restrict__ host, uint8_t* __restrict__ dev, const int* __restrict__ which,
                                                      size_t frame_bytes) {
    const size_t base = (size_t)which[blockIdx.y] * frame_bytes;
    const size_t n16 = frame_bytes / 16;
    const uint4* s = (const uint4*)(host + base);
    uint4* d = (uint4*)(dev + base);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    if (blockIdx.x == 0 && threadIdx.x < (frame_bytes & 15)) dev[base + n16 * 16 + threadIdx.x] = host[base + n16 * 16 + threadIdx.x];
}

__global__ void k_any_nonzero(const uint8_t* __restrict__ frames, size_t frame_bytes, int* __restrict__ flags) {
    int f = blockIdx.y;
    const uint8_t* p = frames + (size_t)f * frame_bytes;
    int any = 0;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)p) & 15) == 0) {
        const uint4* q = (const uint4*)p;
        const size_t n16 = frame_bytes / 16;
        for (size_t i = tid; i < n16; i += nthr) {
            const uint4 v = q[i];
            any |= (v.x | v.y | v.z | v.w) != 0;
        }
        for (size_t i = n16 * 16 + tid; i < frame_bytes; i += nthr) any |= p[i] != 0;
    } else {
        for (size_t i = tid; i < frame_bytes; i += nthr) any |= p[i] != 0;
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(&flags[f], 1);
}

__global__ void k_bcast_frame(const float* __restrict__ src, size_t fe, const int* __restrict__ slots, int nslots,
                              float* __restrict__ stack) {
    int s = blockIdx.y;
    if (s >= nslots) return;
    float* dst = stack + (size_t)slots[s] * fe;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < fe; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// Host frames travel on the copy stream: the frames in `first` (the key frames the rollout
// reads) go ahead and the compute stream waits for them only; rollout_finish_upload sends the rest
// once the predictor launches are queued, so that the bulk of the stack (needed by the delta stage
// only) crosses PCIe while the predictor runs.
// ---- DWP control on the device (compress.py:245-264)
struct DwpState {
    int key_idx;   // first predicted frame of the open window (the reference's key_idx)
    int pad;
    double run;    // squared-error sum of the open window
};

static constexpr int kDwpLds = 2048;  // partial sums staged per pass of k_dwp_decide (16 KB)

__global__ void k_dwp_init(DwpState* st, int p, int nt, int* idx_table, int stride, uint8_t* key) {
    st->key_idx = p + 1;
    st->run = 0.0;
    idx_table[0] = 1;              // idx == key_idx: the input of the first step is the real frame p
    idx_table[stride] = p;
    idx_table[2 * stride] = p + 1;
    if (p < nt) key[p] = 1;        // compress.py:219-220
}

// The DWP step's window SSE and its decision in ONE launch (round 5; until then k_sse, then a one-workgroup k_dwp_decide,
// then a conditional C0 broadcast: three dependent launches behind every predictor step of a B = 1 rollout).  Every
// workgroup writes its block's partial sum (the arithmetic of k_sse: tz_sse_block) and takes a ticket; the workgroup that
// draws the last ticket -- every partial was performed before its ticket, both as device-scope atomics, and is read back
// with atomics -- adds them in block order exactly as tzk_sse does on the host, and lane 0 decides.
// No workgroup waits for another one.  part[nblk]; *ticket is 0 on entry and is left at 0.
__global__ __launch_bounds__(256) void k_sse_decide(const uint8_t* __restrict__ orig, const float* __restrict__ pred, int H, int W, int Hp,
                                                     int Wp, int nblk, double* part, unsigned* ticket, DwpState* st, int idx, int nt,
                                                     double fe_pad, double threshold, int* idx_table, int stride, uint8_t* key,
                                                     uint8_t* gfirst, double* mse, int* c0_flag) {
    __shared__ double s[256];
    __shared__ double s_part[kDwpLds];
    __shared__ unsigned s_last;
    const double mine = tz_sse_block(orig, pred, H, W, Hp, Wp, blockIdx.x, s);
    if (threadIdx.x == 0) {
        // The partial goes out as a device-scope RETURNING exchange; its return value is consumed by an asm the compiler
        // cannot see through, which also drains vmcnt: the exchange has been performed (its old value has come back) before
        // the ticket increment is even issued.  Two relaxed atomics on different addresses are NOT ordered by issue order
        // (different L2 channels); a C-level "dependency" such as `1u + (old & 0)` is folded away by the compiler and left
        // a non-returning swap with no wait in front of the add (round 5's defect; tests/test_build_guard.py now reads
        // the sequence swap sc0 -> s_waitcnt vmcnt(0) -> add off the object code).  No agent-scope fence: that writes
        // back / invalidates a whole per-XCD L2 (measured: 22 us with __threadfence() on both sides).  The reader below
        // loads part[] with agent-scope atomic loads (`sc1`), after its own ticket add has returned and a barrier.
        unsigned long long old = atomicExch((unsigned long long*)(part + blockIdx.x), (unsigned long long)__double_as_longlong(mine));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");
        s_last = atomicAdd(ticket, 1u) == (unsigned)(nblk - 1);
    }
    __syncthreads();
    if (!s_last) return;
    double t = 0.0;
    for (int b0 = 0; b0 < nblk; b0 += kDwpLds) {
        const int nb = min(kDwpLds, nblk - b0);
        __syncthreads();
        for (int b = threadIdx.x; b < nb; b += blockDim.x)
            s_part[b] = __hip_atomic_load(part + b0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load: never this CU's L1
        __syncthreads();
        if (threadIdx.x == 0)
            for (int b = 0; b < nb; ++b) t = t + s_part[b];
    }
    if (threadIdx.x != 0) return;
    *ticket = 0u;
    int key_idx = st->key_idx;
    double run = st->run + t;
    const double stop = run / ((double)(idx - key_idx + 1) * fe_pad);   // compress.py:246
    mse[idx] = stop;
    if (stop > threshold) {                                            // compress.py:249
        gfirst[idx] = 1;
        if (idx == nt - 1) key[idx] = 1;                               // compress.py:260-262
        else c0_flag[idx] = 1;
        key_idx = idx + 1;
        run = 0.0;
    }
    st->key_idx = key_idx;
    st->run = run;
    const int nidx = idx + 1;                                          // selection of the next step (218-222)
    if (nidx < nt) {
        const int from_key = nidx == key_idx;
        idx_table[0] = from_key;
        idx_table[stride] = nidx - 1;
        idx_table[2 * stride] = nidx;
        if (from_key) key[nidx - 1] = 1;
    }
}

// slot 0 of every group the DWP loop opened holds C0 (compress.py:258): one launch behind the loop, frame = blockIdx.y
// (nothing in the loop reads such a slot: the step after a boundary starts from the real frame)
__global__ void k_bcast_frames_flagged(const float* __restrict__ src, size_t fe, const int* __restrict__ flag, float* __restrict__ pred) {
    if (!flag[blockIdx.y]) return;
    float* dst = pred + (size_t)blockIdx.y * fe;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < fe; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

static int rollout_setup(tz_ctx* ctx, const uint8_t* frames, int nt, int H, int W, int warm_up,
                         const std::vector<int>* first = nullptr) {
    if (!ctx->model) return tz_fail(ctx, TZ_ERR_STATE, "no model loaded");
    if (nt < 1 || H < 1 || W < 1 || warm_up < 0 || nt > 32767 || H > 32767 || W > 32767)
        return tz_fail(ctx, TZ_ERR_INVALID, "bad sequence shape nt=%d H=%d W=%d warm_up=%d (int16 trailer limits)", nt, H, W, warm_up);
    int Hp, Wp, maxB;
    TZ_TRY(tz_model_dims(ctx, &Hp, &Wp, &maxB));
    if (pad8(H) != Hp || pad8(W) != Wp)
        return tz_fail(ctx, TZ_ERR_INVALID,
                       "Image size is out of scope for this model: compatible sizes are height %d to %d and width %d to %d",
                       Hp - 7, Hp, Wp - 7, Wp);  // compress.py:178-181
    if (!frames && (!ctx->staged || ctx->nt != nt || ctx->H != H || ctx->W != W))
        return tz_fail(ctx, TZ_ERR_STATE, "no frame stack of this shape was staged (tz_frames_begin / tz_frames_put)");
    ctx->nt = nt;
    ctx->H = H;
    ctx->W = W;
    ctx->Hp = Hp;
    ctx->Wp = Wp;
    ctx->warm_up = warm_up;
    ctx->have_rollout = false;
    ctx->pending_src = nullptr;
    ctx->pending_sent.clear();
    const size_t fsz = (size_t)H * W * 3;
    size_t fb = (size_t)nt * fsz, pb = (size_t)nt * Hp * Wp * 3 * 4;
    if (frames) ctx->staged = false;
    TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_frames, &ctx->cap_frames, fb));
    TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_pred, &ctx->cap_pred, pb));
    if (!frames) {  // staged by tz_frames_put on the copy stream: the compute stream waits for the last of them
        TZ_HIP(ctx, hipEventRecord(ctx->ev_frames, ctx->copy_stream));
        TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0));
        return TZ_OK;
    }
    if (tz_is_device_ptr(frames)) {
        TZ_HIP(ctx, hipMemcpyAsync(ctx->d_frames, frames, fb, hipMemcpyDeviceToDevice, ctx->stream));
        return TZ_OK;
    }
    // earlier work queued on the compute stream may still read d_frames
    TZ_HIP(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));
    TZ_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_compute, 0));
    if (first && !first->empty() && (int)first->size() < nt) {
        ctx->pending_sent.assign(nt, 0);
        // frames at 16-byte-aligned offsets of a pinned stack: the compute stream reads them itself (k_fetch_frames)
        // (the device's view of the block: the same address for tz_host_alloc memory, possibly another one for memory the
        // caller registered himself; no mapping -> the copy engine as before)
        const uint8_t* dev_view = nullptr;
        bool fetch = tz_ptr_kind(frames) == 1 && fsz % 16 == 0 && first->size() <= 4096;
        if (fetch && (hipHostGetDevicePointer((void**)&dev_view, (void*)frames, 0) != hipSuccess || !dev_view || ((uintptr_t)dev_view & 15))) {
            (void)hipGetLastError();
            fetch = false;
        }
        std::vector<int> which;
        for (int f : *first) {
            if (f < 0 || f >= nt || ctx->pending_sent[f]) continue;
            if (fetch) which.push_back(f);
            else TZ_TRY(tz_h2d(ctx, ctx->d_frames + (size_t)f * fsz, frames + (size_t)f * fsz, fsz, ctx->copy_stream));
            ctx->pending_sent[f] = 1;
        }
        if (fetch && !which.empty()) {
            void* d_which;
            TZ_TRY(tz_pool_alloc(ctx, which.size() * sizeof(int), &d_which));
            TZ_TRY(tz_upload(ctx, d_which, which.data(), which.size() * sizeof(int)));
            const unsigned gx = (unsigned)std::min<size_t>((fsz / 16 + 255) / 256, 256);
            hipLaunchKernelGGL(k_fetch_frames, dim3(gx, (unsigned)which.size()), dim3(256), 0, ctx->stream, dev_view, ctx->d_frames,
                               (const int*)d_which, fsz);
            TZ_HIP(ctx, hipGetLastError());
        } else {
            TZ_HIP(ctx, hipEventRecord(ctx->ev_keys, ctx->copy_stream));
            TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_keys, 0));
        }
        ctx->pending_src = frames;
        return TZ_OK;
    }
    TZ_TRY(tz_h2d(ctx, ctx->d_frames, frames, fb, ctx->copy_stream));
    TZ_HIP(ctx, hipEventRecord(ctx->ev_frames, ctx->copy_stream));
    TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0));
    return TZ_OK;
}

// second half of a split upload: every frame not sent by rollout_setup, in contiguous runs; the
// compute stream continues (delta stage, window MSE) only after they have landed
static int rollout_finish_upload(tz_ctx* ctx) {
    if (!ctx->pending_src) return TZ_OK;
    const size_t fsz = (size_t)ctx->H * ctx->W * 3;
    const uint8_t* src = ctx->pending_src;
    ctx->pending_src = nullptr;
    for (int f = 0; f < ctx->nt;) {
        if (ctx->pending_sent[f]) {
            ++f;
            continue;
        }
        int g = f;
        while (g < ctx->nt && !ctx->pending_sent[g]) ++g;
        TZ_TRY(tz_h2d(ctx, ctx->d_frames + (size_t)f * fsz, src + (size_t)f * fsz, (size_t)(g - f) * fsz, ctx->copy_stream));
        f = g;
    }
    TZ_HIP(ctx, hipEventRecord(ctx->ev_frames, ctx->copy_stream));
    TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0));
    return TZ_OK;
}

// copy C0 into the given slots of the prediction stack
static int fill_c0(tz_ctx* ctx, const std::vector<int>& slots) {
    if (slots.empty()) return TZ_OK;
    const float* c0;
    TZ_TRY(tz_model_c0_dev(ctx, &c0));
    size_t fe = (size_t)ctx->Hp * ctx->Wp * 3;
    void* d_slots;
    TZ_TRY(tz_pool_alloc(ctx, slots.size() * sizeof(int), &d_slots));
    TZ_TRY(tz_upload(ctx, d_slots, slots.data(), slots.size() * sizeof(int)));
    int gx = (int)std::min<size_t>((fe + 255) / 256, 1024);
    hipLaunchKernelGGL(k_bcast_frame, dim3(gx, (unsigned)slots.size()), dim3(256), 0, ctx->stream, c0, fe,
                       (const int*)d_slots, (int)slots.size(), ctx->d_pred);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

// Run a static schedule: items (out frame, from_key, in frame) grouped by depth; every depth
// is one batched predictor call over all windows (their recursions are independent).
struct PredItem {
    int out, from_key, in, depth;
};
// The schedule is static: its index table is uploaded once and every depth is one batched
// predictor call over all windows (launch-only, no per-step copies).  Capturing this sequence
// into a hipGraph was measured and brings nothing (cfg1/cfg2 are bound by the latency of their
// tiny grids, not by launch overhead; cfg3+ are GPU-bound), so the launches stay plain.
static int run_schedule(tz_ctx* ctx, std::vector<PredItem>& items) {
    int Hp, Wp, maxB;
    TZ_TRY(tz_model_dims(ctx, &Hp, &Wp, &maxB));
    if (items.empty()) return TZ_OK;
    std::stable_sort(items.begin(), items.end(), [](const PredItem& a, const PredItem& b) { return a.depth < b.depth; });
    // A window is a chain of items (each reads what the one before it wrote); chains never touch each other.
    // TEZIP_SPLIT=1 deals them to two GROUPS that advance on two streams with their own activation slots: the
    // launches of a predictor step depend on each other, so on one stream every launch ramps up and drains alone
    // (0.895 of the MFMA peak at 4 windows against 0.917 at 32), and with two independent launch
    // chains one group's launch could fill the CUs the other's draining launch leaves idle.  Measured in round 3
    // (bit-identical, all GPU tests green with it on): 62.5 -> 65.4 ms per cfg3 step -- two half-sized launches
    // lose more to their own ramps than the overlap returns -- so it is OFF by default and kept as a switch.
    // Per-launch event timing needs launches that run alone: profiling keeps one stream in any case.
    const bool split = ctx->split_rollout && ctx->stream2 && maxB >= 2 && !ctx->prof_on;
    const int capA = split ? (maxB + 1) / 2 : maxB, capB = maxB - capA;
    std::vector<int> group(items.size(), 0);
    if (split) {
        std::vector<int> chain_of(ctx->nt, -1);   // frame slot -> group of the chain that wrote it
        int nchains = 0;
        for (size_t i = 0; i < items.size(); ++i) {
            int g;
            if (items[i].from_key || chain_of[items[i].in] < 0) g = (nchains++) & 1;
            else g = chain_of[items[i].in];
            group[i] = g;
            chain_of[items[i].out] = g;
        }
    }
    // per group: batches of one depth, [is_key | in | out | next slot] x maxB ints each.  next slot: where the item's
    // prediction sits in the NEXT batch of its group when that batch holds its consumer (the next step of the window);
    // the prediction kernel then writes the consumer's level-0 error maps itself, and a batch all of whose items were
    // served that way runs without its k_err0 launch (18 of the 19 steps of a cfg3 rollout).
    std::vector<int> table;
    std::vector<int> counts[2];
    std::vector<size_t> offs[2];
    std::vector<char> all_fed[2];   // per batch: every item's E_0 slot comes from the batch in front
    for (int g = 0; g < (split ? 2 : 1); ++g) {
        const int cap = g == 0 ? capA : capB;
        std::vector<std::vector<size_t>> batches;
        size_t i = 0;
        while (i < items.size()) {
            const int depth = items[i].depth;
            size_t j = i;
            while (j < items.size() && items[j].depth == depth) ++j;
            std::vector<size_t> mine;
            for (size_t k = i; k < j; ++k)
                if (group[k] == g) mine.push_back(k);
            for (size_t q = 0; q < mine.size(); q += cap)
                batches.emplace_back(mine.begin() + q, mine.begin() + std::min(mine.size(), q + (size_t)cap));
            i = j;
        }
        for (size_t b = 0; b < batches.size(); ++b) {
            const size_t nb = batches[b].size(), base = table.size();
            table.resize(base + 4 * (size_t)maxB, 0);
            for (size_t k = 0; k < nb; ++k) {
                const PredItem& it = items[batches[b][k]];
                table[base + k] = it.from_key;
                table[base + maxB + k] = it.in;
                table[base + 2 * maxB + k] = it.out;
                int next = -1;
                if (b + 1 < batches.size())
                    for (size_t k2 = 0; k2 < batches[b + 1].size(); ++k2) {
                        const PredItem& c = items[batches[b + 1][k2]];
                        if (!c.from_key && c.in == it.out) next = (int)k2;
                    }
                table[base + 3 * maxB + k] = next;
            }
            bool fed = b > 0;
            if (b > 0)
                for (size_t k = 0; k < nb && fed; ++k) {
                    const PredItem& c = items[batches[b][k]];
                    bool found = false;
                    for (size_t k0 = 0; k0 < batches[b - 1].size() && !found; ++k0)
                        found = !c.from_key && items[batches[b - 1][k0]].out == c.in;
                    fed = found;
                }
            counts[g].push_back((int)nb);
            offs[g].push_back(base);
            all_fed[g].push_back(fed ? 1 : 0);
        }
    }
    TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_sched, &ctx->cap_sched, table.size() * sizeof(int)));
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));  // an earlier rollout may still read the old table
    TZ_HIP(ctx, hipMemcpy(ctx->d_sched, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice));
    if (split && !counts[1].empty()) {
        TZ_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));          // everything queued so far (frames, C0 slots)
        TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    hipStream_t main_stream = ctx->stream;
    int rc = TZ_OK;
    const size_t steps = std::max(counts[0].size(), counts[1].size());
    bool fused[2] = {false, false};   // did the previous batch of the group write the next one's error maps?
    for (size_t b = 0; b < steps && rc == TZ_OK; ++b) {
        for (int g = 0; g < 2 && rc == TZ_OK; ++g) {
            if (b >= counts[g].size()) continue;
            const int slot0 = g == 0 ? 0 : capA;
            const int* tab = ctx->d_sched + offs[g][b];
            // the slot numbers of the table are positions in the batch: the activation slots of a group start at slot0
            const bool skip = fused[g] && all_fed[g][b];
            if (g == 1) ctx->stream = ctx->stream2;   // the launchers take the context's stream
            rc = tz_model_predict_batch_dev(ctx, counts[g][b], tab, maxB, ctx->d_frames, ctx->H, ctx->W, ctx->d_pred, ctx->d_pred, slot0,
                                            tab + 3 * (size_t)maxB, skip, &fused[g]);
            ctx->stream = main_stream;
        }
    }
    if (split && !counts[1].empty()) {
        hipError_t e = hipEventRecord(ctx->ev_join, ctx->stream2);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
        if (e != hipSuccess && rc == TZ_OK) rc = tz_fail(ctx, TZ_ERR_HIP, "rollout join: %s", hipGetErrorString(e));
    }
    return rc;
}

// ---- streaming ingestion / delivery: the frame stack enters window by window and the payload
// leaves chunk by chunk, so that the host never holds more than a few windows (SURVEY.md §8f-3;
// the reference keeps everything in RAM, compress.py:116-122,329-333).
extern "C" int tz_frames_begin(tz_ctx* ctx, int nt, int H, int W) {
    if (!ctx) return TZ_ERR_INVALID;
    if (nt < 1 || H < 1 || W < 1 || nt > 32767 || H > 32767 || W > 32767)
        return tz_fail(ctx, TZ_ERR_INVALID, "bad sequence shape nt=%d H=%d W=%d (int16 trailer limits)", nt, H, W);
    ctx->have_rollout = false;
    ctx->staged = false;
    TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_frames, &ctx->cap_frames, (size_t)nt * H * W * 3));
    // earlier work queued on the compute stream may still read d_frames
    TZ_HIP(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));
    TZ_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_compute, 0));
    ctx->nt = nt;
    ctx->H = H;
    ctx->W = W;
    ctx->staged = true;
    return TZ_OK;
}

extern "C" int tz_frames_put(tz_ctx* ctx, int first, int count, const uint8_t* frames) {
    if (!ctx || !frames) return TZ_ERR_INVALID;
    if (!ctx->staged) return tz_fail(ctx, TZ_ERR_STATE, "tz_frames_put needs a tz_frames_begin first");
    if (first < 0 || count < 0 || first + count > ctx->nt) return tz_fail(ctx, TZ_ERR_INVALID, "frames [%d, %d) outside the stack", first, first + count);
    const size_t fsz = (size_t)ctx->H * ctx->W * 3;
    return tz_h2d(ctx, ctx->d_frames + (size_t)first * fsz, frames, (size_t)count * fsz, ctx->copy_stream);
}

extern "C" int tz_frames_fence(tz_ctx* ctx) {
    if (!ctx) return TZ_ERR_INVALID;
    TZ_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return TZ_OK;
}

extern "C" int tz_frames_get(tz_ctx* ctx, int first, int count, uint8_t* out) {
    if (!ctx || !out) return TZ_ERR_INVALID;
    if (!ctx->d_frames || first < 0 || count < 0 || first + count > ctx->nt)
        return tz_fail(ctx, TZ_ERR_INVALID, "frames [%d, %d) outside the resident stack", first, first + count);
    const size_t fsz = (size_t)ctx->H * ctx->W * 3;
    TZ_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    TZ_TRY(tz_d2h(ctx, out, ctx->d_frames + (size_t)first * fsz, (size_t)count * fsz, ctx->stream));
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TZ_OK;
}

extern "C" int tz_payload_begin(tz_ctx* ctx, size_t count) {
    if (!ctx) return TZ_ERR_INVALID;
    ctx->enc_pending = false;   // the resident symbols of a tz_encode_begin are about to be overwritten
    TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_payload, &ctx->cap_payload, std::max<size_t>(count, 8) * 2));
    ctx->payload_len = count;
    TZ_HIP(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));  // earlier work may still read the old payload
    TZ_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_compute, 0));
    return TZ_OK;
}

extern "C" int tz_payload_put(tz_ctx* ctx, size_t offset, size_t count, const int16_t* src) {
    if (!ctx || !src) return TZ_ERR_INVALID;
    if (!ctx->d_payload || offset + count > ctx->payload_len) return tz_fail(ctx, TZ_ERR_INVALID, "payload range outside the staged payload");
    return tz_h2d(ctx, ctx->d_payload + offset, src, count * 2, ctx->copy_stream);
}

extern "C" int tz_decoded_get(tz_ctx* ctx, int first, int count, uint8_t* out) {
    if (!ctx || !out) return TZ_ERR_INVALID;
    if (!ctx->d_out || !ctx->have_decoded || first < 0 || count < 0 || first + count > ctx->nt)
        return tz_fail(ctx, TZ_ERR_INVALID, "frames [%d, %d) outside the resident decoded stack", first, first + count);
    const size_t fsz = (size_t)ctx->H * ctx->W * 3;
    TZ_TRY(tz_d2h(ctx, out, ctx->d_out + (size_t)first * fsz, (size_t)count * fsz, ctx->stream));
    return tz_stream_sync(ctx);
}

extern "C" int tz_payload_get(tz_ctx* ctx, size_t offset, size_t count, int16_t* out) {
    if (!ctx || !out) return TZ_ERR_INVALID;
    if (!ctx->d_payload || offset + count > ctx->payload_len) return tz_fail(ctx, TZ_ERR_INVALID, "payload range outside the resident payload");
    TZ_TRY(tz_d2h(ctx, out, ctx->d_payload + offset, count * 2, ctx->stream));
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TZ_OK;
}

extern "C" int tz_rollout(tz_ctx* ctx, const uint8_t* frames, int nt, int H, int W, int warm_up, int window,
                          double threshold, uint8_t* key_mask, double* mse_log) {
    tz_roctx_range roctx_("tz_rollout");
    if (!ctx) return TZ_ERR_INVALID;
    ctx->enc_pending = false;   // a tz_encode_begin belongs to the rollout before it
    if (window < 0) return tz_fail(ctx, TZ_ERR_INVALID, "window must be >= 0");
    if (nt < warm_up + 2)  // the reference breaks here
        return tz_fail(ctx, TZ_ERR_INVALID, "need at least warm_up+2 frames (nt=%d, warm_up=%d)", nt, warm_up);
    const int p = warm_up;
    const bool dwp = window == 0;
    std::vector<int> first;  // SWP reads exactly the frames that start a window (compress.py:218-220)
    if (!dwp)
        for (int f = p; f < nt; f += window) first.push_back(f);
    int rc = rollout_setup(ctx, frames, nt, H, W, warm_up, dwp ? nullptr : &first);
    if (rc != TZ_OK) {
        tz_pool_release_all(ctx);
        return rc;
    }
    const bool want_mse = dwp || mse_log != nullptr;
    std::vector<uint8_t> key(nt, 0), gfirst(nt, 0), qskip(nt, 0);
    std::vector<double> mse(nt, 0.0);
    std::vector<int> c0_slots;
    // compress.py:188-211: warm-up frames are key frames whose "prediction" is C0
    for (int i = 0; i < p; ++i) {
        key[i] = 1;
        qskip[i] = 1;  // error_bound is skipped for group 0 (compress.py:315)
        c0_slots.push_back(i);
    }
    gfirst[0] = 1;
    if (p > 0) {
        gfirst[p] = 1;
        c0_slots.push_back(p);
    } else {
        c0_slots.push_back(0);
    }
    const size_t fe_pad = (size_t)ctx->Hp * ctx->Wp * 3;
    if (!dwp) {
        // SWP: window boundaries are known up front (compress.py:249: (idx-p) % w == 0), so all
        // windows advance together.  The prediction the reference makes and drops at a boundary
        // (251-253) is only evaluated when the MSE log is wanted (-v prints it, 245-247).
        std::vector<PredItem> items;
        std::vector<int> dropped;
        int key_idx = p + 1;
        for (int idx = p + 1; idx < nt; ++idx) {
            bool from_key = idx == key_idx;
            if (from_key) key[idx - 1] = 1;
            bool trig = (idx - p) % window == 0;
            if (trig) gfirst[idx] = 1;
            if (trig && !want_mse) {
                c0_slots.push_back(idx);
            } else {
                items.push_back(PredItem{idx, from_key ? 1 : 0, idx - 1, idx - (key_idx - 1)});
                if (trig && idx != nt - 1) dropped.push_back(idx);  // the last frame keeps it (260-262)
            }
            if (trig) {
                if (idx == nt - 1) key[idx] = 1;  // compress.py:260-262
                key_idx = idx + 1;
            }
        }
        rc = fill_c0(ctx, c0_slots);
        if (rc == TZ_OK) rc = run_schedule(ctx, items);
        if (rc == TZ_OK) rc = rollout_finish_upload(ctx);
        if (rc == TZ_OK && want_mse) {
            std::vector<double> sse(nt, 0.0);
            rc = tzk_sse(ctx, ctx->d_frames, ctx->d_pred, nt, H, W, ctx->Hp, ctx->Wp, sse.data());
            int k0 = p + 1;
            double run = 0.0;
            for (int idx = p + 1; idx < nt && rc == TZ_OK; ++idx) {
                run = run + sse[idx];
                mse[idx] = run / (double)((size_t)(idx - k0 + 1) * fe_pad);
                if (gfirst[idx]) {
                    k0 = idx + 1;
                    run = 0.0;
                }
            }
            if (rc == TZ_OK) rc = fill_c0(ctx, dropped);  // slot 0 of the next group holds C0 (258)
        }
    } else {
        // DWP: boundaries depend on the window MSE of the padded frames (compress.py:245-249).  The
        // decision is taken ON THE DEVICE (k_dwp_decide): it sums the frame's partial squared errors
        // in the fixed order, compares the window mean with the threshold, marks the key frame and
        // writes the input selection of the next predictor step, so the host only queues launches
        // -- no round trip per frame -- and reads the key mask / MSE log once at the end.
        rc = fill_c0(ctx, c0_slots);
        int Hp_, Wp_, maxB;
        if (rc == TZ_OK) rc = tz_model_dims(ctx, &Hp_, &Wp_, &maxB);
        const int nblk = tzk_sse_blocks(ctx->Hp, ctx->Wp);
        void *d_state = nullptr, *d_part = nullptr, *d_key = nullptr, *d_gf = nullptr, *d_mse = nullptr, *d_flag = nullptr;
        if (rc == TZ_OK) rc = tz_ensure(ctx, (void**)&ctx->d_sched, &ctx->cap_sched, 3 * (size_t)maxB * sizeof(int));
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, sizeof(DwpState), &d_state);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, sizeof(double) * nblk, &d_part);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nt, &d_key);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nt, &d_gf);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, sizeof(double) * nt, &d_mse);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, sizeof(int) * nt, &d_flag);
        void *d_ticket = nullptr, *d_slot0 = nullptr;   // (d_slot0: one int 0 = "the prediction is the input of slot 0 of the next step")
        bool prev_fused = false;
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, 256, &d_ticket);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, 256, &d_slot0);
        if (rc == TZ_OK) rc = tz_upload(ctx, d_key, key.data(), nt);
        if (rc == TZ_OK) rc = tz_upload(ctx, d_gf, gfirst.data(), nt);
        if (rc == TZ_OK) {
            hipError_t e = hipMemsetAsync(d_mse, 0, sizeof(double) * nt, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(int) * nt, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_ticket, 0, 256, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_slot0, 0, 256, ctx->stream);
            if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "DWP state: %s", hipGetErrorString(e));
        }
        const float* c0 = nullptr;
        if (rc == TZ_OK) rc = tz_model_c0_dev(ctx, &c0);
        if (rc == TZ_OK) {
            hipLaunchKernelGGL(k_dwp_init, dim3(1), dim3(1), 0, ctx->stream, (DwpState*)d_state, p, nt, ctx->d_sched, maxB,
                               (uint8_t*)d_key);
            if (hipGetLastError() != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "k_dwp_init launch failed");
        }
        const int gx = (int)std::min<size_t>((fe_pad + 255) / 256, 1024);
        for (int idx = p + 1; idx < nt && rc == TZ_OK; ++idx) {
            // the prediction kernel also writes the level-0 error maps of the NEXT step as if that step went on from this
            // prediction (as in the static schedule); the next step's error unit then runs for a key-frame start only --
            // which of the two it is, k_sse_decide says on the device
            bool fused = false;
            static const bool spec = !getenv("TEZIP_DWP_SPEC") || atoi(getenv("TEZIP_DWP_SPEC")) != 0;   // (0: measurements)
            rc = tz_model_predict_batch_dev(ctx, 1, ctx->d_sched, maxB, ctx->d_frames, H, W, ctx->d_pred, ctx->d_pred, 0,
                                            spec ? (const int*)d_slot0 : nullptr, false, &fused, prev_fused);
            prev_fused = fused;
            if (rc != TZ_OK) break;
            {
                tz_prof_scope ps(ctx, TZP_SSE);
                hipLaunchKernelGGL(k_sse_decide, dim3(nblk), dim3(256), 0, ctx->stream, ctx->d_frames + (size_t)idx * H * W * 3,
                                   ctx->d_pred + (size_t)idx * fe_pad, H, W, ctx->Hp, ctx->Wp, nblk, (double*)d_part, (unsigned*)d_ticket,
                                   (DwpState*)d_state, idx, nt, (double)fe_pad, threshold, ctx->d_sched, maxB, (uint8_t*)d_key,
                                   (uint8_t*)d_gf, (double*)d_mse, (int*)d_flag);
            }
            if (hipGetLastError() != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "DWP launch failed");
        }
        // slot 0 of every group opened on the way holds C0 (258); the last frame keeps its prediction (260-262: k_sse_decide
        // does not flag it)
        if (rc == TZ_OK && nt > 1) {
            // (gridDim.y = nt: rollout_setup refuses nt > 32767 -- the trailer stores it as int16, compress.py:390-394 -- so
            // the 65535 limit of a grid's y dimension is never reached)
            static_assert(32767 <= 65535, "nt limit vs gridDim.y");
            hipLaunchKernelGGL(k_bcast_frames_flagged, dim3(std::min(gx, 128), nt), dim3(256), 0, ctx->stream, c0, fe_pad, (const int*)d_flag, ctx->d_pred);
            if (hipGetLastError() != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "DWP launch failed");
        }
        if (rc == TZ_OK) {
            hipError_t e = hipMemcpyAsync(key.data(), d_key, nt, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(gfirst.data(), d_gf, nt, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(mse.data(), d_mse, sizeof(double) * nt, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "DWP result download: %s", hipGetErrorString(e));
        }
    }
    for (int i = 0; i < nt; ++i)
        if (gfirst[i]) qskip[i] = 1;
    if (rc != TZ_OK) {  // nothing of a failed rollout may still read the caller's frames when we return
        (void)hipStreamSynchronize(ctx->copy_stream);
        ctx->pending_src = nullptr;
    }
    if (rc == TZ_OK) {
        ctx->key_mask = key;
        ctx->group_first = gfirst;
        ctx->quant_skip = qskip;
        ctx->have_rollout = true;
        ctx->pred_contract = tz_get_contract(ctx);
        ctx->rollout_is_decode = false;
        if (key_mask) memcpy(key_mask, key.data(), nt);
        if (mse_log) memcpy(mse_log, mse.data(), sizeof(double) * nt);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "rollout failed: %s", hipGetErrorString(e));
    }
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_rollout_decode(tz_ctx* ctx, const uint8_t* key_frames, int nt, int H, int W, int warm_up,
                                 uint8_t* key_mask) {
    tz_roctx_range roctx_("tz_rollout_decode");
    if (!ctx) return TZ_ERR_INVALID;
    ctx->enc_pending = false;
    int rc = rollout_setup(ctx, key_frames, nt, H, W, warm_up);
    if (rc != TZ_OK) return rc;
    // decompress.py:123-129: a frame is a key frame iff it has a non-zero sample
    std::vector<int> flags(nt, 0);
    void* d_flags;
    rc = tz_pool_alloc(ctx, sizeof(int) * nt, &d_flags);
    if (rc == TZ_OK) {
        size_t fb = (size_t)H * W * 3;
        hipError_t e = hipMemsetAsync(d_flags, 0, sizeof(int) * nt, ctx->stream);
        int gx = (int)std::min<size_t>((fb + 255) / 256, 64);
        hipLaunchKernelGGL(k_any_nonzero, dim3(gx, nt), dim3(256), 0, ctx->stream, ctx->d_frames, fb, (int*)d_flags);
        if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), d_flags, sizeof(int) * nt, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "key discovery failed: %s", hipGetErrorString(e));
    }
    if (rc != TZ_OK) {
        tz_pool_release_all(ctx);
        return rc;
    }
    std::vector<int> kfc;
    for (int i = 0; i < nt; ++i)
        if (flags[i]) kfc.push_back(i);
    kfc.push_back(nt);
    // decompress.py:138-179: warm_up copies of C0, then for every key interval: the key frame
    // itself, one prediction from the key frame, then recursion on the previous prediction.
    std::vector<uint8_t> recon_key(nt, 0);
    std::vector<int> c0_slots;
    std::vector<PredItem> items;
    int produced = warm_up;
    for (int i = 0; i < warm_up && i < nt; ++i) c0_slots.push_back(i);
    for (int k = warm_up; k + 1 < (int)kfc.size(); ++k)
        for (int pi = kfc[k]; pi < kfc[k + 1]; ++pi, ++produced) {
            if (produced != pi || pi >= nt) {
                tz_pool_release_all(ctx);
                return tz_fail(ctx, TZ_ERR_INVALID, "key frames do not cover the sequence (frame %d)", pi);
            }
            if (pi == kfc[k]) {
                recon_key[pi] = 1;
                c0_slots.push_back(pi);  // slot content is never used for reconstruction
            } else {
                items.push_back(PredItem{pi, pi == kfc[k] + 1 ? 1 : 0, pi - 1, pi - kfc[k]});
            }
        }
    if (produced != nt) {
        tz_pool_release_all(ctx);
        return tz_fail(ctx, TZ_ERR_INVALID, "key frames do not cover the sequence (%d of %d frames)", produced, nt);
    }
    recon_key[0] = 1;  // decompress.py:186
    rc = fill_c0(ctx, c0_slots);
    if (rc == TZ_OK) rc = run_schedule(ctx, items);
    if (rc == TZ_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "decode rollout failed: %s", hipGetErrorString(e));
    }
    if (rc == TZ_OK) {
        ctx->key_mask = recon_key;
        ctx->have_rollout = true;
        ctx->pred_contract = tz_get_contract(ctx);
        ctx->rollout_is_decode = true;
        if (key_mask)
            for (int i = 0; i < nt; ++i) key_mask[i] = flags[i] ? 1 : 0;
    }
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_get_predictions(tz_ctx* ctx, float* out) {
    if (!ctx || !out) return TZ_ERR_INVALID;
    if (!ctx->have_rollout) return tz_fail(ctx, TZ_ERR_STATE, "no rollout in this context");
    size_t bytes = (size_t)ctx->nt * ctx->Hp * ctx->Wp * 3 * 4;
    TZ_HIP(ctx, hipMemcpyAsync(out, ctx->d_pred, bytes, hipMemcpyDefault, ctx->stream));
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TZ_OK;
}

// --------------------------------------------------------------------- table / LUT (host)
extern "C" int tz_build_table(const unsigned long long* hist, int nbins, int16_t* table, int* table_len) {
    if (!hist || !table || !table_len || nbins < 1 || nbins > 32767) return TZ_ERR_INVALID;
    // compress.py:352-361: symbols with count > 0, count descending; Python's stable sort with
    // reverse=True keeps ascending symbol order among equal counts.
    std::vector<int> syms;
    for (int s = 0; s < nbins; ++s)
        if (hist[s]) syms.push_back(s);
    std::stable_sort(syms.begin(), syms.end(), [&](int a, int b) { return hist[a] > hist[b]; });
    if ((int)syms.size() > TZ_MAX_TABLE) return TZ_ERR_INVALID;
    for (size_t i = 0; i < syms.size(); ++i) table[i] = (int16_t)syms[i];
    *table_len = (int)syms.size();
    return TZ_OK;
}

static int build_enc_lut(tz_ctx* ctx, const int16_t* table, int T, std::vector<int16_t>* lut) {
    lut->resize(TZ_NBINS + 1);
    for (int v = 0; v <= TZ_NBINS; ++v) (*lut)[v] = (int16_t)v;
    for (int idx = 0; idx < T; ++idx) {
        int s = table[idx];
        // compress.py:87-88 applied to arbitrary tables would chain substitutions; the
        // encoder only ever sees its own table (symbols >= 1090 > any rank), so reject others.
        if (s < TZ_MAX_TABLE || s > TZ_NBINS) return tz_fail(ctx, TZ_ERR_INVALID, "table symbol %d outside [%d, %d]", s, TZ_MAX_TABLE, TZ_NBINS);
        (*lut)[s] = (int16_t)idx;
    }
    return TZ_OK;
}

// decompress.py:31-36 sequential-pass semantics (incl. chained substitutions) + optional 1600-x
static void build_dec_lut(const int16_t* table, int T, int apply_offset, std::vector<int16_t>* lut) {
    lut->resize(TZ_NBINS + 1);
    for (int v = 0; v <= TZ_NBINS; ++v) {
        int cur = v, last = -1;
        while (cur >= 0 && cur < T && cur > last) {
            last = cur;
            cur = table[cur];
        }
        (*lut)[v] = (int16_t)(apply_offset ? TZ_OFFSET - cur : cur);
    }
}

// ------------------------------------------------------------------------ encode / decode
// Last stage of tz_encode: rank remap (compress.py:369).  A host payload leaves chunk by chunk on
// the copy stream behind the remap kernel, so that the device -> host transfer overlaps it.
static int remap_out(tz_ctx* ctx, const int16_t* d_sd, size_t N, const int16_t* lut, tz_out* o, bool defer = false) {
    constexpr int kChunks = 8;
    if (!o->host || (N < ((size_t)1 << 22) && !defer)) return tzk_lut(ctx, d_sd, N, lut, 0, (int16_t*)o->dev);
    const size_t per = ((N + kChunks - 1) / kChunks + 7) & ~(size_t)7;
    while (ctx->chunk_ev.size() < (size_t)kChunks) {
        hipEvent_t e;
        TZ_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->chunk_ev.push_back(e);
    }
    int k = 0;
    for (size_t off = 0; off < N; off += per, ++k) {
        const size_t n = std::min(per, N - off);
        TZ_TRY(tzk_lut(ctx, d_sd + off, n, lut, 0, (int16_t*)o->dev + off));
        TZ_HIP(ctx, hipEventRecord(ctx->chunk_ev[k], ctx->stream));
    }
    k = 0;
    for (size_t off = 0; off < N; off += per, ++k) {
        const size_t n = std::min(per, N - off);
        TZ_HIP(ctx, hipStreamWaitEvent(ctx->down_stream, ctx->chunk_ev[k], 0));
        TZ_TRY(tz_d2h(ctx, (int16_t*)o->host + off, (const int16_t*)o->dev + off, n * 2, ctx->down_stream));
    }
    if (defer) {   // the caller collects the payload with tz_payload_wait: the transfer runs under whatever comes next
        if (!ctx->ev_payload) TZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_payload, hipEventDisableTiming));
        TZ_HIP(ctx, hipEventRecord(ctx->ev_payload, ctx->down_stream));
        ctx->payload_inflight = true;
    } else {
        TZ_HIP(ctx, hipStreamSynchronize(ctx->down_stream));
    }
    o->done = true;
    return TZ_OK;
}

// compress.py:292-355 on the context-resident rollout: delta, quantiser, spatial delta over the whole
// flattened stack (no carry), 1600 offset + bincount when `entropy`.  d_sym receives the symbols (or
// the raw spatial delta), d_hist the counters (zeroed here), d_edge[0..1] the first and the last
// element of the quantised delta stack (what a shard boundary needs).  d_delta_tap
// (may be NULL): the quantised delta stack is also wanted there.
static int encode_front(tz_ctx* ctx, int mode, double b0, double b1, int entropy, int16_t* d_delta_tap, int16_t* d_sym,
                        unsigned long long* d_hist, int16_t* d_edge) {
    const int nt = ctx->nt, H = ctx->H, W = ctx->W;
    const size_t N = (size_t)nt * H * W * 3;
    void *d_mask = nullptr, *d_delta = d_delta_tap;
    TZ_TRY(tz_pool_alloc(ctx, nt, &d_mask));
    TZ_TRY(tz_upload(ctx, d_mask, ctx->group_first.data(), nt));
    if (entropy) TZ_HIP(ctx, hipMemsetAsync(d_hist, 0, TZ_NBINS * sizeof(unsigned long long), ctx->stream));
    // error_bound returns its input untouched in these cases (compress.py:24,35) ...
    bool lossless = b0 == 0.0 || (mode == TZ_MODE_ABSREL && b1 == 0.0);
    // ... and COMPUTES its input back wherever the worst-case tolerance of the job cannot merge two different deltas
    // (E <= 0.499: tz_quant_is_identity, with the proof).  The fused pass below then serves such a job as well; a caller
    // that taps the delta stack still gets it through the general quantiser (the parity tests compare the two).
    if (!d_delta_tap && tz_quant_is_identity(mode, b0, b1)) lossless = true;
    bool fused = false;
    // lossless and nobody asked for the delta stack: one fused pass (compress.py:292-355)
    if (lossless && !d_delta_tap)
        TZ_TRY(tzk_delta_sd_fused(ctx, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, nt, H, W, ctx->Hp, ctx->Wp,
                                  entropy ? 1 : 0, d_sym, entropy ? d_hist : nullptr, d_edge, &fused));
    // lossy and nobody asked for the delta stack: quantiser on pred / orig, fill fused with the spatial delta
    if (!lossless && !d_delta_tap)
        TZ_TRY(tzk_quant_sd_fused(ctx, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, ctx->quant_skip.data(), nt, H, W,
                                  ctx->Hp, ctx->Wp, mode, b0, b1, entropy ? 1 : 0, d_sym, entropy ? d_hist : nullptr, d_edge,
                                  &fused));
    if (fused) return TZ_OK;
    if (!d_delta) TZ_TRY(tz_pool_alloc(ctx, N * 2, &d_delta));
    // compress.py:292-314
    TZ_TRY(tzk_delta(ctx, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, nt, H, W, ctx->Hp, ctx->Wp, (int16_t*)d_delta));
    // compress.py:315-319
    TZ_TRY(tzk_error_bound(ctx, ctx->d_frames, (int16_t*)d_delta, ctx->quant_skip.data(), nt, H, W, mode, b0, b1));
    // compress.py:339-355
    TZ_TRY(tzk_spatial_delta(ctx, (const int16_t*)d_delta, N, 0, 0, entropy ? 1 : 0, d_sym, entropy ? d_hist : nullptr));
    TZ_HIP(ctx, hipMemcpyAsync(d_edge, d_delta, 2, hipMemcpyDeviceToDevice, ctx->stream));
    TZ_HIP(ctx, hipMemcpyAsync(d_edge + 1, (const int16_t*)d_delta + (N - 1), 2, hipMemcpyDeviceToDevice, ctx->stream));
    return TZ_OK;
}

extern "C" int tz_encode(tz_ctx* ctx, int mode, double b0, double b1, int entropy, int16_t* payload, int16_t* table,
                         int* table_len, int16_t* delta_out) {
    tz_roctx_range roctx_("tz_encode");
    if (!ctx || !table_len || ((entropy & 1) && !table)) return TZ_ERR_INVALID;
    if (!ctx->have_rollout || ctx->rollout_is_decode) return tz_fail(ctx, TZ_ERR_STATE, "tz_encode needs a tz_rollout first");
    TZ_TRY(tz_check_pred_contract(ctx, "tz_encode"));
    ctx->enc_pending = false;
    if (!payload) {  // keep the payload in the context: it leaves through tz_payload_get
        const size_t n = (size_t)ctx->nt * ctx->H * ctx->W * 3;
        TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_payload, &ctx->cap_payload, n * 2));
        ctx->payload_len = n;
        payload = ctx->d_payload;
    }
    if (mode < 0 || mode > 3) return tz_fail(ctx, TZ_ERR_INVALID, "unknown error-bound mode %d", mode);
    const size_t N = (size_t)ctx->nt * ctx->H * ctx->W * 3;
    const bool shuffle = (entropy & 2) != 0;  // opt-in byte planes (not a reference format)
    entropy &= 1;
    if (shuffle && (N & 7)) return tz_fail(ctx, TZ_ERR_INVALID, "byte shuffle needs a multiple of 8 elements");
    std::vector<tz_out> outs;
    tz_out o_pay, o_delta, o_final;
    void *d_hist = nullptr, *d_sd = nullptr, *d_edge = nullptr;
    // a transfer of the call before may still be reading the staging buffer (and writing the caller's previous host
    // buffer): it has ~a rollout's time to finish, and must have before this call's remap writes the buffer again
    int rc = TZ_OK;
    const bool defer = ctx->defer_payload && entropy && !shuffle && tz_ptr_kind(payload) == 1;
    if (ctx->payload_inflight) {
        if (defer) {
            hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev_payload, 0);
            if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e));
        } else {
            rc = tz_payload_settle(ctx);
        }
    }
    if (rc == TZ_OK && defer) {
        if (ctx->cap_payload_stage < N * 2) rc = tz_payload_settle(ctx);   // (growing frees the old buffer)
        if (rc == TZ_OK) rc = tz_ensure(ctx, (void**)&ctx->d_payload_stage, &ctx->cap_payload_stage, N * 2);
        o_pay.bytes = N * 2;
        o_pay.host = payload;
        o_pay.dev = ctx->d_payload_stage;
    } else if (rc == TZ_OK) {
        rc = tz_dev_out(ctx, payload, N * 2, &o_pay);
    }
    if (rc == TZ_OK && shuffle) {  // the stages below write the plain payload to a scratch buffer instead
        o_final = o_pay;
        o_pay = tz_out();
        o_pay.bytes = N * 2;
        rc = tz_pool_alloc(ctx, N * 2, &o_pay.dev);
    }
    if (rc == TZ_OK && delta_out) {
        rc = tz_dev_out(ctx, delta_out, N * 2, &o_delta);
        if (rc == TZ_OK) outs.push_back(o_delta);
    }
    if (rc == TZ_OK) rc = tz_pool_alloc(ctx, 16, &d_edge);
    if (rc == TZ_OK && entropy) {
        rc = tz_pool_alloc(ctx, TZ_NBINS * sizeof(unsigned long long), &d_hist);
        if (rc == TZ_OK) rc = tz_pool_alloc(ctx, N * 2, &d_sd);
    }
    if (rc == TZ_OK)
        rc = encode_front(ctx, mode, b0, b1, entropy, delta_out ? (int16_t*)o_delta.dev : nullptr,
                          entropy ? (int16_t*)d_sd : (int16_t*)o_pay.dev, (unsigned long long*)d_hist, (int16_t*)d_edge);
    if (rc == TZ_OK && !entropy) {
        *table_len = -1;
    } else if (rc == TZ_OK) {
        std::vector<unsigned long long> hist(TZ_NBINS, 0);
        hipError_t e = hipMemcpyAsync(hist.data(), d_hist, TZ_NBINS * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "hist download: %s", hipGetErrorString(e));
        std::vector<int16_t> lut;
        const auto t0 = std::chrono::steady_clock::now();
        if (rc == TZ_OK) rc = tz_build_table(hist.data(), TZ_NBINS, table, table_len);  // 356-361
        if (rc == TZ_OK) rc = build_enc_lut(ctx, table, *table_len, &lut);
        if (ctx->prof_on) {
            ctx->prof[TZP_TABLE].total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ctx->prof[TZP_TABLE].launches += 1;
        }
        if (rc == TZ_OK) rc = remap_out(ctx, (const int16_t*)d_sd, N, lut.data(), &o_pay, defer);  // 369
    }
    if (rc == TZ_OK && shuffle) {
        rc = tzk_shuffle(ctx, (const int16_t*)o_pay.dev, N, (uint8_t*)o_final.dev, 0);
        o_pay = o_final;
    }
    outs.push_back(o_pay);
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

// ---- tz_encode in two phases, for jobs whose frame windows are sharded over GPUs.
// The spatial delta runs over the WHOLE flattened stack (compress.py:339) and the rank table comes
// from the GLOBAL histogram (compress.py:354-361): a shard therefore runs everything up to its own
// symbols and counters (begin), the ranks exchange one carry element and sum 2111 counters, and
// the shard finishes with the global table (finish).  Same kernels as tz_encode; the symbols stay in
// the context's resident payload buffer between the two calls and are remapped in place.
extern "C" int tz_encode_begin(tz_ctx* ctx, int mode, double b0, double b1, int entropy, unsigned long long* hist,
                               int16_t* edge) {
    tz_roctx_range roctx_("tz_encode_begin");
    if (!ctx || !edge || (entropy && !hist)) return TZ_ERR_INVALID;
    if (!ctx->have_rollout || ctx->rollout_is_decode) return tz_fail(ctx, TZ_ERR_STATE, "tz_encode_begin needs a tz_rollout first");
    TZ_TRY(tz_check_pred_contract(ctx, "tz_encode_begin"));
    if (mode < 0 || mode > 3) return tz_fail(ctx, TZ_ERR_INVALID, "unknown error-bound mode %d", mode);
    ctx->enc_pending = false;
    const size_t N = (size_t)ctx->nt * ctx->H * ctx->W * 3;
    TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_payload, &ctx->cap_payload, N * 2));
    ctx->payload_len = N;
    void *d_hist = nullptr, *d_edge = nullptr;
    int rc = tz_pool_alloc(ctx, 16, &d_edge);
    if (rc == TZ_OK && entropy) rc = tz_pool_alloc(ctx, TZ_NBINS * sizeof(unsigned long long), &d_hist);
    if (rc == TZ_OK)
        rc = encode_front(ctx, mode, b0, b1, entropy ? 1 : 0, nullptr, ctx->d_payload, (unsigned long long*)d_hist, (int16_t*)d_edge);
    if (rc == TZ_OK) {
        hipError_t e = hipMemcpyAsync(edge, d_edge, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && entropy)
            e = hipMemcpyAsync(hist, d_hist, TZ_NBINS * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "tz_encode_begin: %s", hipGetErrorString(e));
    }
    if (rc == TZ_OK) {
        ctx->enc_pending = true;
        ctx->enc_entropy = entropy != 0;
        ctx->enc_first = edge[0];
    }
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_encode_finish(tz_ctx* ctx, int has_carry, int16_t carry, const int16_t* table, int table_len,
                                int16_t* payload) {
    tz_roctx_range roctx_("tz_encode_finish");
    if (!ctx) return TZ_ERR_INVALID;
    if (!ctx->enc_pending) return tz_fail(ctx, TZ_ERR_STATE, "tz_encode_finish needs a tz_encode_begin first");
    if (ctx->enc_entropy != (table_len >= 0) || (table_len > 0 && !table) || table_len > TZ_MAX_TABLE)
        return tz_fail(ctx, TZ_ERR_INVALID, "tz_encode_finish: table does not match the entropy flag of tz_encode_begin");
    const size_t N = ctx->payload_len;
    if (!ctx->have_rollout || ctx->rollout_is_decode || N != (size_t)ctx->nt * ctx->H * ctx->W * 3) {
        ctx->enc_pending = false;
        return tz_fail(ctx, TZ_ERR_STATE, "tz_encode_finish: the resident symbols (%zu) are not those of the current rollout", N);
    }
    int rc = TZ_OK;
    if (has_carry) {
        // the first element of the shard: sd = carry - x[0] instead of x[0] (compress.py:73-77 across the boundary)
        const int16_t sd = (int16_t)(carry - ctx->enc_first);
        const int16_t y = ctx->enc_entropy ? (int16_t)(TZ_OFFSET - sd) : sd;
        rc = tz_upload(ctx, ctx->d_payload, &y, 2);
    }
    std::vector<tz_out> outs;
    tz_out o;
    if (rc == TZ_OK && payload) rc = tz_dev_out(ctx, payload, N * 2, &o);
    if (rc == TZ_OK && ctx->enc_entropy) {
        std::vector<int16_t> lut;
        rc = build_enc_lut(ctx, table, table_len, &lut);
        if (rc == TZ_OK && payload) rc = remap_out(ctx, ctx->d_payload, N, lut.data(), &o);                 // compress.py:369
        else if (rc == TZ_OK) rc = tzk_lut(ctx, ctx->d_payload, N, lut.data(), 0, ctx->d_payload);           // in place: stays resident
    } else if (rc == TZ_OK && payload) {
        hipError_t e = hipMemcpyAsync(o.dev, ctx->d_payload, N * 2, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "payload copy: %s", hipGetErrorString(e));
    }
    if (payload) outs.push_back(o);
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    if (rc == TZ_OK) ctx->enc_pending = false;
    tz_pool_release_all(ctx);
    return rc;
}

// Opt-in byte shuffle of an int16 stream and its inverse (stand-alone; tz_encode applies the
// forward direction itself when bit 1 of `entropy` is set).
extern "C" int tz_byte_shuffle(tz_ctx* ctx, const int16_t* in, size_t n, uint8_t* out) {
    if (!ctx || !in || !out) return TZ_ERR_INVALID;
    const void* din;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, in, n * 2, &din);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, n * 2, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_shuffle(ctx, (const int16_t*)din, n, (uint8_t*)o.dev, 0);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_byte_unshuffle(tz_ctx* ctx, const uint8_t* in, size_t n, int16_t* out) {
    if (!ctx || !in || !out) return TZ_ERR_INVALID;
    const void* din;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, in, n * 2, &din);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, n * 2, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_shuffle(ctx, (const int16_t*)din, n, (uint8_t*)o.dev, 1);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_encode_delta(tz_ctx* ctx, int mode, double b0, double b1, int16_t* delta_out) {
    if (!ctx || !delta_out) return TZ_ERR_INVALID;
    if (!ctx->have_rollout || ctx->rollout_is_decode) return tz_fail(ctx, TZ_ERR_STATE, "tz_encode_delta needs a tz_rollout first");
    TZ_TRY(tz_check_pred_contract(ctx, "tz_encode_delta"));
    const int nt = ctx->nt, H = ctx->H, W = ctx->W;
    const size_t N = (size_t)nt * H * W * 3;
    std::vector<tz_out> outs;
    tz_out o;
    void* d_mask = nullptr;
    int rc = tz_dev_out(ctx, delta_out, N * 2, &o);
    if (rc == TZ_OK) outs.push_back(o);
    if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nt, &d_mask);
    if (rc == TZ_OK) rc = tz_upload(ctx, d_mask, ctx->group_first.data(), nt);
    if (rc == TZ_OK) rc = tzk_delta(ctx, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, nt, H, W, ctx->Hp, ctx->Wp, (int16_t*)o.dev);
    if (rc == TZ_OK) rc = tzk_error_bound(ctx, ctx->d_frames, (int16_t*)o.dev, ctx->quant_skip.data(), nt, H, W, mode, b0, b1);
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_decode_delta(tz_ctx* ctx, const int16_t* delta, uint8_t* frames_out) {
    if (!ctx || !delta || !frames_out) return TZ_ERR_INVALID;
    if (!ctx->have_rollout || !ctx->rollout_is_decode) return tz_fail(ctx, TZ_ERR_STATE, "tz_decode_delta needs a tz_rollout_decode first");
    TZ_TRY(tz_check_pred_contract(ctx, "tz_decode_delta"));
    const int nt = ctx->nt, H = ctx->H, W = ctx->W;
    const size_t N = (size_t)nt * H * W * 3;
    std::vector<tz_out> outs;
    tz_out o;
    const void* d_diff = nullptr;
    void* d_mask = nullptr;
    int rc = tz_dev_in(ctx, delta, N * 2, &d_diff);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, frames_out, N, &o);
    if (rc == TZ_OK) outs.push_back(o);
    if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nt, &d_mask);
    if (rc == TZ_OK) rc = tz_upload(ctx, d_mask, ctx->key_mask.data(), nt);
    if (rc == TZ_OK)
        rc = tzk_reconstruct(ctx, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, (const int16_t*)d_diff, nt, H, W,
                             ctx->Hp, ctx->Wp, (uint8_t*)o.dev);
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_decode(tz_ctx* ctx, const int16_t* payload, size_t payload_len, const int16_t* table, int table_len,
                         uint8_t* frames_out) {
    tz_roctx_range roctx_("tz_decode");
    if (!ctx) return TZ_ERR_INVALID;
    if (!ctx->have_rollout || !ctx->rollout_is_decode) return tz_fail(ctx, TZ_ERR_STATE, "tz_decode needs a tz_rollout_decode first");
    TZ_TRY(tz_check_pred_contract(ctx, "tz_decode"));
    if (table_len > TZ_NBINS || (table_len >= 0 && !table && table_len > 0)) return tz_fail(ctx, TZ_ERR_INVALID, "bad table");
    const int nt = ctx->nt, H = ctx->H, W = ctx->W;
    const size_t N = (size_t)nt * H * W * 3;
    if (!payload) {  // staged with tz_payload_begin / tz_payload_put (on the copy stream)
        if (!ctx->d_payload || ctx->payload_len < N) return tz_fail(ctx, TZ_ERR_STATE, "no staged payload of %zu elements", N);
        TZ_HIP(ctx, hipEventRecord(ctx->ev_frames, ctx->copy_stream));
        TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_frames, 0));
        payload = ctx->d_payload;
    }
    ctx->have_decoded = false;
    if (payload_len != N)  // decompress.py:240: the reshape raises
        return tz_fail(ctx, TZ_ERR_INVALID, "payload holds %zu elements, the key-frame stack implies %zu", payload_len, N);
    const bool resident = frames_out == nullptr;
    if (resident) {  // keep the frames in the context: tz_decoded_get
        TZ_TRY(tz_ensure(ctx, (void**)&ctx->d_out, &ctx->cap_out, N));
        frames_out = ctx->d_out;
    }
    std::vector<tz_out> outs;
    tz_out o;
    const void* d_pay = nullptr;
    void *d_diff = nullptr, *d_mask = nullptr;
    int rc = tz_dev_in(ctx, payload, N * 2, &d_pay);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, frames_out, N, &o);
    if (rc == TZ_OK) outs.push_back(o);
    if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nt, &d_mask);
    if (rc == TZ_OK) rc = tz_upload(ctx, d_mask, ctx->key_mask.data(), nt);
    std::vector<int16_t> lut;
    if (table_len >= 0) build_dec_lut(table, table_len, 1, &lut);  // decompress.py:203-236 rides on the scan of 240-245
    const int16_t* h_lut = table_len >= 0 ? lut.data() : nullptr;
    bool fused = false;
    if (rc == TZ_OK && !ctx->decode_unfused)   // one launch: inverse remap + inverse spatial delta + reconstruct
        rc = tzk_decode_tail_fused(ctx, (const int16_t*)d_pay, h_lut, 1, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, nt,
                                   H, W, ctx->Hp, ctx->Wp, (uint8_t*)o.dev, &fused);
    if (rc == TZ_OK && !fused) {
        rc = tz_pool_alloc(ctx, N * 2, &d_diff);
        if (rc == TZ_OK && h_lut) rc = tzk_unmap_undelta(ctx, (const int16_t*)d_pay, N, h_lut, 1, (int16_t*)d_diff);
        else if (rc == TZ_OK) rc = tzk_undelta(ctx, (const int16_t*)d_pay, N, 0, 0, (int16_t*)d_diff);  // decompress.py:240-245
        if (rc == TZ_OK)                                                    // decompress.py:252-256,269
            rc = tzk_reconstruct(ctx, ctx->d_pred, ctx->d_frames, (const uint8_t*)d_mask, (const int16_t*)d_diff, nt, H, W,
                                 ctx->Hp, ctx->Wp, (uint8_t*)o.dev);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    if (rc == TZ_OK && resident) ctx->have_decoded = true;   // only a decode whose work is queued leaves frames to fetch
    tz_pool_release_all(ctx);
    return rc;
}

// ------------------------------------------------------------------ stand-alone operators
extern "C" int tz_delta_encode(tz_ctx* ctx, const float* pred, const uint8_t* orig, const uint8_t* zero_mask, int nframes,
                               int H, int W, int16_t* out) {
    if (!ctx || !pred || !orig || !out || nframes < 0 || H < 1 || W < 1) return TZ_ERR_INVALID;
    int Hp = pad8(H), Wp = pad8(W);
    size_t N = (size_t)nframes * H * W * 3;
    std::vector<uint8_t> zm(nframes, 0);
    if (zero_mask) memcpy(zm.data(), zero_mask, nframes);
    const void *dp, *dor;
    void* dm;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, pred, (size_t)nframes * Hp * Wp * 3 * 4, &dp);
    if (rc == TZ_OK) rc = tz_dev_in(ctx, orig, N, &dor);
    if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nframes, &dm);
    if (rc == TZ_OK && nframes) rc = tz_upload(ctx, dm, zm.data(), nframes);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, N * 2, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_delta(ctx, (const float*)dp, (const uint8_t*)dor, (const uint8_t*)dm, nframes, H, W, Hp, Wp, (int16_t*)o.dev);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_error_bound(tz_ctx* ctx, const uint8_t* orig, int16_t* diff, const uint8_t* skip_mask, int nframes, int H,
                              int W, int mode, double b0, double b1) {
    if (!ctx || !orig || !diff || nframes < 0 || H < 1 || W < 1) return TZ_ERR_INVALID;
    size_t N = (size_t)nframes * H * W * 3;
    std::vector<uint8_t> sk(nframes, 0);
    if (skip_mask) memcpy(sk.data(), skip_mask, nframes);
    const void* dor;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, orig, N, &dor);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, diff, N * 2, &o);
    if (rc == TZ_OK && o.host) {
        hipError_t e = hipMemcpyAsync(o.dev, diff, N * 2, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "diff upload: %s", hipGetErrorString(e));
    }
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_error_bound(ctx, (const uint8_t*)dor, (int16_t*)o.dev, sk.data(), nframes, H, W, mode, b0, b1);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_spatial_delta(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry, int apply_offset,
                                int16_t* out, unsigned long long* hist) {
    if (!ctx || !in || !out) return TZ_ERR_INVALID;
    const void* din;
    tz_out o, oh;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, in, n * 2, &din);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, n * 2, &o);
    if (rc == TZ_OK) outs.push_back(o);
    void* dh = nullptr;
    if (rc == TZ_OK && hist) {
        rc = tz_dev_out(ctx, hist, TZ_NBINS * sizeof(unsigned long long), &oh);
        if (rc == TZ_OK) {
            dh = oh.dev;
            if (oh.host) {  // counts are ADDED to what the caller holds
                hipError_t e = hipMemcpyAsync(dh, hist, TZ_NBINS * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream);
                if (e != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "hist upload: %s", hipGetErrorString(e));
            }
            outs.push_back(oh);
        }
    }
    if (rc == TZ_OK) rc = tzk_spatial_delta(ctx, (const int16_t*)din, n, has_carry, carry, apply_offset, (int16_t*)o.dev, (unsigned long long*)dh);
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

static int lut_op(tz_ctx* ctx, const int16_t* in, size_t n, const std::vector<int16_t>& lut, int post, int16_t* out) {
    const void* din;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, in, n * 2, &din);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, n * 2, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_lut(ctx, (const int16_t*)din, n, lut.data(), post, (int16_t*)o.dev);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_remap(tz_ctx* ctx, const int16_t* in, size_t n, const int16_t* table, int table_len, int16_t* out) {
    if (!ctx || !in || !out || !table || table_len < 0 || table_len > TZ_MAX_TABLE) return TZ_ERR_INVALID;
    std::vector<int16_t> lut;
    TZ_TRY(build_enc_lut(ctx, table, table_len, &lut));
    return lut_op(ctx, in, n, lut, 0, out);
}

extern "C" int tz_unmap(tz_ctx* ctx, const int16_t* in, size_t n, const int16_t* table, int table_len, int apply_offset,
                        int16_t* out) {
    if (!ctx || !in || !out || !table || table_len < 0 || table_len > TZ_NBINS) return TZ_ERR_INVALID;
    std::vector<int16_t> lut;
    build_dec_lut(table, table_len, apply_offset, &lut);
    return lut_op(ctx, in, n, lut, apply_offset, out);
}

extern "C" int tz_spatial_undelta(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry, int16_t* out) {
    if (!ctx || !in || !out) return TZ_ERR_INVALID;
    const void* din;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, in, n * 2, &din);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, n * 2, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_undelta(ctx, (const int16_t*)din, n, has_carry, carry, (int16_t*)o.dev);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_reconstruct(tz_ctx* ctx, const float* pred, const uint8_t* key_frames, const uint8_t* key_mask,
                              const int16_t* diff, int nframes, int H, int W, uint8_t* out) {
    if (!ctx || !pred || !diff || !out || nframes < 0 || H < 1 || W < 1) return TZ_ERR_INVALID;
    int Hp = pad8(H), Wp = pad8(W);
    size_t N = (size_t)nframes * H * W * 3;
    std::vector<uint8_t> km(nframes, 0);
    if (key_mask && key_frames) memcpy(km.data(), key_mask, nframes);
    const void *dp, *dk = nullptr, *dd;
    void* dm;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, pred, (size_t)nframes * Hp * Wp * 3 * 4, &dp);
    if (rc == TZ_OK && key_frames) rc = tz_dev_in(ctx, key_frames, N, &dk);
    if (rc == TZ_OK) rc = tz_dev_in(ctx, diff, N * 2, &dd);
    if (rc == TZ_OK) rc = tz_pool_alloc(ctx, nframes, &dm);
    if (rc == TZ_OK && nframes) rc = tz_upload(ctx, dm, km.data(), nframes);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, N, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        rc = tzk_reconstruct(ctx, (const float*)dp, (const uint8_t*)dk, (const uint8_t*)dm, (const int16_t*)dd, nframes, H,
                             W, Hp, Wp, (uint8_t*)o.dev);
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_window_sse(tz_ctx* ctx, const uint8_t* orig, const float* pred, int nframes, int H, int W, double* sse) {
    if (!ctx || !orig || !pred || !sse || nframes < 0 || H < 1 || W < 1) return TZ_ERR_INVALID;
    int Hp = pad8(H), Wp = pad8(W);
    const void *dor, *dp;
    int rc = tz_dev_in(ctx, orig, (size_t)nframes * H * W * 3, &dor);
    if (rc == TZ_OK) rc = tz_dev_in(ctx, pred, (size_t)nframes * Hp * Wp * 3 * 4, &dp);
    if (rc == TZ_OK) rc = tzk_sse(ctx, (const uint8_t*)dor, (const float*)dp, nframes, H, W, Hp, Wp, sse);
    tz_pool_release_all(ctx);
    return rc;
}
