// Internal declarations shared by the translation units of libtezip_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/tezip_hip.h"

#define TZ_SENTINEL ((int16_t)0x7FFF)

enum tz_prof_class {
    TZP_CONV = 0,   // implicit-GEMM MFMA convolutions (all PredNet convs)
    TZP_ERR0,       // level-0 error unit
    TZP_DELTA,      // pred/orig -> int16 delta
    TZP_QUANT,      // error-bound quantiser kernels
    TZP_SDELTA,     // spatial delta (+offset, histogram)
    TZP_LUT,        // rank remap / unmap
    TZP_SCAN,       // inverse spatial delta (prefix scan)
    TZP_RECON,      // reconstruct
    TZP_SSE,        // window MSE partial sums
    // sub-classes of TZP_CONV (a launch is counted in TZP_CONV and in exactly one of these)
    TZP_CONV16,     // k_conv16: LDS-DMA kernel, sources with a multiple of 16 channels (levels >= 1)
    TZP_CONV16B,    // k_conv16b: level-0 block-step kernel
    TZP_CONV_SMALL, // k_conv_small: direct VALU 3 -> 3 convolution
    TZP_CONV_GEN,   // k_conv3x3: general kernel
    TZP_CONVLAT,    // k_convlat: one accumulator tile per wave, for grids that cannot fill the chip
    TZP_WINO,       // k_wino: TZ-PA2 form (Winograd F(2x2, 3x3) on the same-resolution source) of the k_conv16 convolutions
    TZP_TABLE,      // HOST time: rank table + LUT from the downloaded histogram (compress.py:356-361)
    TZP_QSERIAL,    // not a time: `launches` counts the chains the quantiser sent through its serial fallback (k_q_serial)
    TZP_COUNT
};

struct tz_model;      // tz_prednet.hip

struct tz_prof_slot {
    double total_ms = 0;
    long long launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<int> pending_sub;  // parallel to `pending`: second class of the interval or -1
};

struct tz_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    // device scratch pool (staging of host-pointer arguments, temporaries): grown on demand and
    // reused; the top bit of the size marks a block that is handed out until tz_pool_release_all
    std::vector<std::pair<void*, size_t>> pool;
    // model + rollout state
    tz_model* model = nullptr;
    int conv_impl = 1;                // tz_set_conv_impl: 1 = LDS-DMA kernels where they apply
    int lat_mode = 1;                 // k_convlat: 0 never, 1 where the cost model says so, 2 wherever eligible (TEZIP_LAT)
    int contract = 0;                 // arithmetic contract of the predictor: 0 = by frame size, 1 = TZ-PA1, 2 = TZ-PA2 (tz_set_contract, TEZIP_PA)
    int num_cus = 256;                // compute units of the device (k_wino: column blocks per workgroup)
    int wino_ipw = 0;                 // TEZIP_WINO_IPW (measurements): column blocks per k_wino workgroup, 0 = chosen per launch
    // rollout-resident data
    int nt = 0, H = 0, W = 0, Hp = 0, Wp = 0, warm_up = 0;
    uint8_t* d_frames = nullptr;      // nt*H*W*3 (encoder: originals; decoder: key stack)
    float* d_pred = nullptr;          // nt*Hp*Wp*3
    size_t cap_frames = 0, cap_pred = 0;
    std::vector<uint8_t> key_mask;    // nt
    std::vector<uint8_t> group_first; // nt: 1 where a group starts (delta slot 0 -> 0)
    std::vector<uint8_t> quant_skip;  // nt: 1 where error_bound is not applied
    bool have_rollout = false, rollout_is_decode = false;
    int pred_contract = 0;            // the arithmetic contract that produced the resident prediction stack (stamped by
                                      // tz_rollout / tz_rollout_decode; tz_encode* / tz_decode* refuse a flip in between)
    // pinned staging ring for small host->device uploads (index arrays, masks, LUTs): the copy
    // out of it is truly asynchronous and the memory outlives the caller's locals
    uint8_t* ring = nullptr;
    size_t ring_size = 0, ring_pos = 0;
    // index table of the static rollout schedule (SWP encoder / decoder replay): [batches][3][stride]
    int* d_sched = nullptr;
    size_t cap_sched = 0;
    // copy engine: host<->device transfers of the frame stack and the payload run on their own
    // stream so that they overlap the predictor (key frames go first, the rest follows while the
    // rollout computes; the payload leaves chunk by chunk behind the remap kernel).  Pinned host
    // memory (tz_host_alloc) is DMA'd directly; pageable memory is pipelined through `stage`.
    hipStream_t copy_stream = nullptr;
    hipStream_t down_stream = nullptr;   // device -> host leg of the payload hand-over: a stream of its own, so that a deferred
                                         // transfer (tz_set_payload_deferred) does not sit in front of the next sequence's uploads
    hipEvent_t ev_keys = nullptr, ev_frames = nullptr, ev_compute = nullptr;
    // second compute stream of a static rollout schedule (run_schedule): the windows advance as two independent
    // groups, so that one group's launch fills the CUs the other group's draining launch leaves idle
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int split_rollout = 0;            // TEZIP_SPLIT=1 turns it on (measured in round 3: slower, see run_schedule)
    // "E-part ahead" (round 5, tz_model_predict_batch_dev): per level, E_l ready on the compute stream / the launch over it
    // finished on stream2
    hipEvent_t ev_epart_src[TZ_MAX_LEVELS] = {nullptr}, ev_epart_done[TZ_MAX_LEVELS] = {nullptr};
    hipEvent_t ev_cal[2] = {nullptr, nullptr};   // epart_measure: timing events on the compute stream (tz_prednet.hip)
    int epart_mode = -1;              // TEZIP_EPART: -1 where launches cannot fill the chip (default), 0 never, 1 wherever possible
    static constexpr int kStages = 4;
    static constexpr size_t kStageBytes = (size_t)8 << 20;
    uint8_t* stage[kStages] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t stage_ev[kStages] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_busy[kStages] = {false, false, false, false};
    int stage_next = 0;
    bool staged = false;                    // d_frames was filled by tz_frames_begin / tz_frames_put
    int16_t* d_payload = nullptr;           // resident payload of a tz_encode(payload = NULL)
    size_t cap_payload = 0, payload_len = 0;
    unsigned* d_scan_status = nullptr;      // inverse scan: one word per resident block, tagged with the launch's epoch
    unsigned scan_epoch = 0;
    unsigned scan_dbg_skew = 0, scan_dbg_limit = 0;   // tz_scan_fault_inject: poll for another epoch / give up sooner
    // fault word: one pinned, device-visible host word that a kernel with a hand-built wait sets when the wait expires
    // (k_scan2p's bounded poll); the host reads it behind every stream synchronisation (tz_stream_sync)
    volatile unsigned* h_fault = nullptr;
    unsigned* d_fault = nullptr;
    int decode_unfused = 0;                 // TEZIP_DECODE_UNFUSED=1: tz_decode as scan + reconstruct launches (cross-check)
    uint8_t* d_out = nullptr;               // resident decoded frames of a tz_decode(frames_out = NULL)
    size_t cap_out = 0;
    bool have_decoded = false;
    bool enc_pending = false, enc_entropy = false;  // tz_encode_begin done, symbols resident in d_payload
    int16_t enc_first = 0;                          // first element of the shard's quantised delta stack
    const uint8_t* pending_src = nullptr;  // host frame stack whose non-key frames are still to be sent
    std::vector<uint8_t> pending_sent;     // nt: 1 = already on its way
    std::vector<hipEvent_t> chunk_ev;  // payload chunk hand-over events (compute -> copy stream)
    // deferred payload hand-over (tz_set_payload_deferred): tz_encode returns once the last chunk of the payload is QUEUED
    // on the copy stream; the device -> host transfer then runs under the next sequence's rollout.  The chunks leave from
    // a staging buffer of the context's own (a pool block would be handed out again by the next call).
    int defer_payload = 0;
    bool payload_inflight = false;
    hipEvent_t ev_payload = nullptr;        // recorded on the copy stream behind the last chunk
    int16_t* d_payload_stage = nullptr;
    size_t cap_payload_stage = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool prof_on = false;
    tz_prof_slot prof[TZP_COUNT];
};

int tz_fail(tz_ctx* ctx, int status, const char* fmt, ...);
// TZ_ERR_STATE when the contract in force is no longer the one the resident prediction stack was made under
int tz_check_pred_contract(tz_ctx* ctx, const char* who);
static constexpr unsigned TZ_FAULT_SCAN_POLL = 1u;   // k_scan2p: a status word never showed the launch's epoch
int tz_fault_word(tz_ctx* ctx);      // makes the fault word on first use
int tz_stream_sync(tz_ctx* ctx);     // hipStreamSynchronize(ctx->stream) + TZ_ERR_HIP if a kernel reported a fault
int tz_payload_settle(tz_ctx* ctx);  // waits for a deferred payload transfer still in flight (no-op without one)

#define TZ_HIP(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return tz_fail(ctx, TZ_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                                \
    } while (0)

#define TZ_TRY(expr)            \
    do {                        \
        int _s = (expr);        \
        if (_s != TZ_OK) return _s; \
    } while (0)

// ---- window SSE: one 4096-element block of one padded frame (compress.py:246), the fixed order of tz_codec.hip k_sse:
// thread t sums elements t, t + 256, ... of the block, then a halving tree over the 256 partials; the result is valid on
// thread 0.  `s` = 256 doubles of LDS.  Shared by k_sse and by the DWP step kernel of tz_api.hip (k_sse_decide).
static __device__ __forceinline__ double tz_sse_block(const uint8_t* __restrict__ o, const float* __restrict__ p, int H, int W, int Hp,
                                                      int Wp, int b, double* s) {
    // (32-bit index arithmetic: a padded frame has fewer than 2^29 elements, tz_model_prepare; the 64-bit divisions this
    // loop used to do per element were 27 us per 512x512 frame -- 4 % of a one-window predictor step)
    const unsigned n = (unsigned)Hp * (unsigned)Wp * 3u, uWp = (unsigned)Wp;
    // all 32 loads of the thread first (they are independent; issued one iteration at a time their round trips added up to
    // most of the kernel), then the sum in its fixed order
    float pv[16], xv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const unsigned i = (unsigned)b * 4096u + (unsigned)j * 256u + threadIdx.x;
        pv[j] = 0.0f;
        xv[j] = 0.0f;
        if (i < n) {
            const unsigned pix = i / 3u, c = i - pix * 3u, y = pix / uWp, x = pix - y * uWp;
            pv[j] = p[i];
            if (y < (unsigned)H && x < (unsigned)W) xv[j] = (float)o[((size_t)y * W + x) * 3 + c] / 255.0f;
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double d = (double)xv[j] - (double)pv[j];
        acc = acc + d * d;   // (an element past the end of the frame adds +0.0, as before)
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + st];
        __syncthreads();
    }
    return s[0];
}

// ---- device memory helpers -------------------------------------------------------------
bool tz_is_device_ptr(const void* p);
int tz_poison_byte();                                 // TEZIP_POISON diagnostic: 0x100 | byte, or 0
int tz_poison(tz_ctx*, void* p, size_t bytes);        // fill a freshly handed-out device buffer when it is on
int tz_pool_alloc(tz_ctx* ctx, size_t bytes, void** out);   // freed by tz_pool_release_all
void tz_pool_release_all(tz_ctx* ctx);
int tz_ensure(tz_ctx* ctx, void** buf, size_t* cap, size_t bytes);  // persistent buffer growth
// Stream-ordered upload of a small host block to `dst` (device) through the pinned ring.
int tz_upload(tz_ctx* ctx, void* dst, const void* src, size_t bytes);

// 0 = pageable host, 1 = pinned / registered host, 2 = device
int tz_ptr_kind(const void* p);
// Stream-ordered host -> device copy on stream `s`.  Pinned source: one asynchronous DMA (the
// source must stay valid until `s` reaches it).  Pageable source: pipelined through the pinned
// staging buffers; the source is free again when the call returns.
int tz_h2d(tz_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s);
// Device -> host copy on stream `s`.  Pinned destination: asynchronous (complete when `s` is
// synchronised).  Pageable destination: pipelined through the staging buffers, complete on return.
int tz_d2h(tz_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s);
// Input argument: returns a device pointer holding `bytes` of *p (staging if host).
int tz_dev_in(tz_ctx* ctx, const void* p, size_t bytes, const void** dev);
// Output argument: returns a device pointer to write; tz_dev_out_finish copies back if host.
struct tz_out {
    void* host = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    bool done = false;  // already copied back (chunked hand-over on the copy stream)
};
int tz_dev_out(tz_ctx* ctx, void* p, size_t bytes, tz_out* o);
int tz_dev_out_finish(tz_ctx* ctx, std::vector<tz_out>& outs);  // D2H copies + stream sync if any host

// ---- profiling wrapper ------------------------------------------------------------------
struct tz_prof_scope {
    tz_ctx* ctx;
    int cls;
    int sub = -1;  // optional second class the same interval is added to (set before the scope ends)
    hipEvent_t a = nullptr, b = nullptr;
    bool rx = false;   // a ROCTx range is open (TEZIP_ROCTX=1)
    tz_prof_scope(tz_ctx* c, int k);
    ~tz_prof_scope();
};

// ROCTx ranges ("same -v lines + rocprofv3 / roctx ranges"): with TEZIP_ROCTX=1 the library opens
// librocprofiler-sdk-roctx.so (else libroctx64.so) and brackets every C-ABI stage call and every stage class of
// tz_prof_scope with roctxRangePushA / roctxRangePop, so that `rocprofv3 --marker-trace --kernel-trace` shows the launches
// of a stage under its name.  Off (the default): one branch on a cached flag.  A library that cannot be opened is said
// once on stderr and the run goes on without ranges -- a diagnostic must never be what a job fails on.
bool tz_roctx_push(const char* name);   // returns whether a range was opened
void tz_roctx_pop();
struct tz_roctx_range {
    bool on;
    explicit tz_roctx_range(const char* name) : on(tz_roctx_push(name)) {}
    ~tz_roctx_range() {
        if (on) tz_roctx_pop();
    }
};

// ---- kernels' host launchers (device pointers only) ---------------------------------------
// tz_codec.hip
int tzk_delta(tz_ctx*, const float* pred, const uint8_t* orig, const uint8_t* d_zero_mask, int nframes,
              int H, int W, int Hp, int Wp, int16_t* out);
int tzk_delta_sd_fused(tz_ctx*, const float* pred, const uint8_t* orig, const uint8_t* d_zero_mask, int nframes, int H,
                       int W, int Hp, int Wp, int apply_offset, int16_t* out, unsigned long long* d_hist, int16_t* d_edge,
                       bool* done);
int tzk_error_bound(tz_ctx*, const uint8_t* orig, int16_t* diff, const uint8_t* h_skip, int nframes, int H,
                    int W, int mode, double b0, double b1);
bool tz_quant_is_identity(int mode, double b0, double b1);   // error_bound leaves every integer delta as it is (tz_codec.hip)
int tzk_quant_sd_fused(tz_ctx*, const float* pred, const uint8_t* orig, const uint8_t* d_zero_mask, const uint8_t* h_skip,
                       int nframes, int H, int W, int Hp, int Wp, int mode, double b0, double b1, int apply_offset,
                       int16_t* sym, unsigned long long* d_hist, int16_t* d_edge, bool* done);
int tzk_spatial_delta(tz_ctx*, const int16_t* in, size_t n, int has_carry, int16_t carry, int apply_offset,
                      int16_t* out, unsigned long long* d_hist);
int tzk_lut(tz_ctx*, const int16_t* in, size_t n, const int16_t* h_lut2112, int post_offset, int16_t* out);
// forward: int16[n] -> low-byte plane | high-byte plane (2n bytes); inverse: planes (passed as `in`) -> int16[n] at `out`
int tzk_shuffle(tz_ctx*, const int16_t* in, size_t n, uint8_t* out, int inverse);
int tzk_undelta(tz_ctx*, const int16_t* in, size_t n, int has_carry, int16_t carry, int16_t* out);
int tzk_unmap_undelta(tz_ctx*, const int16_t* in, size_t n, const int16_t* h_lut2112, int post_offset, int16_t* out);
int tzk_decode_tail_fused(tz_ctx*, const int16_t* in, const int16_t* h_lut2112, int post_offset, const float* pred,
                          const uint8_t* key, const uint8_t* d_key_mask, int nframes, int H, int W, int Hp, int Wp,
                          uint8_t* out, bool* done);
int tzk_reconstruct(tz_ctx*, const float* pred, const uint8_t* key, const uint8_t* d_key_mask, const int16_t* diff,
                    int nframes, int H, int W, int Hp, int Wp, uint8_t* out);
int tzk_sse(tz_ctx*, const uint8_t* orig, const float* pred, int nframes, int H, int W, int Hp, int Wp,
            double* h_sse);
int tzk_sse_blocks(int Hp, int Wp);
int tzk_sse_launch(tz_ctx*, const uint8_t* orig, const float* pred, int nframes, int H, int W, int Hp, int Wp,
                   double* d_part);
// tz_prednet.hip
void tz_model_free(tz_ctx* ctx);
int tz_model_predict_batch(tz_ctx* ctx, int n, const int* h_in_is_key, const int* h_in_idx, const int* h_out_idx,
                           const uint8_t* d_frames_u8, int H, int W, const float* d_in_stack, float* d_out_stack);
// slot0: first activation slot of the batch (a rollout that advances two groups of windows on two streams
// gives each group its own range of the max_batch slots)
// d_next_slot (may be NULL): per item the activation slot its prediction will occupy as the INPUT of the next call on
// this stream (-1: none) -- the prediction kernel then also writes that call's level-0 error maps (*fused_next says whether
// it did), and that call may be told to skip_err0 -- or, when only the device knows whether an item of that call starts from a
// key frame instead (DWP), to launch the error unit for err0_keys_only.
int tz_model_predict_batch_dev(tz_ctx* ctx, int n, const int* d_idx, int stride, const uint8_t* d_frames_u8, int H, int W,
                               const float* d_in_stack, float* d_out_stack, int slot0 = 0, const int* d_next_slot = nullptr,
                               bool skip_err0 = false, bool* fused_next = nullptr, bool err0_keys_only = false);
int tz_model_c0_dev(tz_ctx* ctx, const float** c0);
int tz_model_dims(tz_ctx* ctx, int* Hp, int* Wp, int* max_batch);
