/* tezip_hip.h -- C ABI of the MI355X-native TEZip hot path (libtezip_hip.so).
 *
 * The reference (kento/TEZip, /root/reference) is pure Python and has no FFI: the hot path
 * sits behind compress.run / decompress.run (src/compress.py:93, src/decompress.py:39) and
 * the operator seams inside them.  Each entry point below names the reference code it
 * replaces.  A ctypes binding (tezip_amd/_lib.py) is the "reference-side stub"; see
 * INTEGRATION.md for how the reference's own compress.py would call these.
 *
 * Conventions
 *  - extern "C", int return: 0 = TZ_OK, negative = tz_status; no exceptions cross the ABI.
 *  - Every data pointer may be HOST or DEVICE memory (detected with hipPointerGetAttributes);
 *    host buffers are staged through the context.  Outputs are complete when the call
 *    returns for host pointers; for device pointers the work is enqueued on the context's
 *    stream (tz_ctx_synchronize to wait).
 *  - One context per GPU/process; a context is not thread-safe; contexts are independent.
 *  - Frames are HWC, 3 channels; "padded" means H,W rounded up to a multiple of 8
 *    (data_utils.py:77-107) with pitch Wp*3.
 *  - All integer streams are int16, C-order over (frame, y, x, channel) (compress.py:329-340).
 */
#ifndef TEZIP_HIP_H
#define TEZIP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tz_ctx tz_ctx;

typedef enum {
    TZ_OK = 0,
    TZ_ERR_INVALID = -1,     /* bad argument (the reference prints + exit()s or raises) */
    TZ_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    TZ_ERR_HIP = -3,         /* a HIP runtime call failed: see tz_last_error */
    TZ_ERR_STATE = -4,       /* call order (no model / not prepared / no rollout) */
    TZ_ERR_NOMEM = -5,
    TZ_ERR_UNSUPPORTED = -6  /* model shape outside what the kernels cover */
} tz_status;

/* -m abs|rel|absrel|pwrel (tezip.py:96, compress.py:28-45) */
typedef enum { TZ_MODE_ABS = 0, TZ_MODE_REL = 1, TZ_MODE_ABSREL = 2, TZ_MODE_PWREL = 3 } tz_mode;

#define TZ_OFFSET 1600   /* compress.py:348 */
#define TZ_NBINS 2111    /* symbols 1600 - sd, sd in [-510, 510] (docs/index.rst:1222-1232) */
#define TZ_MAX_TABLE 1021
#define TZ_MAX_LEVELS 8

int tz_version(void);
/* "tezip_hip <version> gfx950 defines:<diagnostic switches this library was compiled with>".  A library whose string
 * names a switch after "defines:" is a measurement build (csrc/tz_wino_kernels.hip.h: TZW_ABL computes WRONG results by
 * design); bench.py and the tests refuse it.  No reference counterpart (the reference is interpreted Python). */
const char* tz_build_info(void);
const char* tz_strerror(int status);
const char* tz_last_error(const tz_ctx* ctx);

/* ---- context -------------------------------------------------------------------------
 * Replaces the device probe + TF session of tezip.py:16-26 / compress.py:281-287.
 * hip_stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL to let the
 * context create its own. */
int tz_ctx_create(int device, void* hip_stream, tz_ctx** out);
int tz_ctx_destroy(tz_ctx* ctx);
int tz_ctx_synchronize(tz_ctx* ctx);
void* tz_ctx_stream(tz_ctx* ctx);
/* Pinned (page-locked) host memory for frame stacks and payloads: the reference builds its
 * stacks with np.array / np.hstack (compress.py:116-122, 329-333); a caller that decodes its images
 * into such a buffer lets the library DMA it directly and overlap the transfer with the
 * predictor (key frames first).  Pageable pointers are accepted everywhere too; they are
 * pipelined through pinned staging buffers inside the context. */
int tz_host_alloc(size_t bytes, void** out);
int tz_host_free(void* p);

/* ---- predictor (prednet.py:24-325 used through Model.predict, compress.py:155-173,227) ----
 * weights: the Keras weight list of the PredNet layer (prednet.py:210-227): for key in
 * sorted(a, ahat, c, f, i, o), for level: kernel (3,3,Cin,Cout) HWIO float32, bias (Cout).
 * 2*(6*nb_layers-1) arrays.  Only 3x3 filters (train.py:53-55). */
int tz_model_load(tz_ctx* ctx, int nb_layers, const int* stack_sizes, const int* r_stack_sizes,
                  const float* const* weights);
/* Fix the padded frame size and the largest number of windows advanced together; allocates
 * activations and evaluates everything that does not depend on the input (t=0 states).
 * Hp, Wp: multiples of 8 and of 2^(levels-1) (compress.py:178-181: "Image size is out of scope").
 * TZ_ERR_UNSUPPORTED when a level's widest per-frame plane (gate columns / error maps) would
 * reach 2^30 floats: the kernels address inside one frame's plane with 32-bit offsets (frames and
 * batch items are 64-bit strides).  Which level binds depends on the model: for the reference's
 * (3,48,96,192) it is level 1's 192 gate columns, i.e. frames up to ~22.3 M pixels (4096 x 4096
 * passes, 8192 x 8192 does not); the error text names the model's limit.  A deviation: the
 * reference's frame size is bounded by memory only. */
int tz_model_prepare(tz_ctx* ctx, int Hp, int Wp, int max_batch);
/* X_hat[0,0] of predict((1,2,Hp,Wp,3)) (compress.py:197): input independent. out: Hp*Wp*3 f32 */
int tz_predict_c0(tz_ctx* ctx, float* out);
/* X_hat[0,1] for n independent padded float32 frames (compress.py:224-229). */
int tz_predict_next(tz_ctx* ctx, const float* frames, int n, float* out);
/* Debug/parity taps of the last tz_predict_next call with n == 1: kind 0 = e_l(t0),
 * 1 = r_l(t1); out sized (Hp>>l)*(Wp>>l)*channels. */
int tz_predict_tap(tz_ctx* ctx, int kind, int level, float* out);
/* Diagnostic: which convolution kernels the predictor launches.  1 (default; env TEZIP_CONV16=0
 * starts a context at 0) = the LDS-DMA kernels k_conv16 / k_conv16b / k_conv_small wherever a
 * convolution qualifies, 0 = the general register-staged kernel k_conv3x3 everywhere.  Both walk
 * the same fmaf chains: results are bit-identical (tests/test_gpu_fullsize.py).
 * Bits 1-2 select the small-grid kernel k_convlat (one accumulator tile per wave): 0 = where a cost
 * model expects it to be faster (default; env TEZIP_LAT=0|1|2 sets a context's start value),
 * 1 (value 2) = never, 2 (value 4) = wherever a convolution is eligible. */
int tz_set_conv_impl(tz_ctx* ctx, int lds_dma);
/* The arithmetic contract of the predictor.  The reference leaves the float32 summation order of its
 * convolutions (prednet.py:254-277) to Keras / TensorFlow / cuDNN; this library fixes it, because a lossless decoder must
 * regenerate the encoder's predictions bit for bit (decompress.py:252-253).  1 = TZ-PA1: every convolution one direct fmaf
 * chain (rounds 1-3; what files written by earlier builds need).  2 = TZ-PA2: the per-frame convolutions of levels >= 1
 * evaluate their same-resolution source as Winograd F(2x2, 3x3) chains (2.25x fewer multiplies; oracle/tz_oracle.c
 * conv3x3_wino), everything else as in TZ-PA1.  Encoder and decoder must use the same contract: the on-disk format of the
 * reference has no place to record it.  0 (the default; or the environment variable TEZIP_PA) = by padded frame size, which
 * both sides know: TZ-PA2 from 256 x 256 pixels on, TZ-PA1 below.  tz_get_contract returns the contract in force (1 or 2)
 * for the prepared model.  Switching re-prepares nothing.
 * The prediction stack a rollout leaves in the context is STAMPED with the contract that produced it: tz_rollout_contract
 * returns that stamp (1 or 2; TZ_ERR_STATE without a rollout) -- it is what the host records next to entropy.dat
 * (tezip_amd.json) and what a decoder adopts --, and tz_encode / tz_encode_begin / tz_encode_delta / tz_decode /
 * tz_decode_delta fail with TZ_ERR_STATE when the contract in force differs from the stamp (a tz_set_contract between a
 * rollout and its encode/decode), because decompress.py:252-253 needs the decoder's predictions bit-identical to the
 * encoder's. */
int tz_set_contract(tz_ctx* ctx, int contract);
int tz_get_contract(tz_ctx* ctx);
int tz_rollout_contract(tz_ctx* ctx);
/* Diagnostic: the inverse scan of decompress.py:22-29 (k_scan2p) lets a workgroup wait for the block sums of the workgroups
 * in front of it; that wait is bounded, and an expiry surfaces as TZ_ERR_HIP at the context's next stream synchronisation
 * (tz_ctx_synchronize, or any call that delivers host results).  This entry makes the next scans wait for the status words
 * of the launch `epoch_skew` launches ahead (never written when != 0) and give up after poll_limit polls (0 = built-in
 * 2^22), so that a test can see the failure path; (0, 0) restores normal operation. */
int tz_scan_fault_inject(tz_ctx* ctx, unsigned epoch_skew, unsigned poll_limit);
/* Diagnostic: the device's own statement of the predictor's scalar functions (prednet.py:79-81,198-205: Keras
 * hard_sigmoid and tanh in the fixed arithmetic of DESIGN.md section 3) on n caller-chosen inputs, so that a test can
 * compare them bit for bit with the oracle's; recip_mismatches (may be NULL) receives the number of float32 values d in
 * [4, 2^27] for which the kernels' division-free 1 - 2/d differs from the IEEE division it stands for (must be 0). */
int tz_act_probe(tz_ctx* ctx, const float* x, size_t n, float* hard_sigmoid, float* tanh_out,
                 unsigned long long* recip_mismatches);

/* ---- rollout (compress.py:183-268 encoder; decompress.py:115-186 decoder) -----------------
 * frames: nt*H*W*3 uint8.  window > 0: SWP (-w); window == 0: DWP with `threshold` (-t).
 * key_mask[nt] (host) receives 1 for key frames.  mse_log (host, nt doubles, may be NULL)
 * receives the per-step window MSE of compress.py:246 (entries 0..warm_up are 0); it is
 * always computed for DWP and only on request for SWP.
 * The prediction stack (nt padded float32 frames; key slots hold C0) and the frames stay in
 * the context for tz_encode.  Rejects nt < warm_up+2 (the reference misbehaves there). */
int tz_rollout(tz_ctx* ctx, const uint8_t* frames, int nt, int H, int W, int warm_up, int window,
               double threshold, uint8_t* key_mask, double* mse_log);
/* Streaming ingestion (the reference holds every frame in RAM, compress.py:116-122): instead of one
 * stack, stage the frames window by window -- tz_frames_begin(nt,H,W), then tz_frames_put for any
 * partition of [0,nt) (asynchronous on the context's copy stream; a pageable source is free again
 * on return, a pinned one after tz_frames_fence) -- and call tz_rollout with frames == NULL.
 * tz_frames_get copies frames of the resident stack back (e.g. the key frames for key_frame.dat). */
int tz_frames_begin(tz_ctx* ctx, int nt, int H, int W);
int tz_frames_put(tz_ctx* ctx, int first, int count, const uint8_t* frames);
int tz_frames_fence(tz_ctx* ctx);
int tz_frames_get(tz_ctx* ctx, int first, int count, uint8_t* out);
/* Decoder replay: key_frames = the key_frame.dat stack (zeros except key frames); key
 * positions are recovered as decompress.py:123-129 does (any non-zero sample). */
int tz_rollout_decode(tz_ctx* ctx, const uint8_t* key_frames, int nt, int H, int W, int warm_up,
                      uint8_t* key_mask);
/* Copy out the prediction stack of the last rollout: nt*Hp*Wp*3 float32. */
int tz_get_predictions(tz_ctx* ctx, float* out);

/* ---- encoder back half on the context-resident rollout (compress.py:289-373) ---------------
 * payload: nt*H*W*3 int16 = rank(1600 - sd) when bit 0 of `entropy` is set, else sd.
 * Bit 1 of `entropy` (value 2; NOT a reference feature, off in the reference's format): the
 * payload is returned byte-shuffled, i.e. as nt*H*W*3 low bytes followed by as many high bytes.
 * table: >= TZ_MAX_TABLE int16 (host), *table_len receives T (or -1 when entropy == 0).
 * delta_out (may be NULL): the int16 delta stack after quantisation, before the spatial delta.
 * A job whose WORST-CASE tolerance is <= 0.499 (abs |b0|; rel / pwrel 255 b0; absrel the smaller of |b0| and 255 b1)
 * is served by the lossless kernels: compress.py:23-70 then leaves every integer delta as it is (two different
 * neighbours always close a run, and a run of equal deltas d gets trunc((fl(d+E) + fl(d-E)) / 2) = d; proof at
 * tz_quant_is_identity in csrc/tz_codec.hip).  The result is the general quantiser's, byte for byte; TEZIP_QMAP=0 runs
 * that instead. */
int tz_encode(tz_ctx* ctx, int mode, double b0, double b1, int entropy, int16_t* payload,
              int16_t* table, int* table_len, int16_t* delta_out);
/* Streaming delivery: tz_encode with payload == NULL keeps the payload in the context; it is then
 * fetched in pieces of `count` int16 elements starting at `offset` (compress.py:375-400 appends and
 * compresses one monolithic array). */
int tz_payload_get(tz_ctx* ctx, size_t offset, size_t count, int16_t* out);
/* Deferred hand-over of the payload (a caller that compresses one sequence after the other; compress.py:375-400 is one
 * blocking pass per job).  With tz_set_payload_deferred(ctx, 1) a tz_encode whose `payload` is PINNED host memory
 * (tz_host_alloc) and whose entropy bit is set returns as soon as the last chunk of the payload is queued on the copy
 * stream: table and *table_len are final, the payload buffer is complete only after tz_payload_wait -- the device -> host
 * transfer (2.4 ms for a cfg3 sequence) then runs under the next sequence's tz_rollout instead of in front of it.  At
 * most one transfer is in flight: the next tz_encode orders its own remap behind it, so a caller alternates between two
 * host buffers and calls tz_payload_wait (free by then) before it reads the older one.  Every other form of tz_encode,
 * tz_ctx_synchronize and tz_ctx_destroy settle a transfer in flight first. */
int tz_set_payload_deferred(tz_ctx* ctx, int on);
int tz_payload_wait(tz_ctx* ctx);
/* First stage of tz_encode only (compress.py:292-319): delta + error-bound quantisation of the
 * context-resident rollout -> int16 delta stack nt*H*W*3.  Used when frame windows are sharded
 * over GPUs: the spatial delta and the histogram then need a carry / a sum across shards
 * (tz_spatial_delta with has_carry, tz_build_table, tz_remap). */
int tz_encode_delta(tz_ctx* ctx, int mode, double b0, double b1, int16_t* delta_out);
/* tz_encode in two phases, for jobs whose frame windows are sharded over GPUs (the
 * reference is single-process).  The spatial delta runs over the whole flattened stack
 * (compress.py:339) and the rank table comes from the global histogram (compress.py:354-361), so a
 * shard runs compress.py:292-355 on its own frames (begin), the ranks exchange one carry element
 * and sum the 2111 counters, and the shard finishes compress.py:356-373 with the global table
 * (finish).  Same kernels as tz_encode.
 * begin: hist (host, TZ_NBINS counters, may be NULL when entropy == 0) receives this shard's counts
 * taken WITHOUT a carry; edge[0], edge[1] (host) the first and the last element of the shard's
 * quantised delta stack.  edge[1] is the carry of the next shard; with its own carry c a shard moves
 * its first symbol in the histogram from 1600 - edge[0] to 1600 - (int16)(c - edge[0]).
 * finish: has_carry/carry as in tz_spatial_delta; table_len >= 0: remap with `table` (the table of the
 * summed histogram, tz_build_table), -1: no remap (must match begin's entropy flag); payload NULL: the
 * payload stays in the context (tz_payload_get). */
int tz_encode_begin(tz_ctx* ctx, int mode, double b0, double b1, int entropy, unsigned long long* hist,
                    int16_t* edge);
int tz_encode_finish(tz_ctx* ctx, int has_carry, int16_t carry, const int16_t* table, int table_len,
                     int16_t* payload);
/* ---- decoder back half (decompress.py:203-256): payload (+table) -> nt*H*W*3 uint8 frames,
 * using the prediction stack of the last tz_rollout_decode.  payload_len (elements) must be
 * nt*H*W*3 of that rollout: the reference fails at its reshape otherwise (decompress.py:240). */
int tz_decode(tz_ctx* ctx, const int16_t* payload, size_t payload_len, const int16_t* table, int table_len,
              uint8_t* frames_out);
/* Streaming decode (decompress.py:87-113 decompresses and holds both files whole): stage the
 * payload in pieces -- tz_payload_begin(count), tz_payload_put(offset, count, src) on the copy
 * stream -- and the key-frame stack with tz_frames_begin / tz_frames_put, then tz_rollout_decode
 * with key_frames == NULL and tz_decode with payload == NULL; with frames_out == NULL the decoded
 * frames stay in the context and are fetched window by window with tz_decoded_get. */
int tz_payload_begin(tz_ctx* ctx, size_t count);
int tz_payload_put(tz_ctx* ctx, size_t offset, size_t count, const int16_t* src);
int tz_decoded_get(tz_ctx* ctx, int first, int count, uint8_t* out);
/* Last stage of tz_decode only (decompress.py:252-256): reconstruct from an already decoded
 * int16 delta stack (sharded decoding: the inverse scan carry comes from the previous shard). */
int tz_decode_delta(tz_ctx* ctx, const int16_t* delta, uint8_t* frames_out);

/* ---- operator seams, usable stand-alone (each mirrors one reference helper) -----------------
 * tz_delta_encode: compress.py:292-314.  pred: nframes padded f32 frames; orig: nframes
 * unpadded u8 frames; zero_mask[nframes] (host): 1 => that frame's delta is forced to 0. */
int tz_delta_encode(tz_ctx* ctx, const float* pred, const uint8_t* orig, const uint8_t* zero_mask,
                    int nframes, int H, int W, int16_t* out);
/* tz_error_bound: compress.py:23-70 applied per frame and channel as compress.py:316-319 does.
 * diff is updated in place; skip_mask[nframes] (host): 1 => frame left untouched.
 * Domain: any int16 stack (the deltas of compress.py:292-314 lie in [-255, 255]; wider values take the reference's
 * double test at every step instead of the integer walk's width table -- same results, slower).
 * Negative tolerances: `abs` takes |b| (compress.py:29).  For rel / pwrel with b < 0 and absrel with a negative relative
 * bound the reference fails only where E < 0 meets the FIRST element of a chain ((inf + -inf)/2 = NaN stored into an int
 * array, compress.py:60-61) and otherwise carries on with every element a run of its own; this library rejects every
 * such call with TZ_ERR_INVALID -- a deliberate superset of the reference's failure, both oracles do the same. */
int tz_error_bound(tz_ctx* ctx, const uint8_t* orig, int16_t* diff, const uint8_t* skip_mask,
                   int nframes, int H, int W, int mode, double b0, double b1);
/* tz_spatial_delta: compress.py:73-77 (+ the 1600 offset of :348 when apply_offset).
 * has_carry: treat `carry` as the element preceding in[0] (shard boundary). hist (may be
 * NULL): TZ_NBINS uint64 counts of the output symbols are ADDED (compress.py:354). */
int tz_spatial_delta(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry,
                     int apply_offset, int16_t* out, unsigned long long* hist);
/* Opt-in byte shuffle (no reference counterpart; BASELINE.json's north star names the stage):
 * int16[n] <-> n low bytes | n high bytes.  n must be a multiple of 8 for the forward direction. */
int tz_byte_shuffle(tz_ctx* ctx, const int16_t* in, size_t n, uint8_t* out);
int tz_byte_unshuffle(tz_ctx* ctx, const uint8_t* in, size_t n, int16_t* out);
/* tz_build_table: compress.py:352-361 (host): count desc, ties ascending symbol. */
int tz_build_table(const unsigned long long* hist, int nbins, int16_t* table, int* table_len);
/* tz_remap: compress.py:84-90 (symbol -> rank). */
int tz_remap(tz_ctx* ctx, const int16_t* in, size_t n, const int16_t* table, int table_len, int16_t* out);
/* tz_unmap: decompress.py:31-36 (+ 1600 - x of :236 when apply_offset). */
int tz_unmap(tz_ctx* ctx, const int16_t* in, size_t n, const int16_t* table, int table_len,
             int apply_offset, int16_t* out);
/* tz_spatial_undelta: decompress.py:22-29 as a wrap-around prefix scan. has_carry: `carry`
 * is the decoded element preceding in[0]. */
int tz_spatial_undelta(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry,
                       int16_t* out);
/* tz_reconstruct: decompress.py:252-256,269.  key_mask[nframes] (host): 1 => the base is
 * the key byte (key_frames), else trunc(pred*255). */
int tz_reconstruct(tz_ctx* ctx, const float* pred, const uint8_t* key_frames, const uint8_t* key_mask,
                   const int16_t* diff, int nframes, int H, int W, uint8_t* out);
/* tz_window_sse: the inner sum of compress.py:246 for nframes (sum over the padded frame of
 * (x/255 - pred)^2 in float64, fixed summation order). sse: nframes doubles (host). */
int tz_window_sse(tz_ctx* ctx, const uint8_t* orig, const float* pred, int nframes, int H, int W, double* sse);

/* ---- timing helper: HIP events on the context's stream (bench.py) -------------------------- */
int tz_timer_start(tz_ctx* ctx);
int tz_timer_stop(tz_ctx* ctx, float* ms);
/* Per-kernel-class accumulated device time since the last reset, measured with HIP events
 * around every launch of that class when profiling is enabled (adds a sync per query only).
 * names: tz_prof_name(i), i < tz_prof_count(). */
int tz_prof_enable(tz_ctx* ctx, int on);
int tz_prof_count(void);
const char* tz_prof_name(int i);
int tz_prof_get(tz_ctx* ctx, int i, double* total_ms, long long* launches);
int tz_prof_reset(tz_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
