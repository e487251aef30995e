// Delta / quantise / entropy-prep kernels and their inverses for gfx950 (MI355X).
// HBM-bound integer/byte work: wide coalesced accesses (16 B per lane where the layout
// allows), LDS-privatised histogram, grid-stride launches of ~8 blocks per CU.
// Reference semantics are cited per kernel (paths into /root/reference/src).
#include <algorithm>

#include "tz_internal.h"

static constexpr int kBlocksPerCU = 8;
static constexpr int kCUs = 256;

static inline int grid_for(size_t work_items, int block) {
    size_t g = (work_items + block - 1) / block;
    size_t cap = (size_t)kCUs * kBlocksPerCU;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

typedef short short8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------- delta
// compress.py:292-314: d = (int)(pred_f32 * 255.0f) - orig ; frames flagged in zero_mask -> 0.
// Fast path: frame needs no padding (H==Hp, W==Wp): pred, orig and out share one flat index.
__global__ __launch_bounds__(256) void k_delta_flat(const float4* __restrict__ pred, const uint2* __restrict__ orig,
                                                    const uint8_t* __restrict__ zero_mask, size_t n8,
                                                    unsigned frame_elems8, short8* __restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float4 p0 = pred[2 * i], p1 = pred[2 * i + 1];
        uint2 o = orig[i];
        unsigned f = (unsigned)(i / frame_elems8);
        short8 r;
        if (zero_mask[f]) {
            r = (short8)(0);
        } else {
            r[0] = (short)((int)(p0.x * 255.0f) - (int)(o.x & 0xff));
            r[1] = (short)((int)(p0.y * 255.0f) - (int)((o.x >> 8) & 0xff));
            r[2] = (short)((int)(p0.z * 255.0f) - (int)((o.x >> 16) & 0xff));
            r[3] = (short)((int)(p0.w * 255.0f) - (int)(o.x >> 24));
            r[4] = (short)((int)(p1.x * 255.0f) - (int)(o.y & 0xff));
            r[5] = (short)((int)(p1.y * 255.0f) - (int)((o.y >> 8) & 0xff));
            r[6] = (short)((int)(p1.z * 255.0f) - (int)((o.y >> 16) & 0xff));
            r[7] = (short)((int)(p1.w * 255.0f) - (int)(o.y >> 24));
        }
        out[i] = r;
    }
}

// General path: crop of a padded prediction (pitch Wp*3).
__global__ __launch_bounds__(256) void k_delta_crop(const float* __restrict__ pred, const uint8_t* __restrict__ orig,
                                                    const uint8_t* __restrict__ zero_mask, size_t n, int H, int W,
                                                    int Hp, int Wp, int16_t* __restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t row = (size_t)W * 3, fe = (size_t)H * row, fp = (size_t)Hp * Wp * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        size_t f = i / fe, r = i - f * fe;
        size_t y = r / row, xc = r - y * row;
        float p = pred[f * fp + y * (size_t)Wp * 3 + xc];
        out[i] = zero_mask[f] ? (int16_t)0 : (int16_t)((int)(p * 255.0f) - (int)orig[i]);
    }
}

int tzk_delta(tz_ctx* ctx, const float* pred, const uint8_t* orig, const uint8_t* d_zero_mask, int nframes, int H,
              int W, int Hp, int Wp, int16_t* out) {
    size_t n = (size_t)nframes * H * W * 3;
    if (n == 0) return TZ_OK;
    tz_prof_scope ps(ctx, TZP_DELTA);
    if (H == Hp && W == Wp && ((size_t)H * W * 3) % 8 == 0) {
        size_t n8 = n / 8;
        hipLaunchKernelGGL(k_delta_flat, dim3(grid_for(n8, 256)), dim3(256), 0, ctx->stream, (const float4*)pred,
                           (const uint2*)orig, d_zero_mask, n8, (unsigned)((size_t)H * W * 3 / 8), (short8*)out);
    } else {
        hipLaunchKernelGGL(k_delta_crop, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, pred, orig, d_zero_mask,
                           n, H, W, Hp, Wp, out);
    }
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

// ------------------------------------------------------------- block-private symbol histogram
// compress.py:354 (bincount of the symbols 1600 - sd).  The symbols of a prediction residual pile up
// on a handful of values around the centre c (1600, or 0 without the offset), so a plain LDS
// histogram spends its time serialising atomics on the same few words (rounds 1-2: one copy per wave,
// 4.7 M bank-conflict cycles per 7.9 M LDS operations, the spatial-delta kernel ran at 41 % of HBM
// against 74 % without the histogram).  Now:
//   * the centre symbol never touches the bins: a wave counts it with a ballot into a scalar register;
//   * the 128 bins around c have 8 copies per wave, interleaved as [bin][copy] with copy = lane & 7, so
//     that a wave's 64 lanes spread over all 32 banks (bank = (bin % 4) * 8 + copy) and only the ~2 lanes
//     that share copy AND bin % 4 meet in a bank;
//   * everything else (rare) goes to one full-range copy per block.
// flush() folds the copies into the full-range one and adds its non-zero bins to the global counters.
static constexpr int HC_HALF = 64, HC_BINS = 2 * HC_HALF, HC_COPIES = 8;
// Blocks of the histogram kernels are 1024 threads, two per CU: what a block's flush costs is global --
// every block adds each of its non-zero bins to the same ~450 counters, and same-address atomics from
// different workgroups are served one after the other (about 12 ns each): with 2048 blocks of 256
// threads that was 25 us of a 76 us kernel, whatever the LDS part did.  (The copies are per BLOCK, not per
// wave: only the lanes of one instruction can conflict, instructions of different waves pass through the
// LDS one after the other anyway.)
static constexpr int HB_THREADS = 1024, HB_GRID = 2 * kCUs;
static constexpr int HL_CENTRAL = 0, HL_FULL = (HC_BINS + 1) * HC_COPIES /* + one junk bin per copy */,
                     HL_SINK = HL_FULL + TZ_NBINS + 1, HL_WORDS = HL_SINK + 64;
struct HistLds {
    // [bin][copy] | full range | one sink word per lane
    unsigned w[HL_WORDS];
};

struct HistAcc {
    unsigned n0 = 0;   // wave-uniform count of the centre symbol
};

__device__ __forceinline__ void hist_clear(HistLds& h) {
    for (int k = threadIdx.x; k < HL_WORDS; k += blockDim.x) h.w[k] = 0;
    __syncthreads();
}

// Eight symbols of one lane, packed two per dword.  The update is ONE unconditional LDS atomic per
// element in straight-line code (a branch per element cost more than the atomics it saved: the
// spatial-delta kernel is as much VALU-issue as HBM bound): the centre symbol goes to the lane's own
// sink word (it is counted by the ballot), symbols within +-64 of the centre to [bin][copy], anything
// farther to the junk bin of the copy -- and, in a branch the wave takes only when one of its 512
// symbols is that far out, to the full-range bins.
__device__ __forceinline__ void hist_add8(HistLds& h, HistAcc& acc, const unsigned* Y, int c) {
    const int lane = threadIdx.x & 63;
    const int base = HL_CENTRAL + (lane & 7), sink = HL_SINK + lane;
    unsigned far = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int y = (k & 1) ? ((int)Y[k >> 1] >> 16) : (int)(short)(Y[k >> 1] & 0xFFFFu);
        const bool hot = y == c;
        acc.n0 += (unsigned)__popcll(__ballot(hot));
        const unsigned rel = (unsigned)(y - (c - HC_HALF));
        const unsigned idx = min(rel, (unsigned)HC_BINS);
        far |= idx >> 7;   // HC_BINS == 128
        atomicAdd(&h.w[hot ? sink : base + (int)idx * HC_COPIES], 1u);
    }
    if (__any(far != 0)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int y = (k & 1) ? ((int)Y[k >> 1] >> 16) : (int)(short)(Y[k >> 1] & 0xFFFFu);
            if ((unsigned)(y - (c - HC_HALF)) >= (unsigned)HC_BINS && (unsigned)y < (unsigned)TZ_NBINS) atomicAdd(&h.w[HL_FULL + y], 1u);
        }
    }
}
static_assert(HC_BINS == 128, "hist_add8 takes the far flag from bit 7 of the clamped index");

// every thread of the block calls this (after its last add)
__device__ __forceinline__ void hist_flush(HistLds& h, const HistAcc& a, int c, unsigned long long* __restrict__ hist) {
    if ((threadIdx.x & 63) == 0 && a.n0 && c >= 0 && c < TZ_NBINS) atomicAdd(&h.w[HL_FULL + c], a.n0);
    __syncthreads();
    for (int b = threadIdx.x; b < HC_BINS; b += blockDim.x) {
        const int y = c - HC_HALF + b;
        unsigned t = 0;
#pragma unroll
        for (int k = 0; k < HC_COPIES; ++k) t += h.w[HL_CENTRAL + b * HC_COPIES + k];
        if (t && y >= 0 && y < TZ_NBINS) atomicAdd(&h.w[HL_FULL + y], t);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < TZ_NBINS; k += blockDim.x) {
        const unsigned v = h.w[HL_FULL + k];
        if (v) atomicAdd(&hist[k], (unsigned long long)v);
    }
}

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) {   // two int16 lanes, wrap-around
    const short2v r = __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b);
    return __builtin_bit_cast(unsigned, r);
}

// spatial delta of 8 consecutive int16 (V, two per dword) given the element in front of them:
// sd[k] = x[k-1] - x[k]; symbols y = 1600 - sd when `offs` = (1600, 1600), y = sd when offs is 0 and `negate`
// is false ... both forms are y = offs - (P - V) resp. P - V; P is V shifted up by one element.
__device__ __forceinline__ void sdelta8(const uint4 V, unsigned prev16, bool apply_offset, unsigned* Y) {
    const unsigned P0 = __builtin_amdgcn_alignbit(V.x, prev16 << 16, 16);
    const unsigned P1 = __builtin_amdgcn_alignbit(V.y, V.x, 16);
    const unsigned P2 = __builtin_amdgcn_alignbit(V.z, V.y, 16);
    const unsigned P3 = __builtin_amdgcn_alignbit(V.w, V.z, 16);
    const unsigned C = ((unsigned)TZ_OFFSET << 16) | (unsigned)TZ_OFFSET;
    Y[0] = pk_sub(P0, V.x);
    Y[1] = pk_sub(P1, V.y);
    Y[2] = pk_sub(P2, V.z);
    Y[3] = pk_sub(P3, V.w);
    if (apply_offset) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Y[j] = pk_sub(C, Y[j]);
    }
}

// --------------------------------------------------------------------------- quantiser
// compress.py:23-70, one chain per (frame, channel) over H*W elements, row-major.
// Stage 1 (k_q_minmax, k_q_bound): per-chain tolerance E for rel / absrel from max-min of the
//                      ORIGINAL slab (compress.py:31-33,36-43).
// Stage 2 (k_q_heads, k_q_stitch): exact wave-parallel form of the greedy interval-intersection
//                      segmentation (see the kernels).  Result: one bit per element (`spec`, a 64-bit
//                      mask per 64-element chunk and chain) that says where a run starts, and the
//                      truncated median of every run at its HEAD position of `tmp` (positions that are
//                      not heads hold garbage: the masks are authoritative, nothing initialises tmp).
// Stage 3 (k_q_last / k_q_carry / k_q_fill): forward-fill the run values.  k_q_fill either writes them
//                      back as the quantised delta stack (stand-alone tz_error_bound, delta tap) or --
//                      fused encode -- goes straight on to the spatial delta, the 1600 offset and the
//                      histogram (compress.py:339-355) and writes symbols: the quantised deltas never
//                      exist in memory.
// The deltas a chain is made of come from a materialised int16 stack or, fused encode, straight from
// prediction and original (compress.py:292-314 evaluated where it is needed): QSrc.
// Round 3 traffic at cfg3, `abs 2`: heads 315 MB + fill 270 MB + remap 252 MB against 1.75 GB through
// k_delta, k_q_init, k_q_last over tmp, k_q_fill, k_sdelta in round 2.
struct QParams {
    int mode;
    double b0, b1;
};

struct QSrc {
    const int16_t* diff;   // nframes * HW * 3, or nullptr: then
    const float* pred;     // ... trunc(pred * 255) - orig of unpadded frames (H == Hp, W == Wp)
};

template <bool FP>
__device__ __forceinline__ int q_delta(const QSrc& s, const uint8_t* __restrict__ orig, size_t e) {
    if (FP) return (int)(s.pred[e] * 255.0f) - (int)orig[e];
    return (int)s.diff[e];
}

// min/max of the original slab per (frame, channel): 12 bytes (4 interleaved RGB pixels) per
// lane and iteration, QBB blocks per frame combined with integer atomics (mm[f][c] = {min, max}).
static constexpr int QBB = 16;
__global__ __launch_bounds__(256) void k_q_minmax(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ skip,
                                                  int HW, int* __restrict__ mm) {
    const int f = blockIdx.y;
    if (skip[f]) return;
    const uint8_t* o = orig + (size_t)f * HW * 3;
    int mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
    const int ngroups = HW / 4;  // 4 pixels = 12 bytes = 3 aligned dwords (frame base is 4-byte aligned when HW*3 % 4 == 0)
    const bool aligned = (((size_t)f * HW * 3) & 3) == 0 && (((uintptr_t)orig) & 3) == 0;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += QBB * 256) {
        unsigned w[3];
        if (aligned) {
            const unsigned* p = (const unsigned*)(o + (size_t)g * 12);
            w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
        } else {
            const uint8_t* p = o + (size_t)g * 12;
            for (int k = 0; k < 3; ++k) w[k] = p[4 * k] | (p[4 * k + 1] << 8) | (p[4 * k + 2] << 16) | ((unsigned)p[4 * k + 3] << 24);
        }
#pragma unroll
        for (int b = 0; b < 12; ++b) {
            int v = (w[b >> 2] >> (8 * (b & 3))) & 0xff, c = b % 3;
            mn[c] = min(mn[c], v);
            mx[c] = max(mx[c], v);
        }
    }
    if (blockIdx.x == 0)
        for (int p = ngroups * 4 + threadIdx.x; p < HW; p += 256)
            for (int c = 0; c < 3; ++c) {
                int v = o[(size_t)p * 3 + c];
                mn[c] = min(mn[c], v);
                mx[c] = max(mx[c], v);
            }
    for (int c = 0; c < 3; ++c) {
        for (int s = 32; s >= 1; s >>= 1) {
            mn[c] = min(mn[c], __shfl_down(mn[c], s, 64));
            mx[c] = max(mx[c], __shfl_down(mx[c], s, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&mm[(f * 3 + c) * 2], mn[c]);
            atomicMax(&mm[(f * 3 + c) * 2 + 1], mx[c]);
        }
    }
}

__global__ void k_q_mm_init(int* __restrict__ mm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mm[i] = (i & 1) ? 0 : 255;
}

// tolerance per chain from max-min of the ORIGINAL slab (compress.py:31-33, 36-43)
__global__ void k_q_bound(const int* __restrict__ mm, const uint8_t* __restrict__ skip, QParams qp, int nframes,
                          double* __restrict__ E) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nframes * 3 || skip[i / 3]) return;
    double range = (double)(mm[2 * i + 1] - mm[2 * i]);
    double e;
    if (qp.mode == TZ_MODE_REL) {
        e = range * qp.b0;
    } else {  // absrel
        double a = fabs(qp.b0), r = range * qp.b1;
        e = a < r ? a : r;
    }
    E[i] = e;
}

// Wave-parallel exact greedy segmentation.  The chain of a (frame, channel) is walked in chunks of
// 64 elements (lane i <-> element).  A run started at s breaks at the first i with
// min(Du[s..i]) < max(Dl[s..i]) (compress.py:60 -- a-b<0 <=> a<b in IEEE), which is monotone in i, so
// per chunk:
//   WORKER waves (carry independent, one chunk each per round; q_chunk):
//   1. range tables T_k[i] = (min Du, max Dl) over [i, i+2^k) by shuffles (k = 0..5),
//   2. nxt[s] = first break after s for EVERY s (fresh start) by binary lifting over T_k,
//      together with the run's (u, l) up to the break,
//   3. inclusive prefix (min Du, max Dl) from the chunk start,
//   2b. heads[s] = bit mask of the run heads reached from a start at s (pointer doubling).
//   RESOLVER wave (sequential over chunks, one round behind the workers):
//   4. the run carried in from earlier chunks breaks at the first i with
//      min(u, P_u[i]) < max(l, P_l[i])  (ballot); the true heads are the nxt-chain from that
//      break, which the workers have already expanded into a bit mask per start: one lookup,
//   5. every head whose run closes inside the chunk stores trunc((u+l)/2) (compress.py:61,
//      truncation by the int64 store) at the head position of `tmp`; the last head carries on.
// The walk of the resolver is the serial critical path (about 0.3 us per chunk, 4096 chunks per
// 512x512 chain), so the chain is cut into QSEG SEGMENTS that are walked concurrently, each from a
// fresh start at its first element (round 2; one workgroup of 1 + QW waves per segment, eight of them
// share a CU).  A fresh start is a guess -- the true run entering a segment began earlier -- and
// k_q_stitch repairs it exactly: two greedy chains over the same data never cross and coincide from
// their first common head on, so the true chain is followed from the segment start only until it
// hits a head of the speculative chain (typically within the first chunk); the speculative heads
// before that point are dropped from the masks, the true ones entered, everything behind it is already
// right.  Nothing depends on run lengths; elements past the chain end are (+inf, -inf) and can neither
// break nor tighten a run.
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

static constexpr int QW = 3;     // worker waves per segment (+1 resolver wave = 256 threads)
static constexpr int QSEG = 8;   // segments per chain

struct QChunk {
    double cu, cl, pu, pl;
    unsigned long long heads;
    int nxt;
};

struct QSlot {
    double cu[64], cl[64], pu[64], pl[64];
    unsigned long long heads[64];  // heads[s]: bit mask of the nxt-chain that starts at s
    int nxt[64];
};

struct QState {  // the run that is open at a segment boundary
    double u, l;
    int head, pad;
};

// steps 1-3 for chunk `ch` of channel c (every lane returns its element's entries).
//
// The cross-lane traffic of these steps is what the quantiser costs: a 64-bit shuffle is two
// ds_bpermute_b32, and with (min Du, max Dl) as two doubles a chunk took 86 of them -- 1.1 ms of LDS
// crossbar per 4096-chunk chain and CU, which is the 1.29 ms k_q_heads ran at in round 1 (cutting the
// resolver's walk into segments alone changed nothing: profiles/r02/quantiser.md).  For abs / rel /
// absrel the tolerance E is one constant per chain, so Du = fl(d + E) and Dl = fl(d - E) are monotone
// in the integer delta d: min Du = fl(min d + E), max Dl = fl(max d - E) EXACTLY.  The tables, the
// prefix and the lifting therefore run on (min d, max d) packed as two int16 in ONE 32-bit word (35
// shuffles per chunk), and the doubles the reference compares (compress.py:55-60) are formed from
// them only where a break is tested or a run value is produced.  pwrel (E = orig * b per element)
// keeps the double tables.
__device__ __forceinline__ int q_pack(int mn, int mx) { return (mn & 0xFFFF) | (mx << 16); }
__device__ __forceinline__ int q_lo(int p) { return (int)(short)(p & 0xFFFF); }
__device__ __forceinline__ int q_hi(int p) { return p >> 16; }

// fe0 = element index of the frame's first sample in the whole stack (f * HW * 3)
template <bool PW, bool FP>
__device__ __forceinline__ QChunk q_chunk(const QSrc& src, const uint8_t* __restrict__ orig, size_t fe0, int c, int ch, int HW,
                                          const QParams& qp, double E, int lane) {
    const int idx = ch * 64 + lane;
    unsigned long long M = 1ull << lane;
    int pos = lane + 1;
    QChunk out;
    if (!PW) {
        // past the chain end: (min, max) = (+32767, -32768) can neither break nor tighten a run
        int mn = 32767, mx = -32768;
        if (idx < HW) mn = mx = q_delta<FP>(src, orig, fe0 + (size_t)idx * 3 + c);
        int tb[6];
        tb[0] = q_pack(mn, mx);
#pragma unroll
        for (int k = 1; k < 6; ++k) {  // shfl_down past lane 63 returns the caller's own value
            const int q = __shfl_down(tb[k - 1], 1 << (k - 1), 64);
            tb[k] = q_pack(min(q_lo(tb[k - 1]), q_lo(q)), max(q_hi(tb[k - 1]), q_hi(q)));
        }
        int pp = tb[0];
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int q = __shfl_up(pp, sft, 64);
            if (lane >= sft) pp = q_pack(min(q_lo(pp), q_lo(q)), max(q_hi(pp), q_hi(q)));
        }
        int cmn = mn, cmx = mx;
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            const int step = 1 << k;
            const int srcl = pos < 63 ? pos : 63;
            const int q = __shfl(tb[k], srcl, 64);
            const int nmn = min(cmn, q_lo(q)), nmx = max(cmx, q_hi(q));
            const double nu = (double)nmn + E, nl = (double)nmx - E;
            const bool ok = (pos + step <= 64) && !(nu - nl < 0.0);
            if (ok) {
                cmn = nmn;
                cmx = nmx;
                pos += step;
            }
        }
        out.cu = (double)cmn + E;
        out.cl = (double)cmx - E;
        out.pu = (double)q_lo(pp) + E;
        out.pl = (double)q_hi(pp) - E;
    } else {
        const double inf = __builtin_huge_val();
        double du = inf, dl = -inf;
        if (idx < HW) {
            const size_t e = fe0 + (size_t)idx * 3 + c;
            const double tol = (double)orig[e] * qp.b0;
            const double df = (double)q_delta<FP>(src, orig, e);
            du = df + tol;
            dl = df - tol;
        }
        double tu[6], tl[6];
        tu[0] = du;
        tl[0] = dl;
#pragma unroll
        for (int k = 1; k < 6; ++k) {
            double a = __shfl_down(tu[k - 1], 1 << (k - 1), 64), b = __shfl_down(tl[k - 1], 1 << (k - 1), 64);
            tu[k] = tu[k - 1] < a ? tu[k - 1] : a;
            tl[k] = tl[k - 1] > b ? tl[k - 1] : b;
        }
        double pu = du, pl = dl;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            double a = __shfl_up(pu, sft, 64), b = __shfl_up(pl, sft, 64);
            if (lane >= sft) {
                pu = pu < a ? pu : a;
                pl = pl > b ? pl : b;
            }
        }
        double cu = du, cl = dl;
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            const int step = 1 << k;
            int srcl = pos < 63 ? pos : 63;
            double xu = shfl_d(tu[k], srcl), xl = shfl_d(tl[k], srcl);
            double nu = cu < xu ? cu : xu, nl = cl > xl ? cl : xl;
            bool ok = (pos + step <= 64) && !(nu - nl < 0.0);
            if (ok) {
                cu = nu;
                cl = nl;
                pos += step;
            }
        }
        out.cu = cu;
        out.cl = cl;
        out.pu = pu;
        out.pl = pl;
    }
    // 2b. pointer doubling: the set of run heads reached from a start at this lane.  (Following the chain
    // of ONE start with scalar instructions in the resolver instead -- v_readlane of nxt, one hop per run --
    // was tried in round 3: 17 instead of 35 shuffles per chunk, but the hops sit on the serial path and
    // noise-like data has up to 64 runs per chunk: `rel 1e-3` went from 0.9 to 1.8 ms per step, `abs 2`
    // gained nothing.)
    int J = pos;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int srcl = J < 64 ? J : lane;
        unsigned long long Mj = __shfl(M, srcl, 64);
        int Jj = __shfl(J, srcl, 64);
        if (J < 64) {
            M |= Mj;
            J = Jj;
        }
    }
    out.heads = M;
    out.nxt = pos;  // in [lane+1, 64]; 64 = the run leaves the chunk
    return out;
}

__device__ __forceinline__ double q_rl_d(double v, int src) {  // src is wave-uniform
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ unsigned long long q_rl_u64(unsigned long long v, int src) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
}

// Speculative walk of segment `seg` of every chain (blockIdx.x = (frame * 3 + channel) * QSEG + seg):
// fresh start at the segment's first element.  spec[chain][chunk] receives the heads the walk put into
// each chunk, seg_out[chain][seg] the run left open at the segment end.
template <bool FP>
__global__ __launch_bounds__(64 * (QW + 1)) void k_q_heads(QSrc src, const uint8_t* __restrict__ orig,
                                                         const uint8_t* __restrict__ skip, int HW, QParams qp,
                                                         const double* __restrict__ Echain, int16_t* __restrict__ tmp,
                                                         unsigned long long* __restrict__ spec, QState* __restrict__ seg_out) {
    const int chain = blockIdx.x / QSEG, seg = blockIdx.x % QSEG;
    const int f = chain / 3, c = chain % 3;
    if (skip[f]) return;
    __shared__ QSlot slots[2][QW];
    const size_t fe0 = (size_t)f * HW * 3;
    int16_t* t = tmp + fe0;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double E = 0.0;
    if (qp.mode == TZ_MODE_ABS) E = fabs(qp.b0);
    else if (qp.mode != TZ_MODE_PWREL) E = Echain[chain];
    const double inf = __builtin_huge_val();
    const int nch = (HW + 63) >> 6;
    const int cps = (nch + QSEG - 1) / QSEG;          // chunks per segment
    const int c0 = seg * cps, c1 = min(nch, c0 + cps);
    unsigned long long* sp = spec + (size_t)chain * nch;
    if (c0 >= c1) {                                    // empty segment (short chain): passes the state on untouched
        if (threadIdx.x == 0) seg_out[chain * QSEG + seg] = QState{inf, -inf, -1, 0};
        return;
    }
    const int nrounds = (c1 - c0 + QW - 1) / QW;
    double u = inf, l = -inf;      // resolver: state of the run that is open at the chunk boundary
    int chead = c0 * 64;           // resolver: its head (chain index): the fresh start
    for (int r = 0; r <= nrounds; ++r) {
        if (wv > 0 && r < nrounds) {
            const int ch = c0 + r * QW + (wv - 1);
            if (ch < c1) {
                const QChunk q = qp.mode == TZ_MODE_PWREL ? q_chunk<true, FP>(src, orig, fe0, c, ch, HW, qp, E, lane)
                                                          : q_chunk<false, FP>(src, orig, fe0, c, ch, HW, qp, E, lane);
                QSlot& sl = slots[r & 1][wv - 1];
                sl.cu[lane] = q.cu;
                sl.cl[lane] = q.cl;
                sl.pu[lane] = q.pu;
                sl.pl[lane] = q.pl;
                sl.nxt[lane] = q.nxt;
                sl.heads[lane] = q.heads;
            }
        } else if (wv == 0 && r > 0) {
            // The walk over the chunks is the serial critical path, so nothing that does not depend on
            // the carried (u, l) may sit on it: every LDS operand of chunk w+1 is loaded while chunk w
            // resolves, and the lookups at the wave-uniform positions j0 / last are register reads
            // (v_readlane), not LDS permutes.
            struct Pre {
                double pu, pl, cu, cl;
                unsigned long long hd;
                int nx;
            };
            auto load = [&](int w) {
                const QSlot& sl = slots[(r - 1) & 1][w];
                return Pre{sl.pu[lane], sl.pl[lane], sl.cu[lane], sl.cl[lane], sl.heads[lane], sl.nxt[lane]};
            };
            Pre cur = load(0), nx = cur;
            for (int w = 0; w < QW; ++w) {
                const int ch = c0 + (r - 1) * QW + w;
                if (ch >= c1) break;
                if (w + 1 < QW && ch + 1 < c1) nx = load(w + 1);
                // 4. where does the carried run break?
                const double eu = u < cur.pu ? u : cur.pu, el = l > cur.pl ? l : cur.pl;
                const unsigned long long brk = __ballot(eu - el < 0.0);
                const unsigned long long first = ch == c0 ? 1ull : 0ull;   // the fresh start is a head of the guess
                if (brk == 0ull) {
                    u = q_rl_d(eu, 63);
                    l = q_rl_d(el, 63);
                    if (lane == 0) sp[ch] = first;
                    cur = nx;
                    continue;
                }
                const int j0 = __ffsll((long long)brk) - 1;
                {
                    double uc = u, lc = l;
                    if (j0 > 0) {
                        uc = q_rl_d(eu, j0 - 1);
                        lc = q_rl_d(el, j0 - 1);
                    }
                    if (lane == 0) t[(size_t)chead * 3 + c] = (int16_t)(long long)((uc + lc) / 2);
                }
                const unsigned long long heads = q_rl_u64(cur.hd, j0);
                const int last = 63 - __clzll((long long)heads);
                // 5. closed runs store their value at their head; the last head carries on
                if (((heads >> lane) & 1ull) && cur.nx < 64)
                    t[(size_t)(ch * 64 + lane) * 3 + c] = (int16_t)(long long)((cur.cu + cur.cl) / 2);
                if (lane == 0) sp[ch] = heads | first;
                u = q_rl_d(cur.cu, last);
                l = q_rl_d(cur.cl, last);
                chead = ch * 64 + last;
                cur = nx;
            }
        }
        __syncthreads();
    }
    if (wv == 0 && lane == 0) seg_out[chain * QSEG + seg] = QState{u, l, chead, 0};
}

// Exact repair of the segment guesses, one wave per chain, segments in order.  `st` is the TRUE open run
// entering segment k.  It is walked through the segment's chunks (tables recomputed by this wave) until
// the true chain shares a head with the guessed one (mask spec[chunk]): guessed heads before that point
// leave the mask, true ones enter it (with their values in tmp); from the common head on the guess is
// right, so the true state leaving the segment is the guessed one.  A segment without any common head is
// walked to its end (worst case: the serial walk of round 1).  Finally the run open at the chain end is
// closed.
template <bool FP>
__global__ __launch_bounds__(64) void k_q_stitch(QSrc src, const uint8_t* __restrict__ orig,
                                               const uint8_t* __restrict__ skip, int HW, QParams qp,
                                               const double* __restrict__ Echain, int16_t* __restrict__ tmp,
                                               unsigned long long* __restrict__ spec, const QState* __restrict__ seg_out) {
    const int chain = blockIdx.x, f = chain / 3, c = chain % 3;
    if (skip[f] || HW <= 0) return;
    const size_t fe0 = (size_t)f * HW * 3;
    int16_t* t = tmp + fe0;
    const int lane = threadIdx.x;
    double E = 0.0;
    if (qp.mode == TZ_MODE_ABS) E = fabs(qp.b0);
    else if (qp.mode != TZ_MODE_PWREL) E = Echain[chain];
    const int nch = (HW + 63) >> 6;
    const int cps = (nch + QSEG - 1) / QSEG;
    unsigned long long* sp = spec + (size_t)chain * nch;
    QState st = seg_out[chain * QSEG];          // segment 0 starts where the chain starts: its guess is the truth
    double u = st.u, l = st.l;
    int chead = st.head;
    for (int k = 1; k < QSEG; ++k) {
        const int c0 = k * cps, c1 = min(nch, c0 + cps);
        if (c0 >= c1) break;
        bool merged = false;
        for (int ch = c0; ch < c1 && !merged; ++ch) {
            const QChunk q = qp.mode == TZ_MODE_PWREL ? q_chunk<true, FP>(src, orig, fe0, c, ch, HW, qp, E, lane)
                                                      : q_chunk<false, FP>(src, orig, fe0, c, ch, HW, qp, E, lane);
            const unsigned long long S = sp[ch];
            const double eu = u < q.pu ? u : q.pu, el = l > q.pl ? l : q.pl;
            const unsigned long long brk = __ballot(eu - el < 0.0);
            if (brk == 0ull) {                  // the true run swallows the chunk: every guessed head in it is wrong
                if (lane == 0) sp[ch] = 0ull;
                u = q_rl_d(eu, 63);
                l = q_rl_d(el, 63);
                continue;
            }
            const int j0 = __ffsll((long long)brk) - 1;
            double uc = u, lc = l;
            if (j0 > 0) {
                uc = q_rl_d(eu, j0 - 1);
                lc = q_rl_d(el, j0 - 1);
            }
            const unsigned long long T = q_rl_u64(q.heads, j0);          // true heads of this chunk
            const unsigned long long common = T & S;
            const int m = common ? __ffsll((long long)common) - 1 : 64;  // first common head
            const unsigned long long below = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
            const bool mine = (below >> lane) & 1ull;
            if (mine && ((T >> lane) & 1ull) && q.nxt < 64)
                t[(size_t)(ch * 64 + lane) * 3 + c] = (int16_t)(long long)((q.cu + q.cl) / 2);
            if (lane == 0) {
                sp[ch] = (S & ~below) | (T & below);
                // the carried run closes at j0: its head keeps its mask bit and gets its value
                t[(size_t)chead * 3 + c] = (int16_t)(long long)((uc + lc) / 2);
            }
            if (common) {
                merged = true;
            } else {
                const int last = 63 - __clzll((long long)T);
                u = q_rl_d(q.cu, last);
                l = q_rl_d(q.cl, last);
                chead = ch * 64 + last;
            }
        }
        if (merged) {
            st = seg_out[chain * QSEG + k];
            u = st.u;
            l = st.l;
            chead = st.head;
        }
    }
    if (lane == 0) t[(size_t)chead * 3 + c] = (int16_t)(long long)((u + l) / 2);
}

static constexpr int QFB = 2048;  // pixels per fill item (32 chunks of 64: a lane per mask word when a wave looks back)

// value of the last run head in each fill item (from the masks: one thread per (frame, item, channel))
__global__ __launch_bounds__(256) void k_q_last(const int16_t* __restrict__ tmp, const unsigned long long* __restrict__ spec,
                                                const uint8_t* __restrict__ skip, int HW, int nch, int nblk, int nframes,
                                                int16_t* __restrict__ carry) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nframes * nblk * 3) return;
    const int c = id % 3, b = (id / 3) % nblk, f = id / (3 * nblk);
    if (skip[f]) return;
    const unsigned long long* sp = spec + (size_t)(f * 3 + c) * nch;
    const int w0 = b * (QFB / 64), w1 = min(nch, w0 + QFB / 64);
    int16_t v = TZ_SENTINEL;
    for (int w = w1 - 1; w >= w0; --w) {
        const unsigned long long m = sp[w];
        if (m) {
            const int pos = w * 64 + 63 - __clzll((long long)m);
            v = tmp[((size_t)f * HW + pos) * 3 + c];
            break;
        }
    }
    carry[((size_t)f * nblk + b) * 3 + c] = v;
}

// exclusive forward fill over the items of a chain, one wave per chain (64 items per pass; the last value seen
// carries from pass to pass); ftail = value of the chain's last run (its last element)
__global__ __launch_bounds__(64) void k_q_carry(int16_t* __restrict__ carry, const uint8_t* __restrict__ skip, int nblk, int nframes,
                                                int16_t* __restrict__ ftail) {
    const int id = blockIdx.x, lane = threadIdx.x;
    if (id >= nframes * 3) return;
    const int f = id / 3, c = id % 3;
    if (skip[f]) return;
    int run = (int)TZ_SENTINEL;
    for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const size_t k = ((size_t)f * nblk + b) * 3 + c;
        const int v = b < nblk ? (int)carry[k] : (int)TZ_SENTINEL;
        const unsigned long long has = __ballot(v != (int)TZ_SENTINEL);
        const unsigned long long below = has & ((1ull << lane) - 1ull);       // items in front of mine that hold a head
        const int src = below ? 63 - __clzll((long long)below) : lane;
        const int got = __shfl(v, src, 64);
        if (b < nblk) carry[k] = (int16_t)(below ? got : run);                 // exclusive: last head value strictly before this item
        if (has) run = __shfl(v, 63 - __clzll((long long)has), 64);
    }
    if (lane == 0) ftail[id] = (int16_t)run;
}

// Forward fill of the run values, one wave per 64-element chunk, no communication between waves: the masks
// say where the run heads are, so a lane finds the head of ITS run with bit operations (highest mask bit at or
// below its lane; else the highest bit of the nearest non-empty mask word in front of the chunk within its
// item of 2048 pixels; else the item's carry value from k_q_last / k_q_carry) and fetches the value there --
// rounds 1-2 propagated the values themselves with 18 shuffles and two barriers per 256 pixels.
//  SYM == false: the filled values go back into the delta stack (frames flagged in `skip` are not touched).
//  SYM == true : fused encode.  Every frame is processed: a skipped frame's deltas are trunc(pred*255) - orig
//                (0 where zero_mask says so: compress.py:314); the values go on through the spatial delta over
//                the whole flattened stack (compress.py:73-77: the element in front of a pixel's first channel
//                is the previous pixel's last channel, across chunk, item and frame boundaries), the 1600 offset
//                and the histogram (compress.py:346-355), and leave as symbols in 16-byte stores (a wave's 192
//                symbols pass through 384 bytes of its own LDS).
struct QFill {
    const int16_t* tmp;
    const unsigned long long* spec;
    const uint8_t *skip, *zero;
    const int16_t *carry, *ftail;
    const float* pred;
    const uint8_t* orig;
    int16_t* out;           // delta stack (in place) or symbols
    unsigned long long* hist;
    int16_t* edge;
    int HW, nch, nblk, nframes, apply_offset;
};

static constexpr int QF_THREADS = 512, QF_WAVES = QF_THREADS / 64;

// what a wave needs of one chunk before it can fetch run values: loaded one chunk AHEAD of its use, so that the
// mask / carry round trip of chunk k+1 overlaps the value gather and the stores of chunk k
struct QFillPre {
    unsigned long long mw[3], m[3];
    int cv[3];
    int skipped;
};

__device__ __forceinline__ QFillPre q_fill_pre(const QFill& a, int f, int ch, int lane) {
    QFillPre r;
    const int nch = a.nch;
    const int b = (ch * 64) / QFB, w0 = b * (QFB / 64);
    r.skipped = a.skip[f];
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // (a skipped frame's entries are never written: loaded all the same, not used)
        const unsigned long long* sp = a.spec + (size_t)(f * 3 + c) * nch;
        const int wi = w0 + lane;
        r.mw[c] = (lane < QFB / 64 && wi < ch) ? sp[wi] : 0ull;   // the item's mask words in front of the chunk, one per lane
        r.m[c] = sp[ch];
        r.cv[c] = (int)a.carry[((size_t)f * a.nblk + b) * 3 + c];  // value of the last head in front of the item
    }
    return r;
}

// first half of a work item: where the run values are, and the loads that fetch them (issued, not yet used)
struct QFillMid {
    int hp[3], before2;
    int16_t tv[3], tp;
    int res[3], prev2;   // a skipped frame's raw deltas (fused encode)
    bool skipped, valid;
};

template <bool SYM>
__device__ __forceinline__ QFillMid q_fill_a(const QFill& a, const QFillPre& cur, int f, int ch, int lane, bool valid) {
    QFillMid m{};
    m.valid = valid;
    m.skipped = cur.skipped != 0;
    if (!valid || (!SYM && m.skipped)) return m;
    const int HW = a.HW;
    const size_t fe0 = (size_t)f * HW * 3;
    const int16_t* t = a.tmp + fe0;
    const int p = ch * 64 + lane;
    const int w0 = ((ch * 64) / QFB) * (QFB / 64);  // first mask word of the item this chunk belongs to
    if (SYM && m.skipped) {
        const bool zero = a.zero[f] != 0, live = p < HW;
#pragma unroll
        for (int c = 0; c < 3; ++c) m.res[c] = (live && !zero) ? (int)(a.pred[fe0 + (size_t)p * 3 + c] * 255.0f) - (int)a.orig[fe0 + (size_t)p * 3 + c] : 0;
        if (ch > 0) m.prev2 = zero ? 0 : (int)(a.pred[fe0 + (size_t)ch * 192 - 1] * 255.0f) - (int)a.orig[fe0 + (size_t)ch * 192 - 1];
        return m;
    }
    m.before2 = -1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const unsigned long long mw = cur.mw[c];
        const unsigned long long nz = __ballot(mw != 0ull);
        // position of the last head in front of the chunk (within the item), or -1
        int before = -1;
        if (nz) {
            const int wl = 63 - __clzll((long long)nz);
            const unsigned long long mm = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mw >> 32), wl) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mw, wl);
            before = (w0 + wl) * 64 + 63 - __clzll((long long)mm);
        }
        const unsigned long long mine = cur.m[c] & ((2ull << lane) - 1ull);
        m.hp[c] = mine ? ch * 64 + 63 - __clzll((long long)mine) : before;
        if (c == 2) m.before2 = before;
    }
    // the gathers of the three channels (and of the pixel in front) go out together
#pragma unroll
    for (int c = 0; c < 3; ++c) m.tv[c] = m.hp[c] >= 0 ? t[(size_t)m.hp[c] * 3 + c] : (int16_t)0;
    m.tp = (SYM && m.before2 >= 0) ? t[(size_t)m.before2 * 3 + 2] : (int16_t)0;
    return m;
}

template <bool SYM, bool HIST>
__device__ __forceinline__ void q_fill_b(const QFill& a, const QFillPre& cur, const QFillMid& m, int f, int ch, int lane, int wv,
                                         uint4* stage, HistLds& hl, HistAcc& acc, int centre) {
    if (!m.valid || (!SYM && m.skipped)) return;
    const int HW = a.HW;
    const size_t fe0 = (size_t)f * HW * 3;
    const int p = ch * 64 + lane;
    int res[3], prev2;
    if (SYM && m.skipped) {
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] = m.res[c];
        prev2 = m.prev2;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] = m.hp[c] >= 0 ? (int)m.tv[c] : cur.cv[c];
        prev2 = m.before2 >= 0 ? (int)m.tp : cur.cv[2];
    }
    if (!SYM) {
        if (p < HW)
#pragma unroll
            for (int c = 0; c < 3; ++c) a.out[fe0 + (size_t)p * 3 + c] = (int16_t)res[c];
        return;
    }
    // the element in front of the chunk's first sample when the chunk starts a frame
    bool have_prev = true;
    if (ch == 0) {
        if (f == 0) have_prev = false;              // start of the stream: sd[0] = x[0]
        else if (a.skip[f - 1]) prev2 = a.zero[f - 1] ? 0 : (int)(a.pred[fe0 - 1] * 255.0f) - (int)a.orig[fe0 - 1];
        else prev2 = (int)a.ftail[(f - 1) * 3 + 2];
    }
    int prev = __shfl_up(res[2], 1);
    if (lane == 0) prev = prev2;
    int sd0 = prev - res[0];
    if (!have_prev && lane == 0) sd0 = res[0];
    const int sd1 = res[0] - res[1], sd2 = res[1] - res[2];
    const short y0 = a.apply_offset ? (short)(TZ_OFFSET - sd0) : (short)sd0;
    const short y1 = a.apply_offset ? (short)(TZ_OFFSET - sd1) : (short)sd1;
    const short y2 = a.apply_offset ? (short)(TZ_OFFSET - sd2) : (short)sd2;
    if (f == 0 && p == 0) a.edge[0] = (int16_t)res[0];
    if (f == a.nframes - 1 && p == HW - 1) a.edge[1] = (int16_t)res[2];
    short* so = (short*)(stage + wv * 24);
    so[lane * 3 + 0] = y0;
    so[lane * 3 + 1] = y1;
    so[lane * 3 + 2] = y2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int npix = min(64, HW - ch * 64);           // HW % 8 == 0: whole 16-byte vectors
    if (lane < npix * 3 / 8) {
        const uint4 y = stage[wv * 24 + lane];
        ((uint4*)(a.out + fe0 + (size_t)ch * 192))[lane] = y;
        if (HIST) {
            const unsigned Y[4] = {y.x, y.y, y.z, y.w};
            hist_add8(hl, acc, Y, centre);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The mask / carry words of a wave's NEXT work item are loaded one iteration ahead.  (The kernel is bound by the
// vector instructions a 64-pixel item costs, about 300 for its 192 symbols, not by memory: two items in flight per
// wave made it slower, 172 -> 211 us at cfg3.)
template <bool SYM, bool HIST>
__global__ __launch_bounds__(QF_THREADS) void k_q_fill(const QFill a) {
    __shared__ unsigned hraw[HIST ? HL_WORDS : 1];
    __shared__ uint4 stage[SYM ? QF_WAVES * 24 : 1];   // 64 pixels x 3 symbols x 2 bytes per wave
    HistLds& hl = *(HistLds*)hraw;
    HistAcc acc;
    const int centre = a.apply_offset ? TZ_OFFSET : 0;
    if (HIST) hist_clear(hl);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nch = a.nch;
    // work item = (frame, chunk of 64 pixels, all three channels); a wave takes every (gridDim * waves)-th of them,
    // (f, ch) advanced without divisions
    const int wstride = (int)gridDim.x * QF_WAVES, df = wstride / nch, dch = wstride % nch;
    const int w_first = (int)blockIdx.x * QF_WAVES + wv;
    int f = w_first / nch, ch = w_first % nch;
    QFillPre nxt{};
    if (f < a.nframes) nxt = q_fill_pre(a, f, ch, lane);
    while (f < a.nframes) {
        const QFillPre cur = nxt;
        const int fc = f, chc = ch;
        f += df;
        ch += dch;
        if (ch >= nch) {
            ch -= nch;
            ++f;
        }
        if (f < a.nframes) nxt = q_fill_pre(a, f, ch, lane);
        const QFillMid m = q_fill_a<SYM>(a, cur, fc, chc, lane, true);
        q_fill_b<SYM, HIST>(a, cur, m, fc, chc, lane, wv, stage, hl, acc, centre);
    }
    if (HIST) hist_flush(hl, acc, centre, a.hist);
}

struct QFused {   // fused encode: symbols + histogram + edge elements instead of a delta stack
    const float* pred;
    const uint8_t* d_zero;
    int apply_offset;
    int16_t* sym;
    unsigned long long* d_hist;
    int16_t* d_edge;
};

static int quant_run(tz_ctx* ctx, const uint8_t* orig, int16_t* diff, const QFused* fu, const uint8_t* h_skip, int nframes,
                     int H, int W, int mode, double b0, double b1) {
    int HW = H * W;
    size_t fe = (size_t)HW * 3;
    int nblk = (HW + QFB - 1) / QFB;
    void *d_skip, *d_E, *d_tmp, *d_carry, *d_mm, *d_spec, *d_seg, *d_ftail;
    TZ_TRY(tz_pool_alloc(ctx, nframes, &d_skip));
    TZ_TRY(tz_pool_alloc(ctx, sizeof(double) * 3 * nframes, &d_E));
    TZ_TRY(tz_pool_alloc(ctx, sizeof(int) * 6 * nframes, &d_mm));
    TZ_TRY(tz_pool_alloc(ctx, fe * nframes * 2, &d_tmp));
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * nblk * 3 * 2, &d_carry));
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * 3 * 2, &d_ftail));
    const int nch = (HW + 63) / 64;
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * 3 * nch * sizeof(unsigned long long), &d_spec));
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * 3 * QSEG * sizeof(QState), &d_seg));
    TZ_TRY(tz_upload(ctx, d_skip, h_skip, nframes));
    QParams qp{mode, b0, b1};
    QSrc src{diff, fu ? fu->pred : nullptr};
    {
        tz_prof_scope ps(ctx, TZP_QUANT);
        if (mode == TZ_MODE_REL || mode == TZ_MODE_ABSREL) {
            hipLaunchKernelGGL(k_q_mm_init, dim3((6 * nframes + 255) / 256), dim3(256), 0, ctx->stream, (int*)d_mm, 6 * nframes);
            hipLaunchKernelGGL(k_q_minmax, dim3(QBB, nframes), dim3(256), 0, ctx->stream, orig, (const uint8_t*)d_skip, HW,
                               (int*)d_mm);
            hipLaunchKernelGGL(k_q_bound, dim3((3 * nframes + 63) / 64), dim3(64), 0, ctx->stream, (const int*)d_mm,
                               (const uint8_t*)d_skip, qp, nframes, (double*)d_E);
        }
        if (fu) {
            hipLaunchKernelGGL(k_q_heads<true>, dim3(nframes * 3 * QSEG), dim3(64 * (QW + 1)), 0, ctx->stream, src, orig,
                               (const uint8_t*)d_skip, HW, qp, (const double*)d_E, (int16_t*)d_tmp, (unsigned long long*)d_spec,
                               (QState*)d_seg);
            hipLaunchKernelGGL(k_q_stitch<true>, dim3(nframes * 3), dim3(64), 0, ctx->stream, src, orig, (const uint8_t*)d_skip, HW,
                               qp, (const double*)d_E, (int16_t*)d_tmp, (unsigned long long*)d_spec, (const QState*)d_seg);
        } else {
            hipLaunchKernelGGL(k_q_heads<false>, dim3(nframes * 3 * QSEG), dim3(64 * (QW + 1)), 0, ctx->stream, src, orig,
                               (const uint8_t*)d_skip, HW, qp, (const double*)d_E, (int16_t*)d_tmp, (unsigned long long*)d_spec,
                               (QState*)d_seg);
            hipLaunchKernelGGL(k_q_stitch<false>, dim3(nframes * 3), dim3(64), 0, ctx->stream, src, orig, (const uint8_t*)d_skip, HW,
                               qp, (const double*)d_E, (int16_t*)d_tmp, (unsigned long long*)d_spec, (const QState*)d_seg);
        }
        hipLaunchKernelGGL(k_q_last, dim3((nframes * nblk * 3 + 255) / 256), dim3(256), 0, ctx->stream, (const int16_t*)d_tmp,
                           (const unsigned long long*)d_spec, (const uint8_t*)d_skip, HW, nch, nblk, nframes, (int16_t*)d_carry);
        hipLaunchKernelGGL(k_q_carry, dim3(nframes * 3), dim3(64), 0, ctx->stream, (int16_t*)d_carry,
                           (const uint8_t*)d_skip, nblk, nframes, (int16_t*)d_ftail);
    }
    QFill a{};
    a.tmp = (const int16_t*)d_tmp;
    a.spec = (const unsigned long long*)d_spec;
    a.skip = (const uint8_t*)d_skip;
    a.carry = (const int16_t*)d_carry;
    a.ftail = (const int16_t*)d_ftail;
    a.HW = HW;
    a.nch = nch;
    a.nblk = nblk;
    a.nframes = nframes;
    const long long nwork = (long long)nframes * nch;
    const int grid = (int)std::min<long long>(4 * kCUs, (nwork + QF_WAVES - 1) / QF_WAVES);
    if (!fu) {
        tz_prof_scope ps(ctx, TZP_QUANT);
        a.out = diff;
        hipLaunchKernelGGL((k_q_fill<false, false>), dim3(grid), dim3(QF_THREADS), 0, ctx->stream, a);
    } else {
        tz_prof_scope ps(ctx, TZP_SDELTA);
        a.zero = fu->d_zero;
        a.pred = fu->pred;
        a.orig = orig;
        a.out = fu->sym;
        a.hist = fu->d_hist;
        a.edge = fu->d_edge;
        a.apply_offset = fu->apply_offset;
        if (fu->d_hist) hipLaunchKernelGGL((k_q_fill<true, true>), dim3(grid), dim3(QF_THREADS), 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_q_fill<true, false>), dim3(grid), dim3(QF_THREADS), 0, ctx->stream, a);
    }
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

int tzk_error_bound(tz_ctx* ctx, const uint8_t* orig, int16_t* diff, const uint8_t* h_skip, int nframes, int H,
                    int W, int mode, double b0, double b1) {
    if (mode < 0 || mode > 3) return tz_fail(ctx, TZ_ERR_INVALID, "unknown error-bound mode %d", mode);
    if (b0 == 0.0) return TZ_OK;                          // compress.py:24
    if (mode == TZ_MODE_ABSREL && b1 == 0.0) return TZ_OK;  // compress.py:35
    if (mode == TZ_MODE_PWREL && b0 < 0.0)
        return tz_fail(ctx, TZ_ERR_INVALID, "pwrel bound must be >= 0 (the reference raises on a negative one)");
    if (nframes <= 0 || H <= 0 || W <= 0) return TZ_OK;
    return quant_run(ctx, orig, diff, nullptr, h_skip, nframes, H, W, mode, b0, b1);
}

// Lossy fused encode (compress.py:292-355 without a delta stack in memory): applies to unpadded frames with
// whole 16-byte groups per frame; *done says whether it ran (else the caller takes the unfused kernels).
int tzk_quant_sd_fused(tz_ctx* ctx, const float* pred, const uint8_t* orig, const uint8_t* d_zero_mask, const uint8_t* h_skip,
                       int nframes, int H, int W, int Hp, int Wp, int mode, double b0, double b1, int apply_offset,
                       int16_t* sym, unsigned long long* d_hist, int16_t* d_edge, bool* done) {
    *done = false;
    if (mode < 0 || mode > 3) return tz_fail(ctx, TZ_ERR_INVALID, "unknown error-bound mode %d", mode);
    if (mode == TZ_MODE_PWREL && b0 < 0.0)
        return tz_fail(ctx, TZ_ERR_INVALID, "pwrel bound must be >= 0 (the reference raises on a negative one)");
    if (H != Hp || W != Wp || ((size_t)H * W) % 8 || nframes <= 0 || (((uintptr_t)sym) & 15)) return TZ_OK;
    QFused fu{pred, d_zero_mask, apply_offset, sym, d_hist, d_edge};
    TZ_TRY(quant_run(ctx, orig, nullptr, &fu, h_skip, nframes, H, W, mode, b0, b1));
    *done = true;
    return TZ_OK;
}

// ------------------------------------------------------- spatial delta (+offset, histogram)
// compress.py:73-77: out[0]=in[0], out[i]=in[i-1]-in[i] over the whole flat array (int16
// wrap); compress.py:348: y = 1600 - sd; compress.py:354: bincount(y).
// 8 elements (16 B) per lane; the element before a lane's first comes from one extra 2-byte
// load (same cache line).  Histogram: HistLds / HistAcc above.
template <bool HIST>
__global__ __launch_bounds__(HIST ? HB_THREADS : 256) void k_sdelta(const int16_t* __restrict__ in, size_t n, int has_carry, int16_t carry,
                                                int apply_offset, int16_t* __restrict__ out,
                                                unsigned long long* __restrict__ hist) {
    __shared__ unsigned hraw[HIST ? HL_WORDS : 1];
    HistLds& hl = *(HistLds*)hraw;
    if (HIST) hist_clear(hl);
    const int centre = apply_offset ? TZ_OFFSET : 0;
    HistAcc acc;
    size_t n8 = n / 8;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4* in8 = (const uint4*)in;
    uint4* out8 = (uint4*)out;
    const unsigned short* inu = (const unsigned short*)in;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const uint4 v = in8[i];
        // the element in front; without one (start of the stream, no carry) sd[0] = x[0], i.e. "prev" = 2 x[0]
        unsigned prev = i ? (unsigned)inu[8 * i - 1] : (has_carry ? (unsigned)(unsigned short)carry : ((v.x << 1) & 0xFFFFu));
        unsigned Y[4];
        sdelta8(v, prev, apply_offset != 0, Y);
        out8[i] = make_uint4(Y[0], Y[1], Y[2], Y[3]);
        if (HIST) hist_add8(hl, acc, Y, centre);
    }
    // tail (n % 8 elements) by the first lanes of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        size_t i = n8 * 8 + threadIdx.x;
        short cur = in[i];
        short sd = i ? (short)(in[i - 1] - cur) : (has_carry ? (short)(carry - cur) : cur);
        short y = apply_offset ? (short)(TZ_OFFSET - sd) : sd;
        out[i] = y;
        if (HIST && y >= 0 && y < TZ_NBINS) atomicAdd(&hl.w[HL_FULL + y], 1u);
    }
    if (HIST) hist_flush(hl, acc, centre, hist);
}

// Lossless fast path (error_bound is the identity, compress.py:24): delta, spatial delta,
// 1600 offset and histogram in ONE pass over pred/orig -- 7 B/element instead of 7 + 4.
// Same arithmetic as k_delta_flat followed by k_sdelta; the element before a lane's first is
// recomputed from pred/orig (one extra float + byte, same cache lines).
template <bool HIST>
__global__ __launch_bounds__(HIST ? HB_THREADS : 256) void k_delta_sd_fused(const float4* __restrict__ pred, const uint2* __restrict__ orig,
                                                        const uint8_t* __restrict__ zero_mask, size_t n8,
                                                        unsigned frame_elems8, int apply_offset,
                                                        short8* __restrict__ out, unsigned long long* __restrict__ hist,
                                                        int16_t* __restrict__ edge) {
    __shared__ unsigned hraw[HIST ? HL_WORDS : 1];
    HistLds& hl = *(HistLds*)hraw;
    constexpr bool do_hist = HIST;
    if (do_hist) hist_clear(hl);
    const int centre = apply_offset ? TZ_OFFSET : 0;
    HistAcc acc;
    const float* predf = (const float*)pred;
    const uint8_t* origb = (const uint8_t*)orig;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float4 p0 = pred[2 * i], p1 = pred[2 * i + 1];
        uint2 o = orig[i];
        const bool zero = zero_mask[i / frame_elems8] != 0;
        short d[8];
        d[0] = (short)((int)(p0.x * 255.0f) - (int)(o.x & 0xff));
        d[1] = (short)((int)(p0.y * 255.0f) - (int)((o.x >> 8) & 0xff));
        d[2] = (short)((int)(p0.z * 255.0f) - (int)((o.x >> 16) & 0xff));
        d[3] = (short)((int)(p0.w * 255.0f) - (int)(o.x >> 24));
        d[4] = (short)((int)(p1.x * 255.0f) - (int)(o.y & 0xff));
        d[5] = (short)((int)(p1.y * 255.0f) - (int)((o.y >> 8) & 0xff));
        d[6] = (short)((int)(p1.z * 255.0f) - (int)((o.y >> 16) & 0xff));
        d[7] = (short)((int)(p1.w * 255.0f) - (int)(o.y >> 24));
        short prev = 0;
        if (i) {
            size_t e = 8 * i - 1;
            if (!zero_mask[(i * 8 - 1) / ((size_t)frame_elems8 * 8)]) prev = (short)((int)(predf[e] * 255.0f) - (int)origb[e]);
        }
        // pack the (masked) deltas two per dword and take the spatial delta in packed int16 arithmetic
        uint4 V;
        {
            unsigned q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                q[j] = zero ? 0u : (((unsigned)(unsigned short)d[2 * j]) | ((unsigned)(unsigned short)d[2 * j + 1] << 16));
            V = make_uint4(q[0], q[1], q[2], q[3]);
        }
        const unsigned pv = i ? (unsigned)(unsigned short)prev : ((V.x << 1) & 0xFFFFu);   // stream start: sd[0] = x[0]
        unsigned Y[4];
        sdelta8(V, pv, apply_offset != 0, Y);
        ((uint4*)out)[i] = make_uint4(Y[0], Y[1], Y[2], Y[3]);
        if (do_hist) hist_add8(hl, acc, Y, centre);
        // first / last element of the delta stack (shard boundaries, tz_encode_begin)
        if (i == 0) edge[0] = zero ? (short)0 : d[0];
        if (i == n8 - 1) edge[1] = zero ? (short)0 : d[7];
    }
    if (do_hist) hist_flush(hl, acc, centre, hist);
}

// returns TZ_OK and sets *done when the fused path applies (unpadded frames, whole 8-element groups)
int tzk_delta_sd_fused(tz_ctx* ctx, const float* pred, const uint8_t* orig, const uint8_t* d_zero_mask, int nframes, int H,
                       int W, int Hp, int Wp, int apply_offset, int16_t* out, unsigned long long* d_hist, int16_t* d_edge,
                       bool* done) {
    *done = false;
    size_t fe = (size_t)H * W * 3;
    if (H != Hp || W != Wp || fe % 8 || nframes <= 0) return TZ_OK;
    size_t n8 = fe * nframes / 8;
    tz_prof_scope ps(ctx, TZP_DELTA);
    if (d_hist)
        hipLaunchKernelGGL(k_delta_sd_fused<true>, dim3(std::min(HB_GRID, grid_for(n8, HB_THREADS))), dim3(HB_THREADS), 0, ctx->stream,
                           (const float4*)pred, (const uint2*)orig, d_zero_mask, n8, (unsigned)(fe / 8), apply_offset, (short8*)out,
                           d_hist, d_edge);
    else
        hipLaunchKernelGGL(k_delta_sd_fused<false>, dim3(grid_for(n8, 256)), dim3(256), 0, ctx->stream, (const float4*)pred,
                           (const uint2*)orig, d_zero_mask, n8, (unsigned)(fe / 8), apply_offset, (short8*)out, d_hist, d_edge);
    TZ_HIP(ctx, hipGetLastError());
    *done = true;
    return TZ_OK;
}

int tzk_spatial_delta(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry, int apply_offset,
                      int16_t* out, unsigned long long* d_hist) {
    if (n == 0) return TZ_OK;
    if (((uintptr_t)in & 15) || ((uintptr_t)out & 15))
        return tz_fail(ctx, TZ_ERR_INVALID, "spatial_delta buffers must be 16-byte aligned");
    tz_prof_scope ps(ctx, TZP_SDELTA);
    if (d_hist)
        hipLaunchKernelGGL(k_sdelta<true>, dim3(std::min(HB_GRID, grid_for(n / 8 + 1, HB_THREADS))), dim3(HB_THREADS), 0, ctx->stream,
                           in, n, has_carry, carry, apply_offset, out, d_hist);
    else
        hipLaunchKernelGGL(k_sdelta<false>, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, ctx->stream, in, n, has_carry, carry,
                           apply_offset, out, d_hist);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

// ----------------------------------------------------------------------- rank remap / unmap
// compress.py:84-90 and decompress.py:31-36 are T sequential `where` passes (O(N*T)); here
// one pass through a 2112-entry LUT held in LDS.  Values outside [0, 2112) pass through
// (then 1600 - v when post_offset, decompress.py:236).  The host builds the LUT so that it
// reproduces the sequential-pass semantics exactly.
__global__ __launch_bounds__(256) void k_lut(const int16_t* __restrict__ in, size_t n, const int16_t* __restrict__ lut,
                                             int post_offset, int16_t* __restrict__ out) {
    __shared__ int16_t sl[TZ_NBINS + 1];
    for (int k = threadIdx.x; k < TZ_NBINS + 1; k += 256) sl[k] = lut[k];
    __syncthreads();
    size_t n8 = n / 8;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const short8* in8 = (const short8*)in;
    short8* out8 = (short8*)out;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        short8 v = in8[i], r;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int x = v[k];
            r[k] = (x >= 0 && x <= TZ_NBINS) ? sl[x] : (short)(post_offset ? TZ_OFFSET - x : x);
        }
        out8[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        size_t i = n8 * 8 + threadIdx.x;
        int x = in[i];
        out[i] = (x >= 0 && x <= TZ_NBINS) ? sl[x] : (int16_t)(post_offset ? TZ_OFFSET - x : x);
    }
}

int tzk_lut(tz_ctx* ctx, const int16_t* in, size_t n, const int16_t* h_lut2112, int post_offset, int16_t* out) {
    if (n == 0) return TZ_OK;
    if (((uintptr_t)in & 15) || ((uintptr_t)out & 15))
        return tz_fail(ctx, TZ_ERR_INVALID, "remap buffers must be 16-byte aligned");
    void* d_lut;
    TZ_TRY(tz_pool_alloc(ctx, (TZ_NBINS + 1) * 2, &d_lut));
    TZ_TRY(tz_upload(ctx, d_lut, h_lut2112, (TZ_NBINS + 1) * 2));
    tz_prof_scope ps(ctx, TZP_LUT);
    hipLaunchKernelGGL(k_lut, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, ctx->stream, in, n, (const int16_t*)d_lut,
                       post_offset, out);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

// ---------------------------------------------------------------------------- byte shuffle
// Opt-in stage that is NOT in the reference (BASELINE.json's north star names it): the int16 payload
// is stored as two byte planes, all low bytes then all high bytes, before zstd sees it.  Ranks are
// < 1021, so the high plane is almost constant and the low plane loses the interleaved zeros.
// 4 B/element of HBM traffic each way; 8 elements per lane (16 B in, two 8 B stores).
__global__ __launch_bounds__(256) void k_shuffle(const int16_t* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
    const size_t n8 = n / 8, stride = (size_t)gridDim.x * blockDim.x;
    const short8* in8 = (const short8*)in;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const short8 v = in8[i];
        unsigned long long lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned u = (unsigned short)v[k];
            lo |= (unsigned long long)(u & 0xFF) << (8 * k);
            hi |= (unsigned long long)(u >> 8) << (8 * k);
        }
        *(unsigned long long*)(out + 8 * i) = lo;
        *(unsigned long long*)(out + n + 8 * i) = hi;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const size_t i = n8 * 8 + threadIdx.x;
        const unsigned u = (unsigned short)in[i];
        out[i] = (uint8_t)(u & 0xFF);
        out[n + i] = (uint8_t)(u >> 8);
    }
}

__global__ __launch_bounds__(256) void k_unshuffle(const uint8_t* __restrict__ in, size_t n, int16_t* __restrict__ out) {
    const size_t n8 = n / 8, stride = (size_t)gridDim.x * blockDim.x;
    short8* out8 = (short8*)out;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        unsigned long long lo, hi;
        memcpy(&lo, in + 8 * i, 8);       // the high plane starts at n, which need not be 8-aligned
        memcpy(&hi, in + n + 8 * i, 8);
        short8 r;
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = (short)(((lo >> (8 * k)) & 0xFF) | (((hi >> (8 * k)) & 0xFF) << 8));
        out8[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const size_t i = n8 * 8 + threadIdx.x;
        out[i] = (int16_t)((unsigned)in[i] | ((unsigned)in[n + i] << 8));
    }
}

int tzk_shuffle(tz_ctx* ctx, const int16_t* in, size_t n, uint8_t* out, int inverse) {
    if (n == 0) return TZ_OK;
    if (((uintptr_t)in & 15) || ((uintptr_t)out & 15))
        return tz_fail(ctx, TZ_ERR_INVALID, "byte-shuffle buffers must be 16-byte aligned");
    tz_prof_scope ps(ctx, TZP_LUT);
    if (!inverse) {
        if (n & 7) return tz_fail(ctx, TZ_ERR_INVALID, "byte shuffle needs a multiple of 8 elements");
        hipLaunchKernelGGL(k_shuffle, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, ctx->stream, in, n, out);
    } else {
        hipLaunchKernelGGL(k_unshuffle, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)in, n,
                           (int16_t*)out);
    }
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

// ------------------------------------------------------------- inverse spatial delta (scan)
// decompress.py:22-29 is a serial loop x[i] = x[i-1] - s[i] (pure Python, forced onto the
// CPU by the reference, docs/index.rst:1392-1396).  It is the wrap-around prefix scan
//   x[i] = c0 - sum_{j<=i} s'[j]   (mod 2^16),  s'[0] = -s[0] and c0 = 0 without a carry.
// ONE launch of SCAN_G resident blocks (round 3; rounds 1-2: three launches, 85 us per 62.9 M elements):
// block g owns a contiguous chunk of tiles; it (1) sums its chunk, (2) publishes the sum and reads
// the sums of the chunks in front of it -- they all run at the same time and do the same work, so they
// are there within one round trip --, (3) walks its chunk again, scanning tile by tile.  6 B/element
// move for 4 algorithmic, as before, but without the launch gaps and the scan-of-sums kernel in
// between: 60 us = 4.2 TB/s algorithmic, which is what 6 B/element at the chip's copy rate
// (6.3 TB/s) allows.  A single pass with decoupled look-back (4 B/element) was built and measured and
// is NOT faster here: a look-back hop between workgroups on different XCDs takes several microseconds
// under streaming load (and a ticket counter serialises at 12 ns per block): 170-220 us in every
// variant (scripts/microbench/scan_lookback.hip, profiles/r03/scan_lookback.txt).
// With LUT the decoder's inverse rank remap (decompress.py:31-36,236: rank -> 1600 - symbol) is
// applied to the elements as they are loaded (both times), which removes the separate k_lut pass
// over the payload (4 B/element) from tz_decode.
// The blocks of phase 2 wait for lower-numbered blocks only; SCAN_G is far below what the chip holds
// (4 of 8 possible workgroups per CU) and workgroups are dispatched in index order, so every block a
// waiter waits for is running.
static constexpr int SCAN_EPT = 16;                 // elements per thread
static constexpr int SCAN_BLK = 256 * SCAN_EPT;     // elements per tile
static constexpr int SCAN_G = 1024;                 // resident blocks
static constexpr unsigned SCAN_VALID = 1u << 16;

__device__ __forceinline__ unsigned block_scan_excl(unsigned v, unsigned* total) {
    __shared__ unsigned wsum[4];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned inc = v;
    for (int s = 1; s < 64; s <<= 1) {
        unsigned up = __shfl_up(inc, s);
        if (lane >= s) inc += up;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned pre = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wv) pre += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

// 16 consecutive elements of a tile for this thread: two 16-byte accesses (scalar at the ragged tail and
// for unaligned buffers), through the decoder LUT when there is one, the first element of the stream
// negated (s'[0] = -s[0]); elements past the end count as 0
template <bool LUT>
__device__ __forceinline__ void scan_load16(const int16_t* __restrict__ in, size_t base, size_t n, bool vec, bool first_neg,
                                            const int16_t* sl, int post_offset, int* v) {
    if (vec && base + SCAN_EPT <= n) {
        short8 a = *(const short8*)(in + base), b = *(const short8*)(in + base + 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = a[k];
            v[8 + k] = b[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) v[k] = base + k < n ? (int)in[base + k] : 0;
    }
    if (LUT) {
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) {
            const int x = v[k];
            const int y = (x >= 0 && x <= TZ_NBINS) ? (int)sl[x] : (int)(short)(post_offset ? TZ_OFFSET - x : x);
            v[k] = base + k < n ? y : 0;
        }
    }
    if (base == 0 && first_neg) v[0] = -v[0];
}

__device__ __forceinline__ unsigned st_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// status[g] = SCAN_VALID | (sum of chunk g mod 2^16); all zero at launch
template <bool LUT>
__global__ __launch_bounds__(256) void k_scan2p(const int16_t* __restrict__ in, size_t n, int tiles_per_block, int has_carry,
                                                int16_t carry, int vec, const int16_t* __restrict__ lut, int post_offset,
                                                unsigned* __restrict__ status, int16_t* __restrict__ out) {
    __shared__ int16_t sl[LUT ? TZ_NBINS + 1 : 1];
    if (LUT) {
        for (int k = threadIdx.x; k < TZ_NBINS + 1; k += 256) sl[k] = lut[k];
        __syncthreads();
    }
    const int g = blockIdx.x;
    const size_t chunk0 = (size_t)g * tiles_per_block * SCAN_BLK;
    // (1) the chunk's sum
    unsigned s = 0;
    for (int t = 0; t < tiles_per_block; ++t) {
        const size_t base = chunk0 + (size_t)t * SCAN_BLK + (size_t)threadIdx.x * SCAN_EPT;
        if (base >= n) break;
        int v[SCAN_EPT];
        scan_load16<LUT>(in, base, n, vec != 0, !has_carry, sl, post_offset, v);
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) s += (unsigned)v[k];
    }
    unsigned tot;
    block_scan_excl(s, &tot);
    if (threadIdx.x == 0) st_store(&status[g], SCAN_VALID | (tot & 0xFFFFu));
    // (2) the sums of the chunks in front: thread t takes chunks t, t + 256, ...
    unsigned mine = 0;
    for (int b = threadIdx.x; b < g; b += 256) {
        unsigned w = st_load(&status[b]);
        while ((w >> 16) == 0) w = st_load(&status[b]);
        mine += w & 0xFFFFu;
    }
    unsigned run;
    block_scan_excl(mine, &run);
    // (3) scan the chunk tile by tile
    const unsigned c0 = has_carry ? (unsigned)(int)carry : 0u;
    for (int t = 0; t < tiles_per_block; ++t) {
        const size_t base = chunk0 + (size_t)t * SCAN_BLK + (size_t)threadIdx.x * SCAN_EPT;
        if (chunk0 + (size_t)t * SCAN_BLK >= n) break;   // uniform for the block: the barriers below stay matched
        int v[SCAN_EPT];
        scan_load16<LUT>(in, base, n, vec != 0, !has_carry, sl, post_offset, v);
        unsigned q = 0;
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) q += (unsigned)v[k];
        unsigned ttot;
        unsigned pre = block_scan_excl(q, &ttot) + run;
        run += ttot;
        short r[SCAN_EPT];
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) {
            pre += (unsigned)v[k];
            r[k] = (short)(uint16_t)(c0 - pre);
        }
        if (vec && base + SCAN_EPT <= n) {
            short8 a, b;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = r[k];
                b[k] = r[8 + k];
            }
            *(short8*)(out + base) = a;
            *(short8*)(out + base + 8) = b;
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_EPT; ++k)
                if (base + k < n) out[base + k] = r[k];
        }
    }
}

static int scan_launch(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry, const int16_t* h_lut2112,
                       int post_offset, int16_t* out) {
    if (n == 0) return TZ_OK;
    const size_t tiles = (n + SCAN_BLK - 1) / SCAN_BLK;
    const int G = (int)std::min<size_t>(SCAN_G, tiles);
    const size_t tpb = (tiles + G - 1) / G;
    if (tpb > 0x7FFFFFFFull) return tz_fail(ctx, TZ_ERR_INVALID, "inverse scan: too many elements");
    void *d_status, *d_lut = nullptr;
    TZ_TRY(tz_pool_alloc(ctx, sizeof(unsigned) * (size_t)G, &d_status));
    if (h_lut2112) {
        TZ_TRY(tz_pool_alloc(ctx, (TZ_NBINS + 1) * 2, &d_lut));
        TZ_TRY(tz_upload(ctx, d_lut, h_lut2112, (TZ_NBINS + 1) * 2));
    }
    tz_prof_scope ps(ctx, TZP_SCAN);
    TZ_HIP(ctx, hipMemsetAsync(d_status, 0, sizeof(unsigned) * (size_t)G, ctx->stream));
    const int vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (h_lut2112)
        hipLaunchKernelGGL(k_scan2p<true>, dim3(G), dim3(256), 0, ctx->stream, in, n, (int)tpb, has_carry, carry, vec,
                           (const int16_t*)d_lut, post_offset, (unsigned*)d_status, out);
    else
        hipLaunchKernelGGL(k_scan2p<false>, dim3(G), dim3(256), 0, ctx->stream, in, n, (int)tpb, has_carry, carry, vec,
                           (const int16_t*)nullptr, 0, (unsigned*)d_status, out);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

int tzk_undelta(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry, int16_t* out) {
    return scan_launch(ctx, in, n, has_carry, carry, nullptr, 0, out);
}

// decoder: inverse rank remap (+ 1600 - x) and inverse spatial delta in one pass over the payload
int tzk_unmap_undelta(tz_ctx* ctx, const int16_t* in, size_t n, const int16_t* h_lut2112, int post_offset, int16_t* out) {
    return scan_launch(ctx, in, n, 0, 0, h_lut2112, post_offset, out);
}

// ----------------------------------------------------------------------------- reconstruct
// decompress.py:252-256,269: pred*255 - diff, clip [0,255], truncate.  pred*255 in float64
// minus an integer, clipped and truncated equals clamp(trunc(f32(pred*255)) - diff, 0, 255)
// (DESIGN.md §"Why reconstruct is integer"); key slots use the key byte as base.
__global__ __launch_bounds__(256) void k_recon(const float* __restrict__ pred, const uint8_t* __restrict__ key,
                                               const uint8_t* __restrict__ key_mask, const int16_t* __restrict__ diff,
                                               size_t n, int H, int W, int Hp, int Wp, uint8_t* __restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t row = (size_t)W * 3, fe = (size_t)H * row, fp = (size_t)Hp * Wp * 3;
    size_t n4 = n / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4 + (n & 3 ? 1 : 0); q += stride) {
        unsigned packed = 0;
        int cnt = q < n4 ? 4 : (int)(n & 3);
        for (int k = 0; k < cnt; ++k) {
            size_t i = q * 4 + k;
            size_t f = i / fe, r = i - f * fe;
            int base;
            if (key_mask[f]) {
                base = key[i];
            } else {
                size_t y = r / row, xc = r - y * row;
                base = (int)(pred[f * fp + y * (size_t)Wp * 3 + xc] * 255.0f);
            }
            int v = base - (int)diff[i];
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            packed |= (unsigned)v << (8 * k);
        }
        if (cnt == 4) ((unsigned*)out)[q] = packed;
        else
            for (int k = 0; k < cnt; ++k) out[q * 4 + k] = (uint8_t)(packed >> (8 * k));
    }
}

// Fast path: unpadded frames, 8 elements per lane (2x16 B pred, 16 B diff, 8 B key, 8 B out).
__global__ __launch_bounds__(256) void k_recon_flat(const float4* __restrict__ pred, const uint2* __restrict__ key,
                                                    const uint8_t* __restrict__ key_mask, const short8* __restrict__ diff,
                                                    size_t n8, unsigned frame_elems8, uint2* __restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        short8 d = diff[i];
        int base[8];
        if (key_mask[i / frame_elems8]) {
            uint2 k = key[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                base[j] = (k.x >> (8 * j)) & 0xff;
                base[4 + j] = (k.y >> (8 * j)) & 0xff;
            }
        } else {
            float4 p0 = pred[2 * i], p1 = pred[2 * i + 1];
            base[0] = (int)(p0.x * 255.0f); base[1] = (int)(p0.y * 255.0f);
            base[2] = (int)(p0.z * 255.0f); base[3] = (int)(p0.w * 255.0f);
            base[4] = (int)(p1.x * 255.0f); base[5] = (int)(p1.y * 255.0f);
            base[6] = (int)(p1.z * 255.0f); base[7] = (int)(p1.w * 255.0f);
        }
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int a = base[j] - (int)d[j], b = base[4 + j] - (int)d[4 + j];
            a = a < 0 ? 0 : (a > 255 ? 255 : a);
            b = b < 0 ? 0 : (b > 255 ? 255 : b);
            lo |= (unsigned)a << (8 * j);
            hi |= (unsigned)b << (8 * j);
        }
        out[i] = make_uint2(lo, hi);
    }
}

int tzk_reconstruct(tz_ctx* ctx, const float* pred, const uint8_t* key, const uint8_t* d_key_mask, const int16_t* diff,
                    int nframes, int H, int W, int Hp, int Wp, uint8_t* out) {
    size_t n = (size_t)nframes * H * W * 3;
    if (n == 0) return TZ_OK;
    tz_prof_scope ps(ctx, TZP_RECON);
    const size_t fe = (size_t)H * W * 3;
    if (H == Hp && W == Wp && fe % 8 == 0 && key && ((((uintptr_t)diff | (uintptr_t)pred) & 15) == 0) &&
        ((((uintptr_t)key | (uintptr_t)out) & 7) == 0)) {
        hipLaunchKernelGGL(k_recon_flat, dim3(grid_for(n / 8, 256)), dim3(256), 0, ctx->stream, (const float4*)pred,
                           (const uint2*)key, d_key_mask, (const short8*)diff, n / 8, (unsigned)(fe / 8), (uint2*)out);
    } else {
        hipLaunchKernelGGL(k_recon, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, ctx->stream, pred, key, d_key_mask, diff,
                           n, H, W, Hp, Wp, out);
    }
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

// ------------------------------------------------------------------------------ window SSE
// compress.py:246: mean((X_test_pad - pred)^2) in float64 over PADDED frames.  Per frame the
// sum is taken in a fixed order so that it is reproducible: 4096-element blocks; thread t sums
// elements t, t+256, ... of its block; halving tree over the 256 partials; blocks are added
// in order on the host.  x = float32(k)/255 inside the image, 0 in the pad region.
__global__ __launch_bounds__(256) void k_sse(const uint8_t* __restrict__ orig, const float* __restrict__ pred, int H,
                                             int W, int Hp, int Wp, int nblk, double* __restrict__ partial) {
    __shared__ double s[256];
    int f = blockIdx.y, b = blockIdx.x;
    size_t n = (size_t)Hp * Wp * 3;
    const float* p = pred + (size_t)f * n;
    const uint8_t* o = orig + (size_t)f * H * W * 3;
    double acc = 0.0;
    for (int j = 0; j < 16; ++j) {
        size_t i = (size_t)b * 4096 + (size_t)j * 256 + threadIdx.x;
        double sq = 0.0;
        if (i < n) {
            size_t pix = i / 3;
            int c = (int)(i - pix * 3), y = (int)(pix / Wp), x = (int)(pix - (size_t)y * Wp);
            float xv = 0.0f;
            if (y < H && x < W) xv = (float)o[((size_t)y * W + x) * 3 + c] / 255.0f;
            double d = (double)xv - (double)p[i];
            sq = d * d;
        }
        acc = acc + sq;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)f * nblk + b] = s[0];
}

int tzk_sse_blocks(int Hp, int Wp) { return (int)(((size_t)Hp * Wp * 3 + 4095) / 4096); }

// launch only: per-block partial sums of nframes frames into d_part[nframes][tzk_sse_blocks]
int tzk_sse_launch(tz_ctx* ctx, const uint8_t* orig, const float* pred, int nframes, int H, int W, int Hp, int Wp,
                   double* d_part) {
    if (nframes <= 0) return TZ_OK;
    const int nblk = tzk_sse_blocks(Hp, Wp);
    tz_prof_scope ps(ctx, TZP_SSE);
    hipLaunchKernelGGL(k_sse, dim3(nblk, nframes), dim3(256), 0, ctx->stream, orig, pred, H, W, Hp, Wp, nblk, d_part);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

int tzk_sse(tz_ctx* ctx, const uint8_t* orig, const float* pred, int nframes, int H, int W, int Hp, int Wp,
            double* h_sse) {
    if (nframes <= 0) return TZ_OK;
    const int nblk = tzk_sse_blocks(Hp, Wp);
    void* d_part;
    TZ_TRY(tz_pool_alloc(ctx, sizeof(double) * nblk * nframes, &d_part));
    TZ_TRY(tzk_sse_launch(ctx, orig, pred, nframes, H, W, Hp, Wp, (double*)d_part));
    std::vector<double> part((size_t)nblk * nframes);
    TZ_HIP(ctx, hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, ctx->stream));
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < nframes; ++f) {
        double t = 0.0;
        for (int b = 0; b < nblk; ++b) t = t + part[(size_t)f * nblk + b];
        h_sse[f] = t;
    }
    return TZ_OK;
}
