// Delta / quantise / entropy-prep kernels and their inverses for gfx950 (MI355X).
// HBM-bound integer/byte work: wide coalesced accesses (16 B per lane where the layout
// allows), LDS-privatised histogram, grid-stride launches of ~8 blocks per CU.
// Reference semantics are cited per kernel (paths into /root/reference/src).
#include <algorithm>
#include <vector>

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

// 2 NDW symbols of one lane, packed two per dword.  The update is ONE unconditional LDS atomic per
// element in straight-line code (a branch per element cost more than the atomics it saved: the
// spatial-delta kernel is as much VALU-issue as HBM bound): the centre symbol goes to the lane's own
// sink word (it is counted by the ballot), symbols within +-64 of the centre to [bin][copy], anything
// farther to the junk bin of the copy -- and, in a branch the wave takes only when one of its 512
// symbols is that far out, to the full-range bins.
template <int NDW>
__device__ __forceinline__ void hist_add(HistLds& h, HistAcc& acc, const unsigned* Y, int c) {
    const int lane = threadIdx.x & 63;
    const int base = HL_CENTRAL + (lane & 7), sink = HL_SINK + lane;
    unsigned far = 0;
#pragma unroll
    for (int k = 0; k < 2 * NDW; ++k) {
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
        for (int k = 0; k < 2 * NDW; ++k) {
            const int y = (k & 1) ? ((int)Y[k >> 1] >> 16) : (int)(short)(Y[k >> 1] & 0xFFFFu);
            if ((unsigned)(y - (c - HC_HALF)) >= (unsigned)HC_BINS && (unsigned)y < (unsigned)TZ_NBINS) atomicAdd(&h.w[HL_FULL + y], 1u);
        }
    }
}
__device__ __forceinline__ void hist_add8(HistLds& h, HistAcc& acc, const unsigned* Y, int c) { hist_add<4>(h, acc, Y, c); }
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
// Stage 2 (k_q_width, k_q_tiles, k_q_chain, k_q_bstitch, k_q_serial): exact parallel form of the greedy
//                      interval-intersection segmentation (see the kernels).  Result: one bit per element (`spec`, a 64-bit
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
// Round 3 at cfg3, `abs 2` (profiles/r03/quantiser.md): the tiles read pred / orig (315 MB) and write the values
// (126 MB), the fill reads them and the masks and writes symbols (270 MB), the remap 252 MB: about 1.0 GB and
// 0.5 ms against 1.75 GB and 1.1 ms through k_delta, k_q_init, k_q_heads, k_q_last over tmp, k_q_fill, k_sdelta
// in round 2.
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

// Two forms of the exact greedy segmentation live here.
//
// (1) WAVE-parallel, one 64-element chunk at a time (lane i <-> element; q_chunk).  A run started at s breaks
// at the first i with min(Du[s..i]) < max(Dl[s..i]) (compress.py:60 -- a-b<0 <=> a<b in IEEE), monotone in i, so
// a wave computes per chunk
//   1. range tables T_k[i] = (min Du, max Dl) over [i, i+2^k) by shuffles (k = 0..5),
//   2. nxt[s] = first break after s for EVERY s (fresh start) by binary lifting over T_k, together with the
//      run's (u, l) up to the break,
//   3. the inclusive prefix (min Du, max Dl) from the chunk start,
//   2b. heads[s] = bit mask of the run heads reached from a start at s (pointer doubling),
// and a run carried in from earlier chunks is then resolved without knowing in advance where it breaks:
//   4. it breaks at the first i with min(u, P_u[i]) < max(l, P_l[i]) (ballot); the true heads of the chunk are
//      heads[that i],
//   5. every head whose run closes inside the chunk gets trunc((u+l)/2) (compress.py:61, truncation by the
//      int64 store) at its position of `tmp`; the last head carries on.
// That is about 250 vector instructions and 35 cross-lane shuffles per chunk.  Rounds 1-2 ran the whole chain
// through it (k_q_heads: worker waves + a resolver wave per segment of a chain, 1.29 -> 0.63 ms per cfg3 step;
// profiles/r02/quantiser.md).  Now it serves where a run of UNKNOWN origin has to be taken into a chunk: at the
// tile boundaries (k_q_bstitch) and in the serial fallback (k_q_serial).
//
// (2) LANE-parallel walks from speculative fresh starts (k_q_tiles, below): a dozen integer instructions per
// element instead, because a walk that starts fresh never has to ask where somebody else's run breaks.  What
// makes the guesses exact: two greedy chains over the same data never cross, and they coincide from their first
// common head on; so the chain that really enters a region is followed only until it hits a head of the region's
// speculative chain -- speculative heads in front of that point leave the masks, true ones enter, behind it
// everything is already right.  Nothing depends on run lengths; elements past the chain end are (+inf, -inf)
// and can neither break nor tighten a run.
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

struct QChunk {
    double cu, cl, pu, pl;
    unsigned long long heads;
    int nxt;
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

// ---------------------------------------------------------------------------------------------------------
// The segmentation as LANE-parallel greedy walks (round 3; form (2) above).
//   k_q_tiles   a wave owns 4096 consecutive elements of a chain, staged transposed in LDS; lane j walks chunk j
//               (64 elements = one mask word) from a fresh start and notes the heads in its mask.  Then the
//               lanes are stitched: lane j takes the run lane j-1 leaves open, lets it pass if its whole chunk fits
//               (one test on the chunk's min / max), else walks until the true chain meets a head of its own
//               speculative chain; heads in front of that point are replaced, behind it the guess stands and the
//               lane's exit state is the speculative one.  Repeated until no lane's incoming run changes (a lane
//               whose chunk the true chain leaves unmerged, or passes through, changes its successor's input:
//               typically 2-3 rounds, 64 at worst).  With the heads final, one more walk forms the run values and
//               puts them at their heads in the LDS tile, which leaves the way it came: 24 bytes per thread, all
//               three channels.  Output per tile: masks, values, the run left open, min / max.
//   k_q_chain   one wave per chain: the tiles' open runs chained -- a run that swallows a whole tile (one test on the
//               tile's min / max) passes, otherwise the tile's own exit state stands (assuming the true chain meets
//               the tile's speculative one inside it) -- giving every tile the run that really enters it.
//   k_q_bstitch one wave per tile boundary, all in parallel: the entering run is walked into the tile with the
//               wave-parallel machinery (q_chunk) until it meets the tile's chain -- usually in the first chunk.
//               A tile it crosses without meeting the speculative chain marks its chain for
//   k_q_serial  the exact serial walk of a whole chain (the algorithm of round 1), which runs only for such chains.
// For abs / rel / absrel the walks are integer: a run is (min d, max d), and "does it break" is
// fl(min d + E) < fl(max d - E) (compress.py:55-60), monotone in both arguments; k_q_width finds per chain the
// widths w = max d - min d that are always fine (w <= wmin) and always a break (w > wmax) by evaluating the
// double test itself on all 511 x 3 candidate pairs -- the two differ only when 2E sits within a rounding error
// of an integer, and then the walk evaluates the double test for the widths in between.  Run values are
// trunc((fl(min d + E) + fl(max d - E)) / 2) in double, as the reference computes them.
struct QWidth {
    int wmin, wmax;
};

__device__ __forceinline__ bool q_ok_exact(int mn, int mx, double E) { return !(((double)mn + E) < ((double)mx - E)); }

__global__ __launch_bounds__(64) void k_q_width(const double* __restrict__ Echain, const uint8_t* __restrict__ skip, QParams qp,
                                              QWidth* __restrict__ width) {
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (skip[chain / 3]) return;
    const double E = qp.mode == TZ_MODE_ABS ? fabs(qp.b0) : Echain[chain];
    int wmin = 511, wmax = -1;
    for (int mn = -255 + lane; mn <= 255; mn += 64) {
        // W(mn) = largest w with ok(mn, mn + w) (monotone in w); 511 when every feasible w is fine
        int W;
        if (!q_ok_exact(mn, mn, E)) W = -1;
        else if (q_ok_exact(mn, 255, E)) W = 511;
        else {
            int lo = 0, hi = 255 - mn;   // ok(lo), !ok(hi)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (q_ok_exact(mn, mn + mid, E)) lo = mid;
                else hi = mid;
            }
            W = lo;
        }
        wmin = min(wmin, W);
        wmax = max(wmax, W);
    }
    for (int s = 32; s >= 1; s >>= 1) {
        wmin = min(wmin, __shfl_down(wmin, s, 64));
        wmax = max(wmax, __shfl_down(wmax, s, 64));
    }
    if (lane == 0) width[chain] = QWidth{wmin, wmax};
}

static constexpr int QT_CHUNKS = 64;                 // chunks (= lanes) per tile
static constexpr int QT_TILE = QT_CHUNKS * 64;       // elements per tile
static constexpr int QT_S16 = 66, QT_S32 = 65;       // LDS row strides (int16 / int32 elements): lanes j, j+32 share a bank, no more

// run state of the integer walk: mn > mx = empty
struct QRunI {
    int mn, mx, hd;   // hd: chain index of the run's head
};
struct QRunD {
    double u, l;
    int hd;
};

template <bool PW, bool FP>
__global__ __launch_bounds__(192) void k_q_tiles(QSrc src, const uint8_t* __restrict__ orig, const uint8_t* __restrict__ skip,
                                                 int HW, int ntiles, QParams qp, const double* __restrict__ Echain,
                                                 const QWidth* __restrict__ width, int16_t* __restrict__ tmp,
                                                 unsigned long long* __restrict__ spec, QState* __restrict__ tile_out,
                                                 double2* __restrict__ tile_agg) {
    __shared__ int lds_raw[3 * (PW ? 64 * QT_S32 : 64 * QT_S16 / 2)];
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;   // one wave per channel of the same pixels (they share cache lines)
    const int f = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
    if (skip[f]) return;
    const int chain = f * 3 + c;
    const size_t fe0 = (size_t)f * HW * 3;
    int16_t* t = tmp + fe0;
    const int nch = (HW + 63) >> 6;
    const int e0 = tile * QT_TILE;                              // first element of the tile in the chain
    short* ls = (short*)lds_raw + c * 64 * QT_S16;              // !PW: int16 deltas
    unsigned* lw = (unsigned*)lds_raw + c * 64 * QT_S32;        //  PW: delta | orig << 16
    // ---- stage the tile transposed: row r = the 64 elements of chunk r (lane r walks it).
    // Frames of whole 4-pixel groups: the block's three waves load the tile TOGETHER -- a thread takes 4 pixels =
    // 12 consecutive samples (three 16-byte loads of the prediction + 12 original bytes, or 24 bytes of the delta
    // stack), forms the deltas and deals them to the three channel planes.  (A wave per channel fetching its own
    // samples with a stride of three touched every cache line three times, 2-byte / 4-byte accesses: 141 of the
    // first version's 280 us.)
    const bool grouped = (HW & 3) == 0 && (((uintptr_t)src.pred & 15) | ((uintptr_t)src.diff & 7) | ((uintptr_t)orig & 3) | ((uintptr_t)tmp & 7)) == 0;
    // The integer walk below rests on k_q_width's table, which covers deltas in [-255, 255] -- everything compress.py:292-314
    // can produce.  The stand-alone tz_error_bound takes ANY int16 stack: a tile that holds a value outside that range
    // (`wide`) evaluates the double test of compress.py:60 itself at every step instead.
    int wide = 0;
    if (grouped) {
        const int npix = min(QT_TILE, HW - e0);
        for (int g = threadIdx.x; g < QT_TILE / 4; g += 192) {
            const int p0 = g * 4;
            int d[12], o[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) d[k] = o[k] = 0;
            if (p0 < npix) {
                const size_t e = fe0 + (size_t)(e0 + p0) * 3;
                if (FP || PW) {
                    const uint3 ob = *(const uint3*)(orig + e);
                    const unsigned ow[3] = {ob.x, ob.y, ob.z};
#pragma unroll
                    for (int k = 0; k < 12; ++k) o[k] = (int)((ow[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                }
                if (FP) {
                    const float4* pp = (const float4*)(src.pred + e);
                    const float4 f0 = pp[0], f1 = pp[1], f2 = pp[2];
                    const float fv[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
#pragma unroll
                    for (int k = 0; k < 12; ++k) d[k] = (int)(fv[k] * 255.0f) - o[k];
                } else {
                    const uint2* dp = (const uint2*)(src.diff + e);
                    const uint2 a0 = dp[0], a1 = dp[1], a2 = dp[2];
                    const unsigned dw[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        d[k] = (int)(short)((dw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
                        wide |= (unsigned)(d[k] + 255) > 510u;
                    }
                }
            }
            const int row = p0 >> 6, col = p0 & 63;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {   // sample k of the group = pixel k / 3, channel k % 3
                if (PW) {
                    unsigned* w = (unsigned*)lds_raw + cc * 64 * QT_S32 + row * QT_S32 + col;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = ((unsigned)d[3 * i + cc] & 0xFFFFu) | ((unsigned)o[3 * i + cc] << 16);
                } else {
                    unsigned* w = (unsigned*)((short*)lds_raw + cc * 64 * QT_S16 + row * QT_S16 + col);   // 4-byte aligned: 132 r + 2 col
                    w[0] = ((unsigned)d[cc] & 0xFFFFu) | ((unsigned)d[3 + cc] << 16);
                    w[1] = ((unsigned)d[6 + cc] & 0xFFFFu) | ((unsigned)d[9 + cc] << 16);
                }
            }
        }
        if (FP || PW) __syncthreads();
        else wide = __syncthreads_or(wide);
    } else {
#pragma unroll 8
        for (int r = 0; r < 64; ++r) {
            const int idx = e0 + r * 64 + lane;
            int d = 0, o = 0;
            if (idx < HW) {
                const size_t e = fe0 + (size_t)idx * 3 + c;
                d = q_delta<FP>(src, orig, e);
                if (PW) o = orig[e];
                if (!FP && !PW) wide |= (unsigned)(d + 255) > 510u;
            }
            if (PW) lw[r * QT_S32 + lane] = ((unsigned)d & 0xFFFFu) | ((unsigned)o << 16);
            else ls[r * QT_S16 + lane] = (short)d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int base = e0 + lane * 64;                            // chain index of my chunk's first element
    const int len = max(0, min(64, HW - base));
    double E = 0.0;
    int wmin = 0, wmax = 0;
    if (!PW) {
        E = qp.mode == TZ_MODE_ABS ? fabs(qp.b0) : Echain[chain];
        const QWidth wd = width[chain];
        wmin = wd.wmin;
        wmax = wd.wmax;
        if (!FP && __any(wide)) {   // no width is taken from the table: "ambiguous" always, q_ok_exact decides
            wmin = -1;
            wmax = 0x7FFFFFFF;
        }
    }
    const double inf = __builtin_huge_val();
    // element access and the run algebra of the two arithmetic forms
    auto elem_i = [&](int tt) { return (int)ls[lane * QT_S16 + tt]; };
    // (wave-uniform, and opaque to the compiler: it would otherwise evaluate the double test speculatively in every step)
    const bool banded = __builtin_amdgcn_readfirstlane(wmin != wmax) != 0;
    auto ok_i = [&](int mn, int mx) {   // may the run (mn, mx) stand?  exact form of compress.py:60
        const int w = mx - mn;
        bool ok = w <= wmin;
        if (banded) {
            const bool amb = !ok && w <= wmax;
            if (__any(amb)) ok = ok || (amb && q_ok_exact(mn, mx, E));
        }
        return ok;
    };
    // |value| <= 255 + E: the int conversion is the int64 store of compress.py:61 (one instruction instead of five)
    auto val_i = [&](int mn, int mx) { return (int)((((double)mn + E) + ((double)mx - E)) / 2); };
    auto elem_d = [&](int tt, double& du, double& dl) {
        const unsigned w = lw[lane * QT_S32 + tt];
        const double df = (double)(int)(short)(w & 0xFFFFu), tol = (double)(int)(w >> 16) * qp.b0;
        du = df + tol;
        dl = df - tol;
    };

    // ---- 1. speculative walk of my chunk from a fresh start: heads only (the values follow once the heads are final)
    unsigned long long Ms = len > 0 ? 1ull : 0ull;
    QRunI xi{32767, -32768, base};      // !PW
    QRunD xd{inf, -inf, base};          //  PW
    int amn = 32767, amx = -32768;      // chunk aggregate (!PW)
    double au = inf, al = -inf;         //                 ( PW)
    for (int tt = 0; tt < 64; ++tt) {
        const bool act = tt < len;
        if (!PW) {
            const int d = elem_i(tt);
            const int nmn = min(xi.mn, d), nmx = max(xi.mx, d);
            const bool brk = act && xi.mn <= xi.mx && !ok_i(nmn, nmx);
            if (brk) {
                Ms |= 1ull << tt;
                xi = QRunI{d, d, base + tt};
            } else if (act) {
                xi.mn = nmn;
                xi.mx = nmx;
            }
            if (act) {
                amn = min(amn, d);
                amx = max(amx, d);
            }
        } else {
            double du, dl;
            elem_d(tt, du, dl);
            const double nu = xd.u < du ? xd.u : du, nl = xd.l > dl ? xd.l : dl;
            const bool brk = act && (nu < nl);   // a fresh run (inf, -inf) cannot break: du >= dl for a tolerance >= 0
            if (brk) {
                Ms |= 1ull << tt;
                xd = QRunD{du, dl, base + tt};
            } else if (act) {
                xd.u = nu;
                xd.l = nl;
            }
            if (act) {
                au = au < du ? au : du;
                al = al > dl ? al : dl;
            }
        }
    }
    const QRunI xsi = xi;
    const QRunD xsd = xd;
    // ---- 2. stitch the lanes: my incoming run is what lane - 1 leaves open; repeat until nothing changes
    unsigned long long M = Ms;
    QRunI pini{0, 0, -2};               // the incoming run my current (M, exit) was computed for; hd -2 = none yet
    QRunD pind{0.0, 0.0, -2};
    for (int iter = 0; iter < 66; ++iter) {
        bool dirty;
        QRunI ini{};
        QRunD ind{};
        if (!PW) {
            ini.mn = __shfl_up(xi.mn, 1, 64);
            ini.mx = __shfl_up(xi.mx, 1, 64);
            ini.hd = __shfl_up(xi.hd, 1, 64);
            dirty = lane > 0 && len > 0 && (ini.mn != pini.mn || ini.mx != pini.mx || ini.hd != pini.hd);
        } else {
            ind.u = __shfl_up(xd.u, 1, 64);
            ind.l = __shfl_up(xd.l, 1, 64);
            ind.hd = __shfl_up(xd.hd, 1, 64);
            dirty = lane > 0 && len > 0 && (ind.u != pind.u || ind.l != pind.l || ind.hd != pind.hd);
        }
        if (!__any(dirty)) break;
        bool walking = false;
        unsigned long long newm = 0;
        if (dirty) {
            pini = ini;
            pind = ind;
            if (!PW) {
                const int qmn = min(ini.mn, amn), qmx = max(ini.mx, amx);
                if (ok_i(qmn, qmx)) {       // the run passes through my whole chunk
                    M = 0;
                    xi = QRunI{qmn, qmx, ini.hd};
                } else {
                    walking = true;
                    xi = ini;
                }
            } else {
                const double qu = ind.u < au ? ind.u : au, ql = ind.l > al ? ind.l : al;
                if (!(qu < ql)) {
                    M = 0;
                    xd = QRunD{qu, ql, ind.hd};
                } else {
                    walking = true;
                    xd = ind;
                }
            }
        }
        for (int tt = 0; tt < 64 && __any(walking); ++tt) {
            const bool act = walking && tt < len;
            bool brk;
            if (!PW) {
                const int d = elem_i(tt);
                const int nmn = min(xi.mn, d), nmx = max(xi.mx, d);
                brk = act && !ok_i(nmn, nmx);
                if (brk) xi = QRunI{d, d, base + tt};
                else if (act) {
                    xi.mn = nmn;
                    xi.mx = nmx;
                }
            } else {
                double du, dl;
                elem_d(tt, du, dl);
                const double nu = xd.u < du ? xd.u : du, nl = xd.l > dl ? xd.l : dl;
                brk = act && (nu < nl);
                if (brk) xd = QRunD{du, dl, base + tt};
                else if (act) {
                    xd.u = nu;
                    xd.l = nl;
                }
            }
            if (brk) {
                if ((Ms >> tt) & 1ull) {   // the true chain starts a run where my speculative chain does: from here on they are one
                    M = newm | (Ms & ~((1ull << tt) - 1ull));
                    xi = xsi;
                    xd = xsd;
                    walking = false;
                } else {
                    newm |= 1ull << tt;
                }
            }
            if (walking && tt == len - 1) {   // left my chunk without meeting the speculative chain
                M = newm;
                walking = false;
            }
        }
    }
    // ---- 3. the run values, now that the heads are final: one more walk of my chunk, which starts with the run that
    // really enters it (pini / pind: what the last stitch round was computed for; lane 0 starts fresh).  A run's value
    // replaces the delta at its head IN the LDS tile -- the element has been consumed by then; the one run that may
    // close in my chunk but started in an earlier lane's row is written after the walk (its row's owner may not have
    // read that element yet).  The run open at the end of the tile is closed by k_q_chain / k_q_bstitch.
    {
        QRunI ri{32767, -32768, base};
        QRunD rd{inf, -inf, base};
        if (lane > 0 && len > 0) {
            ri = pini;
            rd = pind;
        }
        int late_hd = -1, late_v = 0;
        for (int tt = 0; tt < 64; ++tt) {
            const bool act = tt < len;
            const bool head = act && ((M >> tt) & 1ull);
            int d = 0;
            double du = 0.0, dl = 0.0;
            if (!PW) d = elem_i(tt);
            else elem_d(tt, du, dl);
            if (__any(head)) {
                // (the run in front of a head is empty only for lane 0's first element)
                if (head && (!PW ? ri.mn <= ri.mx : (rd.u != inf || rd.l != -inf))) {
                    const int v = !PW ? val_i(ri.mn, ri.mx) : (int)((rd.u + rd.l) / 2);
                    const int hd = !PW ? ri.hd : rd.hd;
                    if (hd >= base) {
                        if (!PW) ls[lane * QT_S16 + (hd - base)] = (short)v;
                        else lw[lane * QT_S32 + (hd - base)] = (unsigned)v & 0xFFFFu;
                    } else {
                        late_hd = hd;
                        late_v = v;
                    }
                }
            }
            if (head) {
                ri = QRunI{d, d, base + tt};
                rd = QRunD{du, dl, base + tt};
            } else if (act) {
                ri.mn = min(ri.mn, d);
                ri.mx = max(ri.mx, d);
                rd.u = rd.u < du ? rd.u : du;
                rd.l = rd.l > dl ? rd.l : dl;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (late_hd >= 0) {
            const int rel = late_hd - e0;
            if (!PW) ls[(rel >> 6) * QT_S16 + (rel & 63)] = (short)late_v;
            else lw[(rel >> 6) * QT_S32 + (rel & 63)] = (unsigned)late_v & 0xFFFFu;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (grouped) {
            // flush, the way the tile came in: all three planes, 12 consecutive samples = 24 bytes per thread.  Dense -- what is
            // not a head carries its delta along, which nobody reads (the masks say where the heads are)
            __syncthreads();
            const int npix = min(QT_TILE, HW - e0);
            for (int g = threadIdx.x; g < QT_TILE / 4; g += 192) {
                const int p0 = g * 4;
                if (p0 >= npix) continue;
                const int row = p0 >> 6, col = p0 & 63;
                unsigned v[12];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    if (PW) {
                        const unsigned* w = (const unsigned*)lds_raw + cc * 64 * QT_S32 + row * QT_S32 + col;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[3 * i + cc] = w[i] & 0xFFFFu;
                    } else {
                        const unsigned* w = (const unsigned*)((const short*)lds_raw + cc * 64 * QT_S16 + row * QT_S16 + col);
                        const unsigned w0 = w[0], w1 = w[1];
                        v[cc] = w0 & 0xFFFFu;
                        v[3 + cc] = w0 >> 16;
                        v[6 + cc] = w1 & 0xFFFFu;
                        v[9 + cc] = w1 >> 16;
                    }
                }
                uint2* dst = (uint2*)(t + (size_t)(e0 + p0) * 3);
                dst[0] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
                dst[1] = make_uint2(v[4] | (v[5] << 16), v[6] | (v[7] << 16));
                dst[2] = make_uint2(v[8] | (v[9] << 16), v[10] | (v[11] << 16));
            }
        } else {
            // flush: row r = chunk r, a lane per element; only heads carry a value
            const unsigned mlo = (unsigned)M, mhi = (unsigned)(M >> 32);
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                const unsigned long long mr = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, r) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane((int)mlo, r);
                if ((mr >> lane) & 1ull) {
                    const int v = !PW ? (int)ls[r * QT_S16 + lane] : (int)(short)(lw[r * QT_S32 + lane] & 0xFFFFu);
                    t[(size_t)(e0 + r * 64 + lane) * 3 + c] = (int16_t)v;
                }
            }
        }
    }
    // ---- 4. results
    const int chunk = tile * QT_CHUNKS + lane;
    if (chunk < nch) spec[(size_t)chain * nch + chunk] = M;
    const int lastl = min(63, (min(HW, e0 + QT_TILE) - e0 - 1) >> 6);   // the last lane that holds elements
    if (!PW) {
        int tmn = len > 0 ? amn : 32767, tmx = len > 0 ? amx : -32768;
        for (int sft = 32; sft >= 1; sft >>= 1) {
            tmn = min(tmn, __shfl_xor(tmn, sft, 64));
            tmx = max(tmx, __shfl_xor(tmx, sft, 64));
        }
        if (lane == lastl) {
            tile_out[(size_t)chain * ntiles + tile] = QState{(double)xi.mn + E, (double)xi.mx - E, xi.hd, 0};
            tile_agg[(size_t)chain * ntiles + tile] = make_double2((double)tmn + E, (double)tmx - E);
        }
    } else {
        double tu = au, tl = al;
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const double a = __shfl_xor(tu, sft, 64), b = __shfl_xor(tl, sft, 64);
            tu = tu < a ? tu : a;
            tl = tl > b ? tl : b;
        }
        if (lane == lastl) {
            tile_out[(size_t)chain * ntiles + tile] = QState{xd.u, xd.l, xd.hd, 0};
            tile_agg[(size_t)chain * ntiles + tile] = make_double2(tu, tl);
        }
    }
}

// The run that really enters every tile (see k_q_tiles); closes the run that is open at the end of the chain.
__global__ __launch_bounds__(64) void k_q_chain(const uint8_t* __restrict__ skip, int HW, int ntiles, const QState* __restrict__ tile_out,
                                              const double2* __restrict__ tile_agg, QState* __restrict__ tile_in,
                                              uint8_t* __restrict__ swallowed, int16_t* __restrict__ tmp) {
    const int chain = blockIdx.x, f = chain / 3, c = chain % 3, lane = threadIdx.x;
    if (skip[f]) return;
    double su = 0.0, sl = 0.0;
    int sh = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 64) {
        const int T = t0 + lane;
        QState o{0.0, 0.0, 0, 0};
        double2 g = make_double2(0.0, 0.0);
        if (T < ntiles) {
            o = tile_out[(size_t)chain * ntiles + T];
            g = tile_agg[(size_t)chain * ntiles + T];
        }
        double iu = 0.0, il = 0.0;
        int ih = 0, sw = 0;
        const int cnt = min(64, ntiles - t0);
        for (int k = 0; k < cnt; ++k) {
            const double ou = q_rl_d(o.u, k), ol = q_rl_d(o.l, k);
            const int oh = __builtin_amdgcn_readlane(o.head, k);
            if (t0 + k == 0) {          // the first tile starts where the chain starts: its guess is the truth
                su = ou;
                sl = ol;
                sh = oh;
                continue;
            }
            const double gu = q_rl_d(g.x, k), gl = q_rl_d(g.y, k);
            const double qu = su < gu ? su : gu, ql = sl > gl ? sl : gl;
            const bool pass = !(qu < ql);
            if (lane == k) {
                iu = su;
                il = sl;
                ih = sh;
                sw = pass;
            }
            if (pass) {
                su = qu;
                sl = ql;
            } else {
                su = ou;
                sl = ol;
                sh = oh;
            }
        }
        if (T < ntiles && T > 0) {
            tile_in[(size_t)chain * ntiles + T] = QState{iu, il, ih, 0};
            swallowed[(size_t)chain * ntiles + T] = (uint8_t)sw;
        }
    }
    if (lane == 0 && HW > 0) tmp[((size_t)f * HW + sh) * 3 + c] = (int16_t)(long long)((su + sl) / 2);
}

// The run entering tile T walked into the tile until it meets the tile's own chain (see k_q_tiles); a tile that is
// crossed without that marks the chain for k_q_serial.
template <bool FP>
__global__ __launch_bounds__(64) void k_q_bstitch(QSrc src, const uint8_t* __restrict__ orig, const uint8_t* __restrict__ skip,
                                                int HW, int ntiles, QParams qp, const double* __restrict__ Echain,
                                                int16_t* __restrict__ tmp, unsigned long long* __restrict__ spec,
                                                const QState* __restrict__ tile_in, const uint8_t* __restrict__ swallowed,
                                                int* __restrict__ bad) {
    const int chain = blockIdx.x / (ntiles - 1), T = 1 + blockIdx.x % (ntiles - 1);
    const int f = chain / 3, c = chain % 3, lane = threadIdx.x;
    if (skip[f]) return;
    const size_t fe0 = (size_t)f * HW * 3;
    int16_t* t = tmp + fe0;
    const int nch = (HW + 63) >> 6;
    unsigned long long* sp = spec + (size_t)chain * nch;
    const int c0 = T * QT_CHUNKS, c1 = min(nch, c0 + QT_CHUNKS);
    if (swallowed[(size_t)chain * ntiles + T]) {   // the entering run covers the whole tile: none of its guessed heads is one
        if (c0 + lane < c1) sp[c0 + lane] = 0ull;
        return;
    }
    double E = 0.0;
    if (qp.mode == TZ_MODE_ABS) E = fabs(qp.b0);
    else if (qp.mode != TZ_MODE_PWREL) E = Echain[chain];
    const QState st = tile_in[(size_t)chain * ntiles + T];
    double u = st.u, l = st.l;
    int chead = st.head;
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
        const unsigned long long Tm = q_rl_u64(q.heads, j0);         // true heads of this chunk
        const unsigned long long common = Tm & S;
        const int m = common ? __ffsll((long long)common) - 1 : 64;  // first common head
        const unsigned long long below = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
        const bool mine = (below >> lane) & 1ull;
        if (mine && ((Tm >> lane) & 1ull) && q.nxt < 64)
            t[(size_t)(ch * 64 + lane) * 3 + c] = (int16_t)(long long)((q.cu + q.cl) / 2);
        if (lane == 0) {
            sp[ch] = (S & ~below) | (Tm & below);
            t[(size_t)chead * 3 + c] = (int16_t)(long long)((uc + lc) / 2);   // the entering run closes at j0
        }
        if (common) {
            merged = true;
        } else {
            const int last = 63 - __clzll((long long)Tm);
            u = q_rl_d(q.cu, last);
            l = q_rl_d(q.cl, last);
            chead = ch * 64 + last;
        }
    }
    if (!merged && lane == 0) atomicOr(&bad[chain], 1);
}

// Exact serial walk of a whole chain (one wave, chunk after chunk: the algorithm of round 1), for the chains
// k_q_bstitch marked: the true chain crossed a whole 4096-element tile out of step with the tile's speculative chain
// (regular ramps do that).  Rewrites every mask word and run value of the chain.
template <bool FP>
__global__ __launch_bounds__(64) void k_q_serial(QSrc src, const uint8_t* __restrict__ orig, const uint8_t* __restrict__ skip, int HW,
                                               QParams qp, const double* __restrict__ Echain, int16_t* __restrict__ tmp,
                                               unsigned long long* __restrict__ spec, const int* __restrict__ bad) {
    const int chain = blockIdx.x, f = chain / 3, c = chain % 3, lane = threadIdx.x;
    if (skip[f] || !bad[chain] || HW <= 0) return;
    const size_t fe0 = (size_t)f * HW * 3;
    int16_t* t = tmp + fe0;
    const int nch = (HW + 63) >> 6;
    unsigned long long* sp = spec + (size_t)chain * nch;
    double E = 0.0;
    if (qp.mode == TZ_MODE_ABS) E = fabs(qp.b0);
    else if (qp.mode != TZ_MODE_PWREL) E = Echain[chain];
    double u = __builtin_huge_val(), l = -__builtin_huge_val();
    int chead = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const QChunk q = qp.mode == TZ_MODE_PWREL ? q_chunk<true, FP>(src, orig, fe0, c, ch, HW, qp, E, lane)
                                                  : q_chunk<false, FP>(src, orig, fe0, c, ch, HW, qp, E, lane);
        const unsigned long long first = ch == 0 ? 1ull : 0ull;
        const double eu = u < q.pu ? u : q.pu, el = l > q.pl ? l : q.pl;
        const unsigned long long brk = __ballot(eu - el < 0.0);
        if (brk == 0ull) {
            u = q_rl_d(eu, 63);
            l = q_rl_d(el, 63);
            if (lane == 0) sp[ch] = first;
            continue;
        }
        const int j0 = __ffsll((long long)brk) - 1;
        double uc = u, lc = l;
        if (j0 > 0) {
            uc = q_rl_d(eu, j0 - 1);
            lc = q_rl_d(el, j0 - 1);
        }
        if (lane == 0) t[(size_t)chead * 3 + c] = (int16_t)(long long)((uc + lc) / 2);
        const unsigned long long heads = q_rl_u64(q.heads, j0);
        const int last = 63 - __clzll((long long)heads);
        if (((heads >> lane) & 1ull) && q.nxt < 64) t[(size_t)(ch * 64 + lane) * 3 + c] = (int16_t)(long long)((q.cu + q.cl) / 2);
        if (lane == 0) sp[ch] = heads | first;
        u = q_rl_d(q.cu, last);
        l = q_rl_d(q.cl, last);
        chead = ch * 64 + last;
    }
    if (lane == 0) t[(size_t)chead * 3 + c] = (int16_t)(long long)((u + l) / 2);
}

static constexpr int QFB = QT_TILE;  // pixels per fill item = one tile of k_q_tiles (64 chunks of 64: a lane per mask word)

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

// Forward fill of the run values back into a DELTA STACK (stand-alone tz_error_bound, delta tap of tz_encode; frames
// flagged in `skip` are not touched), one wave per 64-element chunk, no communication between waves: the masks say where
// the run heads are, so a lane finds the head of ITS run with bit operations (highest mask bit at or below its lane;
// else the highest bit of the nearest non-empty mask word in front of the chunk within its tile; else the tile's carry
// value from k_q_last / k_q_carry) and fetches the value there -- rounds 1-2 propagated the values themselves with 18
// shuffles and two barriers per 256 pixels.  (The fused encode uses k_q_fill_sym below.)
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
        r.mw[c] = (lane < QFB / 64 && wi < ch) ? sp[wi] : 0ull;   // the tile's mask words in front of the chunk, one per lane
        r.m[c] = sp[ch];
        r.cv[c] = (int)a.carry[((size_t)f * a.nblk + b) * 3 + c];  // value of the last head in front of the tile
    }
    return r;
}

__global__ __launch_bounds__(QF_THREADS) void k_q_fill(const QFill a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int HW = a.HW, nch = a.nch;
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
        if (cur.skipped) continue;
        const size_t fe0 = (size_t)fc * HW * 3;
        const int16_t* t = a.tmp + fe0;
        const int p = chc * 64 + lane;
        const int w0 = ((chc * 64) / QFB) * (QFB / 64);   // first mask word of the tile this chunk belongs to
        int hp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned long long mw = cur.mw[c];
            const unsigned long long nz = __ballot(mw != 0ull);
            int before = -1;   // position of the last head in front of the chunk (within the tile), or -1
            if (nz) {
                const int wl = 63 - __clzll((long long)nz);
                const unsigned long long mm = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mw >> 32), wl) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mw, wl);
                before = (w0 + wl) * 64 + 63 - __clzll((long long)mm);
            }
            const unsigned long long mine = cur.m[c] & ((2ull << lane) - 1ull);
            hp[c] = mine ? chc * 64 + 63 - __clzll((long long)mine) : before;
        }
        int16_t tv[3];   // the gathers of the three channels go out together
#pragma unroll
        for (int c = 0; c < 3; ++c) tv[c] = hp[c] >= 0 ? t[(size_t)hp[c] * 3 + c] : (int16_t)0;
        if (p < HW)
#pragma unroll
            for (int c = 0; c < 3; ++c) a.out[fe0 + (size_t)p * 3 + c] = hp[c] >= 0 ? tv[c] : (int16_t)cur.cv[c];
    }
}

// Fused encode: forward fill + spatial delta + 1600 offset + histogram, tile by tile (the tiles of k_q_tiles), by
// resident blocks of three waves.  The values of a tile come in the way k_q_tiles left them -- 24 bytes per thread,
// dealt to three channel planes in LDS, row r = chunk r --; lane j of the channel's wave fills row j from its mask
// word (the value in front of the row comes from the nearest lane above that holds a head, else from the tile's carry);
// then the block walks the tile in memory order again, 4 pixels = 12 consecutive samples per thread: spatial delta in
// packed int16 arithmetic (the sample in front of a group is channel 2 of the pixel before it: LDS, or the tile's /
// frame's carry), symbols out in 24 bytes, histogram.  Frames the quantiser skips take their deltas from pred / orig
// (compress.py:292-314; 0 where zero_mask says so).  About 40 vector instructions per 12 symbols; the first version of
// the fused fill (a wave per 64-pixel chunk gathering run values from global memory, k_q_fill<true>) spent 300 per
// 192 and was bound by them: 193 us at cfg3.
template <bool HIST>
__global__ __launch_bounds__(192) void k_q_fill_sym(const QFill a) {
    __shared__ int planes[3 * 64 * QT_S16 / 2];
    __shared__ unsigned hraw[HIST ? HL_WORDS : 1];
    HistLds& hl = *(HistLds*)hraw;
    HistAcc acc;
    const int centre = a.apply_offset ? TZ_OFFSET : 0;
    if (HIST) hist_clear(hl);
    const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int HW = a.HW, nch = a.nch, ntiles = a.nblk;
    short* ls = (short*)planes + c * 64 * QT_S16;
    auto raw = [&](size_t e) { return (int)(a.pred[e] * 255.0f) - (int)a.orig[e]; };
    for (int item = blockIdx.x; item < a.nframes * ntiles; item += gridDim.x) {
        const int f = item / ntiles, T = item % ntiles;
        const bool skipped = a.skip[f] != 0, zero = skipped && a.zero[f] != 0;
        const size_t fe0 = (size_t)f * HW * 3;
        const int e0 = T * QT_TILE, npix = min(QT_TILE, HW - e0);
        __syncthreads();   // the previous item's readers of the planes are done
        // ---- A. the tile's values (or raw deltas) into the channel planes
        for (int g = threadIdx.x; g < QT_TILE / 4; g += 192) {
            const int p0 = g * 4;
            unsigned v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = 0;
            if (p0 < npix && !zero) {
                const size_t e = fe0 + (size_t)(e0 + p0) * 3;
                if (skipped) {
                    const uint3 ob = *(const uint3*)(a.orig + e);
                    const unsigned ow[3] = {ob.x, ob.y, ob.z};
                    const float4* pp = (const float4*)(a.pred + e);
                    const float4 f0 = pp[0], f1 = pp[1], f2 = pp[2];
                    const float fv[12] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w};
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[k] = (unsigned)((int)(fv[k] * 255.0f) - (int)((ow[k >> 2] >> (8 * (k & 3))) & 0xFFu)) & 0xFFFFu;
                } else {
                    const uint2* dp = (const uint2*)(a.tmp + e);
                    const uint2 a0 = dp[0], a1 = dp[1], a2 = dp[2];
                    const unsigned dw[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[k] = (dw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                }
            }
            const int row = p0 >> 6, col = p0 & 63;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                unsigned* w = (unsigned*)((short*)planes + cc * 64 * QT_S16 + row * QT_S16 + col);
                w[0] = v[cc] | (v[3 + cc] << 16);
                w[1] = v[6 + cc] | (v[9 + cc] << 16);
            }
        }
        __syncthreads();
        // ---- B. forward fill of my row
        const int cv = skipped ? 0 : (int)a.carry[((size_t)f * ntiles + T) * 3 + c];   // value of the last head in front of the tile
        if (!skipped) {
            const int chunk = T * QT_CHUNKS + lane;
            const unsigned long long M = chunk < nch ? a.spec[(size_t)(f * 3 + c) * nch + chunk] : 0ull;
            const int lastv = M ? (int)ls[lane * QT_S16 + 63 - __clzll((long long)M)] : 0;
            const unsigned long long has = __ballot(M != 0ull), below = has & ((1ull << lane) - 1ull);
            const int got = __shfl(lastv, below ? 63 - __clzll((long long)below) : lane, 64);
            int cur = below ? got : cv;
            __builtin_amdgcn_wave_barrier();   // every lane has read its lastv before any row is rewritten
            const int len = max(0, min(64, npix - lane * 64));
            for (int tt = 0; tt < len; ++tt) {
                const int x = (int)ls[lane * QT_S16 + tt];
                cur = ((M >> tt) & 1ull) ? x : cur;
                ls[lane * QT_S16 + tt] = (short)cur;
            }
        }
        __syncthreads();
        // ---- C. memory order again: spatial delta, offset, histogram, symbols
        // the sample in front of the tile's first one
        int tprev = 0;
        bool tfirst = false;   // start of the stream: sd[0] = x[0]
        if (T > 0) {
            tprev = skipped ? (zero ? 0 : raw(fe0 + (size_t)e0 * 3 - 1)) : (int)a.carry[((size_t)f * ntiles + T) * 3 + 2];
        } else if (f > 0) {
            if (a.skip[f - 1]) tprev = a.zero[f - 1] ? 0 : raw(fe0 - 1);
            else tprev = (int)a.ftail[(f - 1) * 3 + 2];
        } else {
            tfirst = true;
        }
        for (int g = threadIdx.x; g < QT_TILE / 4; g += 192) {
            const int p0 = g * 4;
            if (p0 >= npix) continue;
            const int row = p0 >> 6, col = p0 & 63;
            unsigned r[12];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const unsigned* w = (const unsigned*)((const short*)planes + cc * 64 * QT_S16 + row * QT_S16 + col);
                const unsigned w0 = w[0], w1 = w[1];
                r[cc] = w0 & 0xFFFFu;
                r[3 + cc] = w0 >> 16;
                r[6 + cc] = w1 & 0xFFFFu;
                r[9 + cc] = w1 >> 16;
            }
            unsigned V[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) V[k] = r[2 * k] | (r[2 * k + 1] << 16);
            unsigned prev16;
            if (p0 > 0) prev16 = (unsigned)(unsigned short)((const short*)planes)[2 * 64 * QT_S16 + ((p0 - 1) >> 6) * QT_S16 + ((p0 - 1) & 63)];
            else prev16 = tfirst ? ((V[0] << 1) & 0xFFFFu) : ((unsigned)tprev & 0xFFFFu);   // sd[0] = x[0] as "prev = 2 x[0]"
            unsigned Y[6];
            unsigned hi = prev16 << 16;
            const unsigned C2 = ((unsigned)TZ_OFFSET << 16) | (unsigned)TZ_OFFSET;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const unsigned P = __builtin_amdgcn_alignbit(V[k], hi, 16);
                hi = V[k];
                Y[k] = pk_sub(P, V[k]);
                if (a.apply_offset) Y[k] = pk_sub(C2, Y[k]);
            }
            uint2* dst = (uint2*)(a.out + fe0 + (size_t)(e0 + p0) * 3);
            dst[0] = make_uint2(Y[0], Y[1]);
            dst[1] = make_uint2(Y[2], Y[3]);
            dst[2] = make_uint2(Y[4], Y[5]);
            if (HIST) hist_add<6>(hl, acc, Y, centre);
            if (f == 0 && T == 0 && p0 == 0) a.edge[0] = (int16_t)(short)r[0];
            if (f == a.nframes - 1 && e0 + p0 + 4 == HW) a.edge[1] = (int16_t)(short)r[11];
        }
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
    void *d_skip, *d_E, *d_tmp, *d_carry, *d_mm, *d_spec, *d_ftail;
    TZ_TRY(tz_pool_alloc(ctx, nframes, &d_skip));
    TZ_TRY(tz_pool_alloc(ctx, sizeof(double) * 3 * nframes, &d_E));
    TZ_TRY(tz_pool_alloc(ctx, sizeof(int) * 6 * nframes, &d_mm));
    TZ_TRY(tz_pool_alloc(ctx, fe * nframes * 2, &d_tmp));
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * nblk * 3 * 2, &d_carry));
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * 3 * 2, &d_ftail));
    const int nch = (HW + 63) / 64;
    TZ_TRY(tz_pool_alloc(ctx, (size_t)nframes * 3 * nch * sizeof(unsigned long long), &d_spec));
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
        {
            const int ntiles = (nch + QT_CHUNKS - 1) / QT_CHUNKS, nchains = nframes * 3;
            void *d_width, *d_tout, *d_tin, *d_tagg, *d_swal, *d_bad;
            TZ_TRY(tz_pool_alloc(ctx, sizeof(QWidth) * nchains, &d_width));
            TZ_TRY(tz_pool_alloc(ctx, sizeof(QState) * (size_t)nchains * ntiles, &d_tout));
            TZ_TRY(tz_pool_alloc(ctx, sizeof(QState) * (size_t)nchains * ntiles, &d_tin));
            TZ_TRY(tz_pool_alloc(ctx, sizeof(double2) * (size_t)nchains * ntiles, &d_tagg));
            TZ_TRY(tz_pool_alloc(ctx, (size_t)nchains * ntiles, &d_swal));
            TZ_TRY(tz_pool_alloc(ctx, sizeof(int) * nchains, &d_bad));
            TZ_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(int) * nchains, ctx->stream));
            const bool pw = mode == TZ_MODE_PWREL;
            if (!pw)
                hipLaunchKernelGGL(k_q_width, dim3(nchains), dim3(64), 0, ctx->stream, (const double*)d_E, (const uint8_t*)d_skip, qp,
                                   (QWidth*)d_width);
#define TZ_Q_TILES(PWV, FPV)                                                                                                    \
    hipLaunchKernelGGL((k_q_tiles<PWV, FPV>), dim3(nframes * ntiles), dim3(192), 0, ctx->stream, src, orig, (const uint8_t*)d_skip, HW, \
                       ntiles, qp, (const double*)d_E, (const QWidth*)d_width, (int16_t*)d_tmp, (unsigned long long*)d_spec,         \
                       (QState*)d_tout, (double2*)d_tagg)
            if (pw && fu) TZ_Q_TILES(true, true);
            else if (pw) TZ_Q_TILES(true, false);
            else if (fu) TZ_Q_TILES(false, true);
            else TZ_Q_TILES(false, false);
#undef TZ_Q_TILES
            hipLaunchKernelGGL(k_q_chain, dim3(nchains), dim3(64), 0, ctx->stream, (const uint8_t*)d_skip, HW, ntiles,
                               (const QState*)d_tout, (const double2*)d_tagg, (QState*)d_tin, (uint8_t*)d_swal, (int16_t*)d_tmp);
            if (ntiles > 1) {
                if (fu)
                    hipLaunchKernelGGL(k_q_bstitch<true>, dim3(nchains * (ntiles - 1)), dim3(64), 0, ctx->stream, src, orig,
                                       (const uint8_t*)d_skip, HW, ntiles, qp, (const double*)d_E, (int16_t*)d_tmp,
                                       (unsigned long long*)d_spec, (const QState*)d_tin, (const uint8_t*)d_swal, (int*)d_bad);
                else
                    hipLaunchKernelGGL(k_q_bstitch<false>, dim3(nchains * (ntiles - 1)), dim3(64), 0, ctx->stream, src, orig,
                                       (const uint8_t*)d_skip, HW, ntiles, qp, (const double*)d_E, (int16_t*)d_tmp,
                                       (unsigned long long*)d_spec, (const QState*)d_tin, (const uint8_t*)d_swal, (int*)d_bad);
                if (fu)
                    hipLaunchKernelGGL(k_q_serial<true>, dim3(nchains), dim3(64), 0, ctx->stream, src, orig, (const uint8_t*)d_skip, HW, qp,
                                       (const double*)d_E, (int16_t*)d_tmp, (unsigned long long*)d_spec, (const int*)d_bad);
                else
                    hipLaunchKernelGGL(k_q_serial<false>, dim3(nchains), dim3(64), 0, ctx->stream, src, orig, (const uint8_t*)d_skip, HW, qp,
                                       (const double*)d_E, (int16_t*)d_tmp, (unsigned long long*)d_spec, (const int*)d_bad);
                if (ctx->prof_on) {   // how many chains took the fallback (a test asserts that its ramps really do)
                    std::vector<int> hb(nchains, 0);
                    TZ_HIP(ctx, hipMemcpyAsync(hb.data(), d_bad, sizeof(int) * nchains, hipMemcpyDeviceToHost, ctx->stream));
                    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    for (int v : hb) ctx->prof[TZP_QSERIAL].launches += v != 0;
                }
            }
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
        hipLaunchKernelGGL(k_q_fill, dim3(grid), dim3(QF_THREADS), 0, ctx->stream, a);
    } else {
        tz_prof_scope ps(ctx, TZP_SDELTA);
        a.zero = fu->d_zero;
        a.pred = fu->pred;
        a.orig = orig;
        a.out = fu->sym;
        a.hist = fu->d_hist;
        a.edge = fu->d_edge;
        a.apply_offset = fu->apply_offset;
        const int gsym = std::min(4 * kCUs, nframes * nblk);
        if (fu->d_hist) hipLaunchKernelGGL(k_q_fill_sym<true>, dim3(gsym), dim3(192), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_q_fill_sym<false>, dim3(gsym), dim3(192), 0, ctx->stream, a);
    }
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

int tzk_error_bound(tz_ctx* ctx, const uint8_t* orig, int16_t* diff, const uint8_t* h_skip, int nframes, int H,
                    int W, int mode, double b0, double b1) {
    if (mode < 0 || mode > 3) return tz_fail(ctx, TZ_ERR_INVALID, "unknown error-bound mode %d", mode);
    if (b0 == 0.0) return TZ_OK;                          // compress.py:24
    if (mode == TZ_MODE_ABSREL && b1 == 0.0) return TZ_OK;  // compress.py:35
    if ((mode == TZ_MODE_PWREL || mode == TZ_MODE_REL) && b0 < 0.0)
        return tz_fail(ctx, TZ_ERR_INVALID, "%s bound must be >= 0 (the reference raises on a negative one)", mode == TZ_MODE_REL ? "rel" : "pwrel");
    if (mode == TZ_MODE_ABSREL && b1 < 0.0)
        return tz_fail(ctx, TZ_ERR_INVALID, "the rel bound of absrel must be >= 0 (the reference raises on a negative one)");
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
    // a negative tolerance makes the reference assign NaN into its int array (compress.py:61 at the first element): it raises
    if ((mode == TZ_MODE_PWREL || mode == TZ_MODE_REL) && b0 < 0.0)
        return tz_fail(ctx, TZ_ERR_INVALID, "%s bound must be >= 0 (the reference raises on a negative one)", mode == TZ_MODE_REL ? "rel" : "pwrel");
    if (mode == TZ_MODE_ABSREL && b1 < 0.0)
        return tz_fail(ctx, TZ_ERR_INVALID, "the rel bound of absrel must be >= 0 (the reference raises on a negative one)");
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

// When is error_bound (compress.py:23-70) the IDENTITY on a stack of integer deltas?  (round 6)
// The greedy walk closes a run at the first element i with min(u, Du[i]) - max(l, Dl[i]) < 0, Du = d + E, Dl = d - E in
// float64, and gives the run trunc((u + l) / 2).  Deltas are integers (|d| <= 255 in the encoder, compress.py:292-314).
//  (1) With E <= 0.499 two DIFFERENT neighbours a != b always close the run: (min - max) = -|a - b| + 2E <= -0.002, three
//      orders of magnitude beyond the rounding of sums below 256; two EQUAL ones never do: fl(d + E) >= fl(d - E) for
//      E >= 0 (rounding is monotone).  So every run is a run of equal deltas d, u = fl(d + E), l = fl(d - E).
//  (2) fl(u + l) = 2d exactly, hence the run's value is d.  u and l are multiples of their ulp; unless d is a power of two
//      they share one binade (ulp q): |u + l - 2d| = |e_u + e_l| <= q with equality only if both roundings were ties in
//      the same direction -- but 2d is an even multiple of q, so when d + E lies halfway between k q and (k + 1) q, d - E
//      lies halfway between (m - k - 1) q and (m - k) q with m even: round-to-even sends the two opposite ways.  So
//      u + l - 2d is a multiple of q smaller than q: zero.  For d = 2^k (u in the upper binade, ulp 2q; l in the lower,
//      ulp q) u + l - 2d is one of -q, 0, +q; 2d + q rounds down to 2d (spacing 4q above 2^(k+1)), 2d - q is a tie
//      between 2d - 2q and 2d = 2^(k+1), whose mantissa is even.  d = 0: u + l = E - E = 0.
//      (Checked numerically over every integer d in [-255, 255] for 200,000 random and all 3- and 4-digit tolerances.)
// Per-chain tolerances (compress.py:28-48): abs |b0|; rel range * b0 with range <= 255; absrel the smaller of the two --
// so the WORST case over the chains of a job decides, before any data is looked at.  pwrel has a tolerance per ELEMENT,
// E_i = orig_i * b0 <= 255 b0: (1) holds with the elements' own tolerances (all <= 0.499), and a run of equal deltas d has
// u = min fl(d + E_i) = fl(d + min E_i), l = max fl(d - E_i) = fl(d - min E_i) (rounding is monotone) -- the same
// tolerance on both sides, so (2) applies.  BASELINE.json's cfg3 (`rel 1e-3`: E <= 0.255) is such a job: SURVEY.md section 8(d) calls it
// "effectively lossless", and the reference's own runs at such tolerances decode bit-exact (tests/golden/ref_runs4.npz).
static constexpr double kQIdentityMaxE = 0.499;

bool tz_quant_is_identity(int mode, double b0, double b1) {
    static const bool enabled = !getenv("TEZIP_QMAP") || atoi(getenv("TEZIP_QMAP")) != 0;   // (0: A/B against the general quantiser)
    double worst;
    if (mode == TZ_MODE_ABS) worst = fabs(b0);
    else if (mode == TZ_MODE_REL) worst = 255.0 * b0;
    else if (mode == TZ_MODE_ABSREL) worst = std::min(fabs(b0), 255.0 * b1);
    else if (mode == TZ_MODE_PWREL) worst = 255.0 * b0;
    else return false;
    return enabled && worst >= 0.0 && worst <= kQIdentityMaxE;   // (a NaN or a negative tolerance: the general path and its errors)
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
// ONE launch of up to SCAN_G blocks (round 3; rounds 1-2: three launches, 85 us per 62.9 M elements):
// every wave of a block owns a contiguous run of wave-tiles; the block (1) sums its runs, (2) publishes the
// sum and reads the sums of the blocks in front of it -- they run at the same time (or ran before) and do the
// same work, so they are there within a round trip --, (3) walks its runs again, scanning tile by tile with
// wave shuffles only.  6 B/element move for 4 algorithmic, as before, but without the launch gaps and the
// scan-of-sums kernel in between.  A single pass with decoupled look-back (4 B/element) was built and measured
// and is NOT faster here: a look-back hop between workgroups on different XCDs takes several microseconds
// under streaming load (and a ticket counter serialises at 12 ns per block): 170-220 us in every
// variant (scripts/microbench/scan_lookback.hip, profiles/r03/scan_lookback.txt).
// With LUT the decoder's inverse rank remap (decompress.py:31-36,236: rank -> 1600 - symbol) is
// applied to the elements as they are loaded (both times), which removes the separate k_lut pass
// over the payload (4 B/element) from tz_decode; with RECON the reconstruction follows in the same walk.
// Progress: a block waits for lower-numbered blocks only, and the workgroups of a launch start in index order
// (per XCD), so the lowest-numbered unfinished block is always running and waits for nothing unfinished -- also
// when the grid is larger than what the chip holds at once (it is: two rounds measured fastest).
static constexpr unsigned SCAN_POLL_LIMIT = 1u << 22;   // polls of ONE status word before a thread gives up (a healthy wait is
                                                        // tens of polls; 2^22 L2 round trips are seconds)
static constexpr int SCAN_EPT = 16;                 // elements per thread
static constexpr int SCAN_KEEP = 4;                 // wave-tiles of a wave's run that stay in registers between the two phases
static constexpr int SCAN_G = 4096;                 // blocks of a launch: two rounds of the 2048 the chip holds (measured at
                                                    // 62.9 M elements, fused tail: 1024 108 us, 2048 107, 4096 96, 8192 110)

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

// What phase 3 does with the scanned values: store them (tz_spatial_undelta, tz_decode_delta's producer), or -- RECON, the
// decoder's tail in one kernel -- go straight on to decompress.py:252-256,269: the thread's 16 elements lie in one frame
// (frame_elems % 16 == 0, unpadded frames), whose base is 16 key bytes or 16 truncated predictions (loaded before the
// tile's scan so that their latency hides behind its barriers); 16 bytes out.  9 B/element move instead of 13.
struct ScanRecon {
    const float4* pred;       // [frame][H*W*3] floats
    const uint8_t* key;       // key-frame stack (bytes of non-key slots unused)
    const uint8_t* key_mask;  // [frame]
    unsigned long long fe;    // elements per frame
    uint8_t* out;
    int keep_regs;            // (every launch, RECON or not) 1: a wave's run stays in registers between the phases; TEZIP_SCAN_KEEP=0: A/B
};

// status[g] = epoch << 16 | (sum of block g's chunk mod 2^16); the words of a launch carry its epoch (1..65535, the
// context counts them), so nothing has to be cleared between launches.  Inside a block every WAVE owns a contiguous run of
// `wtiles` wave-tiles (64 lanes x 16 elements), so that the walks of phases 1 and 3 need wave shuffles only -- the block
// meets at two barriers, around the exchange of sums.
static constexpr int SCAN_WT = 64 * SCAN_EPT;   // elements per wave-tile

__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
    const int lane = threadIdx.x & 63;
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned up = __shfl_up(v, s);
        if (lane >= s) v += up;
    }
    return v;
}

template <bool LUT, bool RECON>
__global__ __launch_bounds__(256) void k_scan2p(const int16_t* __restrict__ in, size_t n, int wtiles, int has_carry,
                                                int16_t carry, int vec, const int16_t* __restrict__ lut, int post_offset,
                                                unsigned* __restrict__ status, unsigned epoch, unsigned poll_epoch,
                                                unsigned poll_limit, unsigned* __restrict__ fault, int16_t* __restrict__ out,
                                                const ScanRecon rc) {
    __shared__ int16_t sl[LUT ? TZ_NBINS + 1 : 1];
    __shared__ unsigned wsum[4];
    if (LUT) {
        for (int k = threadIdx.x; k < TZ_NBINS + 1; k += 256) sl[k] = lut[k];
        __syncthreads();
    }
    const int g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t chunk0 = ((size_t)g * 4 + wv) * (size_t)wtiles * SCAN_WT;   // of this wave
    // (1) the sum of the wave's run, then of the block's.  Round 6: a wave's run of up to SCAN_KEEP tiles STAYS IN REGISTERS
    // for phase 3 (two int16 per register, 32 registers; the loops are unrolled so that nothing is indexed at run time):
    // the payload is read once instead of twice -- the second read came from HBM (counter traffic 557 MB against 440 MB
    // algorithmic at 62.9 M elements).  Launches with longer runs (more than SCAN_G * 4 * SCAN_KEEP
    // wave-tiles: 67 M elements) read twice as before.
    unsigned s = 0;
    const bool keep = wtiles <= SCAN_KEEP && rc.keep_regs;   // (uniform over the launch)
    unsigned kept[SCAN_KEEP][SCAN_EPT / 2];
    if (keep) {
#pragma unroll
        for (int t = 0; t < SCAN_KEEP; ++t) {
            const size_t base = chunk0 + (size_t)t * SCAN_WT + (size_t)lane * SCAN_EPT;
            int v[SCAN_EPT];
#pragma unroll
            for (int k = 0; k < SCAN_EPT; ++k) v[k] = 0;
            if (t < wtiles && base < n) scan_load16<LUT>(in, base, n, vec != 0, !has_carry, sl, post_offset, v);
#pragma unroll
            for (int k = 0; k < SCAN_EPT; ++k) s += (unsigned)v[k];
#pragma unroll
            for (int k = 0; k < SCAN_EPT / 2; ++k) kept[t][k] = ((unsigned)v[2 * k] & 0xFFFFu) | ((unsigned)v[2 * k + 1] << 16);
        }
    } else {
        for (int t = 0; t < wtiles; ++t) {
            const size_t base = chunk0 + (size_t)t * SCAN_WT + (size_t)lane * SCAN_EPT;
            if (base >= n) break;
            int v[SCAN_EPT];
            scan_load16<LUT>(in, base, n, vec != 0, !has_carry, sl, post_offset, v);
#pragma unroll
            for (int k = 0; k < SCAN_EPT; ++k) s += (unsigned)v[k];
        }
    }
    s = wave_scan_incl(s);
    if (lane == 63) wsum[wv] = s;
    __syncthreads();
    unsigned front = 0, tot = 0;   // of the waves in front of this one in the block; of the block
    for (int w = 0; w < 4; ++w) {
        if (w < wv) front += wsum[w];
        tot += wsum[w];
    }
    if (threadIdx.x == 0) st_store(&status[g], epoch << 16 | (tot & 0xFFFFu));
    // (2) the sums of the blocks in front: thread t takes blocks t, t + 256, ...
    // The poll is BOUNDED: the argument above rests on the dispatch order of today's hardware, which HIP does not
    // promise.  A thread that has polled one word poll_limit times gives up, says so in the context's fault word (pinned
    // host memory, read by the host at its next synchronisation: TZ_ERR_HIP) and goes on with a wrong sum -- every block
    // has published its own sum BEFORE it polls, so a launch always drains.
    unsigned mine = 0;
    for (int b = threadIdx.x; b < g; b += 256) {
        unsigned w = st_load(&status[b]), spins = 0;
        while ((w >> 16) != poll_epoch) {
            if (++spins > poll_limit) {
                *(volatile unsigned*)fault = TZ_FAULT_SCAN_POLL;
                break;
            }
            w = st_load(&status[b]);
        }
        mine += w & 0xFFFFu;
    }
    unsigned run;
    block_scan_excl(mine, &run);
    run += front;
    // (3) scan the wave's run tile by tile
    const unsigned c0 = has_carry ? (unsigned)(int)carry : 0u;
    unsigned long long fr = 0, rr = 0;   // RECON: frame and offset in it of the tile's first element
    if (RECON) {
        fr = chunk0 / rc.fe;
        rr = chunk0 - fr * rc.fe;
    }
    auto tile = [&](int t, const unsigned* kv) {   // kv: the tile's values as phase 1 left them, or nullptr: load them again
        const size_t base = chunk0 + (size_t)t * SCAN_WT + (size_t)lane * SCAN_EPT;
        int bv[SCAN_EPT];
        if (RECON) {
            unsigned long long f = fr, r = rr + (unsigned long long)lane * SCAN_EPT;
            while (r >= rc.fe) {
                r -= rc.fe;
                ++f;
            }
            rr += SCAN_WT;
            while (rr >= rc.fe) {
                rr -= rc.fe;
                ++fr;
            }
            if (base < n) {   // n is a multiple of 16 here: all 16 elements exist
                if (rc.key_mask[f]) {
                    const uint4 k = *(const uint4*)(rc.key + base);
                    const unsigned kw[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
                    for (int j = 0; j < SCAN_EPT; ++j) bv[j] = (kw[j >> 2] >> (8 * (j & 3))) & 0xff;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 p = rc.pred[base / 4 + j];
                        bv[4 * j] = (int)(p.x * 255.0f);
                        bv[4 * j + 1] = (int)(p.y * 255.0f);
                        bv[4 * j + 2] = (int)(p.z * 255.0f);
                        bv[4 * j + 3] = (int)(p.w * 255.0f);
                    }
                }
            }
        }
        int v[SCAN_EPT];
        if (kv) {
#pragma unroll
            for (int k = 0; k < SCAN_EPT / 2; ++k) {
                v[2 * k] = (int)(short)(kv[k] & 0xFFFFu);
                v[2 * k + 1] = (int)(short)(kv[k] >> 16);
            }
        } else {
            scan_load16<LUT>(in, base, n, vec != 0, !has_carry, sl, post_offset, v);
        }
        unsigned q = 0;
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) q += (unsigned)v[k];
        const unsigned incl = wave_scan_incl(q);
        unsigned pre = incl - q + run;
        run += __shfl(incl, 63);
        short r[SCAN_EPT];
#pragma unroll
        for (int k = 0; k < SCAN_EPT; ++k) {
            pre += (unsigned)v[k];
            r[k] = (short)(uint16_t)(c0 - pre);
        }
        if (RECON) {
            if (base < n) {
                unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < SCAN_EPT; ++k) {
                    int x = bv[k] - (int)r[k];
                    x = x < 0 ? 0 : (x > 255 ? 255 : x);
                    w[k >> 2] |= (unsigned)x << (8 * (k & 3));
                }
                *(uint4*)(rc.out + base) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else if (vec && base + SCAN_EPT <= n) {
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
    };
    if (keep) {
#pragma unroll
        for (int t = 0; t < SCAN_KEEP; ++t)
            if (t < wtiles && chunk0 + (size_t)t * SCAN_WT < n) tile(t, kept[t]);   // (uniform for the wave)
    } else {
        for (int t = 0; t < wtiles; ++t) {
            if (chunk0 + (size_t)t * SCAN_WT >= n) break;   // uniform for the wave
            tile(t, nullptr);
        }
    }
}

static int scan_launch(tz_ctx* ctx, const int16_t* in, size_t n, int has_carry, int16_t carry, const int16_t* h_lut2112,
                       int post_offset, int16_t* out, const ScanRecon* recon = nullptr) {
    if (n == 0) return TZ_OK;
    const size_t tiles = (n + SCAN_WT - 1) / SCAN_WT;                 // wave-tiles
    const int G = (int)std::min<size_t>(SCAN_G, (tiles + 3) / 4);
    const size_t tpb = (tiles + (size_t)G * 4 - 1) / ((size_t)G * 4);   // per wave
    if (tpb > 0x7FFFFFFFull) return tz_fail(ctx, TZ_ERR_INVALID, "inverse scan: too many elements");
    void* d_lut = nullptr;
    if (h_lut2112) {
        TZ_TRY(tz_pool_alloc(ctx, (TZ_NBINS + 1) * 2, &d_lut));
        TZ_TRY(tz_upload(ctx, d_lut, h_lut2112, (TZ_NBINS + 1) * 2));
    }
    if (!ctx->d_scan_status) TZ_HIP(ctx, hipMalloc((void**)&ctx->d_scan_status, sizeof(unsigned) * SCAN_G));
    TZ_TRY(tz_fault_word(ctx));
    tz_prof_scope ps(ctx, TZP_SCAN);
    if (ctx->scan_epoch == 0 || ctx->scan_epoch == 0xFFFFu) {   // first launch, or the epochs have gone round
        TZ_HIP(ctx, hipMemsetAsync(ctx->d_scan_status, 0, sizeof(unsigned) * SCAN_G, ctx->stream));
        ctx->scan_epoch = 0;
    }
    const unsigned epoch = ++ctx->scan_epoch;
    const int vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    static const int keep_regs = !getenv("TEZIP_SCAN_KEEP") || atoi(getenv("TEZIP_SCAN_KEEP")) != 0;   // (0: A/B against the two reads)
    ScanRecon rcv = recon ? *recon : ScanRecon{};
    rcv.keep_regs = keep_regs;
#define TZ_SCAN_LAUNCH(L, R)                                                                                              \
    hipLaunchKernelGGL((k_scan2p<L, R>), dim3(G), dim3(256), 0, ctx->stream, in, n, (int)tpb, has_carry, carry, vec,      \
                       (const int16_t*)d_lut, post_offset, ctx->d_scan_status, epoch, (epoch + ctx->scan_dbg_skew) & 0xFFFFu,       \
                       ctx->scan_dbg_limit ? ctx->scan_dbg_limit : SCAN_POLL_LIMIT, ctx->d_fault, out, rcv)
    if (recon) {
        if (h_lut2112) TZ_SCAN_LAUNCH(true, true);
        else TZ_SCAN_LAUNCH(false, true);
    } else {
        if (h_lut2112) TZ_SCAN_LAUNCH(true, false);
        else TZ_SCAN_LAUNCH(false, false);
    }
#undef TZ_SCAN_LAUNCH
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

// the decoder's tail in one launch (inverse remap when there is a table, inverse spatial delta, reconstruct) where the
// layout allows it: unpadded frames of a multiple of 16 elements, 16-byte aligned buffers.  *done = false: the caller
// runs the two separate launches.
int tzk_decode_tail_fused(tz_ctx* ctx, const int16_t* in, const int16_t* h_lut2112, int post_offset, const float* pred,
                          const uint8_t* key, const uint8_t* d_key_mask, int nframes, int H, int W, int Hp, int Wp,
                          uint8_t* out, bool* done) {
    *done = false;
    const size_t fe = (size_t)H * W * 3, n = (size_t)nframes * fe;
    if (n == 0 || H != Hp || W != Wp || fe % SCAN_EPT != 0 || !key ||
        ((((uintptr_t)in | (uintptr_t)pred | (uintptr_t)key | (uintptr_t)out) & 15) != 0))
        return TZ_OK;
    const ScanRecon rc = {(const float4*)pred, key, d_key_mask, (unsigned long long)fe, out};
    TZ_TRY(scan_launch(ctx, in, n, 0, 0, h_lut2112, post_offset, nullptr, &rc));
    *done = true;
    return TZ_OK;
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
    const int f = blockIdx.y, b = blockIdx.x;
    const double t = tz_sse_block(orig + (size_t)f * H * W * 3, pred + (size_t)f * ((size_t)Hp * Wp * 3), H, W, Hp, Wp, b, s);
    if (threadIdx.x == 0) partial[(size_t)f * nblk + b] = t;
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
