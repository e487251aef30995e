// k_wino: the per-frame convolutions of levels >= 1 under arithmetic contract TZ-PA2 (round 4).
//
// What it computes (prednet.py:254-259 gate convolutions over [r_{t-1}, e_{t-1}, up(r_{l+1})] with the LSTM update,
// prednet.py:289-291 + 274-277 A convolutions with pooling and the next level's error unit): the same convolutions as
// k_conv16, with the SAME-RESOLUTION source evaluated as Winograd F(2x2, 3x3) -- 16 multiplies per input channel and 2x2
// outputs instead of 36 -- in the fixed order that oracle/tz_oracle.c::conv3x3_wino states (TZ-PA2):
//   tile (ty, tx) = outputs (2ty + a, 2tx + b); V = B^T d B of its 4x4 input patch, columns first, in float32;
//   D[i][j] = ONE fmaf chain from 0 over the input channels in ascending order: D = fmaf(V[i][j](c), U[i][j](c, col), D)
//             with U = G g G^T evaluated on the host in float32 -- what v_mfma_f32_16x16x4_f32 computes along k;
//   Z[i][0] = (D[i][0] + D[i][1]) + D[i][2],  Z[i][1] = (D[i][1] - D[i][2]) - D[i][3];
//   y[0][b] = ((init + Z[0][b]) + Z[1][b]) + Z[2][b],  y[1][b] = ((init + Z[1][b]) - Z[2][b]) - Z[3][b],
//             init = bias, or G0 = bias + the direct TZ-PA1 chain over the constant r_{t-1} source (tz_model_prepare);
//   an upsampled source then continues every output's chain with its 4 collapsed taps (the four outputs of a tile are the
//   four parity classes over the same 2x2 half-resolution pixels), order: 16-channel block, channel quad, tap, channel.
//
// How (scripts/microbench/wino8.hip is the measured skeleton):
//   workgroup = 8 waves = 16x16 output pixels x 64 (48) columns, ONE per CU (123 KB of LDS, 256 registers per wave), two
//   waves per SIMD.  Wave (mt, ph): the 16 tiles of quadrant mt and the transform rows i = 2 ph, 2 ph + 1: 8 positions x NT
//   column tiles = 128 accumulation registers, held in AGPRs by asm MFMAs with a tied operand.  The channel quad is the
//   outermost loop: a STAGE = 16 KB of transformed weights [position][lane][4 column tiles] + one quad plane of the 18x18
//   halo patch (columns stored evens first: the 16 tiles of a wave sit on consecutive slots, 2-way bank conflicts instead of
//   4-way), DMA'd into a ring of 4 slots 3 stages ahead; per stage and wave 32 MFMAs, 8 ds_read_b128 of weights in one
//   continuous stream 4 positions ahead, 6 ds_read2_b32 of the patch and 10 packed adds for the next stage's B^T d B, one
//   counted vmcnt wait and one barrier.  The stage loops are unrolled once per ring slot; the DMA issue of a stage sits
//   behind its first four MFMAs (right behind the barrier both waves of a SIMD ran its scalar code with the matrix pipe
//   empty: 5.5 % of the kernel).  After the last channel quad the partners exchange two row sums each through LDS and form
//   their two outputs per tile; the upsampled source then runs through the same pipeline (stage = one channel quad: 4
//   classes x 4 taps of collapsed weights, 3 patch reads, no transform), the epilogue is fused.  A workgroup does `ipw`
//   column blocks of its tile one after the other (items): the stage image is [column block][stage], the DMA stream runs
//   on from one item into the next.
// Invariants nobody checks for us (DESIGN.md section 5 "hand-scheduled kernels" applies here too):
//   * every wave issues exactly 3 LDS-DMA instructions per stage, all of them before the stage's tail, so `vmcnt(3)` there
//     = "my pieces of stage s + 2 have landed" (other loads or stores in flight only make that wait longer, never shorter:
//     loads return in order);
//   * S1 and S2 are multiples of 4 (sources of 16 k channels): an item starts in ring slot 0 whatever came before it;
//   * the LDS queue is in order, so the lgkmcnt immediates below are counts of younger reads;
//   * asm MFMAs get no hazard handling from the compiler: a VALU result is never consumed by the very next MFMA (empty
//     asm statements pin the transform away from its first use), accumulators are read behind explicit wait states.
#pragma once
#include "tz_conv_kernels.hip.h"

namespace tzw {
#ifndef TZW_LEAD
#define TZW_LEAD 3
#endif
#ifndef TZW_NS
#define TZW_NS 4
#endif
// ring slots; stages in flight ahead of the one being multiplied (<= NS - 1).  Measured on one box (scripts/gpu_wino_ab.sh,
// ms of k_wino per cfg3 step), first kernel: LEAD 5: 43.64, 4: 43.62, 3: 43.21, 2: 43.34; final schedule: 3: 37.2, 2: 39.2 --
// three stages (3 us) cover the memory latency, more only fills the queues.  (The whole exchange of the output transform in
// one round, 64 KB of LDS: measured 6 % SLOWER, 46.0 against 43.3 ms; it stays one column tile a round.)
static constexpr int NS = TZW_NS, LEAD = TZW_LEAD;
static_assert(LEAD >= 2 && LEAD <= NS - 1, "the slot a stage is DMA'd into must be one nobody reads any more");
static_assert(NS == 4, "the stage loops are unrolled once per ring slot (every LDS address a constant), 4 stages = one 16-channel block");
static constexpr int WBYTES = 16 * 1024;             // weights of a stage
static constexpr int PP = 41, P1BYTES = 8 * PP * 16; // same-resolution patch plane: 8 DMA pieces of 41 slots (>= 18 x 18)
static constexpr int UP = 13, P2BYTES = 8 * UP * 16; // half-resolution patch plane: 8 pieces of 13 slots (>= 10 x 10)
static constexpr int SLOT = WBYTES + P1BYTES + P2BYTES;
static constexpr int XBYTES = 32 * 1024;             // exchange areas of the wave pairs: two of 16 KB, used in turn
static constexpr int LDS_BYTES = NS * SLOT + XBYTES; // 125,952: one workgroup per CU all the same (256 registers per wave)
static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
static constexpr int WPW = 18, WLW = 10;   // halo patch of 18 x 18 pixels, half-resolution patch of 10 x 10

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// (TZW_ABL 32, diagnostic: everything around a DMA stays, the DMA instruction itself is left out)
#if defined(TZW_ABL) && (TZW_ABL & 32)
#define TZW_DMA_INS "; "
#else
#define TZW_DMA_INS "global_load_lds_dwordx4 "
#endif
__device__ __forceinline__ void dma_lanes(const float* ubase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t" TZW_DMA_INS "%0, %1" : : "v"(voff), "s"(ubase), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma_gather(const float* ubase, unsigned voff, unsigned long long mask, unsigned lds) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" TZW_DMA_INS "%1, %2\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "s"(ubase), "s"(lds), "s"(mask) : "memory");
}
template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_read2(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
    f32x4 v;
#if defined(TZW_ABL) && (TZW_ABL & 4)
    v = (f32x4){1.f, 2.f, 3.f, 4.f};
    asm volatile("" : "+v"(v), "+v"(addr));
    return v;
#endif
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// in-place accumulation in an AGPR tuple (see the header: the register allocator must not get a say)
__device__ __forceinline__ void mfma_acc(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the first multiplication of a chain: the accumulator starts at the literal 0 (fma(a, b, +0), the same operation as on a zeroed
// register -- without the 128 register writes per item that zeroing was)
__device__ __forceinline__ void mfma_first(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
// float32 add / subtract the compiler cannot pair into v_pk_add_f32 (it did: 16 packed adds + 12 moves to line their
// operands up, 28 instructions where 20 do -- and a packed fp32 instruction takes the SIMD longer than a plain one, all of
// it time the matrix pipe does not get).  Same IEEE results.
__device__ __forceinline__ float fsub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fadd(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// every wave issues 3 LDS-DMA instructions per stage: "my pieces of all but the N youngest stages have landed"
template <int N>
__device__ __forceinline__ void wait_vm_stages() {
    static_assert(N >= 0 && N <= 3, "immediate of s_waitcnt");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (N == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}
#ifndef TZW_PK
#define TZW_PK 1
#endif
template <int K>
__device__ __forceinline__ void read_rows(unsigned ad, f32x2 (&d)[3][2]) {   // row k of the wave's three patch rows
#if TZW_PK
    d[K][0] = lds_read2<72 * K, 72 * K + 36>(ad);       // (d0, d1): columns 0 and 1 (evens first: column 1 sits 9 slots on)
    d[K][1] = lds_read2<72 * K + 4, 72 * K + 40>(ad);   // (d2, d3)
#else
    d[K][0] = lds_read2<72 * K, 72 * K + 4>(ad);        // (d0, d2)
    d[K][1] = lds_read2<72 * K + 36, 72 * K + 40>(ad);  // (d1, d3)
#endif
}
// packed float32 add / subtract with the operand halves chosen per instruction (same IEEE results as two scalar ones):
//   pk_sub(a, b)  = (a.lo - b.lo, a.hi - b.hi);  pk_add likewise
//   pk_cross(a, b) = (a.hi + b.lo, b.lo - a.hi)   -- the two middle columns of B^T d: (d1 + d2, d2 - d1) from (d0, d1), (d2, d3)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_cross(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
}  // namespace tzw

#define TZW_TIE8(X) asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]))
#define TZW_TIE6(X) asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]))
#define TZW_TIED(d) asm volatile("" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[2][0]), "+v"(d[2][1]))
#define TZW_TIEU(u) asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]))

#ifdef TZW_STAMPS
// diagnostic build: s_memrealtime (100 MHz) at six points of a workgroup's life, wave 0, per (kernel shape slot, workgroup)
__device__ unsigned long long tzw_stamps[8][4096][16];   // [0..6] real time at the phase boundaries, [7] where, [8 + K] the core clock counter at stamp K
#define TZW_STAMP(K) if (tid == 0 && stamp_id < 4096) { tzw_stamps[(a.ncb & 3) | (EPI == EPI_LSTM ? 4 : 0)][stamp_id][K] = __builtin_amdgcn_s_memrealtime(); tzw_stamps[(a.ncb & 3) | (EPI == EPI_LSTM ? 4 : 0)][stamp_id][8 + (K)] = __builtin_amdgcn_s_memtime(); }
#else
#define TZW_STAMP(K)
#endif
// EPI: EPI_LSTM (NT = 4: columns [i | f | g | o] x 16 channels), EPI_POOL_ERR (NT = 3 or 4), EPI_RAW (NT = 4).
// src[0]: the same-resolution source (multiple of 16 channels); src[1] (UPS): the half-resolution source.
// A gate convolution can also run as TWO launches (round 5, tz_prednet.hip "E-part ahead"; launches that cannot fill the
// chip): <4, EPI_RAW, false> over the same-resolution source alone, which leaves every output's chain as it stands behind
// the output transform in out0 ([pixel][column], per batch item), and <4, EPI_LSTM, true, NOSAME> which starts from those
// values (a.init with a.init_nstride per item) and walks the upsampled source only.  Same chains, same order, the
// intermediate is a float32 either way: same bits.  Both walk a PART of the stage image: a.wino_stride / a.wino_first.
template <int NT, int EPI, bool UPS, bool NOSAME = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wino(const ConvArgs a) {
    static_assert(!NOSAME || UPS, "a launch without the same-resolution phase walks the upsampled source");
    using namespace tzw;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wv & 3, ph = wv >> 2;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // a workgroup walks over `ipw` column blocks of ITS tile one after the other (items): geometry, zero fill and the first
    // DMA round trip happen once, the stream of stages runs on across the items (the first stages of the next item are
    // issued under the last ones of the current)
    const int ipw = a.ipw, ncbg = a.ncb / ipw;
    const int cb0 = (bid % ncbg) * ipw;
    bid /= ncbg;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = bid % ntiles, n = bid / ntiles;
    const int ty0 = (tile / a.tiles_x) * 16, tx0 = (tile % a.tiles_x) * 16;
    const int g = lane >> 4, r = lane & 15;
#ifdef TZW_STAMPS
    int stamp_id = blockIdx.x * a.ipw;   // one record per item
#endif
    TZW_STAMP(0)
#ifdef TZW_STAMPS
    if (tid == 0 && stamp_id < 4096) {   // where this workgroup runs: (XCC, SE, SH, CU) -- slot 7
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        for (int q = 0; q < a.ipw && stamp_id + q < 4096; ++q) tzw_stamps[(a.ncb & 3) | (EPI == EPI_LSTM ? 4 : 0)][stamp_id + q][7] = ((unsigned long long)(xcc & 15) << 32) | hw;
    }
#endif
    const unsigned sbase = lds_addr(smem);
    const int S1 = NOSAME ? 0 : a.src[0].C >> 2;          // stages of the same-resolution source
    const int S = S1 + (UPS ? a.src[1].C >> 2 : 0);       // ... and of the upsampled one
    const int SI = a.wino_stride ? a.wino_stride : S;     // stages per column block in the image (a launch may walk a part)

    // ---- DMA geometry, once per workgroup.  Same-resolution plane: piece wv = slots 41 wv ..; slot = row * 18 + column
    // with the columns stored evens first.  Half-resolution plane: piece wv = slots 13 wv .., slot = row * 10 + column.
    unsigned poff, uoff = 0;
    unsigned long long pmask, umask = 0;
    {
        const int slot = wv * PP + lane;
        const int py = slot / WPW, xs = slot - py * WPW;
        const int px = xs < WPW / 2 ? 2 * xs : 2 * (xs - WPW / 2) + 1;
        const int yy = ty0 - 1 + py, xx = tx0 - 1 + px;
        const bool ok = lane < PP && slot < WPW * WPW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        poff = ok ? 4u * (unsigned)((yy * a.W + xx) * a.src[0].pstride) : 0u;
        pmask = __ballot(ok);
    }
    if (UPS) {
        const int slot = wv * UP + lane;
        const int Y = slot / WLW, X = slot - Y * WLW;
        const int ly = (ty0 >> 1) - 1 + Y, lx = (tx0 >> 1) - 1 + X;
        const bool ok = lane < UP && slot < WLW * WLW && ly >= 0 && ly < (a.H >> 1) && lx >= 0 && lx < (a.W >> 1);
        uoff = ok ? 4u * (unsigned)((ly * (a.W >> 1) + lx) * a.src[1].pstride) : 0u;
        umask = __ballot(ok);
    }
    const unsigned lane16 = lane * 16;
    // ---- the DMA stream.  Stage si goes into ring slot si mod 4: 2 KB of weights + one patch piece per wave = 3 DMA
    // instructions.  The stage loops are unrolled once per ring slot, so every LDS address below is a constant; the global
    // addresses are running scalar pointers (the first form recomputed them from the stage number: ~40 scalar instructions
    // per stage and wave, 3.4 % of the kernel -- TZW_ABL 32 against 8).
    // (the stage image is [column block][stage]: the weight pointer simply runs on from one item into the next)
    const float* wp = a.Wwino + (((long long)cb0 * SI + a.wino_first) * 16 + 2 * wv) * 256;   // weights of the next stage to issue
    const float* xp0 = a.src[0].p + (long long)n * a.src[0].nstride;                       // ... its quad of the same-resolution source
    const float* xp1 = UPS ? a.src[1].p + (long long)n * a.src[1].nstride : nullptr;       // ... of the upsampled one
    int si = 0;                                                                            // ... its number
    const unsigned wvw = (unsigned)(2 * wv) * 1024, wvp = WBYTES + (unsigned)wv * (PP * 16), wvu = WBYTES + P1BYTES + (unsigned)wv * (UP * 16);
#define TZW_ISSUE_W0(KI) dma_lanes(wp, lane16, sbase + (unsigned)(KI) * SLOT + wvw);
#define TZW_ISSUE_W1(KI) dma_lanes(wp + 256, lane16, sbase + (unsigned)(KI) * SLOT + wvw + 1024);
#define TZW_ISSUE_P(KI)                                                                                                     \
    {                                                                                                                       \
        const unsigned so = sbase + (unsigned)(KI) * SLOT;                                                                  \
        if (!UPS || si < S1) {                                                                                              \
            dma_gather(xp0, poff, pmask, so + wvp);                                                                         \
            xp0 += 4;                                                                                                       \
        } else {                                                                                                            \
            dma_gather(xp1, uoff, umask, so + wvu);                                                                         \
            xp1 += 4;                                                                                                       \
        }                                                                                                                   \
        wp += 16 * 256;                                                                                                     \
        if (++si == S) {   /* on to the next item: the next column block's stage 0, the sources from their first quad */   \
            si = 0;                                                                                                         \
            wp += (SI - S) * (16 * 256);   /* (a launch over a part of the stages skips the rest of the block's) */          \
            xp0 -= 4 * S1;                                                                                                  \
            if (UPS) xp1 -= 4 * (S - S1);                                                                                   \
        }                                                                                                                   \
    }
#define TZW_ISSUE_TO(KI) { TZW_ISSUE_W0(KI) TZW_ISSUE_W1(KI) TZW_ISSUE_P(KI) }
    // zero the patch areas once: the slots of out-of-image pixels are never written by the DMA (a tile whose halo lies
    // inside the image has none)
    const bool interior = ty0 >= 2 && tx0 >= 2 && ty0 + 18 <= a.H && tx0 + 18 <= a.W;
    if (!interior)
    for (int i = tid; i < NS * ((P1BYTES + P2BYTES) / 16); i += 512) {
        const int sl = i / ((P1BYTES + P2BYTES) / 16), o = i - sl * ((P1BYTES + P2BYTES) / 16);
        *(f32x4*)(smem + sl * SLOT + WBYTES + o * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int nlead = LEAD;   // (S >= 4: sources are multiples of 16 channels)
    TZW_ISSUE_TO(0)
    TZW_ISSUE_TO(1)
    if (LEAD > 2) TZW_ISSUE_TO(2)

    // A address of this lane: tile r = (tyl, txl) of quadrant mt, channel g of the quad, first of the wave's three rows
    const int tyl = r >> 2, txl = r & 3;
    const unsigned abase = WBYTES + (unsigned)(((8 * (mt >> 1) + 2 * tyl + ph) * WPW + 4 * (mt & 1) + txl) * 16 + 4 * g);
    const unsigned ubase = WBYTES + P1BYTES + (unsigned)(((4 * (mt >> 1) + tyl + ph) * WLW + 4 * (mt & 1) + txl) * 16 + 4 * g);
    const unsigned bbase = (unsigned)(8 * ph) * 1024 + lane16;
    // LDS read addresses, one register set for the whole kernel: weights of slots 0, 1 from wbL and of slots 2, 3 from wbH (the
    // 16-bit immediate of ds_read_b128 covers a slot and a half), patch planes per slot (ds_read2 immediates are 8 bits)
    const unsigned wbL = sbase + bbase, wbH = wbL + 2 * SLOT;
    unsigned adA[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) adA[k] = sbase + (unsigned)k * SLOT + abase;
    auto read_d = [&](unsigned ad, f32x2 (&d)[3][2]) {
        read_rows<0>(ad, d);
        read_rows<1>(ad, d);
        read_rows<2>(ad, d);
    };
    // B^T d B for the wave's two transform rows (same association as the oracle): columns first, then rows
    auto transform = [&](const f32x2 (&d)[3][2], float (&V)[8]) {
#if TZW_PK
        // 10 packed instructions instead of 20: per row T1 = (t0, t3) = (d0, d1) - (d2, d3), T2 = (t1, t2) = (d1 + d2, d2 - d1);
        // then the rows the same way on the pairs
        f32x2 T1[3], T2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T1[k] = pk_sub(d[k][0], d[k][1]);
            T2[k] = pk_cross(d[k][0], d[k][1]);
        }
        f32x2 a03, a12, b03, b12;   // (V[.][0], V[.][3]), (V[.][1], V[.][2]) of the wave's two transform rows
        if (ph == 0) {   // rows 0, 1, 2 of the patch: V[0] = t0 - t2, V[1] = t1 + t2
            a03 = pk_sub(T1[0], T1[2]);
            a12 = pk_sub(T2[0], T2[2]);
            b03 = pk_add(T1[1], T1[2]);
            b12 = pk_add(T2[1], T2[2]);
        } else {         // rows 1, 2, 3: V[2] = t2 - t1, V[3] = t1 - t3
            a03 = pk_sub(T1[1], T1[0]);
            a12 = pk_sub(T2[1], T2[0]);
            b03 = pk_sub(T1[0], T1[2]);
            b12 = pk_sub(T2[0], T2[2]);
        }
        V[0] = a03[0]; V[3] = a03[1]; V[1] = a12[0]; V[2] = a12[1];
        V[4] = b03[0]; V[7] = b03[1]; V[5] = b12[0]; V[6] = b12[1];
        return;
#endif
        float t[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d0 = d[k][0][0], d2 = TZW_PK ? d[k][1][0] : d[k][0][1], d1 = TZW_PK ? d[k][0][1] : d[k][1][0], d3 = d[k][1][1];
            t[k][0] = fsub(d0, d2);
            t[k][1] = fadd(d1, d2);
            t[k][2] = fsub(d2, d1);
            t[k][3] = fsub(d1, d3);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ph == 0) {   // rows 0, 1, 2 of the patch: V[0] = t0 - t2, V[1] = t1 + t2
                V[j] = fsub(t[0][j], t[2][j]);
                V[4 + j] = fadd(t[1][j], t[2][j]);
            } else {         // rows 1, 2, 3: V[2] = t2 - t1, V[3] = t1 - t3
                V[j] = fsub(t[1][j], t[0][j]);
                V[4 + j] = fsub(t[0][j], t[2][j]);
            }
        }
    };

    // stages 0 and 1 landed for everyone (from the second item on the last stage of the item before has seen to that)
    if (nlead == LEAD) wait_vm_stages<LEAD - 2>();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < ipw; ++it) {
    const int cb = cb0 + it;
    const bool more = it + 1 < ipw;   // (uniform) the DMA stream runs on into another item
    // (the pixel offsets of the output transform and of the epilogue do not depend on the item: hoisted out of this loop they
    // sat in 16 registers through every stage loop and were spilled to scratch; they are made to depend on this)
    int per_item = 0;
    asm volatile("" : "+s"(per_item));
    f32x4 D[8][4];   // (first written by the first stage of the item: mfma_first)
    TZW_STAMP(1)
    float V0[8], V1[8];
    f32x4 B[8];
    if (!NOSAME) {
        {
            f32x2 d[3][2];
            read_d(adA[0], d);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[2][0]), "+v"(d[2][1]) : : "memory");
            transform(d, V0);
            TZW_TIE8(V0);
        }
        // the weight reads run as one continuous stream, four positions ahead, across the stage boundaries
        B[0] = lds_read16<0>(wbL);
        B[1] = lds_read16<1024>(wbL);
        B[2] = lds_read16<2048>(wbL);
        B[3] = lds_read16<3072>(wbL);
    }
    int s = 0;   // (its ring slot is s mod 4 = the position in the unrolled loop body)
    // One stage of a wave: 8 positions x NT column tiles.  The LDS queue is in order, so every wait is a count:
    //   pos 0, 1  wait 3 (the three younger weight reads); behind pos 1 the patch reads of the NEXT stage (NA instructions)
    //   pos 2-5   wait 3 + NA
    //   pos 6     wait 3: the patch reads are done -> (phase 1) transform for the next stage
    //   pos 4-7   issue B[0..3] of the next stage from the next slot
// TZW_ABL (diagnostic builds, WRONG results, scripts/gpu_wino_ab.sh): 1 no barrier in the loops; 2 no patch reads / transform;
// 4 no weight reads; 8 no DMA in the loops; 16 (with 8) every stage loop twice
#ifndef TZW_ABL
#define TZW_ABL 0
#endif
#define TZW_WAIT(P, N) if (!(TZW_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(B[P]) : : "memory")
#ifndef TZW_ISSUE_AT
#define TZW_ISSUE_AT 1
#endif
#define TZW_ISSUE_HERE(K, AT)                                                                                               \
        if (TZW_ISSUE_AT == (AT) && !(TZW_ABL & 8) && (s + LEAD < S || more)) TZW_ISSUE_TO(((K) + LEAD) % NS)                         \
        if (TZW_ISSUE_AT == 10 + (AT) - 1 && !(TZW_ABL & 8) && (s + LEAD < S || more)) TZW_ISSUE_W0(((K) + LEAD) % NS)                \
        if (TZW_ISSUE_AT == 10 + (AT) - 2 && !(TZW_ABL & 8) && (s + LEAD < S || more)) TZW_ISSUE_W1(((K) + LEAD) % NS)                \
        if (TZW_ISSUE_AT == 10 + (AT) - 3 && !(TZW_ABL & 8) && (s + LEAD < S || more)) TZW_ISSUE_P(((K) + LEAD) % NS)
#define TZW_STAGE_HEAD(K)                                                                                                   \
        TZW_ISSUE_HERE(K, 0)                                                                                                \
        constexpr int KN = ((K) + 1) % NS;                                    /* the next stage's slot */                   \
        const unsigned wb = ((K) & 2) ? wbH : wbL, wn = (KN & 2) ? wbH : wbL;                                               \
        constexpr int WO = ((K) & 1) * SLOT, WN = (KN & 1) * SLOT;            /* ... folded into the read immediates */
#define TZW_STAGE_TAIL                                                                                                      \
        if (s + LEAD < S || more) wait_vm_stages<LEAD - 2>();                                                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
        if (!(TZW_ABL & 1)) __builtin_amdgcn_s_barrier();                                                                   \
        ++s;
#define TZW_MMF(F, ACC, P, AV)                                                                                              \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) F(ACC[t], AV, B[P][t]);
#define TZW_MM(ACC, P, AV) TZW_MMF(mfma_acc, ACC, P, AV)
#define TZW_STAGE1(K, VC, VN) TZW_STAGE1_(K, VC, VN, mfma_acc)
#define TZW_STAGE1_(K, VC, VN, F)                                                                                           \
    {                                                                                                                       \
        TZW_STAGE_HEAD(K)                                                                                                   \
        f32x2 d[3][2];                                                                                                      \
        TZW_WAIT(0, 3); TZW_MMF(F, D[0], 0, VC[0]) B[4] = lds_read16<WO + 4 * 1024>(wb);                                        \
        TZW_ISSUE_HERE(K, 1)                                                                                                \
        TZW_WAIT(1, 3); TZW_MMF(F, D[1], 1, VC[1]) B[5] = lds_read16<WO + 5 * 1024>(wb);                                        \
        TZW_ISSUE_HERE(K, 2)                                                                                                \
        if (!(TZW_ABL & 2)) read_d(adA[KN], d);                                                                             \
        TZW_WAIT(2, 9); TZW_MMF(F, D[2], 2, VC[2]) B[6] = lds_read16<WO + 6 * 1024>(wb);                                        \
        TZW_ISSUE_HERE(K, 3)                                                                                                \
        TZW_WAIT(3, 9); TZW_MMF(F, D[3], 3, VC[3]) B[7] = lds_read16<WO + 7 * 1024>(wb);                                        \
        TZW_ISSUE_HERE(K, 4)                                                                                                \
        TZW_WAIT(4, 9); TZW_MMF(F, D[4], 4, VC[4]) B[0] = lds_read16<WN>(wn);                                                   \
        TZW_WAIT(5, 9); TZW_MMF(F, D[5], 5, VC[5]) B[1] = lds_read16<WN + 1024>(wn);                                            \
        TZW_WAIT(6, 3); TZW_MMF(F, D[6], 6, VC[6]) B[2] = lds_read16<WN + 2048>(wn);                                            \
        /* the patch reads have arrived (empty asm: nothing is computed with d before here) and the transform is done */   \
        /* HERE, not sunk to its first use right in front of an asm MFMA                                              */   \
        if (!(TZW_ABL & 2)) { TZW_TIED(d); transform(d, VN); }                                                              \
        else { _Pragma("unroll") for (int q = 0; q < 8; ++q) VN[q] = VC[q]; }                                               \
        TZW_TIE8(VN);                                                                                                       \
        TZW_WAIT(7, 3); TZW_MMF(F, D[7], 7, VC[7]) B[3] = lds_read16<WN + 3072>(wn);                                            \
        TZW_STAGE_TAIL                                                                                                      \
    }
    if (!NOSAME) {
#pragma unroll 1
    for (int rep = 0; rep < ((TZW_ABL & 16) ? 2 : 1); ++rep) {   // (ablation 16, with 8: every loop twice -> time per stage)
    if (TZW_ABL & 16) s = 0;
    if (rep == 0) {    // the first block of an item: its first stage starts the chains
        TZW_STAGE1_(0, V0, V1, mfma_first)
        TZW_STAGE1(1, V1, V0)
        TZW_STAGE1(2, V0, V1)
        TZW_STAGE1(3, V1, V0)
    }
#pragma unroll 1
    while (s < S1) {   // one 16-channel block = four stages = once round the ring
        TZW_STAGE1(0, V0, V1)
        TZW_STAGE1(1, V1, V0)
        TZW_STAGE1(2, V0, V1)
        TZW_STAGE1(3, V1, V0)
    }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs have written their accumulators
    }
    TZW_STAMP(2)

    // ---- output transform, oracle order, one column tile per round (everything of a round dies with it: the first form
    // computed all row sums ahead and the compiler parked them in AGPRs -- 1100 vector instructions per wave with the matrix
    // pipes idle; this one has ~450).
    //   y[0][b] = ((init + Z[0][b]) + Z[1][b]) + Z[2][b] belongs to the wave with rows 0, 1 and needs Z[2] of its partner;
    //   y[1][b] = ((init + Z[1][b]) - Z[2][b]) - Z[3][b] belongs to the wave with rows 2, 3 and needs Z[1]:
    // exchanged through LDS, two areas of 16 KB in turn, so that one barrier a round is enough (a wave that writes area
    // t & 1 again in round t + 2 has passed the barrier of round t + 1, behind which nobody reads round t's data any more)
    f32x4 Y[2][4];
    {
        // accumulator start of this wave's outputs: row a = ph of the tiles, columns b = 0, 1; register e <-> tile (g, e)
        f32x4 in0[NT], in1[NT];
        {
            const int y = ty0 + 8 * (mt >> 1) + 2 * g + ph + per_item, x0 = tx0 + 8 * (mt & 1);
            if (a.init && !(TZW_ABL & 64)) {
                const float* initn = a.init + (long long)n * a.init_nstride;   // (0 for the per-model G0; a split launch's start values are per item)
                if (ty0 + 16 <= a.H && tx0 + 16 <= a.W) {   // (uniform) the whole tile inside the image: one lane offset, uniform steps
                    const float* ip = initn + (unsigned)((y * a.W + x0) * a.ncols + cb * (16 * NT) + r);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* ie = ip + (unsigned)(2 * e * a.ncols);
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            in0[t][e] = ie[16 * t];
                            in1[t][e] = ie[a.ncols + 16 * t];
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int col = cb * (16 * NT) + 16 * t + r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int x = x0 + 2 * e;
                            const bool ok = y < a.H && x < a.W;   // (W even wherever tiles are cut: x + 1 < W too)
                            const float* ip = initn + ((long long)(ok ? y * a.W + x : 0)) * a.ncols + col;
                            in0[t][e] = ip[0];
                            in1[t][e] = ip[x + 1 < a.W && ok ? a.ncols : 0];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float b = a.bias[cb * (16 * NT) + 16 * t + r];
                    in0[t] = in1[t] = (f32x4){b, b, b, b};
                }
            }
        }
        f32x4* xo = (f32x4*)(smem + NS * SLOT) + (wv * 2) * 64 + lane;
        const f32x4* xi = (const f32x4*)(smem + NS * SLOT) + ((wv ^ 4) * 2) * 64 + lane;
        if (NOSAME) {   // the chains as the launch over the same-resolution source left them
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                Y[0][t] = in0[t];
                Y[1][t] = in1[t];
            }
            // into their accumulation registers HERE, two wait states ahead of any MFMA that takes them as C (the asm MFMAs get
            // no hazard handling: a v_accvgpr_write sunk to right in front of the first stage would be read too early)
            static_assert(!NOSAME || NT == 4, "the split launch exists for gate convolutions");
            asm volatile("" : "+a"(Y[0][0]), "+a"(Y[0][1]), "+a"(Y[0][2]), "+a"(Y[0][3]), "+a"(Y[1][0]), "+a"(Y[1][1]), "+a"(Y[1][2]), "+a"(Y[1][3]));
            asm volatile("s_nop 1" ::: "memory");
        } else
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            constexpr int XA = XBYTES / 2 / 16;   // (f32x4 units) the second area
            const int xa = (t & 1) * XA;
            // (the accumulators of this round are read HERE: hoisted, they and the start values went through ~300 register
            // moves between VGPRs and AGPRs)
            asm volatile("" : "+a"(D[0][t]), "+a"(D[1][t]), "+a"(D[2][t]), "+a"(D[3][t]), "+a"(D[4][t]), "+a"(D[5][t]), "+a"(D[6][t]), "+a"(D[7][t]));
            // row sums of the wave's two transform rows
            const f32x4 za0 = (D[0][t] + D[1][t]) + D[2][t], za1 = (D[1][t] - D[2][t]) - D[3][t];
            const f32x4 zb0 = (D[4][t] + D[5][t]) + D[6][t], zb1 = (D[5][t] - D[6][t]) - D[7][t];
            if (ph == 0) {   // (uniform)
                xo[xa] = zb0;
                xo[xa + 64] = zb1;
            } else {
                xo[xa] = za0;
                xo[xa + 64] = za1;
            }
            if (!(TZW_ABL & 128)) __syncthreads();
            const f32x4 p0 = xi[xa], p1 = xi[xa + 64];
            if (ph == 0) {
                Y[0][t] = ((in0[t] + za0) + zb0) + p0;
                Y[1][t] = ((in1[t] + za1) + zb1) + p1;
            } else {
                Y[0][t] = ((in0[t] + p0) - za0) - zb0;
                Y[1][t] = ((in1[t] + p1) - za1) - zb1;
            }
        }
    }

    TZW_STAMP(3)
    // ---- the upsampled source: every output's chain goes on with its collapsed taps.  Stage = one channel quad; the
    // wave's classes are (a = ph, b = 0, 1); weight sets in consumption order lp = 2 tap + b; A fragments = the 2 x 3
    // half-resolution pixels around the tile (rows ph + tpy, columns b + tpx)
    if (UPS) {
        unsigned udA[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) udA[k] = sbase + (unsigned)k * SLOT + ubase;
        auto read_u = [&](unsigned ad, f32x2 (&u)[3]) {
            u[0] = lds_read2<0, 4>(ad);             // (row 0, columns 0, 1)
            u[1] = lds_read2<8, WLW * 4>(ad);        // (row 0, column 2), (row 1, column 0)
            u[2] = lds_read2<WLW * 4 + 4, WLW * 4 + 8>(ad);   // (row 1, columns 1, 2)
        };
        // (the fragments stay where the reads put them -- pairs; a copy into scalars was a register move per stage behind an
        // asm read, which tests/test_build_guard.py does not let through)
        f32x2 A0[3], A1[3];
        {
            read_u(udA[0], A0);   // (S1 is a multiple of 4: the first stage of this phase sits in slot 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A0[0]), "+v"(A0[1]), "+v"(A0[2]) : : "memory");
            // (the weight stream was drained by that wait: restart it)
            B[0] = lds_read16<0>(wbL);
            B[1] = lds_read16<1024>(wbL);
            B[2] = lds_read16<2048>(wbL);
            B[3] = lds_read16<3072>(wbL);
        }
        // lp = 2 tap + b: tap = (tpy, tpx) -> A[3 tpy + b + tpx]  (A[q] = fragment pair q >> 1, half q & 1)
#define TZW_STAGE2(K, AC, AN)                                                                                               \
    {                                                                                                                       \
        TZW_STAGE_HEAD(K)                                                                                                   \
        TZW_WAIT(0, 3); TZW_MM(Y[0], 0, AC[0][0]) B[4] = lds_read16<WO + 4 * 1024>(wb);                                     \
        TZW_ISSUE_HERE(K, 1)                                                                                                \
        TZW_WAIT(1, 3); TZW_MM(Y[1], 1, AC[0][1]) B[5] = lds_read16<WO + 5 * 1024>(wb);                                     \
        TZW_ISSUE_HERE(K, 2)                                                                                                \
        if (!(TZW_ABL & 2)) read_u(udA[KN], AN);                                                                            \
        TZW_WAIT(2, 6); TZW_MM(Y[0], 2, AC[0][1]) B[6] = lds_read16<WO + 6 * 1024>(wb);                                     \
        TZW_ISSUE_HERE(K, 3)                                                                                                \
        TZW_WAIT(3, 6); TZW_MM(Y[1], 3, AC[1][0]) B[7] = lds_read16<WO + 7 * 1024>(wb);                                     \
        TZW_ISSUE_HERE(K, 4)                                                                                                \
        TZW_WAIT(4, 6); TZW_MM(Y[0], 4, AC[1][1]) B[0] = lds_read16<WN>(wn);                                                \
        TZW_WAIT(5, 6); TZW_MM(Y[1], 5, AC[2][0]) B[1] = lds_read16<WN + 1024>(wn);                                         \
        TZW_WAIT(6, 3); TZW_MM(Y[0], 6, AC[2][0]) B[2] = lds_read16<WN + 2048>(wn);                                         \
        /* the patch reads have arrived (they are older than the three weight reads that wait left outstanding) */          \
        if (TZW_ABL & 2) { AN[0] = AC[0]; AN[1] = AC[1]; AN[2] = AC[2]; }                                                   \
        TZW_TIEU(AN);                                                                                                       \
        TZW_WAIT(7, 3); TZW_MM(Y[1], 7, AC[2][1]) B[3] = lds_read16<WN + 3072>(wn);                                         \
        TZW_STAGE_TAIL                                                                                                      \
    }
#pragma unroll 1
        for (int rep = 0; rep < ((TZW_ABL & 16) ? 2 : 1); ++rep) {
        if (TZW_ABL & 16) s = S1;
#pragma unroll 1
        while (s < S) {
            TZW_STAGE2(0, A0, A1)
            TZW_STAGE2(1, A1, A0)
            TZW_STAGE2(2, A0, A1)
            TZW_STAGE2(3, A1, A0)
        }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads issued for a stage past the end
    TZW_STAMP(4)
#undef TZW_STAGE2
#undef TZW_STAGE1
#undef TZW_STAGE1_
#undef TZW_MMF
#undef TZW_MM
#undef TZW_STAGE_TAIL
#undef TZW_STAGE_HEAD
#undef TZW_ISSUE_TO
#undef TZW_ISSUE_W0
#undef TZW_ISSUE_W1
#undef TZW_ISSUE_P
#undef TZW_ISSUE_HERE
#undef TZW_WAIT

    // ---- epilogues.  This wave's outputs: pixel row a = ph of its 16 tiles, columns b = 0, 1: Y[b][column tile][e],
    // accumulator row 4 g + e = tile (tyl = g, txl = e)
    const int oy = ty0 + 8 * (mt >> 1) + 2 * g + ph + per_item;
    if (EPI == EPI_RAW) {
        float* on = a.out0 + (long long)n * a.out0_nstride;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = tx0 + 8 * (mt & 1) + 2 * e + b;
                if (oy < a.H && x < a.W) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) on[((long long)oy * a.W + x) * a.ncols + cb * (16 * NT) + 16 * t + r] = Y[b][t][e];
                }
            }
    } else if (EPI == EPI_LSTM) {
        // columns of this block: [i | f | g | o] x 16 channels of channel group cb (prednet.py:255-259):
        // c = f * c_prev + i * g ; r = o * tanh(c)
        const int ch = cb * 16 + r, R = a.Cout;
        float* o0 = a.out0 + (long long)n * a.out0_nstride;
        float* o1 = a.out1 ? a.out1 + (long long)n * a.out1_nstride : nullptr;
        if (ty0 + 16 <= a.H && tx0 + 16 <= a.W && !(TZW_ABL & 512)) {
            // (uniform) the whole tile inside the image: ONE lane offset for the eight outputs of a lane, the steps between
            // them uniform -- scalar base + 32-bit lane offset addressing, no per-output predicates (the general form below
            // spends ~25 instructions per output on 64-bit addresses and exec masks; in a serial section, with the matrix
            // pipes idle, instructions are what costs)
            const unsigned voff = 4u * (unsigned)((oy * a.W + tx0 + 8 * (mt & 1)) * R + ch);
            const char* pc = (const char*)a.aux;
            char* p0 = (char*)o0;
            char* p1 = (char*)o1;
            float cpv[2][4];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cpv[b][e] = a.aux && !(TZW_ABL & 256) ? *(const float*)(pc + (size_t)(2 * e + b) * R * 4 + voff) : 0.0f;
#ifdef TZW_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TZW_STAMP(6)
#endif
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gi = tz_hard_sigmoid(Y[b][0][e]);
                    const float gf = tz_hard_sigmoid(Y[b][1 % NT][e]);
                    const float gg = tz_tanh(Y[b][2 % NT][e]);
                    const float go = tz_hard_sigmoid(Y[b][3 % NT][e]);
                    const float t1 = gf * cpv[b][e];
                    const float t2 = gi * gg;
                    const float c = t1 + t2;
                    const float rr = go * tz_tanh(c);
                    *(float*)(p0 + (size_t)(2 * e + b) * R * 4 + voff) = rr;
                    if (o1) *(float*)(p1 + (size_t)(2 * e + b) * R * 4 + voff) = c;
                }
        } else {
        long long pix[2][4];
        bool ok[2][4];
        float cp[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = tx0 + 8 * (mt & 1) + 2 * e + b;
                ok[b][e] = oy < a.H && x < a.W;
                pix[b][e] = ok[b][e] ? (long long)oy * a.W + x : 0;
                // (Asking for these ahead of the output transform through asm loads -- to hide their round trip -- is NOT
                // safe: the compiler is free to reuse an asm load's destination register before the data lands, and did:
                // the late write then hit an address register, a memory fault in the bench.  Plain loads, at their use.)
                cp[b][e] = a.aux && !(TZW_ABL & 256) ? a.aux[pix[b][e] * R + ch] : 0.0f;
            }
#ifdef TZW_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TZW_STAMP(6)
#endif
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (TZW_ABL & 512) {
                    if (ok[b][e]) {
                        o0[pix[b][e] * R + ch] = Y[b][0][e] + Y[b][1 % NT][e] + cp[b][e];
                        if (o1) o1[pix[b][e] * R + ch] = Y[b][2 % NT][e] + Y[b][3 % NT][e];
                    }
                    continue;
                }
                const float gi = tz_hard_sigmoid(Y[b][0][e]);
                const float gf = tz_hard_sigmoid(Y[b][1 % NT][e]);
                const float gg = tz_tanh(Y[b][2 % NT][e]);
                const float go = tz_hard_sigmoid(Y[b][3 % NT][e]);
                const float t1 = gf * cp[b][e];
                const float t2 = gi * gg;
                const float c = t1 + t2;
                const float rr = go * tz_tanh(c);
                if (ok[b][e]) {
                    o0[pix[b][e] * R + ch] = rr;
                    if (o1) o1[pix[b][e] * R + ch] = c;
                }
            }
        }
    } else if (EPI == EPI_POOL_ERR) {
        // prednet.py:289-291 then 274-277 of the next level: A = maxpool2x2(relu(conv)) -- the pooling window IS the tile --,
        // e = [relu(Ahat0 - A), relu(A - Ahat0)] at the pooled resolution.  The wave holds one row of every window; the
        // partner the other: wave ph = 0 finishes column tiles 0, 1, wave ph = 1 the rest
        const int H2 = a.H >> 1, W2 = a.W >> 1, C = a.Cout;
        float* o = a.out0 + (long long)n * a.out0_nstride;
        f32x4 m[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = tz_relu(Y[0][t][e]), v1 = tz_relu(Y[1][t][e]);
                m[t][e] = v1 > v0 ? v1 : v0;
            }
        // (the exchange area the last round of the output transform did NOT use: a partner may still be reading that one)
        constexpr int XP = (NT & 1) * (XBYTES / 2 / 16);
        f32x4* xo = (f32x4*)(smem + NS * SLOT) + XP + (wv * 2) * 64 + lane;
        const f32x4* xi = (const f32x4*)(smem + NS * SLOT) + XP + ((wv ^ 4) * 2) * 64 + lane;
        // (no run-time indices into m[]: it has to stay in registers)
        xo[0] = ph == 0 ? m[2] : m[0];                       // the column tiles the partner finishes
        xo[64] = ph == 0 ? m[NT - 1] : m[1];                 // (NT = 3: the second one of wave 0 is not used)
        __syncthreads();
        const f32x4 q0 = xi[0], q1 = xi[64];
        const int yp = (ty0 >> 1) + 4 * (mt >> 1) + g + per_item;
        if (ty0 + 16 <= a.H && tx0 + 16 <= a.W && (cb + 1) * (16 * NT) <= C && !(TZW_ABL & 512)) {
            // (uniform) the whole tile inside the image and the whole column block inside the stack: ONE lane offset per
            // array for the outputs of a lane, uniform steps between them -- scalar base + 32-bit lane offset addressing, no
            // per-output predicates (what the LSTM epilogue got in round 4; the general form below spends ~25 instructions
            // per output on 64-bit addresses and exec masks, in a serial section with the matrix pipes idle)
            const unsigned pix = (unsigned)(yp * W2 + (tx0 >> 1) + 4 * (mt & 1));
            const unsigned colw = (unsigned)(cb * (16 * NT) + r);
            const char* ph_ = (const char*)a.aux;
            char* po = (char*)o;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (ph == 1 && 2 + k >= NT) break;
                const int t = (ph == 0 ? 0 : 2) + k;
                const f32x4 mine = ph == 0 ? m[k] : m[(2 + k) % NT], other = k == 0 ? q0 : q1;
                const unsigned hoff = 4u * (pix * (unsigned)C + colw + 16u * t), ooff = 4u * (pix * 2u * (unsigned)C + colw + 16u * t);
                float hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = *(const float*)(ph_ + (size_t)e * C * 4 + hoff);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float mm = other[e] > mine[e] ? other[e] : mine[e];
                    const float d1 = hv[e] - mm, d2 = mm - hv[e];
                    *(float*)(po + (size_t)e * C * 8 + ooff) = tz_relu(d1);
                    *(float*)(po + (size_t)e * C * 8 + (size_t)C * 4 + ooff) = tz_relu(d2);
                }
            }
        } else
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (ph == 1 && 2 + k >= NT) break;               // wave 1 finishes column tiles 2 .. NT - 1, wave 0 tiles 0, 1
            const int t = (ph == 0 ? 0 : 2) + k;
            const f32x4 mine = ph == 0 ? m[k] : m[(2 + k) % NT], other = k == 0 ? q0 : q1;
            const int ch = cb * (16 * NT) + 16 * t + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int xp = (tx0 >> 1) + 4 * (mt & 1) + e;
                const bool okp = yp < H2 && xp < W2 && ch < C;
                const long long pp = okp ? (long long)yp * W2 + xp : 0;
                const float h = a.aux[okp ? pp * C + ch : 0];
                const float mm = other[e] > mine[e] ? other[e] : mine[e];
                const float d1 = h - mm, d2 = mm - h;
                if (okp) {
                    o[pp * 2 * C + ch] = tz_relu(d1);
                    o[pp * 2 * C + C + ch] = tz_relu(d2);
                }
            }
        }
    }
#ifdef TZW_STAMPS
    TZW_STAMP(5)
    if (more) {
        ++stamp_id;
        TZW_STAMP(0)   // (of the next item)
    }
#endif
    }   // items
}

// k_wino_ref: the same TZ-PA2 convolutions (and epilogues) as k_wino, written the plain way -- one thread per (tile, column
// group), every chain a loop of fmaf in the order the contract states -- as the device-side cross-check of the hand-scheduled
// kernel (tz_set_conv_impl(0), as k_conv3x3 is for k_conv16).  It reads the same packed stage image (pack_wino), so a packing
// error cannot hide between the two; ~50x slower.
//   EPI_LSTM: thread = (tile, channel of the column block): its four gate columns one after the other, then the cell update
//   EPI_POOL_ERR / EPI_RAW: thread = (tile, column)
template <int NT, int EPI, bool UPS>
__global__ __launch_bounds__(256) void k_wino_ref(const ConvArgs a) {
    const int TX = (a.W + 1) / 2, TY = (a.H + 1) / 2;
    const int gcols = EPI == EPI_LSTM ? 16 : 16 * NT;           // threads per tile and column block
    const long long nthreads = (long long)TX * TY * a.ncb * gcols;
    const int n = blockIdx.y;
    const int S1 = a.src[0].C >> 2, C0 = a.src[0].C;
    const int SR = S1 + (UPS ? a.src[1].C >> 2 : 0);            // stages of a column block in the image
    const float* x0 = a.src[0].p + (long long)n * a.src[0].nstride;
    const float* x1 = UPS ? a.src[1].p + (long long)n * a.src[1].nstride : nullptr;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nthreads; id += (long long)gridDim.x * blockDim.x) {
        const int jc = (int)(id % gcols);
        long long q = id / gcols;
        const int cb = (int)(q % a.ncb);
        q /= a.ncb;
        const int tx = (int)(q % TX), ty = (int)(q / TX);
        float Yg[4][4];   // [gate or 0][output 2 a + b]
        const int ngate = EPI == EPI_LSTM ? 4 : 1;
        for (int gt = 0; gt < ngate; ++gt) {
            const int nt = EPI == EPI_LSTM ? gt : jc >> 4, j = jc & 15;
            const int col = cb * (16 * NT) + 16 * nt + j;
            float D[16];
            for (int p = 0; p < 16; ++p) D[p] = 0.0f;
            for (int c = 0; c < C0; ++c) {
                float d[4][4], t[4][4];
                for (int rr = 0; rr < 4; ++rr)
                    for (int cc = 0; cc < 4; ++cc) {
                        const int yy = 2 * ty - 1 + rr, xx = 2 * tx - 1 + cc;
                        d[rr][cc] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? x0[((long long)yy * a.W + xx) * a.src[0].pstride + c] : 0.0f;
                    }
                for (int rr = 0; rr < 4; ++rr) {
                    t[rr][0] = d[rr][0] - d[rr][2];
                    t[rr][1] = d[rr][1] + d[rr][2];
                    t[rr][2] = d[rr][2] - d[rr][1];
                    t[rr][3] = d[rr][1] - d[rr][3];
                }
                const float* wq = a.Wwino + ((((long long)cb * SR + (c >> 2)) * 16) * 64 + (c & 3) * 16 + j) * 4 + nt;
                for (int jj = 0; jj < 4; ++jj) {
                    const float v0 = t[0][jj] - t[2][jj], v1 = t[1][jj] + t[2][jj], v2 = t[2][jj] - t[1][jj], v3 = t[1][jj] - t[3][jj];
                    D[0 + jj] = fmaf(v0, wq[(0 + jj) * 256], D[0 + jj]);
                    D[4 + jj] = fmaf(v1, wq[(4 + jj) * 256], D[4 + jj]);
                    D[8 + jj] = fmaf(v2, wq[(8 + jj) * 256], D[8 + jj]);
                    D[12 + jj] = fmaf(v3, wq[(12 + jj) * 256], D[12 + jj]);
                }
            }
            float z0[4], z1[4];
            for (int i = 0; i < 4; ++i) {
                z0[i] = (D[4 * i] + D[4 * i + 1]) + D[4 * i + 2];
                z1[i] = (D[4 * i + 1] - D[4 * i + 2]) - D[4 * i + 3];
            }
            float in[4];
            for (int o = 0; o < 4; ++o) {
                const int y = 2 * ty + (o >> 1), x = 2 * tx + (o & 1);
                in[o] = a.init ? a.init[((long long)(y < a.H && x < a.W ? y * a.W + x : 0)) * a.ncols + col] : a.bias[col];
            }
            float* Y = Yg[gt];
            Y[0] = ((in[0] + z0[0]) + z0[1]) + z0[2];
            Y[1] = ((in[1] + z1[0]) + z1[1]) + z1[2];
            Y[2] = ((in[2] + z0[1]) - z0[2]) - z0[3];
            Y[3] = ((in[3] + z1[1]) - z1[2]) - z1[3];
            if (UPS) {   // channel quads ascending, the 4 taps of a quad, its 4 channels
                const int H2 = a.H >> 1, W2 = a.W >> 1, C1 = a.src[1].C;
                for (int o = 0; o < 4; ++o) {
                    const int ay = o >> 1, bx = o & 1;
                    float acc = Y[o];
                    for (int c0 = 0; c0 < C1; c0 += 4)
                        for (int tp = 0; tp < 4; ++tp) {
                            const int ly = ty - 1 + ay + (tp >> 1), lx = tx - 1 + bx + (tp & 1);
                            const bool inside = ly >= 0 && ly < H2 && lx >= 0 && lx < W2;
                            const float* wq = a.Wwino + ((((long long)cb * SR + S1 + (c0 >> 2)) * 16 + 8 * ay + 2 * tp + bx) * 64 + j) * 4 + nt;
                            for (int k = 0; k < 4; ++k) {
                                const float xv = inside ? x1[((long long)ly * W2 + lx) * a.src[1].pstride + c0 + k] : 0.0f;
                                acc = fmaf(xv, wq[k * 64], acc);
                            }
                        }
                    Y[o] = acc;
                }
            }
        }
        if (EPI == EPI_RAW) {
            float* on = a.out0 + (long long)n * a.out0_nstride;
            for (int o = 0; o < 4; ++o) {
                const int y = 2 * ty + (o >> 1), x = 2 * tx + (o & 1);
                if (y < a.H && x < a.W) on[((long long)y * a.W + x) * a.ncols + cb * (16 * NT) + jc] = Yg[0][o];
            }
        } else if (EPI == EPI_LSTM) {
            const int ch = cb * 16 + jc, R = a.Cout;
            float* o0 = a.out0 + (long long)n * a.out0_nstride;
            float* o1 = a.out1 ? a.out1 + (long long)n * a.out1_nstride : nullptr;
            for (int o = 0; o < 4; ++o) {
                const int y = 2 * ty + (o >> 1), x = 2 * tx + (o & 1);
                if (y >= a.H || x >= a.W) continue;
                const long long pix = (long long)y * a.W + x;
                const float cp = a.aux ? a.aux[pix * R + ch] : 0.0f;
                const float gi = tz_hard_sigmoid(Yg[0][o]), gf = tz_hard_sigmoid(Yg[1][o]), gg = tz_tanh(Yg[2][o]), go = tz_hard_sigmoid(Yg[3][o]);
                const float t1 = gf * cp;
                const float t2 = gi * gg;
                const float c = t1 + t2;
                const float rr = go * tz_tanh(c);
                o0[pix * R + ch] = rr;
                if (o1) o1[pix * R + ch] = c;
            }
        } else {   // EPI_POOL_ERR: the pooling window is the tile
            const int H2 = a.H >> 1, W2 = a.W >> 1, C = a.Cout, ch = cb * (16 * NT) + jc;
            if (ty < H2 && tx < W2 && ch < C) {
                float m = tz_relu(Yg[0][0]);
                for (int o = 1; o < 4; ++o) {
                    const float t = tz_relu(Yg[0][o]);
                    if (t > m) m = t;
                }
                const long long pp = (long long)ty * W2 + tx;
                const float h = a.aux[pp * C + ch];
                const float d1 = h - m, d2 = m - h;
                float* o = a.out0 + (long long)n * a.out0_nstride;
                o[pp * 2 * C + ch] = tz_relu(d1);
                o[pp * 2 * C + C + ch] = tz_relu(d2);
            }
        }
    }
}
