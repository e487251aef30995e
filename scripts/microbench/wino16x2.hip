// Winograd F(2x2, 3x3) K loop, all 16 transform positions in ONE wave, TWO column tiles (round 5; a go / no-go for round 6).
// wino16.hip (one wave per SIMD, 16 positions x 4 column tiles = 256 accumulation registers) lost to wino8.hip (pairs of
// waves sharing a tile) because a lone wave per SIMD pays for every instruction between its MFMAs.  This form keeps the 16
// positions together but halves the columns: 16 positions x 2 column tiles = 128 accumulation registers, 256 registers per
// wave, so TWO workgroups of four waves (one 16x16-pixel tile x 32 columns each) share a CU.  What that buys on paper:
//   * the output transform is in registers (no exchange between partner waves, no exchange area in LDS);
//   * the two workgroups of a CU are not coupled by a barrier: one's prologue / output transform / epilogue runs under the
//     other's stage loop (k_wino's serial sections are 5-17 % of a workgroup's life with the matrix pipes idle);
// and what it costs: a stage's weights are 8 KB per workgroup but the 5.2 KB patch plane is now staged once per 32 columns
// (26.9 KB per stage time and CU instead of 21.6), twice the workgroups, and 16 patch reads + 32 adds per 32 MFMAs.
//   stage = one k-step (4 channels): 8 KB of transformed weights [position pair][lane][2 positions x 2 column tiles] +
//   one quad plane of the 18x18 halo patch; ring of NS slots, LEAD stages ahead.
// Checks the result bit for bit against the oracle's statement of the same chains (oracle/tz_oracle.c conv3x3_wino).
//   hipcc --offload-arch=gfx950 -O3 wino16x2.hip -o wino16x2.bin && ./wino16x2.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef W_NS
#define W_NS 5
#endif
#ifndef W_LEAD
#define W_LEAD 4
#endif
static constexpr int NS = W_NS, LEAD = W_LEAD;
static_assert(LEAD >= 2 && LEAD <= NS - 1, "ring");
static constexpr int WBYTES = 8 * 1024;           // weights of a stage (32 columns)
static constexpr int PP = 41;                     // patch slots per DMA piece (8 pieces >= 324 slots)
static constexpr int PBYTES = 8 * PP * 16;        // 5248
static constexpr int SLOT = WBYTES + PBYTES;      // 13440
static constexpr int PW = 18;

// at a stage's tail: everything but the youngest LEAD - 2 stages (4 DMAs each) has landed = stage s + 2 is in LDS
#define VMW_STR2(x) #x
#define VMW_STR(x) VMW_STR2(x)
#if W_LEAD == 2
#define VMWAIT_TAIL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#elif W_LEAD == 3
#define VMWAIT_TAIL() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")
#elif W_LEAD == 4
#define VMWAIT_TAIL() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#elif W_LEAD == 5
#define VMWAIT_TAIL() asm volatile("s_waitcnt vmcnt(12)" ::: "memory")
#else
#error "W_LEAD 2..5"
#endif
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// 1 KB of weights: lane i's 16 bytes from ubase + 16 i to lds + 16 i
__device__ __forceinline__ void dma_lanes(const float* ubase, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(ubase), "s"(lds) : "memory");
}
// gather: lanes in `mask` fetch 16 bytes from ubase + voff_i to lds + 16 i
__device__ __forceinline__ void dma_gather(const float* ubase, unsigned voff, unsigned long long mask, unsigned lds) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
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
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// The accumulators are ALL 256 accumulation registers of the wave: in asm with a tied "+a" operand, so that the register
// allocator cannot decide to accumulate out of place and shuttle tiles through VGPRs (it did, as soon as the loop was
// unrolled twice: 1400 v_accvgpr moves per iteration).  Hazards are ours then: the same accumulator comes round again
// 64 MFMAs later, A / B operands are written by LDS reads (waited for) or by VALU instructions many issues earlier, and
// the accumulators are read only after the loop, behind explicit wait states.
__device__ __forceinline__ void mfma_acc(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int N>
__device__ __forceinline__ void wait_lgkm(f32x4& v) {
    if (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
    else if (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v) : : "memory");
    else if (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(v) : : "memory");
    else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(v) : : "memory");
}

struct Args {
    const float* x;      // [n][H][W][C]
    const float* wimg;   // [stage = C/4][ncb][8 position pairs][64 lanes][2 positions x 2 column tiles]
    const float* init;   // [ncols] (bias)
    float* out;          // [n][H][W][ncols]
    int H, W, C, ncols, ncb, tiles_x, tiles_y;
    int streams;         // 0: no DMA in the loop (ceiling of the loop itself)
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wino(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = bid % ntiles, n = bid / ntiles;
    const int ty0 = (tile / a.tiles_x) * 16, tx0 = (tile % a.tiles_x) * 16;
    const int g = lane >> 4, r = lane & 15;
    const unsigned sbase = lds_addr(smem);
    const int S = a.C >> 2;   // stages

    // ---- patch DMA geometry, once: pieces wv and wv + 4, lane = slot inside the piece
    unsigned poff[2];
    unsigned long long pmask[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = (wv + 4 * j) * PP + lane;
        const int py = slot / PW, px = slot - py * PW;
        const int yy = ty0 - 1 + py, xx = tx0 - 1 + px;
        const bool ok = lane < PP && slot < PW * PW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        poff[j] = ok ? 4u * (unsigned)((yy * a.W + xx) * a.C) : 0u;
        pmask[j] = __ballot(ok);
    }
    const float* xn = a.x + (long long)n * a.H * a.W * a.C;
    const unsigned lane16 = lane * 16;
    auto issue = [&](int s) {   // stage s into its ring slot: 2 KB of weights + two patch pieces per wave = 4 DMA
        const unsigned slot = sbase + (unsigned)(s % NS) * SLOT;
        const float* w = a.wimg + (((long long)s * a.ncb + cb) * 8 + 2 * wv) * 256;
#pragma unroll
        for (int p = 0; p < 2; ++p) dma_lanes(w + p * 256, lane16, slot + (2 * wv + p) * 1024);
        const float* xs = xn + 4 * s;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma_gather(xs, poff[j], pmask[j], slot + WBYTES + (wv + 4 * j) * PP * 16);
    };
    // zero the patch areas once: out-of-image slots are never written by the DMA
    for (int i = tid; i < NS * (PBYTES / 16); i += 256) {
        const int sl = i / (PBYTES / 16), o = i - sl * (PBYTES / 16);
        *(f32x4*)(smem + sl * SLOT + WBYTES + o * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int nlead = LEAD < S ? LEAD : S;
    for (int s = 0; s < nlead; ++s) issue(s);

    f32x4 D[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) D[p][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A address of this lane: tile r of the wave's quadrant, channel g of the quad
    const int tyl = r >> 2, txl = r & 3;
    const unsigned abase = WBYTES + (unsigned)(((8 * (wv >> 1) + 2 * tyl) * PW + 8 * (wv & 1) + 2 * txl) * 16 + 4 * g);
    auto read_d_slot = [&](int slot_, f32x2 (&d)[4][2]) {
        const unsigned ad = sbase + (unsigned)slot_ * SLOT + abase;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            // (template arguments must be constants: spelled out)
            if (rr == 0) { d[0][0] = lds_read2<0, 4>(ad); d[0][1] = lds_read2<8, 12>(ad); }
            if (rr == 1) { d[1][0] = lds_read2<PW * 4, PW * 4 + 4>(ad); d[1][1] = lds_read2<PW * 4 + 8, PW * 4 + 12>(ad); }
            if (rr == 2) { d[2][0] = lds_read2<2 * PW * 4, 2 * PW * 4 + 4>(ad); d[2][1] = lds_read2<2 * PW * 4 + 8, 2 * PW * 4 + 12>(ad); }
            if (rr == 3) { d[3][0] = lds_read2<3 * PW * 4, 3 * PW * 4 + 4>(ad); d[3][1] = lds_read2<3 * PW * 4 + 8, 3 * PW * 4 + 12>(ad); }
        }
    };
    auto read_d = [&](int s_, f32x2 (&d)[4][2]) { read_d_slot(s_ % NS, d); };
    auto transform = [&](const f32x2 (&d)[4][2], float (&V)[16]) {
        float t[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float d0 = d[rr][0][0], d1 = d[rr][0][1], d2 = d[rr][1][0], d3 = d[rr][1][1];
            t[rr][0] = d0 - d2;
            t[rr][1] = d1 + d2;
            t[rr][2] = d2 - d1;
            t[rr][3] = d1 - d3;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            V[0 * 4 + j] = t[0][j] - t[2][j];
            V[1 * 4 + j] = t[1][j] + t[2][j];
            V[2 * 4 + j] = t[2][j] - t[1][j];
            V[3 * 4 + j] = t[1][j] - t[3][j];
        }
    };

    // stages 0 and 1 landed for everyone
    if (nlead == LEAD) VMWAIT_TAIL();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // MODE bits (ablation builds, wrong results): 1 no barrier in the loop; 2 no patch reads / transform; 4 no weight reads
    float V0[16], V1[16];
    {
        f32x2 d[4][2];
        read_d(0, d);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[3][0]), "+v"(d[3][1]) : : "memory");
        transform(d, V0);
        asm volatile("" : "+v"(V0[0]), "+v"(V0[1]), "+v"(V0[2]), "+v"(V0[3]), "+v"(V0[4]), "+v"(V0[5]), "+v"(V0[6]), "+v"(V0[7]),
                     "+v"(V0[8]), "+v"(V0[9]), "+v"(V0[10]), "+v"(V0[11]), "+v"(V0[12]), "+v"(V0[13]), "+v"(V0[14]), "+v"(V0[15]));
    }
    f32x4 B[8];   // B[q] = positions 2 q, 2 q + 1 x column tiles 0, 1
    {   // the weight reads run as one continuous stream, four position pairs ahead, across the stage boundaries
        const unsigned wb = sbase + lane16;
        B[0] = lds_read16<0>(wb);
        B[1] = lds_read16<1024>(wb);
        B[2] = lds_read16<2048>(wb);
        B[3] = lds_read16<3072>(wb);
    }
    int slot = 0;   // ring slot of the current stage
    // One stage: 8 position pairs x (2 positions x 2 column tiles).  In-order LDS queue, so every wait is a count:
    //   q 0, 1   wait 3 (the three younger weight reads); behind q 1 the 8 patch reads of stage s + 1
    //   q 2-5    wait 11 (3 weight reads + the 8 patch reads)
    //   q 6      wait 3: the patch reads are in front of B[6]'s successors, i.e. done -> transform for stage s + 1
    //   q 4-7    issue B[0..3] of stage s + 1 from the next slot
#define WPOS(VC, Q, WAITN, NEXT)                                                                                            \
    {                                                                                                                       \
        if (!(MODE & 4)) {                                                                                                  \
            if (WAITN == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(B[Q]) : : "memory");                                 \
            else asm volatile("s_waitcnt lgkmcnt(11)" : "+v"(B[Q]) : : "memory");                                           \
        }                                                                                                                   \
        mfma_acc(D[2 * (Q)][0], VC[2 * (Q)], B[Q][0]);                                                                      \
        mfma_acc(D[2 * (Q)][1], VC[2 * (Q)], B[Q][1]);                                                                      \
        mfma_acc(D[2 * (Q) + 1][0], VC[2 * (Q) + 1], B[Q][2]);                                                              \
        mfma_acc(D[2 * (Q) + 1][1], VC[2 * (Q) + 1], B[Q][3]);                                                              \
        if (!(MODE & 4)) { NEXT; }                                                                                          \
    }
#define WSTAGE(VC, VN)                                                                                                      \
    {                                                                                                                       \
        if (a.streams && s + LEAD < S) issue(s + LEAD);                                                                     \
        const unsigned wb = sbase + (unsigned)slot * SLOT + lane16;                                                         \
        const int nslot = slot + 1 == NS ? 0 : slot + 1;                                                                    \
        const unsigned wn = sbase + (unsigned)nslot * SLOT + lane16;                                                        \
        f32x2 d[4][2];                                                                                                      \
        WPOS(VC, 0, 3, B[4] = lds_read16<4 * 1024>(wb))                                                                     \
        WPOS(VC, 1, 3, B[5] = lds_read16<5 * 1024>(wb); if (!(MODE & 2)) read_d_slot(nslot, d))                             \
        WPOS(VC, 2, (MODE & 2 ? 3 : 11), B[6] = lds_read16<6 * 1024>(wb))                                                   \
        WPOS(VC, 3, (MODE & 2 ? 3 : 11), B[7] = lds_read16<7 * 1024>(wb))                                                   \
        WPOS(VC, 4, (MODE & 2 ? 3 : 11), B[0] = lds_read16<0>(wn))                                                          \
        WPOS(VC, 5, (MODE & 2 ? 3 : 11), B[1] = lds_read16<1024>(wn))                                                       \
        WPOS(VC, 6, 3, B[2] = lds_read16<2048>(wn))                                                                         \
        /* the patch reads were in front of B[6]'s successors in the queue: arrived */                                      \
        if (!(MODE & 2)) asm volatile("" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[3][0]), "+v"(d[3][1])); \
        if (!(MODE & 2)) { transform(d, VN); }                                                                              \
        else { _Pragma("unroll") for (int q = 0; q < 16; ++q) VN[q] = VC[q]; }                                              \
        /* done HERE: sunk to its first use the transform would sit right in front of an asm MFMA (no hazard handling) */   \
        asm volatile("" : "+v"(VN[0]), "+v"(VN[1]), "+v"(VN[2]), "+v"(VN[3]), "+v"(VN[4]), "+v"(VN[5]), "+v"(VN[6]), "+v"(VN[7]),   \
                     "+v"(VN[8]), "+v"(VN[9]), "+v"(VN[10]), "+v"(VN[11]), "+v"(VN[12]), "+v"(VN[13]), "+v"(VN[14]), "+v"(VN[15])); \
        WPOS(VC, 7, 3, B[3] = lds_read16<3072>(wn))                                                                         \
        if (a.streams && s + LEAD < S) VMWAIT_TAIL();                                                                       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
        if (!(MODE & 1)) __builtin_amdgcn_s_barrier();                                                                      \
        slot = nslot;                                                                                                       \
    }
#pragma unroll 1
    for (int s = 0; s < S; s += 2) {   // S is a multiple of 4 (16-channel blocks)
        WSTAGE(V0, V1)
        ++s;
        WSTAGE(V1, V0)
        --s;
    }
#undef WSTAGE
#undef WPOS
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the reads issued for a stage past the end; the last MFMAs
    // ---- output transform, oracle order, then store
    const float* init = a.init + cb * 32 + r;
    f32x4 Y[4][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float b = init[16 * t];
        f32x4 z0[4], z1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            z0[i] = (D[4 * i + 0][t] + D[4 * i + 1][t]) + D[4 * i + 2][t];
            z1[i] = (D[4 * i + 1][t] - D[4 * i + 2][t]) - D[4 * i + 3][t];
        }
        const f32x4 bb = (f32x4){b, b, b, b};
        Y[0][t] = ((bb + z0[0]) + z0[1]) + z0[2];
        Y[1][t] = ((bb + z1[0]) + z1[1]) + z1[2];
        Y[2][t] = ((bb + z0[1]) - z0[2]) - z0[3];
        Y[3][t] = ((bb + z1[1]) - z1[2]) - z1[3];
    }
    float* on = a.out + (long long)n * a.H * a.W * a.ncols + cb * 32 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // accumulator row 4 g + e = tile (tyl = g, txl = e)
            const int y = ty0 + 8 * (wv >> 1) + 2 * g + (q >> 1), x = tx0 + 8 * (wv & 1) + 2 * e + (q & 1);
            if (y < a.H && x < a.W) {
#pragma unroll
                for (int t = 0; t < 2; ++t) on[((long long)y * a.W + x) * a.ncols + 16 * t] = Y[q][t][e];
            }
        }
}

// ------------------------------------------------------------------------------------------------ host
static void wino_u(const float* Wt, int Cin, int Cout, int ci, int co, float U[4][4]) {   // oracle/tz_oracle.c wino_u
    float g[3][3], w[4][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) g[r][c] = Wt[((size_t)(r * 3 + c) * Cin + ci) * Cout + co];
    for (int c = 0; c < 3; ++c) {
        float s_ = g[0][c] + g[2][c];
        w[0][c] = g[0][c];
        w[1][c] = 0.5f * (s_ + g[1][c]);
        w[2][c] = 0.5f * (s_ - g[1][c]);
        w[3][c] = g[2][c];
    }
    for (int i = 0; i < 4; ++i) {
        float s_ = w[i][0] + w[i][2];
        U[i][0] = w[i][0];
        U[i][1] = 0.5f * (s_ + w[i][1]);
        U[i][2] = 0.5f * (s_ - w[i][1]);
        U[i][3] = w[i][2];
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
static int run(const char* name, int N, int H, int W, int C, int ncols, int reps) {
    const int ncb = ncols / 32, S = C / 4;
    std::vector<float> x((size_t)N * H * W * C), wt((size_t)9 * C * ncols), bias(ncols);
    unsigned s = 12345u + C * 7 + H;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto& v : x) { float t = rnd(); v = t < 0.4f ? 0.0f : t; }                 // relu-like
    for (auto& v : wt) v = (rnd() - 0.5f) * 0.1f;
    for (auto& v : bias) v = rnd() - 0.5f;
    std::vector<float> U((size_t)16 * C * ncols);                                    // [pos][c][col]
    for (int c = 0; c < C; ++c)
        for (int co = 0; co < ncols; ++co) {
            float u[4][4];
            wino_u(wt.data(), C, ncols, c, co, u);
            for (int p = 0; p < 16; ++p) U[((size_t)p * C + c) * ncols + co] = u[p >> 2][p & 3];
        }
    std::vector<float> wimg((size_t)S * ncb * 8 * 256);
    for (int st = 0; st < S; ++st)
        for (int cb = 0; cb < ncb; ++cb)
            for (int q = 0; q < 8; ++q)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e)
                        wimg[((((size_t)st * ncb + cb) * 8 + q) * 64 + l) * 4 + e] =
                            U[((size_t)(2 * q + (e >> 1)) * C + 4 * st + (l >> 4)) * ncols + cb * 32 + 16 * (e & 1) + (l & 15)];
    float *dx, *dw, *db, *dout;
    const size_t nout = (size_t)N * H * W * ncols;
    CK(hipMalloc(&dx, x.size() * 4));
    CK(hipMalloc(&dw, wimg.size() * 4));
    CK(hipMalloc(&db, bias.size() * 4));
    CK(hipMalloc(&dout, nout * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, wimg.data(), wimg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, nout * 4));
    Args a{dx, dw, db, dout, H, W, C, ncols, ncb, (W + 15) / 16, (H + 15) / 16, 1};
    const int grid = N * a.tiles_x * a.tiles_y * ncb;
    const size_t lds = (size_t)NS * SLOT;
    CK(hipFuncSetAttribute((const void*)k_wino<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    for (int streams = 1; streams >= 0; --streams) {
        a.streams = streams;
        for (int it = 0; it < reps + 1; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_wino<MODE>, dim3(grid), dim3(256), lds, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best[streams]) best[streams] = ms;
        }
        if (streams == 1 && MODE == 0) {   // check against the oracle's chains
            std::vector<float> out(nout);
            CK(hipMemcpy(out.data(), dout, nout * 4, hipMemcpyDeviceToHost));
            int bad = 0, checked = 0;
            unsigned q = 99u;
            auto ri = [&](int m) { q = q * 1664525u + 1013904223u; return (int)((q >> 10) % (unsigned)m); };
            for (int k = 0; k < 400; ++k) {
                int n = ri(N), ty = k < 8 ? 0 : (k < 16 ? (H + 1) / 2 - 1 : ri((H + 1) / 2)), tx = k < 4 ? 0 : (k < 12 ? (W + 1) / 2 - 1 : ri((W + 1) / 2));
                int co = ri(ncols);
                float Dp[16];
                for (int p = 0; p < 16; ++p) Dp[p] = 0.f;
                for (int c = 0; c < C; ++c) {
                    float d[4][4], t[4][4], Vv[4][4];
                    for (int rr = 0; rr < 4; ++rr)
                        for (int cc = 0; cc < 4; ++cc) {
                            int yy = 2 * ty - 1 + rr, xx = 2 * tx - 1 + cc;
                            d[rr][cc] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((size_t)n * H + yy) * W + xx) * C + c] : 0.f;
                        }
                    for (int rr = 0; rr < 4; ++rr) {
                        t[rr][0] = d[rr][0] - d[rr][2];
                        t[rr][1] = d[rr][1] + d[rr][2];
                        t[rr][2] = d[rr][2] - d[rr][1];
                        t[rr][3] = d[rr][1] - d[rr][3];
                    }
                    for (int j = 0; j < 4; ++j) {
                        Vv[0][j] = t[0][j] - t[2][j];
                        Vv[1][j] = t[1][j] + t[2][j];
                        Vv[2][j] = t[2][j] - t[1][j];
                        Vv[3][j] = t[1][j] - t[3][j];
                    }
                    for (int p = 0; p < 16; ++p) Dp[p] = fmaf(Vv[p >> 2][p & 3], U[((size_t)p * C + c) * ncols + co], Dp[p]);
                }
                float z0[4], z1[4];
                for (int i = 0; i < 4; ++i) {
                    z0[i] = (Dp[4 * i] + Dp[4 * i + 1]) + Dp[4 * i + 2];
                    z1[i] = (Dp[4 * i + 1] - Dp[4 * i + 2]) - Dp[4 * i + 3];
                }
                float b = bias[co];
                float Y[4] = {((b + z0[0]) + z0[1]) + z0[2], ((b + z1[0]) + z1[1]) + z1[2], ((b + z0[1]) - z0[2]) - z0[3], ((b + z1[1]) - z1[2]) - z1[3]};
                for (int qq = 0; qq < 4; ++qq) {
                    int y = 2 * ty + (qq >> 1), xx = 2 * tx + (qq & 1);
                    if (y >= H || xx >= W) continue;
                    float got = out[(((size_t)n * H + y) * W + xx) * ncols + co];
                    ++checked;
                    if (memcmp(&got, &Y[qq], 4) != 0) {
                        if (bad < 5) printf("  MISMATCH n=%d y=%d x=%d co=%d got %.9g want %.9g\n", n, y, xx, co, got, Y[qq]);
                        ++bad;
                    }
                }
            }
            printf("%s: %d outputs checked bit for bit against the oracle chains, %d mismatches\n", name, checked, bad);
        }
    }
    // executed flops: 16 positions x C x ncols x 2 per tile; direct-equivalent: 9 taps x 4 pixels
    const double tiles = (double)N * ((H + 1) / 2) * ((W + 1) / 2);
    const double fl_exec = tiles * 16.0 * C * ncols * 2.0, fl_dir = tiles * 36.0 * C * ncols * 2.0;
    printf("%s: N=%d %dx%d C=%d cols=%d grid=%d  | streams: %.1f us = %.1f TFLOP/s executed = %.1f direct-equivalent | no streams: %.1f us = %.1f executed\n",
           name, N, H, W, C, ncols, grid, best[1] * 1e3, fl_exec / best[1] / 1e9, fl_dir / best[1] / 1e9, best[0] * 1e3, fl_exec / best[0] / 1e9);
    fflush(stdout);
    hipFree(dx); hipFree(dw); hipFree(db); hipFree(dout);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) {   // ablations on the L3 shape
        run<6>("L3, no LDS reads in the loop", 4, 64, 64, 384, 768, 3);
        run<7>("L3, no LDS reads, no barrier", 4, 64, 64, 384, 768, 3);
        run<2>("L3, weight reads only", 4, 64, 64, 384, 768, 3);
        run<3>("L3, weight reads only, no barrier", 4, 64, 64, 384, 768, 3);
        run<1>("L3 full, no barrier", 4, 64, 64, 384, 768, 3);
        run<0>("L3 full", 4, 64, 64, 384, 768, 3);
        // the loop alone: twice the channels, everything else equal -> (t2 - t1) / (96 stages x 3 rounds) per stage
        run<0>("L3 full, C = 768", 4, 64, 64, 768, 768, 3);
        run<7>("L3 no LDS reads no barrier, C = 768", 4, 64, 64, 768, 768, 3);
        return 0;
    }
    run<0>("small", 1, 32, 32, 32, 64, 2);
    run<0>("edge", 2, 24, 40, 48, 128, 2);
    run<0>("L3 gates (k_conv16: 604 us)", 4, 64, 64, 384, 768, 5);
    run<0>("A2 (k_conv16: 306 us)", 4, 128, 128, 192, 192, 5);
    run<0>("A1 (k_wino: 198 us)", 4, 256, 256, 96, 96, 5);   // three 32-column blocks, nothing padded
    run<0>("L2 gates same-res part", 4, 128, 128, 192, 384, 5);
    run<0>("L1 gates same-res part", 4, 256, 256, 96, 192, 5);
    return 0;
}
