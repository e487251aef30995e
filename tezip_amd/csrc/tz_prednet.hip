// PredNet predictor on MI355X: one LDS-tiled implicit-GEMM 3x3 convolution kernel on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32), with the ConvLSTM / error-unit / max-pool
// epilogues fused, and the host-side model driver.
//
// What is computed (reference: /root/reference/src/prednet.py:235-308, used through
// Model.predict on a 2-step sequence from zero state, compress.py:224-229):
//   * Everything that does not depend on the input frame is evaluated ONCE per
//     (model, Hp, Wp) in tz_model_prepare: the t=0 top-down pass (r0_l, c0_l), Ahat_l at t=0,
//     and G0_l = bias + conv(r0_l part of the t=1 gate convolution).
//   * Per predicted frame only the live work runs: t=0 bottom-up (error units + A convs),
//     t=1 top-down (gate convs over [e_l, up(r_{l+1})] starting from G0_l) and Ahat_0.
//     The t=1 bottom-up pass and the zero second input are dead.
//
// Arithmetic contract TZ-PA1 (bit-exact with oracle/tz_oracle.c): every convolution output is
// ONE float32 fmaf chain: acc = bias; for source in concat order, for each block of 16 input
// channels, for each tap, for ci in the block: acc = fmaf(x, w, acc).  Taps of a same-resolution
// source are the 9 (ky, kx); an upsampled source up(r_{l+1}) is read at HALF resolution with 4
// collapsed taps (dy, dx) whose weights are the float32 sums of the 3x3 taps that land on the
// same half-resolution pixel for the output pixel's parity (2.25x fewer MACs for that source).
// The MFMA k-loop below walks exactly that order; out-of-image taps and channel padding
// contribute fmaf(0, w, acc) = acc.  No split-K, no atomics: results do not depend on batch
// size, grid shape or device.
//
// Tiling: workgroup = 8 waves = 16x16 output pixels (256 GEMM rows) x NT*16 output columns;
// each wave owns 32 rows x NT*16 columns = 2 x NT MFMA tiles (acc in registers).  For every
// block of 16 input channels the 18x18 halo patch is staged in LDS ONCE and all 9 taps read
// their shifted A fragments from it (the first version re-staged A per tap and was bound by
// L2/Infinity-Cache traffic: 53 % L2 hit rate, profiles/r01); the per-tap 16 x (NT*16) weight
// chunk is double-buffered in LDS, one barrier per tap.  The next patch is prefetched into
// registers during the 9 taps.  LDS row strides (18 / NT*16[+16] floats) keep fragment reads
// conflict free.  33-44 KB LDS, <=100 VGPRs, 512 threads => 3 workgroups (24 waves) per CU.
#include <algorithm>
#include <type_traits>

#include "tz_internal.h"
#include "tz_math.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_RAW = 0, EPI_RELU = 1, EPI_LSTM = 2, EPI_LSTM_PACKED = 3, EPI_POOL_ERR = 4 };

struct ConvSrc {
    const float* p;
    long long nstride;  // elements between batch items (0 = broadcast constant)
    int C;              // channels
    int pstride;        // floats between pixels (>= C; level-0 error maps are stored 8 wide)
    int up;             // 1: stored at half resolution, nearest x2 on read (prednet.py:264)
    int cpt;            // 16-channel blocks of this source = ceil(C/16)
};

struct ConvArgs {
    ConvSrc src[2];
    int nsrc;
    int H, W, tiles_x, tiles_y, ncb;
    const float* Wp;    // [weight slots * 16][ncols], slot order = the K-loop order (see pack_conv)
    const float* Wimg;  // the same weights in LDS image order for k_conv16 (see pack_conv), or null
    const float* Wblk;  // block-step image for k_conv16b (see pack_block_image), or null
    const float* zero;  // >= 16 bytes of zeros: LDS-DMA source of out-of-image patch pixels
    int ncols;
    const float* bias;  // [ncols]
    const float* init;  // [H*W][ncols] accumulator start (G0), or null -> bias
    int Cout;
    const float* aux;   // LSTM: previous cell state [H*W][R] or null; POOL_ERR: Ahat(t0) of level l+1
    float* out0;
    long long out0_nstride;
    float* out1;        // LSTM: cell state out (or null)
    long long out1_nstride;
    const int* out_idx; // optional: frame slot of batch item n in out0
    int clip1;          // EPI_RELU: min(.,1)  (prednet.py:270)
    int R;              // EPI_LSTM_PACKED: channels per gate
};

static constexpr int SA = 18;    // LDS row stride of one patch pixel (16 channels + 2 pad floats)
static constexpr int PW = 18;    // same-resolution halo patch: PW x PW pixels around the 16x16 tile
static constexpr int PPIX = PW * PW;
static constexpr int LW = 10;    // half-resolution patch of an upsampled source: LW x LW pixels
static constexpr int LPIX = LW * LW;
static constexpr int NTHR = 512;                         // 8 waves, each owns two 16-row MFMA tiles
static constexpr int MT = 2;
static constexpr int A_ITEMS = PPIX * 4;                 // float4 items of one patch channel block
static constexpr int A_PER_THREAD = (A_ITEMS + NTHR - 1) / NTHR;  // 3

enum { MAP_LINEAR = 0, MAP_POOL = 1, MAP_PARITY = 2 };

// GEMM row m (0..255) of the workgroup -> pixel (py, px) of its 16x16 output tile.
//  LINEAR: wave w = rows 2w, 2w+1 of the tile.
//  POOL:   the 4 accumulator registers of a lane form one 2x2 pooling window.
//  PARITY: every 16-row MFMA tile holds pixels of ONE parity class (py&1, px&1), so that the
//          parity-specific collapsed weights of an upsampled source can be its B operand.
// Tile shapes and patch strides were chosen by exhaustive search so that every ds_read_b32 of
// an A fragment (16 rows x 2 k per 32-lane group) is bank-conflict free at stride 18 for LINEAR
// and POOL (PARITY keeps a 2-way conflict, see below; LDS is not the critical path).
template <int MAP>
__device__ __forceinline__ void row_to_patch(int m, int& py, int& px) {
    int w = m >> 5, mt = (m >> 4) & 1, r16 = m & 15;
    if (MAP == MAP_POOL) {
        // M-tile = 8 rows x 2 columns = four stacked 2x2 windows (conflict-free at stride 18)
        int T = 2 * w + mt;
        py = 8 * (T >> 3) + 2 * (r16 >> 2) + ((r16 & 3) >> 1);
        px = 2 * (T & 7) + (r16 & 1);
    } else if (MAP == MAP_PARITY) {
        // M-tile = two rows x 8 columns of the 8x8 grid of one parity class (2-way conflicts on its
        // A reads; the conflict-free alternative -- rows (sub, sub+4) at stride 17 -- needs 4-byte
        // patch stores and measured 2.5 % slower)
        int T = 2 * w + mt, pc = T >> 2, sub = T & 3;
        py = 2 * (2 * sub + (r16 >> 3)) + (pc >> 1);
        px = 2 * (r16 & 7) + (pc & 1);
    } else {
        py = 2 * w + mt;
        px = r16;
    }
}

// XCD-aware block order: blocks b and b+8 share an XCD (round-robin dispatch), so give every
// XCD a contiguous range of logical ids: the column blocks of one pixel tile and neighbouring
// tiles then share one L2.  Bijective for any grid size; only speed depends on it.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Fused epilogues on the accumulator tiles of one wave (shared by both convolution kernels):
// acc[mt][nt], element r <-> GEMM row (lane>>4)*4 + r of M-tile mt, column lane&15 of N-tile nt.
template <int NT, int EPI, int MAP>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NT], int n, int cb, int ty0, int tx0,
                                              int wv, int lane, float* scratch) {
    constexpr int NTC = NT * 16;
    const int col0 = cb * NTC + (lane & 15);
    auto out_pix = [&](int mt, int r, int& y, int& x) {
        int py, px;
        row_to_patch<MAP>(wv * 32 + mt * 16 + (lane >> 4) * 4 + r, py, px);
        y = ty0 + py;
        x = tx0 + px;
    };
    const int j = lane & 15;
    if (EPI == EPI_RAW) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int y, x;
                out_pix(mt, r, y, x);
                if (y >= a.H || x >= a.W) continue;
                long long pix = (long long)y * a.W + x;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) a.out0[pix * a.ncols + col0 + nt * 16] = acc[mt][nt][r];
            }
    } else if (EPI == EPI_RELU) {
        float* o = a.out0 + (long long)(a.out_idx ? a.out_idx[n] : n) * a.out0_nstride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int y, x;
                out_pix(mt, r, y, x);
                if (y >= a.H || x >= a.W) continue;
                long long pix = (long long)y * a.W + x;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    int ch = col0 + nt * 16;
                    if (ch < a.Cout) {
                        float v = tz_relu(acc[mt][nt][r]);
                        if (a.clip1 && v > 1.0f) v = 1.0f;
                        o[pix * a.Cout + ch] = v;
                    }
                }
            }
    } else if (EPI == EPI_LSTM) {
        // columns of this block: [i | f | g | o] x 16 channels of channel group cb
        // prednet.py:255-259: c = f*c_prev + i*g ; r = o*tanh(c)
        // All c_prev loads are issued first, from clamped (always valid) addresses: a load inside
        // the per-pixel bounds branch costs one memory round trip per pixel (8 per wave).
        const int ch = cb * 16 + j, R = a.Cout;
        float* o0 = a.out0 + (long long)n * a.out0_nstride;
        float* o1 = a.out1 ? a.out1 + (long long)n * a.out1_nstride : nullptr;
        long long pix[MT][4];
        bool ok[MT][4];
        float cp[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int y, x;
                out_pix(mt, r, y, x);
                ok[mt][r] = y < a.H && x < a.W;
                pix[mt][r] = ok[mt][r] ? (long long)y * a.W + x : 0;
                cp[mt][r] = a.aux ? a.aux[pix[mt][r] * R + ch] : 0.0f;
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gi = tz_hard_sigmoid(acc[mt][0 % NT][r]);
                float gf = tz_hard_sigmoid(acc[mt][1 % NT][r]);
                float gg = tz_tanh(acc[mt][2 % NT][r]);
                float go = tz_hard_sigmoid(acc[mt][3 % NT][r]);
                float t1 = gf * cp[mt][r];
                float t2 = gi * gg;
                float c = t1 + t2;
                float rr = go * tz_tanh(c);
                if (ok[mt][r]) {
                    o0[pix[mt][r] * R + ch] = rr;
                    if (o1) o1[pix[mt][r] * R + ch] = c;
                }
            }
    } else if (EPI == EPI_LSTM_PACKED) {
        // one 16-column tile holds [i(R) f(R) g(R) o(R)], R <= 4.  Only R of 16 lanes of the
        // accumulator layout own a channel, so the LSTM update (two tanh per item) is re-distributed:
        // the wave's 32 x 16 tile goes through its private LDS scratch and every lane takes
        // (pixel, channel) items lane, lane + 64 of the 32 * R -- 4x less VALU than 8 masked rows per
        // lane (the level-0 gate launch was VALU-bound in this epilogue).
        const int R = a.R;
        float* o0 = a.out0 + (long long)n * a.out0_nstride;
        float* o1 = a.out1 ? a.out1 + (long long)n * a.out1_nstride : nullptr;
        float* t = scratch + wv * (32 * 17);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[(mt * 16 + (lane >> 4) * 4 + r) * 17 + j] = acc[mt][0][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int row[2], c[2];
        long long pix[2];
        bool ok[2];
        float cp[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = lane + 64 * it;
            row[it] = item / R;
            c[it] = item - row[it] * R;
            int py, px;
            row_to_patch<MAP>(wv * 32 + (row[it] & 31), py, px);
            const int y = ty0 + py, x = tx0 + px;
            ok[it] = item < 32 * R && y < a.H && x < a.W;
            pix[it] = ok[it] ? ((long long)y * a.W + x) * R + c[it] : 0;
            cp[it] = a.aux ? a.aux[pix[it]] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const float* tr = t + (row[it] & 31) * 17 + c[it];
            float gi = tz_hard_sigmoid(tr[0]), gf = tz_hard_sigmoid(tr[R]), gg = tz_tanh(tr[2 * R]), go = tz_hard_sigmoid(tr[3 * R]);
            float t1 = gf * cp[it];
            float t2 = gi * gg;
            float cc = t1 + t2;
            float rr = go * tz_tanh(cc);
            if (ok[it]) {
                o0[pix[it]] = rr;
                if (o1) o1[pix[it]] = cc;
            }
        }
    } else if (EPI == EPI_POOL_ERR) {
        // prednet.py:289-291 then 274-277 of the next level: A = maxpool2x2(relu(conv));
        // e = [relu(Ahat0 - A), relu(A - Ahat0)] written at the pooled resolution.
        const int H2 = a.H >> 1, W2 = a.W >> 1, C = a.Cout;
        float* o = a.out0 + (long long)n * a.out0_nstride;
        long long pp[MT];
        bool ok[MT][NT];
        float h[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int y, x;
            out_pix(mt, 0, y, x);
            const int yp = y >> 1, xp = x >> 1;
            const bool okp = yp < H2 && xp < W2;
            pp[mt] = okp ? (long long)yp * W2 + xp : 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = col0 + nt * 16;
                ok[mt][nt] = okp && ch < C;
                h[mt][nt] = a.aux[ok[mt][nt] ? pp[mt] * C + ch : 0];
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = col0 + nt * 16;
                float m = tz_relu(acc[mt][nt][0]);
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    float t = tz_relu(acc[mt][nt][r]);
                    if (t > m) m = t;
                }
                float d1 = h[mt][nt] - m, d2 = m - h[mt][nt];
                if (ok[mt][nt]) {
                    o[pp[mt] * 2 * C + ch] = tz_relu(d1);
                    o[pp[mt] * 2 * C + C + ch] = tz_relu(d2);
                }
            }
    }
}

// FULLK: every source has a multiple of 16 channels, so every same-resolution step runs all four
// k-steps (lets the compiler schedule the 32 MFMAs of a step as one straight-line block).
template <int NT, int EPI, bool UPS, bool FULLK>
__global__ __launch_bounds__(NTHR, NT == 1 ? 8 : 6) void k_conv3x3(const ConvArgs a) {
    // parity tiles only where an upsampled source needs them (the top level has none)
    constexpr int MAP = EPI == EPI_POOL_ERR ? MAP_POOL : (UPS ? MAP_PARITY : MAP_LINEAR);
    constexpr int SAH = SA;                           // floats per pixel of the same-resolution patch
    constexpr int SAL = SA;                           // ... of the half-resolution patch
    constexpr int NTC = NT * 16;
    constexpr int SB = NTC + ((NTC % 32) == 0 ? 16 : 0);
    constexpr int QPR = NTC / 4;                     // float4 items per weight row
    constexpr int BVEC = 16 * QPR;                   // items of a same-resolution chunk (16 k-rows)
    constexpr int UPH = NT == 4 ? 8 : 16;            // k-rows of an upsampled-source step (per class)
    constexpr int UPN = 16 / UPH;                    // steps per collapsed tap
    constexpr int BUF = (UPS ? 4 * UPH : 16) * SB;   // floats per weight buffer (up: 4 classes x UPH rows)
    static_assert((!UPS || 4 * UPH * QPR <= NTHR) && BVEC <= NTHR, "one weight item per thread");
    __shared__ float sA[PPIX * SA];
    __shared__ float sB[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = bid % ntiles, n = bid / ntiles;
    const int ty0 = (tile / a.tiles_x) * 16, tx0 = (tile % a.tiles_x) * 16;

    // ---- A staging roles.  Same-resolution source: 18x18x16 halo patch = 1296 float4 items,
    // item i -> patch pixel i>>2, channel quad i&3.  Upsampled source: 10x10x16 = 400 items of
    // the half-resolution map (item i = tid).
    const int aq = tid & 3;

    float4 ra[A_PER_THREAD];
    float4 rb = make_float4(0.f, 0.f, 0.f, 0.f);

    const int nb0 = a.nsrc > 0 ? a.src[0].cpt : 0;
    const int nblk = nb0 + (a.nsrc > 1 ? a.src[1].cpt : 0);

    auto load_quad = [&](const ConvSrc& s, const float* ptr, int c0) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((s.C & 3) == 0 && (s.pstride & 3) == 0) {
            v = *(const float4*)ptr;
        } else {
            v.x = ptr[0];
            if (c0 + 1 < s.C) v.y = ptr[1];
            if (c0 + 2 < s.C) v.z = ptr[2];
            if (c0 + 3 < s.C) v.w = ptr[3];
        }
        return v;
    };
    auto load_patch = [&](int blk) {
        const ConvSrc& s = blk >= nb0 ? a.src[1] : a.src[0];
        const int c0 = (blk >= nb0 ? blk - nb0 : blk) * 16 + 4 * aq;
        const float* base = s.p + (long long)n * s.nstride;
        // (pixel coordinates are recomputed per block rather than kept in registers)
        if (UPS && s.up) {
            const int pp = tid >> 2;
            const int ly = (ty0 >> 1) - 1 + pp / LW, lx = (tx0 >> 1) - 1 + pp % LW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pp < LPIX && ly >= 0 && ly < (a.H >> 1) && lx >= 0 && lx < (a.W >> 1) && c0 < s.C)
                v = load_quad(s, base + ((long long)ly * (a.W >> 1) + lx) * s.pstride + c0, c0);
            ra[0] = v;
        } else {
#pragma unroll
            for (int j = 0; j < A_PER_THREAD; ++j) {
                const int i = tid + NTHR * j, pp = i >> 2;
                const int yy = ty0 - 1 + pp / PW, xx = tx0 - 1 + pp % PW;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < A_ITEMS && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && c0 < s.C)
                    v = load_quad(s, base + ((long long)yy * a.W + xx) * s.pstride + c0, c0);
                ra[j] = v;
            }
        }
    };
    auto store_patch = [&](bool up) {
#pragma unroll
        for (int j = 0; j < A_PER_THREAD; ++j) {
            int i = tid + NTHR * j;
            if (up ? (j == 0 && i < LPIX * 4) : (i < A_ITEMS)) {
                if (up || (SAH & 1) == 0) {
                    float* d = sA + (i >> 2) * (up ? SAL : SAH) + 4 * aq;
                    *(float2*)d = make_float2(ra[j].x, ra[j].y);
                    *(float2*)(d + 2) = make_float2(ra[j].z, ra[j].w);
                } else {  // odd stride: only 4-byte alignment
                    float* d = sA + (i >> 2) * SAH + 4 * aq;
                    d[0] = ra[j].x;
                    d[1] = ra[j].y;
                    d[2] = ra[j].z;
                    d[3] = ra[j].w;
                }
            }
        }
    };
    // Weight staging, one float4 per thread and step.  Same-resolution step: the 16 k-rows of
    // one (block, tap).  Upsampled step: UPH k-rows (part `hf` of the block) of one collapsed tap
    // for each of the 4 parity classes; the packed file keeps 16 rows per (tap, class) slot.
    auto load_b = [&](int slot, bool up, int hf) {
        if (UPS && up) {
            int cls = tid / (UPH * QPR), r = tid - cls * (UPH * QPR);
            if (tid < 4 * UPH * QPR)
                rb = *(const float4*)(a.Wp + ((long long)(slot + cls) * 16 + UPH * hf + r / QPR) * a.ncols + cb * NTC + 4 * (r % QPR));
        } else if (tid < BVEC) {
            rb = *(const float4*)(a.Wp + ((long long)slot * 16 + tid / QPR) * a.ncols + cb * NTC + 4 * (tid % QPR));
        }
    };
    auto store_b = [&](int buf, bool up) {
        if (UPS && up) {
            if (tid < 4 * UPH * QPR) *(float4*)(sB + buf * BUF + (tid / QPR) * SB + 4 * (tid % QPR)) = rb;  // row = cls*UPH + r
        } else if (tid < BVEC) {
            *(float4*)(sB + buf * BUF + (tid / QPR) * SB + 4 * (tid % QPR)) = rb;
        }
    };

    // ---- accumulators: acc[mt][nt], element r <-> GEMM row (lane>>4)*4 + r, column lane&15
    f32x4 acc[MT][NT];
    const int col0 = cb * NTC + (lane & 15);
    auto out_pix = [&](int mt, int r, int& y, int& x) {
        int py, px;
        row_to_patch<MAP>(wv * 32 + mt * 16 + (lane >> 4) * 4 + r, py, px);
        y = ty0 + py;
        x = tx0 + px;
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (a.init) {
                // (clamped address instead of a bounds branch: rows outside the image are never stored)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int y, x;
                    out_pix(mt, r, y, x);
                    const long long pix = (y < a.H && x < a.W) ? (long long)y * a.W + x : 0;
                    acc[mt][nt][r] = a.init[pix * a.ncols + col0 + nt * 16];
                }
            } else {
                float b = a.bias[col0 + nt * 16];
                acc[mt][nt] = (f32x4){b, b, b, b};
            }
        }

    // ---- K loop: blocks of 16 input channels (patch staged once per block).  A same-resolution
    // source runs 9 steps per block (one tap each, 2 or 4 k-steps of 4 channels); an upsampled
    // source runs 4*UPN steps (4 collapsed taps x UPN parts of UPH channels; the weights of the 3x3
    // taps that hit the same half-resolution pixel were summed at pack time, per parity class).
    // Every step: prefetch next weights -> MFMAs from LDS -> store next weights -> one barrier.
    if (nblk > 0) {
        int arow_hi[MT], arow_lo[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int py, px;
            row_to_patch<MAP>(wv * 32 + mt * 16 + (lane & 15), py, px);
            arow_hi[mt] = (py * PW + px) * SAH + (lane >> 4);
            arow_lo[mt] = (((py >> 1) + (py & 1)) * LW + (px >> 1) + (px & 1)) * SAL + (lane >> 4);
        }
        const int wcls = MAP == MAP_PARITY ? (wv >> 1) : 0;  // parity class of this wave's rows
        const int boff = (lane >> 4) * SB + (lane & 15);
        bool up = UPS && a.src[nb0 > 0 ? 0 : 1].up != 0;
        load_patch(0);
        load_b(0, up, 0);
        store_patch(up);
        store_b(0, up);
        __syncthreads();
        int cur = 0, slot0 = 0;
        for (int blk = 0; blk < nblk; ++blk) {
            const ConvSrc& s = blk >= nb0 ? a.src[1] : a.src[0];
            up = UPS && s.up != 0;
            const int nsteps = up ? 4 * UPN : 9;
            const int cw = s.C - (blk >= nb0 ? blk - nb0 : blk) * 16;
            const bool more_blk = blk + 1 < nblk;
            const bool up_next = UPS && more_blk && (blk + 1 >= nb0 ? a.src[1].up : a.src[0].up) != 0;
#pragma unroll 1
            for (int st = 0; st < nsteps; ++st) {
                const bool last = st == nsteps - 1;
                if (!last) load_b(up ? slot0 + 4 * ((st + 1) / UPN) : slot0 + st + 1, up, (st + 1) % UPN);
                else if (more_blk) load_b(slot0 + (up ? 16 : 9), up_next, 0);
                // The next block's patch gather is issued AFTER this step's weight load: vmcnt
                // retires in order, so the end-of-step wait for the (older) weight load leaves the
                // patch loads in flight for one more step instead of forcing them after one.
                if (st == 0 && more_blk) load_patch(blk + 1);
                // one compute body for both kinds of step: the first pair of k-steps always runs,
                // the second pair only when the step holds more than 8 channels
                const int tap = up ? st / UPN : st;
                const int toff = up ? ((tap >> 1) * LW + (tap & 1)) * SAL : ((tap / 3) * PW + (tap % 3)) * SAH;
                const bool second = up ? (UPH == 16) : (FULLK || cw > 8);
                const float* pa = sA + toff + (up ? UPH * (st % UPN) : 0);
                const float* pb = sB + cur * BUF + boff + (up ? wcls * UPH * SB : 0);
                int ar[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ar[mt] = up ? arow_lo[mt] : arow_hi[mt];
                // (kept as ONE loop nest: duplicating the MFMA body per k-step count made hipcc
                // spill accumulators inside the loop under the 80-VGPR budget, 4x slower)
#pragma unroll
                for (int pair = 0; pair < 2; ++pair) {
                    if (pair == 0 || second) {
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            const int kk = 2 * pair + k2;
                            float fa[MT], fb[NT];
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) fa[mt] = pa[ar[mt] + 4 * kk];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) fb[nt] = pb[4 * kk * SB + nt * 16];
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
                        }
                    }
                }
                if (!last) store_b(cur ^ 1, up);
                else if (more_blk) store_b(cur ^ 1, up_next);
                __syncthreads();
                cur ^= 1;
            }
            slot0 += up ? 16 : 9;
            if (more_blk) {
                store_patch(up_next);
                __syncthreads();
            }
        }
    }
    if (EPI == EPI_LSTM_PACKED) __syncthreads();  // the patch buffer becomes the epilogue's scratch
    static_assert(8 * 32 * 17 <= PPIX * SA, "epilogue scratch fits the patch buffer");
    conv_epilogue<NT, EPI, MAP>(a, acc, n, cb, ty0, tx0, wv, lane, sA);
}

// ------------------------------------------------------------------------------------------
// k_conv16: the same implicit GEMM (same tiles, same fmaf-chain order, same epilogues) for
// convolutions whose sources all have a multiple of 16 channels -- every hot launch of levels >= 1.
// All staging is LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, so the
// K loop holds only accumulators, fragments and addresses (k_conv3x3 keeps a register-staged
// prefetch of the next patch and weight chunk alive across its steps: 22 spilled VGPRs and a
// vmcnt(0) in front of every use at the 80-VGPR budget; MFMA pipe 68-80 % busy at 2.39 GHz).
//
// An LDS-DMA writes wave-uniform base + lane*16 B, so both LDS images are lane-linear and the
// layout work moves to the SOURCE address:
//  * weights are packed on the host in image order [slot][column block][k-step][lane][4]: lane
//    (g = lane>>4, j = lane&15) of k-step kk holds W[4kk+g][16nt+j], nt = 0..3 -- one
//    ds_read_b128 per k-step gives the B fragments of all four column tiles, conflict free;
//    a (slot, column block) chunk is 4 KB = four 1 KB wave-instructions;
//  * the halo patch is QUAD-PLANAR: the 16-byte item (slot, channel quad q) sits at item index
//    q * NP + slot (NP = 336 slots per plane for the 18x18 patch, 112 for the 10x10 one; 4 planes =
//    exactly 21 / 7 wave-instructions).  An A fragment address is then lane base + tap offset +
//    k-step offset with the last two uniform (a scalar add and an instruction immediate): the K
//    loop carries almost no address arithmetic, which measured as the largest single loss of the
//    DMA'd loop (scripts/microbench/conv_skeleton.hip: 133 -> 140 TFLOP/s; 8-way LDS conflicts on
//    the same reads cost nothing measurable).  A ds_read_b32 of 16 rows x 2 k is 2-way conflicted
//    at best in any 16-byte-granular image (lanes 0-31 only touch elements 0,1 of a quad); the
//    plane layout reaches that for all three row maps, with the columns of a parity-tile patch
//    stored evens first (x -> (x>>1) + 9*(x&1)) so that one parity class is contiguous.
//    Out-of-image pixels read a zero page.
// Protocol per step: issue the DMA of the next step's weights into the other weight buffer, MFMAs
// of this step, s_waitcnt vmcnt(0), s_barrier.  The NEXT block's patch is issued into the other
// patch buffer ahead of a block's first step and retires with that step's wait, one whole step of
// MFMAs later: waves do not stall for a patch.
// LDS = 50 pieces of 1 KB (3 workgroups per CU).  Same-resolution phase: patches P0 = [0,21),
// P1 = [21,42), weight buffers 42 + 4*buf.  The upsampled source's steps (one collapsed tap each:
// 4 k-steps x 4 parity classes = 16 pieces of weights, 32 MFMAs per wave and barrier) use
// 7-piece patches [0,7), [7,14) and weight buffers 14 + 16*buf; the switch between the two
// layouts happens once per workgroup behind a barrier, with an un-overlapped first load.
static constexpr int P16_PIECES = 21;                   // 1 KB pieces of an 18x18 patch (324 px -> 20.25)
static constexpr int U16_PIECES = 7;                    // ... of a 10x10 half-resolution patch
static constexpr int NP16 = P16_PIECES * 16, NPU16 = U16_PIECES * 16;  // slots per quad plane (336 / 112)
static constexpr int C16_LDS_PIECES = 2 * P16_PIECES + 8;   // >= 2 * U16_PIECES + 32

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// waits for all of this wave's vector-memory operations (LDS-DMA included)
__device__ __forceinline__ void wait_vm(int = 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NT, int EPI, bool UPS>
__global__ __launch_bounds__(NTHR, 6) void k_conv16(const ConvArgs a) {
    constexpr int MAP = EPI == EPI_POOL_ERR ? MAP_POOL : (UPS ? MAP_PARITY : MAP_LINEAR);
    __shared__ __attribute__((aligned(16))) float smem[C16_LDS_PIECES * 256];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id, scalar
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = bid % ntiles, n = bid / ntiles;
    const int ty0 = (tile / a.tiles_x) * 16, tx0 = (tile % a.tiles_x) * 16;
    const int nb0 = a.src[0].cpt;
    const int nblk = nb0 + (a.nsrc > 1 ? a.src[1].cpt : 0);
    const int g = lane >> 4;
    // src[] is ordered same-resolution first (or holds only the upsampled source)
    const bool up0 = UPS && a.src[0].up != 0;
    const int nbe = up0 ? 0 : (UPS && a.nsrc > 1 && a.src[1].up ? nb0 : nblk);  // same-resolution blocks

    // ---- LDS-DMA issue.  Item i = piece * 64 + lane of a patch = quad plane i / NP, slot i % NP.
    auto issue_patch = [&](int blk, int piece0) {
        const bool s1 = blk >= nb0;
        const ConvSrc& s = s1 ? a.src[1] : a.src[0];
        const int c0 = (s1 ? blk - nb0 : blk) * 16;
        const float* base = s.p + (long long)n * s.nstride;
        float* dst = smem + piece0 * 256;
        // the patch geometry is recomputed per call (once per block): kept live across the K loop
        // it costs ~12 VGPRs and spills
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (UPS && blk >= nbe) {
            if (wv >= U16_PIECES) return;
            const int i = wv * 64 + ln, q = i / NPU16, slot = i - q * NPU16;
            const int Y = slot / LW, X = slot - Y * LW;
            const int ly = (ty0 >> 1) - 1 + Y, lx = (tx0 >> 1) - 1 + X;
            const bool ok = slot < LPIX && ly >= 0 && ly < (a.H >> 1) && lx >= 0 && lx < (a.W >> 1);
            glds16(ok ? base + ((long long)ly * (a.W >> 1) + lx) * s.pstride + c0 + 4 * q : a.zero, dst + wv * 256);
            return;
        }
#pragma unroll
        for (int j = 0; j < (P16_PIECES + 7) / 8; ++j) {
            const int piece = wv + 8 * j;
            if (piece < P16_PIECES) {
                const int i = piece * 64 + ln, q = i / NP16, slot = i - q * NP16;
                const int y = slot / PW, xs = slot - y * PW;
                const int x = MAP == MAP_PARITY ? (xs < PW / 2 ? 2 * xs : 2 * (xs - PW / 2) + 1) : xs;
                const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                const bool ok = slot < PPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                glds16(ok ? base + ((long long)yy * a.W + xx) * s.pstride + c0 + 4 * q : a.zero, dst + piece * 256);
            }
        }
    };
    // weights of one step into the pieces starting at piece0.  Same-resolution step: the 4 k-steps
    // of (block, tap), one piece each from the waves of half `half` of the workgroup.  Upsampled
    // step: the 4 k-steps of one collapsed tap for each of the 4 parity classes, piece index
    // class * 4 + k-step; wave w fetches k-steps 2(w&1), 2(w&1)+1 of class w>>1.
    auto issue_w = [&](int slot, bool up, int piece0, int half) {
        if (UPS && up) {
            const float* src = a.Wimg + (((long long)(slot + (wv >> 1)) * a.ncb + cb) * 4 + 2 * (wv & 1)) * 256 + lane * 4;
            float* dst = smem + (piece0 + 2 * wv) * 256;
            glds16(src, dst);
            glds16(src + 256, dst + 256);
        } else if ((wv >> 2) == half) {
            glds16(a.Wimg + (((long long)slot * a.ncb + cb) * 4 + (wv & 3)) * 256 + lane * 4,
                   smem + (piece0 + (wv & 3)) * 256);
        }
    };

    // ---- accumulators (as in k_conv3x3)
    f32x4 acc[MT][NT];
    {
        const int col0 = cb * (NT * 16) + (lane & 15);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (a.init) {
                    // (clamped address instead of a bounds branch: rows outside the image are never stored)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int py, px;
                        row_to_patch<MAP>(wv * 32 + mt * 16 + g * 4 + r, py, px);
                        const int y = ty0 + py, x = tx0 + px;
                        const long long pix = (y < a.H && x < a.W) ? (long long)y * a.W + x : 0;
                        acc[mt][nt][r] = a.init[pix * a.ncols + col0 + nt * 16];
                    }
                } else {
                    const float b = a.bias[col0 + nt * 16];
                    acc[mt][nt] = (f32x4){b, b, b, b};
                }
            }
    }

    // float index of this lane's A row (tile pixel of GEMM row lane&15, element g of the quad) in
    // plane 0 of each patch image, for tap (0,0)
    int abase[MT], abase_lo[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int py, px;
        row_to_patch<MAP>(wv * 32 + mt * 16 + (lane & 15), py, px);
        abase[mt] = 4 * (py * PW + (MAP == MAP_PARITY ? (px >> 1) + (PW / 2) * (px & 1) : px)) + g;
        abase_lo[mt] = 4 * (((py >> 1) + (py & 1)) * LW + (px >> 1) + (px & 1)) + g;
    }
    const int wcls = MAP == MAP_PARITY ? (wv >> 1) : 0;  // parity class (py&1, px&1) of this wave's rows

    // The K loop is two loop nests in sequence -- the blocks of the same-resolution sources, then
    // the blocks of the upsampled source -- each with ONE MFMA body: with both kinds of step in one
    // loop body hipcc moves the accumulators between register sets per branch and spills them.
    int slot0 = 0;
    auto run_phase = [&](auto upc, int b0, int b1) {
        constexpr bool UP = decltype(upc)::value;
        constexpr int nsteps = UP ? 4 : 9, bslots = UP ? 16 : 9, WP = UP ? 16 : 4;
        constexpr int pA0 = 0, pA1 = UP ? U16_PIECES : P16_PIECES;   // patch buffers
        constexpr int wA = 2 * pA1;                                  // weight buffers
        if (b0 >= b1) return;
        // first patch and first weights of the phase (nothing of the previous phase is live)
        issue_patch(b0, pA0);
        issue_w(slot0, UP, wA, 0);
        wait_vm(0);
        wg_barrier();
        int pi = 0, cur = 0;
#pragma unroll 1
        for (int blk = b0; blk < b1; ++blk) {
            const bool more_blk = blk + 1 < b1;
            // the next block's patch goes into the other patch buffer while this block computes; it
            // is retired together with the first step's weight DMA (a whole step later)
            if (more_blk) issue_patch(blk + 1, pi ? pA0 : pA1);
#pragma unroll 1
            for (int st = 0; st < nsteps; ++st) {
                if (st + 1 < nsteps) issue_w(UP ? slot0 + 4 * (st + 1) : slot0 + st + 1, UP, wA + WP * (cur ^ 1), cur ^ 1);
                else if (more_blk) issue_w(slot0 + bslots, UP, wA + WP * (cur ^ 1), cur ^ 1);
                const float* pa = smem + (pi ? pA1 : pA0) * 256;
                const float* wb = smem + (wA + WP * cur + (UP ? 4 * wcls : 0)) * 256 + lane * 4;
                // tap offset in slots.  Same resolution: dy rows of 18; a parity-tile patch stores its
                // columns evens first, so one step in x is +9 / -8 from an even / odd column and two
                // steps are +1.  Upsampled: the 2x2 collapsed taps of the 10-wide half-resolution patch.
                int toff;
                if (UP) {
                    toff = 4 * ((st >> 1) * LW + (st & 1));
                } else {
                    const int dy = st / 3, dx = st - 3 * dy;
                    const int xo = MAP == MAP_PARITY ? (dx == 1 ? ((wcls & 1) ? 1 - PW / 2 : PW / 2) : (dx >> 1)) : dx;
                    toff = 4 * (dy * PW + xo);
                }
                constexpr int KOFF = 4 * (UP ? NPU16 : NP16);  // floats between the quad planes
                float fa[MT][4];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[(UP ? abase_lo[mt] : abase[mt]) + toff + kk * KOFF];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 fb = *(const f32x4*)(wb + kk * 256);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
                }
                wait_vm(0);
                wg_barrier();
                cur ^= 1;
            }
            slot0 += bslots;
            pi ^= 1;
        }
    };
    // waves in the K loop outrank the waves of other workgroups that are in their prologue or
    // epilogue (VALU-dense, and older): the matrix pipe is issued first
    __builtin_amdgcn_s_setprio(1);
    run_phase(std::false_type{}, 0, nbe);
    if (UPS) run_phase(std::true_type{}, nbe, nblk);
    __builtin_amdgcn_s_setprio(0);
    conv_epilogue<NT, EPI, MAP>(a, acc, n, cb, ty0, tx0, wv, lane, smem);
}

// ------------------------------------------------------------------------------------------
// k_conv16b: the level-0 convolutions -- A_0 (e_0: 6 channels -> 48 columns) and the level-0
// gates ([e_0 (6), up(r_1) (48)] -> 16 columns) -- as BLOCK STEPS.  With 3..16 useful columns a
// tap step holds 4-12 MFMAs per wave, and k_conv3x3 spends the launch in the latency of 9-21
// barrier-separated staging steps (35-45 % matrix pipe use).  Here everything a block of input
// channels needs -- its patch and the weights of ALL its taps (and parity classes) -- is one
// LDS-DMA batch: one barrier pair per block, 32-108 MFMAs per wave between them, the next
// block's batch in flight meanwhile.  Same tiles, fmaf-chain order and epilogues as k_conv16.
//   block 0: the same-resolution source, stored 8 floats per pixel (6 real channels + 2 zeros):
//            2 quad planes = 11 pieces, 9 taps x 2 k-steps;
//   blocks 1..: 16 channels of the upsampled source: 7 pieces, 4 taps x 4 classes x 4 k-steps.
// Weight image per block: [tap][(class)][k-step][lane][NTI], NTI = 1 float per lane for one
// column tile (ds_read_b32), 4 otherwise (ds_read_b128).
static constexpr int E8_PIECES = 11;

__device__ __forceinline__ void wait_vm_n(int n) {  // wave-uniform n: leave the n youngest operations in flight
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
}

template <int NT>
struct C16b {
    static constexpr int NTI = NT == 1 ? 1 : 4;
    static constexpr int W8 = (18 * 64 * NTI * 4 + 1023) / 1024;  // weight pieces of block 0
    static constexpr int WU = 16 * NTI;                           // ... of an upsampled block
    static constexpr int bufp(bool ups) {
        return !ups ? E8_PIECES + W8 : (E8_PIECES + W8 > U16_PIECES + WU ? E8_PIECES + W8 : U16_PIECES + WU);
    }
};

template <int NT, int EPI, bool UPS>
__global__ __launch_bounds__(NTHR, 6) void k_conv16b(const ConvArgs a) {
    constexpr int MAP = EPI == EPI_POOL_ERR ? MAP_POOL : (UPS ? MAP_PARITY : MAP_LINEAR);
    constexpr int NTI = C16b<NT>::NTI, W8 = C16b<NT>::W8, WU = C16b<NT>::WU, BUFP = C16b<NT>::bufp(UPS);
    constexpr int NBUF = UPS ? 2 : 1;
    constexpr int SCRATCH = EPI == EPI_LSTM_PACKED ? 8 * 32 * 17 : 0;  // epilogue scratch (re-uses the buffers)
    __shared__ __attribute__((aligned(16))) float smem[NBUF * BUFP * 256 > SCRATCH ? NBUF * BUFP * 256 : SCRATCH];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile = bid % ntiles, n = bid / ntiles;
    const int ty0 = (tile / a.tiles_x) * 16, tx0 = (tile % a.tiles_x) * 16;
    const int nblk = 1 + (UPS ? a.src[1].cpt : 0);
    const int g = lane >> 4;

    // one LDS-DMA batch = patch pieces then weight pieces of block b, dealt round-robin to the
    // waves; returns how many this wave issued
    auto issue_block = [&](int b, int buf) -> int {
        float* dst = smem + buf * BUFP * 256;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int P = b == 0 ? E8_PIECES : U16_PIECES, T = P + (b == 0 ? W8 : WU);
        const float* wsrc = b == 0 ? a.Wblk + (long long)cb * W8 * 256
                                   : a.Wblk + ((long long)a.ncb * W8 + ((long long)(b - 1) * a.ncb + cb) * WU) * 256;
        int cnt = 0;
        for (int piece = wv; piece < T; piece += 8, ++cnt) {
            if (piece >= P) {
                glds16(wsrc + (piece - P) * 256 + ln * 4, dst + piece * 256);
            } else if (b == 0) {
                const ConvSrc& s = a.src[0];
                const int i = piece * 64 + ln, q = i / NP16, slot = i - q * NP16;
                const int y = slot / PW, xs = slot - y * PW;
                const int x = MAP == MAP_PARITY ? (xs < PW / 2 ? 2 * xs : 2 * (xs - PW / 2) + 1) : xs;
                const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                const bool ok = q < 2 && slot < PPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                glds16(ok ? s.p + (long long)n * s.nstride + ((long long)yy * a.W + xx) * s.pstride + 4 * q : a.zero, dst + piece * 256);
            } else {
                const ConvSrc& s = a.src[1];
                const int i = piece * 64 + ln, q = i / NPU16, slot = i - q * NPU16;
                const int Y = slot / LW, X = slot - Y * LW;
                const int ly = (ty0 >> 1) - 1 + Y, lx = (tx0 >> 1) - 1 + X;
                const bool ok = slot < LPIX && ly >= 0 && ly < (a.H >> 1) && lx >= 0 && lx < (a.W >> 1);
                glds16(ok ? s.p + (long long)n * s.nstride + ((long long)ly * (a.W >> 1) + lx) * s.pstride + (b - 1) * 16 + 4 * q : a.zero,
                       dst + piece * 256);
            }
        }
        return cnt;
    };

    issue_block(0, 0);

    // ---- accumulators (as in k_conv16; the loads overlap the first DMA batch)
    f32x4 acc[MT][NT];
    {
        const int col0 = cb * (NT * 16) + (lane & 15);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (a.init) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int py, px;
                        row_to_patch<MAP>(wv * 32 + mt * 16 + g * 4 + r, py, px);
                        const int y = ty0 + py, x = tx0 + px;
                        const long long pix = (y < a.H && x < a.W) ? (long long)y * a.W + x : 0;
                        acc[mt][nt][r] = a.init[pix * a.ncols + col0 + nt * 16];
                    }
                } else {
                    const float b = a.bias[col0 + nt * 16];
                    acc[mt][nt] = (f32x4){b, b, b, b};
                }
            }
    }
    int abase[MT], abase_lo[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int py, px;
        row_to_patch<MAP>(wv * 32 + mt * 16 + (lane & 15), py, px);
        abase[mt] = 4 * (py * PW + (MAP == MAP_PARITY ? (px >> 1) + (PW / 2) * (px & 1) : px)) + g;
        abase_lo[mt] = 4 * (((py >> 1) + (py & 1)) * LW + (px >> 1) + (px & 1)) + g;
    }
    const int wcls = MAP == MAP_PARITY ? (wv >> 1) : 0;
    int inflight = 0;
    if (nblk > 1) inflight = issue_block(1, 1);  // younger than everything block 0 waits for

    auto mfma_step = [&](const float (&fa)[MT], const float* wp) {
        float fb[4];
        if (NTI == 1) {
            fb[0] = wp[0];
        } else {
            const f32x4 t = *(const f32x4*)wp;
            fb[0] = t[0]; fb[1] = t[1]; fb[2] = t[2]; fb[3] = t[3];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
    };

    __builtin_amdgcn_s_setprio(1);
    // ---- block 0: 9 taps x 2 k-steps of the 8-wide same-resolution source
    wait_vm_n(inflight);
    wg_barrier();
    {
        const float* pa = smem;
        const float* wb = smem + E8_PIECES * 256 + lane * NTI;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const int xo = MAP == MAP_PARITY ? (dx == 1 ? ((wcls & 1) ? 1 - PW / 2 : PW / 2) : (dx >> 1)) : dx;
            const int toff = 4 * (dy * PW + xo);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float fa[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) fa[mt] = pa[abase[mt] + toff + kk * 4 * NP16];
                mfma_step(fa, wb + (tap * 2 + kk) * 64 * NTI);
            }
        }
    }
    // ---- blocks 1..: 4 collapsed taps x 4 k-steps of 16 channels of the upsampled source
    if (UPS) {
#pragma unroll 1
        for (int b = 1; b < nblk; ++b) {
            if (b + 1 < nblk) {  // buffer (b+1)&1 was last read by block b-1
                wg_barrier();
                inflight = issue_block(b + 1, (b + 1) & 1);
            } else {
                inflight = 0;
            }
            wait_vm_n(inflight);
            wg_barrier();
            const float* pa = smem + (b & 1) * BUFP * 256;
            const float* wb = pa + U16_PIECES * 256 + lane * NTI;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int toff = 4 * ((tap >> 1) * LW + (tap & 1));
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    float fa[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) fa[mt] = pa[abase_lo[mt] + toff + kk * 4 * NPU16];
                    mfma_step(fa, wb + ((tap * 4 + wcls) * 4 + kk) * 64 * NTI);
                }
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (EPI == EPI_LSTM_PACKED) wg_barrier();  // the staging buffers become the epilogue's scratch
    conv_epilogue<NT, EPI, MAP>(a, acc, n, cb, ty0, tx0, wv, lane, smem);
}

// Level-0 prediction Ahat_0 = min(relu(conv3x3(r_0)), 1) (prednet.py:268-271) with CIN, COUT <= 4:
// 81 fmaf per pixel do not need the matrix cores (the MFMA kernel pads K and N to 16 and spends
// its time in 9 barrier-separated staging steps: 55 us per launch at 512^2 x 4).  One thread per
// pixel, 18x18 halo tile in LDS, weights by scalar loads.  Same chain as the MFMA kernel and the
// oracle: acc = bias; for tap (ky, kx) ascending; for ci ascending: acc = fmaf(x, w, acc), where a
// tap outside the image multiplies a stored zero.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_conv_small(const ConvArgs a) {
    __shared__ float tile[PPIX * CIN];
    const int tid = threadIdx.x;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tileid = blockIdx.x % ntiles, n = blockIdx.x / ntiles;
    const int ty0 = (tileid / a.tiles_x) * 16, tx0 = (tileid % a.tiles_x) * 16;
    const float* base = a.src[0].p + (long long)n * a.src[0].nstride;
    for (int i = tid; i < PPIX * CIN; i += 256) {
        const int pp = i / CIN, ci = i - pp * CIN;
        const int yy = ty0 - 1 + pp / PW, xx = tx0 - 1 + pp % PW;
        tile[i] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? base[((long long)yy * a.W + xx) * CIN + ci] : 0.0f;
    }
    __syncthreads();
    const int py = tid >> 4, px = tid & 15;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = a.bias[co];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* x = tile + ((py + tap / 3) * PW + px + tap % 3) * CIN;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = __builtin_fmaf(x[ci], a.Wp[(tap * 16 + ci) * a.ncols + co], acc[co]);
    }
    const int y = ty0 + py, xq = tx0 + px;
    if (y < a.H && xq < a.W) {
        float* o = a.out0 + (long long)(a.out_idx ? a.out_idx[n] : n) * a.out0_nstride + ((long long)y * a.W + xq) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = tz_relu(acc[co]);
            if (a.clip1 && v > 1.0f) v = 1.0f;
            o[co] = v;
        }
    }
}

// level-0 error unit (prednet.py:274-277 with a = input frame, Ahat = Ahat_0(t0)):
// input is either a key frame (uint8, unpadded; x = float32(k)/255, compress.py:138) or a
// padded float32 frame of the prediction stack (compress.py:222).
__global__ __launch_bounds__(256) void k_err0(const uint8_t* __restrict__ frames_u8, int H, int W,
                                              const float* __restrict__ in_stack, const int* __restrict__ is_key,
                                              const int* __restrict__ in_idx, const float* __restrict__ ahat0, int Hp,
                                              int Wp, int C, int Cs, float* __restrict__ e0) {
    int n = blockIdx.y;
    long long npx = (long long)Hp * Wp;
    const bool key = is_key[n] != 0;
    const long long fi = in_idx[n];
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (long long)gridDim.x * blockDim.x) {
        int y = (int)(p / Wp), x = (int)(p - (long long)y * Wp);
        float* o = e0 + ((long long)n * npx + p) * Cs;
        if (C == 3 && Cs == 8) {  // RGB frames (compress.py:114): one pixel = two 16-byte stores
            float av[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (key) av[c] = (y < H && x < W) ? (float)frames_u8[(fi * H * W + (long long)y * W + x) * 3 + c] / 255.0f : 0.0f;
                else av[c] = in_stack[(fi * npx + p) * 3 + c];
            }
            float d1[3], d2[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float h = ahat0[p * 3 + c];
                d1[c] = h - av[c];
                d2[c] = av[c] - h;
            }
            *(float4*)o = make_float4(tz_relu(d1[0]), tz_relu(d1[1]), tz_relu(d1[2]), tz_relu(d2[0]));
            *(float4*)(o + 4) = make_float4(tz_relu(d2[1]), tz_relu(d2[2]), 0.0f, 0.0f);
            continue;
        }
        for (int c = 0; c < C; ++c) {
            float av;
            if (key) {
                av = 0.0f;
                if (y < H && x < W) av = (float)frames_u8[(fi * H * W + (long long)y * W + x) * C + c] / 255.0f;
            } else {
                av = in_stack[(fi * npx + p) * C + c];
            }
            float h = ahat0[p * C + c];
            float d1 = h - av, d2 = av - h;
            o[c] = tz_relu(d1);
            o[C + c] = tz_relu(d2);
        }
        for (int c = 2 * C; c < Cs; ++c) o[c] = 0.0f;  // stride padding reads as zero channels
    }
}

// ------------------------------------------------------------------------------- host side
struct Seg {
    int row_off, C;
    int up;  // 1: half-resolution source read through a x2 nearest upsample (collapsed taps)
};
struct PackedConv {
    float* d_W = nullptr;
    float* d_Wimg = nullptr;  // LDS image order for k_conv16 (every source a multiple of 16 channels), else null
    float* d_Wblk = nullptr;  // block-step image for k_conv16b (first source <= 8 channels at stride 8), else null
    const float* d_zero = nullptr;
    float* d_bias = nullptr;
    int nslots = 0, ncols = 0, NT = 1, ncb = 1;
    std::vector<Seg> segs;
};

struct tz_model {
    int L = 0;
    int stack[TZ_MAX_LEVELS] = {0}, rstack[TZ_MAX_LEVELS] = {0};
    std::vector<std::vector<float>> w;  // Keras list order
    int Hp = 0, Wp = 0, maxB = 0;
    int cap = 0;  // windows advanced together (<= maxB, the allocated batch)
    bool prepared = false;
    float *R0[TZ_MAX_LEVELS] = {0}, *C0[TZ_MAX_LEVELS] = {0}, *Ahat0[TZ_MAX_LEVELS] = {0}, *G0[TZ_MAX_LEVELS] = {0};
    float *E[TZ_MAX_LEVELS] = {0}, *R1[TZ_MAX_LEVELS] = {0};
    PackedConv a_conv[TZ_MAX_LEVELS], gate_t1[TZ_MAX_LEVELS], ahat0_t1;
    float* d_zero = nullptr;  // zero page for LDS-DMA halo pixels
    int e0s = 0;              // floats per pixel of E[0]: 2*stack[0] rounded up to 8 (k_conv16b reads 16-byte quads)
    int* d_idx = nullptr;  // 3*maxB ints: is_key, in_idx, out_idx
    std::vector<void*> allocs;
    // weight list accessors
    const float* a_k(int l) const { return w[2 * l].data(); }
    const float* a_b(int l) const { return w[2 * l + 1].data(); }
    const float* ahat_k(int l) const { return w[2 * (L - 1) + 2 * l].data(); }
    const float* ahat_b(int l) const { return w[2 * (L - 1) + 2 * l + 1].data(); }
    // gate g in our order 0=i 1=f 2=c 3=o ; list order is c, f, i, o (prednet.py:212)
    int gate_base(int g) const {
        static const int pos[4] = {2, 1, 0, 3};
        return 2 * (L - 1) + 2 * L + pos[g] * 2 * L;
    }
    const float* g_k(int g, int l) const { return w[gate_base(g) + 2 * l].data(); }
    const float* g_b(int g, int l) const { return w[gate_base(g) + 2 * l + 1].data(); }
    int gate_cin(int l) const { return rstack[l] + 2 * stack[l] + (l < L - 1 ? rstack[l + 1] : 0); }
};

static int dmalloc(tz_ctx* ctx, tz_model* m, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return tz_fail(ctx, TZ_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    m->allocs.push_back(*p);
    return TZ_OK;
}

struct ColSrc {
    const float* kernel;  // HWIO (3,3,Cin,Cout)
    const float* bias;
    int Cin, Cout, ch;    // ch < 0: zero padding column
};

// Pack weights into the chunk order walked by k_conv3x3: for seg, 16-channel block, tap.
// 3x3 taps of an upsampled source that read the same half-resolution pixel, for output parity
// a (0 even / 1 odd coordinate) and collapsed tap d (0/1): a=0: {0},{1,2}; a=1: {0,1},{2}.
static int collapse_set(int a, int d, int out[2]) {
    if (a == 0) {
        if (d == 0) { out[0] = 0; return 1; }
        out[0] = 1; out[1] = 2; return 2;
    }
    if (d == 0) { out[0] = 0; out[1] = 1; return 2; }
    out[0] = 2; return 1;
}

// Pack weights into the slot order walked by k_conv3x3: for seg, 16-channel block, then either
// 9 taps (same-resolution source) or 4 collapsed taps x 4 parity classes (upsampled source,
// weights summed in float32 in ascending (ky, kx) order: part of TZ-PA1).
static int pack_conv(tz_ctx* ctx, tz_model* m, const std::vector<Seg>& segs, const std::vector<ColSrc>& cols, int NT,
                     PackedConv* pc) {
    int ncols = (int)cols.size();
    int nslots = 0;
    for (auto& s : segs) nslots += (s.up ? 16 : 9) * ((s.C + 15) / 16);
    std::vector<float> W((size_t)std::max(nslots, 1) * 16 * ncols, 0.0f), B(ncols, 0.0f);
    int slot = 0;
    auto put = [&](const Seg& s, int c0, const int* kys, int nky, const int* kxs, int nkx) {
        for (int kc = 0; kc < 16 && c0 + kc < s.C; ++kc)
            for (int col = 0; col < ncols; ++col) {
                const ColSrc& cs = cols[col];
                if (cs.ch < 0) continue;
                float v = 0.0f;
                bool first = true;
                for (int iy = 0; iy < nky; ++iy)
                    for (int ix = 0; ix < nkx; ++ix) {
                        float w = cs.kernel[((size_t)(kys[iy] * 3 + kxs[ix]) * cs.Cin + s.row_off + c0 + kc) * cs.Cout + cs.ch];
                        v = first ? w : v + w;
                        first = false;
                    }
                W[((size_t)slot * 16 + kc) * ncols + col] = v;
            }
        ++slot;
    };
    for (auto& s : segs)
        for (int c0 = 0; c0 < s.C; c0 += 16) {
            if (!s.up) {
                for (int tap = 0; tap < 9; ++tap) {
                    int ky = tap / 3, kx = tap % 3;
                    put(s, c0, &ky, 1, &kx, 1);
                }
            } else {
                for (int tp = 0; tp < 4; ++tp)
                    for (int cls = 0; cls < 4; ++cls) {
                        int kys[2], kxs[2];
                        int nky = collapse_set(cls >> 1, tp >> 1, kys), nkx = collapse_set(cls & 1, tp & 1, kxs);
                        put(s, c0, kys, nky, kxs, nkx);
                    }
            }
        }
    for (int col = 0; col < ncols; ++col)
        if (cols[col].ch >= 0) B[col] = cols[col].bias[cols[col].ch];
    pc->nslots = nslots;
    pc->ncols = ncols;
    pc->NT = NT;
    pc->ncb = ncols / (16 * NT);
    pc->segs = segs;
    TZ_TRY(dmalloc(ctx, m, (void**)&pc->d_W, W.size() * 4));
    TZ_TRY(dmalloc(ctx, m, (void**)&pc->d_bias, B.size() * 4));
    TZ_HIP(ctx, hipMemcpy(pc->d_W, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    TZ_HIP(ctx, hipMemcpy(pc->d_bias, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    // k_conv16's LDS image: [slot][column block][k-step][lane = (k row & 3) * 16 + column][column tile]
    bool all16 = nslots > 0 && NT >= 3;
    for (auto& s : segs) all16 = all16 && (s.C % 16) == 0;
    if (all16) {
        const int ncb = pc->ncb;
        std::vector<float> I((size_t)nslots * ncb * 4 * 256, 0.0f);
        for (int sl = 0; sl < nslots; ++sl)
            for (int cb = 0; cb < ncb; ++cb)
                for (int kk = 0; kk < 4; ++kk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int nt = 0; nt < NT; ++nt)
                            I[((((size_t)sl * ncb + cb) * 4 + kk) * 64 + lane) * 4 + nt] =
                                W[((size_t)sl * 16 + 4 * kk + (lane >> 4)) * ncols + cb * NT * 16 + nt * 16 + (lane & 15)];
        TZ_TRY(dmalloc(ctx, m, (void**)&pc->d_Wimg, I.size() * 4));
        TZ_HIP(ctx, hipMemcpy(pc->d_Wimg, I.data(), I.size() * 4, hipMemcpyHostToDevice));
        pc->d_zero = m->d_zero;
    }
    // k_conv16b's image: block 0 = the <= 8-channel same-resolution source [cb][tap][k-step 0..1][lane][NTI]
    // (padded to whole 1 KB pieces), then per 16-channel block of the upsampled source
    // [block][cb][tap][class][k-step][lane][NTI]
    const bool blk_ok = !segs.empty() && !segs[0].up && segs[0].C <= 8 && (NT == 1 || NT >= 3) &&
                        (segs.size() == 1 || (segs.size() == 2 && segs[1].up && segs[1].C % 16 == 0));
    if (blk_ok) {
        const int ncb = pc->ncb, NTI = NT == 1 ? 1 : 4;
        const int W8 = (18 * 64 * NTI * 4 + 1023) / 1024, WU = 16 * NTI;
        const int nub = segs.size() == 2 ? segs[1].C / 16 : 0;
        std::vector<float> I(((size_t)ncb * W8 + (size_t)nub * ncb * WU) * 256, 0.0f);
        auto wv = [&](int sl, int k, int cb, int nt, int lane) {
            return W[((size_t)sl * 16 + k) * ncols + cb * NT * 16 + nt * 16 + (lane & 15)];
        };
        for (int cb = 0; cb < ncb; ++cb)
            for (int tap = 0; tap < 9; ++tap)
                for (int kk = 0; kk < 2; ++kk)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int nt = 0; nt < std::min(NT, NTI); ++nt)
                            I[(size_t)cb * W8 * 256 + (((size_t)tap * 2 + kk) * 64 + lane) * NTI + nt] = wv(tap, 4 * kk + (lane >> 4), cb, nt, lane);
        for (int b = 0; b < nub; ++b)
            for (int cb = 0; cb < ncb; ++cb)
                for (int tap = 0; tap < 4; ++tap)
                    for (int cls = 0; cls < 4; ++cls)
                        for (int kk = 0; kk < 4; ++kk)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int nt = 0; nt < std::min(NT, NTI); ++nt)
                                    I[((size_t)ncb * W8 + ((size_t)b * ncb + cb) * WU) * 256 +
                                      ((((size_t)tap * 4 + cls) * 4 + kk) * 64 + lane) * NTI + nt] =
                                        wv(9 + 16 * b + 4 * tap + cls, 4 * kk + (lane >> 4), cb, nt, lane);
        TZ_TRY(dmalloc(ctx, m, (void**)&pc->d_Wblk, I.size() * 4));
        TZ_HIP(ctx, hipMemcpy(pc->d_Wblk, I.data(), I.size() * 4, hipMemcpyHostToDevice));
        pc->d_zero = m->d_zero;
    }
    return TZ_OK;
}

static int plain_nt(int Cout) {
    if (Cout % 64 == 0) return 4;
    if (Cout % 48 == 0) return 3;
    return 1;
}

static std::vector<ColSrc> plain_cols(const float* k, const float* b, int Cin, int Cout) {
    int ncols = ((Cout + 15) / 16) * 16;
    std::vector<ColSrc> cols(ncols);
    for (int c = 0; c < ncols; ++c) cols[c] = ColSrc{k, b, Cin, Cout, c < Cout ? c : -1};
    return cols;
}

// packed gate columns: R%16==0 -> [cg][gate][16]; R<=4 -> [gate][R] in one 16-wide tile
static int gate_cols(tz_ctx* ctx, const tz_model* m, int l, std::vector<ColSrc>* cols, int* NT) {
    int R = m->rstack[l], Cin = m->gate_cin(l);
    if (R % 16 == 0) {
        cols->resize((size_t)4 * R);
        for (int cg = 0; cg < R / 16; ++cg)
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 16; ++j)
                    (*cols)[cg * 64 + g * 16 + j] = ColSrc{m->g_k(g, l), m->g_b(g, l), Cin, R, cg * 16 + j};
        *NT = 4;
        return TZ_OK;
    }
    if (R <= 4) {
        cols->assign(16, ColSrc{nullptr, nullptr, 0, 0, -1});
        for (int g = 0; g < 4; ++g)
            for (int c = 0; c < R; ++c) (*cols)[g * R + c] = ColSrc{m->g_k(g, l), m->g_b(g, l), Cin, R, c};
        *NT = 1;
        return TZ_OK;
    }
    return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "R_stack_sizes[%d]=%d: need a multiple of 16 or <= 4", l, R);
}

template <int NT, int EPI, bool UPS, bool FULLK>
static void launch_conv_t(tz_ctx* ctx, const ConvArgs& a, int nbatch) {
    int blocks = a.ncb * a.tiles_x * a.tiles_y * nbatch;
    hipLaunchKernelGGL((k_conv3x3<NT, EPI, UPS, FULLK>), dim3(blocks), dim3(NTHR), 0, ctx->stream, a);
}

template <int NT, int EPI, bool UPS>
static void launch_conv16_t(tz_ctx* ctx, const ConvArgs& a, int nbatch) {
    int blocks = a.ncb * a.tiles_x * a.tiles_y * nbatch;
    hipLaunchKernelGGL((k_conv16<NT, EPI, UPS>), dim3(blocks), dim3(NTHR), 0, ctx->stream, a);
}

extern "C" int tz_set_conv_impl(tz_ctx* ctx, int lds_dma) {
    if (!ctx) return TZ_ERR_INVALID;
    ctx->conv_impl = lds_dma ? 1 : 0;
    return TZ_OK;
}

static int launch_conv(tz_ctx* ctx, int NT, int epi, const ConvArgs& a, int nbatch) {
    tz_prof_scope ps(ctx, TZP_CONV);
    bool ups = false, fullk = true;
    for (int s = 0; s < a.nsrc; ++s) {
        ups = ups || a.src[s].up;
        fullk = fullk && (a.src[s].C % 16) == 0;
    }
    if (epi == EPI_RELU && a.nsrc == 1 && !ups && a.src[0].C == 3 && a.Cout == 3 && ctx->conv_impl) {
        hipLaunchKernelGGL((k_conv_small<3, 3>), dim3(a.tiles_x * a.tiles_y * nbatch), dim3(256), 0, ctx->stream, a);
        TZ_HIP(ctx, hipGetLastError());
        return TZ_OK;
    }
    if (a.Wblk && a.nsrc > 0 && !a.src[0].up && a.src[0].pstride == 8 && ctx->conv_impl) {
        const int blocks = a.ncb * a.tiles_x * a.tiles_y * nbatch;
#define TZ_CASE16B(nt, e, u)                                                                                    \
    if (NT == nt && epi == e && ups == u) {                                                                     \
        hipLaunchKernelGGL((k_conv16b<nt, e, u>), dim3(blocks), dim3(NTHR), 0, ctx->stream, a);                 \
        TZ_HIP(ctx, hipGetLastError());                                                                         \
        return TZ_OK;                                                                                           \
    }
        TZ_CASE16B(1, EPI_LSTM_PACKED, true) TZ_CASE16B(1, EPI_LSTM_PACKED, false)
        TZ_CASE16B(1, EPI_POOL_ERR, false) TZ_CASE16B(3, EPI_POOL_ERR, false) TZ_CASE16B(4, EPI_POOL_ERR, false)
#undef TZ_CASE16B
    }
    if (a.Wimg && a.nsrc > 0 && fullk && ctx->conv_impl) {
#define TZ_CASE16(nt, e, u)                          \
    if (NT == nt && epi == e && ups == u) {          \
        launch_conv16_t<nt, e, u>(ctx, a, nbatch);   \
        TZ_HIP(ctx, hipGetLastError());              \
        return TZ_OK;                                \
    }
        TZ_CASE16(4, EPI_LSTM, false) TZ_CASE16(4, EPI_LSTM, true)
        TZ_CASE16(3, EPI_POOL_ERR, false) TZ_CASE16(4, EPI_POOL_ERR, false)
#undef TZ_CASE16
    }
#define TZ_CASE(nt, e, u)                                                   \
    if (NT == nt && epi == e && ups == u) {                                 \
        if (fullk) launch_conv_t<nt, e, u, true>(ctx, a, nbatch);           \
        else launch_conv_t<nt, e, u, false>(ctx, a, nbatch);                \
        TZ_HIP(ctx, hipGetLastError());                                     \
        return TZ_OK;                                                       \
    }
    TZ_CASE(1, EPI_RAW, false) TZ_CASE(4, EPI_RAW, false)
    TZ_CASE(1, EPI_RELU, false) TZ_CASE(3, EPI_RELU, false) TZ_CASE(4, EPI_RELU, false)
    TZ_CASE(4, EPI_LSTM, false) TZ_CASE(4, EPI_LSTM, true)
    TZ_CASE(1, EPI_LSTM_PACKED, false) TZ_CASE(1, EPI_LSTM_PACKED, true)
    TZ_CASE(1, EPI_POOL_ERR, false) TZ_CASE(3, EPI_POOL_ERR, false) TZ_CASE(4, EPI_POOL_ERR, false)
#undef TZ_CASE
    return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "no conv kernel for NT=%d epilogue=%d upsampled=%d", NT, epi, (int)ups);
}

static void fill_srcs(ConvArgs& a, const PackedConv& pc, const float* const* ptrs, const long long* nstrides,
                      int pstride0 = 0) {
    a.nsrc = (int)pc.segs.size();
    for (int s = 0; s < a.nsrc; ++s) {
        a.src[s].p = ptrs[s];
        a.src[s].nstride = nstrides[s];
        a.src[s].C = pc.segs[s].C;
        a.src[s].pstride = s == 0 && pstride0 ? pstride0 : pc.segs[s].C;
        a.src[s].up = pc.segs[s].up;
        a.src[s].cpt = (pc.segs[s].C + 15) / 16;
    }
    a.Wp = pc.d_W;
    a.Wimg = pc.d_Wimg;
    a.Wblk = pc.d_Wblk;
    a.zero = pc.d_zero;
    a.bias = pc.d_bias;
    a.ncols = pc.ncols;
    a.ncb = pc.ncb;
}

static void set_geom(ConvArgs& a, int H, int W) {
    a.H = H;
    a.W = W;
    a.tiles_x = (W + 15) / 16;
    a.tiles_y = (H + 15) / 16;
}

void tz_model_free(tz_ctx* ctx) {
    tz_model* m = ctx->model;
    if (!m) return;
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
    ctx->model = nullptr;
}

extern "C" int tz_model_load(tz_ctx* ctx, int nb_layers, const int* stack_sizes, const int* r_stack_sizes,
                             const float* const* weights) {
    if (!ctx) return TZ_ERR_INVALID;
    if (nb_layers < 1 || nb_layers > TZ_MAX_LEVELS || !stack_sizes || !r_stack_sizes || !weights)
        return tz_fail(ctx, TZ_ERR_INVALID, "bad model description");
    if (stack_sizes[0] != 3) return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "stack_sizes[0] must be 3 (RGB frames, compress.py:114)");
    tz_model_free(ctx);
    tz_model* m = new tz_model();
    ctx->model = m;
    m->L = nb_layers;
    for (int l = 0; l < nb_layers; ++l) {
        m->stack[l] = stack_sizes[l];
        m->rstack[l] = r_stack_sizes[l];
        if (stack_sizes[l] < 1 || r_stack_sizes[l] < 1) return tz_fail(ctx, TZ_ERR_INVALID, "bad channel count");
        if (l > 0 && stack_sizes[l] % 16) return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "stack_sizes[%d] must be a multiple of 16", l);
        if (!(r_stack_sizes[l] % 16 == 0 || r_stack_sizes[l] <= 4))
            return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "R_stack_sizes[%d] must be a multiple of 16 or <= 4", l);
    }
    const int L = nb_layers;
    int k = 0;
    auto take = [&](size_t count) {
        m->w.emplace_back(weights[k], weights[k] + count);
        ++k;
    };
    for (int l = 0; l < L - 1; ++l) {  // a
        take((size_t)9 * 2 * m->stack[l] * m->stack[l + 1]);
        take(m->stack[l + 1]);
    }
    for (int l = 0; l < L; ++l) {  // ahat
        take((size_t)9 * m->rstack[l] * m->stack[l]);
        take(m->stack[l]);
    }
    for (int g = 0; g < 4; ++g)  // c, f, i, o
        for (int l = 0; l < L; ++l) {
            take((size_t)9 * m->gate_cin(l) * m->rstack[l]);
            take(m->rstack[l]);
        }
    return TZ_OK;
}

extern "C" int tz_model_prepare(tz_ctx* ctx, int Hp, int Wp, int max_batch) {
    if (!ctx) return TZ_ERR_INVALID;
    tz_model* m = ctx->model;
    if (!m) return tz_fail(ctx, TZ_ERR_STATE, "tz_model_prepare before tz_model_load");
    const int L = m->L;
    if (Hp <= 0 || Wp <= 0 || max_batch < 1) return tz_fail(ctx, TZ_ERR_INVALID, "bad prepare arguments");
    if ((Hp % (1 << (L - 1))) || (Wp % (1 << (L - 1))) || (Hp % 8) || (Wp % 8))
        return tz_fail(ctx, TZ_ERR_INVALID,
                       "Image size is out of scope for this model: padded size %dx%d must divide by 8 and 2^(levels-1)", Hp, Wp);
    if (m->prepared && m->Hp == Hp && m->Wp == Wp && m->maxB >= max_batch) {
        m->cap = max_batch;
        return TZ_OK;
    }
    // rebuild from the kept host weights
    {
        std::vector<std::vector<float>> w = std::move(m->w);
        int st[TZ_MAX_LEVELS], rs[TZ_MAX_LEVELS];
        for (int l = 0; l < L; ++l) { st[l] = m->stack[l]; rs[l] = m->rstack[l]; }
        tz_model_free(ctx);
        m = new tz_model();
        ctx->model = m;
        m->L = L;
        for (int l = 0; l < L; ++l) { m->stack[l] = st[l]; m->rstack[l] = rs[l]; }
        m->w = std::move(w);
    }
    m->Hp = Hp;
    m->Wp = Wp;
    m->maxB = max_batch;
    m->cap = max_batch;
    auto hl = [&](int l) { return Hp >> l; };
    auto wl = [&](int l) { return Wp >> l; };
    for (int l = 0; l < L; ++l) {
        size_t npx = (size_t)hl(l) * wl(l);
        int R = m->rstack[l];
        TZ_TRY(dmalloc(ctx, m, (void**)&m->R0[l], npx * R * 4));
        TZ_TRY(dmalloc(ctx, m, (void**)&m->C0[l], npx * R * 4));
        TZ_TRY(dmalloc(ctx, m, (void**)&m->Ahat0[l], npx * m->stack[l] * 4));
        const int ec = l == 0 ? (m->e0s = (2 * m->stack[0] + 7) / 8 * 8) : 2 * m->stack[l];
        TZ_TRY(dmalloc(ctx, m, (void**)&m->E[l], (size_t)max_batch * npx * ec * 4));
        TZ_TRY(dmalloc(ctx, m, (void**)&m->R1[l], (size_t)max_batch * npx * R * 4));
    }
    TZ_TRY(dmalloc(ctx, m, (void**)&m->d_idx, sizeof(int) * 3 * max_batch));
    TZ_TRY(dmalloc(ctx, m, (void**)&m->d_zero, 256));
    TZ_HIP(ctx, hipMemsetAsync(m->d_zero, 0, 256, ctx->stream));

    // ---- t=0 top-down from zero state (prednet.py:143-190, 249-264): only r_up is non-zero
    for (int l = L - 1; l >= 0; --l) {
        std::vector<ColSrc> cols;
        int NT;
        TZ_TRY(gate_cols(ctx, m, l, &cols, &NT));
        std::vector<Seg> segs;
        if (l < L - 1) segs.push_back(Seg{m->rstack[l] + 2 * m->stack[l], m->rstack[l + 1], 1});
        PackedConv pc;
        TZ_TRY(pack_conv(ctx, m, segs, cols, NT, &pc));
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const float* ptrs[2] = {l < L - 1 ? m->R0[l + 1] : nullptr, nullptr};
        long long ns[2] = {0, 0};
        fill_srcs(a, pc, ptrs, ns);
        set_geom(a, hl(l), wl(l));
        a.Cout = m->rstack[l];
        a.R = m->rstack[l];
        a.out0 = m->R0[l];
        a.out1 = m->C0[l];
        TZ_TRY(launch_conv(ctx, NT, NT == 4 ? EPI_LSTM : EPI_LSTM_PACKED, a, 1));
    }
    // ---- Ahat_l at t=0 (prednet.py:268-271)
    for (int l = 0; l < L; ++l) {
        int Cout = m->stack[l], NT = plain_nt(Cout);
        PackedConv pc;
        TZ_TRY(pack_conv(ctx, m, {Seg{0, m->rstack[l], 0}}, plain_cols(m->ahat_k(l), m->ahat_b(l), m->rstack[l], Cout), NT, &pc));
        if (l == 0) m->ahat0_t1 = pc;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const float* ptrs[2] = {m->R0[l], nullptr};
        long long ns[2] = {0, 0};
        fill_srcs(a, pc, ptrs, ns);
        set_geom(a, hl(l), wl(l));
        a.Cout = Cout;
        a.out0 = m->Ahat0[l];
        a.clip1 = l == 0;
        TZ_TRY(launch_conv(ctx, NT, EPI_RELU, a, 1));
    }
    // ---- G0_l = bias + conv over the r_tm1 rows of the t=1 gate convolution, and the t=1 packs
    for (int l = 0; l < L; ++l) {
        std::vector<ColSrc> cols;
        int NT;
        TZ_TRY(gate_cols(ctx, m, l, &cols, &NT));
        PackedConv pg;
        TZ_TRY(pack_conv(ctx, m, {Seg{0, m->rstack[l], 0}}, cols, NT, &pg));
        size_t npx = (size_t)hl(l) * wl(l);
        TZ_TRY(dmalloc(ctx, m, (void**)&m->G0[l], npx * pg.ncols * 4));
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const float* ptrs[2] = {m->R0[l], nullptr};
        long long ns[2] = {0, 0};
        fill_srcs(a, pg, ptrs, ns);
        set_geom(a, hl(l), wl(l));
        a.out0 = m->G0[l];
        TZ_TRY(launch_conv(ctx, NT, EPI_RAW, a, 1));
        std::vector<Seg> segs = {Seg{m->rstack[l], 2 * m->stack[l], 0}};
        if (l < L - 1) segs.push_back(Seg{m->rstack[l] + 2 * m->stack[l], m->rstack[l + 1], 1});
        TZ_TRY(pack_conv(ctx, m, segs, cols, NT, &m->gate_t1[l]));
    }
    // ---- A convs (prednet.py:290)
    for (int l = 0; l < L - 1; ++l) {
        int Cout = m->stack[l + 1], NT = plain_nt(Cout);
        TZ_TRY(pack_conv(ctx, m, {Seg{0, 2 * m->stack[l], 0}}, plain_cols(m->a_k(l), m->a_b(l), 2 * m->stack[l], Cout), NT,
                         &m->a_conv[l]));
    }
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->prepared = true;
    return TZ_OK;
}

int tz_model_dims(tz_ctx* ctx, int* Hp, int* Wp, int* max_batch) {
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    *Hp = m->Hp;
    *Wp = m->Wp;
    *max_batch = m->cap;
    return TZ_OK;
}

int tz_model_c0_dev(tz_ctx* ctx, const float** c0) {
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    *c0 = m->Ahat0[0];
    return TZ_OK;
}

// One predictor step for n <= maxB independent frames (the `predict` seam, compress.py:227).
// Input n: key frame h_in_idx[n] of d_frames_u8 (h_in_is_key) or padded f32 frame
// h_in_idx[n] of d_in_stack; output goes to frame slot h_out_idx[n] of d_out_stack.
int tz_model_predict_batch(tz_ctx* ctx, int n, const int* h_in_is_key, const int* h_in_idx, const int* h_out_idx,
                           const uint8_t* d_frames_u8, int H, int W, const float* d_in_stack, float* d_out_stack) {
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    if (n < 1 || n > m->maxB) return tz_fail(ctx, TZ_ERR_INVALID, "batch %d outside 1..%d", n, m->maxB);
    std::vector<int> idx(3 * (size_t)m->maxB, 0);
    for (int i = 0; i < n; ++i) {
        idx[i] = h_in_is_key[i];
        idx[m->maxB + i] = h_in_idx[i];
        idx[2 * m->maxB + i] = h_out_idx[i];
    }
    TZ_TRY(tz_upload(ctx, m->d_idx, idx.data(), sizeof(int) * idx.size()));
    return tz_model_predict_batch_dev(ctx, n, m->d_idx, m->maxB, d_frames_u8, H, W, d_in_stack, d_out_stack);
}

// Launch-only form: d_idx holds [is_key | in_idx | out_idx], each `stride` ints apart, already
// on the device.  Nothing here allocates or copies.
int tz_model_predict_batch_dev(tz_ctx* ctx, int n, const int* d_idx, int stride, const uint8_t* d_frames_u8, int H, int W,
                               const float* d_in_stack, float* d_out_stack) {
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    if (n < 1 || n > m->maxB) return tz_fail(ctx, TZ_ERR_INVALID, "batch %d outside 1..%d", n, m->maxB);
    const int L = m->L, Hp = m->Hp, Wp = m->Wp;
    auto hl = [&](int l) { return Hp >> l; };
    auto wl = [&](int l) { return Wp >> l; };
    auto npx = [&](int l) { return (long long)hl(l) * wl(l); };
    {
        tz_prof_scope ps(ctx, TZP_ERR0);
        int gx = (int)std::min<long long>((npx(0) + 255) / 256, 2048);
        hipLaunchKernelGGL(k_err0, dim3(gx, n), dim3(256), 0, ctx->stream, d_frames_u8, H, W, d_in_stack, d_idx,
                           d_idx + stride, m->Ahat0[0], Hp, Wp, m->stack[0], m->e0s, m->E[0]);
        TZ_HIP(ctx, hipGetLastError());
    }
    for (int l = 0; l < L - 1; ++l) {  // t0 bottom-up
        const PackedConv& pc = m->a_conv[l];
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const float* ptrs[2] = {m->E[l], nullptr};
        const int ec = l == 0 ? m->e0s : 2 * m->stack[l];  // floats per pixel of E[l]
        long long ns[2] = {npx(l) * ec, 0};
        fill_srcs(a, pc, ptrs, ns, ec);
        set_geom(a, hl(l), wl(l));
        a.Cout = m->stack[l + 1];
        a.aux = m->Ahat0[l + 1];
        a.out0 = m->E[l + 1];
        a.out0_nstride = npx(l + 1) * 2 * m->stack[l + 1];
        TZ_TRY(launch_conv(ctx, pc.NT, EPI_POOL_ERR, a, n));
    }
    for (int l = L - 1; l >= 0; --l) {  // t1 top-down
        const PackedConv& pc = m->gate_t1[l];
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const float* ptrs[2] = {m->E[l], l < L - 1 ? m->R1[l + 1] : nullptr};
        const int ec = l == 0 ? m->e0s : 2 * m->stack[l];
        long long ns[2] = {npx(l) * ec, l < L - 1 ? npx(l + 1) * m->rstack[l + 1] : 0};
        fill_srcs(a, pc, ptrs, ns, ec);
        set_geom(a, hl(l), wl(l));
        a.init = m->G0[l];
        a.Cout = m->rstack[l];
        a.R = m->rstack[l];
        a.aux = m->C0[l];
        a.out0 = m->R1[l];
        a.out0_nstride = npx(l) * m->rstack[l];
        TZ_TRY(launch_conv(ctx, pc.NT, pc.NT == 4 ? EPI_LSTM : EPI_LSTM_PACKED, a, n));
    }
    {  // Ahat_0 at t1 = the prediction (prednet.py:268-271, 293-295)
        const PackedConv& pc = m->ahat0_t1;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const float* ptrs[2] = {m->R1[0], nullptr};
        long long ns[2] = {npx(0) * m->rstack[0], 0};
        fill_srcs(a, pc, ptrs, ns);
        set_geom(a, Hp, Wp);
        a.Cout = m->stack[0];
        a.out0 = d_out_stack;
        a.out0_nstride = npx(0) * m->stack[0];
        a.out_idx = d_idx + 2 * stride;
        a.clip1 = 1;
        TZ_TRY(launch_conv(ctx, pc.NT, EPI_RELU, a, n));
    }
    return TZ_OK;
}

extern "C" int tz_predict_c0(tz_ctx* ctx, float* out) {
    if (!ctx || !out) return TZ_ERR_INVALID;
    const float* c0;
    TZ_TRY(tz_model_c0_dev(ctx, &c0));
    tz_model* m = ctx->model;
    size_t bytes = (size_t)m->Hp * m->Wp * m->stack[0] * 4;
    TZ_HIP(ctx, hipMemcpyAsync(out, c0, bytes, hipMemcpyDefault, ctx->stream));
    if (!tz_is_device_ptr(out)) TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TZ_OK;
}

extern "C" int tz_predict_next(tz_ctx* ctx, const float* frames, int n, float* out) {
    if (!ctx || !frames || !out || n < 0) return TZ_ERR_INVALID;
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    size_t fe = (size_t)m->Hp * m->Wp * m->stack[0];
    const void* din;
    tz_out o;
    std::vector<tz_out> outs;
    int rc = tz_dev_in(ctx, frames, fe * n * 4, &din);
    if (rc == TZ_OK) rc = tz_dev_out(ctx, out, fe * n * 4, &o);
    if (rc == TZ_OK) {
        outs.push_back(o);
        for (int b0 = 0; b0 < n && rc == TZ_OK; b0 += m->maxB) {
            int nb = std::min(m->maxB, n - b0);
            std::vector<int> isk(nb, 0), ii(nb), oi(nb);
            for (int i = 0; i < nb; ++i) ii[i] = oi[i] = b0 + i;
            rc = tz_model_predict_batch(ctx, nb, isk.data(), ii.data(), oi.data(), nullptr, 0, 0, (const float*)din,
                                        (float*)o.dev);
        }
    }
    if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    tz_pool_release_all(ctx);
    return rc;
}

extern "C" int tz_predict_tap(tz_ctx* ctx, int kind, int level, float* out) {
    if (!ctx || !out) return TZ_ERR_INVALID;
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    if (level < 0 || level >= m->L || kind < 0 || kind > 1) return tz_fail(ctx, TZ_ERR_INVALID, "bad tap");
    size_t npx = (size_t)(m->Hp >> level) * (m->Wp >> level);
    const float* src = kind == 0 ? m->E[level] : m->R1[level];
    const size_t ch = kind == 0 ? 2 * m->stack[level] : m->rstack[level];
    const size_t sch = kind == 0 && level == 0 ? m->e0s : ch;  // E[0] is stored with a pixel stride of 8 floats
    TZ_HIP(ctx, hipMemcpy2DAsync(out, ch * 4, src, sch * 4, ch * 4, npx, hipMemcpyDefault, ctx->stream));
    TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TZ_OK;
}
