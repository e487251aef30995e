// Convolution kernels of the PredNet predictor on MI355X (device code; included by tz_prednet.hip,
// which holds the host-side model driver).  All of them are implicit GEMMs over 16x16-pixel
// tiles on the fp32 matrix cores (v_mfma_f32_16x16x4_f32) with the ConvLSTM / error-unit /
// max-pool epilogues fused, and all walk the SAME fmaf chains (arithmetic contract TZ-PA1, stated
// in tz_prednet.hip), so they are interchangeable bit for bit:
//   k_conv16     sources with a multiple of 16 channels: all staging by LDS-DMA (levels >= 1)
//   k_conv16b    the level-0 convolutions as block steps (6-channel source stored 8 wide)
//   k_convlat    the k_conv16 convolutions for grids that cannot fill the chip: one accumulator tile per wave
//   k_conv_small the 3 -> 3 prediction convolution as a direct VALU kernel
//   k_conv3x3    the general kernel (any channel count, register-staged): prepare-time work,
//                other model shapes, and the tz_set_conv_impl(0) cross-check
//   k_err0       level-0 error unit
//
// Tiling (all MFMA kernels): workgroup = 8 waves = 16x16 output pixels (256 GEMM rows) x NT*16
// output columns; each wave owns 32 rows x NT*16 columns = 2 x NT MFMA tiles (acc in registers).
#pragma once
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
    const float* Wlat;  // per-wave fragment image for k_convlat: [slot][column block][4 column tiles][lane][k-step], or null
    const float* Wwino; // TZ-PA2 stage image for k_wino: [column block][stage][16 weight sets][lane][4 column tiles], or null
    int ipw;            // k_wino: column blocks a workgroup does one after the other (divides ncb)
    int wino_stride;    // k_wino: stages per column block in Wwino when the launch walks only PART of them (0 = all of them)
    int wino_first;     // ... and the first stage of that part (a launch over the same-resolution phase: 0; over the upsampled one: S1)
    const float* zero;  // >= 16 bytes of zeros: LDS-DMA source of out-of-image patch pixels
    int ncols;
    const float* bias;  // [ncols]
    const float* init;  // [H*W][ncols] accumulator start (G0), or null -> bias
    long long init_nstride;  // k_convlat: elements between batch items of init (0 = the same for every item)
    int slot0;          // k_convlat: first weight slot of this launch (a launch over the tail of the source list)
    const float* initf; // the same values in accumulator-fragment order (k_to_fragments), or null
    const float* auxf;  // EPI_LSTM: `aux` in fragment order, or null
    int Cout;
    const float* aux;   // LSTM: previous cell state [H*W][R] or null; POOL_ERR: Ahat(t0) of level l+1
    float* out0;
    long long out0_nstride;
    float* out1;        // LSTM: cell state out (or null)
    long long out1_nstride;
    const int* out_idx; // optional: frame slot of batch item n in out0
    int clip1;          // EPI_RELU: min(.,1)  (prednet.py:270)
    int R;              // EPI_LSTM_PACKED: channels per gate
    // k_conv_small: the prediction is also the input of the NEXT predictor step of its window, whose first act is the
    // level-0 error unit e_0 = [relu(Ahat0(t0) - x), relu(x - Ahat0(t0))] (k_err0).  With e0_out set the epilogue writes
    // that too -- into slot e0_slot[n] of the level-0 error maps (8 floats per pixel), unless the slot is negative.
    float* e0_out;
    const float* e0_ahat;   // Ahat_0 at t0 [H*W][3]
    const int* e0_slot;
    long long e0_nstride;
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
    // The two M-tiles of a wave (mt = 0, 1) sit EIGHT patch rows apart in every map: 8 x 18 pixels x 4
    // floats = 576 = 9 x 64 dwords, so k_conv16 reaches both tiles and all four k-step planes (21 x 64 dwords
    // apart) of a tap from ONE address register through the 64-dword-granular immediates of
    // ds_read2st64_b32 -- no vector address arithmetic per k-step (VALU instructions take matrix-pipe time).
    int w = m >> 5, mt = (m >> 4) & 1, r16 = m & 15;
    if (MAP == MAP_POOL) {
        // M-tile = 8 rows x 2 columns = four stacked 2x2 windows; wave w = columns 2w, 2w+1, rows 8 mt ..
        py = 8 * mt + 2 * (r16 >> 2) + ((r16 & 3) >> 1);
        px = 2 * w + (r16 & 1);
    } else if (MAP == MAP_PARITY) {
        // M-tile = two rows x 8 columns of the 8x8 grid of one parity class pc = w >> 1 (wave-uniform: the
        // collapsed weights of an upsampled source depend on it); class rows 2 sub, 2 sub + 1 with
        // sub = (w & 1) + 2 mt
        int pc = w >> 1, sub = (w & 1) + 2 * mt;
        py = 2 * (2 * sub + (r16 >> 3)) + (pc >> 1);
        px = 2 * (r16 & 7) + (pc & 1);
    } else {
        py = w + 8 * mt;
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
                if (!a.auxf) cp[mt][r] = a.aux ? a.aux[pix[mt][r] * R + ch] : 0.0f;
            }
        if (a.auxf) {
            const int tile = (ty0 >> 4) * a.tiles_x + (tx0 >> 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 t = *(const f32x4*)(a.auxf + ((((long long)tile * a.ncb + cb) * 8 + wv) * MT + mt) * 256 + lane * 4);
                cp[mt][0] = t[0]; cp[mt][1] = t[1]; cp[mt][2] = t[2]; cp[mt][3] = t[3];
            }
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

// ------------------------------------------------------------------------------------------
// k_conv3x3: the general kernel.  For every block of 16 input channels the 18x18 halo patch is
// staged in LDS ONCE and all 9 taps read their shifted A fragments from it (the first version
// re-staged A per tap and was bound by L2/Infinity-Cache traffic: 53 % L2 hit rate,
// profiles/r01/conv_v0); the per-tap 16 x (NT*16) weight chunk is double-buffered in LDS, one
// barrier per tap; patch and weights go global -> registers -> LDS, the next patch prefetched
// into registers during the 9 taps.  LDS row strides (18 / NT*16[+16] floats) keep fragment reads
// conflict free.  33-44 KB LDS, 512 threads => 3 workgroups (24 waves) per CU.
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
// 8-piece patches [0,8), [8,16) and weight buffers 16 + 16*buf; the switch between the two
// layouts happens once per workgroup behind a barrier, with an un-overlapped first load.
static constexpr int P16_PIECES = 21;                   // 1 KB pieces of an 18x18 patch (324 px -> 20.25)
static constexpr int U16_PIECES = 7;                    // ... of a 10x10 half-resolution patch (k_conv16b)
static constexpr int NP16 = P16_PIECES * 16, NPU16 = U16_PIECES * 16;  // slots per quad plane (336 / 112)
// k_conv16 stores its 10x10 half-resolution patch with a row stride of 12 slots: the two M-tiles of a wave are
// 4 half-resolution rows apart = 4 x 12 x 4 = 3 x 64 dwords (see row_to_patch); 10 x 12 = 120 -> 128 slots per plane
static constexpr int LWS16 = 12, NPU16S = 128, U16S_PIECES = NPU16S * 4 / 64;
static constexpr int C16_LDS_PIECES = 2 * P16_PIECES + 8;   // >= 2 * U16S_PIECES + 32

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS-DMA of one contiguous kilobyte: lane i's 16 bytes come from ubase + 16 i (ubase wave-uniform, voff = 16 x lane
// kept in a register by the caller).  `global_load_lds_dwordx4 v_off, s[base]`: the address needs no VALU
// instruction (the generic form costs a 64-bit vector add per issue, and VALU instructions take matrix-pipe time).
// Opaque to hipcc like glds16_opaque below; k_conv16 waits with explicit vmcnt(0) in front of its barriers.
__device__ __forceinline__ void glds16_lanes(const float* ubase, unsigned voff, const float* l) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)l);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(ubase), "s"(la) : "memory");
}

// ... and the gather form of the same: lane i fetches 16 bytes from ubase + voff_i if bit i of `mask` is set and
// does nothing otherwise (its LDS slot keeps what it holds: k_conv16 zeroes the slots of out-of-image pixels once
// per workgroup).  The block saves the wave's exec mask in a scalar pair of its own and puts it back (until round 3 it
// ended with `exec = -1`, i.e. assumed a fully enabled wave around it).
// m0 and exec cannot be DECLARED: hipcc (roc-7.2.0) answers a "m0" / "exec" clobber with "inline asm clobber list contains
// reserved registers ... may not be preserved across the asm statement, and clobbering them may lead to undefined
// behaviour" and ignores it.  What makes the blocks safe instead: the compiler treats m0 as reserved and loads it itself
// in front of every instruction of its own that reads it (LDS-DMA builtins, s_movrel, ...), and exec is back to its value
// before the block ends.
__device__ __forceinline__ void glds16_gather(const float* ubase, unsigned voff, unsigned long long mask, const float* l) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)l);
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "s"(ubase), "s"(la), "s"(mask) : "memory");
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
    // The geometry of a lane's items (which pixel, inside the image or not) is the same for every block of the
    // workgroup, so it is worked out ONCE: poff[j] = byte offset of item (wv + 8j) * 64 + lane inside a source
    // image (pixel offset x pixel stride + quad), pmask[j] = the lanes whose pixel exists.  Per block a patch DMA
    // is then scalar work only (exec = mask, scalar base of the 16-channel block): VALU instructions are not
    // free beside the K loops of the other waves -- each one takes 4 cycles of the SIMD that its matrix pipe
    // does not get (scripts/microbench/coexec.hip: 8 v_fma beside every MFMA double the MFMA stream's time; the
    // per-block geometry was ~30 integer instructions per item: 0.845 -> 0.861 of the MFMA peak without it).
    // Lanes outside the image fetch nothing: their LDS slots are zeroed once per phase (below) and never
    // written.  The same-resolution sources of a launch share one pixel stride (E_0's is k_conv16b's business).
    constexpr int NPJ = (P16_PIECES + 7) / 8;
    unsigned poff[NPJ], poff_lo = 0;
    unsigned long long pmask[NPJ], pmask_lo = 0;
    {
        const int ps = a.src[0].pstride;
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const int piece = wv + 8 * j;
            const int i = piece * 64 + lane, q = i / NP16, slot = i - q * NP16;
            const int y = slot / PW, xs = slot - y * PW;
            const int x = MAP == MAP_PARITY ? (xs < PW / 2 ? 2 * xs : 2 * (xs - PW / 2) + 1) : xs;
            const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
            const bool ok = piece < P16_PIECES && slot < PPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            poff[j] = ok ? 4u * (unsigned)((yy * a.W + xx) * ps + 4 * q) : 0u;
            pmask[j] = __ballot(ok);
        }
        if (UPS) {
            const ConvSrc& su = a.src[a.nsrc - 1];
            const int i = wv * 64 + lane, q = i / NPU16S, slot = i - q * NPU16S;
            const int Y = slot / LWS16, X = slot - Y * LWS16;
            const int ly = (ty0 >> 1) - 1 + Y, lx = (tx0 >> 1) - 1 + X;
            const bool ok = wv < U16S_PIECES && Y < LW && X < LW && ly >= 0 && ly < (a.H >> 1) && lx >= 0 && lx < (a.W >> 1);
            poff_lo = ok ? 4u * (unsigned)((ly * (a.W >> 1) + lx) * su.pstride + 4 * q) : 0u;
            pmask_lo = __ballot(ok);
        }
    }
    // zeroes the first `pieces` KB of the workgroup's LDS (the patch buffers of the coming phase)
    auto zero_lds = [&](int pieces) {
        for (int i = tid; i < pieces * 64; i += NTHR) *(f32x4*)(smem + i * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto issue_patch = [&](int blk, int piece0) {
        const bool s1 = blk >= nb0;
        const ConvSrc& s = s1 ? a.src[1] : a.src[0];
        const int c0 = (s1 ? blk - nb0 : blk) * 16;
        const float* base = s.p + (long long)n * s.nstride + c0;
        float* dst = smem + piece0 * 256;
        if (UPS && blk >= nbe) {
            if (wv < U16S_PIECES) glds16_gather(base, poff_lo, pmask_lo, dst + wv * 256);
            return;
        }
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const int piece = wv + 8 * j;
            if (piece < P16_PIECES) glds16_gather(base, poff[j], pmask[j], dst + piece * 256);
        }
    };
    // weights of one step into the pieces starting at piece0.  Same-resolution step: the 4 k-steps
    // of (block, tap), one piece each from the waves of half `half` of the workgroup.  Upsampled
    // step: the 4 k-steps of one collapsed tap for each of the 4 parity classes, piece index
    // class * 4 + k-step; wave w fetches k-steps 2(w&1), 2(w&1)+1 of class w>>1.
    // (scalar base + lane offset: no address arithmetic in the vector ALU, see glds16_lanes)
    const unsigned lane16 = lane * 16;
    auto issue_w = [&](int slot, bool up, int piece0, int half) {
        if (UPS && up) {
            const float* src = a.Wimg + (((long long)(slot + (wv >> 1)) * a.ncb + cb) * 4 + 2 * (wv & 1)) * 256;
            float* dst = smem + (piece0 + 2 * wv) * 256;
            glds16_lanes(src, lane16, dst);
            glds16_lanes(src + 256, lane16, dst + 256);
        } else if ((wv >> 2) == half) {
            glds16_lanes(a.Wimg + (((long long)slot * a.ncb + cb) * 4 + (wv & 3)) * 256, lane16, smem + (piece0 + (wv & 3)) * 256);
        }
    };

    // ---- accumulators (as in k_conv3x3).  The kind of initial value is tested OUTSIDE the tile loops: inside,
    // hipcc keeps one branch diamond per tile and waits for each tile's load at its join (vmcnt(0) x 8,
    // eight memory latencies in a row at the head of every workgroup)
    f32x4 acc[MT][NT];
    {
        const int col0 = cb * (NT * 16) + (lane & 15);
        if (a.initf) {
            // one coalesced 16-byte load per tile (1 KB per wave-instruction) instead of four
            // 64-byte segments per element: the prologue's G0 read was 20+ us under load
            const float* f = a.initf + ((((long long)tile * a.ncb + cb) * 8 + wv) * MT * NT) * 256 + lane * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = *(const f32x4*)(f + (mt * NT + nt) * 256);
        } else if (a.init) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (clamped address instead of a bounds branch: rows outside the image are never stored)
                    int py, px;
                    row_to_patch<MAP>(wv * 32 + mt * 16 + g * 4 + r, py, px);
                    const int y = ty0 + py, x = tx0 + px;
                    const long long pix = (y < a.H && x < a.W) ? (long long)y * a.W + x : 0;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = a.init[pix * a.ncols + col0 + nt * 16];
                }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float b = a.bias[col0 + nt * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = (f32x4){b, b, b, b};
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
        abase_lo[mt] = 4 * (((py >> 1) + (py & 1)) * LWS16 + (px >> 1) + (px & 1)) + g;
    }
    const int wcls = MAP == MAP_PARITY ? (wv >> 1) : 0;  // parity class (py&1, px&1) of this wave's rows

    // The K loop is two loop nests in sequence -- the blocks of the same-resolution sources, then
    // the blocks of the upsampled source -- each with ONE MFMA body: with both kinds of step in one
    // loop body hipcc moves the accumulators between register sets per branch and spills them.
    int slot0 = 0;
    auto run_phase = [&](auto upc, int b0, int b1) {
        constexpr bool UP = decltype(upc)::value;
        constexpr int nsteps = UP ? 4 : 9, bslots = UP ? 16 : 9, WP = UP ? 16 : 4;
        constexpr int pA0 = 0, pA1 = UP ? U16S_PIECES : P16_PIECES;  // patch buffers
        constexpr int wA = 2 * pA1;                                  // weight buffers
        if (b0 >= b1) return;
        // the slots of pixels outside the image stay zero for the whole phase (glds16_gather skips their lanes)
        zero_lds(2 * pA1);
        wg_barrier();
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
                    toff = 4 * ((st >> 1) * LWS16 + (st & 1));
                } else {
                    const int dy = st / 3, dx = st - 3 * dy;
                    const int xo = MAP == MAP_PARITY ? (dx == 1 ? ((wcls & 1) ? 1 - PW / 2 : PW / 2) : (dx >> 1)) : dx;
                    toff = 4 * (dy * PW + xo);
                }
                constexpr int KOFF = 4 * (UP ? NPU16S : NP16);  // floats between the quad planes
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
// k_convlat: the same convolutions (same fmaf chains, same epilogue arithmetic) for launches that
// are too small to fill the chip -- 64x64 / 128x160 frames, where a level is 1-6 of k_conv16's
// 16x16-pixel tiles.  There the launch costs what ONE workgroup costs, and a k_conv16 wave walks the
// K loop with 8 MFMAs per k-step plus a barrier per 32 MFMAs with nobody to overlap with: 234 us for
// the top-level gates whatever the frame size.  The chain order forbids splitting K, so the only
// way to shorten the critical path is to give every wave ONE accumulator tile and to put the tiles
// on as many SIMDs as there are:
//   workgroup = 4 waves = 16 pixels x one column block; wave w owns column tile w (for the gate
//   convolutions: the i | f | g | o columns of 16 channels, so the LSTM update still sees its four
//   gates inside one workgroup, through 4 KB of LDS);
//   16 pixels = a 4x4 block (MAP_LINEAR / MAP_POOL: the 4 accumulator registers of a lane are one
//   2x2 pooling window) or one parity class of an 8x8 region (MAP_PARITY: the collapsed weights of
//   an upsampled source depend on the parity class);
//   B fragments never touch LDS: the weights are packed a third time, per (slot, column block,
//   column tile) as [lane][k-step], so a lane's four k-steps of a slot are ONE 16-byte global load,
//   prefetched one block (9 or 4 slots) ahead in registers (nothing is shared between waves);
//   A fragments come from a halo patch in LDS (6x6 / 10x10 pixels x 16 channels per block, quad-planar),
//   staged one block ahead by LDS-DMA, one barrier per 16-channel block.
// Per slot a wave issues 4 dependent MFMAs (the chain), i.e. the K loop runs at the latency of the
// matrix pipe: 32 cycles per MFMA measured (scripts/microbench/mfma_chain.hip), 54 ns per slot at best.
template <int N>
__device__ __forceinline__ void wait_vmn() {
    static_assert(N == 0 || N == 4 || N == 5 || N == 6 || N == 7 || N == 9 || N == 13 || N == 14 || N == 15, "immediate of s_waitcnt");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}

// 16-byte LDS read the compiler does not see as one: a plain load from the ring would make it wait
// for EVERY outstanding LDS-DMA of the same array (vmcnt(0)) in front of each read, i.e. drain the
// prefetch ring every slot; the explicit vmcnt above is the exact condition.
__device__ __forceinline__ f32x4 lds_read16_opaque(const float* p) {   // issue only: lds_wait(v) before v is used
    const unsigned addr = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait(f32x4& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }
// LDS-DMA hipcc does not see as one: global_load_lds is a FLAT instruction that touches both memories, and
// with one outstanding every later wait for an ordinary load becomes vmcnt(0) (no in-order assumption),
// which would drain k_convlat's register ring at each block.  Unseen, it only makes hipcc's own
// vmcnt(N) wait for more than it has to (N counts the loads it knows; ours are extra).
__device__ __forceinline__ void glds16_opaque(const float* g, const float* l) {   // l: wave-uniform
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)l);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(la) : "memory");   // (m0 is reserved: hipcc sets it in front of each of its own uses)
}
// two floats O0 and O1 x 64 dwords behind a byte address, same reason
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_read2st64_opaque(unsigned addr) {
    static_assert(O0 >= 0 && O1 < 256, "8-bit offsets");
    f32x2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait(f32x2& a, f32x2& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void lds_wait(f32x2& a, f32x2& b, f32x2& c, f32x2& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}

template <int N>
__device__ __forceinline__ void wait_w(f32x4& w) {   // vmcnt(N), and `w` is not read before it
    static_assert(N == 3 || N == 8, "immediate of s_waitcnt");
    if (N == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(w) : : "memory");
    else asm volatile("s_waitcnt vmcnt(8)" : "+v"(w) : : "memory");
}

template <int MAP>
__device__ __forceinline__ void lat_row_to_pixel(int r, int pc, int& py, int& px) {
    if (MAP == MAP_POOL) {  // rows 4g..4g+3 = the 2x2 window g of the 4x4 block
        const int g = r >> 2, q = r & 3;
        py = 2 * (g >> 1) + (q >> 1);
        px = 2 * (g & 1) + (q & 1);
    } else if (MAP == MAP_PARITY) {  // parity class pc of an 8x8 region
        py = 2 * (r >> 2) + (pc >> 1);
        px = 2 * (r & 3) + (pc & 1);
    } else {
        py = r >> 2;
        px = r & 3;
    }
}

// MTL = accumulator tiles per wave: 1 = 16 pixels per workgroup (the latency-bound case: as many SIMDs
// as possible), 2 = 32 pixels (two tiles side by side, two independent MFMA chains interleaved: for
// grids of several rounds, where the matrix pipe rather than the chain latency is the limit and
// half as many workgroups stream the weights).
template <int EPI, bool UPS, int MTL>
__device__ __forceinline__ void convlat_body(const ConvArgs& a, int bid) {
    constexpr int MAP = EPI == EPI_POOL_ERR ? MAP_POOL : (UPS ? MAP_PARITY : MAP_LINEAR);
    constexpr int TS = MAP == MAP_PARITY ? 8 : 4;      // footprint of one accumulator tile in pixels
    constexpr int TSX = TS * MTL;                      // footprint of the workgroup in x
    constexpr int PWL = TSX + 2, PHL = TS + 2, PPL = PWL * PHL;   // same-resolution halo patch
    constexpr int LWL = TSX / 2 + 2, LPL = 6 * LWL;    // half-resolution patch of an upsampled source (parity tiles)
    // patch image: QUAD-PLANAR like k_conv16's -- the 16-byte item (pixel slot p, channel quad q) sits at
    // item index q * NP + p, so that the 16 rows x 4 elements of an A fragment spread over the banks
    // (the memory layout, 64 bytes per pixel, puts them on 8 banks: 8-way conflicts, 256 LDS cycles per
    // slot and workgroup, which is what the first version of this kernel ran at); a parity tile stores
    // its columns evens first so that the pixels of one class are neighbours
    constexpr int NPS = (PPL + 15) / 16 * 16;          // slots per quad plane, same resolution
    constexpr int NPU = (LPL + 15) / 16 * 16;          // ... half resolution
    constexpr int PCS = NPS * 4 / 64, PCU = NPU * 4 / 64;   // 1 KB pieces (= DMA wave-instructions) per patch
    constexpr int NPI = (PCS + 3) / 4;                 // patch DMA instructions per wave
    static_assert(PCU <= PCS, "the half-resolution patch fits the same buffers");
    // LDS: two patch buffers
    constexpr int SE = EPI == EPI_LSTM ? 4 * 16 * MTL * 17 : 0;          // gate exchange of the LSTM epilogue
    constexpr int PBUF = PCS * 256 > (SE + 1) / 2 ? PCS * 256 : ((SE + 1) / 2 + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float sP[2][PBUF];
    float* const sE = &sP[0][0];   // (re-uses the patch buffers: the K loop ends with a barrier)

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTW = a.ncols / (16 * a.ncb);            // column tiles per block: 3 or 4
    const int ltx = (a.W + TSX - 1) / TSX, lty = (a.H + TS - 1) / TS;
    const int ntiles = ltx * lty * (MAP == MAP_PARITY ? 4 : 1);
    const int cb = bid % a.ncb;
    bid /= a.ncb;
    const int tile = bid % ntiles, n = bid / ntiles;
    const int pc = MAP == MAP_PARITY ? (tile & 3) : 0;
    const int reg = MAP == MAP_PARITY ? (tile >> 2) : tile;
    const int ty0 = (reg / ltx) * TS, tx0 = (reg % ltx) * TSX;
    const int g = lane >> 4;
    const bool active = wv < NTW;

    const int nb0 = a.src[0].cpt;
    const int nblk = nb0 + (a.nsrc > 1 ? a.src[1].cpt : 0);
    const bool up0 = UPS && a.src[0].up != 0;
    const int nbe = up0 ? 0 : (UPS && a.nsrc > 1 && a.src[1].up ? nb0 : nblk);  // same-resolution blocks

    // ---- LDS-DMA issue: item i = piece * 64 + lane = quad plane i / NP, slot i % NP.  As in k_conv16 the geometry
    // of a lane's items is worked out once (byte offset inside a source image + the mask of the lanes whose
    // pixel exists), a patch DMA per block is scalar work, and the slots of pixels outside the image are zeroed
    // once per phase: here a wave has its SIMD to itself, so every VALU instruction is on the critical path.
    unsigned poff[NPI], poffu[NPI];
    unsigned long long pmask[NPI], pmasku[NPI];
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
        const int piece = k * 4 + wv;
        {
            const int i = piece * 64 + lane, q = i / NPS, slot = i - q * NPS;
            const int Y = slot / PWL, xs = slot - Y * PWL;
            const int X = MAP == MAP_PARITY ? (xs < PWL / 2 ? 2 * xs : 2 * (xs - PWL / 2) + 1) : xs;
            const int yy = ty0 - 1 + Y, xx = tx0 - 1 + X;
            const bool ok = piece < PCS && slot < PPL && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            poff[k] = ok ? 4u * (unsigned)((yy * a.W + xx) * a.src[0].pstride + 4 * q) : 0u;
            pmask[k] = __ballot(ok);
        }
        poffu[k] = 0;
        pmasku[k] = 0;
        if (UPS) {
            const ConvSrc& su = a.src[a.nsrc - 1];
            const int i = piece * 64 + lane, q = i / NPU, slot = i - q * NPU;
            const int Y = slot / LWL, X = slot - Y * LWL;
            const int ly = (ty0 >> 1) - 1 + Y, lx = (tx0 >> 1) - 1 + X;
            const bool ok = piece < PCU && slot < LPL && ly >= 0 && ly < (a.H >> 1) && lx >= 0 && lx < (a.W >> 1);
            poffu[k] = ok ? 4u * (unsigned)((ly * (a.W >> 1) + lx) * su.pstride + 4 * q) : 0u;
            pmasku[k] = __ballot(ok);
        }
    }
    auto issue_patch = [&](int blk, int buf, bool up) {
        const bool s1 = blk >= nb0;
        const ConvSrc& s = s1 ? a.src[1] : a.src[0];
        const float* base = s.p + (long long)n * s.nstride + (s1 ? blk - nb0 : blk) * 16;
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
            const int piece = k * 4 + wv;
            if (piece >= (up ? PCU : PCS)) continue;               // wave-uniform
            glds16_gather(base, up ? poffu[k] : poff[k], up ? pmasku[k] : pmask[k], sP[buf] + piece * 256);
        }
    };
    auto zero_patches = [&]() {
        for (int i = tid; i < 2 * PBUF / 4; i += 256) *(f32x4*)(&sP[0][0] + i * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // ---- accumulators: rows 4g..4g+3 of tile m (tile m sits m * TS pixels to the right), column lane&15
    // of this wave's column tile
    const int col = cb * (NTW * 16) + wv * 16 + (lane & 15);
    f32x4 acc[MTL];
#pragma unroll
    for (int m = 0; m < MTL; ++m) {
        acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (active) {
            if (a.init) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int py, px;
                    lat_row_to_pixel<MAP>(4 * g + r, pc, py, px);
                    const int y = ty0 + py, x = tx0 + m * TS + px;
                    const long long pix = (y < a.H && x < a.W) ? (long long)y * a.W + x : 0;
                    acc[m][r] = a.init[(long long)n * a.init_nstride + pix * a.ncols + col];
                }
            } else {
                const float b = a.bias[col];
                acc[m] = (f32x4){b, b, b, b};
            }
        }
    }
    // LDS float offsets of this lane's A row (GEMM row lane&15, element g of a quad) for tap (0,0)
    int abase[MTL], abase_lo[MTL];
#pragma unroll
    for (int m = 0; m < MTL; ++m) {
        int py, px;
        lat_row_to_pixel<MAP>(lane & 15, pc, py, px);
        px += m * TS;
        // tap (0,0): patch row py, column px (a parity tile: column slot (px >> 1) + (PWL / 2) (px & 1))
        abase[m] = (py * PWL + (MAP == MAP_PARITY ? (px >> 1) + (PWL / 2) * (px & 1) : px)) * 4 + g;
        abase_lo[m] = (((py >> 1) + (py & 1)) * LWL + (px >> 1) + (px & 1)) * 4 + g;
    }
    const long long wstride = (long long)a.ncb * 1024;                                 // floats between slots of Wlat
    const float* wbase = a.Wlat + ((long long)cb * 4 + wv) * 256;                      // wave-uniform
    const unsigned wlane = lane * 16;                                                  // bytes
    // scalar base + 32-bit lane offset (global_load ... v_off, s[base]): half the address registers of the
    // 64-bit form -- a kilobyte of VMEM traffic costs the SIMD's matrix pipe 20-30 cycles, less in this form
    // (scripts/microbench/lat_occ.hip).  Written as asm: hipcc only selects this form when it sees the
    // zero-extension of the offset next to the load, i.e. at the price of a v_mov per load.  It then does not
    // count these loads either: wait_w<N> in front of a ring entry's first use (N = loads issued after its own).
    auto ldw = [&](const float* u) {
        f32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(wlane), "s"(u) : "memory");
        return v;
    };
    // The K loop: per slot (16 channels x one tap) the wave takes its four B k-steps from a REGISTER ring
    // (one global_load_dwordx4 per slot, issued one block = SPB slots ahead into the entry the slot just
    // consumed) and four A values per tile from the patch, and runs 4
    // dependent MFMAs per tile.  The patch travels by LDS-DMA, whose waits are explicit: memory operations
    // of a wave complete in order, so a patch issued at the start of a block, in front of that block's SPB
    // weight loads, has landed at vmcnt(SPB).  One barrier per 16-channel block (the patch is shared by
    // the 4 waves).  Why registers: scripts/microbench/lat_slot2.hip -- with ONE wave per SIMD every memory
    // instruction between the MFMAs of the chain costs the wave 15-70 cycles that nothing overlaps; a ring
    // in LDS (DMA + ds_read_b128 per slot, the first version) ran at 142 ns per slot, this one at ~90.
    int slot0 = a.slot0;
    auto run_phase = [&](auto upc, int b0, int b1) {
        constexpr bool UP = decltype(upc)::value;
        constexpr int SPB = UP ? 4 : 9;                   // slots this wave uses per block
        if (b0 >= b1) return;
        // weights of step t: same resolution slot0 + t; upsampled slot0 + 16 (t >> 2) + 4 (t & 3) + class = slot0 + pc + 4 t
        const long long wstep = (UP ? 4 : 1) * wstride;
        const float* wblk = wbase + (long long)(slot0 + (UP ? pc : 0)) * wstride;   // first slot of the current block
        __syncthreads();                                  // every wave is done with the previous phase's patches
        zero_patches();                                   // slots of pixels outside the image: zero for the whole phase
        __syncthreads();
        issue_patch(b0, 0, UP);
        f32x4 wr[SPB];
        if (active) {
#pragma unroll
            for (int st = 0; st < SPB; ++st) wr[st] = ldw(wblk + st * wstep);
            wait_vmn<SPB>();
        } else {  // a wave without a column tile (NT = 3) has issued nothing but its part of the patches
            wait_vmn<0>();
        }
        __syncthreads();
        int buf = 0;
        // slot offset of a tap.  Parity tile: its columns are stored evens first, so one step in x is
        // +PWL/2 from an even column and 1 - PWL/2 from an odd one, two steps are +1 (the class is uniform)
        auto t_off = [&](int st) {
            if (UP) return ((st >> 1) * LWL + (st & 1)) * 4;
            const int dy = st / 3, dx = st % 3;
            const int xo = MAP == MAP_PARITY ? (dx == 1 ? ((pc & 1) ? 1 - PWL / 2 : PWL / 2) : (dx >> 1)) : dx;
            return (dy * PWL + xo) * 4;
        };
        constexpr int KOFF = 4 * (UP ? NPU : NPS);         // floats between the quad planes
        // One block = SPB slots.  The MFMAs of a slot form one dependent chain per tile (32 cycles per link);
        // the patch reads of the NEXT slot sit behind the first MFMA and the reload of the ring entry
        // behind the last (sched barriers keep them there).
        auto block = [&](const float* wnext) {
            // the patch is read by instructions hipcc does not see as LDS reads: in front of one it knows
            // it waits for EVERY outstanding LDS-DMA (vmcnt(0): the next block's patch and with it the
            // whole weight ring); the explicit vmcnt(SPB) in front of the barrier is the exact condition
            const unsigned pa = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)sP[buf];
            static_assert(KOFF % 64 == 0 && 3 * KOFF / 64 < 256, "the four k-steps of a lane are two ds_read2st64_b32");
            auto read_a = [&](int st, f32x2 (&f)[MTL][2]) {
#pragma unroll
                for (int m = 0; m < MTL; ++m) {
                    const unsigned ad = pa + 4 * ((UP ? abase_lo[m] : abase[m]) + t_off(st));
                    f[m][0] = lds_read2st64_opaque<0, KOFF / 64>(ad);
                    f[m][1] = lds_read2st64_opaque<2 * KOFF / 64, 3 * KOFF / 64>(ad);
                }
            };
            auto wait_a = [&](f32x2 (&f)[MTL][2]) {
                if constexpr (MTL == 1) lds_wait(f[0][0], f[0][1]);
                else lds_wait(f[0][0], f[0][1], f[1][0], f[1][1]);
            };
            f32x2 fa[MTL][2];
            read_a(0, fa);
            wait_a(fa);
#pragma unroll
            for (int st = 0; st < SPB; ++st) {
                const bool more = st + 1 < SPB;
                wait_w<SPB - 1>(wr[st]);   // the SPB - 1 younger ring loads (and this block's patch DMAs) may still fly
                const f32x4 w = wr[st];
                f32x2 fan[MTL][2];
#pragma unroll
                for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][0][0], w[0], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) read_a(st + 1, fan);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][0][1], w[1], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][1][0], w[2], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MTL; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[m][1][1], w[3], acc[m], 0, 0, 0);
                wr[st] = ldw(wnext + st * wstep);    // the same slot of the next block (of this one again at the end: never used)
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    wait_a(fan);
#pragma unroll
                    for (int m = 0; m < MTL; ++m) {
                        fa[m][0] = fan[m][0];
                        fa[m][1] = fan[m][1];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll 1
        for (int blk = b0; blk < b1; ++blk) {
            if (blk > b0) {
                if (active) wait_vmn<SPB>();   // the SPB reloads of the previous block were issued behind this block's patch
                else wait_vmn<0>();
                __syncthreads();
                buf ^= 1;
            }
            const bool last = blk + 1 == b1;
            if (!last) issue_patch(blk + 1, buf ^ 1, UP);
            const float* wnext = wblk + (last ? 0 : SPB * wstep);
            if (active) block(wnext);
            wblk = wnext;
        }
        // The last block reloaded its ring entries once more (unused).  hipcc does not know that these asm loads
        // are still in flight: the registers must stay the ring's until they have landed, or the late data
        // lands in whatever hipcc put there next.
        if (active) {
            if constexpr (SPB == 9)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]),
                             "+v"(wr[7]), "+v"(wr[8]) : : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]) : : "memory");
        }
        slot0 += (b1 - b0) * (UP ? 16 : 9);
    };
    run_phase(std::false_type{}, 0, nbe);
    if (UPS) run_phase(std::true_type{}, nbe, nblk);
    __syncthreads();

    // ---- epilogues: the arithmetic of conv_epilogue, re-distributed
    if (EPI == EPI_LSTM) {
        // wave w holds gate w (i | f | g | o) of channels cb*16 .. cb*16+15 for the 16 * MTL pixels
        constexpr int NR = 16 * MTL;
        if (active) {
#pragma unroll
            for (int m = 0; m < MTL; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) sE[(wv * NR + m * 16 + 4 * g + r) * 17 + (lane & 15)] = acc[m][r];
        }
        __syncthreads();
        const int j = tid & 15, ch = cb * 16 + j, R = a.Cout;
#pragma unroll
        for (int m = 0; m < MTL; ++m) {
            const int row = tid >> 4;
            int py, px;
            lat_row_to_pixel<MAP>(row, pc, py, px);
            const int y = ty0 + py, x = tx0 + m * TS + px;
            if (y < a.H && x < a.W) {
                const long long pix = (long long)y * a.W + x;
                const int rr_ = m * 16 + row;
                const float cp = a.aux ? a.aux[pix * R + ch] : 0.0f;
                const float gi = tz_hard_sigmoid(sE[(0 * NR + rr_) * 17 + j]);
                const float gf = tz_hard_sigmoid(sE[(1 * NR + rr_) * 17 + j]);
                const float gg = tz_tanh(sE[(2 * NR + rr_) * 17 + j]);
                const float go = tz_hard_sigmoid(sE[(3 * NR + rr_) * 17 + j]);
                const float t1 = gf * cp;
                const float t2 = gi * gg;
                const float c = t1 + t2;
                const float rr = go * tz_tanh(c);
                a.out0[(long long)n * a.out0_nstride + pix * R + ch] = rr;
                if (a.out1) a.out1[(long long)n * a.out1_nstride + pix * R + ch] = c;
            }
        }
    } else if (EPI == EPI_RAW) {
        // the accumulators as they are, [pixel][column]: the first part of a convolution whose sources
        // become available at different times (the launch over the remaining sources starts from them
        // through `init`: the same chain, cut at a source boundary)
        if (active) {
#pragma unroll
            for (int m = 0; m < MTL; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int py, px;
                    lat_row_to_pixel<MAP>(4 * g + r, pc, py, px);
                    const int y = ty0 + py, x = tx0 + m * TS + px;
                    if (y < a.H && x < a.W)
                        a.out0[(long long)n * a.out0_nstride + ((long long)y * a.W + x) * a.ncols + col] = acc[m][r];
                }
        }
    } else if (EPI == EPI_POOL_ERR) {
        // prednet.py:289-291 then 274-277 of the next level; a lane's 4 registers of tile m are window g
        const int H2 = a.H >> 1, W2 = a.W >> 1, C = a.Cout;
#pragma unroll
        for (int m = 0; m < MTL; ++m) {
            const int yp = (ty0 >> 1) + (g >> 1), xp = ((tx0 + m * TS) >> 1) + (g & 1);
            if (active && col < C && yp < H2 && xp < W2) {
                const long long pp = (long long)yp * W2 + xp;
                const float h = a.aux[pp * C + col];
                float mx = tz_relu(acc[m][0]);
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    const float t = tz_relu(acc[m][r]);
                    if (t > mx) mx = t;
                }
                const float d1 = h - mx, d2 = mx - h;
                float* o = a.out0 + (long long)n * a.out0_nstride;
                o[pp * 2 * C + col] = tz_relu(d1);
                o[pp * 2 * C + C + col] = tz_relu(d2);
            }
        }
    }
}

template <int EPI, bool UPS, int MTL>
__global__ __launch_bounds__(256) void k_convlat(const ConvArgs a) {
    convlat_body<EPI, UPS, MTL>(a, blockIdx.x);
}

// Two independent small-grid convolutions in ONE launch (workgroups [0, na) run `a`, the rest `b`): the A
// convolution of a level beside the first part of that level's gate convolution -- both read E_l only --
// so that the gate convolution's chain over E_l leaves the critical path of the top-down pass.
template <int EA, int EB>
__global__ __launch_bounds__(256) void k_convlat_pair(const ConvArgs a, const ConvArgs b, int na) {
    if ((int)blockIdx.x < na) convlat_body<EA, false, 1>(a, blockIdx.x);
    else convlat_body<EB, false, 1>(b, blockIdx.x - na);
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
    static_assert((U16_PIECES + WU + 7) / 8 <= 9 && (E8_PIECES + W8 + 7) / 8 <= 9, "wait_vm_n covers at most 9 DMAs per wave in flight");
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
                if (a.initf) {
                    acc[mt][nt] = *(const f32x4*)(a.initf + (((((long long)tile * a.ncb + cb) * 8 + wv) * MT + mt) * NT + nt) * 256 + lane * 4);
                } else if (a.init) {
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
        const long long pix = (long long)y * a.W + xq;
        float* o = a.out0 + (long long)(a.out_idx ? a.out_idx[n] : n) * a.out0_nstride + pix * COUT;
        float v[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            v[co] = tz_relu(acc[co]);
            if (a.clip1 && v[co] > 1.0f) v[co] = 1.0f;
            o[co] = v[co];
        }
        if (COUT == 3 && a.e0_out) {   // the next step's level-0 error unit, same arithmetic as k_err0
            const int slot = a.e0_slot[n];
            if (slot >= 0) {
                float d1[3], d2[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float h = a.e0_ahat[pix * 3 + c];
                    d1[c] = h - v[c];
                    d2[c] = v[c] - h;
                }
                float* e = a.e0_out + (long long)slot * a.e0_nstride + pix * 8;
                *(float4*)e = make_float4(tz_relu(d1[0]), tz_relu(d1[1]), tz_relu(d1[2]), tz_relu(d2[0]));
                *(float4*)(e + 4) = make_float4(tz_relu(d2[1]), tz_relu(d2[2]), 0.0f, 0.0f);
            }
        }
    }
}

// Prepare-time re-layout of a per-pixel constant image src[H*W][ncols] (G0 accumulator starts,
// previous cell state) into the accumulator-fragment order of the MFMA kernels:
// dst[tile][cb][wave][mt][nt][lane][r] = src[pixel of GEMM row (wave, mt, lane>>4, r)][cb*NT*16 + nt*16 + (lane&15)]
// (rows outside the image hold 0 and are never stored).
template <int MAP>
__global__ __launch_bounds__(256) void k_to_fragments(const float* __restrict__ src, int H, int W, int tiles_x, int tiles_y,
                                                      int ncols, int ncb, int NT, float* __restrict__ dst) {
    const long long total = (long long)tiles_x * tiles_y * ncb * 8 * MT * NT * 256;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i & 3), lane = (int)((i >> 2) & 63);
        long long q = i >> 8;
        const int nt = (int)(q % NT); q /= NT;
        const int mt = (int)(q % MT); q /= MT;
        const int wv = (int)(q & 7); q >>= 3;
        const int cb = (int)(q % ncb);
        const int tile = (int)(q / ncb);
        int py, px;
        row_to_patch<MAP>(wv * 32 + mt * 16 + (lane >> 4) * 4 + r, py, px);
        const int y = (tile / tiles_x) * 16 + py, x = (tile % tiles_x) * 16 + px;
        dst[i] = (y < H && x < W) ? src[((long long)y * W + x) * ncols + cb * NT * 16 + nt * 16 + (lane & 15)] : 0.0f;
    }
}

// level-0 error unit (prednet.py:274-277 with a = input frame, Ahat = Ahat_0(t0)):
// input is either a key frame (uint8, unpadded; x = float32(k)/255, compress.py:138) or a
// padded float32 frame of the prediction stack (compress.py:222).
__global__ __launch_bounds__(256) void k_err0(const uint8_t* __restrict__ frames_u8, int H, int W,
                                              const float* __restrict__ in_stack, const int* __restrict__ is_key,
                                              const int* __restrict__ in_idx, const float* __restrict__ ahat0, int Hp,
                                              int Wp, int C, int Cs, float* __restrict__ e0, int keys_only) {
    int n = blockIdx.y;
    long long npx = (long long)Hp * Wp;
    const bool key = is_key[n] != 0;
    // (keys_only: the prediction kernel of the step before has already written this item's maps from its own output -- the
    // DWP loop, where only the device knows whether the next step starts from a key frame instead)
    if (keys_only && !key) return;
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

