// PredNet predictor on MI355X: the host-side model driver (weight packing, prepare-time
// constants, per-frame launch schedule).  The convolution kernels are in tz_conv_kernels.hip.h.
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
// The MFMA k-loops walk exactly that order; out-of-image taps and channel padding
// contribute fmaf(0, w, acc) = acc.  No split-K, no atomics: results do not depend on batch
// size, grid shape or device.
//
#include <algorithm>
#include <atomic>

#include "tz_conv_kernels.hip.h"
#include "tz_wino_kernels.hip.h"

// ------------------------------------------------------------------------------- host side
struct Seg {
    int row_off, C;
    int up;  // 1: half-resolution source read through a x2 nearest upsample (collapsed taps)
};
struct PackedConv {
    float* d_W = nullptr;
    float* d_Wimg = nullptr;  // LDS image order for k_conv16 (every source a multiple of 16 channels), else null
    float* d_Wblk = nullptr;  // block-step image for k_conv16b (first source <= 8 channels at stride 8), else null
    float* d_Wlat = nullptr;  // per-wave fragment image for k_convlat (same condition as d_Wimg), else null
    float* d_Wwino = nullptr; // TZ-PA2 stage image for k_wino (pack_wino), else null
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
    float *G0f[TZ_MAX_LEVELS] = {0}, *C0f[TZ_MAX_LEVELS] = {0};  // G0 / C0 in accumulator-fragment order (k_to_fragments)
    float *E[TZ_MAX_LEVELS] = {0}, *R1[TZ_MAX_LEVELS] = {0};
    float* P[TZ_MAX_LEVELS] = {0};   // gate accumulators after the E_l part of the chain (split launches of small grids), or null
    int Pcap[TZ_MAX_LEVELS] = {0};   // ... how many batch slots P[l] holds (<= maxB: at most kPBytes per level)
    std::vector<signed char> epart_choice;   // [n]: "E-part ahead" for batches of n items: -1 not measured yet, 0 fused, 1 split
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
    return tz_poison(ctx, *p, bytes);
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
        // k_convlat's image: [slot][column block][column tile (4, the last one empty when NT = 3)][lane][k-step]:
        // the four k-steps of a lane's B fragment are one 16-byte load
        std::fill(I.begin(), I.end(), 0.0f);
        for (int sl = 0; sl < nslots; ++sl)
            for (int cb = 0; cb < ncb; ++cb)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int kk = 0; kk < 4; ++kk)
                            I[((((size_t)sl * ncb + cb) * 4 + nt) * 64 + lane) * 4 + kk] =
                                W[((size_t)sl * 16 + 4 * kk + (lane >> 4)) * ncols + cb * NT * 16 + nt * 16 + (lane & 15)];
        TZ_TRY(dmalloc(ctx, m, (void**)&pc->d_Wlat, I.size() * 4));
        TZ_HIP(ctx, hipMemcpy(pc->d_Wlat, I.data(), I.size() * 4, hipMemcpyHostToDevice));
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

// TZ-PA2 weights of a per-frame convolution for k_wino (tz_wino_kernels.hip.h; oracle/tz_oracle.c::conv3x3_wino states the
// same numbers): stage s < C0/4 = channel quad s of the same-resolution source segs[0], its 16 sets = the transformed
// weights U[i][j] = (G g G^T)[i][j] of the quad's 4 channels, position 4 i + j, evaluated in float32 rows first:
//   s = g[0][c] + g[2][c]; w[1][c] = 0.5 (s + g[1][c]); w[2][c] = 0.5 (s - g[1][c]); w[0][c] = g[0][c]; w[3][c] = g[2][c];
//   then the same along c.
// Stages behind those = the channel quads of the upsampled source segs[1]: 16 sets = (parity class (a, b), collapsed tap tp)
// at index 8 a + 2 tp + b (the order in which wave (.., ph = a) of k_wino consumes them), values = pack_conv's collapsed sums.
// Image: [column block][stage][set][lane = (channel of the quad) * 16 + column][column tile (4, zero beyond NT)]: the stages of
// a column block, and then of the next one, are one contiguous stream (a k_wino workgroup walks it with one running pointer).
static int pack_wino(tz_ctx* ctx, tz_model* m, const std::vector<Seg>& segs, const std::vector<ColSrc>& cols, int NT, PackedConv* pc) {
    if (segs.empty() || segs[0].up || segs[0].C % 16 || (NT != 3 && NT != 4) || segs.size() > 2 ||
        (segs.size() == 2 && (!segs[1].up || segs[1].C % 16)))
        return TZ_OK;   // not a k_wino convolution
    const int ncols = (int)cols.size(), ncb = ncols / (16 * NT);
    const int S1 = segs[0].C / 4, S2 = segs.size() == 2 ? segs[1].C / 4 : 0;
    std::vector<float> I((size_t)(S1 + S2) * ncb * 16 * 256, 0.0f);
    auto at = [&](int st, int cb, int set, int lane, int nt) -> float& {
        return I[((((size_t)cb * (S1 + S2) + st) * 16 + set) * 64 + lane) * 4 + nt];
    };
    for (int col = 0; col < ncols; ++col) {
        const ColSrc& cs = cols[col];
        if (cs.ch < 0) continue;
        const int cb = col / (16 * NT), nt = (col % (16 * NT)) / 16, j = col % 16;
        for (int c = 0; c < segs[0].C; ++c) {
            float g[3][3], w[4][3], U[4][4];
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) g[r][q] = cs.kernel[((size_t)(r * 3 + q) * cs.Cin + segs[0].row_off + c) * cs.Cout + cs.ch];
            for (int q = 0; q < 3; ++q) {
                const float s_ = g[0][q] + g[2][q];
                w[0][q] = g[0][q];
                w[1][q] = 0.5f * (s_ + g[1][q]);
                w[2][q] = 0.5f * (s_ - g[1][q]);
                w[3][q] = g[2][q];
            }
            for (int i = 0; i < 4; ++i) {
                const float s_ = w[i][0] + w[i][2];
                U[i][0] = w[i][0];
                U[i][1] = 0.5f * (s_ + w[i][1]);
                U[i][2] = 0.5f * (s_ - w[i][1]);
                U[i][3] = w[i][2];
            }
            for (int p = 0; p < 16; ++p) at(c / 4, cb, p, (c % 4) * 16 + j, nt) = U[p >> 2][p & 3];
        }
        for (int c = 0; c < (S2 ? segs[1].C : 0); ++c)
            for (int cls = 0; cls < 4; ++cls)
                for (int tp = 0; tp < 4; ++tp) {
                    int kys[2], kxs[2];
                    const int nky = collapse_set(cls >> 1, tp >> 1, kys), nkx = collapse_set(cls & 1, tp & 1, kxs);
                    float v = 0.0f;
                    bool first = true;
                    for (int iy = 0; iy < nky; ++iy)
                        for (int ix = 0; ix < nkx; ++ix) {
                            const float wv_ = cs.kernel[((size_t)(kys[iy] * 3 + kxs[ix]) * cs.Cin + segs[1].row_off + c) * cs.Cout + cs.ch];
                            v = first ? wv_ : v + wv_;
                            first = false;
                        }
                    at(S1 + c / 4, cb, 8 * (cls >> 1) + 2 * tp + (cls & 1), (c % 4) * 16 + j, nt) = v;
                }
    }
    TZ_TRY(dmalloc(ctx, m, (void**)&pc->d_Wwino, I.size() * 4));
    TZ_HIP(ctx, hipMemcpy(pc->d_Wwino, I.data(), I.size() * 4, hipMemcpyHostToDevice));
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

template <int NT, int EPI, bool UPS, bool NOSAME = false>
static int launch_wino_t(tz_ctx* ctx, const ConvArgs& a0, int nbatch) {
    // (per instantiation and device: the attribute belongs to the function on a device, not to a context; a process that
    // opens contexts on several devices sets it on each)
    // (contexts of several threads may launch at once: the flag is atomic and is set only AFTER the attribute is, so a thread
    // that reads it set launches with the attribute in place; two threads that both find it clear both set the attribute,
    // which is idempotent)
    static std::atomic<bool> attr_set[64];
    const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 63;
    if (dev == 63 || !attr_set[dev].load(std::memory_order_acquire)) {
        TZ_HIP(ctx, hipFuncSetAttribute((const void*)k_wino<NT, EPI, UPS, NOSAME>, hipFuncAttributeMaxDynamicSharedMemorySize, tzw::LDS_BYTES));
        attr_set[dev].store(true, std::memory_order_release);
    }
    // Column blocks per workgroup (one workgroup occupies a CU): the divisor of ncb with the shortest launch in items --
    // rounds of workgroups over the CUs x items per workgroup --, the largest one among equals: a workgroup pays its geometry
    // and its first DMA round trip once, and the column blocks of a tile march through the SAME weight stream on all CUs at
    // the same time.
    ConvArgs a = a0;
    const int tiles = a.tiles_x * a.tiles_y * nbatch;
    a.ipw = 1;
    long long best = -1;
    for (int d = 1; d <= a.ncb; ++d) {
        if (a.ncb % d) continue;
        const long long wgs = (long long)tiles * (a.ncb / d), span = (wgs + ctx->num_cus - 1) / ctx->num_cus * d;
        if (best < 0 || span <= best) best = span, a.ipw = d;
    }
    if (a0.ipw > 0 && a.ncb % a0.ipw == 0) a.ipw = a0.ipw;                         // (the caller knows better: side launches want short workgroups)
    if (ctx->wino_ipw > 0 && a.ncb % ctx->wino_ipw == 0) a.ipw = ctx->wino_ipw;   // (TEZIP_WINO_IPW: measurements)
    const int blocks = (a.ncb / a.ipw) * tiles;
    hipLaunchKernelGGL((k_wino<NT, EPI, UPS, NOSAME>), dim3(blocks), dim3(512), tzw::LDS_BYTES, ctx->stream, a);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

template <int NT, int EPI, bool UPS>
static int launch_wino_ref_t(tz_ctx* ctx, const ConvArgs& a, int nbatch) {
    const long long nthreads = (long long)((a.W + 1) / 2) * ((a.H + 1) / 2) * a.ncb * (EPI == EPI_LSTM ? 16 : 16 * NT);
    hipLaunchKernelGGL((k_wino_ref<NT, EPI, UPS>), dim3((unsigned)std::min<long long>((nthreads + 255) / 256, 65535), nbatch), dim3(256), 0,
                       ctx->stream, a);
    TZ_HIP(ctx, hipGetLastError());
    return TZ_OK;
}

static int launch_wino(tz_ctx* ctx, int NT, int epi, bool ups, const ConvArgs& a, int nbatch, bool nosame = false) {
    if (nosame) {   // the second launch of a split gate convolution (tz_model_predict_batch_dev, "E-part ahead")
        if (NT == 4 && epi == EPI_LSTM && ups && ctx->conv_impl) return launch_wino_t<4, EPI_LSTM, true, true>(ctx, a, nbatch);
        return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "no split TZ-PA2 kernel for NT=%d epilogue=%d upsampled=%d", NT, epi, (int)ups);
    }
    if (!ctx->conv_impl) {   // tz_set_conv_impl(0): the plain statement of the same chains
        if (NT == 4 && epi == EPI_LSTM) return ups ? launch_wino_ref_t<4, EPI_LSTM, true>(ctx, a, nbatch) : launch_wino_ref_t<4, EPI_LSTM, false>(ctx, a, nbatch);
        if (NT == 4 && epi == EPI_RAW) return ups ? launch_wino_ref_t<4, EPI_RAW, true>(ctx, a, nbatch) : launch_wino_ref_t<4, EPI_RAW, false>(ctx, a, nbatch);
        if (epi == EPI_POOL_ERR && !ups) return NT == 4 ? launch_wino_ref_t<4, EPI_POOL_ERR, false>(ctx, a, nbatch) : launch_wino_ref_t<3, EPI_POOL_ERR, false>(ctx, a, nbatch);
        return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "no TZ-PA2 reference kernel for NT=%d epilogue=%d upsampled=%d", NT, epi, (int)ups);
    }
    if (NT == 4 && epi == EPI_LSTM) return ups ? launch_wino_t<4, EPI_LSTM, true>(ctx, a, nbatch) : launch_wino_t<4, EPI_LSTM, false>(ctx, a, nbatch);
    if (NT == 4 && epi == EPI_RAW) return ups ? launch_wino_t<4, EPI_RAW, true>(ctx, a, nbatch) : launch_wino_t<4, EPI_RAW, false>(ctx, a, nbatch);
    if (epi == EPI_POOL_ERR && !ups) return NT == 4 ? launch_wino_t<4, EPI_POOL_ERR, false>(ctx, a, nbatch) : launch_wino_t<3, EPI_POOL_ERR, false>(ctx, a, nbatch);
    return tz_fail(ctx, TZ_ERR_UNSUPPORTED, "no TZ-PA2 kernel for NT=%d epilogue=%d upsampled=%d", NT, epi, (int)ups);
}

// The contract in force: what tz_set_contract chose, or -- 0, the default -- a function of the padded frame size that
// encoder and decoder both know: TZ-PA2 from 256 x 256 pixels on, TZ-PA1 below (there the per-frame convolutions are
// latency chains on a mostly idle chip and run on k_convlat).
static constexpr long long TZ_PA2_MIN_PIXELS = 256 * 256;
static constexpr size_t kPBytes = (size_t)160 << 20;      // P[l]: at most this much per level (tz_model_prepare)
static constexpr double kEpartMinIdle = 0.06;   // "E-part ahead": below this idle share of a step's k_wino launches the split is not even tried
static constexpr int kEpartSteps = 3;             // ... steps back to back per timed sample of its measurement
static constexpr float kEpartMinGain = 0.98f;   // ... and it is kept only where it measures at least 2 % faster than the fused step
static int effective_contract(const tz_ctx* ctx) {
    if (ctx->contract) return ctx->contract;
    const tz_model* m = ctx->model;
    return m && (long long)m->Hp * m->Wp >= TZ_PA2_MIN_PIXELS ? 2 : 1;
}

#ifdef TZW_STAMPS
// diagnostic build only (not in tezip_hip.h): the stamps of the last k_wino launches, [8 shape slots][4096 workgroups][16]
extern "C" int tz_debug_wino_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tzw_stamps), sizeof(unsigned long long) * 8 * 4096 * 16) == hipSuccess ? 0 : -3;
}
#endif
extern "C" int tz_set_contract(tz_ctx* ctx, int contract) {
    if (!ctx) return TZ_ERR_INVALID;
    if (contract < 0 || contract > 2)
        return tz_fail(ctx, TZ_ERR_INVALID, "tz_set_contract: %d is not 0 (by frame size), 1 (TZ-PA1) or 2 (TZ-PA2)", contract);
    ctx->contract = contract;   // nothing is re-prepared: both forms of a convolution are packed, the constants are TZ-PA1 in both
    return TZ_OK;
}
extern "C" int tz_get_contract(tz_ctx* ctx) { return ctx ? effective_contract(ctx) : TZ_ERR_INVALID; }

extern "C" int tz_set_conv_impl(tz_ctx* ctx, int lds_dma) {
    if (!ctx) return TZ_ERR_INVALID;
    // a bit field since round 2: reject what is neither 0/1 nor a documented k_convlat selector, so that an old
    // caller's "any non-zero value = on" does not silently pick something else
    if (lds_dma < 0 || lds_dma > 5 || (lds_dma >> 1) > 2)
        return tz_fail(ctx, TZ_ERR_INVALID, "tz_set_conv_impl: %d is not 0 / 1 (+ 2 = never k_convlat, + 4 = k_convlat wherever eligible)", lds_dma);
    ctx->conv_impl = (lds_dma & 1) ? 1 : 0;
    const int lat = lds_dma >> 1;  // 0: cost model (default), 1: never k_convlat, 2: wherever eligible
    ctx->lat_mode = lat == 1 ? 0 : (lat == 2 ? 2 : 1);
    return TZ_OK;
}

// grids that cannot fill the chip: one accumulator tile per wave (k_convlat), see the kernel's header.
// Which kernel is faster is decided by a small cost model fitted to per-launch measurements on the
// MI355X (profiles/r02/small_grid/): k_conv16 costs about 1.08 us per 16-channel x tap slot for
// every workgroup a CU has to run (its K loop is a serial chain of 32-MFMA steps); k_convlat 0.105 us
// per slot while the chain latency bounds it (one workgroup of 16 pixels per CU), 0.078 us per slot and
// workgroup of a CU once the matrix pipe does (0.134 for the 32-pixel workgroups), upsampled sources
// about a tenth more (4-slot blocks: a barrier per 4 slots instead of 9), plus 2.5 us per launch.
struct LatPlan {
    bool use, wide;
    int blocks;
};
static LatPlan lat_plan(const tz_ctx* ctx, int NT, int epi, bool ups, bool fullk, const ConvArgs& a, int nbatch) {
    LatPlan lp = {false, false, 0};
    if (!(a.Wlat && a.nsrc > 0 && fullk && ctx->conv_impl && ctx->lat_mode && (NT == 3 || NT == 4) &&
          (epi == EPI_POOL_ERR || (epi == EPI_LSTM && NT == 4))))
        return lp;
    const int ts_ = (epi != EPI_POOL_ERR && ups) ? 8 : 4;
    const long long wg16 = (long long)a.ncb * a.tiles_x * a.tiles_y * nbatch;
    const long long rows_ = (a.H + ts_ - 1) / ts_ * (ts_ == 8 ? 4 : 1);
    const long long wglat1 = (long long)a.ncb * ((a.W + ts_ - 1) / ts_) * rows_ * nbatch;          // 16 pixels per workgroup
    const long long wglat2 = (long long)a.ncb * ((a.W + 2 * ts_ - 1) / (2 * ts_)) * rows_ * nbatch;  // 32 pixels
    // 32-pixel workgroups (two accumulator chains per wave, half as many workgroups streaming the weights)
    // pay where the matrix pipe is the limit: 166 / 86 us against 175 / 90 with 16 pixels at 512x512, B = 1
    static const long long wide_min = getenv("TEZIP_LAT_WIDE_MIN") ? atoll(getenv("TEZIP_LAT_WIDE_MIN")) : 2560;  // diagnostic
    lp.wide = wglat1 > wide_min;
    int slots = 0;
    for (int s = 0; s < a.nsrc; ++s) slots += a.src[s].cpt * (a.src[s].up ? 4 : 9);
    const double t16 = slots * 1.08 * (double)((wg16 + 255) / 256);
    const double per_slot = lp.wide ? 0.134 * (double)((wglat2 + 255) / 256) : std::max(0.105, 0.078 * (double)((wglat1 + 255) / 256));
    const double tlat = slots * per_slot * (ts_ == 8 ? 1.1 : 1.0) + 2.5;
    lp.use = wg16 <= 768 && (ctx->lat_mode == 2 || tlat < 0.9 * t16);
    lp.blocks = (int)(lp.wide ? wglat2 : wglat1);
    return lp;
}

static int launch_conv(tz_ctx* ctx, int NT, int epi, const ConvArgs& a, int nbatch) {
    tz_prof_scope ps(ctx, TZP_CONV);
    bool ups = false, fullk = true;
    for (int s = 0; s < a.nsrc; ++s) {
        ups = ups || a.src[s].up;
        fullk = fullk && (a.src[s].C % 16) == 0;
        // k_conv16 / k_convlat address a source image through 32-bit byte offsets from its base (patch geometry worked out
        // once per workgroup).  Defensive only since round 4: tz_model_prepare refuses every frame size whose planes could
        // reach that (the general kernel has 32-bit plane offsets of its own), so this never selects another path
        const long long hs = a.src[s].up ? a.H >> 1 : a.H, ws = a.src[s].up ? a.W >> 1 : a.W;
        fullk = fullk && hs * ws * a.src[s].pstride * 4 < (1LL << 32);
    }
    ps.sub = TZP_CONV_GEN;
    if (epi == EPI_RELU && a.nsrc == 1 && !ups && a.src[0].C == 3 && a.Cout == 3 && ctx->conv_impl) {
        ps.sub = TZP_CONV_SMALL;
        hipLaunchKernelGGL((k_conv_small<3, 3>), dim3(a.tiles_x * a.tiles_y * nbatch), dim3(256), 0, ctx->stream, a);
        TZ_HIP(ctx, hipGetLastError());
        return TZ_OK;
    }
    if (a.Wblk && a.nsrc > 0 && !a.src[0].up && a.src[0].pstride == 8 && ctx->conv_impl) {
        const int blocks = a.ncb * a.tiles_x * a.tiles_y * nbatch;
#define TZ_CASE16B(nt, e, u)                                                                                    \
    if (NT == nt && epi == e && ups == u) {                                                                     \
        ps.sub = TZP_CONV16B;                                                                                   \
        hipLaunchKernelGGL((k_conv16b<nt, e, u>), dim3(blocks), dim3(NTHR), 0, ctx->stream, a);                 \
        TZ_HIP(ctx, hipGetLastError());                                                                         \
        return TZ_OK;                                                                                           \
    }
        TZ_CASE16B(1, EPI_LSTM_PACKED, true) TZ_CASE16B(1, EPI_LSTM_PACKED, false)
        TZ_CASE16B(1, EPI_POOL_ERR, false) TZ_CASE16B(3, EPI_POOL_ERR, false) TZ_CASE16B(4, EPI_POOL_ERR, false)
#undef TZ_CASE16B
    }
    // TZ-PA2: the per-frame convolutions of levels >= 1 go through ONE kernel whatever the batch and the frame size
    // (the contract fixes the arithmetic per convolution; k_wino_ref is its cross-check, tz_set_conv_impl(0))
    if (a.Wwino && effective_contract(ctx) == 2 && fullk && (epi == EPI_LSTM || epi == EPI_POOL_ERR || epi == EPI_RAW)) {
        ps.sub = TZP_WINO;
        return launch_wino(ctx, NT, epi, ups, a, nbatch);
    }
    {
        const LatPlan lp = lat_plan(ctx, NT, epi, ups, fullk, a, nbatch);
        if (!lp.use) goto no_lat;
        const bool wide = lp.wide;
        const int blocks = lp.blocks;
        ps.sub = TZP_CONVLAT;
#define TZ_LAT(e, u)                                                                                          \
    do {                                                                                                      \
        if (wide) hipLaunchKernelGGL((k_convlat<e, u, 2>), dim3(blocks), dim3(256), 0, ctx->stream, a);       \
        else hipLaunchKernelGGL((k_convlat<e, u, 1>), dim3(blocks), dim3(256), 0, ctx->stream, a);            \
    } while (0)
        if (epi == EPI_LSTM && ups) TZ_LAT(EPI_LSTM, true);
        else if (epi == EPI_LSTM) TZ_LAT(EPI_LSTM, false);
        else TZ_LAT(EPI_POOL_ERR, false);
#undef TZ_LAT
        TZ_HIP(ctx, hipGetLastError());
        return TZ_OK;
    }
no_lat:
    if (a.Wimg && a.nsrc > 0 && fullk && ctx->conv_impl) {
#define TZ_CASE16(nt, e, u)                          \
    if (NT == nt && epi == e && ups == u) {          \
        ps.sub = TZP_CONV16;                         \
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
    a.Wlat = pc.d_Wlat;
    a.Wwino = pc.d_Wwino;
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
    tz_roctx_range roctx_("tz_model_prepare");
    if (!ctx) return TZ_ERR_INVALID;
    tz_model* m = ctx->model;
    if (!m) return tz_fail(ctx, TZ_ERR_STATE, "tz_model_prepare before tz_model_load");
    const int L = m->L;
    if (Hp <= 0 || Wp <= 0 || max_batch < 1) return tz_fail(ctx, TZ_ERR_INVALID, "bad prepare arguments");
    if ((Hp % (1 << (L - 1))) || (Wp % (1 << (L - 1))) || (Hp % 8) || (Wp % 8))
        return tz_fail(ctx, TZ_ERR_INVALID,
                       "Image size is out of scope for this model: padded size %dx%d must divide by 8 and 2^(levels-1)", Hp, Wp);
    // The convolution kernels address inside ONE frame's plane of a level with 32-bit offsets (LDS-DMA lane offsets, the
    // accumulator starts, the epilogues' lane offsets); batch items and frames are 64-bit strides.  The largest plane is a
    // level's gate columns or error maps: keep it under 2^30 floats (4 GiB).  Which level binds depends on the model; for
    // PredNet (3,48,96,192) it is level 1 (4 x 48 gate columns at a quarter of the pixels): frames up to ~22.3 M pixels
    // (4096 x 4096 passes, 8192 x 8192 does not).  A deviation from the reference, whose frame size is bounded by memory
    // only; there is no 64-bit-offset fallback kernel.
    long long limit_px = -1;   // this model's largest accepted frame, in level-0 pixels
    for (int l = 0; l < L; ++l) {
        const long long widest = std::max<long long>(4LL * m->rstack[l], std::max<long long>(2LL * m->stack[l], 8));
        const long long lim = (((1LL << 30) - 1) / widest) << (2 * l);
        if (limit_px < 0 || lim < limit_px) limit_px = lim;
    }
    for (int l = 0; l < L; ++l) {
        const long long npx = (long long)(Hp >> l) * (Wp >> l);
        const long long widest = std::max<long long>(4LL * m->rstack[l], std::max<long long>(2LL * m->stack[l], 8));
        if (npx * widest >= (1LL << 30))
            return tz_fail(ctx, TZ_ERR_UNSUPPORTED,
                           "frame of %dx%d pixels: level %d holds %lld floats per frame, the kernels address a frame's plane with 32-bit "
                           "offsets (< 2^30 floats); this model accepts frames up to about %lld pixels",
                           Hp, Wp, l, npx * widest, limit_px);
    }
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
    m->epart_choice.assign((size_t)max_batch + 1, (signed char)-1);
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
        // split gate launches (small grids only, see tz_model_predict_batch_dev): [item][pixel][4R] accumulators
        // (also the hand-over buffer of a gate convolution that runs as two k_wino launches: "E-part ahead" below).
        // As many batch slots as fit kPBytes, whatever max_batch is (until round 5: all of max_batch or nothing, so a
        // context prepared for many windows never split its trailing small batches): a batch of n items uses the buffer
        // only where n <= Pcap[l].
        if (l >= 1 && l < L - 1) {
            const size_t per_item = npx * 4 * R * 4;
            const int slots = (int)std::min<size_t>((size_t)max_batch, kPBytes / per_item);
            if (slots >= 1) {
                TZ_TRY(dmalloc(ctx, m, (void**)&m->P[l], (size_t)slots * per_item));
                m->Pcap[l] = slots;
            }
        }
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
        if (l >= 1) TZ_TRY(pack_wino(ctx, m, segs, cols, NT, &m->gate_t1[l]));   // TZ-PA2 form of the same convolution
        // fragment-order copies for the LDS-DMA kernels' prologue / epilogue (row map of the t=1 launch)
        {
            const int tx = (wl(l) + 15) / 16, ty = (hl(l) + 15) / 16, ncb = pg.ncols / (16 * NT);
            const bool parity = l < L - 1;  // an upsampled source => parity tiles
            const size_t nf = (size_t)tx * ty * ncb * 8 * 2 * NT * 256;
            TZ_TRY(dmalloc(ctx, m, (void**)&m->G0f[l], nf * 4));
            if (parity) hipLaunchKernelGGL(k_to_fragments<MAP_PARITY>, dim3(1024), dim3(256), 0, ctx->stream, m->G0[l], hl(l), wl(l), tx, ty, pg.ncols, ncb, NT, m->G0f[l]);
            else hipLaunchKernelGGL(k_to_fragments<MAP_LINEAR>, dim3(1024), dim3(256), 0, ctx->stream, m->G0[l], hl(l), wl(l), tx, ty, pg.ncols, ncb, NT, m->G0f[l]);
            TZ_HIP(ctx, hipGetLastError());
            if (NT == 4) {  // EPI_LSTM reads c_prev per (row, channel of the column block): ncols = R, 16 per block
                const size_t nc = (size_t)tx * ty * ncb * 8 * 2 * 256;
                TZ_TRY(dmalloc(ctx, m, (void**)&m->C0f[l], nc * 4));
                if (parity) hipLaunchKernelGGL(k_to_fragments<MAP_PARITY>, dim3(1024), dim3(256), 0, ctx->stream, m->C0[l], hl(l), wl(l), tx, ty, m->rstack[l], ncb, 1, m->C0f[l]);
                else hipLaunchKernelGGL(k_to_fragments<MAP_LINEAR>, dim3(1024), dim3(256), 0, ctx->stream, m->C0[l], hl(l), wl(l), tx, ty, m->rstack[l], ncb, 1, m->C0f[l]);
                TZ_HIP(ctx, hipGetLastError());
            }
        }
    }
    // ---- A convs (prednet.py:290)
    for (int l = 0; l < L - 1; ++l) {
        int Cout = m->stack[l + 1], NT = plain_nt(Cout);
        TZ_TRY(pack_conv(ctx, m, {Seg{0, 2 * m->stack[l], 0}}, plain_cols(m->a_k(l), m->a_b(l), 2 * m->stack[l], Cout), NT,
                         &m->a_conv[l]));
        if (l >= 1) TZ_TRY(pack_wino(ctx, m, {Seg{0, 2 * m->stack[l], 0}}, plain_cols(m->a_k(l), m->a_b(l), 2 * m->stack[l], Cout), NT, &m->a_conv[l]));
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
    return tz_model_predict_batch_dev(ctx, n, m->d_idx, m->maxB, d_frames_u8, H, W, d_in_stack, d_out_stack, 0);
}

// Launch-only form: d_idx holds [is_key | in_idx | out_idx], each `stride` ints apart, already
// on the device.  Nothing here allocates or copies.
// Which levels CAN run their gate convolution as two launches for a batch of n items (structure only), and how much of the
// chip the step's k_wino launches leave idle: a launch of `items` workgroup-items of `stages` stages each takes
// ceil(items / CUs) rounds of them.
static bool epart_levels(tz_ctx* ctx, tz_model* m, int n, bool* lv, double* idle) {
    const int L = m->L;
    for (int l = 0; l < TZ_MAX_LEVELS; ++l) lv[l] = false;
    *idle = 0.0;
    if (!(effective_contract(ctx) == 2 && ctx->conv_impl && ctx->stream2 && !ctx->split_rollout)) return false;
    auto hl = [&](int l) { return m->Hp >> l; };
    auto wl = [&](int l) { return m->Wp >> l; };
    double busy = 0.0, span = 0.0;
    for (int l = 1; l < L; ++l) {   // the k_wino launches of a step: the gates of level l and A_l (l < L - 1)
        const long long tiles = (long long)((hl(l) + 15) / 16) * ((wl(l) + 15) / 16) * n;
        const int s_e = (2 * m->stack[l]) / 4, s_u = l < L - 1 ? m->rstack[l + 1] / 4 : 0;
        const long long it[2] = {m->gate_t1[l].d_Wwino ? tiles * m->gate_t1[l].ncb : 0,
                                 l < L - 1 && m->a_conv[l].d_Wwino ? tiles * m->a_conv[l].ncb : 0};
        const int st[2] = {s_e + s_u, s_e};
        for (int k = 0; k < 2; ++k) {
            if (!it[k]) continue;
            busy += (double)it[k] / ctx->num_cus * st[k];
            span += (double)((it[k] + ctx->num_cus - 1) / ctx->num_cus) * st[k];
        }
    }
    if (span > 0.0) *idle = 1.0 - busy / span;
    bool any = false;
    for (int l = 1; l < L - 1; ++l) {
        lv[l] = m->P[l] && n <= m->Pcap[l] && m->gate_t1[l].d_Wwino && m->gate_t1[l].NT == 4 && m->gate_t1[l].segs.size() == 2 &&
                (m->rstack[l + 1] % 16) == 0 && (2 * m->stack[l]) % 16 == 0;
        any = any || lv[l];
    }
    return any;
}

static int predict_batch_impl(tz_ctx* ctx, int n, const int* d_idx, int stride, const uint8_t* d_frames_u8, int H, int W,
                              const float* d_in_stack, float* d_out_stack, int slot0, const int* d_next_slot, bool skip_err0,
                              bool* fused_next, bool err0_keys_only, bool use_epart);

// "E-part ahead" by measurement (round 6).  Rounds 5's rule -- split where the step's k_wino launches leave >= 10 % of the
// chip idle -- was calibrated at 512x512 only and is wrong elsewhere in both directions (profiles/r06/epart_shapes.md: 256x256
// one window +16 % left unused, 384x384 two windows -4.7 % taken, 256x256 five windows -7 % / six windows +14 %: whether the
// side workgroups land on CUs the critical path is about to ask for depends on how every launch's last round falls, and on
// the command processor's handling of stream priorities, which HIP does not promise).  So the first call for a batch size
// n of a prepared model MEASURES: the same step -- a pure function of its inputs when the level-0 error maps are formed
// here (skip_err0 off) -- runs fused and split, one untimed and two timed samples of three back-to-back steps each, HIP events on the compute stream,
// and the split is kept for that n where it is at least 2 % faster.  Both forms give the same bits (tests/test_gpu_epart.py),
// so the outcome of the measurement never shows in the results; it costs eighteen extra steps once per (prepare, n).
// TEZIP_EPART=0 / 1 still forbid / force it.  Not tried at all where the idle share is under 6 % (every such case measured
// slower) or no level can split.
static int epart_measure(tz_ctx* ctx, int n, const int* d_idx, int stride, const uint8_t* d_frames_u8, int H, int W,
                         const float* d_in_stack, float* d_out_stack, int slot0, const int* d_next_slot, int* choice) {
    if (!ctx->ev_cal[0]) {
        TZ_HIP(ctx, hipEventCreate(&ctx->ev_cal[0]));
        TZ_HIP(ctx, hipEventCreate(&ctx->ev_cal[1]));
    }
    const bool prof = ctx->prof_on;
    ctx->prof_on = false;                        // (the measurement's launches are not the caller's)
    float best[2] = {1e30f, 1e30f};
    int rc = TZ_OK;
    for (int rep = 0; rep < 3 && rc == TZ_OK; ++rep)
        for (int mode = 0; mode < 2 && rc == TZ_OK; ++mode) {
            bool dummy;
            hipError_t e = hipEventRecord(ctx->ev_cal[0], ctx->stream);
            // kEpartSteps steps back to back between the two events, as a rollout queues them (the host runs ahead of the
            // device): one step alone, with a synchronise on either side, mispredicted a cell by 6 % (384x384, two
            // windows: split 2.9 % faster alone, 3.2 % slower in the rollout; profiles/r06/epart_measure_log.txt)
            for (int k = 0; k < kEpartSteps && rc == TZ_OK; ++k)
                rc = predict_batch_impl(ctx, n, d_idx, stride, d_frames_u8, H, W, d_in_stack, d_out_stack, slot0, d_next_slot, false,
                                        &dummy, false, mode == 1);
            if (e == hipSuccess) e = hipEventRecord(ctx->ev_cal[1], ctx->stream);
            if (e == hipSuccess) e = hipEventSynchronize(ctx->ev_cal[1]);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, ctx->ev_cal[0], ctx->ev_cal[1]);
            if (e != hipSuccess && rc == TZ_OK) rc = tz_fail(ctx, TZ_ERR_HIP, "E-part measurement: %s", hipGetErrorString(e));
            if (rep > 0) best[mode] = std::min(best[mode], ms);
        }
    ctx->prof_on = prof;
    *choice = best[1] < kEpartMinGain * best[0] ? 1 : 0;
    if (getenv("TEZIP_EPART_LOG"))
        fprintf(stderr, "[tezip] E-part ahead, %dx%d batch %d: fused %.1f us, split %.1f us per step -> %s\n", ctx->model->Hp, ctx->model->Wp, n,
                best[0] * 1e3f / kEpartSteps, best[1] * 1e3f / kEpartSteps, *choice ? "split" : "fused");
    return rc;
}

int tz_model_predict_batch_dev(tz_ctx* ctx, int n, const int* d_idx, int stride, const uint8_t* d_frames_u8, int H, int W,
                               const float* d_in_stack, float* d_out_stack, int slot0, const int* d_next_slot, bool skip_err0,
                               bool* fused_next, bool err0_keys_only) {
    tz_model* m = ctx->model;
    if (!m || !m->prepared) return tz_fail(ctx, TZ_ERR_STATE, "model not prepared");
    if (n < 1 || slot0 < 0 || slot0 + n > m->maxB) return tz_fail(ctx, TZ_ERR_INVALID, "batch %d at slot %d outside 1..%d", n, slot0, m->maxB);
    bool use = ctx->epart_mode == 1;
    if (ctx->epart_mode < 0) {
        bool lv[TZ_MAX_LEVELS];
        double idle;
        if (!epart_levels(ctx, m, n, lv, &idle)) {
            use = false;                                       // (depends on the contract in force: not cached)
        } else {
            if (m->epart_choice[n] < 0) {
                if (idle < kEpartMinIdle) {
                    m->epart_choice[n] = 0;
                } else {
                    int choice = 0;
                    TZ_TRY(epart_measure(ctx, n, d_idx, stride, d_frames_u8, H, W, d_in_stack, d_out_stack, slot0, d_next_slot, &choice));
                    m->epart_choice[n] = (signed char)choice;
                    // the measurement's last pass left the NEXT step's level-0 error maps where this step's were: form them again
                    skip_err0 = false;
                    err0_keys_only = false;
                }
            }
            use = m->epart_choice[n] == 1;
        }
    }
    return predict_batch_impl(ctx, n, d_idx, stride, d_frames_u8, H, W, d_in_stack, d_out_stack, slot0, d_next_slot, skip_err0, fused_next,
                              err0_keys_only, use);
}

static int predict_batch_impl(tz_ctx* ctx, int n, const int* d_idx, int stride, const uint8_t* d_frames_u8, int H, int W,
                              const float* d_in_stack, float* d_out_stack, int slot0, const int* d_next_slot, bool skip_err0,
                              bool* fused_next, bool err0_keys_only, bool use_epart) {
    tz_model* m = ctx->model;
    const int L = m->L, Hp = m->Hp, Wp = m->Wp;
    auto hl = [&](int l) { return Hp >> l; };
    auto wl = [&](int l) { return Wp >> l; };
    auto npx = [&](int l) { return (long long)hl(l) * wl(l); };
    if (fused_next) *fused_next = false;
    if (!skip_err0) {   // (skipped when the previous step of every item wrote its E_0 slot from its Ahat_0 epilogue)
        tz_prof_scope ps(ctx, TZP_ERR0);
        int gx = (int)std::min<long long>((npx(0) + 255) / 256, 2048);
        hipLaunchKernelGGL(k_err0, dim3(gx, n), dim3(256), 0, ctx->stream, d_frames_u8, H, W, d_in_stack, d_idx,
                           d_idx + stride, m->Ahat0[0], Hp, Wp, m->stack[0], m->e0s, m->E[0] + (long long)slot0 * npx(0) * m->e0s,
                           err0_keys_only ? 1 : 0);
        TZ_HIP(ctx, hipGetLastError());
    }
    // Small grids (64x64-class frames): the launches of a step are latency chains on a mostly idle chip,
    // so the gate convolution of level l is cut at its source boundary: the part over E_l runs in the
    // SAME launch as A_l (both only need E_l), leaves its accumulators in P_l, and the top-down pass
    // continues the chain over up(R_{l+1}) from there -- same fmaf chain, same bits, 108 + 54 slots off
    // the critical path of a cfg1 step.
    auto gate_args = [&](int l, ConvArgs& a) {
        const PackedConv& pc = m->gate_t1[l];
        memset(&a, 0, sizeof(a));
        const int ec = l == 0 ? m->e0s : 2 * m->stack[l];
        long long ns[2] = {npx(l) * ec, l < L - 1 ? npx(l + 1) * m->rstack[l + 1] : 0};
        const float* ptrs[2] = {m->E[l] + slot0 * ns[0], l < L - 1 ? m->R1[l + 1] + slot0 * ns[1] : nullptr};
        fill_srcs(a, pc, ptrs, ns, ec);
        set_geom(a, hl(l), wl(l));
        a.init = m->G0[l];
        a.initf = m->G0f[l];
        a.Cout = m->rstack[l];
        a.R = m->rstack[l];
        a.aux = m->C0[l];
        a.auxf = m->C0f[l];
        a.out0_nstride = npx(l) * m->rstack[l];
        a.out0 = m->R1[l] + slot0 * a.out0_nstride;
    };
    auto aconv_args = [&](int l, ConvArgs& a) {
        const PackedConv& pc = m->a_conv[l];
        memset(&a, 0, sizeof(a));
        const int ec = l == 0 ? m->e0s : 2 * m->stack[l];  // floats per pixel of E[l]
        long long ns[2] = {npx(l) * ec, 0};
        const float* ptrs[2] = {m->E[l] + slot0 * ns[0], nullptr};
        fill_srcs(a, pc, ptrs, ns, ec);
        set_geom(a, hl(l), wl(l));
        a.Cout = m->stack[l + 1];
        a.aux = m->Ahat0[l + 1];
        a.out0_nstride = npx(l + 1) * 2 * m->stack[l + 1];
        a.out0 = m->E[l + 1] + slot0 * a.out0_nstride;
    };
    // "E-part ahead" (round 5): a gate convolution under TZ-PA2 is two phases in ONE chain per output -- the same-resolution
    // source E_l, then the upsampled R_{l+1} -- and only the second one sits on the step's critical path
    // (A_0 -> A_1 -> ... -> gates L-1 -> ... -> gates 0).  When the launches of a step cannot fill the chip (one window at a time:
    // the top-level gates are 192 workgroups on 256 CUs, and so on down), the first phase of level l runs as a launch of its
    // own on stream2 as soon as E_l exists, leaves every chain as it stands behind the output transform in P_l (float32, what
    // the fused kernel holds in registers at that point), and the launch on the critical path starts from P_l and walks the
    // upsampled source only.  Same chains, same order, same bits; the workgroups of the side launch fill the CUs the
    // critical path leaves idle.  At B = 4 (cfg3) every launch is a whole number of chip-rounds: off.  WHERE it pays is
    // measured, not predicted: tz_model_predict_batch_dev / epart_measure above.
    bool epart[TZ_MAX_LEVELS] = {false};
    if (use_epart) {   // decided by the caller: forced (TEZIP_EPART=1) or measured for this batch size (epart_measure)
        double idle;
        epart_levels(ctx, m, n, epart, &idle);
    }
    auto epart_launch = [&](int l) -> int {   // the launch over E_l, on stream2, behind the A convolution that wrote E_l
        if (!ctx->ev_epart_src[l]) {
            TZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_epart_src[l], hipEventDisableTiming));
            TZ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_epart_done[l], hipEventDisableTiming));
        }
        ConvArgs ge;
        gate_args(l, ge);
        ge.wino_stride = (ge.src[0].C + ge.src[1].C) >> 2;
        ge.wino_first = 0;
        ge.nsrc = 1;                      // the chains over E_l only
        ge.initf = nullptr;
        ge.aux = ge.auxf = nullptr;
        ge.out0_nstride = npx(l) * ge.ncols;
        ge.out0 = m->P[l] + slot0 * ge.out0_nstride;
        ge.ipw = 1;   // short workgroups: a CU a side workgroup holds is one the critical path may be waiting for
        TZ_HIP(ctx, hipEventRecord(ctx->ev_epart_src[l], ctx->stream));
        TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_epart_src[l], 0));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->stream2;       // (the launchers and their profiling scopes take the context's stream)
        int rc = launch_conv(ctx, 4, EPI_RAW, ge, n);
        ctx->stream = main_stream;
        TZ_TRY(rc);
        TZ_HIP(ctx, hipEventRecord(ctx->ev_epart_done[l], ctx->stream2));
        return TZ_OK;
    };
    {
        static const int lv = getenv("TEZIP_EPART_LEVELS") ? atoi(getenv("TEZIP_EPART_LEVELS")) : -1;   // measurements: bit l = level l may split
        if (lv >= 0)
            for (int l = 1; l < L - 1; ++l) epart[l] = epart[l] && ((lv >> l) & 1);
    }
    int epart_top = 0;
    for (int l = 1; l < L - 1; ++l)
        if (epart[l]) epart_top = l;
    bool split[TZ_MAX_LEVELS] = {false};
    for (int l = 0; l < L - 1; ++l) {  // t0 bottom-up
        const PackedConv& pc = m->a_conv[l];
        ConvArgs a;
        aconv_args(l, a);
        // (not under TZ-PA2 where the gate convolution is a k_wino one: the split halves are TZ-PA1 chains)
        if (m->P[l] && slot0 + n <= m->Pcap[l] && m->gate_t1[l].NT == 4 && m->gate_t1[l].segs.size() == 2 &&
            !(effective_contract(ctx) == 2 && m->gate_t1[l].d_Wwino)) {
            ConvArgs ge;
            gate_args(l, ge);
            ge.nsrc = 1;                      // the chain over E_l only
            ge.initf = nullptr;
            ge.out0_nstride = npx(l) * ge.ncols;
            ge.out0 = m->P[l] + slot0 * ge.out0_nstride;
            ConvArgs gu;
            gate_args(l, gu);
            gu.src[0] = gu.src[1];
            gu.nsrc = 1;
            const LatPlan pa = lat_plan(ctx, pc.NT, EPI_POOL_ERR, false, a.src[0].C % 16 == 0, a, n);
            const LatPlan pe = lat_plan(ctx, 4, EPI_LSTM, false, ge.src[0].C % 16 == 0, ge, n);
            const LatPlan pu = lat_plan(ctx, 4, EPI_LSTM, true, gu.src[0].C % 16 == 0, gu, n);
            static const bool split_on = !getenv("TEZIP_LAT_SPLIT") || atoi(getenv("TEZIP_LAT_SPLIT")) != 0;  // diagnostic
            if (split_on && pa.use && pe.use && pu.blocks > 0 && !pa.wide && !pe.wide && pa.blocks + pe.blocks <= 1024) {
                tz_prof_scope ps(ctx, TZP_CONV);
                ps.sub = TZP_CONVLAT;
                hipLaunchKernelGGL((k_convlat_pair<EPI_POOL_ERR, EPI_RAW>), dim3(pa.blocks + pe.blocks), dim3(256), 0, ctx->stream, a, ge,
                                   pa.blocks);
                TZ_HIP(ctx, hipGetLastError());
                split[l] = true;
                continue;
            }
        }
        TZ_TRY(launch_conv(ctx, pc.NT, EPI_POOL_ERR, a, n));
        // The side launches go out behind the A convolution that writes the E of the HIGHEST split level (the A convolutions
        // below it fill the chip by themselves), the highest level first: its second half is the first one the critical path
        // will ask for.
        if (l + 1 == epart_top)
            for (int q = epart_top; q >= 1; --q)
                if (epart[q]) TZ_TRY(epart_launch(q));
    }
    for (int l = L - 1; l >= 0; --l) {  // t1 top-down
        const PackedConv& pc = m->gate_t1[l];
        ConvArgs a;
        gate_args(l, a);
        if (split[l]) {  // continue the chain behind the E_l part
            a.slot0 = a.src[0].cpt * 9;
            a.src[0] = a.src[1];
            a.nsrc = 1;
            a.init_nstride = npx(l) * a.ncols;
            a.init = m->P[l] + slot0 * a.init_nstride;
            a.initf = nullptr;
            const LatPlan pu = lat_plan(ctx, 4, EPI_LSTM, true, a.src[0].C % 16 == 0, a, n);
            tz_prof_scope ps(ctx, TZP_CONV);
            ps.sub = TZP_CONVLAT;
            if (pu.wide) hipLaunchKernelGGL((k_convlat<EPI_LSTM, true, 2>), dim3(pu.blocks), dim3(256), 0, ctx->stream, a);
            else hipLaunchKernelGGL((k_convlat<EPI_LSTM, true, 1>), dim3(pu.blocks), dim3(256), 0, ctx->stream, a);
            TZ_HIP(ctx, hipGetLastError());
            continue;
        }
        if (epart[l]) {   // the chains go on from P_l over the upsampled source
            a.wino_stride = (a.src[0].C + a.src[1].C) >> 2;
            a.wino_first = a.src[0].C >> 2;
            a.init_nstride = npx(l) * a.ncols;
            a.init = m->P[l] + slot0 * a.init_nstride;
            a.initf = nullptr;
            TZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_epart_done[l], 0));
            tz_prof_scope ps(ctx, TZP_CONV);
            ps.sub = TZP_WINO;
            TZ_TRY(launch_wino(ctx, 4, EPI_LSTM, true, a, n, true));
            continue;
        }
        TZ_TRY(launch_conv(ctx, pc.NT, pc.NT == 4 ? EPI_LSTM : EPI_LSTM_PACKED, a, n));
    }
    {  // Ahat_0 at t1 = the prediction (prednet.py:268-271, 293-295)
        const PackedConv& pc = m->ahat0_t1;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        long long ns[2] = {npx(0) * m->rstack[0], 0};
        const float* ptrs[2] = {m->R1[0] + slot0 * ns[0], nullptr};
        fill_srcs(a, pc, ptrs, ns);
        set_geom(a, Hp, Wp);
        a.Cout = m->stack[0];
        a.out0 = d_out_stack;
        a.out0_nstride = npx(0) * m->stack[0];
        a.out_idx = d_idx + 2 * stride;
        a.clip1 = 1;
        // the prediction feeds the next step of its window: k_conv_small can write that step's level-0 error maps as well
        if (d_next_slot && ctx->conv_impl && m->stack[0] == 3 && m->rstack[0] == 3 && m->e0s == 8) {
            a.e0_nstride = npx(0) * m->e0s;
            a.e0_out = m->E[0] + slot0 * a.e0_nstride;   // (the table's slot numbers are positions in the next batch)
            a.e0_ahat = m->Ahat0[0];
            a.e0_slot = d_next_slot;
            if (fused_next) *fused_next = true;
        }
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

// ---- activation probe (diagnostic): the device's TZ-PA1 scalar functions on caller-chosen inputs, and an exhaustive
// check of the division-free 1 - 2/d against the IEEE division it stands for
__global__ __launch_bounds__(256) void k_act_probe(const float* __restrict__ x, size_t n, float* __restrict__ hs,
                                                   float* __restrict__ th) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        hs[i] = tz_hard_sigmoid(x[i]);
        th[i] = tz_tanh(x[i]);
    }
}

__global__ __launch_bounds__(256) void k_recip_check(unsigned lo_bits, unsigned hi_bits, unsigned long long* __restrict__ bad) {
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (unsigned long long b = (unsigned long long)lo_bits + blockIdx.x * blockDim.x + threadIdx.x; b <= hi_bits; b += stride) {
        const float d = __uint_as_float((unsigned)b);
        const float q = 2.0f / d;
        const float want = 1.0f - q;
        mine += __float_as_uint(tz_one_minus_two_over(d)) != __float_as_uint(want);
    }
    if (mine) atomicAdd(bad, mine);
}

extern "C" int tz_act_probe(tz_ctx* ctx, const float* x, size_t n, float* hard_sigmoid, float* tanh_out,
                            unsigned long long* recip_mismatches) {
    if (!ctx || (n && (!x || !hard_sigmoid || !tanh_out))) return TZ_ERR_INVALID;
    int rc = TZ_OK;
    if (n) {
        const void* dx = nullptr;
        std::vector<tz_out> outs;
        tz_out o1, o2;
        rc = tz_dev_in(ctx, x, n * 4, &dx);
        if (rc == TZ_OK) rc = tz_dev_out(ctx, hard_sigmoid, n * 4, &o1);
        if (rc == TZ_OK) outs.push_back(o1);
        if (rc == TZ_OK) rc = tz_dev_out(ctx, tanh_out, n * 4, &o2);
        if (rc == TZ_OK) outs.push_back(o2);
        if (rc == TZ_OK) {
            hipLaunchKernelGGL(k_act_probe, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream,
                               (const float*)dx, n, (float*)o1.dev, (float*)o2.dev);
            if (hipGetLastError() != hipSuccess) rc = tz_fail(ctx, TZ_ERR_HIP, "k_act_probe launch failed");
        }
        if (rc == TZ_OK) rc = tz_dev_out_finish(ctx, outs);
    }
    if (rc == TZ_OK && recip_mismatches) {   // every float32 d in [4, 2^27]: tz_tanh reaches [4.49, 6.6e7]
        void* d_bad = nullptr;
        rc = tz_pool_alloc(ctx, 8, &d_bad);
        if (rc == TZ_OK) {
            TZ_HIP(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
            hipLaunchKernelGGL(k_recip_check, dim3(4096), dim3(256), 0, ctx->stream, 0x40800000u, 0x4D000000u,
                               (unsigned long long*)d_bad);
            TZ_HIP(ctx, hipGetLastError());
            TZ_HIP(ctx, hipMemcpyAsync(recip_mismatches, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
            TZ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
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
