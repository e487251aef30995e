/* CPU ORACLE in C (test infrastructure, NOT product code; see oracle/oracle.py header).
 *
 * Two parts:
 *  1. The integer/byte pipeline of the reference restated for big inputs (the numpy oracle
 *     uses Python loops where the reference does): error_bound, delta, spatial delta,
 *     histogram, remap, inverse scan, reconstruct.  Citations are into /root/reference/src.
 *  2. The PredNet forward pass (prednet.py:143-308) in the canonical arithmetic "TZ-PA1":
 *       conv[y,x,co] = b[co]; for source s in concat order, for each block of 16 channels of
 *                      s, for each tap, for ci in the block: acc = fmaf(x, w, acc)
 *     ('same' zero padding; same-resolution sources use the 9 taps (ky,kx); the upsampled
 *     source up(r_{l+1}) is read at half resolution with 4 collapsed taps whose weights are
 *     float32 sums of the 3x3 taps hitting the same half-resolution pixel: see conv3x3);
 *     activations from
 *     tz_math.h.  The reference's predictor arithmetic lives in keras==2.2.4 /
 *     tensorflow-gpu==1.15 (docs/index.rst:263-264), which are not under /root/reference and
 *     not installable here: PARITY UNPINNED for the predictor.  TZ-PA1 is this build's own
 *     bit-exact definition; it differs from any TF run only by float32 summation order.
 *
 * Build: make -C oracle   (gcc -O3 -ffp-contract=off -mfma -mavx2 -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "tz_math.h"

#define TZO_MAXL 8

/* ------------------------------------------------------------------ integer pipeline */

/* compress.py:292-314: d = (int)(pred_f32*255.0f) - orig over the unpadded crop of one frame */
void tzo_delta_frame(const float* pred_pad, const uint8_t* orig, int H, int W, int Hp, int Wp, int zero,
                     int16_t* out) {
    (void)Hp;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = pred_pad[((size_t)y * Wp + x) * 3 + c] * 255.0f;
                int d = (int)v - (int)orig[((size_t)y * W + x) * 3 + c];
                out[((size_t)y * W + x) * 3 + c] = zero ? 0 : (int16_t)d;
            }
}

/* compress.py:23-70 on one (frame, channel) chain of n elements with element stride `stride`.
 * mode: 0 abs, 1 rel, 2 absrel, 3 pwrel.  Returns 0, or -1 for a negative pwrel / rel bound or a negative
 * rel bound of absrel (the reference raises there). In place on diff. */
int tzo_error_bound(const uint8_t* orig, int16_t* diff, long n, long stride, int mode, double v0, double v1) {
    if (v0 == 0.0) return 0;
    double E = 0.0;
    if (mode == 0) {
        E = fabs(v0);
    } else if (mode == 1 || mode == 2) {
        if (mode == 2 && v1 == 0.0) return 0;
        int mx = 0, mn = 255;
        for (long i = 0; i < n; ++i) {
            int b = orig[i * stride];
            if (b > mx) mx = b;
            if (b < mn) mn = b;
        }
        if (mode == 1) {
            E = (double)(mx - mn) * v0;
        } else {
            double a = fabs(v0), r = (double)(mx - mn) * v1;
            E = a < r ? a : r;
        }
    } else if (mode == 3) {
        if (v0 < 0.0) return -1;
    }
    /* a negative tolerance: the reference assigns (inf + -inf)/2 = NaN into its int array at the first element
     * (compress.py:60-61) and raises */
    if ((mode == 1 && v0 < 0.0) || (mode == 2 && v1 < 0.0)) return -1;
    double u = INFINITY, l = -INFINITY;
    long head = 0;
    for (long i = 0; i < n; ++i) {
        double e = mode == 3 ? (double)orig[i * stride] * v0 : E;
        double df = (double)diff[i * stride];
        double du = df + e, dl = df - e;
        double tu = u < du ? u : du, tl = l > dl ? l : dl;
        if (tu - tl < 0.0) {
            int16_t q = (int16_t)(long)((u + l) / 2);
            for (long j = head; j < i; ++j) diff[j * stride] = q;
            u = INFINITY;
            l = -INFINITY;
            head = i;
        }
        if (du < u) u = du;
        if (l < dl) l = dl;
    }
    if (n > 0) {
        int16_t q = (int16_t)(long)((u + l) / 2);
        for (long j = head; j < n; ++j) diff[j * stride] = q;
    }
    return 0;
}

/* compress.py:73-77 (+ 346-348 when offset!=0): out[0]=in[0], out[i]=in[i-1]-in[i]; y = 1600 - sd */
void tzo_spatial_delta(const int16_t* in, long n, int apply_offset, int16_t* out) {
    int16_t prev = 0;
    for (long i = 0; i < n; ++i) {
        int16_t cur = in[i];
        int16_t sd = i == 0 ? cur : (int16_t)(prev - cur);
        out[i] = apply_offset ? (int16_t)(1600 - sd) : sd;
        prev = cur;
    }
}

/* compress.py:354: bincount over the int16 symbols, bins 0..nbins-1 */
void tzo_histogram(const int16_t* y, long n, long long* hist, int nbins) {
    memset(hist, 0, sizeof(long long) * (size_t)nbins);
    for (long i = 0; i < n; ++i) {
        int v = y[i];
        if (v >= 0 && v < nbins) hist[v]++;
    }
}

/* compress.py:84-90 / decompress.py:31-36 through a 65536-entry LUT indexed by (v + 32768) */
void tzo_lut_apply(const int16_t* in, long n, const int16_t* lut, int16_t* out) {
    for (long i = 0; i < n; ++i) out[i] = lut[(int)in[i] + 32768];
}

/* decompress.py:22-29 (+236 when offset): x[0]=s[0], x[i]=x[i-1]-s[i] with int16 wrap */
void tzo_spatial_undelta(const int16_t* in, long n, int apply_offset, int16_t* out) {
    int16_t prev = 0;
    for (long i = 0; i < n; ++i) {
        int16_t s = apply_offset ? (int16_t)(1600 - in[i]) : in[i];
        int16_t x = i == 0 ? s : (int16_t)(prev - s);
        out[i] = x;
        prev = x;
    }
}

/* decompress.py:252-256,269 for one frame: pred*255 in DOUBLE minus diff, clip, truncate */
void tzo_reconstruct_frame(const float* pred_pad, const uint8_t* key_or_null, const int16_t* diff, int H, int W,
                           int Wp, uint8_t* out) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                size_t o = ((size_t)y * W + x) * 3 + c;
                double base = key_or_null ? ((double)key_or_null[o] / 255) * 255
                                          : (double)pred_pad[((size_t)y * Wp + x) * 3 + c] * 255;
                double r = base - (double)diff[o];
                if (r > 255) r = 255;
                if (r < 0) r = 0;
                out[o] = (uint8_t)r;
            }
}

/* compress.py:246 for one padded frame: sum of (x - pred)^2 in double, in the build's
 * canonical blocked order (4096-element blocks, 256 strided partial sums, halving tree);
 * key_u8 is the UNPADDED frame (pad region compares against 0). */
double tzo_sse_frame(const uint8_t* key_u8, const float* pred_pad, int H, int W, int Hp, int Wp) {
    long n = (long)Hp * Wp * 3;
    double total = 0.0;
    for (long b0 = 0; b0 < n; b0 += 4096) {
        double s[256];
        for (int t = 0; t < 256; ++t) {
            double acc = 0.0;
            for (int j = 0; j < 16; ++j) {
                long i = b0 + (long)j * 256 + t;
                double sq = 0.0;
                if (i < n) {
                    long pix = i / 3;
                    int c = (int)(i % 3), y = (int)(pix / Wp), x = (int)(pix % Wp);
                    float xv = 0.0f;
                    if (y < H && x < W) xv = (float)key_u8[((size_t)y * W + x) * 3 + c] / 255.0f;
                    double d = (double)xv - (double)pred_pad[i];
                    sq = d * d;
                }
                acc = acc + sq;
            }
            s[t] = acc;
        }
        for (int st = 128; st >= 1; st >>= 1)
            for (int t = 0; t < st; ++t) s[t] = s[t] + s[t + st];
        total = total + s[0];
    }
    return total;
}

/* ---------------------------------------------------------------------------- PredNet */
typedef struct {
    int L;
    int stack[TZO_MAXL];  /* A / Ahat channels per level (stack_sizes, train.py:51) */
    int rstack[TZO_MAXL]; /* R channels per level (R_stack_sizes) */
    int Hp, Wp;
    /* weights, HWIO, borrowed pointers; order of the Keras weight list (prednet.py:212):
       a[0..L-2], ahat[0..L-1], c[0..L-1], f[0..L-1], i[0..L-1], o[0..L-1], each kernel then bias */
    const float *a_k[TZO_MAXL], *a_b[TZO_MAXL];
    const float *ahat_k[TZO_MAXL], *ahat_b[TZO_MAXL];
    const float *g_k[4][TZO_MAXL], *g_b[4][TZO_MAXL]; /* gate order here: 0=i 1=f 2=c 3=o */
    /* t=0 state after the top-down pass (input independent), and Ahat at t=0 */
    float *r0[TZO_MAXL], *c0[TZO_MAXL], *ahat0[TZO_MAXL];
    int prepared;
    int contract; /* 1 = TZ-PA1 (every convolution a direct chain), 2 = TZ-PA2 (see conv3x3_wino), tzo_model_set_contract */
} tzo_model;

static int lvl_h(const tzo_model* m, int l) { return m->Hp >> l; }
static int lvl_w(const tzo_model* m, int l) { return m->Wp >> l; }

typedef struct {
    const float* p; /* NULL = all-zero source */
    int C;
    int up; /* 1: stored at half resolution, nearest-upsampled x2 on read (prednet.py:264) */
} tzo_src;

/* 3x3 taps of an upsampled source that read the same half-resolution pixel, for output parity
 * a (0 even / 1 odd coordinate) and collapsed tap d: a=0: {0},{1,2}; a=1: {0,1},{2}. */
static int collapse_set(int a, int d, int out[2]) {
    if (a == 0) {
        if (d == 0) { out[0] = 0; return 1; }
        out[0] = 1; out[1] = 2; return 2;
    }
    if (d == 0) { out[0] = 0; out[1] = 1; return 2; }
    out[0] = 2; return 1;
}

/* Canonical conv (TZ-PA1): out[y][x][co] = chain(bias; sources in concat order; blocks of 16
 * channels; taps; ci).  Same-resolution source: the 9 taps (ky, kx).  Upsampled source
 * (prednet.py:264 followed by the 3x3 conv): read at half resolution with 4 collapsed taps
 * (dy, dx); their weights are the float32 sums (ascending ky, kx) of the 3x3 taps that land on
 * the same half-resolution pixel for the parity of (y, x).  Mathematically the same convolution.
 * W is HWIO with I = sum of source channels. */
static void conv3x3(const tzo_src* src, int nsrc, int H, int W, const float* Wt, const float* bias, int Cout,
                    float* out) {
    int Cin = 0;
    for (int s = 0; s < nsrc; ++s) Cin += src[s].C;
    /* collapsed weights of upsampled sources: Wc[s][cls][tp][ci][co] */
    float* Wc[4] = {0, 0, 0, 0};
    {
        int coff = 0;
        for (int s = 0; s < nsrc; ++s) {
            int C = src[s].C;
            if (src[s].p && src[s].up) {
                Wc[s] = (float*)malloc(sizeof(float) * 16 * (size_t)C * Cout);
                for (int cls = 0; cls < 4; ++cls)
                    for (int tp = 0; tp < 4; ++tp) {
                        int kys[2], kxs[2];
                        int nky = collapse_set(cls >> 1, tp >> 1, kys), nkx = collapse_set(cls & 1, tp & 1, kxs);
                        for (int ci = 0; ci < C; ++ci)
                            for (int co = 0; co < Cout; ++co) {
                                float v = 0.0f;
                                int first = 1;
                                for (int iy = 0; iy < nky; ++iy)
                                    for (int ix = 0; ix < nkx; ++ix) {
                                        float w = Wt[((size_t)(kys[iy] * 3 + kxs[ix]) * Cin + coff + ci) * Cout + co];
                                        v = first ? w : v + w;
                                        first = 0;
                                    }
                                Wc[s][(((size_t)cls * 4 + tp) * C + ci) * Cout + co] = v;
                            }
                    }
            }
            coff += C;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        float* acc = (float*)malloc(sizeof(float) * (size_t)Cout);
        for (int x = 0; x < W; ++x) {
            for (int co = 0; co < Cout; ++co) acc[co] = bias[co];
            int coff = 0;
            for (int s = 0; s < nsrc; ++s) {
                int C = src[s].C;
                if (src[s].p && !src[s].up) { /* an all-zero source leaves every chain unchanged */
                    for (int c0 = 0; c0 < C; c0 += 16) /* blocks of 16 input channels */
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                int yy = y + ky - 1, xx = x + kx - 1;
                                int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
                                const float* ip = inside ? src[s].p + ((size_t)yy * W + xx) * C : NULL;
                                const float* wp = Wt + ((size_t)(ky * 3 + kx) * Cin + coff) * Cout;
                                int c1 = c0 + 16 < C ? c0 + 16 : C;
                                for (int ci = c0; ci < c1; ++ci) {
                                    float xv = inside ? ip[ci] : 0.0f;
                                    const float* wr = wp + (size_t)ci * Cout;
                                    for (int co = 0; co < Cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                                }
                            }
                } else if (src[s].p) {
                    int H2 = H >> 1, W2 = W >> 1, cls = ((y & 1) << 1) | (x & 1);
                    for (int c0 = 0; c0 < C; c0 += 16)
                        for (int tp = 0; tp < 4; ++tp) {
                            int ly = (y >> 1) - 1 + (y & 1) + (tp >> 1), lx = (x >> 1) - 1 + (x & 1) + (tp & 1);
                            int inside = ly >= 0 && ly < H2 && lx >= 0 && lx < W2;
                            const float* ip = inside ? src[s].p + ((size_t)ly * W2 + lx) * C : NULL;
                            const float* wp = Wc[s] + ((size_t)cls * 4 + tp) * C * Cout;
                            int c1 = c0 + 16 < C ? c0 + 16 : C;
                            for (int ci = c0; ci < c1; ++ci) {
                                float xv = inside ? ip[ci] : 0.0f;
                                const float* wr = wp + (size_t)ci * Cout;
                                for (int co = 0; co < Cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                            }
                        }
                }
                coff += C;
            }
            memcpy(out + ((size_t)y * W + x) * Cout, acc, sizeof(float) * (size_t)Cout);
        }
        free(acc);
    }
    for (int s = 0; s < 4; ++s) free(Wc[s]);
}

tzo_model* tzo_model_create(int L, const int* stack, const int* rstack, int Hp, int Wp, const float* const* w) {
    if (L < 1 || L > TZO_MAXL || (Hp % (1 << (L - 1))) || (Wp % (1 << (L - 1)))) return NULL;
    tzo_model* m = (tzo_model*)calloc(1, sizeof(tzo_model));
    m->L = L;
    m->Hp = Hp;
    m->Wp = Wp;
    for (int l = 0; l < L; ++l) {
        m->stack[l] = stack[l];
        m->rstack[l] = rstack[l];
    }
    int k = 0;
    for (int l = 0; l < L - 1; ++l) { m->a_k[l] = w[k++]; m->a_b[l] = w[k++]; }
    for (int l = 0; l < L; ++l) { m->ahat_k[l] = w[k++]; m->ahat_b[l] = w[k++]; }
    const int order[4] = {2, 1, 0, 3}; /* list order c, f, i, o -> our gate slots */
    for (int g = 0; g < 4; ++g)
        for (int l = 0; l < L; ++l) { m->g_k[order[g]][l] = w[k++]; m->g_b[order[g]][l] = w[k++]; }
    /* the contract in force when nothing else is asked for is a function of the padded frame size, which encoder and decoder
       both know (tz_prednet.hip effective_contract): TZ-PA2 from 256 x 256 pixels on, TZ-PA1 below */
    m->contract = (long long)Hp * Wp >= 256LL * 256 ? 2 : 1;
    return m;
}

/* ------------------------------------------------------------------------------------------------------
 * A SECOND statement of the same convolution, arithmetic contract "TZ-PA2": Winograd F(2x2, 3x3) on the
 * per-frame same-resolution sources (2.25x fewer multiplies, still one fmaf chain along the input channels per
 * transformed position, i.e. what an MFMA k-loop computes).  tzo_model_set_contract(m, 2) makes the predictor use
 * it for the per-frame convolutions of levels >= 1 (wino_gate_ok / wino_a_ok); the device kernel k_wino
 * (tezip_amd/csrc/tz_wino_kernels.hip.h) is bit-exact with it.
 *   sources marked in `direct_mask` (the constant r_{t-1} of a gate convolution) come FIRST and the direct way:
 *   init[y][x] = the TZ-PA1 chain from the bias over them (conv3x3; the device keeps it as G0 per model);
 *   tile (ty, tx) = outputs (2ty+a, 2tx+b), a, b in {0,1}; its input patch d[r][c] = x[2ty-1+r][2tx-1+c], zero outside
 *   input transform V = B^T d B, columns first:  t[r][0] = d[r][0]-d[r][2]  t[r][1] = d[r][1]+d[r][2]
 *                                                t[r][2] = d[r][2]-d[r][1]  t[r][3] = d[r][1]-d[r][3]
 *       then rows the same way: V[0][j] = t[0][j]-t[2][j]  V[1][j] = t[1][j]+t[2][j]  V[2][j] = t[2][j]-t[1][j]  V[3][j] = t[1][j]-t[3][j]
 *   weights U = G g G^T per (ci, co), float32, rows first: s = g[0][c]+g[2][c]; w[1][c] = 0.5(s+g[1][c]); w[2][c] = 0.5(s-g[1][c]);
 *       w[0][c] = g[0][c]; w[3][c] = g[2][c]; then the same along c
 *   products: D[i][j] = chain over sources in concat order, channels ascending, from 0:  D = fmaf(V[i][j](ci), U[i][j](ci, co), D)
 *   outputs, transform row i = 0..3 in turn (one pass over the channels each):
 *       Z[i][0] = (D[i][0]+D[i][1])+D[i][2]     Z[i][1] = (D[i][1]-D[i][2])-D[i][3]
 *       y[0][b] = ((init + Z[0][b]) + Z[1][b]) + Z[2][b]     y[1][b] = ((init + Z[1][b]) - Z[2][b]) - Z[3][b]
 *   an upsampled source then continues each output's chain with the collapsed taps and weights of conv3x3 (the four
 *   outputs of a tile are the four parity classes over the same 2x2 half-resolution pixels), in the order: channel
 *   quads ascending, the 4 taps of a quad, the 4 channels of a tap (conv3x3: blocks of 16, taps, 16 channels). */
static void wino_u(const float* Wt, int Cin, int Cout, int ci, int co, float U[4][4]) {
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

static void conv3x3_wino(const tzo_src* src, int nsrc, int direct_mask, int H, int W, const float* Wt, const float* bias, int Cout,
                         float* out) {
    int Cin = 0, Csame = 0;
    for (int s = 0; s < nsrc; ++s) Cin += src[s].C;
    /* the direct part first: init = bias + the TZ-PA1 chains over the sources of direct_mask */
    float* init = NULL;
    tzo_src wsrc[4];
    for (int s = 0; s < nsrc; ++s) wsrc[s] = src[s];
    if (direct_mask) {
        tzo_src dsrc[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int s = 0; s < nsrc; ++s) {
            dsrc[s] = src[s];
            if (!((direct_mask >> s) & 1)) dsrc[s].p = NULL;   /* conv3x3 leaves the chains alone for a NULL source */
            else wsrc[s].p = NULL;
        }
        init = (float*)malloc(sizeof(float) * (size_t)H * W * Cout);
        conv3x3(dsrc, nsrc, H, W, Wt, bias, Cout, init);
    }
    src = wsrc;
    /* transformed weights of the same-resolution sources: Us[pos][c][co], c counting the live same-resolution channels in
       concat order; collapsed weights of upsampled sources as in conv3x3 */
    int coff = 0;
    for (int s = 0; s < nsrc; ++s)
        if (src[s].p && !src[s].up) Csame += src[s].C;
    float* Us = (float*)malloc(sizeof(float) * 16 * (size_t)(Csame ? Csame : 1) * Cout);
    {
        int c = 0;
        coff = 0;
        for (int s = 0; s < nsrc; ++s) {
            if (src[s].p && !src[s].up)
                for (int ci = 0; ci < src[s].C; ++ci, ++c)
                    for (int co = 0; co < Cout; ++co) {
                        float U[4][4];
                        wino_u(Wt, Cin, Cout, coff + ci, co, U);
                        for (int q = 0; q < 16; ++q) Us[((size_t)q * Csame + c) * Cout + co] = U[q >> 2][q & 3];
                    }
            coff += src[s].C;
        }
    }
    float* Wc[4] = {0, 0, 0, 0};
    coff = 0;
    for (int s = 0; s < nsrc; ++s) {
        int C = src[s].C;
        if (src[s].p && src[s].up) {
            Wc[s] = (float*)malloc(sizeof(float) * 16 * (size_t)C * Cout);
            for (int cls = 0; cls < 4; ++cls)
                for (int tp = 0; tp < 4; ++tp) {
                    int kys[2], kxs[2];
                    int nky = collapse_set(cls >> 1, tp >> 1, kys), nkx = collapse_set(cls & 1, tp & 1, kxs);
                    for (int ci = 0; ci < C; ++ci)
                        for (int co = 0; co < Cout; ++co) {
                            float v = 0.0f;
                            int first = 1;
                            for (int iy = 0; iy < nky; ++iy)
                                for (int ix = 0; ix < nkx; ++ix) {
                                    float w = Wt[((size_t)(kys[iy] * 3 + kxs[ix]) * Cin + coff + ci) * Cout + co];
                                    v = first ? w : v + w;
                                    first = 0;
                                }
                            Wc[s][(((size_t)cls * 4 + tp) * C + ci) * Cout + co] = v;
                        }
                }
        }
        coff += C;
    }
    const int TY = (H + 1) / 2, TX = (W + 1) / 2;
#pragma omp parallel for schedule(static)
    for (int ty = 0; ty < TY; ++ty) {
        float* D = (float*)malloc(sizeof(float) * 4 * (size_t)Cout);
        float* Y = (float*)malloc(sizeof(float) * 4 * (size_t)Cout);
        float* V = (float*)malloc(sizeof(float) * 4 * (size_t)(Csame ? Csame : 1));
        for (int tx = 0; tx < TX; ++tx) {
            for (int q = 0; q < 4; ++q) {
                const int yy = 2 * ty + (q >> 1), xx = 2 * tx + (q & 1);
                const float* ip = (init && yy < H && xx < W) ? init + ((size_t)yy * W + xx) * Cout : bias;
                for (int co = 0; co < Cout; ++co) Y[q * Cout + co] = ip[co];
            }
            for (int i = 0; i < 4 && Csame; ++i) {   /* one pass over the channels per transform row */
                int c = 0;
                for (int s = 0; s < nsrc; ++s) {
                    if (!(src[s].p && !src[s].up)) continue;
                    int C = src[s].C;
                    for (int ci = 0; ci < C; ++ci, ++c) {
                        float d[4][4], t[4][4];
                        for (int r = 0; r < 4; ++r)
                            for (int cc = 0; cc < 4; ++cc) {
                                int yy = 2 * ty - 1 + r, xx = 2 * tx - 1 + cc;
                                d[r][cc] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[s].p[((size_t)yy * W + xx) * C + ci] : 0.0f;
                            }
                        for (int r = 0; r < 4; ++r) {
                            t[r][0] = d[r][0] - d[r][2];
                            t[r][1] = d[r][1] + d[r][2];
                            t[r][2] = d[r][2] - d[r][1];
                            t[r][3] = d[r][1] - d[r][3];
                        }
                        for (int j = 0; j < 4; ++j) {
                            float v;
                            if (i == 0) v = t[0][j] - t[2][j];
                            else if (i == 1) v = t[1][j] + t[2][j];
                            else if (i == 2) v = t[2][j] - t[1][j];
                            else v = t[1][j] - t[3][j];
                            V[(size_t)j * Csame + c] = v;
                        }
                    }
                }
                for (int j = 0; j < 4; ++j) {
                    const float* up = Us + (size_t)(i * 4 + j) * Csame * Cout;
                    for (int co = 0; co < Cout; ++co) D[j * Cout + co] = 0.0f;
                    for (int cc = 0; cc < Csame; ++cc) {
                        const float xv = V[(size_t)j * Csame + cc];
                        const float* wr = up + (size_t)cc * Cout;
                        for (int co = 0; co < Cout; ++co) D[j * Cout + co] = fmaf(xv, wr[co], D[j * Cout + co]);
                    }
                }
                for (int co = 0; co < Cout; ++co) {
                    const float d0 = D[co], d1 = D[Cout + co], d2 = D[2 * Cout + co], d3 = D[3 * Cout + co];
                    float z0 = d0 + d1;
                    z0 = z0 + d2;
                    float z1 = d1 - d2;
                    z1 = z1 - d3;
                    if (i == 0) {
                        Y[0 * Cout + co] = Y[0 * Cout + co] + z0;
                        Y[1 * Cout + co] = Y[1 * Cout + co] + z1;
                    } else if (i == 1) {
                        Y[0 * Cout + co] = Y[0 * Cout + co] + z0;
                        Y[1 * Cout + co] = Y[1 * Cout + co] + z1;
                        Y[2 * Cout + co] = Y[2 * Cout + co] + z0;
                        Y[3 * Cout + co] = Y[3 * Cout + co] + z1;
                    } else if (i == 2) {
                        Y[0 * Cout + co] = Y[0 * Cout + co] + z0;
                        Y[1 * Cout + co] = Y[1 * Cout + co] + z1;
                        Y[2 * Cout + co] = Y[2 * Cout + co] - z0;
                        Y[3 * Cout + co] = Y[3 * Cout + co] - z1;
                    } else {
                        Y[2 * Cout + co] = Y[2 * Cout + co] - z0;
                        Y[3 * Cout + co] = Y[3 * Cout + co] - z1;
                    }
                }
            }
            /* upsampled sources: each output's chain goes on with its collapsed taps */
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int y = 2 * ty + a, x = 2 * tx + b;
                    if (y >= H || x >= W) continue;
                    float* acc = Y + (a * 2 + b) * Cout;
                    for (int s = 0; s < nsrc; ++s) {
                        if (!(src[s].p && src[s].up)) continue;
                        int C = src[s].C, H2 = H >> 1, W2 = W >> 1, cls = (a << 1) | b;
                        for (int c0 = 0; c0 < C; c0 += 4)   /* channel quads; taps inside */
                            for (int tp = 0; tp < 4; ++tp) {
                                int ly = (y >> 1) - 1 + (y & 1) + (tp >> 1), lx = (x >> 1) - 1 + (x & 1) + (tp & 1);
                                int inside = ly >= 0 && ly < H2 && lx >= 0 && lx < W2;
                                const float* ip = inside ? src[s].p + ((size_t)ly * W2 + lx) * C : NULL;
                                const float* wp = Wc[s] + ((size_t)cls * 4 + tp) * C * Cout;
                                int c1 = c0 + 4 < C ? c0 + 4 : C;
                                for (int ci = c0; ci < c1; ++ci) {
                                    float xv = inside ? ip[ci] : 0.0f;
                                    const float* wr = wp + (size_t)ci * Cout;
                                    for (int co = 0; co < Cout; ++co) acc[co] = fmaf(xv, wr[co], acc[co]);
                                }
                            }
                    }
                    memcpy(out + ((size_t)y * W + x) * Cout, acc, sizeof(float) * (size_t)Cout);
                }
        }
        free(D);
        free(Y);
        free(V);
    }
    free(Us);
    free(init);
    for (int s = 0; s < 4; ++s) free(Wc[s]);
}

/* Which per-frame convolutions TZ-PA2 evaluates this way: the same predicate as pack_wino (tz_prednet.hip) -- level >= 1,
 * every source a multiple of 16 channels, columns in blocks of 64 (gates: 4 x 16) or 48 / 64 (A convolutions). */
static int wino_gate_ok(const tzo_model* m, int l) {
    return m->contract == 2 && l >= 1 && (2 * m->stack[l]) % 16 == 0 && m->rstack[l] % 16 == 0 &&
           (l == m->L - 1 || m->rstack[l + 1] % 16 == 0);
}
static int wino_a_ok(const tzo_model* m, int l) {
    return m->contract == 2 && l >= 1 && (2 * m->stack[l]) % 16 == 0 && (m->stack[l + 1] % 64 == 0 || m->stack[l + 1] % 48 == 0);
}
void tzo_model_set_contract(tzo_model* m, int contract) {
    m->contract = contract == 0 ? ((long long)m->Hp * m->Wp >= 256LL * 256 ? 2 : 1) : (contract == 2 ? 2 : 1);
}
int tzo_model_get_contract(const tzo_model* m) { return m->contract; }

/* probe for tests: one convolution in either statement.  x: [H][W][C] same-resolution source (or NULL), xu: [H/2][W/2][Cu]
 * upsampled source (or NULL), Wt: HWIO (3,3,C+Cu,Cout) */
void tzo_conv_probe(int winograd, const float* x, int C, const float* xu, int Cu, int H, int W, const float* Wt,
                    const float* bias, int Cout, float* out) {
    tzo_src src[2];
    int ns = 0;
    if (C > 0) { src[ns].p = x; src[ns].C = C; src[ns].up = 0; ++ns; }
    if (Cu > 0) { src[ns].p = xu; src[ns].C = Cu; src[ns].up = 1; ++ns; }
    if (winograd) conv3x3_wino(src, ns, 0, H, W, Wt, bias, Cout, out);
    else conv3x3(src, ns, H, W, Wt, bias, Cout, out);
}

void tzo_model_destroy(tzo_model* m) {
    if (!m) return;
    for (int l = 0; l < TZO_MAXL; ++l) { free(m->r0[l]); free(m->c0[l]); free(m->ahat0[l]); }
    free(m);
}

/* One ConvLSTM update at level l (prednet.py:249-261).  r_prev/c_prev/e_prev may be NULL (zeros). */
static void lstm_level_c(const tzo_model* m, int l, const float* r_prev, const float* c_prev, const float* e_prev,
                         const float* r_up_half, float* r_out, float* c_out, int pa2) {
    int H = lvl_h(m, l), W = lvl_w(m, l), R = m->rstack[l];
    tzo_src src[3];
    int ns = 0;
    src[ns++] = (tzo_src){r_prev, R, 0};
    src[ns++] = (tzo_src){e_prev, 2 * m->stack[l], 0};
    if (l < m->L - 1) src[ns++] = (tzo_src){r_up_half, m->rstack[l + 1], 1};
    size_t n = (size_t)H * W * R;
    float* g[4];
    for (int k = 0; k < 4; ++k) {
        g[k] = (float*)malloc(sizeof(float) * n);
        /* TZ-PA2, per-frame call: the constant r_{t-1} (source 0) directly, e_{t-1} as Winograd chains, then the taps of up(r) */
        if (pa2) conv3x3_wino(src, ns, 1, H, W, m->g_k[k][l], m->g_b[k][l], R, g[k]);
        else conv3x3(src, ns, H, W, m->g_k[k][l], m->g_b[k][l], R, g[k]);
    }
    for (size_t j = 0; j < n; ++j) {
        float i_ = tzo_hard_sigmoid(g[0][j]), f_ = tzo_hard_sigmoid(g[1][j]), o_ = tzo_hard_sigmoid(g[3][j]);
        float g_ = tzo_tanh(g[2][j]);
        float t1 = f_ * (c_prev ? c_prev[j] : 0.0f);
        float t2 = i_ * g_;
        float c = t1 + t2;
        c_out[j] = c;
        r_out[j] = o_ * tzo_tanh(c);
    }
    for (int k = 0; k < 4; ++k) free(g[k]);
}

static void lstm_level(const tzo_model* m, int l, const float* r_prev, const float* c_prev, const float* e_prev,
                       const float* r_up_half, float* r_out, float* c_out) {
    lstm_level_c(m, l, r_prev, c_prev, e_prev, r_up_half, r_out, c_out, 0);
}

/* ahat_l = relu(conv(r_l)) (+ min(.,1) at l=0)   prednet.py:268-271 */
static void ahat_level(const tzo_model* m, int l, const float* r, float* out) {
    int H = lvl_h(m, l), W = lvl_w(m, l);
    tzo_src s = {r, m->rstack[l], 0};
    conv3x3(&s, 1, H, W, m->ahat_k[l], m->ahat_b[l], m->stack[l], out);
    size_t n = (size_t)H * W * m->stack[l];
    for (size_t j = 0; j < n; ++j) {
        float v = tzo_relu(out[j]);
        if (l == 0 && v > 1.0f) v = 1.0f;
        out[j] = v;
    }
}

/* e_l = concat(relu(ahat - a), relu(a - ahat))   prednet.py:274-277 */
static void err_level(const tzo_model* m, int l, const float* ahat, const float* a, float* e) {
    int C = m->stack[l];
    size_t npx = (size_t)lvl_h(m, l) * lvl_w(m, l);
    for (size_t p = 0; p < npx; ++p)
        for (int c = 0; c < C; ++c) {
            float h = ahat[p * C + c], av = a[p * C + c];
            float d1 = h - av, d2 = av - h;
            e[p * 2 * C + c] = tzo_relu(d1);
            e[p * 2 * C + C + c] = tzo_relu(d2);
        }
}

/* a_{l+1} = maxpool2x2(relu(conv(e_l)))   prednet.py:289-291 */
static void a_level(const tzo_model* m, int l, const float* e, float* a_next, int pa2) {
    int H = lvl_h(m, l), W = lvl_w(m, l), C = m->stack[l + 1];
    float* full = (float*)malloc(sizeof(float) * (size_t)H * W * C);
    tzo_src s = {e, 2 * m->stack[l], 0};
    if (pa2) conv3x3_wino(&s, 1, 0, H, W, m->a_k[l], m->a_b[l], C, full);
    else conv3x3(&s, 1, H, W, m->a_k[l], m->a_b[l], C, full);
    int H2 = H / 2, W2 = W / 2;
    for (int y = 0; y < H2; ++y)
        for (int x = 0; x < W2; ++x)
            for (int c = 0; c < C; ++c) {
                float v = 0.0f; /* relu outputs are >= 0, so 0 is the identity of the max */
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx) {
                        float t = tzo_relu(full[((size_t)(2 * y + dy) * W + 2 * x + dx) * C + c]);
                        if (t > v) v = t;
                    }
                a_next[((size_t)y * W2 + x) * C + c] = v;
            }
    free(full);
}

static float* falloc(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

/* t=0 top-down from zero state + Ahat at t=0 for every level (input independent). */
void tzo_model_prepare(tzo_model* m) {
    if (m->prepared) return;
    for (int l = m->L - 1; l >= 0; --l) {
        size_t n = (size_t)lvl_h(m, l) * lvl_w(m, l) * m->rstack[l];
        m->r0[l] = falloc(n);
        m->c0[l] = falloc(n);
        lstm_level(m, l, NULL, NULL, NULL, l < m->L - 1 ? m->r0[l + 1] : NULL, m->r0[l], m->c0[l]);
    }
    for (int l = 0; l < m->L; ++l) {
        m->ahat0[l] = falloc((size_t)lvl_h(m, l) * lvl_w(m, l) * m->stack[l]);
        ahat_level(m, l, m->r0[l], m->ahat0[l]);
    }
    m->prepared = 1;
}

/* X_hat[0,0] (compress.py:197): the t=0 frame prediction = Ahat_0 at t=0 */
void tzo_model_c0(tzo_model* m, float* out) {
    tzo_model_prepare(m);
    memcpy(out, m->ahat0[0], sizeof(float) * (size_t)m->Hp * m->Wp * m->stack[0]);
}

/* X_hat[0,1] = f(frame) (compress.py:224-229): t0 bottom-up with a_0 = frame, then the t1
 * top-down pass and Ahat_0.  Live work only; tzo_predict2_literal does the same the long way.
 * dbg (optional, may be NULL): array of 3*L pointers receiving e_l(t0), r_l(t1), c_l(t1). */
void tzo_model_next(tzo_model* m, const float* frame, float* pred, float** dbg) {
    tzo_model_prepare(m);
    int L = m->L;
    float *e[TZO_MAXL] = {0}, *r1[TZO_MAXL] = {0}, *c1[TZO_MAXL] = {0};
    const float* a = frame;
    float* a_own = NULL;
    for (int l = 0; l < L; ++l) {
        size_t npx = (size_t)lvl_h(m, l) * lvl_w(m, l);
        e[l] = falloc(npx * 2 * m->stack[l]);
        err_level(m, l, m->ahat0[l], a, e[l]);
        if (l < L - 1) {
            float* an = falloc((npx / 4) * m->stack[l + 1]);
            a_level(m, l, e[l], an, wino_a_ok(m, l));
            free(a_own);
            a_own = an;
            a = an;
        }
    }
    free(a_own);
    for (int l = L - 1; l >= 0; --l) {
        size_t n = (size_t)lvl_h(m, l) * lvl_w(m, l) * m->rstack[l];
        r1[l] = falloc(n);
        c1[l] = falloc(n);
        lstm_level_c(m, l, m->r0[l], m->c0[l], e[l], l < L - 1 ? r1[l + 1] : NULL, r1[l], c1[l], wino_gate_ok(m, l));
    }
    ahat_level(m, 0, r1[0], pred);
    for (int l = 0; l < L; ++l) {
        if (dbg) {
            size_t npx = (size_t)lvl_h(m, l) * lvl_w(m, l);
            if (dbg[l]) memcpy(dbg[l], e[l], sizeof(float) * npx * 2 * m->stack[l]);
            if (dbg[L + l]) memcpy(dbg[L + l], r1[l], sizeof(float) * npx * m->rstack[l]);
            if (dbg[2 * L + l]) memcpy(dbg[2 * L + l], c1[l], sizeof(float) * npx * m->rstack[l]);
        }
        free(e[l]);
        free(r1[l]);
        free(c1[l]);
    }
}

/* Literal evaluation of predict on (1,2,Hp,Wp,C): two generic steps from zero state
 * (prednet.py:235-308 via K.rnn), second input all zeros (compress.py:225-226).
 * out0 = X_hat[0,0], out1 = X_hat[0,1].  Used to show the live-work shortcut is exact. */
void tzo_predict2_literal(tzo_model* m, const float* frame, float* out0, float* out1) {
    int L = m->L;
    float *r[TZO_MAXL], *c[TZO_MAXL], *e[TZO_MAXL];
    for (int l = 0; l < L; ++l) {
        size_t npx = (size_t)lvl_h(m, l) * lvl_w(m, l);
        r[l] = falloc(npx * m->rstack[l]);
        c[l] = falloc(npx * m->rstack[l]);
        e[l] = falloc(npx * 2 * m->stack[l]);
    }
    float* zeros = falloc((size_t)m->Hp * m->Wp * m->stack[0]);
    for (int t = 0; t < 2; ++t) {
        float *rn[TZO_MAXL], *cn[TZO_MAXL];
        for (int l = L - 1; l >= 0; --l) {
            size_t n = (size_t)lvl_h(m, l) * lvl_w(m, l) * m->rstack[l];
            rn[l] = falloc(n);
            cn[l] = falloc(n);
            lstm_level(m, l, r[l], c[l], e[l], l < L - 1 ? rn[l + 1] : NULL, rn[l], cn[l]);
        }
        const float* a = t == 0 ? frame : zeros;
        float* a_own = NULL;
        for (int l = 0; l < L; ++l) {
            size_t npx = (size_t)lvl_h(m, l) * lvl_w(m, l);
            float* ah = falloc(npx * m->stack[l]);
            ahat_level(m, l, rn[l], ah);
            if (l == 0) memcpy(t == 0 ? out0 : out1, ah, sizeof(float) * npx * m->stack[0]);
            err_level(m, l, ah, a, e[l]);
            free(ah);
            if (l < L - 1) {
                float* an = falloc((npx / 4) * m->stack[l + 1]);
                a_level(m, l, e[l], an, 0);   /* (the literal form is the TZ-PA1 cross-check) */
                free(a_own);
                a_own = an;
                a = an;
            }
        }
        free(a_own);
        for (int l = 0; l < L; ++l) {
            free(r[l]);
            free(c[l]);
            r[l] = rn[l];
            c[l] = cn[l];
        }
    }
    for (int l = 0; l < L; ++l) { free(r[l]); free(c[l]); free(e[l]); }
    free(zeros);
}

/* scalar function probes so tests can compare activations bit for bit with the device */
void tzo_act_probe(const float* x, long n, float* hs, float* th) {
    for (long i = 0; i < n; ++i) {
        hs[i] = tzo_hard_sigmoid(x[i]);
        th[i] = tzo_tanh(x[i]);
    }
}

/* key-frame byte -> float32 model input: float32(k)/255 (compress.py:138) */
void tzo_u8_to_f32_frame(const uint8_t* key, int H, int W, int Hp, int Wp, float* out) {
    memset(out, 0, sizeof(float) * (size_t)Hp * Wp * 3);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c)
                out[((size_t)y * Wp + x) * 3 + c] = (float)key[((size_t)y * W + x) * 3 + c] / 255.0f;
}
