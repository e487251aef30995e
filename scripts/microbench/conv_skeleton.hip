// Skeleton of k_conv16's K loop, built up feature by feature to find what keeps the MFMA pipe idle:
//   F_ADDR  real A-fragment addressing (tap offsets, XOR swizzle) instead of fixed addresses
//   F_WDMA  weight LDS-DMA per step (4 x 1 KB from a global stream) + vmcnt(0) before the barrier
//   F_PDMA  patch LDS-DMA per 9 steps (21 x 1 KB gather) into the other patch buffer
//   F_BAR   one workgroup barrier per step
// 512-thread workgroups, 50 KB LDS (3 per CU), 2 x 4 MFMA tiles per wave, 32 MFMAs per step.
//   hipcc --offload-arch=gfx950 -O3 conv_skeleton.hip -o conv_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { F_ADDR = 1, F_WDMA = 2, F_PDMA = 4, F_BAR = 8, F_PRIO = 16, F_PLANAR = 32, F_PREF = 64, F_W3 = 128 };

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int F>
__global__ __launch_bounds__(512, 6) void k(float* out, unsigned long long* stamps, int nblk, const float* act, const float* wimg,
                                            int W, int C) {
    __shared__ __attribute__((aligned(16))) float smem[50 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int tile = blockIdx.x % 256, cb = blockIdx.x / 256 % 3;
    const int ty0 = (tile / 16) * 16, tx0 = (tile % 16) * 16;
    for (int i = tid; i < 50 * 256; i += 512) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tpy[2] = {2 * wv, 2 * wv + 1}, tpx[2] = {lane & 15, lane & 15};
    const int qsrc = 4 * ((lane & 3) ^ ((lane >> 3) & 3));
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int cur = 0, pi = 0, slot0 = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        if (F & F_PDMA) {
            float* dst = smem + (pi ? 0 : ((F & F_W3) ? 19 : 21)) * 256;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int piece = wv + 8 * j;
                if (piece < ((F & F_W3) ? 19 : 21)) {
                    const int slot = piece * 16 + (lane >> 2);
                    const int y = slot / 18, x = slot - y * 18;
                    const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                    const bool ok = slot < 324 && yy >= 0 && yy < W && xx >= 0 && xx < W;
                    glds16(ok ? act + ((long long)yy * W + xx) * C + (blk % (C / 16)) * 16 + qsrc : act, dst + piece * 256);
                }
            }
        }
#pragma unroll 1
        for (int st = 0; st < 9; ++st) {
            const int sg = slot0 + st;  // global step number
            if (F & F_W3) {
                // three weight buffers: the DMA issued in step s fills the buffer of step s+2
                if ((wv >> 2) == (sg & 1))
                    glds16(wimg + (((long long)(sg + 2) * 3 + cb) * 4 + (wv & 3)) * 256 + lane * 4,
                           smem + (38 + 4 * ((sg + 2) % 3) + (wv & 3)) * 256);
            } else if (F & F_WDMA) {
                if ((wv >> 2) == (cur ^ 1))
                    glds16(wimg + (((long long)(slot0 + st + 1) * 3 + cb) * 4 + (wv & 3)) * 256 + lane * 4,
                           smem + (42 + 4 * (cur ^ 1) + (wv & 3)) * 256);
            }
            const float* pa = smem + (pi ? ((F & F_W3) ? 19 : 21) : 0) * 256;
            const float* wb = smem + ((F & F_W3) ? 38 + 4 * (sg % 3) : 42 + 4 * cur) * 256 + lane * 4;
            float fa[2][4];
            if (F & F_PLANAR) {
                // quad-planar patch image: item (slot, quad q) at q * 336 + slot -> tap and k-step are pure offsets
                const int toff = ((st / 3) * 18 + st % 3) * 4;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[(tpy[mt] * 18 + tpx[mt]) * 4 + g + toff + kk * 1344];
            } else if (F & F_ADDR) {
                const int dy = st / 3, dx = st - 3 * dy;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int slot = (tpy[mt] + dy) * 18 + tpx[mt] + dx, f = (slot >> 1) & 3;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[slot * 16 + 4 * (kk ^ f) + g];
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[(tpy[mt] * 18 + tpx[mt]) * 16 + 4 * kk + g];
            }
            if (F & F_PRIO) __builtin_amdgcn_s_setprio(1);
            if (F & F_PREF) {
                // fragments of k-step kk+1 are requested before the MFMAs of k-step kk are issued
                f32x4 fb = *(const f32x4*)(wb);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    f32x4 fbn = fb;
                    if (kk < 3) fbn = *(const f32x4*)(wb + (kk + 1) * 256);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    fb = fbn;
                }
            } else
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 fb = *(const f32x4*)(wb + kk * 256);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
            }
            if (F & F_PRIO) __builtin_amdgcn_s_setprio(0);
            if (F & F_W3) {
                if ((wv >> 2) == (sg & 1)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (F & (F_WDMA | F_PDMA)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (F & F_BAR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            cur ^= 1;
        }
        slot0 += 9;
        pi ^= 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}


// Variant: one step = one ROW of taps (3 taps, 96 MFMAs per barrier, 12 KB of weights per step),
// single patch buffer re-filled between blocks.  LDS 45 KB.
template <int F>
__global__ __launch_bounds__(512, 6) void k3(float* out, unsigned long long* stamps, int nblk, const float* act, const float* wimg,
                                             int W, int C) {
    __shared__ __attribute__((aligned(16))) float smem[45 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int tile = blockIdx.x % 256, cb = blockIdx.x / 256 % 3;
    const int ty0 = (tile / 16) * 16, tx0 = (tile % 16) * 16;
    for (int i = tid; i < 45 * 256; i += 512) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tpy[2] = {2 * wv, 2 * wv + 1}, tpx[2] = {lane & 15, lane & 15};
    const int qsrc = 4 * ((lane & 3) ^ ((lane >> 3) & 3));
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int cur = 0, slot0 = 0;
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll 1
        for (int st = 0; st < 3; ++st) {
            {   // weights of the next row: 12 pieces, wave w takes w and (w < 4) w + 8
                const float* src = wimg + (((long long)(slot0 + 3 * (st + 1)) * 3 + cb) * 4) * 256 + lane * 4;
                float* dst = smem + (21 + 12 * (cur ^ 1)) * 256;
                glds16(src + wv * 256, dst + wv * 256);
                if (wv < 4) glds16(src + (wv + 8) * 256, dst + (wv + 8) * 256);
            }
            const float* pa = smem;
            const float* wb = smem + (21 + 12 * cur) * 256 + lane * 4;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float fa[2][4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int slot = (tpy[mt] + st) * 18 + tpx[mt] + dx, f = (slot >> 1) & 3;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[slot * 16 + 4 * (kk ^ f) + g];
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 fb = *(const f32x4*)(wb + (dx * 4 + kk) * 256);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        slot0 += 9;
        if (F & F_PDMA) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int piece = wv + 8 * j;
                if (piece < 21) {
                    const int slot = piece * 16 + (lane >> 2);
                    const int y = slot / 18, x = slot - y * 18;
                    const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                    const bool ok = slot < 324 && yy >= 0 && yy < W && xx >= 0 && xx < W;
                    glds16(ok ? act + ((long long)yy * W + xx) * C + (blk % (C / 16)) * 16 + qsrc : act, smem + piece * 256);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Upsampled-phase skeleton: 16 KB of weights per 32-MFMA step (4 parity classes), 7-piece patch per
// 4 steps.  WPS = weight pieces per wave and step (2 = as k_conv16; 1, 0 = what less DMA would buy).
template <int WPS>
__global__ __launch_bounds__(512, 6) void k4(float* out, unsigned long long* stamps, int nblk, const float* act, const float* wimg,
                                             int W, int C) {
    __shared__ __attribute__((aligned(16))) float smem[50 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int cb = blockIdx.x / 256 % 3;
    for (int i = tid; i < 50 * 256; i += 512) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int abase[2] = {4 * ((wv & 3) * 10 + (lane & 7)) + g, 4 * (((wv & 3) + 1) * 10 + (lane & 7)) + g};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int cur = 0, pi = 0, slot0 = 0;
    nblk = nblk * 9 / 4;  // same MFMA count as the 9-step blocks
    for (int blk = 0; blk < nblk; ++blk) {
        if (wv < 7) glds16(act + ((long long)(blockIdx.x % 256) * 4096 + wv * 64 + lane) * 4, smem + ((pi ? 0 : 7) + wv) * 256);
#pragma unroll 1
        for (int st = 0; st < 4; ++st) {
            if (WPS >= 1) {
                const float* src = wimg + (((long long)((slot0 + st + 1) % 100) * 3 + cb) * 4 + 2 * (wv & 1)) * 256 + lane * 4;
                float* dst = smem + (14 + 16 * (cur ^ 1) + 2 * wv) * 256;
                glds16(src, dst);
                if (WPS >= 2) glds16(src + 256, dst + 256);
            }
            const float* pa = smem + (pi ? 7 : 0) * 256;
            const float* wb = smem + (14 + 16 * cur + 4 * (wv >> 1)) * 256 + lane * 4;
            const int toff = 4 * ((st >> 1) * 10 + (st & 1));
            float fa[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[abase[mt] + toff + kk * 448];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 fb = *(const f32x4*)(wb + kk * 256);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        slot0 += 4;
        pi ^= 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Variant: weights two steps ahead in THREE buffers with compile-time buffer indices (tap (dy,dx) lives
// in buffer dx; the dx loop is unrolled), quad-planar A addressing, double-buffered 20.5-piece patches.
// Waves 0-3 issue on even global steps, 4-7 on odd ones; a wave waits only for a piece it issued a
// whole step earlier.  LDS 53 KB.
__global__ __launch_bounds__(512, 6) void k6(float* out, unsigned long long* stamps, int nblk, const float* act, const float* wimg,
                                             int W, int C) {
    __shared__ __attribute__((aligned(16))) float smem[53 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int tile = blockIdx.x % 256, cb = blockIdx.x / 256 % 3;
    const int ty0 = (tile / 16) * 16, tx0 = (tile % 16) * 16;
    for (int i = tid; i < 53 * 256; i += 512) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int abase[2] = {4 * ((2 * wv) * 18 + (lane & 15)) + g, 4 * ((2 * wv + 1) * 18 + (lane & 15)) + g};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int pi = 0, slot0 = 0;
    const int half = wv >> 2;
    for (int blk = 0; blk < nblk; ++blk) {
        {
            float* dst = smem + (pi ? 0 : 5248);   // 20.5 pieces = 5248 floats
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int piece = wv + 8 * j;
                if (piece < 21) {
                    const int i = piece * 64 + lane, q = i / 328, slot = i - q * 328;
                    const int y = slot / 18, x = slot - y * 18;
                    const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                    const bool ok = i < 1312 && slot < 324 && yy >= 0 && yy < W && xx >= 0 && xx < W;
                    if (i < 1312) glds16(ok ? act + ((long long)yy * W + xx) * C + (blk % (C / 16)) * 16 + 4 * q : act, dst + piece * 256);
                }
            }
        }
        const float* pa = smem + (pi ? 5248 : 0);
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int sg = slot0 + 3 * dy + dx;   // global step; 9 per block so parity alternates across blocks too
                if (half == (sg & 1))
                    glds16(wimg + (((long long)(sg + 2) * 3 + cb) * 4 + (wv & 3)) * 256 + lane * 4,
                           smem + (41 + 4 * ((dx + 2) % 3) + (wv & 3)) * 256);
                const float* wb = smem + (41 + 4 * dx) * 256 + lane * 4;
                const int toff = 4 * (dy * 18 + dx);
                float fa[2][4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = pa[abase[mt] + toff + kk * 1312];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 fb = *(const f32x4*)(wb + kk * 256);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
                }
                if (half == (sg & 1)) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        slot0 += 9;
        pi ^= 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Variant: 4 x 4 MFMA tiles per wave (64 pixels x 64 columns; workgroup = 32x16 pixels), so the B
// fragments, the weight DMA and the barrier are amortised over 64 MFMAs per wave and step.
// ~100 VGPRs -> 2 workgroups per CU; single 34x18 patch buffer (39 pieces) + 2 x 4 weight pieces.
template <int DBUF>
__global__ __launch_bounds__(512, 4) void k7(float* out, unsigned long long* stamps, int nblk, const float* act, const float* wimg,
                                             int W, int C) {
    __shared__ __attribute__((aligned(16))) float smem[(DBUF ? 78 : 39) * 256 + 8 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int tile = blockIdx.x % 128, cb = blockIdx.x / 128 % 3;
    const int ty0 = (tile / 16) * 32, tx0 = (tile % 16) * 16;
    for (int i = tid; i < 47 * 256; i += 512) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int abase[4];
    for (int mt = 0; mt < 4; ++mt) abase[mt] = 4 * ((4 * wv + mt) * 18 + (lane & 15)) + g;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int cur = 0, pi = 0, slot0 = 0;
    constexpr int WOFF = (DBUF ? 78 : 39) * 256;
    for (int blk = 0; blk < nblk; ++blk) {
        if (DBUF) {
            float* dst = smem + (pi ? 0 : 39) * 256;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int piece = wv + 8 * j;
                if (piece < 39) {
                    const int i = piece * 64 + lane, q = i / 624, slot = i - q * 624;
                    const int y = slot / 18, x = slot - y * 18;
                    const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                    const bool ok = slot < 612 && yy >= 0 && yy < W && xx >= 0 && xx < W;
                    glds16(ok ? act + ((long long)yy * W + xx) * C + (blk % (C / 16)) * 16 + 4 * q : act, dst + piece * 256);
                }
            }
        }
        const float* pa = smem + (DBUF && pi ? 39 : 0) * 256;
#pragma unroll 1
        for (int st = 0; st < 9; ++st) {
            if ((wv >> 2) == (cur ^ 1))
                glds16(wimg + (((long long)(slot0 + st + 1) * 3 + cb) * 4 + (wv & 3)) * 256 + lane * 4,
                       smem + WOFF + (4 * (cur ^ 1) + (wv & 3)) * 256);
            const float* wb = smem + WOFF + 4 * cur * 256 + lane * 4;
            const int toff = 4 * ((st / 3) * 18 + st % 3);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 fb = *(const f32x4*)(wb + kk * 256);
                float fa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) fa[mt] = pa[abase[mt] + toff + kk * 2496];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        slot0 += 9;
        pi ^= 1;
        if (!DBUF) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int piece = wv + 8 * j;
                if (piece < 39) {
                    const int i = piece * 64 + lane, q = i / 624, slot = i - q * 624;
                    const int y = slot / 18, x = slot - y * 18;
                    const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                    const bool ok = slot < 612 && yy >= 0 && yy < W && xx >= 0 && xx < W;
                    glds16(ok ? act + ((long long)yy * W + xx) * C + (blk % (C / 16)) * 16 + 4 * q : act, smem + piece * 256);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Variant: the k_conv16 K loop at 4 workgroups per CU (8 waves/SIMD): <= 64 VGPRs, single patch
// buffer (21 + 8 = 29 pieces), patch refilled between blocks behind one extra barrier.
__global__ __launch_bounds__(512, 8) void k8(float* out, unsigned long long* stamps, int nblk, const float* act, const float* wimg,
                                             int W, int C) {
    __shared__ __attribute__((aligned(16))) float smem[29 * 256];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    const int tile = blockIdx.x % 256, cb = blockIdx.x / 256 % 3;
    const int ty0 = (tile / 16) * 16, tx0 = (tile % 16) * 16;
    for (int i = tid; i < 29 * 256; i += 512) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int abase[2] = {4 * ((2 * wv) * 18 + (lane & 15)) + g, 4 * ((2 * wv + 1) * 18 + (lane & 15)) + g};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int cur = 0, slot0 = 0;
    for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll 1
        for (int st = 0; st < 9; ++st) {
            if ((wv >> 2) == (cur ^ 1))
                glds16(wimg + (((long long)(slot0 + st + 1) * 3 + cb) * 4 + (wv & 3)) * 256 + lane * 4,
                       smem + (21 + 4 * (cur ^ 1) + (wv & 3)) * 256);
            const float* wb = smem + (21 + 4 * cur) * 256 + lane * 4;
            const int toff = 4 * ((st / 3) * 18 + st % 3);
            float fa[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fa[mt][kk] = smem[abase[mt] + toff + kk * 1344];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 fb = *(const f32x4*)(wb + kk * 256);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt][kk], fb[nt], acc[mt][nt], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur ^= 1;
        }
        slot0 += 9;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int piece = wv + 8 * j;
            if (piece < 21) {
                const int i = piece * 64 + lane, q = i / 336, slot = i - q * 336;
                const int y = slot / 18, x = slot - y * 18;
                const int yy = ty0 - 1 + y, xx = tx0 - 1 + x;
                const bool ok = slot < 324 && yy >= 0 && yy < W && xx >= 0 && xx < W;
                glds16(ok ? act + ((long long)yy * W + xx) * C + (blk % (C / 16)) * 16 + 4 * q : act, smem + piece * 256);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static float *g_out, *g_act, *g_w;
static unsigned long long* g_st;
typedef void (*kern_t)(float*, unsigned long long*, int, const float*, const float*, int, int);
template <int F>
static void run(const char* name, int blocks, int nblk, kern_t kf = nullptr, double mt_scale = 1.0) {
    if (!kf) kf = k<F>;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kf, dim3(blocks), dim3(512), 0, 0, g_out, g_st, nblk, g_act, g_w, 256, 96);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kf, dim3(blocks), dim3(512), 0, 0, g_out, g_st, nblk, g_act, g_w, 256, 96);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    unsigned long long* h = (unsigned long long*)malloc(16 * blocks);
    (void)hipMemcpy(h, g_st, 16 * blocks, hipMemcpyDeviceToHost);
    double clk = 0;
    for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
    clk /= blocks;
    double flops = (double)blocks * 8 * nblk * 9 * 32.0 * 2048.0 * mt_scale;
    printf("%-44s %7.3f ms  %6.1f TFLOP/s  clock %.0f MHz  pipe busy %.1f %%\n", name, ms, flops / (ms * 1e-3) / 1e12, clk,
           100.0 * (flops / (ms * 1e-3)) / (256.0 * 4 * 64 * clk * 1e6));
    free(h);
}

int main() {
    const int blocks = 768 * 4, nblk = 12;  // 4 rounds of 3 workgroups per CU, 108 steps each
    size_t nact = (size_t)256 * 256 * 96, nw = (size_t)(nblk * 9 + 12) * 3 * 4 * 256;
    (void)hipMalloc(&g_out, sizeof(float) * 512 * blocks);
    (void)hipMalloc(&g_st, 16 * blocks);
    (void)hipMalloc(&g_act, nact * 4);
    (void)hipMalloc(&g_w, nw * 4);
    float* h = (float*)malloc((nact > nw ? nact : nw) * 4);
    srand(1);
    for (size_t i = 0; i < nact; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(g_act, h, nact * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(g_w, h, nw * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<F_BAR>("reads + barrier", blocks, nblk);
        run<F_BAR | F_ADDR>("+ real A addressing", blocks, nblk);
        run<F_BAR | F_WDMA>("+ weight DMA (fixed A addr)", blocks, nblk);
        run<F_BAR | F_ADDR | F_WDMA>("+ real A addressing + weight DMA", blocks, nblk);
        run<F_BAR | F_ADDR | F_WDMA | F_PDMA>("+ patch DMA (= k_conv16 K loop)", blocks, nblk);
        run<F_BAR | F_ADDR | F_WDMA | F_PDMA | F_PRIO>("+ setprio(1) around the MFMAs", blocks, nblk);
        run<F_BAR | F_PLANAR | F_WDMA | F_PDMA>("quad-planar patch image (offset-only A addr)", blocks, nblk);
        run<F_BAR | F_PLANAR | F_WDMA | F_PDMA | F_PREF>("quad-planar + fragment prefetch", blocks, nblk);
        run<F_BAR | F_PLANAR | F_WDMA | F_PDMA | F_W3>("quad-planar + weights 2 steps ahead (3 bufs)", blocks, nblk);
        run<0>("3 weight buffers (compile-time), 2 steps ahead", blocks, nblk, k6);
        run<0>("4 WG/CU (<=64 VGPR), single patch buffer", 1024 * 3, nblk, k8);
        run<0>("4x4 tiles per wave, 2 WG/CU, single patch buf", blocks / 2, nblk, k7<0>, 2.0);
        run<0>("4x4 tiles per wave, 1 WG/CU, double patch buf", blocks / 2, nblk, k7<1>, 2.0);
        run<0>("up phase: 16 KB weights / step (k_conv16)", blocks, nblk, k4<2>);
        run<0>("up phase:  8 KB weights / step", blocks, nblk, k4<1>);
        run<0>("up phase: no weight DMA", blocks, nblk, k4<0>);
        run<0>("row steps (96 MFMA/barrier), no patch DMA", blocks, nblk, k3<0>);
        run<F_PDMA>("row steps + single-buffer patch DMA", blocks, nblk, k3<F_PDMA>);
    }
    return 0;
}
