// Skeleton of a Winograd F(2x2, 3x3) K loop in the "four groups of four positions" order (profiles/r03/
// winograd_skeleton.md), WITH its LDS-DMA streams: per step = (transform row i, 16-channel block) the workgroup stages
// the block's patch again and the four transformed weight blocks of the row, and every wave runs 4 k-steps x 4 positions
// x 4 column tiles = 64 MFMAs on ONE 16-row tile (16 tiles of 2x2 outputs), the four A fragments of a k-step formed from
// 8 LDS reads + 12 adds.  Variants: WG = 4 waves (16x16 pixels, 2 workgroups per CU) or 8 waves (16x32 pixels, 1 per CU);
// streams on / off.  Prints executed TFLOP/s; one executed FLOP stands for 2.25 of the direct form.
//   hipcc --offload-arch=gfx950 -O3 wino_skeleton.hip -o wino_skeleton.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// NW waves per workgroup; region 16 x (NW * 4) pixels = 8 x (NW * 2) tiles; patch 18 x (NW * 4 + 2) pixels x 16 channels
template <int NW, bool DMA>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 2) void k(float* out, int nsteps, const float* act, const float* wimg, int W, int C) {
    constexpr int PWX = NW * 4 + 2, PIX = 18 * PWX, NP = (PIX + 15) / 16 * 16, PPIECES = NP * 4 / 64;   // 1 KB pieces of a patch
    constexpr int WPIECES = 16;                                                                       // 4 positions x 4 KB
    constexpr int BUF = (PPIECES + WPIECES) * 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, r = lane & 15;
    for (int i = tid; i < 2 * BUF; i += NW * 64) smem[i] = act[i & 4095];
    __syncthreads();
    f32x4 d[4][4], y[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) d[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 16; ++q) {
        y[q] = (f32x4){act[tid + q], 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(y[q]));
    }
    // this wave's 16 tiles: tile rows 0..7, tile columns 2 wv, 2 wv + 1 -> lane r: (ty = r & 7, tx = 2 wv + (r >> 3))
    const int ty = r & 7, tx = 2 * wv + (r >> 3);
    const int abase = ((2 * ty) * PWX + 2 * tx) * 4 + g;
    const int tile = blockIdx.x % 256, ty0 = (tile / 16) * 16, tx0 = (tile % 16) * 16;
    const int qsrc = 4 * (lane & 3);
    int cur = 0;
#pragma unroll 1
    for (int st = 0; st < nsteps; ++st) {
        if (DMA) {   // next step's patch and weights into the other buffer
            float* dst = smem + (cur ^ 1) * BUF;
            for (int piece = wv; piece < PPIECES; piece += NW) {
                const int slot = piece * 16 + (lane >> 2);
                const int py = slot / PWX, px = slot - py * PWX;
                const int yy = ty0 - 1 + py, xx = tx0 - 1 + px;
                const bool ok = slot < PIX && yy >= 0 && yy < W && xx >= 0 && xx < W;
                glds16(ok ? act + ((long long)yy * W + xx) * C + ((st + 1) % (C / 16)) * 16 + qsrc : act, dst + piece * 256);
            }
            for (int piece = wv; piece < WPIECES; piece += NW)
                glds16(wimg + ((long long)(st + 1) * WPIECES + piece) * 256 + lane * 4, dst + (PPIECES + piece) * 256);
        }
        const float* pa = smem + cur * BUF + abase;
        const float* pb = smem + cur * BUF + PPIECES * 256 + lane * 4;
        const int ro = (st & 1) ? PWX * 4 : 0;   // the pass picks two of the four patch rows
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float* q = pa + ro + kk * NP * 4;
            const float a0 = q[0], a1 = q[4], a2 = q[8], a3 = q[12];
            const float* q2 = q + 2 * PWX * 4;
            const float c0 = q2[0], c1 = q2[4], c2 = q2[8], c3 = q2[12];
            const float ta0 = a0 - a2, ta1 = a1 + a2, ta2 = a2 - a1, ta3 = a1 - a3;
            const float tc0 = c0 - c2, tc1 = c1 + c2, tc2 = c2 - c1, tc3 = c1 - c3;
            const float fa[4] = {ta0 - tc0, ta1 - tc1, ta2 - tc2, ta3 - tc3};
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const f32x4 b = *(const f32x4*)(pb + (pos * 4 + kk) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[pos][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[pos], b[j], d[pos][j], 0, 0, 0);
            }
        }
        if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        __syncthreads();
        cur ^= 1;
    }
    float s = 0.f;
    for (int q = 0; q < 16; ++q) {
        asm volatile("" : "+v"(y[q]));
        s += y[q][0];
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += d[i][j][0] + d[i][j][1] + d[i][j][2] + d[i][j][3];
    out[(size_t)blockIdx.x * NW * 64 + tid] = s;
}

template <int NW, bool DMA>
static void run(const char* name, float* out, const float* act, const float* wimg, int blocks, int nsteps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NW, DMA>), dim3(blocks), dim3(NW * 64), 0, 0, out, nsteps, act, wimg, 256, 384);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<NW, DMA>), dim3(blocks), dim3(NW * 64), 0, 0, out, nsteps, act, wimg, 256, 384);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * NW * nsteps * 64.0 * 2048.0;
    printf("%-44s %7.2f ms  %6.1f TFLOP/s executed = %6.1f direct-equivalent (%s)\n", name, ms, flops / (ms * 1e-3) / 1e12,
           2.25 * flops / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int nsteps = 4 * 24;   // four passes over 24 channel blocks (384 channels: the top-level gates' live source)
    float *out, *act, *wimg;
    (void)hipMalloc(&out, sizeof(float) * 512 * 4096);
    (void)hipMalloc(&act, sizeof(float) * 256 * 256 * 384);
    (void)hipMalloc(&wimg, sizeof(float) * (size_t)(nsteps + 2) * 16 * 256);
    float* h = (float*)malloc(sizeof(float) * 256 * 256 * 384);
    srand(1);
    for (size_t i = 0; i < (size_t)256 * 256 * 384; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(act, h, sizeof(float) * 256 * 256 * 384, hipMemcpyHostToDevice);
    (void)hipMemcpy(wimg, h, sizeof(float) * (size_t)(nsteps + 2) * 16 * 256, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, false>("4 waves (16x16 px), 2 WG/CU, no streams", out, act, wimg, 256 * 2 * 6, nsteps);
        run<4, true>("4 waves, patch + weight LDS-DMA per step", out, act, wimg, 256 * 2 * 6, nsteps);
        run<8, false>("8 waves (16x32 px), 1-2 WG/CU, no streams", out, act, wimg, 256 * 1 * 6, nsteps);
        run<8, true>("8 waves, patch + weight LDS-DMA per step", out, act, wimg, 256 * 1 * 6, nsteps);
    }
    return 0;
}
