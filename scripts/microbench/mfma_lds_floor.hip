// Ceiling of an fp32-MFMA implicit-GEMM wave tile on this device, without any global traffic:
//   MODE 0  v_mfma_f32_16x16x4_f32 from registers only
//   MODE 1  + A/B fragments re-read from LDS per k-step as 6 x ds_read_b32 per 8 MFMAs (k_conv3x3 v3)
//   MODE 2  + fragments read as ds_read_b128 (A: 4 k-steps per read, B: 4 column tiles per read)
//   MODE 3  A fragments as Winograd input-transform elements formed on the fly (4 x ds_read_b32 + 3 adds each)
// each with or without one workgroup barrier per 32 MFMAs.  512-thread workgroups, 2 x 4 tiles per
// wave, 3 workgroups per CU -- the shape of the conv kernel.  Prints TFLOP/s and the in-kernel clock
// (s_memtime / s_memrealtime), so that the DVFS clock under a sustained fp32-MFMA load is known.
//   hipcc --offload-arch=gfx950 -O3 mfma_lds_floor.hip -o mfma_lds_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool BAR>
__global__ __launch_bounds__(512, MODE == 4 ? 2 : (MODE == 5 ? 3 : 6)) void k(float* out, unsigned long long* stamps, int steps, const float* seed) {
    __shared__ __attribute__((aligned(16))) float sA[18 * 18 * 20];
    __shared__ __attribute__((aligned(16))) float sB[2 * 16 * 80];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 18 * 18 * 20; i += 512) sA[i] = seed[i & 4095];
    for (int i = tid; i < 2 * 16 * 80; i += 512) sB[i] = seed[(i * 7) & 4095];
    __syncthreads();
    f32x4 acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 y[MODE == 4 ? 4 : 1][2][4];
    for (int q = 0; q < (MODE == 4 ? 4 : 1); ++q)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 4; ++j) y[q][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 d5[4][4], y5[MODE == 5 ? 16 : 1];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) d5[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < (MODE == 5 ? 16 : 1); ++q) {
        y5[q] = (f32x4){seed[tid + q], 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(y5[q]));
    }
    float ra = seed[tid], rb = seed[tid + 512];
    const int arow0 = ((2 * wv) * 18 + (lane & 15)) * 20, arow1 = ((2 * wv + 1) * 18 + (lane & 15)) * 20;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int cur = 0;
    for (int st = 0; st < steps; ++st) {
        const int tap = st % 9, toff = ((tap / 3) * 18 + tap % 3) * 20;
        if (MODE == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra, rb, acc[i][j], 0, 0, 0);
        } else if (MODE == 1) {
            const float* pa = sA + toff + (lane >> 4);
            const float* pb = sB + cur * 16 * 80 + (lane >> 4) * 80 + (lane & 15);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float fa[2] = {pa[arow0 + 4 * kk], pa[arow1 + 4 * kk]};
                float fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = pb[4 * kk * 80 + j * 16];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else if (MODE == 3) {
            // Winograd F(2x2, 3x3) in the transformed domain, position xi outermost: the A fragment of a k-step is one element
            // of B^T d B = a signed sum of FOUR patch values (rows / columns xi selects), formed on the fly: 4 LDS reads and
            // 3 vector adds per A fragment instead of 1 read; B fragments as one ds_read_b128 per k-step (the weights of xi).
            // An MFMA of this loop stands for 2.25 MFMAs of the direct form (16 instead of 36 multiplies per 2x2 outputs).
            const int xi = st % 16, r0 = (xi >> 2), c0 = (xi & 3);
            const int o00 = (r0 * 18 + c0) * 20, o02 = o00 + 2 * 20, o20 = o00 + 2 * 18 * 20, o22 = o20 + 2 * 20;
            const float* pa = sA + (lane >> 4);
            const float* pb = sB + cur * 16 * 80 + lane * 4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float fa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* q = pa + (i ? arow1 : arow0) + 4 * kk;
                    const float t0 = q[o00] - q[o02];
                    const float t1 = q[o20] - q[o22];
                    fa[i] = t0 - t1;
                }
                const f32x4 b = *(const f32x4*)(pb + kk * 256);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if (MODE == 4) {
            // MODE 3 with the channel BLOCK outermost (the patch of a block is staged once, as in k_conv16): every step is one
            // transform position of one 16-channel block -- D = 4 k-steps from a zero accumulator, then D is folded into the 2x2
            // output accumulators Y (A^T D A: a corner position feeds 1 of the 4 outputs, an edge 2, a centre position 4 --
            // 36 signed adds per 16 positions = 2.25 vector adds per D register and step).
            const int xi = st % 16, r0 = (xi >> 2), c0 = (xi & 3);
            const int o00 = (r0 * 18 + c0) * 20, o02 = o00 + 2 * 20, o20 = o00 + 2 * 18 * 20, o22 = o20 + 2 * 20;
            const float* pa = sA + (lane >> 4);
            const float* pb = sB + cur * 16 * 80 + lane * 4;
            f32x4 d[2][4];
            const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float fa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* q = pa + (i ? arow1 : arow0) + 4 * kk;
                    const float t0 = q[o00] - q[o02];
                    const float t1 = q[o20] - q[o22];
                    fa[i] = t0 - t1;
                }
                const f32x4 b = *(const f32x4*)(pb + kk * 256);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], b[j], kk ? d[i][j] : zero, 0, 0, 0);
            }
            // fold: the number of outputs this position feeds (1, 2 or 4), uniform per step
            const int nr = (r0 == 0 || r0 == 3) ? 1 : 2, nc = (c0 == 0 || c0 == 3) ? 1 : 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    y[0][i][j] += d[i][j];
                    if (nc == 2) y[1][i][j] -= d[i][j];
                    if (nr == 2) {
                        y[2][i][j] += d[i][j];
                        if (nc == 2) y[3][i][j] -= d[i][j];
                    }
                }
        } else if (MODE == 5) {
            // The middle road: positions in four groups of four (one transform ROW per pass over the channels).  Four accumulator
            // sets d[pos] stay live (ONE 16-row tile per wave x 4 column tiles), the 64 output registers y[] are only live, not
            // touched; per k-step the four A fragments of the row come from 8 reads + 12 adds (column transform shared).
            const int r0 = st & 3;
            const float* pa = sA + (lane >> 4) + arow0 + (r0 * 18) * 20;
            const float* pb = sB + cur * 16 * 80 + lane * 4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float* q = pa + 4 * kk;
                const float a0 = q[0], a1 = q[20], a2 = q[40], a3 = q[60];
                const float c0 = q[2 * 18 * 20], c1 = q[2 * 18 * 20 + 20], c2 = q[2 * 18 * 20 + 40], c3 = q[2 * 18 * 20 + 60];
                const float ta0 = a0 - a2, ta1 = a1 + a2, ta2 = a2 - a1, ta3 = a1 - a3;
                const float tc0 = c0 - c2, tc1 = c1 + c2, tc2 = c2 - c1, tc3 = c1 - c3;
                const float fa[4] = {ta0 - tc0, ta1 - tc1, ta2 - tc2, ta3 - tc3};
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const f32x4 b = *(const f32x4*)(pb + ((pos * 4 + kk) & 7) * 256);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d5[pos][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[pos], b[j], d5[pos][j], 0, 0, 0);
                }
            }
        } else {
            // A image: pixel stride 20 floats, channel (4kk+g) at 4g+kk ; B image [kk][g][j][nt]
            const f32x4 a0 = *(const f32x4*)(sA + toff + arow0 + 4 * (lane >> 4));
            const f32x4 a1 = *(const f32x4*)(sA + toff + arow1 + 4 * (lane >> 4));
            const float* pb = sB + cur * 16 * 80 + lane * 4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 b = *(const f32x4*)(pb + kk * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kk], b[j], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kk], b[j], acc[1][j], 0, 0, 0);
                }
            }
        }
        if (BAR) __syncthreads();
        cur ^= 1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    for (int q = 0; q < (MODE == 4 ? 4 : 1); ++q)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 4; ++j) s += y[q][i][j][0] + y[q][i][j][1] + y[q][i][j][2] + y[q][i][j][3];
    for (int q = 0; q < (MODE == 5 ? 16 : 1); ++q) {
        asm volatile("" : "+v"(y5[q]));
        s += y5[q][0] + y5[q][1] + y5[q][2] + y5[q][3];
    }
    if (MODE == 5)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) s += d5[i][j][0] + d5[i][j][1] + d5[i][j][2] + d5[i][j][3];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE, bool BAR>
static void run(const char* name, float* out, unsigned long long* st, const float* seed, int blocks, int steps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, BAR>), dim3(blocks), dim3(512), 0, 0, out, st, steps, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE, BAR>), dim3(blocks), dim3(512), 0, 0, out, st, steps, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    unsigned long long* h = (unsigned long long*)malloc(16 * blocks);
    hipMemcpy(h, st, 16 * blocks, hipMemcpyDeviceToHost);
    double clk = 0;
    for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;  // MHz
    clk /= blocks;
    double flops = (double)blocks * 8 * steps * (MODE == 5 ? 64.0 : 32.0) * 2048.0;
    printf("%-28s %7.2f ms  %6.1f TFLOP/s  in-kernel clock %.0f MHz  -> MFMA pipe busy %.1f %%\n", name, ms,
           flops / (ms * 1e-3) / 1e12, clk, 100.0 * (flops / (ms * 1e-3)) / (256.0 * 4 * 64 * clk * 1e6));
    free(h);
}

int main() {
    const int blocks = 256 * 3 * 4, steps = 9 * 60;
    float *out, *seed;
    unsigned long long* st;
    hipMalloc(&out, sizeof(float) * 512 * blocks);
    hipMalloc(&st, 16 * blocks);
    hipMalloc(&seed, 4096 * 4);
    float h[4096];
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("regs only", out, st, seed, blocks, steps);
        run<0, true>("regs only + barrier/step", out, st, seed, blocks, steps);
        run<1, false>("ds_read_b32 x6 per 8 MFMA", out, st, seed, blocks, steps);
        run<1, true>("ds_read_b32 + barrier/step", out, st, seed, blocks, steps);
        run<2, false>("ds_read_b128", out, st, seed, blocks, steps);
        run<2, true>("ds_read_b128 + barrier/step", out, st, seed, blocks, steps);
        run<3, false>("winograd A: 4 reads + 3 adds", out, st, seed, blocks, steps);
        run<3, true>("winograd A + barrier/step", out, st, seed, blocks, steps);
        run<4, false>("winograd, block outermost + fold", out, st, seed, blocks, steps);
        run<4, true>("... + barrier/step", out, st, seed, blocks, steps);
        run<5, false>("winograd, 4 groups of 4 positions", out, st, seed, blocks, steps);
        run<5, true>("... + barrier/step", out, st, seed, blocks, steps);
    }
    return 0;
}
