// Round 5 go/no-go (second option): would the F(2x2) chains of k_wino run faster on
// v_mfma_f32_32x32x2_f32 than on v_mfma_f32_16x16x4_f32?  Both issue 64 FLOP per cycle and SIMD; the 32x32x2 form reads
// HALF the A/B operand registers per FLOP (and twice the accumulator registers).  k_wino's stage loops hold 2.13-2.20 GHz
// instead of the 2.4 GHz the guide's peak is quoted at (profiles/r04/winograd_product.md), so the question is whether the
// other shape holds a higher clock under the same kind of load: random-mantissa operands that CHANGE with every
// instruction (a register-only loop on constant operands runs at 155 TFLOP/s, mfma_f32_peak.hip -- data toggling is what
// costs), two waves per SIMD, 128 accumulation registers per wave as in k_wino.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape_clock.hip -o mfma_shape_clock && ./mfma_shape_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return __uint_as_float(0x3f000000u | (s >> 9)) - 0.75f;   // [-0.25, 0.25): accumulators stay finite
}

// 32 accumulators of 4 registers (= k_wino's 8 positions x 4 column tiles), A changes per position, B per (position, tile)
template <bool RANDOM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k16(float* out, int iters, unsigned long long* clk) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[8], b[8][4];
    for (int p = 0; p < 8; ++p) {
        a[p] = RANDOM ? rnd(s) : 0.125f;
        for (int t = 0; t < 4; ++t) b[p][t] = RANDOM ? rnd(s) : 0.0625f;
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[4 * p + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p], b[p][t], acc[4 * p + t], 0, 0, 0);
        if (RANDOM) {   // new operands for the next k-step: one cheap VALU op per operand register (k_wino: transform adds + LDS reads)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                a[p] = -a[p];
#pragma unroll
                for (int t = 0; t < 4; ++t) b[p][t] = __uint_as_float(__float_as_uint(b[p][t]) ^ 0x00155555u);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < 32; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x < 1024) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// the same FLOPs per iteration on 32x32x2: 8 accumulators of 16 registers (128 registers), k = 2 per instruction, so TWO
// instructions per accumulator make one k = 4 step: A and B change per instruction
template <bool RANDOM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k32(float* out, int iters, unsigned long long* clk) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a[8][2], b[8][2];
    for (int p = 0; p < 8; ++p)
        for (int h = 0; h < 2; ++h) {
            a[p][h] = RANDOM ? rnd(s) : 0.125f;
            b[p][h] = RANDOM ? rnd(s) : 0.0625f;
        }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][h], b[p][h], acc[p], 0, 0, 0);
        if (RANDOM) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    a[p][h] = -a[p][h];
                    b[p][h] = __uint_as_float(__float_as_uint(b[p][h]) ^ 0x00155555u);
                }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 16; ++j) sum += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x < 1024) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename K>
static void run(const char* name, K kern, double flops_per_wave_iter, float* d, unsigned long long* dclk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 4;   // one workgroup of 8 waves per CU at a time (2 waves per SIMD), 4 rounds
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, 200, dclk);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, iters, dclk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    static unsigned long long h[2048];
    hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < 1024; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double flops = (double)blocks * 8 * iters * flops_per_wave_iter;
    printf("%-44s %7.1f TFLOP/s  (%.2f ms)  core clock in the loop %.0f MHz\n", name, flops / (best * 1e-3) / 1e12, best, cyc / (real / 100.0));
}

int main() {
    float* d;
    unsigned long long* dclk;
    hipMalloc(&d, sizeof(float) * 1024 * 512);
    hipMalloc(&dclk, sizeof(unsigned long long) * 2048);
    const double f16 = 32.0 * (16 * 16 * 4 * 2), f32 = 16.0 * (32 * 32 * 2 * 2);   // FLOPs per wave and iteration: equal
    run("16x16x4, constant operands", k16<false>, f16, d, dclk);
    run("32x32x2, constant operands", k32<false>, f32, d, dclk);
    run("16x16x4, operands change every k-step", k16<true>, f16, d, dclk);
    run("32x32x2, operands change every k-step", k32<true>, f32, d, dclk);
    return 0;
}
