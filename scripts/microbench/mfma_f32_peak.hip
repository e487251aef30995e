// Sustained rate of v_mfma_f32_16x16x4_f32 from registers only (no memory traffic): the ceiling
// any fp32-MFMA kernel on this device can approach.  One wave per SIMD up to 8 waves per SIMD,
// 8 independent accumulators per wave.   hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0 + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, sizeof(float) * 256 * 256 * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
        int blocks = 256 * blocks_per_cu;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 0.5f, 0.25f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * (16 * 16 * 4 * 2);
        printf("waves/SIMD %d: %.1f TFLOP/s (%.2f ms)\n", blocks_per_cu, flops / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
