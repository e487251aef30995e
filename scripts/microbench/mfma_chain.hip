// How fast does ONE wave per SIMD walk a chain of dependent v_mfma_f32_16x16x4_f32 (k_convlat's K loop)?
// hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LDSREADS>
__global__ __launch_bounds__(256) void chain(float* out, int n, float a, float b) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = a;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float x = a, y = b;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (LDSREADS) x = s[(threadIdx.x * 4 + i + k) & 4095];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 20000;  // x4 MFMAs
    for (int nb : {1, 96, 256}) {
        for (int v = 0; v < 2; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(chain<0>, dim3(nb), dim3(256), 0, 0, d, n, 1.0f, 2.0f);
                else hipLaunchKernelGGL(chain<1>, dim3(nb), dim3(256), 0, 0, d, n, 1.0f, 2.0f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("blocks %3d  lds_reads %d: %.2f ns per dependent MFMA (%.1f cycles at 2.4 GHz)\n", nb, v, ms * 1e6 / (4.0 * n), ms * 1e6 / (4.0 * n) * 2.4);
            }
        }
    }
    return 0;
}
