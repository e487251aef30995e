// Do VALU instructions of one wave run beside the MFMAs of another wave of the same SIMD?
// Workgroup = 8 waves (2 per SIMD): waves 0-3 run a dependent-free MFMA stream, waves 4-7 run NV VALU (or SALU)
// instructions per MFMA of their SIMD-mate.  hipcc --offload-arch=gfx950 -O3 coexec.hip -o coexec && ./coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>   // KIND 0: v_fma_f32, 1: s_add_u32, 2: v_mov (DPP-free), 3: ds_read_b32
__global__ __launch_bounds__(512) void k(float* out, int n) {
    __shared__ float lds[1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    if (wv < 4) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const float x = 1.0f + lane, y = 0.5f;
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (NV > 0) {
        float v0 = lane, v1 = lane + 1.f, v2 = lane + 2.f, v3 = lane + 3.f;
        unsigned s0 = blockIdx.x, s1 = 1;
        unsigned addr = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)lds + 4 * lane;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) {
                    asm volatile("v_fma_f32 %0, %0, %0, %1\n\tv_fma_f32 %1, %1, %1, %2\n\tv_fma_f32 %2, %2, %2, %3\n\tv_fma_f32 %3, %3, %3, %0"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                } else if (KIND == 1) {
                    asm volatile("s_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0" : "+s"(s0), "+s"(s1));
                } else if (KIND == 2) {
                    asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                } else {
                    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %4\n\tds_read_b32 %3, %4\n\ts_waitcnt lgkmcnt(0)"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
                }
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + s0 + s1;
    }
}

template <int NV, int KIND>
static void run(float* d, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(512), 0, 0, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfma_ms = (double)n * 4 * 32 / 2.4e6;   // one MFMA wave per SIMD: its pipe time
    static const char* names[] = {"v_fma_f32", "s_add_u32", "v_mov_b32", "ds_read_b32"};
    printf("%2d x %-11s per MFMA beside it: %.3f ms for %.3f ms of MFMA (x%.2f)\n", NV, names[KIND], ms, mfma_ms, ms / mfma_ms);
    fflush(stdout);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const int n = 20000;
    run<0, 0>(d, n);
    run<1, 0>(d, n);
    run<2, 0>(d, n);
    run<4, 0>(d, n);
    run<8, 0>(d, n);
    run<2, 2>(d, n);
    run<8, 2>(d, n);
    run<2, 1>(d, n);
    run<8, 1>(d, n);
    run<1, 3>(d, n);
    run<2, 3>(d, n);
    return 0;
}
