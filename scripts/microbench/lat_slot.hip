// What does one k_convlat slot cost beyond its 4 dependent MFMAs (4 x 32 cycles)?  One workgroup of 4
// waves per CU; variants add the slot's other instructions one by one.
// hipcc --offload-arch=gfx950 -O3 lat_slot.hip -o lat_slot && ./lat_slot
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ f32x4 lds_read16_opaque(const float* p) {
    const unsigned addr = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait(f32x4& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }

// V: bit0 = DMA of a weight slot per slot, bit1 = vmcnt(13) wait, bit2 = ring read (ds_read_b128) prefetched one slot
// ahead, bit3 = four ds_read_b32 of A prefetched one slot ahead
template <int V>
__global__ __launch_bounds__(256) void slots(const float* __restrict__ w, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float ring[4 * 16 * 256];
    __shared__ float patch[4096];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) patch[i] = 1.0f;
    for (int i = threadIdx.x; i < 4 * 16 * 256; i += 256) ring[i] = 0.5f;
    __syncthreads();
    float* wr = ring + wv * 16 * 256;
    const float* wp = w + ((size_t)blockIdx.x * 4 + wv) * 256 + lane * 4;
    const size_t wstride = (size_t)gridDim.x * 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 wcur = {0.5f, 0.5f, 0.5f, 0.5f};
    float fa[4] = {1.f, 1.f, 1.f, 1.f};
    if (V & 1)
        for (int j = 0; j < 15; ++j, wp += wstride) glds16(wp, wr + j * 256);
    for (int t = 0; t < n; ++t) {
        const int r = t & 15, rp = (r + 15) & 15, rn = (r + 1) & 15;
        f32x4 wn = wcur;
        float fan[4] = {fa[0], fa[1], fa[2], fa[3]};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], wcur[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (V & 2) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        if (V & 4) wn = lds_read16_opaque(wr + rn * 256 + lane * 4);
        if (V & 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) fan[k] = patch[((lane & 15) * 4 + (lane >> 4) + 192 * k + 4 * (t & 7)) & 4095];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], wcur[1], acc, 0, 0, 0);
        if (V & 1) {
            glds16(wp, wr + rp * 256);
            wp += wstride;
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], wcur[2], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], wcur[3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (V & 4) {
            lds_wait(wn);
            wcur = wn;
        }
        if (V & 8)
            for (int k = 0; k < 4; ++k) fa[k] = fan[k];
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int V>
static void run(const float* w, float* d, int nb, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(slots<V>, dim3(nb), dim3(256), 0, 0, w, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("blocks %3d variant %2d (dma %d vmcnt %d ringread %d patchreads %d): %.1f ns per slot = %.0f cycles\n", nb, V, V & 1,
           (V >> 1) & 1, (V >> 2) & 1, (V >> 3) & 1, ms * 1e6 / n, ms * 1e6 / n * 2.4);
    fflush(stdout);
}

int main() {
    const int nb = 96, n = 4000;
    float *w, *d;
    hipMalloc(&w, ((size_t)n + 16) * 256 * 1024 * 4 + (1 << 20));
    hipMemset(w, 0, ((size_t)n + 16) * 256 * 1024 * 4);
    hipMalloc(&d, 256 * 256 * 4);
    run<0>(w, d, nb, n);
    run<1>(w, d, nb, n);
    run<3>(w, d, nb, n);
    run<4>(w, d, nb, n);
    run<8>(w, d, nb, n);
    run<12>(w, d, nb, n);
    run<5>(w, d, nb, n);
    run<7>(w, d, nb, n);
    run<15>(w, d, nb, n);
    run<15>(w, d, 256, n);
    return 0;
}
