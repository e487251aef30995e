// Follow-up of lat_slot.hip: weights shared by all workgroups (L2 hits, as in k_convlat), and the LDS operand
// reads issued either one or two slots ahead of their use, or the weights taken straight from global memory into
// a register ring.  hipcc --offload-arch=gfx950 -O3 lat_slot2.hip -o lat_slot2 && ./lat_slot2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ f32x4 rd16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ float rd4(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void lgkm(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

// MODE 0: MFMAs only; 1: + DMA ring (shared weights) with vmcnt(13); 2: 1 + operand reads one slot ahead;
// 3: 1 + operand reads two slots ahead; 4: operand A reads two slots ahead, weights by global_load_dwordx4 into an
// 8-deep register ring (no LDS weights at all); 5: as 4 but A reads one slot ahead; 6: as 5 with the four A values in
// one ds_read_b128 (a channel-permuted patch)
template <int MODE>
__global__ __launch_bounds__(256) void slots(const float* __restrict__ w, float* out, int n) {
    __shared__ __attribute__((aligned(16))) float ring[4 * 16 * 256];
    __shared__ float patch[4096];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) patch[i] = 1.0f;
    for (int i = threadIdx.x; i < 4 * 16 * 256; i += 256) ring[i] = 0.5f;
    __syncthreads();
    float* wr = ring + wv * 16 * 256;
    const float* wp = w + wv * 256 + lane * 4;
    constexpr size_t wstride = 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 w0 = {0.5f, 0.5f, 0.5f, 0.5f}, w1 = w0, w2 = w0;
    f32x4 a0 = {1.f, 1.f, 1.f, 1.f}, a1 = a0, a2 = a0;
    const unsigned pa = lds_addr(patch) + 4 * ((lane & 15) * 4 + (lane >> 4));
    const unsigned ra = lds_addr(wr) + 16 * lane;
    if (MODE >= 1 && MODE <= 3)
        for (int j = 0; j < 15; ++j, wp += wstride) glds16(wp, wr + j * 256);
    if (MODE >= 4) {
        f32x4 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j, wp += wstride) g[j] = *(const f32x4*)wp;
        for (int t = 0; t < n; t += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 wc = g[j];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], wc[0], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned p = pa + 16 * ((t + j) & 7);
                f32x4 an;
                if (MODE == 6) {
                    an = rd16(pa * 4 + 1024 * ((t + j) & 7));
                } else {
                    an[0] = rd4(p); an[1] = rd4(p + 768); an[2] = rd4(p + 1536); an[3] = rd4(p + 2304);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], wc[1], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], wc[2], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[3], wc[3], acc, 0, 0, 0);
                g[j] = *(const f32x4*)wp;  // reload this ring entry once its last use has issued
                wp += wstride;
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 4) {
                    lgkm<4>(a1, a1);
                    a0 = a1;
                    a1 = an;
                } else {
                    lgkm<0>(an, an);
                    a0 = an;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int t = 0; t < n; ++t) {
            const int r = t & 15, rp = (r + 15) & 15, rn = (r + (MODE == 3 ? 2 : 1)) & 15;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], w0[0], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 wn = w0, an = a0;
            if (MODE >= 1) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            if (MODE >= 2) {
                wn = rd16(ra + rn * 1024);
                const unsigned p = pa + 16 * (t & 7);
                an[0] = rd4(p); an[1] = rd4(p + 768); an[2] = rd4(p + 1536); an[3] = rd4(p + 2304);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], w0[1], acc, 0, 0, 0);
            if (MODE >= 1) {
                glds16(wp, wr + rp * 256);
                wp += wstride;
            }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], w0[2], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[3], w0[3], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) {
                lgkm<0>(wn, an);
                w0 = wn;
                a0 = an;
            } else if (MODE == 3) {
                lgkm<5>(w1, a1);  // the five reads issued in this slot may stay in flight
                w0 = w1; a0 = a1;
                w1 = wn; a1 = an;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + w1[0] + a1[0] + w2[0] + a2[0];
}

template <int MODE>
static void run(const float* w, float* d, int nb, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(slots<MODE>, dim3(nb), dim3(256), 0, 0, w, d, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("blocks %3d mode %d: %.1f ns per slot = %.0f cycles\n", nb, MODE, ms * 1e6 / n, ms * 1e6 / n * 2.4);
    fflush(stdout);
}

int main() {
    const int n = 4000;
    float *w, *d;
    hipMalloc(&w, ((size_t)n + 32) * 4096);
    hipMemset(w, 0, ((size_t)n + 32) * 4096);
    hipMalloc(&d, 1024 * 256 * 4);
    for (int nb : {96, 256}) {
        run<0>(w, d, nb, n);
        run<1>(w, d, nb, n);
        run<2>(w, d, nb, n);
        run<3>(w, d, nb, n);
        run<4>(w, d, nb, n);
        run<5>(w, d, nb, n);
        run<6>(w, d, nb, n);
    }
    return 0;
}
