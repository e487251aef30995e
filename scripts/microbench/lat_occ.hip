// k_convlat's slot at several workgroups per CU: which of its instructions costs matrix-pipe time when 3-6 waves
// share a SIMD?  One slot = 4 k-steps x CH independent chains of dependent v_mfma_f32_16x16x4_f32.
// hipcc --offload-arch=gfx950 -O3 lat_occ.hip -o lat_occ && ./lat_occ
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
template <int O0, int O1>
__device__ __forceinline__ f32x2 rd2(unsigned addr) {
    f32x2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
__device__ __forceinline__ void lgkm0(f32x2& a, f32x2& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }

// LDSR: A values of the next slot read from LDS (two ds_read2st64_b32 per chain) | GLD: weights reloaded into a 9-deep
// register ring, one kilobyte per wave and slot: 1 = global_load_dwordx4 with a 64-bit VGPR address (what hipcc emits),
// 2 = scalar base + 32-bit lane offset, 3 = as two dwordx2, 4 = LDS-DMA of the kilobyte instead | BAR: a workgroup
// barrier every 9 slots
template <int CH, bool LDSR, int GLD, bool BAR>
__global__ __launch_bounds__(256) void slots(const float* __restrict__ w, float* out, int nblk) {
    __shared__ float patch[4096];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) patch[i] = 1.0f;
    __syncthreads();
    const float* wp = w + wv * 256 + lane * 4;
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 g[9];
#pragma unroll
    for (int j = 0; j < 9; ++j, wp += 1024) g[j] = GLD ? *(const f32x4*)wp : (f32x4){0.5f, 0.5f, 0.5f, 0.5f};
    const unsigned pa = lds_addr(patch) + 4 * ((lane & 15) * 4 + (lane >> 4));
    const unsigned voff = wv * 1024 + lane * 16;
    unsigned long long sbase = (unsigned long long)w + 9 * 4096;
    asm volatile("" : "+s"(sbase));
    __shared__ __attribute__((aligned(16))) float dma[4 * 256];
    const unsigned dma_dst = __builtin_amdgcn_readfirstlane(lds_addr(dma) + wv * 1024);
    f32x2 fa[CH][2];
    for (int c = 0; c < CH; ++c) fa[c][0] = fa[c][1] = (f32x2){1.f, 1.f};
    for (int b = 0; b < nblk; ++b) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (GLD >= 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(GLD == 3 ? 16 : 8) : "memory");
            const f32x4 wc = g[j];
            f32x2 fn[CH][2];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                fn[c][0] = fa[c][0];
                fn[c][1] = fa[c][1];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][0][0], wc[0], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LDSR) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const unsigned p = pa + 16 * ((j + c) & 7);
                    fn[c][0] = rd2<0, 3>(p);
                    fn[c][1] = rd2<6, 9>(p);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][0][1], wc[1], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][1][0], wc[2], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c][1][1], wc[3], acc[c], 0, 0, 0);
            if (GLD == 1) {
                g[j] = *(const f32x4*)wp;
                wp += 1024;
            } else if (GLD == 2) {   // scalar base + 32-bit lane offset: one address VGPR instead of two
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[j]) : "v"(voff), "s"(sbase) : "memory");
                sbase += 4096;
            } else if (GLD == 3) {   // the same bytes as two 8-byte loads
                f32x2 lo, hi;
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(lo) : "v"(voff), "s"(sbase) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, %2 offset:8" : "=v"(hi) : "v"(voff), "s"(sbase) : "memory");
                g[j] = (f32x4){lo[0], lo[1], hi[0], hi[1]};
                sbase += 4096;
            } else if (GLD == 4) {   // LDS-DMA of the same kilobyte (what k_conv16 does)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(wp), "s"(dma_dst) : "memory");
                wp += 1024;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (LDSR) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    lgkm0(fn[c][0], fn[c][1]);
                    fa[c][0] = fn[c][0];
                    fa[c][1] = fn[c][1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int j = 0; j < 9; ++j) s += g[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, bool LDSR, int GLD, bool BAR>
static void run(const float* w, float* d, int nb, int nblk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((slots<CH, LDSR, GLD, BAR>), dim3(nb), dim3(256), 0, 0, w, d, nblk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // pipe time of the launch: per SIMD (nb / 256) waves x nblk x 9 slots x 4 CH MFMAs x 32 cycles
    const double ideal_ms = (double)(nb / 256.0) * nblk * 9 * 4 * CH * 32 / 2.4e6;
    printf("wg/CU %4.1f chains %d lds-reads %d weight-loads(mode) %d barrier %d: %7.3f ms, matrix pipe %.2f busy\n", nb / 256.0, CH, LDSR, GLD,
           BAR, ms, ideal_ms / ms);
    fflush(stdout);
}

int main() {
    const int nblk = 200;
    float *w, *d;
    hipMalloc(&w, ((size_t)nblk * 9 + 32) * 4096);
    hipMemset(w, 0, ((size_t)nblk * 9 + 32) * 4096);
    hipMalloc(&d, 4096 * 256 * 4);
    for (int nb : {256, 1536}) {
        run<1, false, 0, false>(w, d, nb, nblk);
        run<1, true, 0, false>(w, d, nb, nblk);
        run<1, false, 1, false>(w, d, nb, nblk);
        run<1, false, 2, false>(w, d, nb, nblk);
        run<1, false, 3, false>(w, d, nb, nblk);
        run<1, false, 4, false>(w, d, nb, nblk);
        run<1, true, 1, true>(w, d, nb, nblk);
        run<2, false, 0, false>(w, d, nb, nblk);
        run<2, false, 1, false>(w, d, nb, nblk);
        run<2, false, 2, false>(w, d, nb, nblk);
        run<2, false, 4, false>(w, d, nb, nblk);
        run<2, true, 1, true>(w, d, nb, nblk);
    }
    return 0;
}
