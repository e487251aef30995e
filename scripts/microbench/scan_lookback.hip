// What does a decoupled look-back cost on MI355X (8 XCDs, one L2 each)?  Variants of the mod-2^16
// prefix scan of tz_codec.hip (k_scan1) on 62.9 M int16, timed with HIP events, checked against the CPU.
//   V0 look-back, 256 threads poll with agent-scope loads        V1 same, only wave 0 polls (64 wide)
//   V2 no look-back at all (floor of the kernel structure; wrong result)
//   V3 wave 0 polls with atomic RMW (fetch_add 0)                V4 V1 + s_sleep between polls
//   V5 the three-pass form of rounds 1-2 (sums, scan of sums, apply)
//   V6 two-pass "reduce then scan" with G persistent blocks (second read should hit the Infinity Cache)
// build + run:  MB=scan_lookback bash scripts/gpu_floor.sh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef short short8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static constexpr unsigned AGG = 1u << 16, INC = 2u << 16;

__device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned rmw(unsigned* p) { return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ unsigned block_excl(unsigned v, unsigned* total) {
    __shared__ unsigned wsum[4];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned inc = v;
    for (int s = 1; s < 64; s <<= 1) {
        unsigned up = __shfl_up(inc, s);
        if (lane >= s) inc += up;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned pre = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wv) pre += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

template <int EPT>
__device__ __forceinline__ void load(const int16_t* in, size_t base, int* v) {
#pragma unroll
    for (int q = 0; q < EPT / 8; ++q) {
        short8 a = *(const short8*)(in + base + 8 * q);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[8 * q + k] = a[k];
    }
}
template <int EPT>
__device__ __forceinline__ void store(int16_t* out, size_t base, unsigned pre, const int* v) {
#pragma unroll
    for (int q = 0; q < EPT / 8; ++q) {
        short8 a;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pre += (unsigned)v[8 * q + k];
            a[k] = (short)(uint16_t)(0u - pre);
        }
        *(short8*)(out + base + 8 * q) = a;
    }
}

template <int V, int EPT>
__global__ __launch_bounds__(256) void k_scan(const int16_t* __restrict__ in, size_t n, unsigned* __restrict__ status,
                                              int16_t* __restrict__ out) {
    __shared__ unsigned s_bid, s_prefix, s_wsum[4];
    __shared__ int s_winc[4];
    if (threadIdx.x == 0) s_bid = atomicAdd(&status[0], 1u);
    __syncthreads();
    const unsigned bid = s_bid;
    unsigned* stw = status + 1;
    const size_t base = ((size_t)bid * 256 + threadIdx.x) * EPT;
    int v[EPT];
    load<EPT>(in, base, v);
    if (base == 0) v[0] = -v[0];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) s += (unsigned)v[k];
    unsigned tot;
    unsigned pre = block_excl(s, &tot);
    unsigned acc = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (V == 2) {
        if (threadIdx.x == 0) st(&stw[bid], INC | (tot & 0xFFFFu));
    } else if (V == 0) {
        if (bid == 0) {
            if (threadIdx.x == 0) st(&stw[0], INC | (tot & 0xFFFFu));
        } else {
            if (threadIdx.x == 0) st(&stw[bid], AGG | (tot & 0xFFFFu));
            int look = (int)bid - 1;
            for (;;) {
                const int b = look - (int)threadIdx.x;
                unsigned w = b >= 0 ? ld(&stw[b]) : INC;
                while (__any((w >> 16) == 0))
                    if ((w >> 16) == 0) w = ld(&stw[b]);
                const unsigned long long incl = __ballot((w >> 16) == 2);
                const int first = incl ? __ffsll((long long)incl) - 1 : 63;
                unsigned c = lane <= first ? (w & 0xFFFFu) : 0u;
                for (int sft = 32; sft >= 1; sft >>= 1) c += __shfl_down(c, sft);
                if (lane == 0) {
                    s_wsum[wv] = c;
                    s_winc[wv] = incl != 0ull;
                }
                __syncthreads();
                bool found = false;
                for (int k = 0; k < 4 && !found; ++k) {
                    acc += s_wsum[k];
                    found = s_winc[k] != 0;
                }
                __syncthreads();
                if (found) break;
                look -= 256;
            }
            if (threadIdx.x == 0) st(&stw[bid], INC | ((acc + tot) & 0xFFFFu));
        }
    } else {  // V 1, 3, 4: wave 0 looks back
        if (threadIdx.x < 64) {
            if (bid == 0) {
                if (lane == 0) st(&stw[0], INC | (tot & 0xFFFFu));
            } else {
                if (lane == 0) st(&stw[bid], AGG | (tot & 0xFFFFu));
                int look = (int)bid - 1;
                for (;;) {
                    const int b = look - lane;
                    unsigned w = b >= 0 ? (V == 3 ? rmw(&stw[b]) : ld(&stw[b])) : INC;
                    while (__any((w >> 16) == 0)) {
                        if (V == 4) __builtin_amdgcn_s_sleep(8);
                        if ((w >> 16) == 0) w = V == 3 ? rmw(&stw[b]) : ld(&stw[b]);
                    }
                    const unsigned long long incl = __ballot((w >> 16) == 2);
                    const int first = incl ? __ffsll((long long)incl) - 1 : 63;
                    unsigned c = lane <= first ? (w & 0xFFFFu) : 0u;
                    for (int sft = 32; sft >= 1; sft >>= 1) c += __shfl_down(c, sft);
                    acc += __shfl(c, 0);
                    if (incl) break;
                    look -= 64;
                }
                if (lane == 0) st(&stw[bid], INC | ((acc + tot) & 0xFFFFu));
            }
            if (lane == 0) s_prefix = acc;
        }
        __syncthreads();
        acc = s_prefix;
    }
    store<EPT>(out, base, pre + acc, v);
}

// ---- V5: three passes
template <int EPT>
__global__ __launch_bounds__(256) void k_sums(const int16_t* __restrict__ in, unsigned* __restrict__ bsum) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * EPT;
    int v[EPT];
    load<EPT>(in, base, v);
    if (base == 0) v[0] = -v[0];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) s += (unsigned)v[k];
    unsigned tot;
    block_excl(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void k_blocks(unsigned* __restrict__ bsum, int nb) {
    const int per = (nb + 255) / 256;
    const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
    unsigned s = 0;
    for (int b = b0; b < b1; ++b) s += bsum[b];
    unsigned tot;
    unsigned run = block_excl(s, &tot);
    for (int b = b0; b < b1; ++b) {
        unsigned v = bsum[b];
        bsum[b] = run;
        run += v;
    }
}
template <int EPT>
__global__ __launch_bounds__(256) void k_apply(const int16_t* __restrict__ in, const unsigned* __restrict__ bsum, int16_t* __restrict__ out) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * EPT;
    int v[EPT];
    load<EPT>(in, base, v);
    if (base == 0) v[0] = -v[0];
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) s += (unsigned)v[k];
    unsigned tot;
    unsigned pre = block_excl(s, &tot) + bsum[blockIdx.x];
    store<EPT>(out, base, pre, v);
}

// ---- V6: G persistent blocks, each owns a contiguous chunk of `tiles` tiles: reduce the chunk, publish its sum,
// read the sums of all chunks in front (they are all resident and do the same work: available together), re-read
// the chunk (Infinity Cache) and write
template <int EPT>
__global__ __launch_bounds__(256) void k_two_pass(const int16_t* __restrict__ in, int tiles, unsigned* __restrict__ status,
                                                  int16_t* __restrict__ out) {
    __shared__ unsigned s_red[256];
    const int g = blockIdx.x;
    const size_t chunk0 = (size_t)g * tiles * 256 * EPT;
    unsigned s = 0;
    for (int t = 0; t < tiles; ++t) {
        const size_t base = chunk0 + ((size_t)t * 256 + threadIdx.x) * EPT;
        int v[EPT];
        load<EPT>(in, base, v);
        if (base == 0) v[0] = -v[0];
#pragma unroll
        for (int k = 0; k < EPT; ++k) s += (unsigned)v[k];
    }
    unsigned tot;
    block_excl(s, &tot);
    if (threadIdx.x == 0) st(&status[g], AGG | (tot & 0xFFFFu));
    // sum of the chunks in front: thread t takes chunks t, t + 256, ...
    unsigned mine = 0;
    for (int b = threadIdx.x; b < g; b += 256) {
        unsigned w = ld(&status[b]);
        while ((w >> 16) == 0) w = ld(&status[b]);
        mine += w & 0xFFFFu;
    }
    unsigned all;
    block_excl(mine, &all);
    unsigned carry = all;
    for (int t = 0; t < tiles; ++t) {
        const size_t base = chunk0 + ((size_t)t * 256 + threadIdx.x) * EPT;
        int v[EPT];
        load<EPT>(in, base, v);
        if (base == 0) v[0] = -v[0];
        unsigned q = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) q += (unsigned)v[k];
        unsigned ttot;
        unsigned pre = block_excl(q, &ttot);
        store<EPT>(out, base, pre + carry, v);
        carry += ttot;
    }
}


// ---- V7 / V8 / V9: no ticket.  V9: tile = blockIdx.x (relies on in-order dispatch).  V7: G resident blocks walk the
// tiles blockIdx.x + k G (deadlock-free whatever the dispatch order: every predecessor tile belongs to a resident block).
// V8: V7 with the loads of the NEXT tile issued before the look-back of the current one.
template <int EPT>
__device__ __forceinline__ unsigned lookback256(unsigned* stw, int tile, unsigned tot, unsigned* s_wsum, int* s_winc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned acc = 0;
    if (tile == 0) {
        if (threadIdx.x == 0) st(&stw[0], INC | (tot & 0xFFFFu));
        return 0;
    }
    if (threadIdx.x == 0) st(&stw[tile], AGG | (tot & 0xFFFFu));
    int look = tile - 1;
    for (;;) {
        const int b = look - (int)threadIdx.x;
        unsigned w = b >= 0 ? ld(&stw[b]) : INC;
        while (__any((w >> 16) == 0))
            if ((w >> 16) == 0) w = ld(&stw[b]);
        const unsigned long long incl = __ballot((w >> 16) == 2);
        const int first = incl ? __ffsll((long long)incl) - 1 : 63;
        unsigned c = lane <= first ? (w & 0xFFFFu) : 0u;
        for (int sft = 32; sft >= 1; sft >>= 1) c += __shfl_down(c, sft);
        if (lane == 0) {
            s_wsum[wv] = c;
            s_winc[wv] = incl != 0ull;
        }
        __syncthreads();
        bool found = false;
        for (int k = 0; k < 4 && !found; ++k) {
            acc += s_wsum[k];
            found = s_winc[k] != 0;
        }
        __syncthreads();
        if (found) break;
        look -= 256;
    }
    if (threadIdx.x == 0) st(&stw[tile], INC | ((acc + tot) & 0xFFFFu));
    return acc;
}

template <int V, int EPT>
__global__ __launch_bounds__(256) void k_scan_p(const int16_t* __restrict__ in, int ntiles, unsigned* __restrict__ status,
                                                int16_t* __restrict__ out) {
    __shared__ unsigned s_wsum[4];
    __shared__ int s_winc[4];
    unsigned* stw = status + 1;
    const int G = V == 9 ? ntiles : (int)gridDim.x;
    int tile = blockIdx.x;
    int v[EPT], nx[EPT];
    if (V == 8 && tile < ntiles) load<EPT>(in, ((size_t)tile * 256 + threadIdx.x) * EPT, nx);
    for (; tile < ntiles; tile += G) {
        const size_t base = ((size_t)tile * 256 + threadIdx.x) * EPT;
        if (V == 8) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) v[k] = nx[k];
            if (tile + G < ntiles) load<EPT>(in, ((size_t)(tile + G) * 256 + threadIdx.x) * EPT, nx);
        } else {
            load<EPT>(in, base, v);
        }
        if (base == 0) v[0] = -v[0];
        unsigned s = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) s += (unsigned)v[k];
        unsigned tot;
        unsigned pre = block_excl(s, &tot);
        const unsigned acc = lookback256<EPT>(stw, tile, tot, s_wsum, s_winc);
        store<EPT>(out, base, pre + acc, v);
        if (V == 9) break;
    }
}

static int g_G = 1024;
template <int V, int EPT>
static float run(const int16_t* d_in, size_t n, unsigned* d_status, int16_t* d_out, int reps) {
    const int nb = (int)(n / (256 * EPT));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(d_status, 0, sizeof(unsigned) * (nb + 1), 0));
        CK(hipEventRecord(a, 0));
        if (V == 5) {
            hipLaunchKernelGGL(k_sums<EPT>, dim3(nb), dim3(256), 0, 0, d_in, d_status);
            hipLaunchKernelGGL(k_blocks, dim3(1), dim3(256), 0, 0, d_status, nb);
            hipLaunchKernelGGL((k_apply<EPT>), dim3(nb), dim3(256), 0, 0, d_in, d_status, d_out);
        } else if (V == 6) {
            const int G = g_G, tiles = nb / G;  // n is chosen so that this divides
            hipLaunchKernelGGL(k_two_pass<EPT>, dim3(G), dim3(256), 0, 0, d_in, tiles, d_status, d_out);
        } else if (V == 9) {
            hipLaunchKernelGGL((k_scan_p<9, EPT>), dim3(nb), dim3(256), 0, 0, d_in, nb, d_status, d_out);
        } else if (V == 7 || V == 8) {
            int per_cu = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_p<V, EPT>, 256, 0));
            hipDeviceProp_t prop;
            CK(hipGetDeviceProperties(&prop, 0));
            const int G = per_cu * prop.multiProcessorCount;
            if (r == 0) printf("   [V%d EPT %d: %d resident blocks per CU x %d CUs]\n", V, EPT, per_cu, prop.multiProcessorCount);
            hipLaunchKernelGGL((k_scan_p<V, EPT>), dim3(G < nb ? G : nb), dim3(256), 0, 0, d_in, nb, d_status, d_out);
        } else {
            hipLaunchKernelGGL((k_scan<V, EPT>), dim3(nb), dim3(256), 0, 0, d_in, n, d_status, d_out);
        }
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t n = (size_t)80 * 512 * 512 * 3;  // 62,914,560 = 15360 tiles of 4096 = 1024 * 15 * 4096
    std::vector<int16_t> h(n), ref(n), got(n);
    unsigned x = 12345;
    for (size_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = (int16_t)((int)((x >> 16) % 41) - 20);
    }
    {
        uint16_t acc = 0;
        for (size_t i = 0; i < n; ++i) {
            acc = (uint16_t)(acc + (uint16_t)(i == 0 ? -h[i] : h[i]));
            ref[i] = (int16_t)(uint16_t)(0u - acc);
        }
    }
    int16_t *d_in, *d_out;
    unsigned* d_status;
    CK(hipMalloc(&d_in, n * 2));
    CK(hipMalloc(&d_out, n * 2));
    CK(hipMalloc(&d_status, sizeof(unsigned) * (n / 2048 + 2)));
    CK(hipMemcpy(d_in, h.data(), n * 2, hipMemcpyHostToDevice));
    auto check = [&](const char* name, float ms, bool expect_ok) {
        CK(hipMemcpy(got.data(), d_out, n * 2, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i];
        printf("%-44s %8.1f us  %6.2f TB/s algorithmic (4 B/el)  %s\n", name, ms * 1e3, 4.0 * n / (ms * 1e-3) / 1e12,
               bad == 0 ? "exact" : (expect_ok ? "WRONG" : "(not a scan)"));
        CK(hipMemset(d_out, 0, n * 2));
    };
    const int R = 8;
    check("V2 no look-back (floor), 16/thread", run<2, 16>(d_in, n, d_status, d_out, R), false);
    check("V2 no look-back (floor), 32/thread", run<2, 32>(d_in, n, d_status, d_out, R), false);
    check("V0 look-back 256 wide, 16/thread", run<0, 16>(d_in, n, d_status, d_out, R), true);
    check("V5 three passes, 16/thread", run<5, 16>(d_in, n, d_status, d_out, R), true);
    check("V5 three passes, 32/thread", run<5, 32>(d_in, n, d_status, d_out, R), true);
    check("V6 two passes, 1024 persistent blocks, 16", run<6, 16>(d_in, n, d_status, d_out, R), true);
    for (int G : {256, 512, 768, 1024, 1536, 2048}) {
        g_G = G;
        char nm[64];
        snprintf(nm, sizeof(nm), "V6 two passes, %d blocks, 16/thread", G);
        if ((n / 4096) % G == 0) check(nm, run<6, 16>(d_in, n, d_status, d_out, R), true);
        snprintf(nm, sizeof(nm), "V6 two passes, %d blocks, 32/thread", G);
        if ((n / 8192) % G == 0) check(nm, run<6, 32>(d_in, n, d_status, d_out, R), true);
        snprintf(nm, sizeof(nm), "V6 two passes, %d blocks, 8/thread", G);
        if ((n / 2048) % G == 0) check(nm, run<6, 8>(d_in, n, d_status, d_out, R), true);
    }
    return 0;
}
