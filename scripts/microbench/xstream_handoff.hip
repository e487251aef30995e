// Round 6 microbenchmark: what does a cross-stream hand-off cost between two dependent kernels?
//   (a) both kernels on ONE stream (no hand-off)
//   (b) hipEventRecord on stream 1 + hipStreamWaitEvent on stream 2, and back            (what "E-part ahead" uses)
//   (c) hipStreamWriteValue32 on stream 1 + hipStreamWaitValue32 on stream 2, and back   (stream memory operations, BETA)
//   (d) the events of (b) attached to the kernels themselves (hipExtLaunchKernelGGL's stopEvent: no marker packet behind the kernel)
//   (e) what an event RECORD between two kernels of ONE stream costs the second one (nobody waits for the event)
//   (f) the same with the event as the first kernel's stopEvent
// Each iteration = kernel A (s1) -> kernel B (s2, needs A) -> next A (s1, needs B).  Reported: microseconds per iteration
// minus twice the kernel's own duration = the two hand-offs.
// build: hipcc --offload-arch=gfx950 -O2 -o xstream_handoff scripts/microbench/xstream_handoff.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 0.5f;
    p[threadIdx.x] = v;
}

int main(int argc, char** argv) {
    const int N = 2000, iters = argc > 1 ? atoi(argv[1]) : 4000;
    float* d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d, stream priorities %d..%d\n", can, lo, hi);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    // kernel alone
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, d, iters);
    CK(hipStreamSynchronize(s1));
    auto t0 = now();
    for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, d, iters);
    CK(hipStreamSynchronize(s1));
    const double one = us(t0, now()) / (2 * N);
    printf("(a) one stream            : %.2f us per kernel (launch to launch)\n", one);
    // events
    hipEvent_t ea, eb;
    CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, d, iters);
            CK(hipEventRecord(ea, s1));
            CK(hipStreamWaitEvent(s2, ea, 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s2, d + 512, iters);
            CK(hipEventRecord(eb, s2));
            CK(hipStreamWaitEvent(s1, eb, 0));
        }
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        if (rep) printf("(b) events                : %.2f us per iteration = 2 kernels + %.2f us for the two hand-offs\n", us(t0, now()) / N, us(t0, now()) / N - 2 * one);
    }
    for (int rep = 0; rep < 2; ++rep) {   // (d)
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, nullptr, ea, 0, d, iters);
            CK(hipStreamWaitEvent(s2, ea, 0));
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s2, nullptr, eb, 0, d + 512, iters);
            CK(hipStreamWaitEvent(s1, eb, 0));
        }
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        if (rep) printf("(d) stopEvent of the launch: %.2f us per iteration = 2 kernels + %.2f us for the two hand-offs\n", us(t0, now()) / N, us(t0, now()) / N - 2 * one);
    }
    for (int rep = 0; rep < 2; ++rep) {   // (e)
        t0 = now();
        for (int i = 0; i < 2 * N; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, d, iters);
            CK(hipEventRecord(ea, s1));
        }
        CK(hipStreamSynchronize(s1));
        if (rep) printf("(e) record behind each kernel, one stream : %.2f us per kernel (+%.2f)\n", us(t0, now()) / (2 * N), us(t0, now()) / (2 * N) - one);
    }
    for (int rep = 0; rep < 2; ++rep) {   // (f)
        t0 = now();
        for (int i = 0; i < 2 * N; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, nullptr, ea, 0, d, iters);
        CK(hipStreamSynchronize(s1));
        if (rep) printf("(f) stopEvent on each kernel, one stream  : %.2f us per kernel (+%.2f)\n", us(t0, now()) / (2 * N), us(t0, now()) / (2 * N) - one);
    }
    if (can) {
        unsigned *sig = nullptr, *sig2 = nullptr;   // (signal memory is handed out in 8-byte objects)
        CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
        CK(hipExtMallocWithFlags((void**)&sig2, 8, hipMallocSignalMemory));
        CK(hipMemset(sig, 0, 8));
        CK(hipMemset(sig2, 0, 8));
        for (int rep = 0; rep < 2; ++rep) {
            unsigned base = rep * 2 * N;
            t0 = now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, d, iters);
                CK(hipStreamWriteValue32(s1, sig, base + 2 * i + 1, 0));
                CK(hipStreamWaitValue32(s2, sig, base + 2 * i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s2, d + 512, iters);
                CK(hipStreamWriteValue32(s2, sig2, base + 2 * i + 2, 0));
                CK(hipStreamWaitValue32(s1, sig2, base + 2 * i + 2, hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
            if (rep) printf("(c) stream memory ops     : %.2f us per iteration = 2 kernels + %.2f us for the two hand-offs\n", us(t0, now()) / N, us(t0, now()) / N - 2 * one);
        }
    }
    return 0;
}
