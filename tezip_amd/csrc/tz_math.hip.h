// Device statement of the prediction arithmetic "TZ-PA1" scalar functions.
//
// TZ-PA1 fixes the float32 operation sequence of the predictor so that encoder and decoder
// (any GPU, or a CPU) regenerate bit-identical predictions -- a hard requirement for
// lossless decoding (decompress.py:252-253 rebuilds frames from pred*255 - delta).
//   * every op is IEEE binary32, round-to-nearest-even; fused multiply-add only where fmaf is
//     written (build with -ffp-contract=off); division is correctly rounded
//     (-fhip-fp32-correctly-rounded-divide-sqrt, the hipcc default); denormals preserved.
//   * convolution sums are single fmaf chains in a fixed order (tz_prednet.hip), which is
//     exactly what v_mfma_f32_16x16x4_f32 computes along k.
// Reference semantics: prednet.py:79-81,198-205 (relu, tanh, Keras hard_sigmoid).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float tz_relu(float x) { return x > 0.0f ? x : 0.0f; }

// clip(0.2 x + 0.5, 0, 1) with the product rounded before the sum.  The clip is a median of three, which hipcc folds
// into the clamp modifier of the addition: 2 vector instructions instead of 6 (two compares, two selects).  The sum is
// never -0 (0.5 is positive), so the median returns what the comparisons would.
__device__ __forceinline__ float tz_hard_sigmoid(float x) {
    float t = 0.2f * x;
    t = t + 0.5f;
    return __builtin_amdgcn_fmed3f(t, 0.0f, 1.0f);
}

// exp(x), 0 <= x <= 20: n = floor(x*log2(e)+0.5), two-step Cody-Waite reduction, degree-5
// polynomial on the remainder, exact scaling by 2^n.
__device__ __forceinline__ float tz_exp_pos(float x) {
    float fn = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(fn, -0.693359375f, x);
    r = fmaf(fn, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = fmaf(p, r2, r);
    e = e + 1.0f;
    return __builtin_ldexpf(e, (int)fn);   // exact: e in [0.7, 1.42], 0 <= fn <= 29
}

// 1 - 2/d for d in [4, 2^27] with both operations correctly rounded, as the contract states them (q = 2/d; t = 1 - q):
// 2/d = 2 RN(1/d) exactly, so t = RN(1 - 2 RN(1/d)) is one fmaf on the correctly rounded reciprocal, and that is two
// Newton steps on v_rcp_f32 (1 ulp).  6 vector instructions instead of the 12 of an IEEE division (operand scaling and
// special-case fix-up included) plus a subtraction.  That the two steps round correctly for EVERY d in the range is
// checked exhaustively against the division on the device (tests/test_gpu_parity.py::test_activations_...).
__device__ __forceinline__ float tz_one_minus_two_over(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = fmaf(-d, r, 1.0f);
    r = fmaf(e, r, r);
    e = fmaf(-d, r, 1.0f);
    r = fmaf(e, r, r);
    return fmaf(r, -2.0f, 1.0f);
}

__device__ __forceinline__ float tz_tanh(float x) {
    float a = fabsf(x);
    float t;
    if (a >= 9.0f) {
        t = 1.0f;
    } else if (a >= 0.625f) {
        float e = tz_exp_pos(a + a);
        float d = e + 1.0f;
        t = tz_one_minus_two_over(d);   // q = 2 / d; t = 1 - q
    } else {
        float z = a * a;
        float p = -5.70498872745e-3f;
        p = fmaf(p, z, 2.06390887954e-2f);
        p = fmaf(p, z, -5.37397155531e-2f);
        p = fmaf(p, z, 1.33314422036e-1f);
        p = fmaf(p, z, -3.33332819422e-1f);
        float pz = p * z;
        t = fmaf(pz, a, a);
    }
    return copysignf(t, x);
}
