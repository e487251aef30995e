/* ORACLE-SIDE copy of the prediction arithmetic "TZ-PA1" scalar functions (test infrastructure).
 *
 * The product's device code has its own statement of the same functions in
 * tezip_amd/csrc/tz_math.hip.h; the two are kept textually independent on purpose (the
 * oracle must not be the thing shipped) and tests compare them bit for bit.
 *
 * Every operation is an IEEE-754 binary32 operation with round-to-nearest-even; a*b+c is
 * fused ONLY where fmaf() is written.  Compile with -ffp-contract=off.
 *
 * Semantics restated from the reference (Keras 2.2.4 semantics per SURVEY.md §8c):
 *   relu          prednet.py:201-205 (A_activation / error_activation)
 *   hard_sigmoid  prednet.py:80,198  clip(0.2*x + 0.5, 0, 1)   [Keras backend definition]
 *   tanh          prednet.py:80,198  (own polynomial: no libm dependency, deterministic)
 */
#ifndef TZ_ORACLE_MATH_H
#define TZ_ORACLE_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float tzo_relu(float x) { return x > 0.0f ? x : 0.0f; }

static inline float tzo_hard_sigmoid(float x) {
    float t = 0.2f * x;
    t = t + 0.5f;
    return t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
}

static inline float tzo_pow2i(int n) { /* 2^n for -126 <= n <= 127 */
    uint32_t b = (uint32_t)(n + 127) << 23;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

/* exp(x) for 0 <= x <= 20: Cody-Waite reduction + degree-5 polynomial (Cephes expf constants) */
static inline float tzo_exp_pos(float x) {
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
    return e * tzo_pow2i((int)fn);
}

static inline float tzo_tanh(float x) {
    float a = fabsf(x);
    float t;
    if (a >= 9.0f) {
        t = 1.0f;
    } else if (a >= 0.625f) {
        float e = tzo_exp_pos(a + a);
        float d = e + 1.0f;
        float q = 2.0f / d;
        t = 1.0f - q;
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
#endif
