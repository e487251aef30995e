#!/usr/bin/env python3
"""Round 5 go/no-go, numerical half: float32 error of Winograd F(4x4, 3x3) against a float64
convolution, beside F(2x2, 3x3) (= arithmetic contract TZ-PA2) and the direct form (TZ-PA1), on the PredNet gate shapes.
Every variant accumulates along the input channels in float32 in ascending order (what an MFMA k-loop does); transforms
in float32 with the standard matrices (Lavin & Gray 2016; F(4x4): interpolation points 0, +-1, +-2, inf).  CPU only.

  python scripts/wino_f4_error.py  > profiles/r05/f4_error.txt
"""
import numpy as np

F32 = np.float32
# F(2x2, 3x3)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
# F(3x3, 3x3) (interpolation points 0, +-1, 2, inf): 25 multiplies per 9 outputs
BT3 = np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], np.float64)
G3 = np.array([[1 / 2, 0, 0], [-1 / 2, -1 / 2, -1 / 2], [-1 / 6, 1 / 6, -1 / 6], [1 / 6, 1 / 3, 2 / 3], [0, 0, 1]], np.float64)
AT3 = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]], np.float64)
# F(4x4, 3x3)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
               [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def chain32(V, U):
    """sum over channels in float32, ascending, fused multiply-add emulated as float64 product rounded once per step
    (fmaf on float32 inputs: the float64 product is exact, the add rounds once)."""
    acc = np.zeros(V.shape[:-1] + (U.shape[-1],), F32)
    for c in range(V.shape[-1]):
        acc = (acc.astype(np.float64) + V[..., c:c + 1].astype(np.float64) * U[c].astype(np.float64)).astype(F32)
    return acc


def wino(x, w, m, BT, G, AT):
    """x (H, W, C) float32, w (3,3,C,O) float32 -> (H, W, O) float32 via F(m x m, 3x3); H, W multiples of m."""
    H, W, C = x.shape
    O = w.shape[3]
    a = m + 2
    U = np.einsum("ik,klco,jl->ijco", G, w.astype(np.float64), G).astype(F32)          # weights transformed on the host (float64 -> f32)
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    out = np.zeros((H, W, O), F32)
    BTf, ATf = BT.astype(F32), AT.astype(F32)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[ty:ty + a, tx:tx + a]                                                    # (a, a, C)
            t = np.einsum("ik,klc->ilc", BTf, d).astype(F32)                                # float32 transform, rows then columns
            V = np.einsum("ilc,jl->ijc", t, BTf).astype(F32)
            D = np.stack([[chain32(V[i, j][None], U[i, j])[0] for j in range(a)] for i in range(a)])   # (a, a, O)
            z = np.einsum("ik,klo->ilo", ATf, D).astype(F32)
            out[ty:ty + m, tx:tx + m] = np.einsum("ilo,jl->ijo", z, ATf).astype(F32)
    return out


def direct32(x, w):
    H, W, C = x.shape
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    acc = np.zeros((H, W, w.shape[3]), F32)
    for ky in range(3):
        for kx in range(3):
            for c in range(C):
                acc = (acc.astype(np.float64) + xp[ky:ky + H, kx:kx + W, c:c + 1].astype(np.float64) * w[ky, kx, c].astype(np.float64)).astype(F32)
    return acc


def ref64(x, w):
    H, W, C = x.shape
    xp = np.pad(x.astype(np.float64), ((1, 1), (1, 1), (0, 0)))
    out = np.zeros((H, W, w.shape[3]))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("hwc,co->hwo", xp[ky:ky + H, kx:kx + W], w[ky, kx].astype(np.float64))
    return out


def main():
    print("float32 error (rms/max) against a float64 convolution, 12x12 outputs, 24 output columns, glorot-scaled weights")
    print("%-34s %15s %15s %15s %15s   %s" % ("source (channels, value model)", "direct", "F(2x2)", "F(3x3)", "F(4x4)",
                                                "F(3x3) / F(2x2) and F(4x4) / F(2x2)  [rms | max]"))
    rng = np.random.default_rng(5)
    for C, kind in ((96, "e: relu of normal"), (192, "e: relu of normal"), (384, "e: relu of normal"), (96, "uniform [0,1]"),
                    (192, "r: tanh-like in [-1,1]")):
        if kind.startswith("e"):
            x = np.maximum(rng.normal(0, 0.3, (12, 12, C)), 0).astype(F32)
        elif kind.startswith("uniform"):
            x = rng.random((12, 12, C)).astype(F32)
        else:
            x = np.tanh(rng.normal(0, 1, (12, 12, C))).astype(F32)
        lim = np.sqrt(6.0 / (9 * C + 9 * 48))
        w = rng.uniform(-lim, lim, (3, 3, C, 24)).astype(F32)
        ref = ref64(x, w)
        errs = []
        for got in (direct32(x, w), wino(x, w, 2, BT2, G2, AT2), wino(x, w, 3, BT3, G3, AT3), wino(x, w, 4, BT4, G4, AT4)):
            e = np.abs(got.astype(np.float64) - ref)
            errs.append((np.sqrt((e ** 2).mean()), e.max()))
        print("%-34s %15s %15s %15s %15s   %.1f | %.1f   %.1f | %.1f" % ("%d, %s" % (C, kind), *("%.1e/%.1e" % e for e in errs),
              errs[2][0] / errs[1][0], errs[2][1] / errs[1][1], errs[3][0] / errs[1][0], errs[3][1] / errs[1][1]))


if __name__ == "__main__":
    main()
