#!/usr/bin/env python3
"""Round 5 (the form passes this check and was dropped for its weight stream): the UPSAMPLED source of a gate convolution -- 3x3 taps on a x2 nearest
upsampled map = per output parity class a 2x2-tap filter on the half-resolution grid (the collapsed taps of TZ-PA1 / TZ-PA2)
-- evaluated as Winograd F(2x2, 2x2) per class: 9 multiplies per 2x2 outputs of a class instead of 16.  Float32 error of
both forms against a float64 evaluation, chains along the channels as an MFMA k-loop accumulates them.  CPU only.

  python scripts/wino_ups_error.py > profiles/r05/ups_winograd_error.txt
"""
import numpy as np

F32 = np.float32
BT = np.array([[1, -1, 0], [0, 1, 0], [0, -1, 1]], np.float64)     # F(2, 2): m1 = (d0 - d1) g0, m2 = d1 (g0 + g1), m3 = (d2 - d1) g1
G = np.array([[1, 0], [1, 1], [0, 1]], np.float64)
AT = np.array([[1, 1, 0], [0, 1, 1]], np.float64)


def chain32(V, U):
    acc = np.zeros(V.shape[:-1] + (U.shape[-1],), F32)
    for c in range(V.shape[-1]):
        acc = (acc.astype(np.float64) + V[..., c:c + 1].astype(np.float64) * U[c].astype(np.float64)).astype(F32)
    return acc


def collapsed(w):
    """3x3 taps (3,3,C,O) -> per parity class (a,b) the 2x2 filter on the half-resolution grid, float32 sums (ascending ky, kx)."""
    out = {}
    for a in (0, 1):
        for b in (0, 1):
            f = np.zeros((2, 2) + w.shape[2:], F32)
            for ky in range(3):
                for kx in range(3):
                    u, v = (a + ky - 1) // 2 - (a - 1), (b + kx - 1) // 2 - (b - 1)   # which of the class's 2x2 half-res pixels the tap lands on
                    f[u, v] = (f[u, v] + w[ky, kx]).astype(F32)
            out[(a, b)] = f
    return out


def main():
    rng = np.random.default_rng(9)
    print("float32 error (rms/max) of the upsampled source's contribution against float64, 8x8 half-resolution pixels, 24 columns")
    print("%-10s %18s %18s   %s" % ("channels", "collapsed taps", "F(2x2,2x2)", "ratio rms | max"))
    for C in (96, 192):
        r = np.tanh(rng.normal(0, 1, (10, 10, C))).astype(F32)          # half-resolution map with a halo of one
        lim = np.sqrt(6.0 / (9 * C + 9 * 48))
        w = rng.uniform(-lim, lim, (3, 3, C, 24)).astype(F32)
        fl = collapsed(w)
        e_dir, e_win = [], []
        for (a, b), f in fl.items():
            ref = np.zeros((8, 8, 24))
            direct = np.zeros((8, 8, 24), F32)
            for u in range(2):
                for v in range(2):
                    x = r[a + u: a + u + 8, b + v: b + v + 8]
                    ref += np.einsum("hwc,co->hwo", x.astype(np.float64), f[u, v].astype(np.float64))
            # direct: chain over (tap, channel) as the kernel does (taps inside a channel quad are reordered there; the error
            # statistics do not depend on it)
            taps = np.concatenate([r[a + u: a + u + 8, b + v: b + v + 8] for u in range(2) for v in range(2)], axis=-1)
            wt = np.concatenate([f[u, v] for u in range(2) for v in range(2)], axis=0)
            direct = chain32(taps, wt)
            U = np.einsum("ik,klco,jl->ijco", G, f.astype(np.float64), G).astype(F32)
            win = np.zeros((8, 8, 24), F32)
            BTf, ATf = BT.astype(F32), AT.astype(F32)
            for i in range(0, 8, 2):
                for j in range(0, 8, 2):
                    d = r[a + i: a + i + 3, b + j: b + j + 3]
                    V = np.einsum("ilc,jl->ijc", np.einsum("ik,klc->ilc", BTf, d).astype(F32), BTf).astype(F32)
                    D = np.stack([[chain32(V[p, q][None], U[p, q])[0] for q in range(3)] for p in range(3)])
                    win[i:i + 2, j:j + 2] = np.einsum("ilo,jl->ijo", np.einsum("ik,klo->ilo", ATf, D).astype(F32), ATf).astype(F32)
            e_dir.append(np.abs(direct - ref))
            e_win.append(np.abs(win - ref))
        ed, ew = np.stack(e_dir), np.stack(e_win)
        print("%-10d %8.1e/%8.1e %8.1e/%8.1e   %.2f | %.2f" % (C, np.sqrt((ed ** 2).mean()), ed.max(), np.sqrt((ew ** 2).mean()), ew.max(),
                                                             np.sqrt((ew ** 2).mean()) / np.sqrt((ed ** 2).mean()), ew.max() / ed.max()))


if __name__ == "__main__":
    main()
