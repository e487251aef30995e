"""Deterministic stand-in for the PredNet `predict` seam, shared by the golden
generator (where it is plugged under the reference's compress.run/decompress.run)
and by the tests (where it is plugged under the oracle's rollout).

It has the two properties of the real predictor that the reference's control flow
relies on: output t=0 is an input-independent constant image C0,
output t=k is a pure function g of input t=k-1.  All arithmetic is float32 and uses
only IEEE add/mul/clip so numpy 1.x and 2.x give identical bits.
"""
import numpy as np


def c0_image(hp, wp):
    """The constant t=0 'prediction' (plays the role of Ahat_0 at t0)."""
    yy, xx = np.meshgrid(np.arange(hp, dtype=np.float32), np.arange(wp, dtype=np.float32), indexing="ij")
    base = (np.float32(0.25) + np.float32(0.002) * yy + np.float32(0.001) * xx).astype(np.float32)
    out = np.stack([base, base * np.float32(0.5), base + np.float32(0.125)], axis=-1)
    return np.clip(out, np.float32(0), np.float32(1)).astype(np.float32)


def g_next(frame):
    """Next-frame function on one padded frame (Hp, Wp, 3) -> float32 (Hp, Wp, 3)."""
    x = np.asarray(frame).astype(np.float32)
    shifted = np.roll(x, 1, axis=1)
    y = np.float32(0.8125) * x + np.float32(0.125) * shifted + np.float32(0.03125)
    return np.clip(y, np.float32(0), np.float32(1)).astype(np.float32)


def predict(X):
    """Keras-like predict on (1, T, Hp, Wp, 3): out[:,0]=C0, out[:,t]=g(X[:,t-1])."""
    X = np.asarray(X)
    out = np.empty(X.shape, dtype=np.float32)
    out[0, 0] = c0_image(X.shape[2], X.shape[3])
    for t in range(1, X.shape[1]):
        out[0, t] = g_next(X[0, t - 1])
    return out
