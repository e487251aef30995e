"""Seeded synthetic frame stacks for the BASELINE.json configs: the reference
ships no data and there is no network, so benchmarks and tests use these 8-bit stacks."""
import numpy as np


def moving_blobs(nt=40, h=64, w=64, seed=1):
    """cfg1: moving-MNIST-like: two bright blobs bouncing on black; returned as RGB (the
    reference converts 'L' to RGB, compress.py:114)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(14, min(h, w) - 14, (2, 2))
    vel = rng.uniform(-2.5, 2.5, (2, 2))
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out = np.zeros((nt, h, w), np.float64)
    for t in range(nt):
        for b in range(2):
            out[t] += 255 * np.exp(-((yy - pos[b, 0]) ** 2 + (xx - pos[b, 1]) ** 2) / (2 * 5.0 ** 2))
            pos[b] += vel[b]
            for d, lim in ((0, h), (1, w)):
                if pos[b, d] < 8 or pos[b, d] > lim - 8:
                    vel[b, d] = -vel[b, d]
    g = np.clip(out, 0, 255).astype(np.uint8)
    return np.repeat(g[..., None], 3, axis=-1)


def translating_scene(nt=40, h=128, w=160, seed=2):
    """cfg2: KITTI-like: smooth textured field shifted 2 px/frame plus sensor noise (sigma 2)."""
    rng = np.random.default_rng(seed)
    big = rng.normal(0, 1, (h + 16, w + 2 * nt + 16, 3))
    k = np.hanning(15)
    k /= k.sum()
    for ax in (0, 1):
        big = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, big)
    big = (big - big.min()) / (big.max() - big.min()) * 255
    frames = [big[8:8 + h, 8 + 2 * t: 8 + 2 * t + w] + rng.normal(0, 2, (h, w, 3)) for t in range(nt)]
    return np.clip(np.round(np.stack(frames)), 0, 255).astype(np.uint8)


def turbulence(nt=80, h=512, w=512, seed=3):
    """cfg3/cfg5: sum of 6 advected sinusoid octaves with random phases per channel."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    out = np.zeros((nt, h, w, 3), np.float32)
    ts = np.arange(nt, dtype=np.float32)
    for c in range(3):
        for o in range(6):
            f = (2.0 ** o) * 2 * np.pi / max(h, w)
            th = rng.uniform(0, 2 * np.pi)
            ph = rng.uniform(0, 2 * np.pi)
            vx, vy = rng.uniform(-1.5, 1.5, 2)
            amp = np.float32(1.0 / (1.5 ** o))
            kx, ky = np.float32(f * np.cos(th)), np.float32(f * np.sin(th))
            base = kx * xx + ky * yy + np.float32(ph)
            sb, cb = amp * np.sin(base), amp * np.cos(base)
            om = np.float32(kx * vx + ky * vy) * ts  # sin(base - om t) = sb cos(om t) - cb sin(om t)
            out[..., c] += sb[None] * np.cos(om)[:, None, None] - cb[None] * np.sin(om)[:, None, None]
    out = (out - out.min()) / (out.max() - out.min()) * 255
    return np.clip(np.round(out), 0, 255).astype(np.uint8)


def detector(nt=320, h=1024, w=1024, seed=4):
    """cfg4: XFEL-style detector frames: Poisson background + drifting Gaussian peaks, as RGB."""
    rng = np.random.default_rng(seed)
    npk = 50
    pos = rng.uniform(0, [h, w], (npk, 2))
    vel = rng.normal(0, 0.3, (npk, 2))
    amp = rng.uniform(40, 220, npk)
    sig = rng.uniform(1.5, 4.0, npk)
    out = np.empty((nt, h, w), np.uint8)
    for t in range(nt):
        img = rng.poisson(3.0, (h, w)).astype(np.float32)
        for k in range(npk):
            y0, x0 = pos[k] + vel[k] * t
            r = int(4 * sig[k]) + 1
            ys = slice(max(0, int(y0) - r), min(h, int(y0) + r + 1))
            xs = slice(max(0, int(x0) - r), min(w, int(x0) + r + 1))
            yy, xx = np.meshgrid(np.arange(ys.start, ys.stop), np.arange(xs.start, xs.stop), indexing="ij")
            img[ys, xs] += amp[k] * np.exp(-((yy - y0) ** 2 + (xx - x0) ** 2) / (2 * sig[k] ** 2))
        out[t] = np.clip(img, 0, 255).astype(np.uint8)
    return np.repeat(out[..., None], 3, axis=-1)
