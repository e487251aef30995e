"""The oracle's SECOND statement of the 3x3 convolution -- Winograd F(2x2, 3x3) on the same-resolution sources, still one
fmaf chain along the input channels per transformed position (oracle/tz_oracle.c: conv3x3_wino; profiles/r03/
winograd_skeleton.md) -- arithmetic contract TZ-PA2 since round 4 (tzo_model_set_contract; the device kernel k_wino is bit-exact
with it: tests/test_gpu_wino.py).  Here it is held against the direct statement and a float64 convolution."""
import itertools

import numpy as np
import pytest

from oracle import coracle


def _ref64(x, xu, w, b):
    H, W = (x.shape[:2] if x is not None else (2 * xu.shape[0], 2 * xu.shape[1]))
    C = 0 if x is None else x.shape[2]
    ref = np.zeros((H, W, w.shape[3])) + b.astype(np.float64)
    for src, lo in ((x, 0), (None if xu is None else np.repeat(np.repeat(xu, 2, 0), 2, 1), C)):
        if src is None:
            continue
        xp = np.pad(src.astype(np.float64), ((1, 1), (1, 1), (0, 0)))
        for ky, kx in itertools.product(range(3), range(3)):
            ref += np.einsum("hwc,co->hwo", xp[ky:ky + H, kx:kx + W], w[ky, kx, lo:lo + src.shape[2]].astype(np.float64))
    return ref


@pytest.mark.parametrize("H,W,C,Cu,Co", [(8, 8, 16, 0, 8), (9, 13, 32, 0, 12), (1, 1, 16, 0, 3), (2, 7, 16, 0, 4), (16, 24, 48, 32, 20),
                                         (6, 10, 0, 16, 5), (32, 40, 96, 0, 48), (8, 8, 5, 0, 7)])
def test_winograd_statement_agrees_with_direct_and_float64(H, W, C, Cu, Co):
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.normal(0, 1, (H, W, C)).astype(np.float32) if C else None
    xu = rng.normal(0, 1, (H // 2, W // 2, Cu)).astype(np.float32) if Cu else None
    w = (rng.normal(0, 1, (3, 3, C + Cu, Co)) / np.sqrt(9 * (C + Cu))).astype(np.float32)
    b = rng.normal(0, 0.1, Co).astype(np.float32)
    direct = coracle.conv_probe(x, xu, w, b, False)
    wino = coracle.conv_probe(x, xu, w, b, True)
    ref = _ref64(x, xu, w, b)
    assert np.abs(direct - ref).max() < 1e-5 and np.abs(wino - ref).max() < 1e-5
    # (C == 0, nothing to transform: still not the same bits -- TZ-PA2 walks an upsampled source quad by quad, taps inside,
    # TZ-PA1 in blocks of 16 channels: round 4, when the statement met its kernel)
    np.testing.assert_array_equal(wino, coracle.conv_probe(x, xu, w, b, True))   # deterministic (OpenMP over tile rows)


def test_winograd_statement_on_structured_inputs():
    """Constant images, single impulses at the borders and a checkerboard: the places where a wrong patch origin, a wrong
    sign in a transform or a wrong zero padding would show at once."""
    C, Co = 16, 6
    rng = np.random.default_rng(5)
    w = rng.normal(0, 0.2, (3, 3, C, Co)).astype(np.float32)
    b = np.zeros(Co, np.float32)
    for H, W in ((6, 6), (7, 5)):
        imgs = [np.ones((H, W, C), np.float32)]
        for y, x in ((0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (H // 2, W // 2)):
            im = np.zeros((H, W, C), np.float32)
            im[y, x, 3] = 1.0
            imgs.append(im)
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        imgs.append(np.repeat(((yy + xx) % 2)[..., None], C, 2).astype(np.float32))
        for im in imgs:
            np.testing.assert_allclose(coracle.conv_probe(im, None, w, b, True), _ref64(im, None, w, b), atol=2e-6)
