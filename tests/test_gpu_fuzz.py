"""Seeded random sweeps of the codec kernels against the C oracle (-m gpu): many shapes, data styles, modes and bounds per
test instead of a few hand-picked ones -- ragged sizes, one-pixel frames, chains shorter than a lane group and longer than
a tile, saturated and constant data, tolerances from below one grey level to beyond the value range."""
import numpy as np
import pytest

from oracle import coracle
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tezip_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _style(rng, kind, shape):
    n = int(np.prod(shape))
    if kind == 0:      # noise
        x = rng.integers(0, 256, n)
    elif kind == 1:    # smooth random walk
        x = 128 + np.cumsum(rng.normal(0, 1.2, n))
    elif kind == 2:    # constant with isolated spikes
        x = np.full(n, int(rng.integers(0, 256)), float)
        x[rng.integers(0, n, max(1, n // 97))] = rng.integers(0, 256, max(1, n // 97))
    elif kind == 3:    # saturated halves
        x = np.where(np.arange(n) % max(2, n // 3) < max(1, n // 6), 255, 0)
    elif kind == 4:    # triangle ramps (runs that break at regular distances)
        p = int(rng.integers(3, 200))
        x = np.abs((np.arange(n) % (2 * p)) - p) * (255.0 / p)
    else:              # low-amplitude noise around a level (long runs under abs bounds)
        x = int(rng.integers(20, 230)) + rng.normal(0, 1.0, n)
    return np.clip(np.round(x), 0, 255).astype(np.uint8).reshape(shape)


class _Predictor:
    def __init__(self, net):
        self.net = net

    def c0(self, hp, wp):
        return self.net.c0()

    def next(self, frame):
        return self.net.next(np.asarray(frame, dtype=np.float32))


MODES = [("abs", lambda r: [float(r.choice([0.0, 0.3, 0.5, 1.0, 1.5, 2.0, 3.7, 8.0, 40.0, 300.0, -2.0]))]),
         ("rel", lambda r: [float(r.choice([0.0, 1e-4, 1e-3, 0.01, 0.05, 0.3, 1.0, 2.5]))]),
         ("absrel", lambda r: [float(r.choice([0.0, 1.0, 2.5, 6.0])), float(r.choice([0.0, 0.001, 0.02, 0.5]))]),
         ("pwrel", lambda r: [float(r.choice([0.0, 0.001, 0.02, 0.1, 0.5, 1.0, 3.0]))])]


@pytest.mark.parametrize("seed", range(12))
def test_error_bound_random_shapes_styles_and_bounds(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(24):
        h = int(rng.choice([1, 2, 3, 7, 16, 21, 33, 64, 65, 96, 130, 256]))
        w = int(rng.choice([1, 2, 5, 16, 31, 64, 100, 128, 171, 256, 512]))
        n = int(rng.integers(1, 4))
        orig = _style(rng, int(rng.integers(0, 6)), (n, h, w, 3))
        pred = _style(rng, int(rng.integers(0, 6)), (n, h, w, 3)).astype(np.int16)
        jitter = rng.integers(-3, 4, (n, h, w, 3)) if rng.random() < 0.7 else rng.integers(-255, 256, (n, h, w, 3))
        diff = np.clip((pred - orig.astype(np.int16)) // int(rng.choice([1, 1, 4, 32])) + jitter, -255, 255).astype(np.int16)
        mode, mk = MODES[int(rng.integers(0, 4))]
        bound = mk(rng)
        skip = (rng.random(n) < 0.2).astype(np.uint8)
        got = ctx.error_bound(orig, diff.copy(), mode, bound, skip)
        for i in range(n):
            ref = diff[i] if skip[i] else coracle.error_bound_frame(orig[i], diff[i], mode, bound)
            np.testing.assert_array_equal(got[i], ref, err_msg="seed %d case %d: %dx%dx%d %s %s frame %d" % (seed, case, n, h, w, mode, bound, i))


@pytest.mark.parametrize("seed", range(4))
def test_spatial_delta_table_remap_and_inverses_random_lengths(ctx, seed):
    rng = np.random.default_rng(2000 + seed)
    for case in range(20):
        n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 65537, 300007]))
        sigma = float(rng.choice([0.0, 0.5, 3.0, 40.0, 200.0]))
        x = np.clip(np.round(rng.normal(0, sigma, n)), -255, 255).astype(np.int16)
        carry = int(rng.integers(-255, 256)) if rng.random() < 0.5 else None
        hist = np.zeros(2111, np.uint64)
        y = ctx.spatial_delta(x, 1, hist=hist, carry=carry)
        ref_y = coracle.spatial_delta(x, 1)
        if carry is not None:   # a shard in the middle of the stream: its first symbol is relative to the element in front
            ref_y[0] = np.int32(1600 - (carry - int(x[0]))).astype(np.int16)
        np.testing.assert_array_equal(y, ref_y, err_msg="seed %d case %d n %d" % (seed, case, n))
        np.testing.assert_array_equal(hist, coracle.histogram(y).astype(np.uint64))
        table = ctx.build_table(hist)
        np.testing.assert_array_equal(table, O.build_table(y))
        ranks = ctx.remap(y, table)
        np.testing.assert_array_equal(ranks, O.remap_enc(y, table))
        sd = ctx.unmap(ranks, table, offset=True)
        np.testing.assert_array_equal(sd, (1600 - y.astype(np.int32)).astype(np.int16))
        back = ctx.spatial_undelta(sd, carry=carry) if carry is not None else ctx.spatial_undelta(sd)
        np.testing.assert_array_equal(back, x)


@pytest.mark.parametrize("seed", range(6))
def test_whole_jobs_random_small_shapes_vs_oracle(ctx, seed):
    """rollout + encode + decode of small random jobs (frames of a multiple of 16 elements take the decoder's one-launch
    tail, the others the separate launches) against the oracle's streams and decoder."""
    from tezip_amd.prednet import PredNetConfig
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    rng = np.random.default_rng(3000 + seed)
    wts = cfg.init_weights(seed=40 + seed, bias_scale=0.2)
    for case in range(5):
        h, w = int(rng.choice([8, 16, 24, 21, 40])), int(rng.choice([8, 16, 30, 32, 48]))
        nt, p = int(rng.integers(6, 13)), int(rng.integers(0, 3))
        window = int(rng.integers(1, 6))
        mode, mk = MODES[int(rng.integers(0, 4))]
        bound = mk(rng)
        entropy = bool(rng.random() < 0.8)
        frames = _style(rng, 1, (nt, h, w, 3))
        frames[:, 0, 0, 0] |= 1   # an all-black key frame is invisible to the reference's decoder (decompress.py:123-129)
        hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
        net = coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)

        pred = _Predictor(net)
        ref = O.compress_oracle(frames, p, window, None, mode, bound, pred, entropy)
        ctx.load_model(cfg, wts)
        ctx.prepare(hp, wp, max_batch=4)
        key, _ = ctx.rollout(frames, p, window, None)
        np.testing.assert_array_equal(key, ref["key"])
        payload, table, _ = ctx.encode(mode, bound, entropy)
        sp, rt, _, _ = O.parse_stream(ref["stream"])
        msg = "seed %d case %d: nt %d %dx%d p %d w %d %s %s" % (seed, case, nt, h, w, p, window, mode, bound)
        np.testing.assert_array_equal(payload, sp, err_msg=msg)
        if entropy:
            np.testing.assert_array_equal(table, rt, err_msg=msg)
        ctx.rollout_decode(ref["key_frame"].reshape(nt, h, w, 3), p)
        dec = ctx.decode(payload, table)
        np.testing.assert_array_equal(dec, O.decode_stream(ref["stream"], ref["key_frame"], pred), err_msg=msg)


SMALL_E = [("abs", lambda r: [float(r.choice([0.1, 0.25, 0.255, 0.3, 0.45, 0.499, 0.4991, 0.5, 1e-9]))]),
           ("rel", lambda r: [float(r.choice([1e-4, 1e-3, 0.0019, 0.00195, 0.00196, 0.002]))]),
           ("absrel", lambda r: [float(r.choice([0.3, 0.499, 5.0])), float(r.choice([0.0005, 0.0019, 0.5]))]),
           ("pwrel", lambda r: [float(r.choice([1e-4, 1e-3, 0.0019, 0.00196, 0.002]))])]


@pytest.mark.parametrize("seed", range(6))
def test_whole_jobs_small_tolerances_vs_oracle(ctx, seed):
    """Round 6: tolerances around the limit of the identity shortcut (tz_quant_is_identity: worst-case E <= 0.499 -- below
    it error_bound leaves every delta as it is and the one-pass lossless kernel runs; just above it the general quantiser) on UNPADDED frames (the
    fused encode), full-range and narrow-range data (rel: E = range * b per chain), with warm-up frames (not quantised);
    payload, table and decoded frames against the oracle's."""
    from tezip_amd.prednet import PredNetConfig
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    rng = np.random.default_rng(4000 + seed)
    wts = cfg.init_weights(seed=60 + seed, bias_scale=0.2)
    for case in range(6):
        h, w = int(rng.choice([8, 16, 24, 40])), int(rng.choice([8, 16, 32, 48]))
        nt, p = int(rng.integers(6, 12)), int(rng.integers(0, 3))
        window = int(rng.integers(2, 6))
        mode, mk = SMALL_E[int(rng.integers(0, 4))]
        bound = mk(rng)
        entropy = bool(rng.random() < 0.7)
        frames = _style(rng, int(rng.choice([0, 1, 5])), (nt, h, w, 3))
        frames[:, 0, 0, 0] |= 1
        net = coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, h, w)
        pred = _Predictor(net)
        ref = O.compress_oracle(frames, p, window, None, mode, bound, pred, entropy)
        ctx.load_model(cfg, wts)
        ctx.prepare(h, w, max_batch=4)
        key, _ = ctx.rollout(frames, p, window, None)
        np.testing.assert_array_equal(key, ref["key"])
        payload, table, _ = ctx.encode(mode, bound, entropy)
        sp, rt, _, _ = O.parse_stream(ref["stream"])
        msg = "seed %d case %d: nt %d %dx%d p %d w %d %s %s" % (seed, case, nt, h, w, p, window, mode, bound)
        np.testing.assert_array_equal(payload, sp, err_msg=msg)
        if entropy:
            np.testing.assert_array_equal(table, rt, err_msg=msg)
        ctx.rollout_decode(ref["key_frame"].reshape(nt, h, w, 3), p)
        dec = ctx.decode(payload, table)
        np.testing.assert_array_equal(dec, O.decode_stream(ref["stream"], ref["key_frame"], pred), err_msg=msg)
