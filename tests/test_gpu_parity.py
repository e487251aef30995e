"""GPU parity tests (run with -m gpu on a MI355X): every HIP entry point against the CPU
oracle on the same seeded inputs.  Integer/byte work and -- thanks to the TZ-PA1 fixed
fmaf-chain arithmetic -- the float32 predictor are compared BIT-EXACT."""
import numpy as np
import pytest

from oracle import coracle
from oracle import oracle as O
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu

SMALL = PredNetConfig(stack_sizes=(3, 16, 32))
FULL = PredNetConfig()


@pytest.fixture(scope="module")
def ctx():
    from tezip_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _frames(rng, nt, h, w):
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out = []
    for t in range(nt):
        base = 120 + 60 * np.sin((xx + 2 * t) / 5.0) + 40 * np.cos((yy - t) / 4.0)
        img = np.stack([base, base * 0.7 + 30, 255 - base * 0.5], axis=-1) + rng.normal(0, 3.0, (h, w, 3))
        out.append(np.clip(np.round(img), 0, 255).astype(np.uint8))
    return np.stack(out)


def _pad(h):
    return (h + 7) // 8 * 8


# ------------------------------------------------------------------------------ codec ops
@pytest.mark.parametrize("h,w", [(64, 64), (21, 30), (128, 160)])
def test_delta_encode(ctx, h, w):
    rng = np.random.default_rng(1)
    n = 5
    pred = rng.random((n, _pad(h), _pad(w), 3), dtype=np.float32)
    pred[0, 0, 0] = [0.0, 1.0, 0.5]
    orig = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    zero = np.array([1, 0, 0, 1, 0], np.uint8)
    got = ctx.delta_encode(pred, orig, zero)
    ref = np.stack([coracle.delta_frame(pred[i], orig[i], bool(zero[i])) for i in range(n)])
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("mode,bound", [("abs", [2.0]), ("abs", [0.4]), ("abs", [-3.0]), ("abs", [0.0]),
                                        ("rel", [0.01]), ("rel", [0.1]), ("absrel", [3.0, 0.01]),
                                        ("absrel", [1.0, 0.5]), ("absrel", [2.0, 0.0]), ("pwrel", [0.05]),
                                        ("pwrel", [1.0])])
def test_error_bound(ctx, mode, bound):
    rng = np.random.default_rng(2)
    n, h, w = 4, 45, 61  # chains longer than one staged chunk (1024 px) and one fill block (2048 px)
    orig = _frames(rng, n, h, w)
    noise = rng.integers(-255, 256, (n, h, w, 3))
    smooth = np.clip(np.round(np.cumsum(rng.normal(0, 0.7, (n, h * w, 3)), axis=1)), -255, 255).reshape(n, h, w, 3)
    diff = np.where(np.arange(n)[:, None, None, None] % 2 == 0, noise, smooth).astype(np.int16)
    diff[3] = 0
    diff[3, 10, 5:9, 1] = [3, -7, 9, 1]
    skip = np.array([0, 0, 1, 0], np.uint8)
    got = ctx.error_bound(orig, diff.copy(), mode, bound, skip)
    for i in range(n):
        ref = diff[i] if skip[i] else coracle.error_bound_frame(orig[i], diff[i], mode, bound)
        np.testing.assert_array_equal(got[i], ref, err_msg="frame %d" % i)


@pytest.mark.parametrize("mode,bound", [("abs", [300.0]), ("abs", [700.0]), ("abs", [2.0]), ("abs", [0.49999999999999994]),
                                        ("rel", [1.5]), ("absrel", [400.0, 3.0])])
def test_error_bound_takes_any_int16_stack(ctx, mode, bound):
    """The stand-alone operator is compress.py:23-70 on ANY int16 slab, not only on the [-255, 255] deltas of
    compress.py:292-314 that the integer walk's width table covers: a tile that holds a wider value evaluates the
    reference's double test directly.  Diffs of +-2000 and up to the int16 limits, runs wider than 511 that
    compress.py:60 merges under a tolerance of 300 / 700, wide and narrow tiles side by side (frame 1 keeps
    some tiles in range), vs the C oracle."""
    rng = np.random.default_rng(22)
    n, h, w = 3, 96, 128          # chains of 12,288 elements = 3 tiles of the quantiser
    orig = _frames(rng, n, h, w)
    walk = np.round(np.cumsum(rng.normal(0, 40.0, (h * w, 3)), axis=0))
    d0 = np.clip(walk + rng.integers(-1000, 1001, (h * w, 3)), -2000, 2000).reshape(h, w, 3)
    d1 = np.clip(np.round(np.cumsum(rng.normal(0, 1.0, (h * w, 3)), axis=0)), -255, 255).reshape(h, w, 3)
    d1[40:50] += rng.integers(-2000, 2001, (10, w, 3))           # only the middle tile is wide
    d2 = rng.integers(-32768, 32768, (h, w, 3))
    d2[::2] = np.clip(d2[::2], -600, 600)
    diff = np.stack([d0, d1, d2]).astype(np.int16)
    assert int(np.abs(diff[0]).max()) > 1500 and int(np.abs(diff[1][:30]).max()) <= 255
    got = ctx.error_bound(orig, diff.copy(), mode, bound)
    for i in range(n):
        ref = coracle.error_bound_frame(orig[i], diff[i], mode, bound)
        np.testing.assert_array_equal(got[i], ref, err_msg="frame %d" % i)
    if mode == "abs" and bound[0] >= 300:
        assert (np.diff(got[0].reshape(-1, 3), axis=0) == 0).mean() > 0.5   # runs really merge across widths > 511


def _long_chain_cases():
    import os
    from conftest import GOLDEN
    lc = np.load(os.path.join(GOLDEN, "ref_long.npz"))
    return lc, range(int(lc["lc_n"]))


@pytest.mark.parametrize("i", _long_chain_cases()[1])
def test_error_bound_against_reference_long_chains(ctx, i):
    """tz_error_bound on chains the REFERENCE itself quantised (tests/golden/ref_long.npz, made by
    make_golden.py::_long_chains from compress.py:23-70): 12,288 / 13,100 / 65,536 elements per
    chain, with runs shorter than, exactly and far longer than the kernel's 64-element chunks and
    1/8-chain segments, pwrel with black pixels (tolerance 0).  Chunk carry, segment speculation
    and k_q_stitch all fire here, and the expected output is the reference's, not the oracle's."""
    lc, _ = _long_chain_cases()
    mode, val = str(lc["lc_%d_mode" % i]), lc["lc_%d_val" % i].tolist()
    orig, diff, res = (lc["lc_%d_%s" % (i, k)] for k in ("orig", "diff", "res"))
    got = ctx.error_bound(orig[None], diff[None].copy(), mode, val)
    np.testing.assert_array_equal(got[0], res)
    # several frames per launch, one of them skipped (compress.py:315-319 skips slot 0 of a group)
    o3 = np.stack([orig, orig[::-1].copy(), orig])
    d3 = np.stack([diff, diff[::-1].copy(), diff])
    got3 = ctx.error_bound(o3, d3.copy(), mode, val, np.array([0, 0, 1], np.uint8))
    np.testing.assert_array_equal(got3[0], res)
    np.testing.assert_array_equal(got3[1], coracle.error_bound_frame(o3[1], d3[1], mode, val))
    np.testing.assert_array_equal(got3[2], diff)


def test_error_bound_chains_that_never_meet_their_speculative_start(ctx):
    """The lane-parallel quantiser (k_q_tiles) walks every 64-element chunk from a speculative fresh start and
    relies on the true chain meeting the speculative one; regular ramps keep two greedy chains out of step for
    ever.  Slow triangle waves over 4+ tiles of 4096 elements: whatever the kernels do -- in-tile stitch rounds
    that never merge, tiles crossed without a common head, the serial fallback k_q_serial -- the result must be
    the oracle's, and the fallback must really have run for some of these chains."""
    h, w = 160, 128                                   # 20,480 elements per chain = 5 tiles
    n = h * w
    i = np.arange(n)
    frames = []
    for period, step in ((10, 1), (7, 1), (23, 2), (3, 1), (64, 1), (50, 3)):
        k = (i // period) * step
        tri = np.abs((k % 400) - 200) - 100           # slope +-step per `period` elements, no jump anywhere
        frames.append(np.stack([tri, -tri, np.roll(tri, 1234)], axis=-1).reshape(h, w, 3))
    diff = np.stack(frames).astype(np.int16)
    orig = np.full(diff.shape, 200, np.uint8)
    orig[:, 0, 0] = 0                                  # range 200 for `rel`
    ctx.prof_enable(True)
    ctx.prof_reset()
    for mode, bound in (("abs", [2.0]), ("abs", [7.0]), ("rel", [0.0126]), ("pwrel", [0.011])):
        got = ctx.error_bound(orig, diff.copy(), mode, bound)
        for f in range(diff.shape[0]):
            np.testing.assert_array_equal(got[f], coracle.error_bound_frame(orig[f], diff[f], mode, bound),
                                          err_msg="%s %s, frame %d" % (mode, bound, f))
    serial = ctx.prof_get()["quant_serial_chains"][1]
    ctx.prof_enable(False)
    assert serial > 0, "none of the ramp chains took the serial fallback: the test no longer covers it"


@pytest.mark.parametrize("mode,bound", [("abs", [0.49999999999999994]), ("abs", [1.4999999999999998]), ("rel", [0.5 / 255 * (1 - 2 ** -53)]),
                                        ("absrel", [1.4999999999999998, 0.9]), ("abs", [300.0])])
def test_error_bound_tolerances_on_a_rounding_edge(ctx, mode, bound):
    """compress.py:60 compares fl(min d + E) with fl(max d - E).  For an E a rounding error away from k/2 the
    widest run that may stand depends on WHERE the deltas sit (0.49999999999999994: 0 and 1 break, 100 and 101
    merge), which is the case k_q_width reports as a band and k_q_tiles resolves with the double test; a huge
    tolerance makes a chain one run.  All against the oracle, which evaluates the doubles literally."""
    rng = np.random.default_rng(77)
    h, w = 64, 96                                      # 6144 elements: two tiles
    walk = np.clip(np.round(np.cumsum(rng.normal(0, 0.6, (2, h * w, 3)), axis=1)), -255, 255)
    walk[1] += 100                                     # the same kind of data around 0 and around 100
    near = rng.integers(0, 2, (1, h * w, 3)) + np.array([0, 100, -100])
    diff = np.concatenate([walk, near]).reshape(3, h, w, 3).astype(np.int16)
    orig = rng.integers(0, 256, diff.shape).astype(np.uint8)
    orig[:, 0, 0], orig[:, 0, 1] = 0, 255
    got = ctx.error_bound(orig, diff.copy(), mode, bound)
    for f in range(3):
        np.testing.assert_array_equal(got[f], coracle.error_bound_frame(orig[f], diff[f], mode, bound), err_msg="frame %d" % f)


def test_error_bound_rejects_negative_pwrel(ctx):
    """A negative tolerance makes the reference assign NaN into its int array at the first element
    (compress.py:60-61): it raises.  pwrel, rel and the rel bound of absrel take the sign of their bound."""
    from tezip_amd._lib import TezipError
    o = np.zeros((1, 8, 8, 3), np.uint8)
    d = np.zeros((1, 8, 8, 3), np.int16)
    for mode, bound in (("pwrel", [-0.1]), ("rel", [-0.01]), ("absrel", [3.0, -0.2])):
        with pytest.raises(TezipError):
            ctx.error_bound(o, d, mode, bound)
    ctx.error_bound(o, d, "abs", [-3.0])   # abs takes |b| (compress.py:29)


@pytest.mark.parametrize("n", [1, 7, 8, 4099, 3 * 64 * 64 * 5 + 3])
@pytest.mark.parametrize("offset", [0, 1])
def test_spatial_delta_histogram_and_inverse(ctx, n, offset):
    rng = np.random.default_rng(3)
    x = np.clip(np.round(rng.normal(0, 6, n)), -255, 255).astype(np.int16)
    hist = np.zeros(2111, np.uint64)
    hist[5] = 7  # counts are ADDED
    y = ctx.spatial_delta(x, offset, hist=hist if offset else None)
    np.testing.assert_array_equal(y, coracle.spatial_delta(x, offset))
    if offset:
        ref_h = coracle.histogram(y).astype(np.uint64)
        ref_h[5] += 7
        np.testing.assert_array_equal(hist, ref_h)
        table = ctx.build_table(coracle.histogram(y))
        np.testing.assert_array_equal(table, O.build_table(y))
        ranks = ctx.remap(y, table)
        np.testing.assert_array_equal(ranks, O.remap_enc(y, table))
        np.testing.assert_array_equal(ctx.unmap(ranks, table, offset=True), (1600 - y.astype(np.int32)).astype(np.int16))
        back = ctx.spatial_undelta(ctx.unmap(ranks, table, offset=True))
    else:
        back = ctx.spatial_undelta(y)
    np.testing.assert_array_equal(back, x)
    # shard carry: second half given the last element of the first half
    if n > 16:
        k = (n // 2) & ~7
        y2 = ctx.spatial_delta(x[k:].copy(), offset, carry=int(x[k - 1]))
        np.testing.assert_array_equal(y2, y[k:])
        sd2 = y[k:] if not offset else (1600 - y[k:].astype(np.int32)).astype(np.int16)
        np.testing.assert_array_equal(ctx.spatial_undelta(sd2.copy(), carry=int(x[k - 1])), x[k:])


@pytest.mark.parametrize("kind", ["wide", "uniform", "spikes", "constant"])
def test_histogram_of_heavy_tailed_symbols(ctx, kind):
    """The block-private histogram keeps the 128 bins around the centre symbol in interleaved copies and
    sends everything farther out through a rarely taken path: distributions that live out there --
    sigma 100, uniform over the whole delta range, isolated +-255 jumps in flat data, one single symbol --
    must count exactly (compress.py:354 bincount)."""
    rng = np.random.default_rng(31)
    n = 3 * 64 * 64 * 7 + 5
    if kind == "wide":
        x = np.clip(np.round(rng.normal(0, 100, n)), -255, 255)
    elif kind == "uniform":
        x = rng.integers(-255, 256, n)
    elif kind == "spikes":
        x = np.zeros(n)
        x[rng.integers(0, n, 200)] = rng.choice([-255, 255, 70, -64, 64, -65, 65, 63], 200)
    else:
        x = np.full(n, 17)
    x = x.astype(np.int16)
    hist = np.zeros(2111, np.uint64)
    y = ctx.spatial_delta(x, 1, hist=hist)
    np.testing.assert_array_equal(y, coracle.spatial_delta(x, 1))
    np.testing.assert_array_equal(hist, coracle.histogram(y).astype(np.uint64))
    assert int(hist.sum()) == n


def test_undelta_wraparound_matches_reference_loop(ctx):
    rng = np.random.default_rng(4)
    s = rng.integers(-32768, 32768, 10007).astype(np.int16)
    np.testing.assert_array_equal(ctx.spatial_undelta(s), coracle.spatial_undelta(s, 0))
    np.testing.assert_array_equal(ctx.spatial_undelta(s), O.finding_difference_dec(s))


def test_inverse_scan_wait_is_bounded_and_fails_loudly(ctx):
    """k_scan2p's workgroups wait for the block sums of the workgroups in front of them.  The wait is bounded: with the
    poll pointed at an epoch nobody publishes (tz_scan_fault_inject) every waiting thread gives up after `poll_limit`
    polls, the launch drains, and the call reports TZ_ERR_HIP instead of hanging the GPU or returning wrong data as
    good; with a device output the error arrives at the context's next synchronisation.  Afterwards the scan works."""
    import torch
    from tezip_amd._lib import TezipError
    rng = np.random.default_rng(14)
    s = rng.integers(-300, 300, 3 * 1024 * 1024 + 5).astype(np.int16)     # thousands of wave-tiles: hundreds of blocks
    want = coracle.spatial_undelta(s, 0)
    np.testing.assert_array_equal(ctx.spatial_undelta(s), want)
    ctx.scan_fault_inject(epoch_skew=1, poll_limit=64)
    try:
        with pytest.raises(TezipError) as e:
            ctx.spatial_undelta(s)                                         # host output: the call itself synchronises
        assert e.value.status == -3 and "k_scan2p" in str(e.value)
        d_in = torch.from_numpy(s).cuda()
        d_out = torch.empty_like(d_in)
        torch.cuda.synchronize()
        ctx.spatial_undelta(d_in, out=d_out)                               # device output: asynchronous ...
        with pytest.raises(TezipError):
            ctx.synchronize()                                              # ... the fault is reported here
    finally:
        ctx.scan_fault_inject(0, 0)
    ctx.synchronize()                                                      # the fault word was consumed
    np.testing.assert_array_equal(ctx.spatial_undelta(s), want)
    d_out.zero_()
    torch.cuda.synchronize()
    ctx.spatial_undelta(d_in, out=d_out)
    ctx.synchronize()
    np.testing.assert_array_equal(d_out.cpu().numpy(), want)


def test_unmap_chained_table(ctx):
    rng = np.random.default_rng(5)
    table = np.array([3, 0, 5, 4, 9, 1], dtype=np.int16)
    ranks = rng.integers(-2, 12, size=999).astype(np.int16)
    np.testing.assert_array_equal(ctx.unmap(ranks, table, offset=False), O.remap_dec(ranks, table))


@pytest.mark.parametrize("h,w", [(64, 64), (21, 30)])
def test_reconstruct_and_sse(ctx, h, w):
    rng = np.random.default_rng(6)
    n = 4
    pred = rng.random((n, _pad(h), _pad(w), 3), dtype=np.float32)
    key = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    diff = rng.integers(-300, 300, (n, h, w, 3)).astype(np.int16)
    km = np.array([1, 0, 0, 1], np.uint8)
    got = ctx.reconstruct(pred, key, km, diff)
    for i in range(n):
        ref = coracle.reconstruct_frame(pred[i], key[i] if km[i] else None, diff[i])
        np.testing.assert_array_equal(got[i], ref)
    sse = ctx.window_sse(key, pred)
    for i in range(n):
        assert sse[i] == coracle.sse_frame(key[i], pred[i])  # same summation order => bit-exact
        x = np.zeros((_pad(h), _pad(w), 3))
        x[:h, :w] = key[i].astype(np.float32) / np.float32(255)
        np.testing.assert_allclose(sse[i], ((x - pred[i].astype(np.float64)) ** 2).sum(), rtol=1e-12)


# ------------------------------------------------------------------------------ predictor
def test_activations_bit_exact_vs_oracle_and_division_free_reciprocal(ctx):
    """The device's hard_sigmoid / tanh against the C oracle's on every float32 of tanh's middle branch [0.625, 9) and a
    dense sample elsewhere (both signs, zeros, the branch points, huge and tiny values); and the kernels' 1 - 2/d without
    a division against the IEEE division for EVERY float32 d in [4, 2^27] (tz_math.hip.h)."""
    lo, hi = np.float32(0.625).view(np.uint32), np.float32(9.0).view(np.uint32)
    mid = np.arange(int(lo) - 64, int(hi) + 64, dtype=np.uint32).view(np.float32)
    rng = np.random.default_rng(3)
    rest = np.concatenate([rng.uniform(-0.7, 0.7, 1 << 20), rng.uniform(-30, 30, 1 << 20), rng.normal(0, 1e-3, 1 << 16),
                           [0.0, -0.0, 0.625, -0.625, 9.0, -9.0, 1e-8, 1e-30, 1e-41, 30.0, -1e30, 2.5, -2.5, 7.5]]).astype(np.float32)
    x = np.concatenate([mid, -mid[::7], rest])
    hs, th, bad = ctx.act_probe(x, check_reciprocal=True)
    assert bad == 0
    ref_hs, ref_th = coracle.act_probe(x)
    np.testing.assert_array_equal(hs.view(np.uint32), ref_hs.view(np.uint32))
    np.testing.assert_array_equal(th.view(np.uint32), ref_th.view(np.uint32))


@pytest.mark.parametrize("cfg,hp,wp,bias", [(SMALL, 16, 24, 0.3), (SMALL, 40, 48, 0.2), (FULL, 64, 64, 0.1),
                                            (FULL, 72, 88, 0.0)])
def test_prednet_bit_exact_vs_canonical_oracle(ctx, cfg, hp, wp, bias):
    rng = np.random.default_rng(7)
    w = cfg.init_weights(seed=11, bias_scale=bias)
    net = coracle.CPredNet(w, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)
    ctx.load_model(cfg, w)
    ctx.prepare(hp, wp, max_batch=2)
    np.testing.assert_array_equal(ctx.predict_c0(), net.c0())
    frames = rng.integers(0, 256, (3, hp, wp, 3)).astype(np.float32) / np.float32(255)
    got = ctx.predict_next(frames)  # 3 frames through a max_batch of 2: batch invariance
    for i in range(3):
        ref, dbg = net.next(frames[i], debug=True)
        if i == 2:  # taps hold the last batch (n == 1)
            for l in range(cfg.nb_layers):
                np.testing.assert_array_equal(ctx.predict_tap(0, l), dbg["e"][l], err_msg="e level %d" % l)
            for l in reversed(range(cfg.nb_layers)):
                np.testing.assert_array_equal(ctx.predict_tap(1, l), dbg["r"][l], err_msg="r level %d" % l)
        np.testing.assert_array_equal(got[i], ref, err_msg="frame %d" % i)
    # recursion: feed predictions back
    again = ctx.predict_next(got[:1])
    np.testing.assert_array_equal(again[0], net.next(got[0]))


SHAPES = [
    # (stack_sizes, R_stack_sizes, hp, wp): which convolution kernels the predictor dispatches to
    ((3, 16), None, 24, 40),                      # 2 levels: k_conv16b (A_0 NT=1, gates), k_conv16 top gates with one 16-block
    ((3, 32, 64), (3, 48, 32), 40, 56),           # R != stack: NT=3 gate columns blocks, 32/48-channel sources (2-3 blocks)
    ((3, 64, 16), (4, 16, 64), 32, 48),           # R_0 = 4 packed gates, NT=4 A_0 (64 columns), 16-column top level
    ((3,), (3,), 16, 24),                         # single level: no upsampled source anywhere
    ((3, 48, 96, 192), (16, 48, 96, 192), 32, 32),  # R_0 = 16: level-0 gates with a 6-channel source on the general kernel
    ((3, 48, 96, 192), None, 8, 8),               # the smallest legal frame: the top level is one pixel, every tile mostly halo
    ((3, 48, 96, 192), None, 8, 40),              # one-tile-high strip
]


@pytest.mark.parametrize("stack,rstack,hp,wp", SHAPES)
def test_prednet_other_model_shapes_bit_exact(ctx, stack, rstack, hp, wp):
    """Every dispatch path of the convolution kernels (k_conv16 / k_conv16b / k_conv_small and the
    general k_conv3x3) against the canonical C oracle, and against each other."""
    cfg = PredNetConfig(stack_sizes=stack, R_stack_sizes=rstack)
    rng = np.random.default_rng(21)
    w = cfg.init_weights(seed=13, bias_scale=0.25)
    net = coracle.CPredNet(w, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)
    ctx.load_model(cfg, w)
    ctx.prepare(hp, wp, max_batch=3)
    np.testing.assert_array_equal(ctx.predict_c0(), net.c0())
    frames = rng.integers(0, 256, (4, hp, wp, 3)).astype(np.float32) / np.float32(255)
    outs = []
    for impl in (1, 0):
        ctx.set_conv_impl(impl)
        outs.append(ctx.predict_next(frames))
    ctx.set_conv_impl(1)
    np.testing.assert_array_equal(outs[0], outs[1])
    for i in range(4):
        np.testing.assert_array_equal(outs[0][i], net.next(frames[i]), err_msg="frame %d" % i)


def test_prednet_matches_independent_numpy_restatement(ctx):
    from oracle import prednet_np
    cfg, hp, wp = SMALL, 16, 24
    rng = np.random.default_rng(8)
    w = cfg.init_weights(seed=5, bias_scale=0.3)
    ctx.load_model(cfg, w)
    ctx.prepare(hp, wp, 1)
    f = rng.integers(0, 256, (hp, wp, 3)).astype(np.float32) / np.float32(255)
    ref = prednet_np.predict(w, cfg.stack_sizes, cfg.R_stack_sizes, np.stack([f, np.zeros_like(f)]))
    np.testing.assert_allclose(ctx.predict_c0(), ref[0], atol=2e-5)  # tolerance: float32 summation order
    np.testing.assert_allclose(ctx.predict_next(f[None])[0], ref[1], atol=2e-5)


# ------------------------------------------------------------------- rollout + full pipeline
class _OraclePredictor:
    def __init__(self, net):
        self.net = net

    def c0(self, hp, wp):
        return self.net.c0()

    def next(self, frame):
        return self.net.next(np.asarray(frame, dtype=np.float32))


CASES = [
    # nt, h, w, p, window, thr, mode, bound, entropy
    (11, 21, 30, 2, 4, None, "abs", [0.0], True),
    (12, 16, 24, 0, 5, None, "abs", [4.0], True),
    (9, 16, 16, 0, 4, None, "rel", [0.02], False),   # boundary on the last frame
    (11, 21, 30, 1, 3, None, "pwrel", [0.05], True),
    (10, 24, 16, 0, None, "auto", "absrel", [3.0, 0.05], True),
    (10, 24, 16, 2, None, "auto", "abs", [0.0], True),
    # edge cases
    (6, 8, 8, 3, 1, None, "abs", [0.0], True),        # window 1: every frame after the warm-up is a key frame
    (7, 9, 9, 0, 50, None, "abs", [1.0], True),       # one window longer than the sequence
    (4, 8, 16, 2, 2, None, "abs", [0.0], False),      # shortest legal sequence: nt = warm_up + 2
    (8, 16, 8, 0, None, 0.0, "abs", [0.0], True),     # DWP threshold 0: every prediction is rejected
    (8, 16, 8, 1, None, 1e9, "rel", [0.05], True),    # DWP threshold never reached: one window
    (5, 8, 8, 0, 2, None, "absrel", [0.0, 0.5], True),  # absrel with abs bound 0: lossless shortcut
    # round 6: tolerances that cannot merge two different deltas (worst-case E <= 0.499) leave every delta as it is --
    # the value of a run of equal deltas d, trunc((fl(d+E) + fl(d-E)) / 2), is d (tz_quant_is_identity) -- so the fused
    # encode of an unpadded job takes the one-pass lossless kernel.  Compared below with the oracle AND with the general
    # quantiser (the delta tap takes tzk_error_bound)
    (10, 16, 24, 0, 4, None, "abs", [0.3], True),
    (10, 16, 24, 2, 3, None, "abs", [0.255], True),      # warm-up frames: not quantised, non-zero deltas pass through
    (9, 64, 64, 1, 4, None, "abs", [0.499], True),       # the last tolerance the map takes
    (9, 64, 64, 0, 4, None, "abs", [0.4995], True),      # ... and the first one the general quantiser keeps
    (10, 16, 24, 0, 5, None, "rel", [1e-3], True),       # BASELINE.json cfg3's bound: E = range * 1e-3 <= 0.255 per chain
    (10, 32, 40, 2, 4, None, "rel", [0.00195], False),   # 255 * b = 0.497
    (9, 16, 24, 0, None, "auto", "rel", [1e-3], True),   # DWP (8 window MSEs: their median is no element -- no tie with the threshold)
    (9, 16, 24, 0, 4, None, "absrel", [0.4, 0.9], True),
    (9, 16, 24, 1, 4, None, "absrel", [3.0, 0.001], True),
    (6, 16, 16, 0, 3, None, "abs", [1e-300], True),      # E so small that d + E == d: still not the lossless shortcut
    (10, 16, 24, 1, 4, None, "pwrel", [0.0019], True),   # a tolerance per element, every one <= 0.4845
    (10, 16, 24, 0, 4, None, "pwrel", [0.002], True),    # 255 * b = 0.51: the general quantiser
]


@pytest.mark.parametrize("nt,h,w,p,window,thr,mode,bound,entropy", CASES)
def test_compress_decompress_matches_oracle(ctx, nt, h, w, p, window, thr, mode, bound, entropy):
    cfg = SMALL
    rng = np.random.default_rng(9)
    frames = _frames(rng, nt, h, w)
    hp, wp = _pad(h), _pad(w)
    wts = cfg.init_weights(seed=3, bias_scale=0.2)
    pred = _OraclePredictor(coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp))
    if thr == "auto":  # pick a threshold inside the observed MSE range so windows have mixed lengths
        probe = O.rollout(frames, p, None, 1e9, pred)
        thr = float(np.median(probe["mse"]))
    ref = O.compress_oracle(frames, p, window, thr, mode, bound, pred, entropy)
    ctx.load_model(cfg, wts)
    ctx.prepare(hp, wp, max_batch=4)
    key, mse = ctx.rollout(frames, p, window, thr, want_mse=True)
    np.testing.assert_array_equal(key, ref["key"])
    # predictions: every slot the encoder uses (slot 0 of each group is a placeholder)
    stack = ctx.get_predictions()
    for start, preds in ref["rollout"]["groups"]:
        for j, pr in enumerate(preds):
            if j > 0 or start < p:
                np.testing.assert_array_equal(stack[start + j], pr, err_msg="prediction of frame %d" % (start + j))
    # DWP decisions come from the MSE; with SWP it is the -v log (compress.py:245-247), boundary
    # frames included (the prediction the reference makes there and then drops)
    got_mse = [m for m in mse[p + 1:]]
    np.testing.assert_allclose(got_mse, ref["mse"], rtol=1e-12)
    if window is not None:  # the log must not change what is encoded: same stack without it
        key_q, _ = ctx.rollout(frames, p, window, thr)
        np.testing.assert_array_equal(key_q, key)
        quiet = ctx.get_predictions()
        for start, preds in ref["rollout"]["groups"]:
            for j, pr in enumerate(preds):
                if j > 0 or start < p:
                    np.testing.assert_array_equal(quiet[start + j], pr)
    payload, table, delta = ctx.encode(mode, bound, entropy, want_delta=True)
    np.testing.assert_array_equal(delta, ref["delta"])
    # without the delta tap the lossless case takes the fused delta+spatial-delta+histogram kernel
    payload2, table2, _ = ctx.encode(mode, bound, entropy)
    np.testing.assert_array_equal(payload2, payload)
    assert (table2 is None) == (table is None) and (table is None or (table2 == table).all())
    stream_payload, ref_table, shape, warm = O.parse_stream(ref["stream"])
    np.testing.assert_array_equal(payload, stream_payload)
    if entropy:
        np.testing.assert_array_equal(table, ref_table)
    else:
        assert table is None and ref_table is None
    # decode on the GPU and compare with the oracle's decoder and (lossless) the input
    key_stack = ref["key_frame"].reshape(nt, h, w, 3)
    kd = ctx.rollout_decode(key_stack, p)
    np.testing.assert_array_equal(kd, ref["key"])
    dec = ctx.decode(payload, table)
    np.testing.assert_array_equal(dec, O.decode_stream(ref["stream"], ref["key_frame"], pred))
    if bound[0] == 0:
        np.testing.assert_array_equal(dec, frames)


@pytest.mark.parametrize("entropy", [True, False])
def test_decode_tail_in_one_launch_equals_the_separate_launches(ctx, entropy, monkeypatch):
    """tz_decode runs inverse remap + inverse spatial delta + reconstruct as ONE launch where the layout allows it
    (unpadded frames of a multiple of 16 elements); TEZIP_DECODE_UNFUSED=1 keeps the scan and reconstruct launches."""
    from tezip_amd import _lib
    cfg = SMALL
    nt, h, w = 13, 64, 80   # 15360 elements per frame: tiles of 4096 straddle frames and key frames
    frames = _frames(np.random.default_rng(21), nt, h, w)
    wts = cfg.init_weights(seed=5, bias_scale=0.2)
    ctx.load_model(cfg, wts)
    ctx.prepare(h, w, max_batch=4)
    key = ctx.rollout(frames, 1, 4, None)[0]
    payload, table, _ = ctx.encode("abs", [2.0], entropy)
    key_stack = np.where(key[:, None, None, None] > 0, frames, 0).astype(np.uint8)

    def decode(c):
        c.rollout_decode(key_stack, 1)
        c.prof_enable(True)
        c.prof_reset()
        out = c.decode(payload, table)
        prof = c.prof_get()
        c.prof_enable(False)
        return out, prof

    one, prof_one = decode(ctx)
    assert prof_one["undelta_scan"][1] == 1 and prof_one["reconstruct"][1] == 0
    monkeypatch.setenv("TEZIP_DECODE_UNFUSED", "1")
    c2 = _lib.Context(0)
    try:
        c2.load_model(cfg, wts)
        c2.prepare(h, w, max_batch=4)
        two, prof_two = decode(c2)
    finally:
        c2.close()
    assert prof_two["undelta_scan"][1] == 1 and prof_two["reconstruct"][1] == 1
    np.testing.assert_array_equal(one, two)
    assert int(np.abs(one.astype(int) - frames.astype(int)).max()) <= 2
    for _ in range(3):   # the status words of the scan carry the launch's epoch: repeated launches without a clear
        np.testing.assert_array_equal(decode(ctx)[0], one)


def test_rollout_rejects_short_sequences_and_bad_sizes(ctx):
    from tezip_amd._lib import TezipError
    cfg = SMALL
    ctx.load_model(cfg, cfg.init_weights(seed=1))
    ctx.prepare(16, 16, 1)
    f = np.zeros((3, 16, 16, 3), np.uint8)
    with pytest.raises(TezipError):
        ctx.rollout(f[:1], 0, 2)
    with pytest.raises(TezipError):
        ctx.rollout(f, 2, 2)
    with pytest.raises(TezipError):  # compress.py:178-181
        ctx.rollout(np.zeros((3, 24, 16, 3), np.uint8), 0, 2)


def test_prepare_rejects_frames_whose_planes_pass_32_bit_offsets(ctx):
    """The convolution kernels address inside one frame's plane of a level with 32-bit offsets: a frame size whose widest
    plane would pass 2^30 floats is refused before anything is allocated (status UNSUPPORTED, message with the numbers);
    the context stays usable."""
    from tezip_amd._lib import TezipError
    ctx.load_model(FULL, FULL.init_weights(seed=1))
    with pytest.raises(TezipError) as ei:
        ctx.prepare(8192, 8192, 1)            # level 1: 4096 x 4096 x 192 gate columns = 3.2e9 floats
    assert "32-bit" in str(ei.value)
    ctx.prepare(16, 16, 1)
    assert ctx.predict_c0().shape == (16, 16, 3)


def test_decode_rejects_key_stacks_that_do_not_cover_the_sequence(ctx):
    from tezip_amd._lib import TezipError
    cfg = SMALL
    ctx.load_model(cfg, cfg.init_weights(seed=1))
    ctx.prepare(16, 16, 2)
    keys = np.zeros((6, 16, 16, 3), np.uint8)
    with pytest.raises(TezipError):      # no key frame at all
        ctx.rollout_decode(keys, 0)
    keys[2] = 7
    with pytest.raises(TezipError):      # first key frame is not frame `warm_up` (decompress.py:147-148 breaks)
        ctx.rollout_decode(keys, 0)
    keys[0] = 9
    assert ctx.rollout_decode(keys, 0).tolist() == [True, False, True, False, False, False]
    with pytest.raises(TezipError):      # payload decode without a decode rollout of matching state is fine, but
        ctx.encode("abs", [0.0], True)   # ... encode after a decode rollout is a call-order error
