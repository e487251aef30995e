"""TZ-PA2 on the MI355X (-m gpu): the per-frame convolutions of levels >= 1 with their same-resolution source as
Winograd F(2x2, 3x3) chains (k_wino, tezip_amd/csrc/tz_wino_kernels.hip.h; prednet.py:254-277 are the convolutions) against
the C oracle's statement of the same chains (oracle/tz_oracle.c conv3x3_wino, tzo_model_set_contract(2)) BIT FOR BIT --
predictions, every level's error units and representations, whole compress -> decompress jobs -- and against TZ-PA1
to float32 accuracy (the two contracts are the same function in a different summation order)."""
import numpy as np
import pytest

from oracle import coracle
from oracle import oracle as O
from tezip_amd import synth
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu

FULL = PredNetConfig()


@pytest.fixture()
def ctx():
    from tezip_amd import _lib
    c = _lib.Context(0)
    c.set_contract(2)
    yield c
    c.close()


def _net(w, cfg, hp, wp, contract=2):
    return coracle.CPredNet(w, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp).set_contract(contract)


@pytest.mark.parametrize("hp,wp,bias", [(64, 64, 0.1), (72, 88, 0.0), (8, 8, 0.2), (8, 40, 0.2), (24, 40, 0.3), (128, 160, 0.1)])
def test_predictor_bit_exact_vs_the_oracle_under_pa2(ctx, hp, wp, bias):
    """Frame sizes whose levels are whole 16x16 tiles, ragged tiles, odd top levels (24x40 -> 3x5), one-pixel top levels
    (8x8) and one-tile strips; three frames through a batch of two (batch invariance); the recursion fed back."""
    assert ctx.get_contract() == 2
    rng = np.random.default_rng(7)
    w = FULL.init_weights(seed=11, bias_scale=bias)
    net = _net(w, FULL, hp, wp)
    ctx.load_model(FULL, w)
    ctx.prepare(hp, wp, max_batch=2)
    np.testing.assert_array_equal(ctx.predict_c0(), net.c0())          # constants are TZ-PA1 in both contracts
    frames = rng.integers(0, 256, (3, hp, wp, 3)).astype(np.float32) / np.float32(255)
    got = ctx.predict_next(frames)
    for i in range(3):
        ref, dbg = net.next(frames[i], debug=True)
        if i == 2:  # the taps hold the last batch (n == 1)
            for l in range(FULL.nb_layers):
                np.testing.assert_array_equal(ctx.predict_tap(0, l), dbg["e"][l], err_msg="e level %d" % l)
            for l in reversed(range(FULL.nb_layers)):
                np.testing.assert_array_equal(ctx.predict_tap(1, l), dbg["r"][l], err_msg="r level %d" % l)
        np.testing.assert_array_equal(got[i], ref, err_msg="frame %d" % i)
    again = ctx.predict_next(got[:1])
    np.testing.assert_array_equal(again[0], net.next(got[0]))
    # the kernels really were the TZ-PA2 ones
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.predict_next(frames[:1])
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    # (5 k_wino launches per step fused, 7 where the measured "E-part ahead" decision split the level-1 / level-2 gates)
    assert prof["wino_pa2"][1] in (5, 7) and prof["conv16_lds_dma"][1] == 0 and prof["convlat_small_grid"][1] == 0


def test_pa2_is_the_same_function_as_pa1_in_another_summation_order(ctx):
    rng = np.random.default_rng(9)
    w = FULL.init_weights(seed=5, bias_scale=0.2)
    ctx.load_model(FULL, w)
    ctx.prepare(64, 96, max_batch=2)
    frames = rng.integers(0, 256, (2, 64, 96, 3)).astype(np.float32) / np.float32(255)
    p2 = ctx.predict_next(frames)
    taps2 = [ctx.predict_tap(1, l) for l in range(4)]
    ctx.set_contract(1)
    p1 = ctx.predict_next(frames)
    taps1 = [ctx.predict_tap(1, l) for l in range(4)]
    np.testing.assert_array_equal(p1[1], _net(w, FULL, 64, 96, 1).next(frames[1]))
    assert not np.array_equal(p1, p2) or not np.array_equal(taps1[3], taps2[3])     # different bits somewhere ...
    assert np.abs(p1 - p2).max() < 2e-5                                              # ... the same numbers
    for a, b in zip(taps1, taps2):
        assert np.abs(a - b).max() < 2e-5


@pytest.mark.parametrize("stack,rstack,hp,wp", [
    ((3, 32, 64), (3, 48, 32), 40, 56),             # R != stack: 48-column A blocks (NT = 3), 32 / 48-channel sources
    ((3, 64, 16), (4, 16, 64), 32, 48),             # 64 -> 16 columns at level 1: NOT a k_wino convolution (falls back), top gates are
    ((3, 16), None, 24, 40),                        # two levels: only the top gates qualify
    ((3, 48, 96, 192), (16, 48, 96, 192), 32, 32),
])
def test_other_model_shapes_under_pa2(ctx, stack, rstack, hp, wp):
    """Which convolutions TZ-PA2 evaluates as Winograd chains is a function of the model shape (pack_wino in tz_prednet.hip
    = wino_gate_ok / wino_a_ok in the oracle): both sides must agree on it, whatever the shape."""
    cfg = PredNetConfig(stack_sizes=stack, R_stack_sizes=rstack)
    rng = np.random.default_rng(21)
    w = cfg.init_weights(seed=13, bias_scale=0.25)
    net = _net(w, cfg, hp, wp)
    ctx.load_model(cfg, w)
    ctx.prepare(hp, wp, max_batch=3)
    frames = rng.integers(0, 256, (4, hp, wp, 3)).astype(np.float32) / np.float32(255)
    got = ctx.predict_next(frames)
    for i in range(4):
        np.testing.assert_array_equal(got[i], net.next(frames[i]), err_msg="frame %d" % i)


@pytest.mark.parametrize("nt,h,w,p,window,thr,mode,bound", [
    (9, 21, 30, 0, 4, None, "abs", [2.0]),
    (11, 61, 90, 1, 5, None, "abs", [0.0]),
    (8, 64, 64, 0, None, 0.004, "rel", [0.01]),
])
def test_compress_decompress_matches_oracle_under_pa2(ctx, nt, h, w, p, window, thr, mode, bound):
    """Whole jobs: rollout (SWP batched / DWP), encode, decoder replay and decode under TZ-PA2 vs the numpy oracle driving
    the C predictor with the same contract: key mask, payload, table and decoded frames."""
    from tezip_amd import _lib
    hp, wp = _lib.pad8(h), _lib.pad8(w)
    frames = synth.translating_scene(nt, h, w, seed=31)
    wts = FULL.init_weights(seed=3, bias_scale=0.2)
    ctx.load_model(FULL, wts)
    ctx.prepare(hp, wp, max_batch=3)

    class P:
        net = _net(wts, FULL, hp, wp)

        def c0(self, a, b):
            return self.net.c0()

        def next(self, f):
            return self.net.next(np.asarray(f, np.float32))

    key, _ = ctx.rollout(frames, p, window, thr)
    payload, table, _ = ctx.encode(mode, bound, True)
    ref = O.compress_oracle(frames, p, window, thr, mode, bound, P(), True)
    ref_payload, ref_table, _, _ = O.parse_stream(ref["stream"])
    assert (key == ref["key"]).all()
    np.testing.assert_array_equal(table, ref_table)
    np.testing.assert_array_equal(payload, ref_payload)
    ctx.rollout_decode(ref["key_frame"].reshape(nt, h, w, 3), p)
    dec = ctx.decode(payload, table)
    np.testing.assert_array_equal(dec, O.decode_stream(ref["stream"], ref["key_frame"], P()))
    if bound[0] == 0:
        np.testing.assert_array_equal(dec, frames)
    # a decoder on the OTHER contract does not reproduce a lossless stream bit for bit: the contract is part of the job
    if bound[0] == 0:
        ctx.set_contract(1)
        ctx.rollout_decode(ref["key_frame"].reshape(nt, h, w, 3), p)
        other = ctx.decode(payload, table)
        assert np.abs(other.astype(int) - frames.astype(int)).max() <= 1


@pytest.mark.parametrize("ipw", [2, 3, 6, 12])
def test_column_blocks_per_workgroup(ipw, monkeypatch):
    """A k_wino workgroup does several column blocks of its tile one after the other, the DMA stream running on from one
    into the next (launch_wino_t picks how many from the grid; at these sizes it would pick 1): forced here through
    TEZIP_WINO_IPW (every convolution whose column-block count it divides), ragged tiles included -- same bits."""
    from tezip_amd import _lib
    monkeypatch.setenv("TEZIP_WINO_IPW", str(ipw))
    c = _lib.Context(0)
    try:
        c.set_contract(2)
        rng = np.random.default_rng(17)
        w = FULL.init_weights(seed=19, bias_scale=0.15)
        for hp, wp in ((72, 88), (32, 32)):
            net = _net(w, FULL, hp, wp)
            c.load_model(FULL, w)
            c.prepare(hp, wp, max_batch=2)
            frames = rng.integers(0, 256, (3, hp, wp, 3)).astype(np.float32) / np.float32(255)
            got = c.predict_next(frames)
            for i in range(3):
                np.testing.assert_array_equal(got[i], net.next(frames[i]), err_msg="%dx%d frame %d" % (hp, wp, i))
    finally:
        c.close()
