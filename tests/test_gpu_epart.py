"""'E-part ahead' (round 5, tz_prednet.hip): under TZ-PA2 a gate convolution of a level with an upsampled source can run
as two k_wino launches -- the same-resolution phase on a second stream as soon as E_l exists (RAW epilogue into P_l), the
upsampled phase on the critical path starting from P_l (k_wino<..., NOSAME>).  Same chains, same order: every result
must be bit-identical to the fused launch and to the C oracle's TZ-PA2 statement.  On by itself only where a step's
launches cannot fill the chip (one window at a time at 512x512); TEZIP_EPART=1 forces it, =0 forbids it."""
import numpy as np
import pytest

from oracle import coracle
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu
FULL = PredNetConfig()


def _ctx(monkeypatch, mode):
    if mode is None:
        monkeypatch.delenv("TEZIP_EPART", raising=False)
    else:
        monkeypatch.setenv("TEZIP_EPART", str(mode))
    return _lib.Context(0)


@pytest.mark.parametrize("hp,wp,batch", [(64, 64, 1), (72, 88, 3), (32, 32, 2), (128, 160, 2)])
def test_forced_split_is_bit_identical_to_the_fused_launch_and_the_oracle(monkeypatch, hp, wp, batch):
    rng = np.random.default_rng(hp * 7 + wp)
    w = FULL.init_weights(seed=31, bias_scale=0.15)
    frames = rng.integers(0, 256, (batch + 1, hp, wp, 3)).astype(np.float32) / np.float32(255)
    outs, taps = {}, {}
    for mode in (0, 1):
        c = _ctx(monkeypatch, mode)
        try:
            c.load_model(FULL, w)
            c.prepare(hp, wp, max_batch=batch)
            c.set_contract(2)
            c.prof_enable(True)
            c.prof_reset()
            outs[mode] = c.predict_next(frames)        # batch + 1 frames through max_batch: a full batch and a batch of one
            n_wino = c.prof_get()["wino_pa2"][1]
            c.prof_enable(False)
            taps[mode] = [c.predict_tap(1, l) for l in range(4)]
            # 5 k_wino launches per predictor call fused, 7 split (levels 1 and 2 have an upsampled source), two calls
            assert n_wino == (14 if mode else 10), (mode, n_wino)
            again = c.predict_next(outs[mode][:1])       # the hand-over buffers are reused step after step
            outs[(mode, "again")] = again
        finally:
            c.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[(0, "again")], outs[(1, "again")])
    for a, b in zip(taps[0], taps[1]):
        np.testing.assert_array_equal(a, b)
    net = coracle.CPredNet(w, FULL.stack_sizes, FULL.R_stack_sizes, hp, wp).set_contract(2)
    for i in range(batch + 1):
        np.testing.assert_array_equal(outs[1][i], net.next(frames[i]), err_msg="frame %d" % i)


def test_default_engages_at_one_window_of_512_and_not_at_four(monkeypatch):
    """The heuristic: a step whose k_wino launches leave more than a fifth of a chip-round idle (B = 1 at 512x512: the
    level-3 gates are 192 workgroups on 256 CUs) splits, the cfg3 bench shape (4 windows: whole rounds everywhere) does not.
    Either way the rollout equals the C oracle's."""
    frames = synth.turbulence(6, 512, 512, seed=9)
    w = FULL.init_weights(seed=123)
    net = coracle.CPredNet(w, FULL.stack_sizes, FULL.R_stack_sizes, 512, 512)
    c = _ctx(monkeypatch, None)
    try:
        c.load_model(FULL, w)
        for batch, nt, window, expect in ((1, 3, 3, 7), (4, 6, 2, 5)):   # (2 steps of one window) / (1 step of three windows... of 4 slots)
            c.prepare(512, 512, max_batch=batch)
            assert c.get_contract() == 2
            c.prof_enable(True)
            c.prof_reset()
            key, _ = c.rollout(frames[:nt], 0, window)
            p = c.prof_get()
            c.prof_enable(False)
            steps = window - 1
            if batch == 1:
                assert p["wino_pa2"][1] == expect * steps, p["wino_pa2"]
            pred = c.get_predictions()
            cur = coracle.u8_to_f32_frame(frames[0], 512, 512)
            for d in range(1, steps + 1):
                cur = net.next(cur)
                np.testing.assert_array_equal(pred[d], cur, err_msg="batch %d depth %d" % (batch, d))
    finally:
        c.close()
    # four windows in flight: fused
    c = _ctx(monkeypatch, None)
    try:
        c.load_model(FULL, w)
        c.prepare(512, 512, max_batch=4)
        f8 = synth.turbulence(8, 512, 512, seed=10)
        c.prof_enable(True)
        c.prof_reset()
        c.rollout(f8, 0, 2)
        assert c.prof_get()["wino_pa2"][1] == 5
    finally:
        c.close()
