"""'E-part ahead' (round 5, tz_prednet.hip): under TZ-PA2 a gate convolution of a level with an upsampled source can run
as two k_wino launches -- the same-resolution phase on a second stream as soon as E_l exists (RAW epilogue into P_l), the
upsampled phase on the critical path starting from P_l (k_wino<..., NOSAME>).  Same chains, same order: every result
must be bit-identical to the fused launch and to the C oracle's TZ-PA2 statement.  TEZIP_EPART=1 forces it, =0 forbids it;
by default (round 6) the first step of a batch size MEASURES both forms and keeps the split where it is >= 2 % faster
(tz_prednet.hip epart_measure; profiles/r06/epart_shapes.md) -- never where the step's launches fill the chip anyway."""
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
    """The measured decision where its outcome is not in doubt: B = 1 at 512x512 (the level-3 gates are 192 workgroups on 256
    CUs; the split measures +5..6 %) splits, the cfg3 bench shape (4 windows: whole rounds everywhere, nothing idle, never
    even tried) does not.  Either way the rollout equals the C oracle's; the measurement's own launches are not counted."""
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
                # (+5..6 % measured for the split here against the 2 % it must show to be kept: on a box that is busy during the
                # six passes of the measurement the decision may come out fused -- either way the bits below are the oracle's;
                # that the decision DOES engage where it pays is asserted at 256x256, +16 %, in the next test)
                assert p["wino_pa2"][1] in (5 * steps, expect * steps), p["wino_pa2"]
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


@pytest.mark.parametrize("hp,wp,batch,expect", [(256, 256, 1, 7), (1024, 1024, 1, 5), (384, 384, 2, None), (256, 256, 5, None)])
def test_measured_decision_outside_512(monkeypatch, hp, wp, batch, expect):
    """shapes the round-5 rule was never measured at.  256x256 one window: the split is worth +16 %
    (the old rule left it off); 1024x1024 one window: whole chip-rounds, fused, not measured at all; 384x384 two windows
    and 256x256 five windows: the old rule's worst cases (-4.7 %, -7 % when forced) -- whatever the measurement picks
    there, the predictions are the fused launch's bits, also on the step that ran the measurement."""
    rng = np.random.default_rng(hp + 3 * batch)
    w = FULL.init_weights(seed=123)
    nt, window = 3 * batch, 3
    frames = rng.integers(0, 256, (nt, hp, wp, 3), dtype=np.uint8)
    stacks, counts = {}, {}
    for mode in (0, None):
        c = _ctx(monkeypatch, mode)
        try:
            c.load_model(FULL, w)
            c.prepare(hp, wp, max_batch=batch)
            assert c.get_contract() == 2
            c.prof_enable(True)
            c.prof_reset()
            c.rollout(frames, 0, window)              # the first step of the default context measures (prof off meanwhile)
            counts[mode] = c.prof_get()["wino_pa2"][1]
            c.prof_enable(False)
            stacks[mode] = c.get_predictions()
            c.rollout(frames, 0, window)              # ... the second rollout runs on the cached decision
            np.testing.assert_array_equal(c.get_predictions(), stacks[mode])
        finally:
            c.close()
    np.testing.assert_array_equal(stacks[None], stacks[0])
    steps = window - 1
    assert counts[0] == 5 * steps
    assert counts[None] in (5 * steps, 7 * steps)
    if expect is not None:
        assert counts[None] == expect * steps, counts


def test_split_launches_over_random_shapes_soak():
    """scripts/soak_epart.py as a test (40 random frame sizes 8..408 x 8..408, batches 1..5, forced split vs forbidden, bit
    for bit): the split now runs wherever it measures faster, so a mistake in its hand-overs -- round 6 built one: a side
    launch whose completion nobody waited for -- must show here and not in a user's images."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_epart.py"), "--cases", "40", "--seed", "7"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatching" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
