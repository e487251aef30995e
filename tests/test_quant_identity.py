"""When is the reference's error_bound (compress.py:23-70) the identity?  The HIP build sends a lossy job whose worst-case
tolerance is <= 0.499 through its one-pass LOSSLESS kernel (tz_codec.hip tz_quant_is_identity, with the proof by binades
and round-to-even).  The CPU half of the evidence: the arithmetic fact, both oracles, and what the reference itself did
in six whole runs at such tolerances (tests/golden/ref_runs4.npz).  The GPU half: tests/test_gpu_qmap.py,
tests/test_gpu_parity.py CASES, tests/test_gpu_fuzz.py, tests/test_gpu_ref_runs.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import coracle
from oracle import oracle as O

R4 = np.load(os.path.join(GOLDEN, "ref_runs4.npz"))


def test_the_value_of_a_run_of_equal_deltas_is_that_delta():
    """compress.py:61,67: a run gets trunc((u + l) / 2), u = min(d + E), l = max(d - E) in float64.  For a run of equal
    integer deltas d that is d, whatever E in [0, 0.5): fl(fl(d + E) + fl(d - E)) == 2 d."""
    rng = np.random.default_rng(6)
    d = np.arange(-255, 256).astype(np.float64)
    tol = np.concatenate([rng.uniform(0.0, 0.499, 20000), rng.uniform(0.0, 1e-6, 2000), np.arange(0, 500) / 1000.0,
                          np.arange(1, 256) * 1e-3, np.arange(1, 256) * 0.00195, [0.499, 0.255, 2.0 ** -53, 0.5 - 2.0 ** -54]])
    for E in tol:
        u, l = d + E, d - E
        assert ((u + l) == 2 * d).all(), E
        assert (np.trunc((u + l) / 2) == d).all(), E
        # ... and two different integers always close a run at E <= 0.499 (compress.py:60: min(u) - max(l) < 0)
        if E <= 0.499:
            assert ((d[:-1] + E) - (d[1:] - E) < 0.0).all(), E


@pytest.mark.parametrize("mode,bound", [("abs", [0.3]), ("abs", [0.499]), ("abs", [1e-12]), ("rel", [1e-3]), ("rel", [0.00195]),
                                        ("absrel", [0.4, 0.9]), ("absrel", [7.0, 0.0015]), ("pwrel", [1e-3]), ("pwrel", [0.00195])])
def test_both_oracles_return_their_input_at_such_tolerances(mode, bound):
    rng = np.random.default_rng(11)
    for trial in range(6):
        h, w = int(rng.choice([1, 7, 16, 33])), int(rng.choice([1, 5, 64, 100]))
        orig = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        if trial % 2:     # long runs of equal deltas, the case the walk merges
            diff = np.repeat(rng.integers(-255, 256, (h, (w + 7) // 8, 3)), 8, axis=1)[:, :w].astype(np.int16)
        else:
            diff = rng.integers(-255, 256, (h, w, 3)).astype(np.int16)
        np.testing.assert_array_equal(coracle.error_bound_frame(orig, diff, mode, bound), diff)
        for c in range(3):
            np.testing.assert_array_equal(O.error_bound(orig[..., c], diff[..., c], mode, bound), diff[..., c])


def test_just_above_the_limit_the_quantiser_does_merge():
    """E = 0.5: neighbours d, d + 1 satisfy (d + 0.5) - (d + 1 - 0.5) = 0, not < 0 -- they merge; the shortcut's limit 0.499
    keeps well away from it."""
    orig = np.zeros((1, 8, 3), np.uint8)
    diff = np.tile(np.array([3, 4, 3, 4, 3, 4, 3, 4], np.int16)[None, :, None], (1, 1, 3))
    out = coracle.error_bound_frame(orig, diff, "abs", [0.5])
    assert not (out == diff).all()


@pytest.mark.parametrize("name", [str(n) for n in R4["run_names"]])
def test_the_reference_itself_is_lossless_at_these_tolerances(name):
    """Six executions of the reference's compress.run / decompress.run (fake predictor) with abs 0.255 / 0.3 / 0.499,
    rel 1e-3 / 0.0019, absrel [0.4, 0.5]: its decoded images are its input images."""
    pre = "run_%s_" % name
    frames, dec = R4[pre + "frames"], R4[pre + "decoded"]
    if frames.ndim == 3:   # grayscale input: decompress.py:272-278 saves RGB
        frames = np.repeat(frames[..., None], 3, -1)
    np.testing.assert_array_equal(dec, frames)
    mode, bound = str(R4[pre + "mode"]), R4[pre + "bound"].tolist()
    worst = {"abs": lambda b: abs(b[0]), "rel": lambda b: 255.0 * b[0], "absrel": lambda b: min(abs(b[0]), 255.0 * b[1])}[mode](bound)
    assert 0.0 < worst <= 0.499
