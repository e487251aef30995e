"""HIP predictor against the reference's OWN PredNet class, with no oracle in between.

tests/golden/ref_prednet.npz was computed by /root/reference/src/prednet.py itself (build, get_initial_state, step run
over numpy stand-ins for the Keras primitives: tests/golden/make_golden.py `_prednet_fixture`).  Here the C ABI's
tz_predict_c0 / tz_predict_next / tz_predict_tap are held to it at 2e-5 under BOTH arithmetic contracts: TZ-PA1 (direct
fmaf chains) and TZ-PA2 (Winograd F(2x2,3x3) for levels >= 1).  Tolerance: float32 summation order (the fixture
accumulates each convolution in float64); what the weight list order means is exactly what this test would trip on --
the gate kernels i, f, o, c of a level have one shape, so only the outputs can tell a wrong order.
Replaces on the HIP side: prednet.py:143-190, 192-233, 235-308; compress.py:163-173, 195, 224-229."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(GOLDEN, "ref_prednet.npz"))
ATOL = 2e-5


@pytest.fixture(scope="module")
def ctx():
    from tezip_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("contract", [1, 2])
@pytest.mark.parametrize("name", ["small", "full64", "full72x88"])
def test_hip_predictor_matches_the_reference_class(ctx, name, contract):
    pre = "pn_%s_" % name
    stack = tuple(int(v) for v in FIX[pre + "stack"])
    hp, wp = (int(v) for v in FIX[pre + "hw"])
    seed, bias = FIX[pre + "wseed_bias"]
    cfg = PredNetConfig(stack_sizes=stack)
    L = cfg.nb_layers
    w = cfg.init_weights(seed=int(seed), bias_scale=float(bias))   # sha256 checked in tests/test_ref_prednet.py
    ctx.load_model(cfg, w)
    ctx.prepare(hp, wp, max_batch=1)
    ctx.set_contract(contract)
    try:
        assert ctx.get_contract() == contract
        X, X_hat = FIX[pre + "X"][0], FIX[pre + "X_hat"][0]
        np.testing.assert_allclose(ctx.predict_c0(), X_hat[0], atol=ATOL, rtol=0)             # compress.py:195
        np.testing.assert_allclose(ctx.predict_next(X[:1])[0], X_hat[1], atol=ATOL, rtol=0)   # compress.py:227-229
        for l in range(L):
            # e of step one (errors against the real frame) and r of step two, per level
            np.testing.assert_allclose(ctx.predict_tap(0, l), FIX[pre + "t1_e%d" % l], atol=ATOL, rtol=0,
                                       err_msg="e level %d" % l)
            np.testing.assert_allclose(ctx.predict_tap(1, l), FIX[pre + "t2_r%d" % l], atol=ATOL, rtol=0,
                                       err_msg="r level %d" % l)
    finally:
        ctx.set_contract(0)
