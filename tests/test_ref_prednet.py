"""The reference's OWN PredNet class (prednet.py build / get_initial_state / step), executed in the build container
over numpy stand-ins for the Keras primitives (tests/golden/make_golden.py `_prednet_fixture`,
tests/golden/keras_standin.py), against the three predictor restatements in oracle/ -- numpy einsum, torch conv2d and
the canonical C oracle under BOTH arithmetic contracts (TZ-PA1 direct chains, TZ-PA2 Winograd).

Pinned by this: the wiring of /root/reference/src/prednet.py:143-190 (state list and shapes), :192-233 (which
convolutions exist, their shapes, the order `trainable_weights` puts them in -- the order compress.py:163 hands the
weight list over in) and :235-308 (the step).  NOT pinned: what Keras 2.2.4's Conv2D / hard_sigmoid / UpSampling2D /
MaxPooling2D compute; the stand-ins state those from knowledge.  Tolerance 2e-5 (float32 summation
order; the fixture accumulates every convolution in float64)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import coracle, prednet_np
from tezip_amd.prednet import PredNetConfig

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "ref_prednet.npz"))
CASES = [str(c) for c in FIX["pn_cases"]]
ATOL = 2e-5


def case(name):
    pre = "pn_%s_" % name
    stack = tuple(int(v) for v in FIX[pre + "stack"])
    hp, wp = (int(v) for v in FIX[pre + "hw"])
    seed, bias = FIX[pre + "wseed_bias"]
    cfg = PredNetConfig(stack_sizes=stack)
    w = cfg.init_weights(seed=int(seed), bias_scale=float(bias))
    return pre, cfg, hp, wp, w


def states(pre, t, L):
    return [FIX[pre + "t%d_%s%d" % (t, u, l)] for u in "rce" for l in range(L)]


@pytest.mark.parametrize("name", CASES)
def test_weight_list_is_the_order_build_leaves_trainable_weights_in(name):
    """prednet.py:212-227: sorted(conv_layers.keys()) = a, ahat, c, f, i, o; per level [kernel, bias]."""
    pre, cfg, hp, wp, w = case(name)
    # the values the fixture was computed with are the ones this numpy draws
    assert hashlib.sha256(b"".join(x.tobytes() for x in w)).hexdigest() == str(FIX[pre + "weights_sha256"])
    names = [str(n) for n in FIX[pre + "weight_names"]]
    shapes = [tuple(int(v) for v in row if v) for row in FIX[pre + "weight_shapes"]]
    mine = cfg.weight_shapes()
    assert len(mine) == len(names)
    for (my_name, my_shape), ref_name, ref_shape in zip(mine, names, shapes):
        key, kind = my_name.split("/")                      # "ahat0/kernel" vs the reference's "layer_ahat_0/kernel"
        assert ref_name == "layer_%s_%s/%s" % (key.rstrip("0123456789"), key[len(key.rstrip("0123456789")):], kind)
        assert my_shape == ref_shape, (my_name, my_shape, ref_shape)


@pytest.mark.parametrize("name", CASES)
def test_initial_state_layout(name):
    """prednet.py:157-181: r then c then e, level by level, (1, Hp>>l, Wp>>l, channels), all zero."""
    pre, cfg, hp, wp, _ = case(name)
    L, st, rs = cfg.nb_layers, cfg.stack_sizes, cfg.R_stack_sizes
    expect = [[1, hp >> l, wp >> l, rs[l]] for l in range(L)] * 2 + [[1, hp >> l, wp >> l, 2 * st[l]] for l in range(L)]
    assert FIX[pre + "state_shapes"].tolist() == expect


@pytest.mark.parametrize("name", CASES)
def test_numpy_restatement(name):
    pre, cfg, hp, wp, w = case(name)
    X = FIX[pre + "X"][0]
    out, st = prednet_np.predict(w, cfg.stack_sizes, cfg.R_stack_sizes, X, return_states=True)
    np.testing.assert_allclose(out, FIX[pre + "X_hat"][0], atol=ATOL, rtol=0)
    for mine, ref in zip(st, states(pre, len(X), cfg.nb_layers)):
        np.testing.assert_allclose(mine, ref, atol=ATOL, rtol=0)
    if name.startswith("small"):                            # every intermediate step too
        for t in range(1, len(X)):
            _, st = prednet_np.predict(w, cfg.stack_sizes, cfg.R_stack_sizes, X[:t], return_states=True)
            for mine, ref in zip(st, states(pre, t, cfg.nb_layers)):
                np.testing.assert_allclose(mine, ref, atol=ATOL, rtol=0)


@pytest.mark.parametrize("name", CASES)
def test_torch_restatement(name):
    from oracle import prednet_torch
    pre, cfg, hp, wp, w = case(name)
    X = FIX[pre + "X"][0]
    tn = prednet_torch.TorchPredNet(w, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)
    out, st = tn.predict_seq(X, return_states=True)
    np.testing.assert_allclose(out, FIX[pre + "X_hat"][0], atol=ATOL, rtol=0)
    for mine, ref in zip(st, states(pre, len(X), cfg.nb_layers)):
        np.testing.assert_allclose(mine, ref, atol=ATOL, rtol=0)


@pytest.mark.parametrize("contract", [1, 2])
@pytest.mark.parametrize("name", [c for c in CASES if c != "small_t3"])
def test_c_oracle_both_contracts(name, contract):
    """The canonical oracle (what the HIP kernels are bit-exact with): the t0 constant image = X_hat[0,0], the
    prediction = X_hat[0,1] (compress.py:195, 227-229) and the per-level r, c, e of the second step, under TZ-PA1 and
    TZ-PA2 alike."""
    pre, cfg, hp, wp, w = case(name)
    L = cfg.nb_layers
    net = coracle.CPredNet(w, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp).set_contract(contract)
    X, X_hat = FIX[pre + "X"][0], FIX[pre + "X_hat"][0]
    np.testing.assert_allclose(net.c0(), X_hat[0], atol=ATOL, rtol=0)
    out, taps = net.next(X[0], debug=True)
    np.testing.assert_allclose(out, X_hat[1], atol=ATOL, rtol=0)
    ref = states(pre, 2, L)
    for l in range(L):
        np.testing.assert_allclose(taps["r"][l], ref[l], atol=ATOL, rtol=0)
        np.testing.assert_allclose(taps["c"][l], ref[L + l], atol=ATOL, rtol=0)
    # the oracle's e tap is the live one: the errors of step one, against the real frame, which the gates of step two
    # read (the e of step two is taken against the zero frame of compress.py:225-226 and feeds nothing)
    for l in range(L):
        np.testing.assert_allclose(taps["e"][l], FIX[pre + "t1_e%d" % l], atol=ATOL, rtol=0)


def test_c_oracle_rollout_feeds_on_the_reference_prediction():
    """compress.py:218-229: the next input is the previous X_hat[0,1].  Two chained predictions through the oracle from
    the fixture's frame equal two chained evaluations of the reference-wired numpy restatement (itself held to the
    fixture above) -- the recursion adds nothing the fixture does not cover."""
    pre, cfg, hp, wp, w = case("small")
    net = coracle.CPredNet(w, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp).set_contract(1)
    f = FIX[pre + "X"][0, 0]
    p1 = net.next(f)
    p2 = net.next(p1)
    q1 = prednet_np.predict(w, cfg.stack_sizes, cfg.R_stack_sizes, np.stack([f, np.zeros_like(f)]))[1]
    q2 = prednet_np.predict(w, cfg.stack_sizes, cfg.R_stack_sizes, np.stack([q1, np.zeros_like(f)]))[1]
    np.testing.assert_allclose(p1, FIX[pre + "X_hat"][0, 1], atol=ATOL, rtol=0)
    np.testing.assert_allclose(p2, q2, atol=4e-5, rtol=0)
