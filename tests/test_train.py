"""Trainer (§8f-2) on CPU: the torch model is the same function as the oracle's PredNet, the L_0
loss follows train.py:56-72, a few epochs reduce it, and the saved model directory is what
compress.run reads."""
import os

import numpy as np
import pytest

from oracle import coracle
from tezip_amd import train, train_data_create, weights
from tezip_amd.prednet import PredNetConfig

SMALL = PredNetConfig(stack_sizes=(3, 16, 32))


def test_torch_model_is_the_same_function_as_the_oracle():
    import torch
    rng = np.random.default_rng(1)
    w = SMALL.init_weights(seed=5, bias_scale=0.2)
    net = coracle.CPredNet(w, SMALL.stack_sizes, SMALL.R_stack_sizes, 16, 24)
    m = train.build_model(SMALL, w)
    got_list = m.keras_list()
    assert all((a == b).all() for a, b in zip(got_list, w))  # layout round trip HWIO <-> OIHW
    f = rng.integers(0, 256, (16, 24, 3)).astype(np.float32) / np.float32(255)
    x = torch.from_numpy(np.stack([f, np.zeros_like(f)])).permute(0, 3, 1, 2)[None]
    with torch.no_grad():
        pred = m(x, output="prediction")[0].permute(0, 2, 3, 1).numpy()
        err = m(x)[0].numpy()
    np.testing.assert_allclose(pred[0], net.c0(), atol=2e-5)
    np.testing.assert_allclose(pred[1], net.next(f), atol=2e-5)
    _, dbg = net.next(f, debug=True)
    np.testing.assert_allclose(err[0, 0], dbg["e"][0].mean(), rtol=1e-4)  # t=0 error-unit mean of level 0


def test_l0_loss_weights():
    import torch
    e = torch.tensor([[[0.5, 9.0, 9.0], [0.25, 9.0, 9.0]]])  # (B=1, T=2, L=3)
    assert float(train.l0_loss(e, 2)) == 0.25  # only level 0, only t>=1 (train.py:56-60)
    # data_utils.py:30 iterates range(N - nt): the last valid start (3) is never offered
    assert train.possible_starts(np.array(["a", "a", "a", "b", "b"]), 2).tolist() == [0, 1]


def test_data_builder_and_short_training_run(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    yy, xx = np.meshgrid(np.arange(21), np.arange(30), indexing="ij")
    for s in range(3):
        d = tmp_path / "raw" / ("seq%d" % s)
        d.mkdir(parents=True)
        for t in range(6):
            img = np.stack([120 + 80 * np.sin((xx + 3 * t + 5 * s) / 4.0), 100 + 60 * np.cos((yy - 2 * t) / 3.0),
                            128 + 0 * xx], -1) + rng.normal(0, 2, (21, 30, 3))
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(d / ("f%02d.png" % t))
    out = str(tmp_path / "set")
    train_data_create.process_data(str(tmp_path / "raw"), out, val_folders=["seq2"])
    X = np.load(os.path.join(out, "X_train.npy"))
    assert X.shape == (12, 24, 32, 3) and (X[:, 21:] == 0).all() and (X[:, :, 30:] == 0).all()
    assert np.load(os.path.join(out, "sources_val.npy")).tolist() == ["val-seq2"] * 6
    mdir = str(tmp_path / "model")
    hist = train.run(mdir, out, False, nb_epoch=6, samples_per_epoch=4, stack_sizes=(3, 16), device="cpu")
    assert hist[-1][0] < hist[0][0] and np.isfinite(hist).all()
    cfg, w, shape = weights.load_model(mdir)
    assert cfg.stack_sizes == (3, 16) and shape == (24, 32) and len(w) == 22
