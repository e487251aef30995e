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
    from tezip_amd import hkl
    assert sorted(os.listdir(out)) == ["X_train.hkl", "X_val.hkl", "sources_train.hkl", "sources_val.hkl"]  # train.py:24-27
    X = hkl.load(os.path.join(out, "X_train.hkl"))
    assert X.dtype == np.uint8 and X.shape == (12, 24, 32, 3) and (X[:, 21:] == 0).all() and (X[:, :, 30:] == 0).all()
    assert hkl.load(os.path.join(out, "sources_val.hkl")) == ["val-seq2"] * 6
    mdir = str(tmp_path / "model")
    hist = train.run(mdir, out, False, nb_epoch=6, samples_per_epoch=4, stack_sizes=(3, 16), device="cpu")
    assert hist[-1][0] < hist[0][0] and np.isfinite(hist).all()
    cfg, w, shape = weights.load_model(mdir)
    assert cfg.stack_sizes == (3, 16) and shape == (24, 32) and len(w) == 22


def test_hickle_layout_files_open_with_real_h5py(tmp_path):
    """X_*.hkl / sources_*.hkl written by tezip_amd/hkl.py, opened by libhdf5 through h5py (conda
    interpreter of the build container; hickle itself is installable nowhere here, so its layout --
    /data dataset + HICKLE_VERSION / base_type / type attributes -- is restated, parity unpinned)."""
    import subprocess
    from tezip_amd import hkl
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py on this machine")
    X = np.random.default_rng(5).integers(0, 256, (9, 16, 24, 3)).astype(np.uint8)
    names = ["train-a"] * 5 + ["train-longer-name"] * 4
    hkl.dump(X, str(tmp_path / "X.hkl"))
    hkl.dump(names, str(tmp_path / "s.hkl"))
    code = ("import h5py, numpy as np, sys, hashlib\n"
            "f = h5py.File(sys.argv[1], 'r'); d = f['data']\n"
            "print(f.attrs['HICKLE_VERSION'].decode(), d.attrs['base_type'].decode(), d.dtype, d.shape, hashlib.sha1(np.asarray(d).tobytes()).hexdigest())\n"
            "g = h5py.File(sys.argv[2], 'r')['data']\n"
            "print(g.attrs['base_type'].decode(), g.attrs['str_type'].decode(), '|'.join(x.decode() for x in np.asarray(g)))\n")
    r = subprocess.run([py, "-c", code, str(tmp_path / "X.hkl"), str(tmp_path / "s.hkl")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    import hashlib
    l1, l2 = r.stdout.strip().splitlines()
    assert l1 == "4.0.1 ndarray uint8 (9, 16, 24, 3) " + hashlib.sha1(X.tobytes()).hexdigest()
    assert l2 == "list <class 'str'> " + "|".join(names)
    assert (hkl.load(str(tmp_path / "X.hkl")) == X).all() and hkl.load(str(tmp_path / "s.hkl")) == names
