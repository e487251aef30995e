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


def test_autograd_gradient_equals_finite_differences_of_the_oracle_loss():
    """The trainer's gradient (torch autograd through train.build_model) against central finite differences of the SAME loss
    evaluated by the C ORACLE's predictor (train.py:56-72 at nt = 2: time weights [0, 1], layer weights [1, 0, ..] =>
    L_0 = mean of the level-0 error units at t = 1 = mean |X_hat[1] - x[1]| / 2, prednet.py:286-301): for every one of the
    34 weight arrays of the SMALL model, the entry with the largest gradient.  rtol 1e-2 (+ 2e-7: float32 noise of the
    oracle's prediction / the step)."""
    import torch
    rng = np.random.default_rng(11)
    hp, wp = 16, 24
    w = SMALL.init_weights(seed=5, bias_scale=0.2)
    yy, xx = np.meshgrid(np.arange(hp), np.arange(wp), indexing="ij")
    f0 = (np.stack([120 + 80 * np.sin(xx / 4.0), 100 + 60 * np.cos(yy / 3.0), 128 + 0 * xx], -1)
          + rng.normal(0, 4, (hp, wp, 3))).clip(0, 255).astype(np.uint8)
    x = train.sample(np.stack([f0, np.roll(f0, 2, axis=1)]), 0, 2)
    m = train.build_model(SMALL, w)
    loss = train.l0_loss(m(torch.from_numpy(x).permute(0, 3, 1, 2)[None]), 2)
    loss.backward()

    def oracle_loss(ws):
        net = coracle.CPredNet(ws, SMALL.stack_sizes, SMALL.R_stack_sizes, hp, wp)
        return 0.5 * np.abs(net.next(x[0]).astype(np.float64) - x[1].astype(np.float64)).mean()

    assert float(loss.detach()) == pytest.approx(oracle_loss(w), rel=1e-6)
    grads = []
    for mod in m._ordered():   # Keras list order, HWIO
        grads += [mod.weight.grad.permute(2, 3, 1, 0).contiguous().numpy(), mod.bias.grad.numpy().copy()]
    assert len(grads) == len(w) == 34
    for k, g in enumerate(grads):
        idx = np.unravel_index(np.argmax(np.abs(g)), g.shape)
        assert abs(g[idx]) > 0, k
        up, dn = [a.copy() for a in w], [a.copy() for a in w]
        up[k][idx] += 1e-3
        dn[k][idx] -= 1e-3
        fd = (oracle_loss(up) - oracle_loss(dn)) / (float(up[k][idx]) - float(dn[k][idx]))
        assert abs(fd - g[idx]) <= 1e-2 * abs(g[idx]) + 2e-7, (k, idx, float(g[idx]), fd)


def _ref_train():
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "ref_train.npz"))


def _per_folder(X, sources):
    """{label: stack of that folder's images} from a (N, Hp, Wp, 3) stack and its N source labels; also checks that a
    folder's images are consecutive."""
    out, labels = {}, [str(x) for x in sources]
    for lab in dict.fromkeys(labels):
        idx = [i for i, v in enumerate(labels) if v == lab]
        assert idx == list(range(idx[0], idx[0] + len(idx))), lab
        out[lab] = np.asarray(X[idx[0]: idx[0] + len(idx)])
    return out


def _write_tree(root, g):
    from PIL import Image
    for folder in g["tr_folders"].tolist():
        d = root / folder
        d.mkdir(parents=True)
        imgs = g["tr_img_" + folder]
        for i in range(imgs.shape[0]):
            Image.fromarray(imgs[i], mode="L" if imgs.ndim == 3 else "RGB").save(d / ("im_%02d.png" % i))


@pytest.mark.parametrize("tag", ["v", "r"])
def test_data_builder_equals_the_reference(tmp_path, tag, capsys):
    """tezip_amd.train_data_create.process_data against the REFERENCE's process_data (train_data_create.py:11-95), run by
    tests/golden/make_golden.py on the same PNG tree (five folders of different image sizes, one of them grayscale) with
    hickle.dump captured: padded size, the stack of every folder (zero padding, sorted file order, RGB conversion) and
    its source label `<split>-<folder>`.  The ORDER of the folders inside a split is what the reference leaves to set()
    iteration / os.listdir (train_data_create.py:23,14), so the comparison is per folder.  tag v: validation folders
    given (-v); tag r: the reference's random split, whose pick is passed to ours."""
    from tezip_amd import hkl
    g = _ref_train()
    _write_tree(tmp_path / "data", g)
    ref = {sp: _per_folder(g["tr_%s_X_%s" % (tag, sp)], g["tr_%s_src_%s" % (tag, sp)]) for sp in ("train", "val")}
    val = [lab.split("-", 1)[1] for lab in ref["val"]]
    assert len(val) == (2 if tag == "v" else 1)            # int(5 / 10) < 1 -> one validation folder (train_data_create.py:26-28)
    out = str(tmp_path / "set")
    train_data_create.process_data(str(tmp_path / "data"), out, val_folders=[str(tmp_path / "data" / v) for v in val])
    for sp in ("train", "val"):
        X = hkl.load(os.path.join(out, "X_%s.hkl" % sp))
        src = hkl.load(os.path.join(out, "sources_%s.hkl" % sp))
        assert X.dtype == np.uint8 and X.shape == g["tr_%s_X_%s" % (tag, sp)].shape
        assert sorted(src) == sorted(g["tr_%s_src_%s" % (tag, sp)].tolist())
        mine = _per_folder(X, src)
        assert set(mine) == set(ref[sp])
        for lab in mine:
            np.testing.assert_array_equal(mine[lab], ref[sp][lab], err_msg=lab)
    # the messages of train_data_create.py:75,99-100
    log, ref_log = capsys.readouterr().out, str(g["tr_log"])
    for line in ("Creating train data: %d images" % len(g["tr_%s_src_train" % tag]), "After Padding ：height: 16  width: 24"):
        assert line in log and line in ref_log, line


def test_possible_starts_and_samples_equal_the_reference_generator():
    """train.possible_starts / train.sample against data_utils.SequenceGenerator of the REFERENCE (data_utils.py:28-45,
    58-71; fixture ref_train.npz): both start modes at nt = 2, 3, 5 on the reference's own source list (three folders of
    6, 7 and 4 images), the N_seq cut of train.py:90, and the first two batches of a non-shuffled generator."""
    g = _ref_train()
    X, src = g["tr_v_X_train"], g["tr_v_src_train"].tolist()
    for nt in (2, 3, 5):
        assert train.possible_starts(src, nt).tolist() == g["sg_all_nt%d" % nt].tolist()
        assert train.possible_starts(src, nt, "unique").tolist() == g["sg_unique_nt%d" % nt].tolist()
    assert train.possible_starts(src, 2, N_seq=2).tolist() == g["sg_nseq2"].tolist()
    assert g["sg_im_shape"].tolist() == list(X.shape[1:])
    starts = train.possible_starts(src, 3)
    b0 = np.stack([train.sample(X, int(starts[0]), 3), train.sample(X, int(starts[1]), 3)])
    b1 = np.stack([train.sample(X, int(starts[2]), 3), train.sample(X, int(starts[3]), 3)])
    assert b0.dtype == np.float32 and b0.tobytes() == g["sg_batch0_x"].tobytes()
    assert b1.tobytes() == g["sg_batch1_x"].tobytes()
    assert (g["sg_batch0_y"] == 0).all()                   # output_mode 'error': the target is zero (data_utils.py:62-63)
    with pytest.raises(ValueError):
        train.possible_starts(src, 2, "some")


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
