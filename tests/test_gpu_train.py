"""§8f rows on the MI355X, through the reference-shaped entry points only:
  train_data_create (folder of sequence folders -> padded uint8 stacks)           f4
  tezip.py -l  (tezip_amd/train.py on the GPU: L_0 loss, Adam schedule, best ckpt) f2
  -> model directory = prednet_model.json + Keras-layout prednet_weights.hdf5      f1 ("and back")
  tezip.py -c / -u with that model: lossless round trip, and the trained model predicts better
  than the untrained one (smaller entropy stream for the same frames)."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from tezip_amd import synth, tezip, train_data_create, weights, zstd
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu


def _cli(argv):
    buf = io.StringIO()
    with redirect_stdout(buf):
        tezip.main(tezip.build_parser().parse_args(argv))
    return buf.getvalue()


def _write_pngs(folder, frames):
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    for t in range(frames.shape[0]):
        Image.fromarray(frames[t]).save(os.path.join(folder, "f%03d.png" % t))


def test_learn_then_compress_then_uncompress_on_the_gpu(tmp_path):
    import torch
    from PIL import Image
    raw = tmp_path / "raw"
    for s in range(5):
        _write_pngs(str(raw / ("seq%d" % s)), synth.turbulence(10, 61, 90, seed=200 + s))  # pads to 64 x 96
    data = str(tmp_path / "set")
    train_data_create.process_data(str(raw), data, val_folders=["seq4"])
    from tezip_amd import hkl
    X = hkl.load(os.path.join(data, "X_train.hkl"))
    assert X.shape == (40, 64, 96, 3) and (X[:, 61:] == 0).all() and (X[:, :, 90:] == 0).all()
    mdir = str(tmp_path / "model")
    out = _cli(["-l", mdir, data, "-v"])                       # tezip.py -l model dir (train.py:18)
    assert out.splitlines()[:2] == ["GPU MODE", "train mode"] and "Epoch 100/100" in out
    losses = [float(l.split("loss:")[1].split("-")[0]) for l in out.splitlines() if l.startswith("Epoch")]
    assert len(losses) == 100 and np.mean(losses[-5:]) < 0.5 * np.mean(losses[:5])
    assert torch.cuda.is_available()   # train.run picks the GPU when there is one
    # the reference's two files, in Keras' layout
    assert sorted(os.listdir(mdir)) == ["prednet_model.json", "prednet_weights.hdf5"]
    cfg, trained, shape = weights.load_model(mdir)
    assert cfg.stack_sizes == (3, 48, 96, 192) and shape == (64, 96) and len(trained) == 46
    # compress / uncompress a held-out sequence of the same size with the trained model
    test = synth.turbulence(12, 61, 90, seed=300)
    ddir = str(tmp_path / "imgs")
    _write_pngs(ddir, test)
    cdir, udir = str(tmp_path / "comp"), str(tmp_path / "out")
    out = _cli(["-c", mdir, ddir, cdir, "-p", "0", "-w", "6", "-m", "abs", "-b", "0"])
    assert "compress mode" in out and sorted(os.listdir(cdir)) == ["entropy.dat", "filename.txt", "key_frame.dat", "tezip_amd.json"]
    _cli(["-u", mdir, cdir, udir])
    got = np.stack([np.array(Image.open(os.path.join(udir, "f%03d.png" % t))) for t in range(12)])
    assert np.array_equal(got, test)                            # lossless
    # ... and with the weights `-l` just trained the files hold exactly the ORACLE's streams (trained weights are not the
    # near-constant glorot predictions the other parity tests see: the whole predictor + codec chain on a real model)
    from oracle import coracle
    from oracle import oracle as O

    class P:
        net = coracle.CPredNet(trained, cfg.stack_sizes, cfg.R_stack_sizes, 64, 96)

        def c0(self, a, b):
            return self.net.c0()

        def next(self, f):
            return self.net.next(np.asarray(f, np.float32))

    for mode, bound, sub in (("abs", [0.0], cdir), ("abs", [2.0], str(tmp_path / "comp_abs2"))):
        if sub != cdir:
            _cli(["-c", mdir, ddir, sub, "-p", "0", "-w", "6", "-m", mode, "-b", "2"])
        ref = O.compress_oracle(test, 0, 6, None, mode, bound, P(), True)
        stream = np.frombuffer(zstd.decompress(open(os.path.join(sub, "entropy.dat"), "rb").read()), "<i2")
        keyb = np.frombuffer(zstd.decompress(open(os.path.join(sub, "key_frame.dat"), "rb").read()), np.uint8)
        np.testing.assert_array_equal(stream, ref["stream"], err_msg="pre-zstd entropy stream, %s %r" % (mode, bound))
        np.testing.assert_array_equal(keyb, ref["key_frame"])
    assert int(np.abs(np.diff(P.net.next(coracle.u8_to_f32_frame(test[0], 64, 96)), axis=1)).max() * 255) > 3  # not a constant image
    # the same frames with untrained weights of the same architecture compress worse
    rdir = str(tmp_path / "random_model")
    weights.save_model(rdir, cfg, cfg.init_weights(seed=123), 64, 96)
    c2 = str(tmp_path / "comp_random")
    _cli(["-c", rdir, ddir, c2, "-p", "0", "-w", "6", "-m", "abs", "-b", "0"])

    def size(d):
        blob = open(os.path.join(d, "entropy.dat"), "rb").read()
        return len(zstd.decompress(blob)), len(blob)

    assert abs(size(cdir)[0] - size(c2)[0]) < 2 * 1021 + 2   # same payload length, tables differ
    assert size(cdir)[1] < size(c2)[1]
