#!/opt/conda/bin/python3.9
"""Writes tests/golden/keras_style_model/{prednet_model.json,prednet_weights.hdf5} with REAL h5py
(conda python3.9 in the build container) in the layout Keras 2.2.4 `ModelCheckpoint` produces
(train.py:109): /model_weights/<layer>/<layer>/layer_<key>_<level>/{kernel:0,bias:0}, plus the
attributes Keras adds.  The weights are PredNetConfig((3,16)).init_weights(seed=77, bias 0.1),
so tests can compare what the pure-Python reader returns.  Run:
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py"""
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tezip_amd.prednet import PredNetConfig  # noqa: E402
from tezip_amd import weights as W  # noqa: E402

cfg = PredNetConfig(stack_sizes=(3, 16))
ws = cfg.init_weights(seed=77, bias_scale=0.1)
out = os.path.join(HERE, "keras_style_model")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, W.JSON_NAME), "w").write(W.make_model_json(cfg, 16, 24))
with h5py.File(os.path.join(out, W.H5_NAME), "w") as f:
    f.attrs["keras_version"] = np.string_("2.2.4")
    f.attrs["backend"] = np.string_("tensorflow")
    mw = f.create_group("model_weights")
    mw.attrs["layer_names"] = [np.string_(n) for n in ("input_1", "pred_net_1", "time_distributed_1")]
    mw.create_group("input_1").attrs["weight_names"] = np.array([], dtype="S1")
    g = mw.create_group("pred_net_1")
    names = []
    for (n, shape), w in zip(cfg.weight_shapes(), ws):
        key, kind = n.split("/")
        stem = key.rstrip("0123456789")
        path = "pred_net_1/layer_%s_%s/%s:0" % (stem, key[len(stem):], kind)
        names.append(np.string_(path))
        g.create_dataset(path, data=w)
    g.attrs["weight_names"] = names
    td = mw.create_group("time_distributed_1")
    td.attrs["weight_names"] = [np.string_("time_distributed_1/kernel:0")]
    td.create_dataset("time_distributed_1/kernel:0", data=np.array([[1.0], [0.0], [0.0]], np.float32))
    f.create_group("optimizer_weights")
print("wrote", out, os.path.getsize(os.path.join(out, W.H5_NAME)), "bytes")
