"""Model directory I/O.

The reference keeps a model as `prednet_model.json` (Keras `model.to_json()`, train.py:114-117)
plus `prednet_weights.hdf5` (a full-model Keras checkpoint, train.py:109) and rebuilds the
PredNet layer from them (compress.py:143-173).  Here:
  * prednet_model.json is read the same way (the PredNet layer's config gives the channel
    stacks, the InputLayer's batch_input_shape the padded frame size);
  * weights are read from `prednet_weights.npz` (this build's native format: arrays
    w000..wNNN in the Keras weight-list order of prednet.py:210-227), or from the reference's
    `prednet_weights.hdf5` -- with h5py when it is importable, otherwise with the built-in
    minimal reader tezip_amd/h5lite.py (h5py is not part of the image this was built in).
"""
import json
import os

import numpy as np

from .prednet import PredNetConfig

JSON_NAME = "prednet_model.json"
H5_NAME = "prednet_weights.hdf5"
NPZ_NAME = "prednet_weights.npz"


def _find_layers(model_json):
    cfg = model_json.get("config", model_json)
    layers = cfg.get("layers", []) if isinstance(cfg, dict) else cfg
    return layers


def parse_model_json(text):
    """-> (PredNetConfig, (Hp, Wp) or None)."""
    mj = json.loads(text)
    pred, shape = None, None
    for layer in _find_layers(mj):
        cname = layer.get("class_name")
        lc = layer.get("config", {})
        if cname == "PredNet":
            pred = lc
        elif cname == "InputLayer" and shape is None:
            bis = lc.get("batch_input_shape")
            if bis and len(bis) == 5:
                shape = (bis[2], bis[3])
    if pred is None and mj.get("class_name") == "PredNet":
        pred = mj.get("config", {})
    if pred is None:
        raise ValueError("no PredNet layer in the model json")
    fmt = pred.get("data_format", pred.get("dim_ordering", "channels_last"))
    if fmt not in ("channels_last", "tf"):
        raise NotImplementedError("only channels_last models are supported")
    cfg = PredNetConfig(pred["stack_sizes"], pred.get("R_stack_sizes"), pred.get("A_filt_sizes"),
                        pred.get("Ahat_filt_sizes"), pred.get("R_filt_sizes"), pred.get("pixel_max", 1.0))
    return cfg, shape


def make_model_json(cfg, hp, wp, nt=2):
    """A minimal Keras-style model json that parse_model_json (and the reference's field
    accesses, compress.py:163-169) understand."""
    return json.dumps({"class_name": "Model", "config": {"name": "model_1", "layers": [
        {"class_name": "InputLayer", "name": "input_1",
         "config": {"batch_input_shape": [None, nt, hp, wp, cfg.stack_sizes[0]], "dtype": "float32", "name": "input_1"}},
        {"class_name": "PredNet", "name": "pred_net_1",
         "config": dict(cfg.to_json_dict()["config"], output_mode="error", return_sequences=True)}]},
        "keras_version": "2.2.4", "backend": "tensorflow"})


def save_model(model_dir, cfg, weights, hp, wp):
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, JSON_NAME), "w") as f:
        f.write(make_model_json(cfg, hp, wp))
    np.savez(os.path.join(model_dir, NPZ_NAME), **{"w%03d" % i: np.asarray(w, np.float32) for i, w in enumerate(weights)})


def _load_h5(path, cfg):
    import h5py  # optional
    names = [n for n, _ in cfg.weight_shapes()]
    out = []
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        layer = None
        for k in g.keys():
            if "pred_net" in k.lower() or "prednet" in k.lower():
                layer = g[k]
        if layer is None:
            raise ValueError("no PredNet layer group in %s" % path)
        found = {}

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                found[name] = np.array(obj, dtype=np.float32)
        layer.visititems(visit)
        for n in names:  # 'a0/kernel' -> '.../layer_a_0/kernel:0'
            key, kind = n.split("/")
            gate, lvl = key.rstrip("0123456789"), key[len(key.rstrip("0123456789")):]
            want = "layer_%s_%s" % (gate, lvl)
            hit = [v for k, v in found.items() if want in k and kind in k.split("/")[-1]]
            if len(hit) != 1:
                raise ValueError("cannot locate %s in %s" % (n, path))
            out.append(hit[0])
    return out


def load_model(model_dir):
    """-> (PredNetConfig, weights list, (Hp, Wp) or None). Raises FileNotFoundError/OSError like
    the reference's open()/load_weights() do (callers print the reference's messages)."""
    with open(os.path.join(model_dir, JSON_NAME), "r") as f:
        cfg, shape = parse_model_json(f.read())
    npz = os.path.join(model_dir, NPZ_NAME)
    h5 = os.path.join(model_dir, H5_NAME)
    if os.path.exists(npz):
        z = np.load(npz)
        weights = [z["w%03d" % i] for i in range(len(z.files))]
    elif os.path.exists(h5):
        try:
            weights = _load_h5(h5, cfg)
        except ImportError:  # no h5py in this image: the built-in reader covers Keras' files
            from . import h5lite
            weights = h5lite.load_prednet_weights(h5, [n for n, _ in cfg.weight_shapes()])
    else:
        raise OSError("No such file or directory: %s" % h5)
    shapes = cfg.weight_shapes()
    if len(weights) != len(shapes) or any(tuple(w.shape) != s for w, (_, s) in zip(weights, shapes)):
        raise ValueError("weights do not match the model json")
    return cfg, weights, shape


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 3 and sys.argv[1] == "convert":
        d = sys.argv[2]
        with open(os.path.join(d, JSON_NAME)) as f:
            c, s = parse_model_json(f.read())
        w = _load_h5(os.path.join(d, H5_NAME), c)
        np.savez(os.path.join(d, NPZ_NAME), **{"w%03d" % i: x for i, x in enumerate(w)})
        print("wrote", os.path.join(d, NPZ_NAME))
    else:
        print("usage: python -m tezip_amd.weights convert <model_dir>")
