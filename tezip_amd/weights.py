"""Model directory I/O.

The reference keeps a model as `prednet_model.json` (Keras `model.to_json()`, train.py:114-117)
plus `prednet_weights.hdf5` (a full-model Keras checkpoint, train.py:109) and rebuilds the
PredNet layer from them (compress.py:143-173).  Here:
  * prednet_model.json is read the same way (the PredNet layer's config gives the channel
    stacks, the InputLayer's batch_input_shape the padded frame size);
  * weights are read from the reference's `prednet_weights.hdf5` -- with h5py when it is
    importable, otherwise with the built-in minimal reader tezip_amd/h5lite.py (h5py is not part
    of the image this was built in) -- or, when there is no such file, from a legacy
    `prednet_weights.npz` (arrays w000..wNNN in the Keras weight-list order of
    prednet.py:210-227).  When both exist the hdf5 wins: it is the file the reference (re)writes.
  * `save_model` writes the same two files the reference's trainer leaves behind: a
    `model.to_json()`-shaped prednet_model.json (every layer of train.py:62-71 with the config keys
    Keras 2.2.4 emits, so that `model_from_json` can rebuild it) and a Keras-layout full-model
    prednet_weights.hdf5 (tezip_amd/h5lite.py writer), laid out for /root/reference/src/compress.py:143-173
    to consume.  **Parity unpinned**: neither Keras 2.2.4 nor h5py can be imported in the image this was
    built in and the reference holds no .hdf5 / .json fixture, so both layouts are restated from knowledge
    of that Keras version; what IS checked is that libhdf5 opens the files (tests/golden/check_h5_with_h5py.py,
    run by hand where h5py exists) and that this package reads back what it writes.  The .npz form stays
    as the documented fallback.
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


SUPPORTED_ACTIVATIONS = {"error_activation": "relu", "A_activation": "relu", "LSTM_activation": "tanh",
                         "LSTM_inner_activation": "hard_sigmoid"}


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
    # the kernels hard-code the reference's defaults (prednet.py:78-79, 95-98); a model that asks
    # for anything else must not be predicted with the wrong function
    for key, want in SUPPORTED_ACTIVATIONS.items():
        if pred.get(key, want) != want:
            raise NotImplementedError("%s=%r: the HIP predictor implements %r only" % (key, pred[key], want))
    if pred.get("extrap_start_time") is not None:
        raise NotImplementedError("extrap_start_time is not supported (the reference never sets it, train.py:62-64)")
    cfg = PredNetConfig(pred["stack_sizes"], pred.get("R_stack_sizes"), pred.get("A_filt_sizes"),
                        pred.get("Ahat_filt_sizes"), pred.get("R_filt_sizes"), pred.get("pixel_max", 1.0))
    return cfg, shape


def _dense_config(name, units=1):
    """Dense.get_config() of Keras 2.2.4 with default arguments."""
    return {"name": name, "trainable": False, "units": units, "activation": "linear", "use_bias": True,
            "kernel_initializer": {"class_name": "VarianceScaling",
                                   "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}},
            "bias_initializer": {"class_name": "Zeros", "config": {}},
            "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
            "kernel_constraint": None, "bias_constraint": None}


def make_model_json(cfg, hp, wp, nt=2):
    """prednet_model.json as `model.to_json()` writes it for the training graph of
    train.py:62-71 (Input -> PredNet(output_mode='error') -> TimeDistributed(Dense(1)) -> Flatten ->
    Dense(1)), Keras 2.2.4 functional-model schema.  The reference rebuilds the model from this
    with `model_from_json(..., custom_objects={'PredNet': PredNet})` and reads `layers[0]`'s
    batch_input_shape and `layers[1].get_config()` (compress.py:155-169)."""
    pred_cfg = {"name": "pred_net_1", "trainable": True, "return_sequences": True, "return_state": False,
                "go_backwards": False, "stateful": False, "unroll": False, "implementation": 0,
                "stack_sizes": list(cfg.stack_sizes), "R_stack_sizes": list(cfg.R_stack_sizes),
                "A_filt_sizes": list(cfg.A_filt_sizes), "Ahat_filt_sizes": list(cfg.Ahat_filt_sizes),
                "R_filt_sizes": list(cfg.R_filt_sizes), "pixel_max": float(cfg.pixel_max),
                "error_activation": "relu", "A_activation": "relu", "LSTM_activation": "tanh",
                "LSTM_inner_activation": "hard_sigmoid", "data_format": "channels_last",
                "extrap_start_time": None, "output_mode": "error"}
    layers = [
        {"name": "input_1", "class_name": "InputLayer",
         "config": {"batch_input_shape": [None, nt, hp, wp, cfg.stack_sizes[0]], "dtype": "float32", "sparse": False,
                    "name": "input_1"}, "inbound_nodes": []},
        {"name": "pred_net_1", "class_name": "PredNet", "config": pred_cfg, "inbound_nodes": [[["input_1", 0, 0, {}]]]},
        {"name": "time_distributed_1", "class_name": "TimeDistributed",
         "config": {"name": "time_distributed_1", "trainable": False,
                    "layer": {"class_name": "Dense", "config": _dense_config("dense_1")}},
         "inbound_nodes": [[["pred_net_1", 0, 0, {}]]]},
        {"name": "flatten_1", "class_name": "Flatten",
         "config": {"name": "flatten_1", "trainable": True, "data_format": "channels_last"},
         "inbound_nodes": [[["time_distributed_1", 0, 0, {}]]]},
        {"name": "dense_2", "class_name": "Dense", "config": _dense_config("dense_2"),
         "inbound_nodes": [[["flatten_1", 0, 0, {}]]]},
    ]
    return json.dumps({"class_name": "Model",
                       "config": {"name": "model_1", "layers": layers, "input_layers": [["input_1", 0, 0]],
                                  "output_layers": [["dense_2", 0, 0]]},
                       "keras_version": "2.2.4", "backend": "tensorflow"})


def save_model(model_dir, cfg, weights, hp, wp):
    """The reference's model directory: prednet_model.json + prednet_weights.hdf5 (train.py:109,114-117)."""
    from . import h5lite
    os.makedirs(model_dir, exist_ok=True)
    text = make_model_json(cfg, hp, wp)
    with open(os.path.join(model_dir, JSON_NAME), "w") as f:
        f.write(text)
    h5lite.save_prednet_checkpoint(os.path.join(model_dir, H5_NAME), cfg, weights, nt=2, model_config=text)
    stale = os.path.join(model_dir, NPZ_NAME)
    if os.path.exists(stale):  # a converted copy of OLDER weights: load_model prefers the hdf5, so it is only misleading
        os.replace(stale, stale + ".superseded")   # ... but it is the user's file: renamed, not deleted


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
    if os.path.exists(h5):  # the reference's file wins over a converted copy (which may be stale)
        try:
            weights = _load_h5(h5, cfg)
        except ImportError:  # no h5py in this image: the built-in reader covers Keras' files
            from . import h5lite
            weights = h5lite.load_prednet_weights(h5, [n for n, _ in cfg.weight_shapes()])
    elif os.path.exists(npz):
        z = np.load(npz)
        weights = [z["w%03d" % i] for i in range(len(z.files))]
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
