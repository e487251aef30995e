#!/opt/conda/bin/python3.9
"""Reads a prednet_weights.hdf5 with REAL h5py the way Keras 2.2.4 does
(`Model.load_weights` -> `load_weights_from_hdf5_group`: descend into /model_weights when the
root has no `layer_names`, walk `layer_names`, per layer `weight_names`, np.asarray(g[name])) and
prints one line per weighted layer: name, number of arrays, sha1 of their float32 bytes.  Used by
tests/test_host.py to check files WRITTEN by tezip_amd/h5lite.py against libhdf5 itself.
Run: /opt/conda/bin/python3.9 tests/golden/check_h5_with_h5py.py <file.hdf5>"""
import hashlib
import sys

import h5py
import numpy as np

with h5py.File(sys.argv[1], "r") as f:
    print("keras_version", f.attrs["keras_version"].decode("utf8"), "backend", f.attrs["backend"].decode("utf8"))
    if "layer_names" not in f.attrs and "model_weights" in f:
        f = f["model_weights"]
    layer_names = [n.decode("utf8") for n in f.attrs["layer_names"]]
    print("layers", ",".join(layer_names))
    for name in layer_names:
        g = f[name]
        weight_names = [n.decode("utf8") for n in g.attrs["weight_names"]]
        if not weight_names:
            continue
        h = hashlib.sha1()
        for wn in weight_names:
            a = np.asarray(g[wn])
            assert a.dtype == np.float32
            h.update(np.ascontiguousarray(a).tobytes())
        print(name, len(weight_names), h.hexdigest(), weight_names[0], weight_names[-1])
