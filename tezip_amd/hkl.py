"""The reference's training-set container: hickle 4.0.1 files (docs/index.rst:265 pins
`hickle==4.0.1`; written by train_data_create.py:82-83 `hkl.dump(X, ...)`, `hkl.dump(source_list,
...)`, read by data_utils.py:14-15 / train.py:29 `hkl.load`).  hickle is not installable here, so
this is a restatement of its on-disk layout from knowledge of that version -- **parity unpinned** --
on top of the built-in HDF5 subset (tezip_amd/h5lite.py):

  /                      attrs HICKLE_VERSION = "4.0.1", HICKLE_PYTHON_VERSION
  /data                  one dataset per dumped object
        ndarray  ->  the array itself (contiguous), attrs base_type = b"ndarray",
                     type = pickle.dumps(numpy.ndarray), np_dtype = b"uint8"
        list of str -> fixed-length byte strings (utf-8), attrs base_type = b"list",
                     type = pickle.dumps(list), str_type = b"<class 'str'>"

`load` never unpickles anything: it reads the `data` dataset and the plain-bytes `base_type` /
`str_type` attributes are not even needed for the two shapes the reference stores (a uint8 image
stack and a list of source labels)."""
import pickle
import sys

import numpy as np

from . import h5lite


def dump(obj, path):
    root = h5lite.Group({"HICKLE_VERSION": b"4.0.1",
                         "HICKLE_PYTHON_VERSION": ("%d.%d.%d" % sys.version_info[:3]).encode()})
    if isinstance(obj, np.ndarray):
        root.dataset("data", obj, {"base_type": b"ndarray", "type": pickle.dumps(np.ndarray, protocol=3),
                                   "np_dtype": str(obj.dtype).encode()})
    elif isinstance(obj, (list, tuple)) and all(isinstance(x, str) for x in obj):
        raw = [x.encode("utf8") for x in obj]
        arr = np.array(raw, dtype="S%d" % max([len(x) for x in raw] + [1]))
        root.dataset("data", arr, {"base_type": b"list", "type": pickle.dumps(list, protocol=3),
                                   "str_type": b"<class 'str'>"})
    else:
        raise NotImplementedError("hkl.dump: only numpy arrays and lists of str (what train_data_create.py stores)")
    h5lite.write_file(path, root)


def load(path):
    """-> numpy array (image stack) or list of str (source labels)."""
    data = h5lite.H5File(path).walk()
    hit = [v for k, v in data.items() if k.rsplit("/", 1)[-1] in ("data", "data_0")]
    if len(hit) != 1:
        raise ValueError("%s: expected one hickle `data` dataset, found %d" % (path, len(hit)))
    arr = hit[0]
    if arr.dtype.kind == "S":
        return [x.decode("utf8") for x in arr.reshape(-1)]
    return arr
