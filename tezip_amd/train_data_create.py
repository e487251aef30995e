"""Training-set builder: same job as the reference's train_data_create.py (a folder of
time-series sub-folders of images -> zero-padded uint8 stacks + a per-image source label,
random ~10 % of the folders held out for validation unless given), written for this build:
the stacks are saved under the reference's names and in its container, hickle 4.0.1 files
(train_data_create.py:82-83; written by tezip_amd/hkl.py on the built-in HDF5 writer, since hickle
is not installable here -- layout restated from knowledge of that version, parity unpinned):
    X_train.hkl  (N, Hp, Wp, 3) uint8      sources_train.hkl  list of N str
    X_val.hkl    ...                        sources_val.hkl
Usage: python -m tezip_amd.train_data_create DATA_DIR OUTPUT_DIR [-v VAL_FOLDER ...]"""
import argparse
import os
import random

import numpy as np

from . import hkl
from .data_utils import padding_shape


def _images(folder):
    return sorted(f for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f)))


def process_data(data_dir, output_dir, val_folders=None, seed=None):
    from PIL import Image, UnidentifiedImageError
    folders = sorted(d for d in os.listdir(data_dir) if os.path.isdir(os.path.join(data_dir, d)))
    if len(folders) < 2:
        print("ERROR: Two or more time-series folders are required in the specified folder.")
        print("Please prepare at least two for tarin and val.")
        exit()
    if val_folders:
        val = [os.path.basename(v.rstrip("/")) for v in val_folders]
    else:
        rng = random.Random(seed)
        val = rng.sample(folders, max(1, len(folders) // 10))
    split = {"train": [f for f in folders if f not in val], "val": [f for f in folders if f in val]}
    try:
        sizes = [Image.open(os.path.join(data_dir, f, _images(os.path.join(data_dir, f))[0])).size for f in folders]
        hp, wp = padding_shape(max(s[1] for s in sizes), max(s[0] for s in sizes))
        print("After Padding ：height:", hp, " width:", wp)
        os.makedirs(output_dir, exist_ok=True)
        for name, members in split.items():
            paths, sources = [], []
            for f in members:
                files = _images(os.path.join(data_dir, f))
                paths += [os.path.join(data_dir, f, x) for x in files]
                sources += [name + "-" + f] * len(files)
            print("Creating " + name + " data: " + str(len(paths)) + " images")
            X = np.zeros((len(paths), hp, wp, 3), np.uint8)
            for i, p in enumerate(paths):
                im = np.array(Image.open(p).convert("RGB"))
                X[i, : im.shape[0], : im.shape[1]] = im
            hkl.dump(X, os.path.join(output_dir, "X_%s.hkl" % name))
            hkl.dump(sources, os.path.join(output_dir, "sources_%s.hkl" % name))
    except (PermissionError, IndexError, UnidentifiedImageError):
        print("ERROR: Contains non-image files or inappropriate folders.")
        exit()


if __name__ == "__main__":
    ap = argparse.ArgumentParser(prog="TRAIN_DATA_CREATE")
    ap.add_argument("data_dir")
    ap.add_argument("output_dir")
    ap.add_argument("-v", "--val_dir_path", nargs="*")
    a = ap.parse_args()
    process_data(a.data_dir, a.output_dir, a.val_dir_path)
