#!/usr/bin/env python3
"""cProfile of compress.run / decompress.run on the cfg3 data (80 PNGs of 512x512): where the
wall time of the CLI goes outside the GPU path."""
import cProfile
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import compress, decompress, synth, weights  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

tmp = tempfile.mkdtemp()
d = os.path.join(tmp, "in")
os.makedirs(d)
frames = synth.turbulence(80, 512, 512)
from PIL import Image  # noqa: E402
for t in range(80):
    Image.fromarray(frames[t]).save(os.path.join(d, "f%03d.png" % t))
cfg = PredNetConfig()
m = os.path.join(tmp, "model")
weights.save_model(m, cfg, cfg.init_weights(123), 512, 512)
compress.run(m, d, os.path.join(tmp, "c0"), 0, 20, None, "abs", [2.0], True, False, True)  # warm (library load, HIP init)
for name, fn in (("compress.run", lambda: compress.run(m, d, os.path.join(tmp, "c"), 0, 20, None, "abs", [2.0], True, False, True)),
                 ("decompress.run", lambda: decompress.run(m, os.path.join(tmp, "c"), os.path.join(tmp, "u"), True, False))):
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    print("=====", name)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
