#!/usr/bin/env python3
"""Wall-time breakdown of the command-line path on the cfg3 workload (80 PNGs of 512x512):
where a user's time goes outside the GPU hot path."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import compress, decompress, synth, weights, zstd, _lib
from tezip_amd.prednet import PredNetConfig
from PIL import Image

tmp = tempfile.mkdtemp(prefix="tzcli_")
frames = synth.turbulence(80, 512, 512, seed=3)
d = os.path.join(tmp, "data"); os.mkdir(d)
for t in range(80):
    Image.fromarray(frames[t]).save(os.path.join(d, "f%03d.png" % t))
cfg = PredNetConfig(); m = os.path.join(tmp, "model"); weights.save_model(m, cfg, cfg.init_weights(123), 512, 512)
T = {}
t0 = time.time(); stack, files, rgb = compress.load_images(d); T["load 80 PNGs"] = time.time() - t0
t0 = time.time(); c, w, s = weights.load_model(m); T["read model dir"] = time.time() - t0
t0 = time.time(); ctx = compress.make_context(c, w, 512, 512, 4); T["context + model prepare"] = time.time() - t0
t0 = time.time(); key, _ = ctx.rollout(stack, 0, 20); payload, table, _ = ctx.encode("abs", [2.0], True); T["GPU rollout+encode (host buffers)"] = time.time() - t0
kf = np.zeros_like(stack); kf[key] = stack[key]
t0 = time.time(); a = zstd.compress_array(kf, 9); T["zstd-9 key_frame (%d MB -> %.1f MB)" % (kf.nbytes >> 20, len(a) / 2**20)] = time.time() - t0
st = compress.build_stream(payload, table, (1, 80, 512, 512, 3), 0)
t0 = time.time(); b = zstd.compress_array(st, 9); T["zstd-9 entropy, 1 thread (%d MB -> %.1f MB)" % (st.nbytes >> 20, len(b) / 2**20)] = time.time() - t0
t0 = time.time(); b = zstd.compress_array(st, 9, zstd.default_threads()); T["zstd-9 entropy, %d worker threads (-> %.1f MB)" % (zstd.default_threads() if zstd.multithreaded() else 1, len(b) / 2**20)] = time.time() - t0
ctx.close()
t0 = time.time(); compress.run(m, d, os.path.join(tmp, "c"), 0, 20, None, "abs", [2.0], True, False, True); T["compress.run total"] = time.time() - t0
t0 = time.time(); decompress.run(m, os.path.join(tmp, "c"), os.path.join(tmp, "u"), True, False); T["decompress.run total"] = time.time() - t0
for k, v in T.items():
    print("%-48s %7.3f s" % (k, v))
