#!/usr/bin/env python3
"""The output stage of compress.run in isolation (cfg3, abs 2): payload pieces out of HBM, the zstd
stream of entropy.dat, the zstd stream of key_frame.dat -- each alone and together."""
import io
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, compress, synth, zstd  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

f = synth.turbulence(80, 512, 512)
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
key, _ = ctx.rollout(f, 0, 20)
_, table, _ = ctx.encode("abs", [2.0], True, payload="resident")
nt, H, W = 80, 512, 512
n = nt * H * W * 3
CH = compress.PAYLOAD_CHUNK
bufs = [np.empty(min(CH, n), np.int16) for _ in range(2)]


class Null(io.RawIOBase):
    def write(self, b):
        return len(b)


def pieces_only():
    for k, off in enumerate(range(0, n, CH)):
        cnt = min(CH, n - off)
        ctx.payload_get(off, cnt, out=bufs[k % 2][:cnt])


def entropy(threads):
    sc = zstd.StreamCompressor(Null(), n * 2, 9, threads)
    for k, off in enumerate(range(0, n, CH)):
        cnt = min(CH, n - off)
        sc.write(ctx.payload_get(off, cnt, out=bufs[k % 2][:cnt]))
    return sc.close()


def keyfile(threads):
    zero = np.zeros((H, W, 3), np.uint8)
    sc = zstd.StreamCompressor(Null(), n, 9, threads)
    for i in range(nt):
        sc.write(f[i] if key[i] else zero)
    return sc.close()


def timed(name, fn, *a):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = fn(*a)
        best = min(best, time.perf_counter() - t0)
    print("%-46s %.3f s  %s" % (name, best, r if r is not None else ""))


T = zstd.default_threads()
print("chunk %d elements, %d zstd threads" % (CH, T))
timed("payload pieces HBM -> host only", pieces_only)
timed("entropy stream, %d threads" % T, entropy, T)
timed("entropy stream, 1 thread", entropy, 1)
timed("key_frame stream, %d threads" % max(1, T // 4), keyfile, max(1, T // 4))
timed("key_frame stream, 1 thread", keyfile, 1)
with ThreadPoolExecutor(2) as pool:
    def both():
        k = pool.submit(keyfile, max(1, T // 4))
        e = entropy(T)
        return e, k.result()
    timed("both (as compress.run does)", both)
ctx.close()
