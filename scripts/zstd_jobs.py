#!/usr/bin/env python3
"""zstd level 9 of a real cfg3 payload (80 frames of 512x512, abs 2): wall time and size for the job
sizes / overlaps of libzstd's multi-threaded compressor (same standard frame either way)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth, zstd  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

f = synth.turbulence(80, 512, 512)
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
ctx.rollout(f, 0, 20)
payload, table, _ = ctx.encode("abs", [2.0], True)
ctx.close()
a = np.ascontiguousarray(payload)
L = zstd._lib()
threads = zstd.default_threads() if hasattr(zstd, "default_threads") else 16
print("payload %d bytes, %d threads, libzstd %d" % (a.nbytes, threads, L.ZSTD_versionNumber()))
cap = L.ZSTD_compressBound(a.nbytes)
dst = C.create_string_buffer(cap)
t0 = time.perf_counter()
m = L.ZSTD_compress(dst, cap, a.ctypes.data, a.nbytes, 9)
print("single thread            : %.3f s, %d bytes" % (time.perf_counter() - t0, m))
base = m
for job_mb, ov in ((0, 0), (16, 0), (8, 0), (4, 0), (2, 0), (1, 0), (4, 9), (2, 9), (2, 8), (1, 9)):
    cctx = L.ZSTD_createCCtx()
    L.ZSTD_CCtx_setParameter(cctx, 100, 9)
    L.ZSTD_CCtx_setParameter(cctx, 400, threads)
    if job_mb:
        L.ZSTD_CCtx_setParameter(cctx, 401, job_mb << 20)
    if ov:
        L.ZSTD_CCtx_setParameter(cctx, 402, ov)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        m = L.ZSTD_compress2(cctx, dst, cap, a.ctypes.data, a.nbytes)
        best = min(best, time.perf_counter() - t0)
    L.ZSTD_freeCCtx(cctx)
    print("jobSize %2d MB overlapLog %d: %.3f s, %d bytes (%+.2f %% vs single thread)" % (job_mb, ov, best, m, 100.0 * (m - base) / base))
