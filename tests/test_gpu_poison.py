"""Results must not depend on what device memory held before: the library's TEZIP_POISON diagnostic fills every device
buffer it hands out (fresh or recycled) with a byte first.  Each poison value runs in its own process (the switch is read
once per process) over a lossless and three lossy jobs, SWP and DWP, at a size that uses the LDS-DMA convolution kernels,
the tile quantiser and the fused encode / decode tails; the digests of everything the jobs return must agree."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JOB = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=11, bias_scale=0.1))
h = hashlib.sha256()
for (nt, H, W, p, window, thr, mode, bound) in [(14, 128, 160, 1, 5, None, "abs", [0.0]), (14, 128, 160, 0, 4, None, "abs", [2.0]),
                                               (12, 64, 80, 0, None, 0.15, "rel", [0.01]), (9, 61, 90, 2, 3, None, "pwrel", [0.1])]:
    frames = synth.turbulence(nt, H, W, seed=4)
    ctx.prepare(_lib.pad8(H), _lib.pad8(W), max_batch=4)
    for rep in range(2):   # the second pass runs on recycled pool blocks
        key, _ = ctx.rollout(frames, p, window, thr)
        payload, table, _ = ctx.encode(mode, bound, True)
        keys = np.where(key[:, None, None, None], frames, 0).astype(np.uint8)
        ctx.rollout_decode(keys, p)
        dec = ctx.decode(payload, table)
        for a in (key, payload, table, dec):
            h.update(np.ascontiguousarray(a).tobytes())
        if bound[0] == 0:
            assert (dec == frames).all()
print("digest", h.hexdigest())
'''


def _run(poison):
    env = dict(os.environ)
    env.pop("TEZIP_POISON", None)
    if poison is not None:
        env["TEZIP_POISON"] = str(poison)
    out = subprocess.run([sys.executable, "-c", JOB % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [ln for ln in out.stdout.splitlines() if ln.startswith("digest")][-1]


def test_results_do_not_depend_on_stale_device_memory():
    ref = _run(None)
    for poison in (255, 0, 165):
        assert _run(poison) == ref, "TEZIP_POISON=%d changes the result" % poison


DWP_JOB = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from oracle import coracle
from oracle import oracle as O
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(stack_sizes=(3, 16, 32))
wts = cfg.init_weights(seed=3, bias_scale=0.2)
ctx = _lib.Context(0)
ctx.load_model(cfg, wts)
for (nt, H, W, p) in [(10, 64, 96, 0), (9, 45, 61, 2)]:   # (the C oracle's predictor is what this test's time goes to)
    hp, wp = _lib.pad8(H), _lib.pad8(W)
    frames = synth.turbulence(nt, H, W, seed=6)
    net = coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)
    class P:
        def c0(self, a, b):
            return net.c0()
        def next(self, f):
            return net.next(np.asarray(f, np.float32))
    probe = O.rollout(frames, p, None, 1e9, P())
    thr = float(np.median(probe["mse"]))
    ref = O.rollout(frames, p, None, thr, P())
    ctx.prepare(hp, wp, max_batch=4)
    for rep in range(3):   # part[] is re-poisoned every time the pool hands it out: every DWP step meets NaN bit patterns
        key, mse = ctx.rollout(frames, p, None, thr, want_mse=True)
        assert np.isfinite(mse).all(), mse
        np.testing.assert_array_equal(key, ref["key"])
        np.testing.assert_allclose(mse[p + 1:], ref["mse"], rtol=1e-12)
print("dwp ok")
# ... and at 512x512 (192 partials per step, the cfg5 shape; no oracle at this size): the log must be finite and the same
# with and without the poison
import hashlib
full = PredNetConfig()
ctx.load_model(full, full.init_weights(seed=123))
ctx.prepare(512, 512, max_batch=1)
img = np.random.default_rng(3).integers(0, 256, (64, 64, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)   # (cheap frames: a drifting block image)
f = np.stack([np.roll(img, 3 * t, axis=1) for t in range(14)])
_, probe = ctx.rollout(f, 0, None, 1e9, want_mse=True)
thr = float(np.sort(probe[1:])[5])
h = hashlib.sha256()
for rep in range(3):
    key, mse = ctx.rollout(f, 0, None, thr, want_mse=True)
    assert np.isfinite(mse).all(), mse
    h.update(key.tobytes()); h.update(mse.tobytes())
print("digest512", h.hexdigest())
'''


def test_dwp_decision_never_reads_a_partial_it_did_not_wait_for():
    """k_sse_decide's part[] filled with 0xFF bytes (NaN bit patterns) before every rollout: a last-ticket workgroup that
    read a slot before its writer's exchange had been performed would put a NaN into mse[] (and take no boundary there).
    The window MSE log and the key mask must equal the oracle's (compress.py:245-264) at two small sizes, and at 512x512
    (192 partials per step) be finite and equal to the unpoisoned run's.  The ordering itself is read off the object code
    in tests/test_build_guard.py; this is its dynamic half."""
    digests = {}
    for poison in ("255", None):
        env = dict(os.environ)
        env.pop("TEZIP_POISON", None)
        env["OMP_NUM_THREADS"] = "4"   # (the C oracle at these frame sizes: a GPU box's many cores only spin on its tiny loops)
        if poison:
            env["TEZIP_POISON"] = poison
        out = subprocess.run([sys.executable, "-c", DWP_JOB % ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "dwp ok" in out.stdout, out.stderr[-2000:]
        digests[poison] = [ln for ln in out.stdout.splitlines() if ln.startswith("digest512")][-1]
    assert digests["255"] == digests[None]


ROCTX_JOB = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig
cfg = PredNetConfig(stack_sizes=(3, 16, 32))
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=5, bias_scale=0.1))
frames = synth.turbulence(9, 64, 80, seed=2)
ctx.prepare(64, 80, max_batch=2)
key, _ = ctx.rollout(frames, 0, 4)
payload, table, _ = ctx.encode("abs", [2.0], True)
ctx.rollout_decode(np.where(key[:, None, None, None], frames, 0).astype(np.uint8), 0)
dec = ctx.decode(payload, table)
h = hashlib.sha256()
for a in (key, payload, table, dec):
    h.update(np.ascontiguousarray(a).tobytes())
print("digest", h.hexdigest())
'''


def test_roctx_ranges_change_nothing_and_the_library_opens():
    """TEZIP_ROCTX=1 (roctx ranges around the stages): the ROCTx library of this image opens without a
    warning and the job's results are the ones without it.  (That the ranges show up in a trace is checked where a trace
    is taken: scripts/gpu_r06_final.sh -> profiles/r06/roctx_ranges.txt.)"""
    outs = {}
    for flag in (None, "1"):
        env = dict(os.environ)
        env.pop("TEZIP_ROCTX", None)
        if flag:
            env["TEZIP_ROCTX"] = flag
        r = subprocess.run([sys.executable, "-c", ROCTX_JOB % ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "no ROCTx library" not in r.stderr, r.stderr[-500:]
        outs[flag] = [ln for ln in r.stdout.splitlines() if ln.startswith("digest")][-1]
    assert outs[None] == outs["1"]
