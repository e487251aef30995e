"""The identity shortcut of the error-bound quantiser (round 6, tz_quant_is_identity in tz_codec.hip): for a job whose worst-case
tolerance is <= 0.499 no two different deltas can merge (compress.py:55-67), every run is a run of equal deltas d, and its
value trunc((fl(d + E) + fl(d - E)) / 2) is d -- the fused encode then takes the one-pass LOSSLESS kernel.  At full size
(512x512, where the C oracle is too slow for a whole job) the shortcut must give the bytes of the general quantiser
(TEZIP_QMAP=0: speculative walks + stitch + fill), which the small-size parity tests (tests/test_gpu_parity.py CASES, the
fuzz) pin to the oracle and tests/golden/ref_runs4.npz to the reference's own runs at such tolerances."""
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
ctx.load_model(cfg, cfg.init_weights(seed=123))
h = hashlib.sha256()
frames = synth.turbulence(24, 512, 512, seed=3)
ctx.prepare(512, 512, max_batch=3)
for (p, window, mode, bound, entropy) in [(0, 8, "rel", [1e-3], True), (2, 6, "abs", [0.3], True), (0, 8, "absrel", [0.45, 0.5], False),
                                          (1, 5, "rel", [0.0019], True), (0, 8, "abs", [0.499], True), (0, 8, "pwrel", [0.0015], True)]:
    key, _ = ctx.rollout(frames, p, window)
    ctx.prof_enable(True)
    ctx.prof_reset()
    payload, table, _ = ctx.encode(mode, bound, entropy)
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    print("launches", mode, bound, "sdelta", prof["spatial_delta_hist"][1], "delta", prof["delta"][1])
    keys = np.where(key[:, None, None, None], frames, 0).astype(np.uint8)
    ctx.rollout_decode(keys, p)
    dec = ctx.decode(payload, table)
    err = int(np.abs(dec.astype(np.int16) - frames.astype(np.int16)).max())
    assert err == 0, err        # error_bound is the identity at these tolerances: the job is lossless
    for a in (key, payload, dec) + ((table,) if table is not None else ()):
        h.update(np.ascontiguousarray(a).tobytes())
print("digest", h.hexdigest())
'''


def _run(qmap):
    env = dict(os.environ)
    env.pop("TEZIP_QMAP", None)
    if qmap is not None:
        env["TEZIP_QMAP"] = qmap
    out = subprocess.run([sys.executable, "-c", JOB % ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    return [ln for ln in lines if ln.startswith("digest")][-1], [ln for ln in lines if ln.startswith("launches")]


def test_identity_shortcut_gives_the_general_quantisers_bytes_at_full_size():
    fast, fast_launches = _run(None)
    general, general_launches = _run("0")
    assert fast == general
    # the shortcut really ran (one fused delta pass, no k_q_fill_sym), and TEZIP_QMAP=0 really took the general quantiser
    assert all(ln.split()[-3] == "0" and ln.split()[-1] == "1" for ln in fast_launches), fast_launches
    assert all(ln.split()[-3] == "1" and ln.split()[-1] == "0" for ln in general_launches), general_launches
