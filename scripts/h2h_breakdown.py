#!/usr/bin/env python3
"""Where the host-to-host step of the cfg3 job loses its 2.5 ms against the device-resident one: wall time of
tz_rollout / tz_encode with device-resident, pinned and deferred-pinned buffers, each alone and back to back."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tezip_amd import _lib, synth  # noqa: E402
from tezip_amd.prednet import PredNetConfig  # noqa: E402

cfg = PredNetConfig()
ctx = _lib.Context(0)
ctx.load_model(cfg, cfg.init_weights(seed=123))
ctx.prepare(512, 512, 4)
f = synth.turbulence(80, 512, 512, seed=3)
fd = torch.from_numpy(f).cuda()
fp = _lib.pinned_copy(f)
n = f.size
pd = torch.empty(n, dtype=torch.int16, device="cuda")
pp = [_lib.pinned_empty(n, np.int16) for _ in range(2)]


def timeit(fn, reps=5):
    fn()
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("rollout device frames      %.2f ms" % timeit(lambda: ctx.rollout(fd, 0, 20)))
print("rollout pinned frames      %.2f ms" % timeit(lambda: ctx.rollout(fp, 0, 20)))
ctx.rollout(fd, 0, 20)
print("encode device payload      %.2f ms" % timeit(lambda: ctx.encode("rel", [1e-3], True, payload=pd)))
print("encode pinned payload      %.2f ms" % timeit(lambda: ctx.encode("rel", [1e-3], True, payload=pp[0])))
ctx.set_payload_deferred(True)
print("encode pinned, deferred    %.2f ms (5 back to back: each orders itself behind the transfer before it)" %
      timeit(lambda: ctx.encode("rel", [1e-3], True, payload=pp[0])))
t0 = time.perf_counter()
ctx.encode("rel", [1e-3], True, payload=pp[0])
t1 = time.perf_counter()
ctx.payload_wait()
t2 = time.perf_counter()
print("  one deferred encode: call %.2f ms, payload_wait %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
ctx.set_payload_deferred(False)


def step_dev():
    ctx.rollout(fd, 0, 20)
    ctx.encode("rel", [1e-3], True, payload=pd)


def step_pin():
    ctx.rollout(fp, 0, 20)
    ctx.encode("rel", [1e-3], True, payload=pp[0])


k = [0]


def step_def():
    ctx.rollout(fp, 0, 20)
    ctx.payload_wait()
    ctx.encode("rel", [1e-3], True, payload=pp[k[0] & 1])
    k[0] += 1


def step_dev_in_def_out():
    ctx.rollout(fd, 0, 20)
    ctx.payload_wait()
    ctx.encode("rel", [1e-3], True, payload=pp[k[0] & 1])
    k[0] += 1


print("step device -> device      %.2f ms" % timeit(step_dev))
print("step pinned -> pinned      %.2f ms" % timeit(step_pin))
print("step pinned -> device      %.2f ms" % timeit(lambda: (ctx.rollout(fp, 0, 20), ctx.encode("rel", [1e-3], True, payload=pd))))
print("step device -> pinned      %.2f ms" % timeit(lambda: (ctx.rollout(fd, 0, 20), ctx.encode("rel", [1e-3], True, payload=pp[0]))))
ctx.set_payload_deferred(True)
print("step pinned -> deferred    %.2f ms" % timeit(step_def))
print("step device -> deferred    %.2f ms" % timeit(step_dev_in_def_out))
ctx.set_payload_deferred(False)
ctx.close()
