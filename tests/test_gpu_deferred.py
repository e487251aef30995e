"""Deferred payload hand-over (tz_set_payload_deferred / tz_payload_wait): the device -> host transfer of a payload
runs under the next sequence's rollout.  Same bytes as the blocking form, at most one transfer in flight, every
synchronising call settles it.  (No reference counterpart: compress.py:375-400 is one blocking pass per job.)"""
import numpy as np
import pytest

from tezip_amd import _lib, synth
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu
CFG = PredNetConfig(stack_sizes=(3, 16, 32))


def test_deferred_payloads_equal_blocking_ones():
    ctx = _lib.Context(0)
    try:
        ctx.load_model(CFG, CFG.init_weights(seed=3, bias_scale=0.2))
        nt, h, w = 24, 256, 320                      # 5.9 M elements: above the chunked hand-over's threshold
        ctx.prepare(h, w, max_batch=4)
        seqs = [synth.turbulence(nt, h, w, seed=40 + i) for i in range(3)]
        want = []
        for f in seqs:
            ctx.rollout(f, 0, 6)
            p, t, _ = ctx.encode("abs", [1.0], True)
            want.append((np.array(p), t))
        pinned_in = [_lib.pinned_copy(f) for f in seqs]
        bufs = [_lib.pinned_empty(nt * h * w * 3, np.int16) for _ in range(2)]
        for b in bufs:
            b[...] = -7
        ctx.set_payload_deferred(True)
        tables = []
        for i, f in enumerate(pinned_in):
            ctx.rollout(f, 0, 6)
            ctx.payload_wait()
            if i > 0:                                 # the older buffer is complete now
                np.testing.assert_array_equal(bufs[(i - 1) & 1], want[i - 1][0], err_msg="sequence %d" % (i - 1))
            _, t, _ = ctx.encode("abs", [1.0], True, payload=bufs[i & 1])
            tables.append(t)                          # table and table_len are final when the call returns
            np.testing.assert_array_equal(t, want[i][1])
        ctx.payload_wait()
        np.testing.assert_array_equal(bufs[(len(seqs) - 1) & 1], want[-1][0])
        # synchronize settles a transfer in flight too
        ctx.rollout(pinned_in[0], 0, 6)
        ctx.encode("abs", [1.0], True, payload=bufs[0])
        ctx.synchronize()
        np.testing.assert_array_equal(bufs[0], want[0][0])
        # a pageable buffer, the lossless-without-entropy form and the blocking mode are untouched by the switch
        ctx.rollout(pinned_in[1], 0, 6)
        p, t, _ = ctx.encode("abs", [1.0], True)
        np.testing.assert_array_equal(p, want[1][0])
        ctx.set_payload_deferred(False)
        ctx.rollout(pinned_in[2], 0, 6)
        ctx.encode("abs", [1.0], True, payload=bufs[1])
        np.testing.assert_array_equal(bufs[1], want[2][0])
    finally:
        ctx.close()


@pytest.mark.parametrize("h,w", [(64, 96), (61, 90), (21, 30)])
def test_pinned_pageable_and_device_stacks_roll_out_alike(h, w):
    """The key frames of a PINNED stack are fetched by the compute stream itself (k_fetch_frames; frames of a size that
    is a multiple of 16 bytes), every other host stack goes through the copy engine, a device stack is copied on the
    device: same key mask, same predictions, same payload."""
    import torch
    ctx = _lib.Context(0)
    try:
        ctx.load_model(CFG, CFG.init_weights(seed=5, bias_scale=0.2))
        ctx.prepare(_lib.pad8(h), _lib.pad8(w), max_batch=3)
        nt = 11
        frames = synth.translating_scene(nt, h, w, seed=77)
        res = []
        for stack in (frames, _lib.pinned_copy(frames), torch.from_numpy(frames).cuda()):
            key, _ = ctx.rollout(stack, 1, 4)
            pred = ctx.get_predictions()
            payload, table, _ = ctx.encode("abs", [1.0], True)
            res.append((key, pred, np.array(payload), table))
        for other in res[1:]:
            np.testing.assert_array_equal(other[0], res[0][0])
            np.testing.assert_array_equal(other[1], res[0][1])
            np.testing.assert_array_equal(other[2], res[0][2])
            np.testing.assert_array_equal(other[3], res[0][3])
    finally:
        ctx.close()
