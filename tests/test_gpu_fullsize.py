"""BASELINE.json frame sizes on the GPU, checked through size-independent properties (the oracle
would take minutes here): compress -> decompress round trips, the error bound as the reference
defines it (|err| <= E, +1 for the truncated median, compress.py:61), encoder and decoder
regenerating identical predictions, batch invariance, and a spot check of a few windows of the
quantiser and one prediction against the C oracle."""
import numpy as np
import pytest

from tezip_amd import synth
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu

CFG = PredNetConfig()


@pytest.fixture(scope="module")
def ctx():
    from tezip_amd import _lib
    c = _lib.Context(0)
    c.load_model(CFG, CFG.init_weights(seed=123))
    yield c
    c.close()


def _roundtrip(ctx, frames, p, window, thr, mode, bound, entropy=True):
    key, _ = ctx.rollout(frames, p, window, thr)
    enc_pred = ctx.get_predictions()
    payload, table, delta = ctx.encode(mode, bound, entropy, want_delta=True)
    key_stack = np.zeros_like(frames)
    key_stack[key] = frames[key]
    kd = ctx.rollout_decode(key_stack, p)
    dec_pred = ctx.get_predictions()
    dec = ctx.decode(payload, table)
    return key, kd, enc_pred, dec_pred, payload, table, delta, dec


@pytest.mark.parametrize("mode,bound,tol", [("abs", [0.0], 0), ("rel", [1e-3], 0), ("abs", [2.0], 3), ("pwrel", [0.02], 7)])
def test_cfg3_512_roundtrip(ctx, mode, bound, tol):
    frames = synth.turbulence(24, 512, 512, seed=3)
    ctx.prepare(512, 512, max_batch=4)
    key, kd, ep, dp, payload, table, delta, dec = _roundtrip(ctx, frames, 0, 6, None, mode, bound)
    assert key.tolist() == [i % 6 == 0 for i in range(24)] and (kd == key).all()
    # the decoder regenerates the encoder's predictions bit for bit (every non-key slot)
    assert np.array_equal(ep[~key], dp[~key])
    assert payload.shape == (24 * 512 * 512 * 3,) and 0 < len(table) <= 1021
    assert int(payload.min()) >= 0 and int(payload.max()) < len(table)
    err = np.abs(dec.astype(np.int16) - frames.astype(np.int16))
    assert int(err.max()) <= tol
    if tol == 0:
        assert np.array_equal(dec, frames)
    assert (delta[key] == 0).all()


def test_cfg3_quantiser_and_prediction_spot_check_vs_oracle(ctx):
    from oracle import coracle
    frames = synth.turbulence(8, 512, 512, seed=4)
    ctx.prepare(512, 512, max_batch=2)
    key, _ = ctx.rollout(frames, 0, 4)
    pred = ctx.get_predictions()
    raw = ctx.delta_encode(pred, frames, key.astype(np.uint8))
    _, _, delta = ctx.encode("abs", [2.0], True, want_delta=True)
    for f in (1, 6):  # whole-frame chains of 262,144 elements
        np.testing.assert_array_equal(delta[f], coracle.error_bound_frame(frames[f], raw[f], "abs", [2.0]))
    net = coracle.CPredNet(CFG.init_weights(seed=123), CFG.stack_sizes, CFG.R_stack_sizes, 512, 512)
    np.testing.assert_array_equal(pred[1], net.next(coracle.u8_to_f32_frame(frames[0], 512, 512)))
    np.testing.assert_array_equal(pred[0], net.c0())


def test_cfg4_1024_lossy_and_dwp(ctx):
    frames = synth.detector(10, 1024, 1024, seed=4)
    ctx.prepare(1024, 1024, max_batch=2)
    key, kd, ep, dp, payload, table, delta, dec = _roundtrip(ctx, frames, 0, 5, None, "abs", [2.0])
    assert key.tolist() == [True, False, False, False, False] * 2
    assert np.array_equal(ep[~key], dp[~key])
    assert int(np.abs(dec.astype(np.int16) - frames.astype(np.int16)).max()) <= 3
    # DWP on the same data: windows come from the MSE threshold; still a lossless round trip
    ctx.prepare(1024, 1024, max_batch=2)
    k1, mse = ctx.rollout(frames, 1, None, 1e9, want_mse=True)
    assert k1.tolist() == [True, True] + [False] * 8 and (np.diff(mse[2:]) != 0).any()
    thr = float(np.sort(mse[2:])[len(mse[2:]) // 2])
    key, kd, ep, dp, payload, table, delta, dec = _roundtrip(ctx, frames, 1, None, thr, "abs", [0.0])
    assert 2 < key.sum() < 10 and np.array_equal(dec, frames)


def test_batch_invariance_of_the_rollout(ctx):
    frames = synth.turbulence(16, 512, 512, seed=5)
    ctx.prepare(512, 512, max_batch=4)
    ctx.rollout(frames, 0, 4)
    a = ctx.get_predictions()
    ctx.prepare(512, 512, max_batch=1)  # same windows, one at a time
    ctx.rollout(frames, 0, 4)
    assert np.array_equal(a, ctx.get_predictions())


def test_cfg4_full_length_on_one_gpu(ctx):
    """BASELINE configs[3] at its full size (320 frames of 1024x1024, eight 40-frame windows):
    1.0e9 elements through every kernel with device-resident buffers.  The integer back half is
    compared with the C ORACLE at this size -- spatial delta, 1600 offset, histogram, table, remap
    and trailer of the whole 1,006,632,960-element stack (64-bit offsets, multi-block histogram
    merge), payload and table byte for byte (compress.py:329-395) -- plus the quantiser of two whole
    frames and two predicted frames of the first window (the convolutions at 1024x1024); then the round trip within
    the bound."""
    import torch
    from oracle import coracle
    from oracle import oracle as O
    from tezip_amd import compress
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(4)
    nt, h, w = 320, 1024, 1024
    base = torch.poisson(torch.full((nt, h, w), 3.0, device=dev), generator=g)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    for k in range(12):  # a few drifting Bragg-like peaks
        y0, x0 = 100 + 70 * k, 90 + 75 * k
        t = torch.arange(nt, device=dev).view(-1, 1, 1)
        base += 180 * torch.exp(-(((yy - y0 - 0.2 * t) ** 2 + (xx - x0 + 0.1 * t) ** 2) / 18.0))
    frames = base.clamp(0, 255).to(torch.uint8)[..., None].expand(nt, h, w, 3).contiguous()
    del base
    ctx.prepare(1024, 1024, max_batch=8)
    payload = torch.empty(nt * h * w * 3, dtype=torch.int16, device=dev)
    delta = torch.empty(nt * h * w * 3, dtype=torch.int16, device=dev)
    # this context runs on its OWN HIP stream: order it against torch's stream explicitly
    torch.cuda.synchronize()
    key, _ = ctx.rollout(frames, 0, 40)
    assert key.nonzero()[0].tolist() == list(range(0, 320, 40))
    _, table, _ = ctx.encode("abs", [2.0], True, payload=payload, delta_out=delta)
    ctx.synchronize()  # device outputs are asynchronous on the context's stream
    assert 0 < len(table) <= 1021 and int(payload.max()) < len(table) and int(payload.min()) >= 0
    # --- the whole job's integer back half vs the C oracle
    delta_h = delta.cpu().numpy()
    ref_payload, ref_table = coracle.encode_tail(delta_h, True)
    np.testing.assert_array_equal(table, ref_table)
    payload_h = payload.cpu().numpy()
    assert np.array_equal(payload_h, ref_payload)
    stream = compress.build_stream(payload_h, table, (1, nt, h, w, 3), 0)
    assert stream[-7 - len(table):].tobytes() == O.build_stream(ref_payload[:0], ref_table, nt, h, w, 0).tobytes()
    del ref_payload, payload_h, stream
    # without the delta tap the encode takes the fused lossy kernels: same payload, same table
    payload2 = torch.empty_like(payload)
    _, table2, _ = ctx.encode("abs", [2.0], True, payload=payload2)
    ctx.synchronize()
    assert np.array_equal(table2, table) and bool(torch.equal(payload2, payload))
    del payload2
    # --- quantiser of two whole frames (3 chains of 1,048,576 elements each) vs the C oracle
    pred = ctx.get_predictions()
    # --- the predictor itself at this size: the first window's predictions to depth 2 (frame 1 from the key frame,
    # frame 2 from that prediction: prednet.py:235-308 on 1024x1024 levels), bit for bit vs the C oracle
    net = coracle.CPredNet(CFG.init_weights(seed=123), CFG.stack_sizes, CFG.R_stack_sizes, 1024, 1024)
    np.testing.assert_array_equal(pred[0], net.c0())
    cur = coracle.u8_to_f32_frame(frames[0].cpu().numpy(), 1024, 1024)
    for d in (1, 2):
        cur = net.next(cur)
        np.testing.assert_array_equal(pred[d], cur, err_msg="1024x1024 prediction, depth %d" % d)
    del net, cur
    for f in (1, 279):
        fr = frames[f].cpu().numpy()
        raw = coracle.delta_frame(pred[f], fr)
        got = delta_h[f * h * w * 3:(f + 1) * h * w * 3].reshape(h, w, 3)
        np.testing.assert_array_equal(got, coracle.error_bound_frame(fr, raw, "abs", [2.0]), err_msg="frame %d" % f)
    del pred, delta_h, delta
    keys = torch.zeros_like(frames)
    kidx = torch.from_numpy(key).to(dev)
    keys[kidx] = frames[kidx]
    out = torch.empty_like(frames)
    torch.cuda.synchronize()
    ctx.rollout_decode(keys, 0)
    ctx.decode(payload, table, out=out)
    ctx.synchronize()
    err = (out.to(torch.int16) - frames.to(torch.int16)).abs()
    assert int(err.max()) <= 3 and int(err[kidx].max()) == 0


@pytest.mark.parametrize("h,w,batch,seed", [(512, 512, 4, 5), (512, 512, 3, 6), (376, 1248, 2, 7), (1024, 1024, 1, 8)])
def test_lds_dma_kernels_agree_with_the_general_kernel_at_full_size(ctx, h, w, batch, seed):
    """k_conv16 / k_conv16b / k_conv_small and the general k_conv3x3 walk the same fmaf chains: the
    whole recursive rollout must agree bit for bit (k_conv3x3 is pinned to the C oracle at the
    small sizes of test_gpu_parity.py; this carries the pin to the BASELINE.json frame sizes,
    including the KITTI size whose levels are not multiples of the 16-pixel tiles)."""
    nt = 2 * batch + 1
    frames = synth.turbulence(nt, h, w, seed=seed)
    ctx.prepare((h + 7) // 8 * 8, (w + 7) // 8 * 8, max_batch=batch)
    preds = []
    for impl in (1, 0):
        ctx.set_conv_impl(impl)
        ctx.rollout(frames, 0, 3)
        preds.append(ctx.get_predictions())
    ctx.set_conv_impl(1)
    assert np.array_equal(preds[0], preds[1])
    assert preds[0].std() > 0  # not trivially equal


def test_host_buffers_pinned_pageable_and_device_give_the_same_bytes(ctx):
    """The copy engine (key frames first, rest of the stack under the predictor, payload leaving in
    chunks behind the remap kernel; pageable memory pipelined through the pinned staging buffers)
    must not change a byte: pinned host, pageable host and device-resident runs agree."""
    import torch
    from tezip_amd import _lib
    frames = synth.turbulence(24, 512, 512, seed=11)
    n = frames.size
    ctx.prepare(512, 512, max_batch=4)
    res = {}
    for kind in ("pageable", "pinned", "device"):
        if kind == "pageable":
            f, pay, out = frames, np.empty(n, np.int16), np.empty(frames.shape, np.uint8)
        elif kind == "pinned":
            f, pay, out = _lib.pinned_copy(frames), _lib.pinned_empty(n, np.int16), _lib.pinned_empty(frames.shape, np.uint8)
        else:
            dev = torch.device("cuda", 0)
            f = torch.from_numpy(frames).to(dev)
            pay = torch.empty(n, dtype=torch.int16, device=dev)
            out = torch.empty(frames.shape, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
        key, mse = ctx.rollout(f, 0, 6, want_mse=(kind == "pinned"))
        _, table, _ = ctx.encode("abs", [2.0], True, payload=pay)
        ctx.synchronize()
        ks = np.zeros_like(frames)
        ks[key] = frames[key]
        if kind == "device":
            ks = torch.from_numpy(ks).to(dev)
            torch.cuda.synchronize()  # the context runs on its own stream
        ctx.rollout_decode(ks, 0)
        ctx.decode(pay, table, out=out)
        ctx.synchronize()
        res[kind] = (key, table, pay.cpu().numpy() if kind == "device" else np.array(pay),
                     out.cpu().numpy() if kind == "device" else np.array(out))
    for kind in ("pinned", "device"):
        for a, b in zip(res["pageable"], res[kind]):
            assert np.array_equal(a, b), kind
    assert int(np.abs(res["pinned"][3].astype(np.int16) - frames.astype(np.int16)).max()) <= 3


@pytest.mark.parametrize("h,w,batch,nt,seed", [(128, 160, 4, 13, 21), (512, 512, 1, 3, 22), (61, 90, 2, 9, 23), (40, 56, 2, 7, 24)])
def test_small_grid_kernel_agrees_with_k_conv16(ctx, h, w, batch, nt, seed):
    """k_convlat (one accumulator tile per wave, weights streamed per wave through a register ring, for
    launches that cannot fill the chip) walks the same fmaf chains as k_conv16: forced on wherever it is
    eligible -- every level >= 1, also at 512x512 where it runs thousands of 32-pixel workgroups in
    several rounds -- the whole recursive rollout must equal the k_conv16 result bit for bit, image
    borders and sizes that are no multiple of its 4x4 / 8x8 tiles included (40x56: 10x14 and 20x28
    levels).  The two small cases also take the split gate launches (E part beside the A convolution,
    accumulators through HBM, upsampled part behind: tz_model_predict_batch_dev)."""
    frames = synth.turbulence(nt, h, w, seed=seed)
    ctx.prepare((h + 7) // 8 * 8, (w + 7) // 8 * 8, max_batch=batch)
    preds = {}
    for lat in ("never", "always", None):
        ctx.set_conv_impl(1, lat=lat)
        ctx.rollout(frames, 0, (nt + batch - 1) // batch)
        preds[lat] = ctx.get_predictions()
    ctx.set_conv_impl(1)
    assert np.array_equal(preds["never"], preds["always"]) and np.array_equal(preds["never"], preds[None])
    assert preds["never"].std() > 0


def test_soak_lds_dma_kernels_forced_small_grid_kernel_included():
    """scripts/soak_conv.py once per GPU-suite run: 114 rollouts over eleven frame sizes and batch
    sizes, each through the LDS-DMA kernels, the general kernel and k_convlat forced wherever it is eligible
    (lat='always': register weight ring, asm LDS-DMA with hand-counted vmcnt waits) -- all bit-identical.  A DMA /
    wait-count race shows up as rare differing tiles; see tests/test_build_guard.py for the static half."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_conv.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatching" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_hot_launches_run_on_the_kernels_the_design_names(ctx):
    """Dispatch guard: at 512x512 with four windows every convolution of levels >= 1 is k_wino (TZ-PA2 is the contract in
    force from 256x256 pixels on; 95 launches per 19-step rollout) -- k_conv16 when TZ-PA1 is asked for --, the level-0
    ones k_conv16b / k_conv_small, and nothing falls back to the general k_conv3x3; at 64x64 (TZ-PA1 by size) the levels
    >= 1 run on k_convlat (split gate launches included).  A silent fallback would keep every parity test green and
    cost a large part of the headline."""
    frames = synth.turbulence(80, 512, 512, seed=9)
    ctx.prepare(512, 512, max_batch=4)
    assert ctx.get_contract() == 2
    ctx.rollout(frames, 0, 20)
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.rollout(frames, 0, 20)
    p = ctx.prof_get()
    assert p["wino_pa2"][1] == 95 and p["conv16_lds_dma"][1] == 0 and p["conv16b_level0"][1] == 38 and p["conv_small_valu"][1] == 19
    assert p["conv3x3_general"][1] == 0 and p["convlat_small_grid"][1] == 0
    ctx.set_contract(1)
    ctx.prof_reset()
    ctx.rollout(frames, 0, 20)
    p = ctx.prof_get()
    ctx.set_contract(0)
    assert p["conv16_lds_dma"][1] == 95 and p["wino_pa2"][1] == 0 and p["conv16b_level0"][1] == 38 and p["conv_small_valu"][1] == 19
    assert p["conv3x3_general"][1] == 0 and p["convlat_small_grid"][1] == 0
    # the level-0 error unit is launched for the step that starts from the key frames only: every later step finds its
    # error maps written by the prediction kernel of the step before (round 3)
    assert p["err0"][1] == 1
    small = synth.moving_blobs(40, 64, 64)
    ctx.prepare(64, 64, max_batch=2)
    assert ctx.get_contract() == 1
    ctx.rollout(small, 0, 20)
    ctx.prof_reset()
    ctx.rollout(small, 0, 20)
    p = ctx.prof_get()
    ctx.prof_enable(False)
    # per step: pair (A1 + gates-1 E part), pair (A2 + gates-2 E part), gates 3, gates-2 / gates-1 upsampled parts
    assert p["convlat_small_grid"][1] == 19 * 5 and p["conv16_lds_dma"][1] == 0 and p["conv3x3_general"][1] == 0


def test_more_than_2_31_elements_on_one_gpu():
    """720 frames of 1024x1024x3 = 2,264,924,160 elements -- past what a signed or unsigned 31-bit index reaches -- through
    rollout (18 windows in one batch), the fused lossless and lossy encodes, the tapped encode, and the decoder's one-launch
    tail: lossless round trip bit-exact, `abs 2` within the bound, fused payload = tapped payload = the C oracle's back half
    (spatial delta, offset, histogram, table, remap) of the device's own delta stack, table for table, byte for byte."""
    import torch
    from oracle import coracle
    from tezip_amd import _lib
    dev = torch.device("cuda", 0)
    cfg = PredNetConfig()
    c = _lib.Context(0)
    try:
        c.load_model(cfg, cfg.init_weights(seed=123))
        nt, h, w = 720, 1024, 1024
        n = nt * h * w * 3
        assert n > 2 ** 31
        g = torch.Generator(device=dev).manual_seed(4)
        frames = torch.empty((nt, h, w, 3), dtype=torch.uint8, device=dev)
        yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        for a in range(0, nt, 40):   # detector-like: Poisson background + drifting peaks (cfg4's kind of data)
            base = torch.poisson(torch.full((40, h, w), 3.0, device=dev), generator=g)
            t = torch.arange(a, a + 40, device=dev).view(-1, 1, 1)
            for k in range(6):
                y0, x0 = 100 + 140 * k, 90 + 150 * k
                base += 180 * torch.exp(-(((yy - y0 - 0.2 * t) ** 2 + (xx - x0 + 0.1 * t) ** 2) / 18.0))
            frames[a:a + 40] = base.clamp(0, 255).to(torch.uint8)[..., None]
        del base
        c.prepare(h, w, max_batch=18)
        payload = torch.empty(n, dtype=torch.int16, device=dev)
        out = torch.empty_like(frames)
        torch.cuda.synchronize()   # the context launches on a stream of its own
        key, _ = c.rollout(frames, 0, 40)
        assert key.nonzero()[0].tolist() == list(range(0, nt, 40))
        keys = torch.zeros_like(frames)
        kidx = torch.from_numpy(key).to(dev)
        keys[kidx] = frames[kidx]
        torch.cuda.synchronize()

        def max_err():
            return max(int((out[a:a + 80].to(torch.int16) - frames[a:a + 80].to(torch.int16)).abs().max()) for a in range(0, nt, 80))

        for bound in (0.0, 2.0):
            c.rollout(frames, 0, 40)
            _, table, _ = c.encode("abs", [bound], True, payload=payload)
            c.synchronize()
            if bound:
                delta = torch.empty(n, dtype=torch.int16, device=dev)
                tapped = torch.empty(n, dtype=torch.int16, device=dev)
                _, table2, _ = c.encode("abs", [bound], True, payload=tapped, delta_out=delta)
                c.synchronize()
                assert torch.equal(payload, tapped) and np.array_equal(table, table2)
                delta_h = delta.cpu().numpy()
                del delta, tapped
                ref_payload, ref_table = coracle.encode_tail(delta_h, True)
                np.testing.assert_array_equal(table, ref_table)
                assert np.array_equal(payload.cpu().numpy(), ref_payload)
                del delta_h, ref_payload
            c.rollout_decode(keys, 0)
            c.decode(payload, table, out=out)
            c.synchronize()
            assert max_err() <= bound
    finally:
        c.close()
