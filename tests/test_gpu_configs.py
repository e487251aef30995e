"""Every BASELINE.json configuration as a WORKLOAD on the MI355X, with the FULL reference model
(PredNet (3,48,96,192), train.py:51-55; glorot weights seed 123 -- the reference ships none):

  cfg1  64x64 'L' moving blobs, nt=40, -p 0 -w 20, lossless     end to end against the C oracle
  cfg2  128x160 RGB KITTI-like, nt=40, -p 0 -w 10, lossless     end to end against the C oracle
  cfg3  512x512 RGB turbulence, nt=80, -w 20, `rel 1e-3`/`abs 2` one whole window (predictions, delta, quantiser)
                                                                 and the complete payload + table + stream vs the C oracle
  cfg5  512x512, lossless: DWP with a threshold that gives 5..40-frame windows, and the SWP sweep
        -w in {5,...,40} (tezip_amd/sweep.py)
(cfg4, 320 x 1024x1024, -w 40, `abs 2`: tests/test_gpu_fullsize.py::test_cfg4_full_length_on_one_gpu.)

"End to end against the oracle" = key mask, pre-zstd entropy stream (payload + table + trailer),
key-frame stream and the decoded frames are compared byte for byte with oracle/oracle.py driven
by the C PredNet (oracle/tz_oracle.c); at 512x512 the oracle's predictor needs > 1 s per frame, so
cfg3 checks one whole window of predictions and the integer back half of the whole job, cfg5 uses
properties."""
import numpy as np
import pytest

from oracle import coracle
from oracle import oracle as O
from tezip_amd import compress, decompress, synth, sweep, zstd
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu

CFG = PredNetConfig()
WTS = CFG.init_weights(seed=123)


@pytest.fixture(scope="module")
def ctx():
    from tezip_amd import _lib
    c = _lib.Context(0)
    c.load_model(CFG, WTS)
    yield c
    c.close()


class _Oracle:
    def __init__(self, hp, wp):
        self.net = coracle.CPredNet(WTS, CFG.stack_sizes, CFG.R_stack_sizes, hp, wp)

    def c0(self, a, b):
        return self.net.c0()

    def next(self, f):
        return self.net.next(np.asarray(f, np.float32))


def _end_to_end_vs_oracle(ctx, frames, p, window, mode, bound):
    nt, h, w = frames.shape[:3]
    hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
    pred = _Oracle(hp, wp)
    ref = O.compress_oracle(frames, p, window, None, mode, bound, pred, True)
    ctx.prepare(hp, wp, max_batch=(nt - p + window - 1) // window)
    key, _ = ctx.rollout(frames, p, window)
    np.testing.assert_array_equal(key, ref["key"])
    payload, table, _ = ctx.encode(mode, bound, True)
    stream = compress.build_stream(payload, table, (1, nt, h, w, 3), p)
    np.testing.assert_array_equal(stream, ref["stream"])            # what entropy.dat holds before zstd
    key_stack = np.zeros_like(frames)
    key_stack[key] = frames[key]
    np.testing.assert_array_equal(key_stack.reshape(-1), ref["key_frame"])  # what key_frame.dat holds before zstd
    pl, tb, shape, warm = decompress.parse_stream(stream.tobytes())
    decompress.check_stream(shape, warm, pl.size, key_stack.size)
    ctx.rollout_decode(key_stack, warm)
    dec = ctx.decode(np.ascontiguousarray(pl), np.ascontiguousarray(tb))
    np.testing.assert_array_equal(dec, O.decode_stream(ref["stream"], ref["key_frame"], pred))
    return dec, key


def test_cfg1_64x64_w20_lossless_end_to_end_vs_oracle(ctx):
    frames = synth.moving_blobs(40, 64, 64, seed=1)   # 'L' source expanded to RGB (compress.py:114)
    dec, key = _end_to_end_vs_oracle(ctx, frames, 0, 20, "abs", [0.0])
    assert key.nonzero()[0].tolist() == [0, 20] and np.array_equal(dec, frames)


def test_cfg2_128x160_w10_lossless_end_to_end_vs_oracle(ctx):
    frames = synth.translating_scene(40, 128, 160, seed=2)
    dec, key = _end_to_end_vs_oracle(ctx, frames, 0, 10, "abs", [0.0])
    assert key.nonzero()[0].tolist() == [0, 10, 20, 30] and np.array_equal(dec, frames)


def test_cfg3_512_nt80_w20_workload(ctx):
    """BASELINE configs[2] at its full size against the ORACLE (not against itself):
      * predictions of one whole 20-frame window (depth 1..19) bit for bit vs the C PredNet;
      * that window's complete delta stack (delta + quantiser, 19 x 3 chains of 262,144 elements) for
        both `rel 1e-3` and `abs 2` vs tzo_delta_frame / tzo_error_bound;
      * the payload, the table and the pre-zstd entropy.dat stream of ALL 80 frames (62.9 M elements)
        byte for byte vs the C oracle's spatial delta -> 1600 offset -> histogram -> table -> remap ->
        trailer (compress.py:329-395) run on the int16 delta stack;
      * the decoder regenerating the encoder's predictions, and the round trip within the bound."""
    frames = synth.turbulence(80, 512, 512, seed=3)
    ctx.prepare(512, 512, max_batch=4)
    key, _ = ctx.rollout(frames, 0, 20)
    assert key.nonzero()[0].tolist() == [0, 20, 40, 60]
    enc_pred = ctx.get_predictions()
    key_stack = np.zeros_like(frames)
    key_stack[key] = frames[key]
    results = {}
    for mode, bound in (("rel", [1e-3]), ("abs", [2.0])):
        payload, table, delta = ctx.encode(mode, bound, True, want_delta=True)
        assert 0 < len(table) <= 1021 and int(payload.min()) >= 0 and int(payload.max()) < len(table)
        assert (delta[key] == 0).all()
        # integer back half of the whole job vs the C oracle
        ref_payload, ref_table = coracle.encode_tail(delta, True)
        np.testing.assert_array_equal(table, ref_table, err_msg=mode)
        np.testing.assert_array_equal(payload, ref_payload, err_msg=mode)
        stream = compress.build_stream(payload, table, (1, 80, 512, 512, 3), 0)
        assert stream.tobytes() == O.build_stream(ref_payload, ref_table, 80, 512, 512, 0).tobytes()
        results[mode] = (np.array(payload), table, delta)
        # the same job WITHOUT the delta tap takes the fused kernels (quantiser on pred / orig, run values
        # straight into spatial delta + histogram: no delta stack in memory): same bytes
        fused_payload, fused_table, _ = ctx.encode(mode, bound, True)
        np.testing.assert_array_equal(fused_table, table, err_msg=mode)
        np.testing.assert_array_equal(fused_payload, results[mode][0], err_msg=mode)
        del fused_payload
    # one complete window: predictor recursion to depth 19, then delta + quantiser of every frame
    net = coracle.CPredNet(WTS, CFG.stack_sizes, CFG.R_stack_sizes, 512, 512)
    np.testing.assert_array_equal(enc_pred[0], net.c0())
    cur = coracle.u8_to_f32_frame(frames[0], 512, 512)
    for d in range(1, 20):
        cur = net.next(cur)
        np.testing.assert_array_equal(enc_pred[d], cur, err_msg="depth %d" % d)
        raw = coracle.delta_frame(cur, frames[d])
        for mode, bound in (("rel", [1e-3]), ("abs", [2.0])):
            np.testing.assert_array_equal(results[mode][2][d], coracle.error_bound_frame(frames[d], raw, mode, bound),
                                          err_msg="%s, frame %d" % (mode, d))
    # decoder: regenerates the encoder's predictions bit for bit; lossless / within the bound
    ctx.rollout_decode(key_stack, 0)
    assert np.array_equal(ctx.get_predictions()[~key], enc_pred[~key])
    for mode, tol in (("rel", 0), ("abs", 3)):
        dec = ctx.decode(results[mode][0], results[mode][1])
        err = int(np.abs(dec.astype(np.int16) - frames.astype(np.int16)).max())
        assert err <= tol, (mode, err)


def _windows(key, nt):
    k = key.nonzero()[0].tolist()
    return [b - a for a, b in zip(k, k[1:] + [nt])]


def test_cfg5_512_dwp_lossless_windows_5_to_40(ctx):
    frames = synth.turbulence(80, 512, 512, seed=3)
    fe_pad = 512 * 512 * 3
    ctx.prepare(512, 512, max_batch=1)
    _, probe = ctx.rollout(frames[:41], 0, None, 1e9, want_mse=True)   # one 41-frame window: the MSE curve
    assert (np.diff(probe[1:]) > 0).all()                               # recursion error grows with depth
    key = None
    for depth in (12, 16, 20, 8, 24, 30):                               # a threshold reached at about that depth
        thr = float(probe[depth])
        key, mse = ctx.rollout(frames, 0, None, thr, want_mse=True)
        lens = _windows(key, 80)
        if all(5 <= n <= 40 for n in lens[:-1]) and lens[-1] <= 40 and len(lens) > 2:
            break
    else:
        pytest.fail("no threshold gave 5..40-frame windows: %r" % (lens,))
    # the decisions replayed from the device's own predictions (compress.py:245-263)
    pred = ctx.get_predictions()
    sse = ctx.window_sse(frames, pred)
    run, k0, expect = 0.0, 1, [0]
    for idx in range(1, 80):
        if key[idx] and not (idx == 79 and mse[idx] <= thr):
            # a rejected frame: its prediction was dropped (slot idx now holds C0, compress.py:258),
            # so only the logged value can be checked against the threshold
            assert mse[idx] > thr
            expect.append(idx)
            k0, run = idx + 1, 0.0
            continue
        run += sse[idx]
        stop = run / ((idx - k0 + 1) * fe_pad)
        assert stop == pytest.approx(mse[idx], rel=1e-12) and stop <= thr
    assert key.nonzero()[0].tolist() == expect   # a rejected frame becomes the next key frame
    # the integer back half of the DWP job vs the C oracle (compress.py:329-373) on the delta tap, and the same job
    # through the fused kernels (no delta stack in memory)
    payload, table, delta = ctx.encode("abs", [0.0], True, want_delta=True)
    assert (delta[key] == 0).all()
    ref_payload, ref_table = coracle.encode_tail(delta, True)
    np.testing.assert_array_equal(table, ref_table)
    np.testing.assert_array_equal(payload, ref_payload)
    payload = np.array(payload)
    fused_payload, fused_table, _ = ctx.encode("abs", [0.0], True)
    np.testing.assert_array_equal(fused_table, table)
    np.testing.assert_array_equal(fused_payload, payload)
    # the B = 1 rollout takes other kernels than the batched SWP job (k_convlat on the upper levels): its first window's
    # predictions to depth 3, and the deltas formed from them, vs the C oracle
    net = coracle.CPredNet(WTS, CFG.stack_sizes, CFG.R_stack_sizes, 512, 512)
    cur = coracle.u8_to_f32_frame(frames[0], 512, 512)
    for d in range(1, min(4, expect[1] if len(expect) > 1 else 4)):
        cur = net.next(cur)
        np.testing.assert_array_equal(pred[d], cur, err_msg="depth %d" % d)
        np.testing.assert_array_equal(delta[d], coracle.delta_frame(cur, frames[d]))
    key_stack = np.zeros_like(frames)
    key_stack[key] = frames[key]
    kd = ctx.rollout_decode(key_stack, 0)
    assert (kd == key).all() and np.array_equal(ctx.get_predictions()[~key], pred[~key])
    assert np.array_equal(ctx.decode(payload, table), frames)


def test_cfg5_512_swp_sweep_5_to_40_lossless(ctx):
    frames = synth.turbulence(80, 512, 512, seed=3)
    rows, (best_w, key_bytes, ent_bytes) = sweep.sweep(ctx, frames, 0, sweep.DEFAULT_WINDOWS, "abs", [0.0])
    assert [r["window"] for r in rows] == list(sweep.DEFAULT_WINDOWS)
    for r in rows:
        n_windows = (80 + r["window"] - 1) // r["window"]
        assert r["key_frames"] in (n_windows, n_windows + 1)  # + the last frame when it starts a window
    best = min(rows, key=lambda r: (r["total_bytes"], r["window"]))
    assert best["window"] == best_w and len(key_bytes) == best["key_bytes"] and len(ent_bytes) == best["entropy_bytes"]
    # key_frame.dat grows with the number of key frames
    for a in rows:
        for b in rows:
            assert a["key_frames"] <= b["key_frames"] or a["key_bytes"] > b["key_bytes"]
    # the kept output is a complete reference-format pair: decode it
    key_stack = np.frombuffer(zstd.decompress(key_bytes), np.uint8).reshape(frames.shape)
    pl, tb, shape, warm = decompress.parse_stream(zstd.decompress(ent_bytes))
    decompress.check_stream(shape, warm, pl.size, key_stack.size)
    ctx.prepare(512, 512, max_batch=min(64, (80 + best_w - 1) // best_w))
    ctx.rollout_decode(np.ascontiguousarray(key_stack), warm)
    assert np.array_equal(ctx.decode(np.ascontiguousarray(pl), np.ascontiguousarray(tb)), frames)
