#!/opt/conda/bin/python3.9
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference);
its outputs (*.npz in this directory) are committed and are what travels to the GPU box.

What it does:
  * pre-seeds sys.modules with empty stand-ins for the heavy third-party imports of the
    reference modules (keras / numba / hickle / zstd are absent from this image), puts
    /root/reference/src on sys.path and imports `compress`, `decompress`, `data_utils`;
  * calls the reference's own numpy helpers (error_bound, finding_difference x2,
    replacing_based_on_frequency x2, data_padding, padding_size) on seeded inputs and
    records inputs + outputs;
  * runs the reference's compress.run / decompress.run END TO END with the Keras model
    replaced by tests/golden/fake_predictor.py (the predictor arithmetic lives in
    keras==2.2.4/tensorflow==1.15, which are not under /root/reference: parity of the
    predictor itself is unpinned) and `zstd` replaced by the identity, so
    the recorded `entropy.dat` / `key_frame.dat` are the PRE-zstd byte streams.

  * (round 4) runs the reference's train_data_create.process_data and data_utils.SequenceGenerator on a small PNG
    tree (hickle.dump captured, keras' Iterator replaced by a sequential stand-in) -> ref_train.npz.

  * (round 5) instantiates the reference's own `PredNet` class (prednet.py) over numpy stand-ins for the Keras surface it
    touches (tests/golden/keras_standin.py) and runs its build / get_initial_state / step -> ref_prednet.npz.

Every file regenerates bit for bit except ref_train.npz, whose folder order inside a split follows the reference's `set()`
iteration (string hashing: PYTHONHASHSEED); the tests compare it per folder.

Run:  /opt/conda/bin/python3.9 tests/golden/make_golden.py      (--train-only: just ref_train.npz; --prednet-only: just ref_prednet.npz; --runs4-only: just ref_runs4.npz)
(python3.9 + numpy 1.26 because the reference calls ndarray.tostring(), removed in numpy 2.)
Nothing from the reference's source text is written to the fixtures: only arrays.
"""
import io
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fake_predictor  # noqa: E402

REF = "/root/reference/src"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ---- fake Keras surface used by compress.run / decompress.run -------------------------
class _FakeLayer:
    def __init__(self, hp, wp):
        self.batch_input_shape = (None, 2, hp, wp, 3)

    def get_config(self):
        return {"output_mode": "error", "data_format": "channels_last"}

    def get_weights(self):
        return []


class _FakeTrainModel:
    hp = wp = None

    def __init__(self):
        self.layers = [_FakeLayer(self.hp, self.wp), _FakeLayer(self.hp, self.wp)]

    def load_weights(self, path):
        return None


class _Shape(tuple):
    pass


class _FakeTensor:
    def __init__(self, shape):
        self.shape = shape


class _FakeTestModel:
    calls = []

    def __init__(self, inputs=None, outputs=None):
        self.input = _FakeTensor((None, None, _FakeTrainModel.hp, _FakeTrainModel.wp, 3))

    def predict(self, X, batch_size=None):
        _FakeTestModel.calls.append(tuple(X.shape))
        return fake_predictor.predict(X)


def _install_stubs():
    K = _stub("keras.backend", image_data_format=lambda: "channels_last")
    keras = _stub("keras", backend=K)
    _stub("keras.models", Model=_FakeTestModel, model_from_json=lambda s, custom_objects=None: _FakeTrainModel())
    _stub("keras.layers", Input=lambda shape=None: _FakeTensor(shape), Dense=None, Flatten=None,
          Recurrent=object, Conv2D=None, UpSampling2D=None, MaxPooling2D=None)
    _stub("keras.engine", InputSpec=None)
    _stub("keras.legacy")
    _stub("keras.legacy.interfaces", generate_legacy_interface=lambda **k: (lambda f: f),
          recurrent_args_preprocessor=None)
    _stub("keras.preprocessing")
    _stub("keras.preprocessing.image", Iterator=object)
    keras.activations = _stub("keras.activations")
    _stub("numba", cuda=None)
    _stub("hickle")
    # identity "zstd": the files then hold the pre-zstd streams
    _stub("zstd", compress=lambda data, level=3: bytes(data), decompress=lambda data: bytes(data))
    # round 5: numpy stand-ins for the Keras surface prednet.py touches, so that the reference's own PredNet class
    # can be instantiated and stepped (tests/golden/keras_standin.py says what that does and does not pin)
    import keras_standin
    keras_standin.install(sys.modules)
    sys.path.insert(0, REF)


def _helpers(compress, decompress, data_utils, out):
    rng = np.random.default_rng(20261004)

    # --- error_bound: the real call site passes int64 (1,H,W) slabs (compress.py:310-319)
    cases = []
    specs = [
        ("abs", [2.0]), ("abs", [0.4]), ("abs", [-3.0]), ("abs", [7.5]), ("abs", [0.0]),
        ("rel", [0.01]), ("rel", [0.1]), ("rel", [0.0]),
        ("absrel", [3.0, 0.01]), ("absrel", [1.0, 0.5]), ("absrel", [2.0, 0.0]), ("absrel", [0.0, 0.1]),
        ("pwrel", [0.05]), ("pwrel", [0.5]), ("pwrel", [1.0]),  # pwrel<0 raises inside the reference (NaN -> int)
    ]
    for k, (mode, val) in enumerate(specs):
        for shape, kind in [((1, 9, 13), "noise"), ((1, 16, 24), "smooth"), ((1, 1, 1), "noise"), ((1, 7, 64), "flat")]:
            n = int(np.prod(shape))
            orig = rng.integers(0, 256, size=shape).astype(np.int64)
            if kind == "noise":
                diff = rng.integers(-255, 256, size=shape).astype(np.int64)
            elif kind == "smooth":
                diff = np.round(np.cumsum(rng.normal(0, 1.2, size=n))).astype(np.int64).reshape(shape)
                diff = np.clip(diff, -255, 255)
            else:
                diff = np.full(shape, int(rng.integers(-3, 4)), dtype=np.int64)
                diff.reshape(-1)[rng.integers(0, n, size=3)] += rng.integers(-9, 10, size=3)
                orig = np.full(shape, 128, dtype=np.int64)
                orig.reshape(-1)[::7] = 3
            res = compress.error_bound(orig.copy(), diff.copy(), mode, val, False, np)
            # the caller assigns the result into an int64 array (compress.py:319): truncation
            res_int = np.empty(shape, dtype=np.int64)
            res_int[...] = res
            cases.append((mode, val, orig, diff, res_int))
    out["eb_n"] = np.array(len(cases))
    for i, (mode, val, orig, diff, res) in enumerate(cases):
        out["eb_%d_mode" % i] = np.array(mode)
        out["eb_%d_val" % i] = np.array(val, dtype=np.float64)
        out["eb_%d_orig" % i] = orig
        out["eb_%d_diff" % i] = diff
        out["eb_%d_res" % i] = res
    # doc KAT (docs/img/img33.png) with the float inputs of the figure
    E = np.array([6, 4, 2, 4, 2, 6, 2, 2, 6], dtype=np.float64)
    D = np.array([0, 0, -5, -5, 10, 5, -5, -5, 0], dtype=np.float64)
    out["eb_doc_float"] = compress.error_bound(E.copy(), D.copy(), "pwrel", [1.0], False, np)

    # --- finding_difference, encoder and decoder (int16, wrap-around included)
    for i, n in enumerate([1, 2, 9, 1000]):
        a = rng.integers(-255, 256, size=(1, n)).astype(np.int16)
        out["fd_enc_in_%d" % i] = a
        out["fd_enc_out_%d" % i] = compress.finding_difference(a.copy())
        out["fd_dec_out_%d" % i] = decompress.finding_difference(out["fd_enc_out_%d" % i].copy())
    w = rng.integers(-32768, 32768, size=(3, 5, 7)).astype(np.int16)
    out["fd_wrap_in"] = w
    with np.errstate(over="ignore"):
        out["fd_wrap_enc"] = compress.finding_difference(w.copy())
        out["fd_wrap_dec"] = decompress.finding_difference(w.copy())

    # --- rank remap, both directions
    sym = (1600 - rng.integers(-40, 41, size=5000)).astype(np.int16)
    table = np.array([1600, 1599, 1601, 1580, 1625, 1610], dtype=np.int16)
    out["rp_in"] = sym
    out["rp_table"] = table
    out["rp_enc"] = compress.replacing_based_on_frequency(sym.copy(), table, np)
    ranks = rng.integers(0, len(table), size=5000).astype(np.int16)
    out["rp_dec_in"] = ranks
    out["rp_dec"] = decompress.replacing_based_on_frequency(ranks.copy(), table, np)

    # --- padding
    x = rng.random((1, 3, 21, 30, 3)).astype(np.float32)
    out["pad_in"] = x
    out["pad_out"] = data_utils.data_padding(x)
    out["pad_sizes_in"] = np.array([1, 7, 8, 9, 64, 375, 1242, 512, 1023])
    out["pad_sizes_out"] = np.array([data_utils.padding_size(int(v)) for v in out["pad_sizes_in"]])


def _chain(rng, kind, n, e_hint):
    """One (frame, channel) chain of `n` integer deltas whose greedy runs under a tolerance of
    about `e_hint` have a known character (the HIP quantiser works in 64-element chunks, walks a
    chain in 8 segments from speculative starts and stitches them: tz_codec.hip)."""
    if kind == "walk_slow":      # random walk that stays inside a 2E band for hundreds of elements
        d = np.round(np.cumsum(rng.normal(0, 0.25 * max(e_hint, 1.0) / 2.0, size=n)))
    elif kind == "walk_fast":    # runs of about ten elements
        d = np.round(np.cumsum(rng.normal(0, 1.2, size=n)))
        d = (d + 255) % 1020
        d = np.where(d > 510, 1020 - d, d) - 255          # reflect into [-255, 255]
    elif kind in ("steps64", "steps64_off"):              # runs of exactly 64, chunk-aligned or not
        off = 0 if kind == "steps64" else 17
        lev = rng.integers(-200, 201, size=n // 64 + 2)
        jump = int(2 * e_hint) + 3
        lev[1:] = np.where(np.abs(np.diff(lev)) <= jump, lev[1:] + 3 * jump, lev[1:])
        for k in range(1, len(lev)):                       # neighbours differ by more than 2E
            if abs(int(lev[k]) - int(lev[k - 1])) <= jump:
                lev[k] = lev[k - 1] + jump + 1 if lev[k - 1] < 0 else lev[k - 1] - jump - 1
        d = np.repeat(lev, 64)[64 - off: 64 - off + n] if off else np.repeat(lev, 64)[:n]
    elif kind == "flat_spikes":  # runs much longer than a segment (1/8 of the chain)
        d = np.full(n, int(rng.integers(-20, 21)), dtype=np.int64)
        for pos in rng.integers(0, n, size=3):
            d[pos] += int(4 * e_hint) + 9
    elif kind == "one_run":      # the whole chain is a single run
        d = np.full(n, int(rng.integers(-5, 6)), dtype=np.int64)
        d[rng.integers(0, n, size=n // 50)] += 1
    elif kind == "noise":        # runs of one or two elements
        d = rng.integers(-255, 256, size=n)
    elif kind == "mixed":        # a long run, a stretch of noise, steps, a slow walk
        q = n // 4
        d = np.concatenate([_chain(rng, "flat_spikes", q, e_hint), _chain(rng, "noise", q, e_hint),
                            _chain(rng, "steps64_off", q, e_hint), _chain(rng, "walk_slow", n - 3 * q, e_hint)])
    else:
        raise ValueError(kind)
    return np.clip(np.asarray(d, dtype=np.int64), -255, 255)


def _orig_slab(rng, kind, n):
    if kind == "full":           # range 255
        o = rng.integers(0, 256, size=n)
        o[0], o[-1] = 0, 255
    elif kind == "narrow":       # range 40: `rel 0.01` gives E = 0.4, only equal neighbours merge
        o = rng.integers(100, 141, size=n)
        o[3], o[5] = 100, 140
    elif kind == "zeros":        # pwrel: tolerance 0 wherever the pixel is black
        o = rng.integers(0, 256, size=n)
        o[rng.random(n) < 0.08] = 0
        o[n // 3: n // 3 + 700] = 0
    elif kind == "bright":       # pwrel with large tolerances: long runs
        o = rng.integers(180, 256, size=n)
    else:
        raise ValueError(kind)
    return o.astype(np.int64)


def _run_lengths(res):
    f = res.reshape(-1)
    cut = np.flatnonzero(np.diff(f) != 0) + 1
    return np.diff(np.concatenate([[0], cut, [f.size]]))


def _long_chains(compress, out):
    """error_bound of the REFERENCE on chains long enough for the parallel quantiser's chunk carry
    (64 elements), segment speculation (1/8 of a chain) and stitch to fire: 96x128 and 100x131
    (not a multiple of 64) slabs in all four modes and one 256x256 frame per mode.  Three slabs of
    one (mode, bound) make the three channels of an HWC frame, the form tz_error_bound takes."""
    rng = np.random.default_rng(20261005)
    specs = [
        # mode, value, H, W, (delta kind, orig kind) x 3 channels
        ("abs", [2.0], 96, 128, [("walk_slow", "full"), ("steps64", "full"), ("flat_spikes", "full")]),
        ("abs", [0.4], 96, 128, [("walk_slow", "full"), ("one_run", "full"), ("steps64_off", "full")]),
        ("abs", [12.0], 96, 128, [("noise", "full"), ("walk_fast", "full"), ("mixed", "full")]),
        ("abs", [3.5], 100, 131, [("mixed", "full"), ("steps64_off", "full"), ("walk_slow", "full")]),
        ("rel", [0.01], 96, 128, [("walk_slow", "full"), ("walk_fast", "narrow"), ("steps64_off", "full")]),
        ("rel", [0.1], 96, 128, [("noise", "full"), ("mixed", "full"), ("walk_fast", "narrow")]),
        ("absrel", [3.0, 0.01], 96, 128, [("mixed", "full"), ("walk_slow", "narrow"), ("steps64", "full")]),
        ("absrel", [1.0, 0.5], 96, 128, [("flat_spikes", "full"), ("walk_fast", "full"), ("one_run", "narrow")]),
        ("pwrel", [0.05], 96, 128, [("walk_slow", "zeros"), ("mixed", "zeros"), ("steps64_off", "bright")]),
        ("pwrel", [0.5], 100, 131, [("noise", "zeros"), ("walk_fast", "bright"), ("flat_spikes", "zeros")]),
        ("abs", [2.0], 256, 256, [("mixed", "full"), ("flat_spikes", "full"), ("walk_slow", "full")]),
        ("rel", [0.02], 256, 256, [("walk_slow", "full"), ("mixed", "full"), ("steps64_off", "narrow")]),
        ("absrel", [4.0, 0.05], 256, 256, [("flat_spikes", "full"), ("walk_fast", "full"), ("mixed", "narrow")]),
        ("pwrel", [0.1], 256, 256, [("mixed", "zeros"), ("walk_slow", "bright"), ("one_run", "zeros")]),
    ]
    out["lc_n"] = np.array(len(specs))
    for i, (mode, val, h, w, chans) in enumerate(specs):
        n = h * w
        e_hint = abs(val[0]) if mode in ("abs", "absrel") else 255 * val[0]
        orig = np.empty((h, w, 3), np.uint8)
        diff = np.empty((h, w, 3), np.int16)
        res = np.empty((h, w, 3), np.int16)
        stats = []
        for c, (dk, ok) in enumerate(chans):
            o = _orig_slab(rng, ok, n).reshape(1, h, w)
            d = _chain(rng, dk, n, e_hint).reshape(1, h, w)
            r = compress.error_bound(o.copy(), d.copy(), mode, val, False, np)
            r_int = np.empty((1, h, w), dtype=np.int64)     # compress.py:319: stored into an int64 array
            r_int[...] = r
            assert np.abs(r_int).max() <= 255
            orig[..., c], diff[..., c], res[..., c] = o[0], d[0], r_int[0]
            rl = _run_lengths(r_int)
            stats.append((int(rl.size), int(rl.max()), int((rl == 64).sum())))
        out["lc_%d_mode" % i] = np.array(mode)
        out["lc_%d_val" % i] = np.array(val, dtype=np.float64)
        out["lc_%d_orig" % i] = orig
        out["lc_%d_diff" % i] = diff
        out["lc_%d_res" % i] = res
        print("long chain %2d %-6s %-12s %3dx%3d  (runs, longest, runs of 64) per channel: %s" % (
            i, mode, val, h, w, stats))


def _make_frames(rng, nt, h, w, gray):
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    frames = []
    for t in range(nt):
        base = 120 + 60 * np.sin((xx + 2 * t) / 5.0) + 40 * np.cos((yy - t) / 4.0)
        img = np.stack([base, base * 0.7 + 30, 255 - base * 0.5], axis=-1)
        img = img + rng.normal(0, 3.0, size=img.shape)
        img = np.clip(np.round(img), 0, 255).astype(np.uint8)
        if t == 0:
            img[0, 0] = (0, 0, 0)
            img[-1, -1] = (255, 255, 255)
        frames.append(img[..., 0] if gray else img)
    return np.stack(frames)


def _make_frames_uneven(rng, nt, h, w, gray):
    """Motion that speeds up, stalls and jumps, so that the window MSE of compress.py:246 reaches a
    threshold after a different number of frames in every window (DWP windows of mixed length)."""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    speed = np.array([0.2, 0.2, 0.3, 3.0, 0.2, 0.1, 0.1, 0.2, 0.2, 4.0, 2.5, 0.3, 0.2, 0.2, 0.2, 0.1, 5.0, 0.3,
                      0.3, 0.2, 0.2, 0.2, 0.2, 3.5, 0.2, 0.2])
    phase = np.concatenate([[0.0], np.cumsum(speed)])
    frames = []
    for t in range(nt):
        ph = phase[t % len(phase)]
        base = 120 + 70 * np.sin((xx + 3 * ph) / 4.0) + 45 * np.cos((yy - 2 * ph) / 3.0)
        img = np.stack([base, base * 0.6 + 40, 250 - base * 0.5], axis=-1) + rng.normal(0, 1.5, size=(h, w, 3))
        img = np.clip(np.round(img), 0, 255).astype(np.uint8)
        if t == 0:
            img[0, 0] = (0, 0, 0)
        frames.append(img[..., 0] if gray else img)
    return np.stack(frames)


RUNS_1 = [
        # name, nt, H, W, gray, p, w, t, mode, bound, entropy
        ("swp_p2_w4_lossless", 14, 21, 30, False, 2, 4, None, "abs", [0.0], True),
        ("swp_p0_w5_abs4", 12, 16, 24, False, 0, 5, None, "abs", [4.0], True),
        ("dwp_p0_rel", 13, 21, 30, False, 0, None, 0.011, "rel", [0.02], False),
        ("swp_p1_w3_pwrel_gray", 11, 19, 17, True, 1, 3, None, "pwrel", [0.05], True),
        ("swp_p0_w6_absrel", 13, 8, 40, False, 0, 6, None, "absrel", [3.0, 0.01], True),
        ("swp_p0_w4_lastkey", 9, 16, 16, False, 0, 4, None, "abs", [1.0], True),
        ("dwp_p2_lossless", 12, 10, 12, False, 2, None, 0.0045, "abs", [0.0], True),
]
# round 3: DWP runs whose windows have MIXED lengths (the ones above came out uniform), one of them
# with warm-up, a lossy bound and the entropy remap all together
RUNS_2 = [
        ("dwp_mixed_p0_lossless", 26, 24, 40, False, 0, None, 0.0085, "abs", [0.0], True),
        ("dwp_mixed_p2_abs3_entropy", 26, 21, 30, False, 2, None, 0.0085, "abs", [3.0], True),
        ("dwp_mixed_p1_pwrel_gray", 24, 19, 27, True, 1, None, 0.012, "pwrel", [0.04], True),
]
# round 3: the edge cases the GPU parity tests exercise against the oracle, as the REFERENCE runs them -- one-frame
# windows (every frame after the warm-up a key frame), a window longer than the sequence, the shortest legal sequence
# (nt = warm_up + 2), DWP with a threshold that every prediction exceeds / none reaches, absrel with an absolute bound of 0
RUNS_3 = [
        ("swp_p3_w1_every_frame_a_key", 7, 8, 8, False, 3, 1, None, "abs", [0.0], True),
        ("swp_p0_w50_one_window", 7, 9, 9, False, 0, 50, None, "abs", [1.0], True),
        ("swp_p2_w2_shortest_sequence", 4, 8, 16, False, 2, 2, None, "abs", [0.0], False),
        ("dwp_p0_thr0_every_prediction_rejected", 8, 16, 8, False, 0, None, 0.0, "abs", [0.0], True),
        ("dwp_p1_thr_never_reached", 8, 16, 8, False, 1, None, 1e9, "rel", [0.05], True),
        ("swp_p0_w2_absrel_abs0", 6, 8, 8, False, 0, 2, None, "absrel", [0.0, 0.5], True),
        ("swp_p1_w3_rel_no_entropy_gray", 9, 10, 14, True, 1, 3, None, "rel", [0.1], False),
]
# round 6: tolerances that cannot merge two different deltas (E <= 0.499 in the reference's 0..255 units -- BASELINE.json's cfg3
# `rel 1e-3` is one), on UNPADDED frame sizes: the HIP build sends these through its one-pass lossless kernel, because error_bound
# is then the identity (tz_quant_is_identity in tz_codec.hip); here is what the REFERENCE's error_bound makes of them, end to end
RUNS_4 = [
        ("swp_p0_w4_rel1e-3", 10, 16, 24, False, 0, 4, None, "rel", [1e-3], True),
        ("swp_p2_w3_abs0.3", 11, 24, 32, False, 2, 3, None, "abs", [0.3], True),
        ("swp_p1_w5_abs0.499_no_entropy", 9, 16, 16, False, 1, 5, None, "abs", [0.499], False),
        ("swp_p0_w6_absrel_small", 13, 8, 40, False, 0, 6, None, "absrel", [0.4, 0.5], True),
        ("swp_p0_w4_rel0.0019_gray", 10, 16, 16, True, 0, 4, None, "rel", [0.0019], True),
        ("swp_p0_w3_abs0.255", 8, 32, 32, False, 0, 3, None, "abs", [0.255], True),
]


def _runs(compress, decompress, out, runs=None, seed=777, make_frames=None):
    from PIL import Image
    rng = np.random.default_rng(seed)
    make_frames = make_frames or _make_frames
    runs = RUNS_1 if runs is None else runs
    out["run_names"] = np.array([r[0] for r in runs])
    for name, nt, h, w, gray, p, win, thr, mode, bound, entropy in runs:
        frames = make_frames(rng, nt, h, w, gray)
        hp, wp = ((h + 7) // 8) * 8, ((w + 7) // 8) * 8
        _FakeTrainModel.hp, _FakeTrainModel.wp = hp, wp
        _FakeTestModel.calls = []
        tmp = tempfile.mkdtemp(prefix="tzgold_")
        try:
            ddir, cdir, udir, mdir = (os.path.join(tmp, d) for d in ("data", "comp", "out", "model"))
            os.mkdir(ddir)
            os.mkdir(mdir)
            open(os.path.join(mdir, "prednet_model.json"), "w").write("{}")
            names = []
            for t in range(nt):
                fn = "frame_%03d.png" % t
                Image.fromarray(frames[t], mode="L" if gray else "RGB").save(os.path.join(ddir, fn))
                names.append(fn)
            stdout = sys.stdout
            sys.stdout = io.StringIO()
            try:
                compress.run(mdir, ddir, cdir, p, win, thr, mode, bound, False, True, entropy)
                mse_log = sys.stdout.getvalue()
                enc_calls = list(_FakeTestModel.calls)
                _FakeTestModel.calls = []
                decompress.run(mdir, cdir, udir, False, False)
            finally:
                sys.stdout = stdout
            dec_calls = list(_FakeTestModel.calls)
            key = np.frombuffer(open(os.path.join(cdir, "key_frame.dat"), "rb").read(), dtype=np.uint8)
            ent = np.frombuffer(open(os.path.join(cdir, "entropy.dat"), "rb").read(), dtype="<i2")
            ftxt = open(os.path.join(cdir, "filename.txt"), encoding="UTF-8").read()
            dec = np.stack([np.array(Image.open(os.path.join(udir, fn))) for fn in names])
            mses = [float(l.split("MSE:")[1]) for l in mse_log.splitlines() if l.startswith("MSE:")]
            pre = "run_%s_" % name
            out[pre + "frames"] = frames
            out[pre + "params"] = np.array([p, -1 if win is None else win, int(gray), int(entropy)], dtype=np.int64)
            out[pre + "thr"] = np.array(-1.0 if thr is None else thr)
            out[pre + "mode"] = np.array(mode)
            out[pre + "bound"] = np.array(bound, dtype=np.float64)
            out[pre + "key_frame"] = key
            out[pre + "entropy"] = ent
            out[pre + "filename_txt"] = np.array(ftxt)
            out[pre + "decoded"] = dec
            out[pre + "mse"] = np.array(mses, dtype=np.float64)
            out[pre + "enc_calls"] = np.array(enc_calls, dtype=np.int64).reshape(-1, 5)
            out[pre + "dec_calls"] = np.array(dec_calls, dtype=np.int64).reshape(-1, 5)
            kf = key.reshape(1, nt, h, w, 3)
            keys = [i for i in range(nt) if kf[0, i].any()]
            print("%-24s nt=%d keys=%s entropy_len=%d max|dec-orig|=%d" % (
                name, nt, keys, ent.size,
                int(np.abs(dec.astype(int) - (frames if not gray else np.repeat(frames[..., None], 3, -1)).astype(int)).max())))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


# ---- round 4: the training-data side -----------------------------------------------------
class _SeqIterator:
    """Stand-in for keras.preprocessing.image.Iterator (Keras 2.2.4, absent here) with the behaviour
    data_utils.SequenceGenerator relies on: n, batch_size, a lock, batch_index and an index generator that walks
    arange(n) (a permutation when shuffle) in batches, wrapping around."""

    def __init__(self, n, batch_size, shuffle, seed):
        import threading
        self.n, self.batch_size, self.shuffle, self.seed = n, batch_size, shuffle, seed
        self.batch_index, self.total_batches_seen, self.lock = 0, 0, threading.Lock()
        self.index_generator = self._flow_index()

    def _flow_index(self):
        self.batch_index = 0
        while True:
            if self.batch_index == 0:
                self.index_array = np.random.permutation(self.n) if self.shuffle else np.arange(self.n)
            cur = (self.batch_index * self.batch_size) % self.n
            self.batch_index = self.batch_index + 1 if self.n > cur + self.batch_size else 0
            self.total_batches_seen += 1
            yield self.index_array[cur: cur + self.batch_size]


TRAIN_TREE = [
    # folder, number of images, H, W, grayscale
    ("seq_a", 6, 13, 19, False),
    ("seq_b", 5, 16, 24, False),
    ("seq_c", 7, 11, 22, True),
    ("seq_d", 4, 16, 17, False),
    ("seq_e", 5, 9, 24, False),
]


def _train_fixture(out):
    """train_data_create.process_data (train_data_create.py:11-95) and data_utils.SequenceGenerator
    (data_utils.py:8-71) of the REFERENCE on a small PNG tree: hickle.dump is captured, hickle.load answers from
    the capture, keras' Iterator is the stand-in above.  Recorded: the images of the tree, what the reference
    stacked per split (explicit validation folders; the random split under random.seed(5)), possible_starts in
    both start modes / with N_seq, and two batches."""
    import random
    from PIL import Image
    store = {}
    sys.modules["hickle"].dump = lambda obj, path, **k: store.__setitem__(path, obj)
    sys.modules["hickle"].load = lambda path: store[path]
    sys.modules["keras.preprocessing.image"].Iterator = _SeqIterator
    for m in ("train_data_create", "data_utils"):
        sys.modules.pop(m, None)
    import train_data_create
    import data_utils
    rng = np.random.default_rng(20261006)
    tmp = tempfile.mkdtemp(prefix="tztrain_")
    stdout = sys.stdout
    try:
        ddir = os.path.join(tmp, "data")
        os.mkdir(ddir)
        out["tr_folders"] = np.array([t[0] for t in TRAIN_TREE])
        for folder, n, h, w, gray in TRAIN_TREE:
            os.mkdir(os.path.join(ddir, folder))
            imgs = rng.integers(0, 256, size=(n, h, w) if gray else (n, h, w, 3)).astype(np.uint8)
            out["tr_img_" + folder] = imgs
            for i in range(n):
                Image.fromarray(imgs[i], mode="L" if gray else "RGB").save(os.path.join(ddir, folder, "im_%02d.png" % i))
        sys.stdout = io.StringIO()
        # (1) explicit validation folders (-v): train = set(folders) ^ set(val), in whatever order set() iterates
        o1 = os.path.join(tmp, "o1")
        os.mkdir(o1)
        train_data_create.args = types.SimpleNamespace(val_dir_path=[os.path.join(ddir, "seq_b"), os.path.join(ddir, "seq_e")])
        train_data_create.process_data(ddir, o1)
        # (2) the random split (train_data_create.py:26-35)
        o2 = os.path.join(tmp, "o2")
        os.mkdir(o2)
        train_data_create.args = types.SimpleNamespace(val_dir_path=None)
        random.seed(5)
        train_data_create.process_data(ddir, o2)
        log = sys.stdout.getvalue()
        sys.stdout = stdout
        for tag, od in (("v", o1), ("r", o2)):
            for split in ("train", "val"):
                out["tr_%s_X_%s" % (tag, split)] = store[os.path.join(od, "X_%s.hkl" % split)]
                out["tr_%s_src_%s" % (tag, split)] = np.array(store[os.path.join(od, "sources_%s.hkl" % split)])
        out["tr_log"] = np.array(log)
        # (3) SequenceGenerator on (1)'s training files
        xf, sf = os.path.join(o1, "X_train.hkl"), os.path.join(o1, "sources_train.hkl")
        for nt in (2, 3, 5):
            g = data_utils.SequenceGenerator(xf, sf, nt, batch_size=2, shuffle=False, data_format="channels_last")
            out["sg_all_nt%d" % nt] = np.asarray(g.possible_starts, dtype=np.int64)
            gu = data_utils.SequenceGenerator(xf, sf, nt, batch_size=2, shuffle=False, sequence_start_mode="unique",
                                              data_format="channels_last")
            out["sg_unique_nt%d" % nt] = np.asarray(gu.possible_starts, dtype=np.int64)
        g = data_utils.SequenceGenerator(xf, sf, 2, batch_size=1, N_seq=2, data_format="channels_last")   # train.py:90
        out["sg_nseq2"] = np.asarray(g.possible_starts, dtype=np.int64)
        assert g.N_sequences == 2
        g = data_utils.SequenceGenerator(xf, sf, 3, batch_size=2, shuffle=False, data_format="channels_last")
        bx, by = g.next()
        out["sg_batch0_x"], out["sg_batch0_y"] = bx, by
        bx, by = g[None]
        out["sg_batch1_x"] = bx
        out["sg_im_shape"] = np.array(g.im_shape)
        print("train fixture: train sources", list(dict.fromkeys(out["tr_v_src_train"].tolist())), "| random val",
              list(dict.fromkeys(out["tr_r_src_val"].tolist())), "| starts nt=3:", out["sg_all_nt3"].tolist())
    finally:
        sys.stdout = stdout
        shutil.rmtree(tmp, ignore_errors=True)


# ---- round 5: the reference's own PredNet class, executed ----------------------------------------------------------
PREDNET_CASES = [
    # name, stack_sizes (= R_stack_sizes, train.py:51-52), Hp, Wp, T, weight seed, bias scale, frame seed
    ("small", (3, 16, 32), 16, 24, 2, 7, 0.3, 31),
    ("small_t3", (3, 16, 32), 16, 24, 3, 8, 0.3, 32),        # three real frames: K.rnn carries r, c, e across steps
    ("full64", (3, 48, 96, 192), 64, 64, 2, 123, 0.2, 33),   # train.py:51-55's model
    ("full72x88", (3, 48, 96, 192), 72, 88, 2, 124, 0.1, 34),
]


def _prednet_fixture(out):
    """Instantiates /root/reference/src/prednet.py's PredNet the way compress.py:163-173 does (weights=..., the train
    layer's config with output_mode='prediction'), lets its build() create the convolutions, and runs
    get_initial_state + step over the (1, T, Hp, Wp, 3) input of compress.py:224-227 (frame, then zeros).  Recorded
    per case: the names/shapes of `trainable_weights` in the order build() leaves them, the initial-state shapes, the
    state list (r, c, e per level) after every step, and X_hat.  The weight VALUES are not stored (27 MB for the full
    model): they are tezip_amd.prednet.PredNetConfig.init_weights(seed, bias_scale), whose sha256 is."""
    import hashlib
    sys.modules.pop("prednet", None)
    import prednet as ref_prednet
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tezip_amd.prednet import PredNetConfig
    out["pn_cases"] = np.array([c[0] for c in PREDNET_CASES])
    for name, stack, hp, wp, T, wseed, bscale, fseed in PREDNET_CASES:
        cfg = PredNetConfig(stack_sizes=stack)
        weights = cfg.init_weights(seed=wseed, bias_scale=bscale)
        L = len(stack)
        # train.py:57-60 builds the layer with output_mode='error', return_sequences=True; compress.py:163-168 takes
        # that layer's config, switches output_mode and passes the trained weight list back in through `weights=`
        layer_config = dict(stack_sizes=stack, R_stack_sizes=stack, A_filt_sizes=(3,) * (L - 1),
                            Ahat_filt_sizes=(3,) * L, R_filt_sizes=(3,) * L, output_mode="prediction",
                            return_sequences=True, data_format="channels_last")
        layer = ref_prednet.PredNet(weights=weights, **layer_config)
        rng = np.random.default_rng(fseed)
        X = np.zeros((1, T, hp, wp, 3), np.float32)
        nreal = T - 1 if T == 2 else T
        X[0, :nreal] = rng.integers(0, 256, size=(nreal, hp, wp, 3)).astype(np.float32) / np.float32(255)
        rec = []
        X_hat = layer(X, record=rec)
        pre = "pn_%s_" % name
        out[pre + "stack"] = np.array(stack, dtype=np.int64)
        out[pre + "hw"] = np.array([hp, wp], dtype=np.int64)
        out[pre + "wseed_bias"] = np.array([wseed, bscale], dtype=np.float64)
        out[pre + "weights_sha256"] = np.array(hashlib.sha256(b"".join(w.tobytes() for w in weights)).hexdigest())
        out[pre + "weight_names"] = np.array([v.name for v in layer.trainable_weights])
        out[pre + "weight_shapes"] = np.array([list(v.value.shape) + [0] * (4 - v.value.ndim)
                                               for v in layer.trainable_weights], dtype=np.int64)
        out[pre + "state_shapes"] = np.array([list(s.shape) for s in rec[0]], dtype=np.int64)
        assert all(not s.any() for s in rec[0])
        out[pre + "X"] = X
        out[pre + "X_hat"] = X_hat.astype(np.float32)
        for t in range(1, T + 1):
            for k, s in enumerate(rec[t]):
                # the full model: every state of the last step, of the first step only e (the errors against the real
                # frame, which are what the second step's gates read; r and c of step one do not depend on the input)
                if name.startswith("small") or t == T or "rce"[k // L] == "e":
                    out[pre + "t%d_%s%d" % (t, "rce"[k // L], k % L)] = s[0].astype(np.float32)
        print("prednet %-10s %d weights, X_hat %s, |X_hat[0,1]-X[0,0]| mean %.4f" % (
            name, len(layer.trainable_weights), X_hat.shape, float(np.abs(X_hat[0, 1] - X[0, 0]).mean())))


def main():
    _install_stubs()
    if "--prednet-only" in sys.argv:
        pn = {}
        _prednet_fixture(pn)
        np.savez_compressed(os.path.join(HERE, "ref_prednet.npz"), **pn)
        print("ref_prednet.npz:", len(pn), "arrays")
        return
    if "--train-only" in sys.argv:
        tr = {}
        _train_fixture(tr)
        np.savez_compressed(os.path.join(HERE, "ref_train.npz"), **tr)
        print("ref_train.npz:", len(tr), "arrays")
        return
    import compress
    import decompress
    import data_utils
    # the predictor layer is replaced wholesale by the fake model (see module docstring)
    compress.PredNet = decompress.PredNet = lambda weights=None, **cfg: (lambda inputs: inputs)
    if "--runs4-only" in sys.argv:
        runs4 = {}
        _runs(compress, decompress, runs4, RUNS_4, seed=780)
        np.savez_compressed(os.path.join(HERE, "ref_runs4.npz"), **runs4)
        print("ref_runs4.npz:", len(runs4), "arrays")
        return
    helpers = {}
    _helpers(compress, decompress, data_utils, helpers)
    np.savez_compressed(os.path.join(HERE, "ref_helpers.npz"), **helpers)
    print("ref_helpers.npz:", len(helpers), "arrays")
    longc = {}
    _long_chains(compress, longc)
    np.savez_compressed(os.path.join(HERE, "ref_long.npz"), **longc)
    print("ref_long.npz:", len(longc), "arrays")
    if not hasattr(np.ndarray, "tostring"):
        print("numpy >= 2: skipping compress.run/decompress.run goldens (needs ndarray.tostring)")
        return
    runs = {}
    _runs(compress, decompress, runs)
    np.savez_compressed(os.path.join(HERE, "ref_runs.npz"), **runs)
    print("ref_runs.npz:", len(runs), "arrays")
    runs2 = {}
    _runs(compress, decompress, runs2, RUNS_2, seed=778, make_frames=_make_frames_uneven)
    np.savez_compressed(os.path.join(HERE, "ref_runs2.npz"), **runs2)
    print("ref_runs2.npz:", len(runs2), "arrays")
    runs3 = {}
    _runs(compress, decompress, runs3, RUNS_3, seed=779)
    np.savez_compressed(os.path.join(HERE, "ref_runs3.npz"), **runs3)
    print("ref_runs3.npz:", len(runs3), "arrays")
    runs4 = {}
    _runs(compress, decompress, runs4, RUNS_4, seed=780)
    np.savez_compressed(os.path.join(HERE, "ref_runs4.npz"), **runs4)
    print("ref_runs4.npz:", len(runs4), "arrays")
    tr = {}
    _train_fixture(tr)
    np.savez_compressed(os.path.join(HERE, "ref_train.npz"), **tr)
    print("ref_train.npz:", len(tr), "arrays")
    pn = {}
    _prednet_fixture(pn)
    np.savez_compressed(os.path.join(HERE, "ref_prednet.npz"), **pn)
    print("ref_prednet.npz:", len(pn), "arrays")


if __name__ == "__main__":
    main()
