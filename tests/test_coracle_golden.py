"""Pins the C oracle's codec functions (oracle/tz_oracle.c through oracle/coracle.py) to the
reference's own outputs: tests/golden/ref_helpers.npz, ref_long.npz (helper outputs) and
ref_runs.npz / ref_runs2.npz / ref_runs3.npz (whole compress.run / decompress.run executions of the reference with
the stand-in predictor).  The GPU tests compare the HIP kernels with THIS library at sizes the
numpy oracle cannot reach, so the chain HIP -> C oracle -> reference must not have an unpinned link:
every tzo_* integer function is checked here against data the reference itself produced."""
import os

import numpy as np
import pytest

import fake_predictor
from conftest import GOLDEN
from oracle import coracle
from oracle import oracle as O

H = np.load(os.path.join(GOLDEN, "ref_helpers.npz"))
LC = np.load(os.path.join(GOLDEN, "ref_long.npz"))
R = np.load(os.path.join(GOLDEN, "ref_runs.npz"))
R2 = np.load(os.path.join(GOLDEN, "ref_runs2.npz"))
RUNS = {str(n): R for n in R["run_names"]}
RUNS.update({str(n): R2 for n in R2["run_names"]})
R3 = np.load(os.path.join(GOLDEN, "ref_runs3.npz"))
RUNS.update({str(n): R3 for n in R3["run_names"]})
R4 = np.load(os.path.join(GOLDEN, "ref_runs4.npz"))
RUNS.update({str(n): R4 for n in R4["run_names"]})
PRED = O.FnPredictor(fake_predictor.c0_image, fake_predictor.g_next)


def _eb_chain(orig, diff, mode, val):
    """tzo_error_bound on ONE chain (stride 1), as compress.py:316-319 calls error_bound per slab."""
    import ctypes as C
    o = np.ascontiguousarray(np.asarray(orig).reshape(-1), np.uint8)
    d = np.ascontiguousarray(np.asarray(diff).reshape(-1), np.int16).copy()
    v1 = float(val[1]) if len(val) > 1 else 0.0
    rc = coracle.lib().tzo_error_bound(C.c_void_p(o.ctypes.data), C.c_void_p(d.ctypes.data), o.size, 1,
                                       coracle.MODES[mode], float(val[0]), v1)
    assert rc == 0
    return d


@pytest.mark.parametrize("i", range(int(H["eb_n"])))
def test_c_error_bound_matches_reference(i):
    mode, val = str(H["eb_%d_mode" % i]), H["eb_%d_val" % i].tolist()
    got = _eb_chain(H["eb_%d_orig" % i], H["eb_%d_diff" % i], mode, val)
    np.testing.assert_array_equal(got, H["eb_%d_res" % i].reshape(-1))


@pytest.mark.parametrize("i", range(int(LC["lc_n"])))
def test_c_error_bound_long_chains_match_reference(i):
    """The frame form the GPU tests use (three channel chains of an HWC frame, stride 3)."""
    mode, val = str(LC["lc_%d_mode" % i]), LC["lc_%d_val" % i].tolist()
    got = coracle.error_bound_frame(LC["lc_%d_orig" % i], LC["lc_%d_diff" % i], mode, val)
    np.testing.assert_array_equal(got, LC["lc_%d_res" % i])


def test_c_error_bound_rejects_negative_pwrel():
    for mode, bound in (("pwrel", [-0.5]), ("rel", [-0.01]), ("absrel", [3.0, -0.2])):
        with pytest.raises(ValueError):
            coracle.error_bound_frame(np.zeros((2, 2, 3), np.uint8), np.zeros((2, 2, 3), np.int16), mode, bound)


def test_c_spatial_delta_both_ways_match_reference():
    for i in range(4):
        a = H["fd_enc_in_%d" % i]
        np.testing.assert_array_equal(coracle.spatial_delta(a.reshape(-1), 0), H["fd_enc_out_%d" % i].reshape(-1))
        np.testing.assert_array_equal(coracle.spatial_undelta(H["fd_enc_out_%d" % i].reshape(-1), 0),
                                      H["fd_dec_out_%d" % i].reshape(-1))
    w = H["fd_wrap_in"].reshape(-1)                      # int16 wrap-around in both directions
    np.testing.assert_array_equal(coracle.spatial_delta(w, 0), H["fd_wrap_enc"].reshape(-1))
    np.testing.assert_array_equal(coracle.spatial_undelta(w, 0), H["fd_wrap_dec"].reshape(-1))
    # the 1600 offset of compress.py:348 / decompress.py:236 folded into the same pass
    x = H["fd_enc_in_3"].reshape(-1)
    y = coracle.spatial_delta(x, 1)
    np.testing.assert_array_equal(y, (1600 - H["fd_enc_out_3"].reshape(-1).astype(np.int32)).astype(np.int16))
    np.testing.assert_array_equal(coracle.spatial_undelta(y, 1), x)


def _lut_enc(table):
    lut = np.arange(65536, dtype=np.int64) - 32768
    for idx, sym in enumerate(np.asarray(table).tolist()):
        lut[sym + 32768] = idx
    return lut.astype(np.int16)


def test_c_histogram_and_lut_match_reference_remap():
    y, table = H["rp_in"], H["rp_table"]
    hist = coracle.histogram(y)
    np.testing.assert_array_equal(hist, np.bincount(y.astype(np.int64), minlength=2111))
    np.testing.assert_array_equal(coracle.lut_apply(y, _lut_enc(table)), H["rp_enc"])
    np.testing.assert_array_equal(coracle.lut_apply(H["rp_dec_in"], O.unmap_lut(table).astype(np.int16)), H["rp_dec"])


def _table_from_hist(hist):
    syms = np.nonzero(hist)[0]
    order = sorted(zip(syms.tolist(), hist[syms].tolist()), key=lambda e: e[1], reverse=True)   # compress.py:356-359
    return np.array([s for s, _ in order], dtype=np.int16)


@pytest.mark.parametrize("name", sorted(RUNS))
def test_c_codec_reproduces_the_reference_run(name):
    """Every integer stage of a reference run through the C functions: per-frame delta, quantiser,
    spatial delta + offset, histogram -> table, remap -> the reference's entropy.dat (pre-zstd);
    then unmap, inverse scan and reconstruct -> the reference's decoded images; and the per-frame
    squared error sums -> the reference's MSE log."""
    G = RUNS[name]
    pre = "run_%s_" % name
    p, win, gray, entropy = (int(v) for v in G[pre + "params"])
    thr = float(G[pre + "thr"])
    mode, bound = str(G[pre + "mode"]), G[pre + "bound"].tolist()
    f = G[pre + "frames"]
    frames = f if f.ndim == 4 else np.repeat(f[..., None], 3, axis=-1)
    nt, h, w, _ = frames.shape
    ro = O.rollout(frames, p, None if win < 0 else win, None if thr < 0 else thr, PRED)  # pinned by test_oracle_golden
    delta = np.empty((nt, h, w, 3), np.int16)
    pred_of = {}
    for g, (start, preds) in enumerate(ro["groups"]):
        for j, pr in enumerate(preds):
            t = start + j
            pred_of[t] = pr
            d = coracle.delta_frame(pr, frames[t], zero=(j == 0))
            if j > 0 and not (p != 0 and g == 0):
                d = coracle.error_bound_frame(frames[t], d, mode, bound)
            delta[t] = d
    ref_payload, ref_table, shape, warm = O.parse_stream(G[pre + "entropy"])
    assert shape == (1, nt, h, w, 3) and warm == p
    if entropy:
        y = coracle.spatial_delta(delta.reshape(-1), 1)
        table = _table_from_hist(coracle.histogram(y))
        np.testing.assert_array_equal(table, ref_table)
        payload = coracle.lut_apply(y, _lut_enc(table))
    else:
        assert ref_table is None
        payload = coracle.spatial_delta(delta.reshape(-1), 0)
    np.testing.assert_array_equal(payload, ref_payload)
    # decoder side from the reference's own files
    if entropy:
        sym = coracle.lut_apply(ref_payload, O.unmap_lut(ref_table).astype(np.int16))
        back = coracle.spatial_undelta(sym, 1)
    else:
        back = coracle.spatial_undelta(ref_payload, 0)
    np.testing.assert_array_equal(back, delta.reshape(-1))
    key = ro["key"]
    dec = np.empty_like(frames)
    for t in range(nt):
        # decompress.py:141-158,186: frame 0 and the key frames from index warm_up on are rebuilt from
        # the key byte; warm-up frames 1..p-1 are key frames too but are decoded against C0
        from_key = t == 0 or (key[t] and t >= p)
        dec[t] = coracle.reconstruct_frame(None if from_key else pred_of[t], frames[t] if from_key else None, delta[t])
    np.testing.assert_array_equal(dec, G[pre + "decoded"])
    # compress.py:245-249: window MSE = running sum of per-frame squared-error sums / element count,
    # over the predictions in the order the reference made them (dropped boundary ones included)
    hp, wp = ro["x_pad"].shape[1:3]
    got, run, last = [], 0.0, None
    key_idx = p + 1
    for idx in range(p + 1, nt):
        if idx == key_idx:
            last, run = coracle.u8_to_f32_frame(frames[idx - 1], hp, wp), 0.0
        last = np.asarray(PRED.next(last), np.float32)
        run += coracle.sse_frame(frames[idx], last)
        got.append(run / ((idx - key_idx + 1) * hp * wp * 3))
        if (thr >= 0 and got[-1] > thr) or (win > 0 and (idx - p) % win == 0):
            key_idx = idx + 1
    np.testing.assert_allclose(got, G[pre + "mse"], rtol=1e-12)
