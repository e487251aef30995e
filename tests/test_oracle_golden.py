"""Pins the CPU oracle (oracle/oracle.py) against the reference's own outputs
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference) and the
doc KATs listed in SURVEY.md §4.2."""
import os

import numpy as np
import pytest

import fake_predictor
from conftest import GOLDEN
from oracle import oracle as O

H = np.load(os.path.join(GOLDEN, "ref_helpers.npz"))
R = np.load(os.path.join(GOLDEN, "ref_runs.npz"))
R2 = np.load(os.path.join(GOLDEN, "ref_runs2.npz"))   # DWP runs with windows of mixed length (round 3)
LC = np.load(os.path.join(GOLDEN, "ref_long.npz"))    # error_bound of the reference on 12k..65k-element chains
RUNS = {str(n): R for n in R["run_names"]}
RUNS.update({str(n): R2 for n in R2["run_names"]})
R3 = np.load(os.path.join(GOLDEN, "ref_runs3.npz"))   # edge cases: one-frame windows, one window, shortest sequence, DWP extremes
RUNS.update({str(n): R3 for n in R3["run_names"]})
R4 = np.load(os.path.join(GOLDEN, "ref_runs4.npz"))   # round 6: tolerances that cannot merge different deltas (E <= 0.499), unpadded frames
RUNS.update({str(n): R4 for n in R4["run_names"]})
PRED = O.FnPredictor(fake_predictor.c0_image, fake_predictor.g_next)


@pytest.mark.parametrize("i", range(int(H["eb_n"])))
def test_error_bound_matches_reference(i):
    mode, val = str(H["eb_%d_mode" % i]), H["eb_%d_val" % i].tolist()
    got = O.error_bound(H["eb_%d_orig" % i], H["eb_%d_diff" % i], mode, val)
    np.testing.assert_array_equal(got, H["eb_%d_res" % i])


@pytest.mark.parametrize("i", range(int(LC["lc_n"])))
def test_error_bound_long_chains_match_reference(i):
    """Chains of 12,288 / 13,100 / 65,536 elements computed by the reference itself: runs shorter
    than, equal to and far longer than the 64-element chunks and 1/8-chain segments of the HIP
    quantiser (tests/golden/make_golden.py::_long_chains)."""
    mode, val = str(LC["lc_%d_mode" % i]), LC["lc_%d_val" % i].tolist()
    orig, diff, res = (LC["lc_%d_%s" % (i, k)] for k in ("orig", "diff", "res"))
    for c in range(3):
        got = O.error_bound(orig[..., c], diff[..., c], mode, val)
        np.testing.assert_array_equal(got, res[..., c], err_msg="channel %d" % c)


def test_error_bound_doc_kat():
    # docs/img/img33.png: float inputs give x.5 medians; the int path truncates them
    E = np.array([6, 4, 2, 4, 2, 6, 2, 2, 6])
    D = np.array([0, 0, -5, -5, 10, 5, -5, -5, 0])
    np.testing.assert_array_equal(H["eb_doc_float"], [-3.5] * 4 + [9.5] * 2 + [-4.5] * 3)
    np.testing.assert_array_equal(O.error_bound(E, D, "pwrel", [1.0]), [-3] * 4 + [9] * 2 + [-4] * 3)


def test_error_bound_rejects_negative_pwrel():
    for mode, bound in (("pwrel", [-0.1]), ("rel", [-0.01]), ("absrel", [3.0, -0.2])):
        with pytest.raises(ValueError):
            O.error_bound(np.array([1, 2]), np.array([0, 1]), mode, bound)


def test_finding_difference_both_ways():
    for i in range(4):
        a = H["fd_enc_in_%d" % i]
        np.testing.assert_array_equal(O.finding_difference_enc(a), H["fd_enc_out_%d" % i])
        np.testing.assert_array_equal(O.finding_difference_dec(H["fd_enc_out_%d" % i]), H["fd_dec_out_%d" % i])
        np.testing.assert_array_equal(H["fd_dec_out_%d" % i], a)
    np.testing.assert_array_equal(O.finding_difference_enc(H["fd_wrap_in"]), H["fd_wrap_enc"])
    np.testing.assert_array_equal(O.finding_difference_dec(H["fd_wrap_in"]), H["fd_wrap_dec"])
    # doc KATs (docs/index.rst:1185-1198 / 1362-1372); the figure's 5th entry has a sign slip,
    # the reference code gives in[i-1]-in[i] = 8-4 = +4
    np.testing.assert_array_equal(O.finding_difference_enc(np.array([2, 5, 8, 8, 4, 4, 5, 6, 6])),
                                  [2, -3, -3, 0, 4, 0, -1, -1, 0])


def test_remap_both_ways():
    np.testing.assert_array_equal(O.remap_enc(H["rp_in"], H["rp_table"]), H["rp_enc"])
    np.testing.assert_array_equal(O.remap_dec(H["rp_dec_in"], H["rp_table"]), H["rp_dec"])
    # doc KAT (docs/index.rst:1200-1236, offset omitted in the figure)
    y = np.array([0, 5, 5, 5, 4, 5, 4, 4, 5]) + 1090
    t = O.build_table(y)
    np.testing.assert_array_equal(t, np.array([5, 4, 0]) + 1090)
    np.testing.assert_array_equal(O.remap_enc(y, t), [2, 0, 0, 0, 1, 0, 1, 1, 0])
    # tie-break: equal counts keep ascending symbol order
    y = np.array([1600] * 4 + [1595] * 2 + [1605, 1598])
    np.testing.assert_array_equal(O.build_table(y), [1600, 1595, 1598, 1605])


def test_unmap_chain_semantics_match_sequential_passes():
    # a (non-reference) table whose symbols are themselves valid ranks
    rng = np.random.default_rng(5)
    table = np.array([3, 0, 5, 4, 9, 1], dtype=np.int16)
    ranks = rng.integers(-2, 12, size=200).astype(np.int16)
    ref = ranks.copy()
    for idx, num in enumerate(table):  # decompress.py:31-36 semantics
        ref = np.where(ref == idx, num, ref)
    np.testing.assert_array_equal(O.remap_dec(ranks, table), ref)


def test_padding():
    np.testing.assert_array_equal(O.data_padding(H["pad_in"]), H["pad_out"])
    assert O.data_padding(H["pad_in"]).dtype == np.float64
    np.testing.assert_array_equal([O.padding_size(int(v)) for v in H["pad_sizes_in"]], H["pad_sizes_out"])


def test_uint8_scale_identities():
    k = np.arange(256, dtype=np.uint8)
    assert ((k / 255 * 255.0).astype(int) == k).all()  # compress.py:294,308,310
    assert (((k / 255) * 255).astype(np.uint8) == k).all()  # decompress.py:117,252,269
    # decoder feeds float64(k/255) to a float32 model; encoder feeds float32(k)/255
    assert ((k / 255).astype(np.float32) == k.astype(np.float32) / np.float32(255)).all()


def _frames3(name):
    f = RUNS[name]["run_%s_frames" % name]
    return f if f.ndim == 4 else np.repeat(f[..., None], 3, axis=-1)


@pytest.mark.parametrize("name", sorted(RUNS))
def test_full_run_matches_reference(name):
    pre = "run_%s_" % name
    R = RUNS[name]
    p, win, gray, entropy = (int(v) for v in R[pre + "params"])
    thr = float(R[pre + "thr"])
    frames = _frames3(name)
    enc = O.compress_oracle(frames, p, None if win < 0 else win, None if thr < 0 else thr,
                            str(R[pre + "mode"]), R[pre + "bound"].tolist(), PRED, bool(entropy))
    np.testing.assert_array_equal(enc["key_frame"], R[pre + "key_frame"])
    np.testing.assert_array_equal(enc["stream"], R[pre + "entropy"])
    np.testing.assert_allclose(enc["mse"], R[pre + "mse"], rtol=1e-12)
    names = ["frame_%03d.png" % t for t in range(frames.shape[0])]
    assert O.filename_txt(names, not gray) == str(R[pre + "filename_txt"])
    dec = O.decode_stream(R[pre + "entropy"], R[pre + "key_frame"], PRED)
    np.testing.assert_array_equal(dec, R[pre + "decoded"])
    # reference call pattern: one (1,2,..) predict per frame after the first (+p warm-ups)
    assert len(R[pre + "enc_calls"]) == frames.shape[0] - 1
    assert (R[pre + "enc_calls"][:, 1] == 2).all()


def test_mixed_dwp_goldens_really_have_windows_of_mixed_length():
    for name in (str(n) for n in R2["run_names"]):
        pre = "run_%s_" % name
        p = int(R2[pre + "params"][0])
        f = R2[pre + "frames"]
        kf = R2[pre + "key_frame"].reshape((f.shape[0], f.shape[1], f.shape[2], 3))
        keys = [i for i in range(f.shape[0]) if kf[i].any()]
        lens = set(np.diff(keys[p:]).tolist())
        assert len(lens) >= 3, (name, keys)


def test_short_sequences_rejected():
    f = np.zeros((2, 8, 8, 3), np.uint8)
    with pytest.raises(ValueError):
        O.rollout(f[:1], 0, 2, None, PRED)
    with pytest.raises(ValueError):
        O.rollout(f, 1, 2, None, PRED)


def test_f32_product_truncation_is_exact():
    """trunc(f32(p*255)) == floor(exact p*255): the encoder's float32 product
    (compress.py:307) and the decoder's float64 product (decompress.py:252) agree, so the
    float64 reconstruct equals the integer form used by the HIP kernel."""
    rng = np.random.default_rng(11)
    k = np.arange(0, 256, dtype=np.float32)
    centre = (k / np.float32(255)).view(np.uint32).astype(np.int64)
    near = (centre[:, None] + np.arange(-512, 513)[None, :]).clip(0, 0x3F800000).astype(np.uint32).view(np.float32)
    p = np.concatenate([near.reshape(-1), rng.random(4_000_000, dtype=np.float32), np.float32([0, 1])])
    a = (p * np.float32(255.0)).astype(np.int64)
    b = np.floor(p.astype(np.float64) * 255).astype(np.int64)
    np.testing.assert_array_equal(a, b)
    d = rng.integers(-300, 300, size=p.shape).astype(np.int16)
    np.testing.assert_array_equal(O.reconstruct(p.astype(np.float64), d), O.reconstruct_int(a, d))
