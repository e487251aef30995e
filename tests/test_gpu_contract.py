"""The arithmetic contract is part of a job (decompress.py:252-253 needs the decoder's predictions bit-identical to
the encoder's): recorded next to entropy.dat, adopted by -u, enforced inside the C ABI between a rollout and its
encode / decode.  Round 5; VERDICT r04 item 4 / ADVICE r04 (medium)."""
import json
import os

import numpy as np
import pytest

from tezip_amd import _lib, compress, decompress, sidecar, synth, tezip, weights
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu
SMALL = PredNetConfig(stack_sizes=(3, 16, 32))
FULL = PredNetConfig()


def _write(tmp, frames):
    from PIL import Image
    d = tmp / "data"
    d.mkdir()
    for t in range(frames.shape[0]):
        Image.fromarray(frames[t], mode="RGB").save(d / ("frame_%03d.png" % t))
    return str(d)


def _read(udir, nt):
    from PIL import Image
    return np.stack([np.array(Image.open(os.path.join(udir, "frame_%03d.png" % t))) for t in range(nt)])


@pytest.fixture
def job(tmp_path, monkeypatch):
    """A lossless job of the FULL model at 64x96 (where TZ-PA1 and TZ-PA2 differ in their bits) compressed under --pa 2."""
    monkeypatch.delenv("TEZIP_PA", raising=False)
    nt, h, w = 7, 64, 96
    frames = synth.translating_scene(nt, h, w, seed=21)
    wts = FULL.init_weights(seed=6, bias_scale=0.1)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, FULL, wts, 64, 96)
    ddir = _write(tmp_path, frames)
    cdir = str(tmp_path / "comp")
    args = tezip.build_parser().parse_args(["-c", mdir, ddir, cdir, "-p", "0", "-w", "3", "-m", "abs", "-b", "0", "--pa", "2"])
    tezip.main(args)
    assert "TEZIP_PA" not in os.environ                 # --pa lives for the run only (tezip.main puts the variable back)
    return dict(frames=frames, mdir=mdir, cdir=cdir, nt=nt, tmp=tmp_path)


def test_compress_records_the_contract_and_uncompress_adopts_it(job, monkeypatch):
    doc = json.load(open(os.path.join(job["cdir"], sidecar.NAME)))
    assert doc["contract"] == 2 and doc["arithmetic_contract"] == "TZ-PA2" and doc["padded_frame"] == [64, 96]
    # no --pa on the decoding side; by frame size 64x96 would be TZ-PA1 -- the recorded contract wins
    udir = str(job["tmp"] / "out")
    tezip.main(tezip.build_parser().parse_args(["-u", job["mdir"], job["cdir"], udir]))
    np.testing.assert_array_equal(_read(udir, job["nt"]), job["frames"])


def test_contradicting_pa_is_refused_with_a_message(job, capsys):
    udir = str(job["tmp"] / "out")
    with pytest.raises(SystemExit) as stop:
        tezip.main(tezip.build_parser().parse_args(["-u", job["mdir"], job["cdir"], udir, "--pa", "1"]))
    assert stop.value.code == 2                       # a refused decode is not a success
    out = capsys.readouterr().out
    assert "ERROR:" in out and "TZ-PA2" in out and "TZ-PA1" in out
    assert not os.path.exists(udir) or not [f for f in os.listdir(udir) if f.endswith(".png")]


def test_another_model_is_refused(job, capsys, monkeypatch):
    monkeypatch.delenv("TEZIP_PA", raising=False)
    other = str(job["tmp"] / "other_model")
    weights.save_model(other, FULL, FULL.init_weights(seed=7, bias_scale=0.1), 64, 96)
    with pytest.raises(SystemExit) as stop:
        decompress.run(other, job["cdir"], str(job["tmp"] / "out"), True, False)
    assert stop.value.code == 2
    assert "not the model" in capsys.readouterr().out


def test_refused_decode_ends_the_process_with_status_2(job):
    """The same refusal seen by a launcher: `python -m tezip_amd.tezip -u ... --pa 1` returns 2, not 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "tezip_amd.tezip", "-u", job["mdir"], job["cdir"], str(job["tmp"] / "out2"), "--pa", "1"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert "ERROR:" in r.stdout


def test_directory_without_sidecar_decodes_by_the_old_rule(job, monkeypatch):
    """What a directory written by the reference (or by a build of rounds 1-4) looks like: three files.  It still
    decodes -- under --pa if given, else by frame size; for THIS job (encoded under TZ-PA2 at a TZ-PA1 size) that means
    exact with --pa 2 and within one grey level without."""
    os.remove(os.path.join(job["cdir"], sidecar.NAME))
    monkeypatch.setenv("TEZIP_PA", "2")
    u2 = str(job["tmp"] / "out2")
    decompress.run(job["mdir"], job["cdir"], u2, True, False)
    np.testing.assert_array_equal(_read(u2, job["nt"]), job["frames"])
    monkeypatch.delenv("TEZIP_PA")
    u0 = str(job["tmp"] / "out0")
    decompress.run(job["mdir"], job["cdir"], u0, True, False)
    assert np.abs(_read(u0, job["nt"]).astype(int) - job["frames"].astype(int)).max() <= 1


@pytest.mark.parametrize("h,w,expect", [(256, 256, 2), (248, 256, 1)])
def test_default_contract_boundary_round_trip(tmp_path, monkeypatch, h, w, expect):
    """contract == 0 on both sides: TZ-PA2 from 256 x 256 padded pixels on, TZ-PA1 below (tz_prednet.hip
    effective_contract); the stamp, the sidecar and a lossless round trip on either side of the boundary."""
    monkeypatch.delenv("TEZIP_PA", raising=False)
    nt = 5
    frames = synth.turbulence(nt, h, w, seed=4)
    wts = FULL.init_weights(seed=123)
    ctx = _lib.Context(0)
    try:
        ctx.load_model(FULL, wts)
        ctx.prepare(_lib.pad8(h), _lib.pad8(w), max_batch=2)
        assert ctx.get_contract() == expect
        key, _ = ctx.rollout(frames, 0, 2)
        assert ctx.rollout_contract() == expect
        payload, table, _ = ctx.encode("abs", [0.0], True)
        kf = np.zeros_like(frames)
        kf[key] = frames[key]
        ctx.rollout_decode(kf, 0)
        assert ctx.rollout_contract() == expect
        np.testing.assert_array_equal(ctx.decode(payload, table), frames)
    finally:
        ctx.close()
    mdir, cdir, udir = (str(tmp_path / n) for n in ("model", "comp", "out"))
    weights.save_model(mdir, FULL, wts, _lib.pad8(h), _lib.pad8(w))
    compress.run(mdir, _write(tmp_path, frames), cdir, 0, 2, None, "abs", [0.0], True, False, True)
    assert sidecar.read(cdir)["contract"] == expect
    decompress.run(mdir, cdir, udir, True, False)
    np.testing.assert_array_equal(_read(udir, nt), frames)


def test_abi_refuses_a_flip_between_rollout_and_encode_or_decode():
    ctx = _lib.Context(0)
    try:
        w = FULL.init_weights(seed=9, bias_scale=0.1)
        frames = synth.translating_scene(6, 64, 96, seed=3)
        ctx.load_model(FULL, w)
        ctx.prepare(64, 96, max_batch=2)
        with pytest.raises(_lib.TezipError) as e:
            ctx.rollout_contract()                       # nothing rolled out yet
        assert e.value.status == -4
        ctx.set_contract(2)
        key, _ = ctx.rollout(frames, 0, 3)
        assert ctx.rollout_contract() == 2
        ctx.set_contract(1)
        for call in (lambda: ctx.encode("abs", [0.0], True), lambda: ctx.encode_delta("abs", [0.0]),
                     lambda: ctx.encode_begin("abs", [0.0], True)):
            with pytest.raises(_lib.TezipError, match="TZ-PA2.*TZ-PA1") as e:
                call()
            assert e.value.status == -4                  # TZ_ERR_STATE
        ctx.set_contract(2)                               # back: the same stack encodes
        payload, table, _ = ctx.encode("abs", [0.0], True)
        kf = np.zeros_like(frames)
        kf[key] = frames[key]
        ctx.rollout_decode(kf, 0)
        ctx.set_contract(0)                               # 64 x 96: by size = TZ-PA1, the stack is TZ-PA2
        with pytest.raises(_lib.TezipError) as e:
            ctx.decode(payload, table)
        assert e.value.status == -4
        ctx.set_contract(2)
        np.testing.assert_array_equal(ctx.decode(payload, table), frames)
    finally:
        ctx.close()


def test_library_is_a_product_build():
    assert _lib.diagnostic_defines() == [] and _lib.build_info().startswith("tezip_hip ")
