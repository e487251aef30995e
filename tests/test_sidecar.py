"""tezip_amd.json (tezip_amd/sidecar.py): the record of the arithmetic contract next to entropy.dat, and the rule a
decoder resolves it by.  CPU only: no context is opened."""
import json
import os

import numpy as np
import pytest

from tezip_amd import sidecar
from tezip_amd.prednet import PredNetConfig

CFG = PredNetConfig(stack_sizes=(3, 16, 32))
W = CFG.init_weights(seed=2, bias_scale=0.1)


def test_written_fields_and_round_trip(tmp_path):
    doc = sidecar.write(str(tmp_path), 2, W, 512, 512)
    assert doc["arithmetic_contract"] == "TZ-PA2" and doc["contract"] == 2 and doc["arch"] == "gfx950"
    assert doc["padded_frame"] == [512, 512] and doc["tz_version"] >= 101 and len(doc["weights_sha256"]) == 64
    assert json.load(open(tmp_path / sidecar.NAME)) == doc == sidecar.read(str(tmp_path))
    with pytest.raises(ValueError):
        sidecar.write(str(tmp_path), 0, W, 8, 8)      # "by frame size" is a rule, not a contract: never recorded


def test_absent_sidecar_keeps_the_rule_of_earlier_rounds(tmp_path, monkeypatch):
    """A directory written by the reference (three files) or by a build before round 5."""
    assert sidecar.read(str(tmp_path)) is None
    monkeypatch.delenv("TEZIP_PA", raising=False)
    assert sidecar.resolve(None, W) is None            # by frame size
    monkeypatch.setenv("TEZIP_PA", "1")
    assert sidecar.resolve(None, W) == 1
    monkeypatch.setenv("TEZIP_PA", "0")
    assert sidecar.resolve(None, W) is None


def test_recorded_contract_is_adopted_and_a_contradiction_refused(tmp_path, monkeypatch):
    sidecar.write(str(tmp_path), 2, W, 256, 256)
    doc = sidecar.read(str(tmp_path))
    monkeypatch.delenv("TEZIP_PA", raising=False)
    assert sidecar.resolve(doc, W) == 2
    monkeypatch.setenv("TEZIP_PA", "2")
    assert sidecar.resolve(doc, W) == 2
    monkeypatch.setenv("TEZIP_PA", "0")
    assert sidecar.resolve(doc, W) == 2
    monkeypatch.setenv("TEZIP_PA", "1")
    with pytest.raises(sidecar.SidecarMismatch, match="TZ-PA2.*TZ-PA1"):
        sidecar.resolve(doc, W)


def test_another_model_is_refused(tmp_path, monkeypatch):
    monkeypatch.delenv("TEZIP_PA", raising=False)
    sidecar.write(str(tmp_path), 1, W, 64, 64)
    other = [w.copy() for w in W]
    other[3][0] += np.float32(1e-3)
    with pytest.raises(sidecar.SidecarMismatch, match="not the model"):
        sidecar.resolve(sidecar.read(str(tmp_path)), other)
    assert sidecar.resolve(sidecar.read(str(tmp_path)), [w.copy() for w in W]) == 1


@pytest.mark.parametrize("text", ["", "{", '{"format": 2, "contract": 1}', '{"format": 1, "contract": 3}', '{"format": 1}'])
def test_damaged_sidecar_is_an_error_not_a_guess(tmp_path, text):
    open(tmp_path / sidecar.NAME, "w").write(text)
    with pytest.raises(sidecar.SidecarMismatch, match="damaged"):
        sidecar.read(str(tmp_path))


def test_reference_decoder_never_opens_it():
    """The claim the design rests on: decompress.py of the reference opens filename.txt, key_frame.dat and entropy.dat
    by name and never lists its input directory (checked where the reference is present: the build container)."""
    ref = "/root/reference/src/decompress.py"
    if not os.path.exists(ref):
        pytest.skip("reference not present on this box")
    src = open(ref, encoding="utf-8").read()
    assert "listdir" not in src and "glob" not in src and "scandir" not in src
    for name in ("filename.txt", "key_frame.dat", "entropy.dat"):
        assert name in src


def test_a_refused_decode_exits_with_status_2(tmp_path, capsys, monkeypatch):
    """decompress.adopt_contract: a damaged / contradicting tezip_amd.json ends the run with exit status 2, not the
    reference's `exit()` (status 0) -- this error class does not exist in the reference.  No context is
    opened on the way: the refusal comes before any GPU work."""
    from tezip_amd import decompress
    monkeypatch.delenv("TEZIP_PA", raising=False)
    open(tmp_path / sidecar.NAME, "w").write("{")
    with pytest.raises(SystemExit) as stop:
        decompress.adopt_contract(str(tmp_path), W, False)
    assert stop.value.code == 2 and "damaged" in capsys.readouterr().out
    sidecar.write(str(tmp_path), 2, W, 256, 256)
    monkeypatch.setenv("TEZIP_PA", "1")
    with pytest.raises(SystemExit) as stop:
        decompress.adopt_contract(str(tmp_path), W, False)
    assert stop.value.code == 2
    monkeypatch.setenv("TEZIP_PA", "2")
    assert decompress.adopt_contract(str(tmp_path), W, False) == 2


def test_stack_is_recorded_and_only_a_plausible_one_is_used(tmp_path):
    """Round 6: `stack` = [frames, height, width, warm_up] lets the decoder queue its rollout while entropy.dat (whose LAST
    values hold the same numbers, compress.py:390-394) is still being decompressed.  Optional and additive: sidecars of
    round 5 have none; junk is ignored, never trusted."""
    doc = sidecar.write(str(tmp_path), 2, W, 512, 512, (80, 512, 510, 2))
    assert doc["stack"] == [80, 512, 510, 2] and sidecar.stack_of(sidecar.read(str(tmp_path))) == (80, 512, 510, 2)
    assert sidecar.stack_of(None) is None
    assert sidecar.stack_of(sidecar.write(str(tmp_path), 1, W, 64, 64)) is None
    for junk in ([80, 512, 510], [0, 512, 510, 0], [80, 512, 510, 80], [80, 512, 40000, 0], ["80", 512, 510, 0], "80x512", [80.5, 512, 510, 0]):
        assert sidecar.stack_of({"stack": junk}) is None, junk


def test_prefetch_hands_over_the_stream_in_order_and_raises_where_the_consumer_iterates(tmp_path):
    """decompress._Prefetch: a zstd frame decompressed on a worker thread through a ring of buffers.  The pieces arrive in
    order and complete; a truncated file raises in the consumer; a consumer that stops early does not leave the worker
    blocked."""
    from tezip_amd import decompress, zstd
    rng = np.random.default_rng(5)
    data = rng.integers(0, 7, 3_000_001, dtype=np.uint8)          # not a multiple of the piece size
    path = tmp_path / "x.zst"
    path.write_bytes(zstd.compress_array(data, 3, 0))
    pre = decompress._Prefetch(str(path), piece_bytes=1 << 18, depth=2)
    got, sizes = [], set()
    try:
        for size, piece in pre:
            sizes.add(size)
            got.append(piece.copy())     # the ring reuses a buffer depth + 2 pieces later
    finally:
        pre.close()
    assert sizes == {data.size} and len(got) == 12
    np.testing.assert_array_equal(np.concatenate(got), data)
    (tmp_path / "cut.zst").write_bytes(path.read_bytes()[: path.stat().st_size // 2])
    pre = decompress._Prefetch(str(tmp_path / "cut.zst"), piece_bytes=1 << 18, depth=2)
    with pytest.raises(RuntimeError, match="zstd"):
        try:
            for _ in pre:
                pass
        finally:
            pre.close()
    pre = decompress._Prefetch(str(path), piece_bytes=1 << 16, depth=2)   # 46 pieces, the consumer takes one and leaves
    it = iter(pre)
    next(it)
    pre.close()
    assert not pre.t.is_alive()
