"""End-to-end through the reference-shaped entry points on a MI355X: PNG directory ->
compress.run -> {filename.txt, key_frame.dat, entropy.dat} -> decompress.run -> PNGs, and the
files are decoded by the ORACLE's decoder too (the restatement of the reference's decompressor,
pinned to it by tests/test_oracle_golden.py), i.e. the on-disk format is the reference's."""
import os

import numpy as np
import pytest

from oracle import coracle
from oracle import oracle as O
from tezip_amd import compress, decompress, synth, weights, zstd
from tezip_amd.prednet import PredNetConfig

pytestmark = pytest.mark.gpu


def _write(tmp, frames, gray):
    from PIL import Image
    d = tmp / "data"
    d.mkdir()
    for t in range(frames.shape[0]):
        img = frames[t, :, :, 0] if gray else frames[t]
        Image.fromarray(img, mode="L" if gray else "RGB").save(d / ("frame_%03d.png" % t))
    return str(d)


@pytest.mark.parametrize("gray,p,window,thr,mode,bound,entropy", [
    (False, 0, 5, None, "abs", [0.0], True),
    (True, 2, 4, None, "abs", [3.0], True),
    (False, 0, None, 0.02, "rel", [0.01], False),
])
def test_cli_roundtrip_and_reference_format(tmp_path, gray, p, window, thr, mode, bound, entropy):
    from PIL import Image
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    nt, h, w = 13, 29, 43
    frames = synth.translating_scene(nt, h, w, seed=5)
    if gray:
        frames = np.repeat(frames[..., :1], 3, axis=-1)
    hp, wp = 32, 48
    wts = cfg.init_weights(seed=4, bias_scale=0.1)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, wts, hp, wp)
    ddir = _write(tmp_path, frames, gray)
    cdir, udir = str(tmp_path / "comp"), str(tmp_path / "out")
    compress.run(mdir, ddir, cdir, p, window, thr, mode, bound, True, True, entropy)
    assert sorted(os.listdir(cdir)) == ["entropy.dat", "filename.txt", "key_frame.dat", "tezip_amd.json"]
    names = ["frame_%03d.png" % t for t in range(nt)]
    assert open(os.path.join(cdir, "filename.txt")).read() == O.filename_txt(names, not gray)
    key_bytes = np.frombuffer(zstd.decompress(open(os.path.join(cdir, "key_frame.dat"), "rb").read()), np.uint8)
    stream = np.frombuffer(zstd.decompress(open(os.path.join(cdir, "entropy.dat"), "rb").read()), "<i2")

    class P:
        net = coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, hp, wp)

        def c0(self, a, b):
            return self.net.c0()

        def next(self, f):
            return self.net.next(np.asarray(f, np.float32))

    ref = O.compress_oracle(frames, p, window, thr, mode, bound, P(), entropy)
    np.testing.assert_array_equal(key_bytes, ref["key_frame"])
    np.testing.assert_array_equal(stream, ref["stream"])       # pre-zstd bytes identical to the oracle's
    oracle_dec = O.decode_stream(stream, key_bytes, P())         # the reference's decoder (restated) reads our files
    decompress.run(mdir, cdir, udir, True, False)
    got = np.stack([np.array(Image.open(os.path.join(udir, n))) for n in names])
    assert got.shape == (nt, h, w, 3)                            # decompress.py:278 always saves RGB
    np.testing.assert_array_equal(got, oracle_dec)
    if bound[0] == 0:
        np.testing.assert_array_equal(got, frames)
    else:
        assert np.abs(got.astype(int) - frames.astype(int)).max() <= (int(bound[0]) + 1 if mode == "abs" else 4)


def test_wrong_model_size_is_reported_like_the_reference(tmp_path, capsys):
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=1), 64, 64)
    ddir = _write(tmp_path, synth.translating_scene(4, 16, 16, seed=1), False)
    with pytest.raises(SystemExit):
        compress.run(mdir, ddir, str(tmp_path / "c"), 0, 2, None, "abs", [0.0], True, False, True)
    assert "ERROR:Image size is out of scope for this model." in capsys.readouterr().out


def test_cli_with_the_reference_model_files(tmp_path):
    """Model directory exactly as the reference leaves it: prednet_model.json + Keras
    prednet_weights.hdf5 (fixture written by real h5py, read by the built-in reader)."""
    from PIL import Image
    from conftest import GOLDEN
    mdir = os.path.join(GOLDEN, "keras_style_model")
    frames = synth.translating_scene(9, 16, 24, seed=12)
    ddir = _write(tmp_path, frames, False)
    cdir, udir = str(tmp_path / "comp"), str(tmp_path / "out")
    compress.run(mdir, ddir, cdir, 0, 3, None, "abs", [0.0], True, False, True)
    decompress.run(mdir, cdir, udir, True, False)
    got = np.stack([np.array(Image.open(os.path.join(udir, "frame_%03d.png" % t))) for t in range(9)])
    np.testing.assert_array_equal(got, frames)


def test_cli_under_torchrun_shards_windows_and_writes_identical_files(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m tezip_amd.tezip -c/-u ...`: the two
    ranks (gloo transport, both on GPU 0 here) shard the windows; the files must be byte-identical
    to the single-process run and decode to the same images."""
    import socket
    import subprocess
    import sys
    from PIL import Image
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    nt, h, w = 14, 24, 40
    frames = synth.translating_scene(nt, h, w, seed=15)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=8, bias_scale=0.1), 24, 40)
    ddir = _write(tmp_path, frames, False)
    one, two, out2 = str(tmp_path / "one"), str(tmp_path / "two"), str(tmp_path / "out2")
    compress.run(mdir, ddir, one, 1, 4, None, "abs", [2.0], True, False, True)

    def torchrun(args):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, TEZIP_DIST_BACKEND="gloo", TEZIP_SINGLE_DEVICE="1", TEZIP_IO_LOG=str(iolog))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "-m", "tezip_amd.tezip"] + args
        r = subprocess.run(cmd, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

    def ranges(what, rank):
        return [tuple(int(v) for v in r.split(":")) for r in open(iolog / ("%s.rank%d" % (what, rank))).read().split()]

    iolog = tmp_path / "iolog"
    iolog.mkdir()
    torchrun(["-c", mdir, ddir, two, "-p", "1", "-w", "4", "-m", "abs", "-b", "2"])
    for name in ("filename.txt", "key_frame.dat", "entropy.dat"):
        assert open(os.path.join(one, name), "rb").read() == open(os.path.join(two, name), "rb").read(), name
    # rank-local I/O (compress.py:97-122 is one loop over every file): a rank decodes the images of its own windows,
    # rank 0 in addition the key frames of the other rank's windows (one file per window) for key_frame.dat
    from tezip_amd import dist as tzdist
    (a0, b0), (a1, b1) = tzdist.plan_shards(nt, 1, 4, 2)
    assert (a0, b0, b1) == (0, a1, nt) and 0 < a1 < nt
    assert ranges("compress", 1) == [(a1, b1)]
    r0 = ranges("compress", 0)
    assert r0[0] == (a0, b0) and all(b - a == 1 and a >= a1 for a, b in r0[1:]) and len(r0) - 1 <= (b1 - a1 + 3) // 4 + 1
    torchrun(["-u", mdir, two, out2])
    # ... and writes the images of its own key intervals (decompress.py:266-279), nothing gathered on rank 0
    w0, w1 = ranges("decompress", 0), ranges("decompress", 1)
    assert len(w0) == len(w1) == 1 and w0[0][0] == 0 and w0[0][1] == w1[0][0] and w1[0][1] == nt and 0 < w0[0][1] < nt
    got = np.stack([np.array(Image.open(os.path.join(out2, "frame_%03d.png" % t))) for t in range(nt)])
    assert np.abs(got.astype(int) - frames.astype(int)).max() <= 3
    udir = str(tmp_path / "out1")
    decompress.run(mdir, one, udir, True, False)
    ref = np.stack([np.array(Image.open(os.path.join(udir, "frame_%03d.png" % t))) for t in range(nt)])
    np.testing.assert_array_equal(got, ref)


def test_sweep_under_torchrun_writes_the_files_of_one_process(tmp_path):
    """BASELINE configs[4] on N GPUs: `torch.distributed.run --nproc-per-node 2 -m tezip_amd.tezip -c ... --sweep 5 10 20 40`
    (gloo transport, both ranks on GPU 0 here) -- one candidate window size per rank and turn, the sizes all-gathered
    (sweep.sweep_sharded), the owner of the best candidate writes its files: the three files and sweep.txt must be
    byte-identical to the one-process sweep, and equal to what `-w <best>` writes (compress.py:249)."""
    import socket
    import subprocess
    import sys
    from tezip_amd import sweep
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    nt, h, w = 44, 24, 40
    frames = synth.translating_scene(nt, h, w, seed=25)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=8, bias_scale=0.1), 24, 40)
    ddir = _write(tmp_path, frames, False)
    one, two, direct = str(tmp_path / "one"), str(tmp_path / "two"), str(tmp_path / "direct")
    rows, bw = sweep.run(mdir, ddir, one, 0, [5, 10, 20, 40], "abs", [0.0], False, True)
    assert [r["window"] for r in rows] == [5, 10, 20, 40] and bw in (5, 10, 20, 40)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TEZIP_DIST_BACKEND="gloo", TEZIP_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "tezip_amd.tezip", "-c", mdir, ddir, two, "-p", "0", "--sweep", "5",
           "10", "20", "40", "-m", "abs", "-b", "0"]
    r = subprocess.run(cmd, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert sorted(os.listdir(two)) == ["entropy.dat", "filename.txt", "key_frame.dat", "sweep.txt", "tezip_amd.json"]
    for name in ("filename.txt", "key_frame.dat", "entropy.dat", "sweep.txt"):
        assert open(os.path.join(one, name), "rb").read() == open(os.path.join(two, name), "rb").read(), name
    assert open(os.path.join(two, "sweep.txt")).read().count("<- best") == 1
    compress.run(mdir, ddir, direct, 0, bw, None, "abs", [0.0], True, False, True)
    assert open(os.path.join(direct, "filename.txt"), "rb").read() == open(os.path.join(two, "filename.txt"), "rb").read()
    # compress.run streams its two frames through ZSTD_compressStream2, the sweep packs them in one shot: same content
    for name in ("key_frame.dat", "entropy.dat"):
        a = zstd.decompress(open(os.path.join(direct, name), "rb").read())
        assert a == zstd.decompress(open(os.path.join(two, name), "rb").read()), name


def test_opt_in_byte_shuffle_roundtrip_and_default_off(tmp_path):
    """--shuffle (not a reference feature): entropy.dat holds the payload as byte planes and says so
    in its trailer (first shape entry 2); this build decodes it; without the flag the stream is the
    reference's, byte for byte."""
    from PIL import Image
    from tezip_amd import _lib
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    nt, h, w = 9, 24, 40
    frames = synth.translating_scene(nt, h, w, seed=8)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=8, bias_scale=0.1), 24, 40)
    ddir = _write(tmp_path, frames, False)
    plain, shuf, out = str(tmp_path / "plain"), str(tmp_path / "shuf"), str(tmp_path / "out")
    compress.run(mdir, ddir, plain, 0, 4, None, "abs", [0.0], True, False, True)
    compress.run(mdir, ddir, shuf, 0, 4, None, "abs", [0.0], True, False, True, SHUFFLE=True)
    a = np.frombuffer(zstd.decompress(open(os.path.join(plain, "entropy.dat"), "rb").read()), "<i2")
    b = np.frombuffer(zstd.decompress(open(os.path.join(shuf, "entropy.dat"), "rb").read()), "<i2")
    n = nt * h * w * 3
    assert a.size == b.size and a[-6] == 1 and b[-6] == 2 and (a[n:-6] == b[n:-6]).all() and (a[-5:] == b[-5:]).all()
    planes = b[:n].view(np.uint8)
    assert (planes[:n] == (a[:n].astype(np.uint16) & 0xFF)).all() and (planes[n:] == (a[:n].astype(np.uint16) >> 8)).all()
    ctx = _lib.Context(0)
    assert (ctx.byte_unshuffle(planes) == a[:n]).all() and (ctx.byte_shuffle(np.ascontiguousarray(a[:n])) == planes).all()
    ctx.close()
    decompress.run(mdir, shuf, out, True, False)
    got = np.stack([np.array(Image.open(os.path.join(out, "frame_%03d.png" % t))) for t in range(nt)])
    assert np.array_equal(got, frames)


def test_host_memory_does_not_grow_with_the_number_of_frames(tmp_path):
    """the reference holds every frame and the whole int16 stream in RAM
    (compress.py:116-122, 329-333, 375-400); here the frames stream through a ring of window
    buffers into HBM and both output files are compressed from pieces.  Peak RSS of a whole
    `tezip.py -c` process must not depend on nt (4x the frames = 151 MB more frames + payload if it
    were held), and the longer run still round-trips."""
    import subprocess
    import sys
    from PIL import Image
    from conftest import ROOT
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    h = w = 256
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=8, bias_scale=0.1), h, w)
    frames = synth.turbulence(256, h, w, seed=12)
    peak = {}
    for nt in (64, 256):
        d = tmp_path / ("data%d" % nt)
        d.mkdir()
        for t in range(nt):
            Image.fromarray(frames[t]).save(d / ("frame_%03d.png" % t), compress_level=1)
        code = ("import resource, sys; sys.path.insert(0, %r); from tezip_amd import tezip;"
                "tezip.main(tezip.build_parser().parse_args(['-c', %r, %r, %r, '-p', '0', '-w', '16', '-m', 'abs', '-b', '0']));"
                "print('MAXRSS_KB', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)"
                % (ROOT, mdir, str(d), str(tmp_path / ("comp%d" % nt))))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        peak[nt] = int([l for l in r.stdout.splitlines() if l.startswith("MAXRSS_KB")][0].split()[1])
    held = (256 - 64) * h * w * 3 * 3 / 1024          # frames + int16 payload the reference would hold, KiB
    assert peak[256] - peak[64] < 0.5 * held, peak
    decompress.run(mdir, str(tmp_path / "comp256"), str(tmp_path / "out"), True, False)
    got = np.stack([np.array(Image.open(os.path.join(str(tmp_path / "out"), "frame_%03d.png" % t))) for t in (0, 100, 255)])
    assert np.array_equal(got, frames[[0, 100, 255]])


def test_streaming_decompress_rejects_damaged_files(tmp_path):
    """decompress.run reads both files piece by piece straight into HBM; a truncated zstd frame, an
    entropy.dat whose trailer does not fit its payload and a key_frame.dat of another size must raise
    (the reference fails at its reshapes, decompress.py:115,240) -- before any kernel sees them."""
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    nt, h, w = 9, 24, 40
    frames = synth.translating_scene(nt, h, w, seed=8)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=8, bias_scale=0.1), 24, 40)
    ddir = _write(tmp_path, frames, False)
    good = str(tmp_path / "good")
    compress.run(mdir, ddir, good, 0, 4, None, "abs", [0.0], True, False, True)
    names = ("entropy.dat", "filename.txt", "key_frame.dat")

    def variant(name, **repl):
        d = tmp_path / name
        d.mkdir()
        for n in names:
            data = open(os.path.join(good, n), "rb").read()
            (d / n).write_bytes(repl.get(n, data))
        return str(d)

    ent = open(os.path.join(good, "entropy.dat"), "rb").read()
    key = open(os.path.join(good, "key_frame.dat"), "rb").read()
    stream = np.frombuffer(zstd.decompress(ent), "<i2")
    cases = {
        "cut_zstd": {"entropy.dat": ent[: len(ent) // 2]},
        "short_payload": {"entropy.dat": zstd.compress_array(np.ascontiguousarray(stream[40:]), 9)},
        "one_channel": {"entropy.dat": zstd.compress_array(np.concatenate([stream[:-2], [1, stream[-1]]]).astype("<i2"), 9)},
        "other_key_size": {"key_frame.dat": zstd.compress_array(np.zeros(nt * h * w * 3 - 3, np.uint8), 9)},
        "cut_key": {"key_frame.dat": key[: len(key) // 2]},
    }
    for name, repl in cases.items():
        with pytest.raises((ValueError, RuntimeError)):
            decompress.run(mdir, variant(name, **repl), str(tmp_path / ("out_" + name)), True, False)
    decompress.run(mdir, good, str(tmp_path / "out_good"), True, False)   # the context is still usable afterwards


def test_early_rollout_from_the_sidecar_gives_the_same_images_and_survives_damage(tmp_path, monkeypatch):
    """Round 6: with `stack` in tezip_amd.json the streaming decoder stages the key frames and queues its rollout WHILE
    entropy.dat is being decompressed on a worker thread (the reference keeps the shape in the last values of that file,
    compress.py:390-394).  Same images as the late path (TEZIP_NO_EARLY_ROLLOUT=1, and a directory without the sidecar);
    a sidecar whose stack contradicts the trailer, a truncated entropy.dat and a truncated key_frame.dat raise -- the
    worker thread is gone afterwards and the next decode works."""
    import json
    import shutil
    import threading
    from PIL import Image
    from tezip_amd import sidecar
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    nt, h, w, p = 11, 24, 40, 2
    frames = synth.translating_scene(nt, h, w, seed=12)
    mdir = str(tmp_path / "model")
    weights.save_model(mdir, cfg, cfg.init_weights(seed=12, bias_scale=0.1), 24, 40)
    ddir = _write(tmp_path, frames, False)
    good = str(tmp_path / "good")
    compress.run(mdir, ddir, good, p, 4, None, "abs", [0.0], True, False, True)
    doc = json.load(open(os.path.join(good, sidecar.NAME)))
    assert doc["stack"] == [nt, h, w, p]

    def decode(src, out, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            decompress.run(mdir, src, str(tmp_path / out), True, False)
        finally:
            for k in env:
                monkeypatch.delenv(k)
        return np.stack([np.array(Image.open(os.path.join(str(tmp_path / out), "frame_%03d.png" % t))) for t in range(nt)])

    early = decode(good, "out_early")
    np.testing.assert_array_equal(early, frames)
    np.testing.assert_array_equal(decode(good, "out_late", TEZIP_NO_EARLY_ROLLOUT="1"), frames)
    bare = tmp_path / "bare"
    shutil.copytree(good, bare)
    os.remove(bare / sidecar.NAME)
    np.testing.assert_array_equal(decode(str(bare), "out_bare"), frames)

    def variant(name, stack=None, **repl):
        d = tmp_path / name
        shutil.copytree(good, d)
        for n, data in repl.items():
            (d / n).write_bytes(data)
        if stack is not None:
            dd = dict(doc, stack=stack)
            (d / sidecar.NAME).write_text(json.dumps(dd))
        return str(d)

    ent = open(os.path.join(good, "entropy.dat"), "rb").read()
    key = open(os.path.join(good, "key_frame.dat"), "rb").read()
    with pytest.raises(ValueError, match="tezip_amd.json describes"):      # same sizes, another warm-up: the trailer decides
        decode(variant("wrong_warm_up", stack=[nt, h, w, p - 1]), "o1")
    # a hint that contradicts the model's frame size (height and width swapped: the same number of key bytes) is dropped
    np.testing.assert_array_equal(decode(variant("swapped", stack=[nt, w, h, p]), "o2"), frames)
    # a stack that does not even fit key_frame.dat's size is not used at all: the late path decodes
    np.testing.assert_array_equal(decode(variant("other_size", stack=[nt + 1, h, w, p]), "o3"), frames)
    with pytest.raises((ValueError, RuntimeError)):
        decode(variant("cut_entropy", **{"entropy.dat": ent[: len(ent) // 2]}), "o4")
    with pytest.raises((ValueError, RuntimeError)):
        decode(variant("cut_key", **{"key_frame.dat": key[: len(key) // 2]}), "o5")
    assert not [t for t in threading.enumerate() if t is not threading.current_thread() and t.daemon and "Prefetch" in repr(t._target)]
    np.testing.assert_array_equal(decode(good, "out_again"), frames)
