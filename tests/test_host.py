"""Host-side mirror of the reference interface (no GPU needed): CLI validation cascade
(tezip.py:28-84), model directory parsing, trailer/stream layout (compress.py:381-394,
decompress.py:105-113) against the goldens, zstd frames, padding helpers."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O
from tezip_amd import compress, data_utils, decompress, tezip, weights, zstd
from tezip_amd.prednet import PredNetConfig

R = np.load(os.path.join(GOLDEN, "ref_runs.npz"))
H = np.load(os.path.join(GOLDEN, "ref_helpers.npz"))


def _cli(argv):
    buf = io.StringIO()
    with redirect_stdout(buf):
        args = tezip.build_parser().parse_args(argv + ["-f"])  # -f: no device probe, 'CPU MODE'
        try:
            tezip.main(args)
        except SystemExit:
            pass
    return buf.getvalue().splitlines()


def test_cli_validation_cascade_messages():
    assert _cli([])[1:] == ['ERROR', 'Please mode select!', 'learn or compress or uncompress.',
                            'Command to check the options is -h or --help']
    assert _cli(["-c", "m", "d", "o"])[1:] == ['compress mode', 'ERROR', 'Please specify the -p or --preprocess option!',
                                               'warm up num.']
    out = _cli(["-c", "m", "d", "o", "-p", "0"])
    assert out[2:4] == ['ERROR', 'Please specify the window size(-w or --window) or MSE threshold(-t or --threshold) option!']
    out = _cli(["-c", "m", "d", "o", "-p", "0", "-w", "5", "-t", "0.1", "-m", "abs", "-b", "0"])
    assert out[3] == 'Please select only one of window size(-w or --window) or MSE threshold(-t or --threshold)!'
    out = _cli(["-c", "m", "d", "o", "-p", "0", "-w", "5", "-m", "xyz", "-b", "0"])
    assert out[2:5] == ['xyz', 'ERROR', 'Please specify the -m or --mode correctly!']
    out = _cli(["-c", "m", "d", "o", "-p", "0", "-w", "5", "-m", "abs"])
    assert out[3:5] == ['ERROR', 'Please specify the -b or --bound option!']
    out = _cli(["-c", "m", "d", "o", "-p", "0", "-w", "5", "-m", "absrel", "-b", "1"])
    assert out[3] == 'ERROR' and out[5].startswith("If the -m or --mode is 'absrel'")
    out = _cli(["-c", "m", "d", "o", "-u", "m", "f", "d"])
    assert out[1:3] == ['ERROR', 'Please select only one of learn or compress or uncompress.']
    assert _cli([])[0] == 'CPU MODE'
    # flags: -n is store_false (default True), as tezip.py:99
    a = tezip.build_parser().parse_args(["-n"])
    assert a.no_entropy is False and tezip.build_parser().parse_args([]).no_entropy is True


def test_force_cpu_is_refused_not_emulated(tmp_path):
    out = _cli(["-c", str(tmp_path), str(tmp_path), str(tmp_path / "o"), "-p", "0", "-w", "5", "-m", "abs", "-b", "0"])
    assert any("MI355X only" in l for l in out)


def test_padding_helpers_match_reference():
    assert [data_utils.padding_size(int(v)) for v in H["pad_sizes_in"]] == H["pad_sizes_out"].tolist()
    assert data_utils.padding_shape(375, 1242) == (376, 1248)


def test_model_dir_roundtrip_and_keras_style_json(tmp_path):
    import json
    cfg = PredNetConfig()
    w = cfg.init_weights(seed=2, bias_scale=0.1)
    weights.save_model(str(tmp_path), cfg, w, 128, 160)
    # the reference's two files (train.py:109,114-117), nothing else
    assert sorted(os.listdir(tmp_path)) == ["prednet_model.json", "prednet_weights.hdf5"]
    cfg2, w2, shape = weights.load_model(str(tmp_path))
    assert cfg2.stack_sizes == (3, 48, 96, 192) and shape == (128, 160)
    assert all((a == b).all() for a, b in zip(w, w2))
    # model.to_json() schema of the training graph (train.py:62-71)
    mj = json.loads(open(tmp_path / "prednet_model.json").read())
    layers = mj["config"]["layers"]
    assert [l["class_name"] for l in layers] == ["InputLayer", "PredNet", "TimeDistributed", "Flatten", "Dense"]
    assert layers[0]["config"]["batch_input_shape"] == [None, 2, 128, 160, 3]      # compress.py:168
    assert layers[1]["config"]["output_mode"] == "error" and layers[1]["config"]["data_format"] == "channels_last"
    assert layers[2]["config"]["layer"]["class_name"] == "Dense" and mj["keras_version"] == "2.2.4"
    assert mj["config"]["output_layers"] == [["dense_2", 0, 0]] and layers[4]["inbound_nodes"] == [[["flatten_1", 0, 0, {}]]]
    # a stale converted copy must not shadow the reference's file
    other = cfg.init_weights(seed=99)
    np.savez(tmp_path / "prednet_weights.npz", **{"w%03d" % i: x for i, x in enumerate(other)})
    _, w3, _ = weights.load_model(str(tmp_path))
    assert all((a == b).all() for a, b in zip(w, w3))
    os.remove(tmp_path / "prednet_weights.hdf5")      # legacy directory: only the npz
    _, w4, _ = weights.load_model(str(tmp_path))
    assert all((a == b).all() for a, b in zip(other, w4))
    # activations the kernels do not implement are refused, not silently replaced (prednet.py:95-98)
    for key, val in (("LSTM_activation", "relu"), ("LSTM_inner_activation", "sigmoid"), ("A_activation", "tanh"),
                     ("error_activation", "linear"), ("extrap_start_time", 5)):
        bad = json.loads(weights.make_model_json(cfg, 128, 160))
        bad["config"]["layers"][1]["config"][key] = val
        with pytest.raises(NotImplementedError):
            weights.parse_model_json(json.dumps(bad))
    # a json shaped like Keras 2.2.4's model.to_json() with extra layers (train.py:63-70)
    js = ('{"class_name":"Model","config":{"layers":[{"class_name":"InputLayer","config":{"batch_input_shape":'
          '[null,2,64,64,3]}},{"class_name":"PredNet","config":{"stack_sizes":[3,48,96,192],"R_stack_sizes":[3,48,96,192],'
          '"A_filt_sizes":[3,3,3],"Ahat_filt_sizes":[3,3,3,3],"R_filt_sizes":[3,3,3,3],"pixel_max":1.0,'
          '"data_format":"channels_last","output_mode":"error"}},{"class_name":"TimeDistributed","config":{}}]}}')
    c3, s3 = weights.parse_model_json(js)
    assert c3.nb_layers == 4 and s3 == (64, 64)
    with pytest.raises(FileNotFoundError):
        weights.load_model(str(tmp_path / "missing"))


@pytest.mark.parametrize("name", [str(n) for n in R["run_names"]])
def test_stream_layout_matches_reference_files(name):
    pre = "run_%s_" % name
    ref = R[pre + "entropy"]
    payload, table, shape, warm = decompress.parse_stream(ref.astype('<i2').tobytes())
    p2, t2, s2, w2 = O.parse_stream(ref)
    assert shape == s2 and warm == w2 and (payload == p2).all()
    assert (table is None) == (t2 is None) and (table is None or (table == t2).all())
    rebuilt = compress.build_stream(payload, table, shape, warm)
    np.testing.assert_array_equal(rebuilt, ref)
    nt, h, w = shape[1:4]
    assert payload.size == nt * h * w * 3


def test_zstd_frames_are_standard_and_carry_content_size():
    data = np.arange(10000, dtype=np.int16)
    blob = zstd.compress_array(data, 9)
    assert blob[:4] == b"\x28\xb5\x2f\xfd"  # zstd magic
    assert zstd.decompress(blob) == data.tobytes()
    assert zstd.decompress(zstd.compress(b"", 9)) == b""


def test_zstd_job_parallel_frame_is_one_standard_frame():
    """compress.run compresses entropy.dat with libzstd's worker threads when the library has them:
    still one frame with its content size (what the reference's zstd.decompress needs,
    decompress.py:89,98); falls back to one thread otherwise."""
    rng = np.random.default_rng(0)
    data = (rng.integers(0, 9, 6_000_000) ** 2 % 23).astype(np.int16)  # 12 MB: above the threading threshold
    blob = zstd.compress_array(data, 9, threads=4)
    assert blob[:4] == b"\x28\xb5\x2f\xfd"
    assert zstd.decompress(blob) == data.tobytes()
    one = zstd.compress_array(data, 9)
    assert abs(len(blob) - len(one)) <= len(one) // 100
    assert zstd.default_threads() >= 1 or "TEZIP_ZSTD_THREADS" in os.environ


def test_load_images_rules(tmp_path):
    from PIL import Image
    d = tmp_path / "imgs"
    d.mkdir()
    rng = np.random.default_rng(0)
    for i in (2, 0, 1):
        Image.fromarray(rng.integers(0, 256, (9, 7), dtype=np.uint8), mode="L").save(d / ("f%d.png" % i))
    stack, files, is_rgb = compress.load_images(str(d))
    assert files == ["f0.png", "f1.png", "f2.png"] and not is_rgb
    assert stack.shape == (3, 9, 7, 3) and (stack[..., 0] == stack[..., 2]).all()  # compress.py:114
    buf = io.StringIO()
    with redirect_stdout(buf), pytest.raises(SystemExit):
        compress.load_images(str(tmp_path / "nothing"))
    assert "is an empty or non-existent directory" in buf.getvalue()


def test_reads_the_reference_hdf5_checkpoint_without_h5py():
    """tests/golden/keras_style_model was written by real h5py in the layout of Keras 2.2.4's
    ModelCheckpoint (train.py:109); the built-in reader must return exactly those arrays."""
    mdir = os.path.join(GOLDEN, "keras_style_model")
    cfg, w, shape = weights.load_model(mdir)
    assert cfg.stack_sizes == (3, 16) and shape == (16, 24)
    ref = PredNetConfig(stack_sizes=(3, 16)).init_weights(seed=77, bias_scale=0.1)
    assert len(w) == len(ref) == 22
    for a, b in zip(w, ref):
        assert a.dtype == np.float32 and a.shape == b.shape and (a == b).all()
    from tezip_amd import h5lite
    every = h5lite.H5File(os.path.join(mdir, weights.H5_NAME)).walk()
    assert "/model_weights/time_distributed_1/time_distributed_1/kernel:0" in every
    with pytest.raises(ValueError):
        h5lite.H5File(os.path.join(GOLDEN, "ref_runs.npz"))


def test_corrupt_or_truncated_streams_are_rejected_before_the_native_call():
    """The trailer comes from a file: a truncated payload, a one-channel trailer or a key stack
    of another size must raise a ValueError (the reference fails at its reshape,
    decompress.py:115,240) instead of handing a short buffer to the library."""
    payload = np.arange(2 * 4 * 5 * 3, dtype=np.int16)
    table = np.array([1600, 1599], np.int16)
    stream = compress.build_stream(payload, table, (1, 2, 4, 5, 3), 0)
    p, t, shape, warm = decompress.parse_stream(stream.tobytes())
    decompress.check_stream(shape, warm, p.size, 2 * 4 * 5 * 3)  # consistent: passes
    with pytest.raises(ValueError, match="truncated"):
        decompress.check_stream(shape, warm, p.size - 7, 2 * 4 * 5 * 3)
    with pytest.raises(ValueError, match="key_frame.dat"):
        decompress.check_stream(shape, warm, p.size, 2 * 4 * 5 * 3 - 1)
    with pytest.raises(ValueError, match="shape"):
        decompress.check_stream((1, 2, 4, 5, 1), warm, 2 * 4 * 5, 2 * 4 * 5)
    with pytest.raises(ValueError, match="shape"):
        decompress.check_stream((3, 2, 4, 5, 3), warm, p.size, p.size)
    with pytest.raises(ValueError, match="warm-up"):
        decompress.check_stream(shape, 2, p.size, p.size)
    with pytest.raises(ValueError, match="warm-up"):
        decompress.check_stream(shape, -1, p.size, p.size)
    # a stream cut in the middle of the payload: the trailer is then garbage or inconsistent
    cut = stream[: stream.size - 30].tobytes()
    with pytest.raises(ValueError):
        p2, t2, shape2, warm2 = decompress.parse_stream(cut)
        decompress.check_stream(shape2, warm2, p2.size, 2 * 4 * 5 * 3)


def test_hdf5_written_here_is_read_by_real_h5py(tmp_path):
    """The "and back" half of the model import/export row: prednet_weights.hdf5 written by
    tezip_amd/h5lite.py, opened by libhdf5 through h5py (conda interpreter of the build container)
    following Keras 2.2.4's load_weights traversal.  Skipped where that interpreter is missing."""
    import hashlib
    import subprocess
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py on this machine")
    cfg = PredNetConfig()
    w = cfg.init_weights(seed=31, bias_scale=0.2)
    weights.save_model(str(tmp_path), cfg, w, 64, 64)
    r = subprocess.run([py, os.path.join(GOLDEN, "check_h5_with_h5py.py"), str(tmp_path / "prednet_weights.hdf5")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "keras_version 2.2.4 backend tensorflow"
    assert lines[1] == "layers input_1,pred_net_1,time_distributed_1,flatten_1,dense_2"
    h = hashlib.sha1()
    for a in w:
        h.update(np.ascontiguousarray(a, np.float32).tobytes())
    got = dict((l.split()[0], l.split()[1:]) for l in lines[2:])
    assert got["pred_net_1"][:2] == ["46", h.hexdigest()]
    assert got["pred_net_1"][2:] == ["pred_net_1/layer_a_0/kernel:0", "pred_net_1/layer_o_3/bias:0"]  # prednet.py:212 order
    assert got["time_distributed_1"][0] == "2" and got["dense_2"][0] == "2"
    # and the built-in reader returns the same arrays from the same file
    _, back, _ = weights.load_model(str(tmp_path))
    assert all((a == b).all() for a, b in zip(w, back))


def test_frame_source_streams_in_order_and_rejects_what_the_reference_rejects(tmp_path):
    """compress.py:97-131 as a stream: windows of frames in sorted-name order through a ring of three
    buffers, grayscale expanded to RGB, and the reference's messages for empty directories, non-image
    files, other modes and mixed sizes."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    rng = np.random.default_rng(3)
    fr = rng.integers(0, 256, (11, 9, 14, 3)).astype(np.uint8)
    d = tmp_path / "rgb"
    d.mkdir()
    for t in range(11):
        Image.fromarray(fr[t]).save(d / ("f%02d.png" % (10 - t)))      # reversed names: order comes from sorted()
    src = compress.FrameSource(str(d))
    assert (src.nt, src.H, src.W, src.is_rgb) == (11, 9, 14, True) and src.files[0] == "f00.png"
    with ThreadPoolExecutor(4) as pool:
        seen = [(f0, v.copy()) for f0, v in src.chunks(4, pool)]
    assert [f0 for f0, _ in seen] == [0, 4, 8] and [v.shape[0] for _, v in seen] == [4, 4, 3]
    assert np.array_equal(np.concatenate([v for _, v in seen]), fr[::-1])
    g = tmp_path / "gray"
    g.mkdir()
    for t in range(3):
        Image.fromarray(fr[t, :, :, 0], mode="L").save(g / ("g%d.png" % t))
    gs = compress.FrameSource(str(g))
    with ThreadPoolExecutor(2) as pool:
        got = np.concatenate([v.copy() for _, v in gs.chunks(2, pool)])
    assert not gs.is_rgb and np.array_equal(got, np.repeat(fr[:3, :, :, :1], 3, axis=-1))   # compress.py:114

    def messages(fn):
        buf = io.StringIO()
        with redirect_stdout(buf):
            with pytest.raises(SystemExit):
                fn()
        return buf.getvalue()

    assert "is an empty or non-existent directory" in messages(lambda: compress.FrameSource(str(tmp_path / "none")))
    bad = tmp_path / "bad"
    bad.mkdir()
    Image.fromarray(fr[0]).save(bad / "a.png")
    (bad / "b.png").write_bytes(b"not an image")

    def consume(path):
        s = compress.FrameSource(str(path))
        with ThreadPoolExecutor(2) as pool:
            list(s.chunks(4, pool))
    assert "contains files or folders that are not images" in messages(lambda: consume(bad))
    mixed = tmp_path / "mixed"
    mixed.mkdir()
    Image.fromarray(fr[0]).save(mixed / "a.png")
    Image.fromarray(fr[1][:5]).save(mixed / "b.png")
    assert "contains files or folders that are not images" in messages(lambda: consume(mixed))
    rgba = tmp_path / "rgba"
    rgba.mkdir()
    Image.fromarray(np.dstack([fr[0], fr[0][..., :1]]), mode="RGBA").save(rgba / "a.png")
    assert "Only RGB and grayscale are supported" in messages(lambda: compress.FrameSource(str(rgba)))


def test_zstd_streaming_frames_are_what_the_reference_reads(tmp_path):
    """ZSTD_compressStream2 with a pledged size: ONE standard frame with the content size in its
    header (the reference's zstd.decompress needs it), identical content, decodable piece by piece."""
    a = (np.random.default_rng(1).integers(0, 30, 3_000_001)).astype(np.int16)
    p = tmp_path / "s.zst"
    with open(p, "wb") as f:
        sc = zstd.StreamCompressor(f, a.nbytes, 9, 4)
        for i in range(0, a.size, 700_000):
            sc.write(a[i:i + 700_000])
        n = sc.close()
    blob = open(p, "rb").read()
    assert n == len(blob) and blob[:4] == b"\x28\xb5\x2f\xfd" and zstd.content_size(blob[:32]) == a.nbytes
    assert zstd.decompress(blob) == a.tobytes()
    with open(p, "rb") as f:
        pieces = [(size, bytes(piece)) for size, piece in zstd.stream_decompress(f, piece_bytes=1 << 20, read_bytes=1 << 16)]
    assert all(s == a.nbytes for s, _ in pieces) and b"".join(x for _, x in pieces) == a.tobytes()
    with pytest.raises(RuntimeError):
        list(zstd.stream_decompress(io.BytesIO(blob[: len(blob) // 2])))
    with open(tmp_path / "bad.zst", "wb") as f:
        sc = zstd.StreamCompressor(f, 100, 9, 0)
        sc.write(np.zeros(10, np.uint8))
        with pytest.raises(RuntimeError):
            sc.close()   # fewer bytes than pledged: libzstd refuses to end the frame


def test_h5lite_writer_limits_and_dtypes(tmp_path):
    from tezip_amd import h5lite
    root = h5lite.Group({"note": b"x"})
    g = root.group("g")
    for i in range(2 * h5lite.LEAF_K + 1):
        g.dataset("d%02d" % i, np.zeros(1, np.float32))
    with pytest.raises(NotImplementedError):
        h5lite.write_file(str(tmp_path / "big.h5"), root)
    root = h5lite.Group()
    root.dataset("u8", np.arange(6, dtype=np.uint8).reshape(2, 3), {"tag": b"bytes\x00inside"})
    root.dataset("i4", np.array([-5, 7], np.int32))
    root.dataset("f8", np.array([1.5, -2.25]))
    root.dataset("s", np.array([b"ab", b"c"], dtype="S2"))
    root.dataset("empty", np.zeros((0, 4), np.float32))
    h5lite.write_file(str(tmp_path / "t.h5"), root)
    back = h5lite.H5File(str(tmp_path / "t.h5")).walk()
    assert back["/u8"].dtype == np.uint8 and back["/u8"].tolist() == [[0, 1, 2], [3, 4, 5]]
    assert back["/i4"].tolist() == [-5, 7] and back["/f8"].tolist() == [1.5, -2.25]
    assert back["/s"].tolist() == [b"ab", b"c"] and back["/empty"].shape == (0, 4)
    with pytest.raises(NotImplementedError):
        r2 = h5lite.Group()
        r2.dataset("c", np.zeros(2, np.complex64))
        h5lite.write_file(str(tmp_path / "c.h5"), r2)


def test_result_pool_recycles_blocks():
    """_lib._ResultPool: big results come from recycled host blocks (no page faults on the second call);
    a block returns when the last view of it is dropped, never while a view is alive."""
    from tezip_amd import _lib
    pool = _lib._ResultPool()
    a = pool.empty((3, 1 << 20), np.int16)
    assert a.shape == (3, 1 << 20) and a.dtype == np.int16 and not pool.free
    a[...] = 7
    view, addr = a[1], a.ctypes.data
    del a
    assert not pool.free and (view == 7).all()       # the view keeps the block out of the pool
    del view
    assert len(pool.free) == 1
    b = pool.empty(5 << 20, np.uint8)                # 5 MB fits the 6 MB block (at most twice the size asked for)
    assert b.ctypes.data == addr and not pool.free
    c = pool.empty(1 << 20, np.uint8)                # a second request while b is out: a new block
    assert c.ctypes.data != addr
    del b, c
    assert len(pool.free) == 2
    small = pool.empty(100, np.uint8)                # small results are ordinary arrays
    assert small.base is None and len(pool.free) == 2
    for _ in range(10):                              # the pool keeps a bounded number of blocks
        del small
        small = [pool.empty((1 << 20) * (k + 1), np.uint8) for k in range(8)]
    del small
    assert len(pool.free) <= pool.MAX_FREE
