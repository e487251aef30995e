"""decompress.run -- same signature, inputs and outputs as the reference's
/root/reference/src/decompress.py:39 `run(...)`; the rollout replay, inverse remap, inverse
spatial delta (a prefix scan on the GPU instead of the reference's Python loop) and the
reconstruction run in libtezip_hip.so."""
import os
import sys
import time

import numpy as np

from . import _lib, sidecar, zstd
from . import dist as tzdist
from .compress import SHUFFLE_MARK, make_context, open_model
from .data_utils import padding_shape


def parse_stream(data):
    """decompress.py:105-113, 203-221: -> (payload, table|None, shape5, warm_up)."""
    s = np.frombuffer(data, dtype='<i2')
    if s.size < 8:
        raise ValueError("entropy.dat is too short")
    warm_up = int(s[-1])
    shape = tuple(int(v) for v in s[-6:-1])
    tlen = int(s[-7])
    if tlen == -1:
        return s[:-7], None, shape, warm_up
    if tlen < 0 or tlen > s.size - 7:
        raise ValueError("corrupt table length %d" % tlen)
    return s[: -7 - tlen], s[-7 - tlen: -7], shape, warm_up


def check_stream(shape, warm_up, payload_len, key_len):
    """The trailer is data from a file: cross-check it before any pointer derived from it reaches
    the native library.  The reference fails with a ValueError at its reshapes
    (decompress.py:115,240) for the same inconsistencies."""
    one, nt, H, W, C = shape
    if one not in (1, SHUFFLE_MARK) or C != 3 or nt < 1 or H < 1 or W < 1:
        raise ValueError("entropy.dat: unsupported stack shape %r (expected (1, nt, H, W, 3))" % (tuple(shape),))
    n = nt * H * W * C
    if payload_len != n:
        raise ValueError("entropy.dat: payload holds %d elements, the trailer says %d (truncated or corrupt file)"
                         % (payload_len, n))
    if key_len != n:
        raise ValueError("key_frame.dat holds %d bytes, entropy.dat's trailer implies %d" % (key_len, n))
    if not 0 <= warm_up < nt:
        raise ValueError("entropy.dat: warm-up count %d outside [0, %d)" % (warm_up, nt))


TAIL_ELEMS = _lib.TZ_NBINS + 8  # the longest trailer: table (<= 2111 symbols) + T + shape(5) + warm_up


def adopt_contract(DATA_DIR, wts, VERBOSE):
    """The arithmetic contract this directory must be decoded under (tezip_amd/sidecar.py): the one tezip_amd.json
    records; without that file --pa / TEZIP_PA, else None = by frame size.  A contradiction (a --pa that disagrees with the
    sidecar, another model's weights, a damaged tezip_amd.json) ends the run with a message and EXIT STATUS 2: this error
    class has no counterpart in the reference, so its `print` + `exit()` habit (status 0) is not mirrored -- a launcher
    must not see success when no frame was written."""
    try:
        contract = sidecar.resolve(sidecar.read(DATA_DIR), wts)
    except sidecar.SidecarMismatch as e:
        print("ERROR:", e)
        sys.exit(2)
    if VERBOSE and contract:
        print("arithmetic contract: TZ-PA%d" % contract)
    return contract


class _Prefetch:
    """zstd.stream_decompress of one file on a worker thread (libzstd releases the GIL), its pieces copied into a ring of
    host buffers and handed over through a bounded queue: the caller -- the only thread that touches the context -- can
    queue the decoder's rollout first and collect the payload afterwards.  At most `depth` + 2 pieces exist at a time,
    whatever the length of the stream.  An error on the worker is raised where the caller iterates."""

    def __init__(self, path, piece_bytes=16 << 20, depth=8):
        import queue
        import threading
        self.path, self.piece_bytes = path, piece_bytes
        self.q = queue.Queue(maxsize=depth)
        self.nbuf = depth + 2     # `depth` queued + the one the caller holds + the one being filled
        self.stop = False
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _put(self, item):
        import queue
        while not self.stop:
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def _run(self):
        try:
            bufs = []
            with open(self.path, "rb") as f:
                for k, (size, piece) in enumerate(zstd.stream_decompress(f, piece_bytes=self.piece_bytes)):
                    if len(bufs) < self.nbuf:
                        bufs.append(np.empty(self.piece_bytes, np.uint8))
                    buf = bufs[k % self.nbuf]
                    buf[:piece.size] = piece
                    if not self._put((size, buf[:piece.size])):
                        return
            self._put(None)
        except BaseException as e:   # handed to the consumer
            self._put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        self.stop = True
        self.t.join(timeout=5.0)


def _run_streaming(DATA_DIR, OUTPUT_DIR, file_names, isRGB, cfg, wts, model_shape, VERBOSE, device, contract=None, stack=None):
    """decompress.py:87-279 with nothing of size nt*H*W on the host: entropy.dat is decompressed
    piece by piece straight into HBM (the trailer is read from the last piece), key_frame.dat
    likewise, the decoded frames come back window by window and are PNG-encoded on a thread pool
    while the next window is fetched.  Returns False when the stream needs the whole-array path
    (this build's opt-in byte-shuffled payload).
    `stack` = (nt, H, W) from tezip_amd.json (round 6), or None: the reference keeps the shape in the LAST values of
    entropy.dat (compress.py:390-394), so without it nothing can start before the whole payload is decompressed; with it
    the key frames are staged and the decoder's rollout is queued FIRST, entropy.dat being decompressed on a worker
    thread meanwhile, and the trailer is checked against it when it arrives."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from .compress import _Stages, io_threads
    stages = _Stages("decompress")
    paths = {n: os.path.join(DATA_DIR, n) for n in ("key_frame.dat", "entropy.dat")}
    for n in ("key_frame.dat", "entropy.dat"):
        if not os.path.exists(paths[n]):
            print("ERROR: No such file or directory:", paths[n])
            exit()
    ctx = _lib.Context(device)
    try:
        ctx.load_model(cfg, wts)
        if contract:
            ctx.set_contract(contract)
        stages.mark("context + model load")
        with open(paths["key_frame.dat"], "rb") as f:
            head = f.read(64)
        key_len = zstd.content_size(head)

        def checks(nt, H, W):
            hp, wp = padding_shape(H, W)
            if model_shape is not None and (model_shape[0] != hp or model_shape[1] != wp):
                print("ERROR:keyframe size and model size do not match.")
                print("model size: height ", model_shape[0] - 7, "～", model_shape[0], " width ", model_shape[1] - 7, "～", model_shape[1])
                print("key frame size: height ", H, " width ", W)
                exit()
            if len(file_names) != nt:
                print("ERROR：The lengths of filename.txt and images do not match.")
                print("filename.txt：", len(file_names))
                print("number of images", nt)
                exit()
            return hp, wp

        def stage_keys(nt, H, W, hp, wp):
            """key_frame.dat into HBM (decompress.py:97-103); returns frames per fetch window"""
            ctx.prepare(hp, wp, 64 if nt > 64 else max(1, nt))
            stages.mark("model prepare")
            fb = H * W * 3
            per = max(1, (16 << 20) // fb)
            ctx.frames_begin(nt, H, W)
            first = 0
            with open(paths["key_frame.dat"], "rb") as f:
                for _, piece in zstd.stream_decompress(f, piece_bytes=per * fb):
                    k = piece.size // fb
                    ctx.frames_put(first, piece[: k * fb].reshape(k, H, W, 3))
                    first += k
            if first != nt:
                raise ValueError("key_frame.dat: truncated stream")
            stages.mark("zstd-d key_frame.dat + stage to HBM", ctx)
            return per

        def rollout(warm_up):
            if VERBOSE:
                ctx.prof_enable(True)
            t0 = time.time()
            ctx.rollout_decode(None, warm_up)     # (key discovery, then the predictor launches are queued)
            if VERBOSE:
                print("predict:{0}".format(time.time() - t0) + "[sec]")

        with open(paths["entropy.dat"], "rb") as f:
            ent_size = zstd.content_size(f.read(64))
        if ent_size % 2 or ent_size < 16:
            raise ValueError("entropy.dat is too short")
        total = ent_size // 2
        # the stack is known up front only from this build's sidecar -- a HINT: it is used when it agrees with everything
        # that can be checked now (key_frame.dat's size, the model's frame size, filename.txt), and the trailer has the last
        # word; a hint that does not fit is dropped and the late path below reports whatever is really wrong
        early = None
        if stack is not None and key_len == stack[0] * stack[1] * stack[2] * 3 and len(file_names) == stack[0]:
            hp_e, wp_e = padding_shape(stack[1], stack[2])
            if model_shape is None or (model_shape[0] == hp_e and model_shape[1] == wp_e):
                early = stack
        pre = _Prefetch(paths["entropy.dat"])      # entropy.dat is being decompressed from here on
        try:
            per = None
            ctx.payload_begin(total)               # (in front of the rollout: the copy stream need not wait for it)
            if early is not None:
                nt, H, W, warm_early = early
                hp, wp = checks(nt, H, W)
                per = stage_keys(nt, H, W, hp, wp)
                rollout(warm_early)
                stages.mark("rollout (decoder) queued")
            tail = np.zeros(0, np.int16)
            off = 0
            for size, piece in pre:
                if size != ent_size or piece.size % 2 or off * 2 + piece.size > ent_size:
                    raise ValueError("entropy.dat: inconsistent stream")
                p16 = piece.view(np.int16)
                ctx.payload_put(off, p16)       # staged on the copy stream; the piece buffer is free on return
                off += p16.size
                tail = np.concatenate([tail, p16[-TAIL_ELEMS:]])[-TAIL_ELEMS:]
        finally:
            pre.close()
        if off != total:
            raise ValueError("entropy.dat: truncated stream")
        stages.mark("zstd-d entropy.dat + stage to HBM", ctx)
        warm_up, shape, tlen = int(tail[-1]), tuple(int(v) for v in tail[-6:-1]), int(tail[-7])
        if tlen < -1 or tlen > tail.size - 7:
            raise ValueError("corrupt table length %d" % tlen)
        table = None if tlen == -1 else np.ascontiguousarray(tail[tail.size - 7 - tlen: tail.size - 7])
        payload_len = total - 7 - max(tlen, 0)
        check_stream(shape, warm_up, payload_len, key_len)
        if shape[0] == SHUFFLE_MARK:
            return False
        _, nt, H, W, C = shape
        if early is not None and (nt, H, W, warm_up) != tuple(early):
            raise ValueError("tezip_amd.json describes the stack as %r (frames, height, width, warm-up), entropy.dat's trailer as %r"
                             % (tuple(early), (nt, H, W, warm_up)))
        if early is None:
            hp, wp = checks(nt, H, W)
            per = stage_keys(nt, H, W, hp, wp)
            rollout(warm_up)
        fb = H * W * C
        stages.mark("rollout (decoder)", ctx)
        ctx.decode(None, table, out="resident")
        stages.mark("decode tail (frames resident)", ctx)
        if VERBOSE:
            prof = ctx.prof_get()
            if table is not None:
                print("replacing_based_on_frequency:{0}".format(prof["lut_remap"][0] / 1e3) + "[sec]")
            print("finding_difference:{0}".format(prof["undelta_scan"][0] / 1e3) + "[sec]")
        print("save as RGB" if isRGB else "save as gray")
        ring = [np.empty((per, H, W, C), np.uint8) for _ in range(3)]
        busy = [[], [], []]

        def save(buf, j, name):
            # decompress.py:272-278: the grayscale save is overwritten by an unconditional RGB save
            Image.fromarray(buf[j]).save(os.path.join(OUTPUT_DIR, name))

        with ThreadPoolExecutor(max_workers=io_threads()) as pool:  # PIL's encoder releases the GIL
            for ci, f0 in enumerate(range(0, nt, per)):
                k = min(per, nt - f0)
                slot = ci % 3
                for ft in busy[slot]:
                    ft.result()           # the encoders of the window that used this buffer are done
                ctx.decoded_get(f0, k, out=ring[slot][:k])
                busy[slot] = [pool.submit(save, ring[slot], j, file_names[f0 + j]) for j in range(k)]
            for fs in busy:
                for ft in fs:
                    ft.result()
        stages.mark("frames fetch + PNG encode")
    finally:
        ctx.close()
    return True


def run(WEIGHTS_DIR, DATA_DIR, OUTPUT_DIR, GPU_FLAG, VERBOSE, device=0):
    if not GPU_FLAG:
        print("ERROR: this build runs the decompression path on an AMD MI355X only (no CPU path).")
        exit()
    job = tzdist.active()
    rank0 = job is None or job[0] == 0
    # every rank of a sharded job writes the images of its own windows: each makes the directory (on one node they race for
    # the same one, hence exist_ok; on node-local paths each node gets its share -- INTEGRATION.md section 3)
    os.makedirs(OUTPUT_DIR, exist_ok=True)
    isRGB = True
    try:
        with open(os.path.join(DATA_DIR, 'filename.txt'), 'r', encoding='UTF-8') as f:
            file_names = [s.strip() for s in f.readlines()]
    except FileNotFoundError:
        print("ERROR:No such file or directory:", os.path.join(DATA_DIR, 'filename.txt'))
        exit()
    # decompress.py:55: `isdigit` is not called there, so any 1-character first line is the flag
    if file_names and len(file_names[0]) == 1:
        isRGB = bool(int(file_names.pop(0)))

    cfg, wts, model_shape = open_model(WEIGHTS_DIR)
    contract = adopt_contract(DATA_DIR, wts, VERBOSE)
    if job is None and not os.environ.get("TEZIP_NO_STREAMING"):
        # (the sidecar passed adopt_contract: it is readable or absent)
        stack = None if os.environ.get("TEZIP_NO_EARLY_ROLLOUT") else sidecar.stack_of(sidecar.read(DATA_DIR))
        done = _run_streaming(DATA_DIR, OUTPUT_DIR, file_names, isRGB, cfg, wts, model_shape, VERBOSE, device, contract, stack)
        if done:
            return

    def read(name):
        try:
            with open(os.path.join(DATA_DIR, name), mode='rb') as f:
                return zstd.decompress(f.read())
        except FileNotFoundError:
            print("ERROR: No such file or directory:", os.path.join(DATA_DIR, name))
            exit()

    key_bytes = read("key_frame.dat")
    payload, table, shape, warm_up = parse_stream(read("entropy.dat"))
    check_stream(shape, warm_up, payload.size, len(key_bytes))
    _, nt, H, W, C = shape
    key_frames = np.frombuffer(key_bytes, dtype=np.uint8).reshape(nt, H, W, C)
    hp, wp = padding_shape(H, W)
    if model_shape is not None and (model_shape[0] != hp or model_shape[1] != wp):
        print("ERROR:keyframe size and model size do not match.")
        print("model size: height ", model_shape[0] - 7, "～", model_shape[0], " width ", model_shape[1] - 7, "～", model_shape[1])
        print("key frame size: height ", H, " width ", W)
        exit()
    if len(file_names) != nt:
        print("ERROR：The lengths of filename.txt and images do not match.")
        print("filename.txt：", len(file_names))
        print("number of images", nt)
        exit()

    if job:
        device = tzdist.init_from_env()
    ctx = make_context(cfg, wts, hp, wp, 64 if nt > 64 else max(1, nt), device)
    first = 0           # index of frames[0] in the sequence: a rank of a sharded job holds (and saves) its own windows only
    try:
        if contract:
            ctx.set_contract(contract)
        if shape[0] == SHUFFLE_MARK:  # this build's opt-in byte planes -> the int16 payload
            payload = ctx.byte_unshuffle(np.ascontiguousarray(payload).view(np.uint8))
        if job:
            # key intervals sharded over the ranks (tezip_amd/dist.py); no frame travels: each rank saves its own
            first, _, frames = tzdist.decompress_sharded(tzdist.HipEngine(ctx, device), key_frames, payload, table, warm_up,
                                                         gather=False)
        else:
            if VERBOSE:
                ctx.prof_enable(True)
            t0 = time.time()
            ctx.rollout_decode(np.ascontiguousarray(key_frames), warm_up)
            if VERBOSE:
                print("predict:{0}".format(time.time() - t0) + "[sec]")
            frames = ctx.decode(np.ascontiguousarray(payload), None if table is None else np.ascontiguousarray(table))
            if VERBOSE:
                prof = ctx.prof_get()
                if table is not None:
                    print("replacing_based_on_frequency:{0}".format(prof["lut_remap"][0] / 1e3) + "[sec]")
                print("finding_difference:{0}".format(prof["undelta_scan"][0] / 1e3) + "[sec]")
    finally:
        ctx.close()

    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from .compress import _log_io, io_threads
    if rank0:
        print("save as RGB" if isRGB else "save as gray")

    def save(j):
        # decompress.py:272-278: the grayscale save is overwritten by an unconditional RGB save
        Image.fromarray(frames[j]).save(os.path.join(OUTPUT_DIR, file_names[first + j]))

    if job:
        # the ranks meet once more so that none returns (and the launcher none reports success) before every file is
        # written -- or learns that a rank could not
        import torch.distributed as dist
        err = None
        try:
            with ThreadPoolExecutor(max_workers=io_threads()) as pool:
                list(pool.map(save, range(len(frames))))
        except Exception as e:
            err = e
        _log_io("decompress", job, [(first, first + len(frames))])
        try:
            tzdist._all_ok(err is None, dist, "writing the decoded images")
        except RuntimeError:
            if err is not None:
                raise err
            raise
        return
    with ThreadPoolExecutor(max_workers=io_threads()) as pool:  # PIL's encoder releases the GIL
        list(pool.map(save, range(nt)))
