"""compress.run -- same signature, files and messages as the reference's
/root/reference/src/compress.py:93 `run(...)`; everything between "uint8 frame stack" and
"int16 payload + table" runs in libtezip_hip.so on the MI355X (no CPU fallback).

Output directory:
  filename.txt   line 1 "1"|"0" (RGB|L source), then one basename per line  (compress.py:133-136)
  key_frame.dat  zstd-9 of uint8[nt*H*W*3], zero except key frames           (compress.py:271-278)
  entropy.dat    zstd-9 of int16: payload | table | T  (or | -1) | 1,nt,H,W,3 | warm_up
                                                                              (compress.py:381-400)
"""
import glob
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, sidecar, weights, zstd
from . import dist as tzdist
from .data_utils import padding_shape


def io_threads():
    """Threads for PNG decode/encode: the CPUs this process may use, at most 16."""
    try:
        n = len(os.sched_getaffinity(0))
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def load_images(data_dir):
    """compress.py:97-131: sorted(glob), RGB or L only (L is expanded to RGB), all one size."""
    from PIL import Image, UnidentifiedImageError
    file_paths = sorted(glob.glob(os.path.join(data_dir, '*')))
    if len(file_paths) == 0:
        print("ERROR:", data_dir, "is an empty or non-existent directory")
        exit()
    try:
        first = Image.open(file_paths[0])
        image_mode = first.mode
        if all([image_mode != 'RGB', image_mode != 'L']):
            print("ERROR: input image is {0}. Only RGB and grayscale are supported.".format(image_mode))
            exit()
        is_rgb = image_mode == 'RGB'

        def decode(path):
            img = Image.open(path)
            arr = np.array(img if is_rgb else img.convert('RGB'))
            if arr.ndim != 3 or arr.shape[2] != 3:
                raise IndexError(path)
            return arr

        # PIL's codecs release the GIL: decode on a few threads (order is kept by map)
        with ThreadPoolExecutor(max_workers=io_threads()) as pool:
            frames = list(pool.map(decode, file_paths))
        files = [os.path.basename(path) for path in file_paths]
        stack = np.ascontiguousarray(np.stack(frames), dtype=np.uint8)
    except (PermissionError, IndexError, UnidentifiedImageError, IsADirectoryError, ValueError):
        print(data_dir, "contains files or folders that are not images.")
        exit()
    return stack, files, is_rgb


def open_model(weights_dir):
    """compress.py:143-160: same error messages."""
    json_file = os.path.join(weights_dir, weights.JSON_NAME)
    try:
        return weights.load_model(weights_dir)
    except FileNotFoundError:
        print("ERROR: No such file or directory:", json_file)
        exit()
    except OSError as e:
        print("ERROR: No such file or directory:", os.path.join(weights_dir, weights.H5_NAME))
        print(e)
        exit()


def make_context(cfg, wts, hp, wp, max_batch, device=0):
    ctx = _lib.Context(device)
    ctx.load_model(cfg, wts)
    ctx.prepare(hp, wp, max_batch)
    return ctx


SHUFFLE_MARK = 2  # first trailer shape entry of a byte-shuffled stream (the reference always writes 1)


def build_stream(payload, table, shape5, warm_up):
    """compress.py:381-394: payload | table | len(table)  (or | -1) | shape | PREPROCESS, int16.
    shape5[0] is 1 in the reference's format; SHUFFLE_MARK flags the opt-in byte-shuffled payload
    (this build only; `payload` then carries the two byte planes in an int16-typed buffer)."""
    if table is not None:
        tail = np.concatenate([table.astype(np.int64), [len(table)]])
    else:
        tail = np.array([-1], dtype=np.int64)
    trailer = np.concatenate([tail, list(shape5), [warm_up]]).astype(np.int16)
    return np.concatenate([np.asarray(payload, dtype=np.int16).reshape(-1), trailer])


def pack_outputs(frames, key, payload, table, warm_up, shuffled=False):
    """compress.py:271-278 and 375-400: the two zstd-9 frames (key_frame.dat, entropy.dat) as bytes."""
    nt, H, W = frames.shape[:3]
    key_frame = np.zeros_like(frames)
    key_frame[key] = frames[key]
    key_bytes = zstd.compress_array(key_frame, 9, zstd.default_threads())
    stream = build_stream(payload, table, (SHUFFLE_MARK if shuffled else 1, nt, H, W, 3), warm_up)
    return key_bytes, zstd.compress_array(stream, 9, zstd.default_threads())


def write_outputs(out_dir, frames, key, payload, table, warm_up, shuffled=False):
    key_bytes, entropy_bytes = pack_outputs(frames, key, payload, table, warm_up, shuffled)
    with open(os.path.join(out_dir, "key_frame.dat"), mode='wb') as f:
        f.write(key_bytes)
    with open(os.path.join(out_dir, "entropy.dat"), mode='wb') as f:
        f.write(entropy_bytes)
    return len(key_bytes), len(entropy_bytes)


class FrameSource:
    """compress.py:97-131 as a stream: the sorted image files of a directory decoded on a thread pool
    into a small ring of window-sized buffers (the reference appends every image to one growing
    array, compress.py:116-122).  Same acceptance rules and messages as load_images."""

    def __init__(self, data_dir):
        from PIL import Image, UnidentifiedImageError
        self.Image, self.errors = Image, (PermissionError, IndexError, UnidentifiedImageError, IsADirectoryError, ValueError)
        self.data_dir = data_dir
        self.paths = sorted(glob.glob(os.path.join(data_dir, '*')))
        if len(self.paths) == 0:
            print("ERROR:", data_dir, "is an empty or non-existent directory")
            exit()
        try:
            first = Image.open(self.paths[0])
            mode = first.mode
            if all([mode != 'RGB', mode != 'L']):
                print("ERROR: input image is {0}. Only RGB and grayscale are supported.".format(mode))
                exit()
            self.is_rgb = mode == 'RGB'
            self.W, self.H = first.size
        except self.errors:
            self.fail()
        self.nt = len(self.paths)
        self.files = [os.path.basename(p) for p in self.paths]

    def fail(self):
        print(self.data_dir, "contains files or folders that are not images.")
        exit()

    def _decode_into(self, dst, path):
        img = self.Image.open(path)
        arr = np.asarray(img if self.is_rgb else img.convert('RGB'))
        if arr.shape != dst.shape:  # another size or mode: np.array of such a list is not a stack (compress.py:122)
            raise ValueError(path)
        dst[...] = arr

    def chunks(self, per_chunk, pool, ring=3):
        """Yields (first frame index, uint8 (k,H,W,3) view) in order; a yielded buffer is reused
        `ring` chunks later, so the consumer must be done with it when it asks for the next one."""
        bufs = [np.empty((per_chunk, self.H, self.W, 3), np.uint8) for _ in range(ring)]
        starts = list(range(0, self.nt, per_chunk))

        def submit(ci):
            f0 = starts[ci]
            n = min(per_chunk, self.nt - f0)
            buf = bufs[ci % ring]
            return [pool.submit(self._decode_into, buf[j], self.paths[f0 + j]) for j in range(n)], buf[:n], f0

        pending = [submit(ci) for ci in range(min(ring - 1, len(starts)))]
        nxt = len(pending)
        while pending:
            futs, view, f0 = pending.pop(0)
            try:
                for ft in futs:
                    ft.result()
            except self.errors:
                self.fail()
            if nxt < len(starts):  # its buffer was handed out `ring` chunks ago
                pending.append(submit(nxt))
                nxt += 1
            yield f0, view


PAYLOAD_CHUNK = 8 << 20  # int16 elements fetched and fed to zstd at a time
KEY_PREFETCH_BYTES = 256 << 20  # key frames held on the host at once (more than that: streamed one by one)


STAGE_LOG = None   # a list installed by a caller (bench.py's host_pipeline leg): every run appends (run, stage, seconds)


class _Stages:
    """Stage wall times of compress.run / decompress.run: on stderr when TEZIP_TIMING is set (scripts/host_pipeline.py),
    into compress.STAGE_LOG when a caller installed a list there.  `add` records a duration measured elsewhere (a worker
    thread's: such a stage overlaps the ones marked around it)."""

    def __init__(self, run="compress"):
        self.on = bool(os.environ.get("TEZIP_TIMING"))
        self.run = run
        self.t0 = self.last = time.perf_counter()

    def add(self, name, seconds):
        if STAGE_LOG is not None:
            STAGE_LOG.append((self.run, name, float(seconds)))
        if self.on:
            import sys
            print("[tezip timing] %-34s %7.3f s  (overlapped)" % (name, seconds), file=sys.stderr)

    def mark(self, name, ctx=None):
        """ctx: the stage queued device work that may still be running -- when (and only when) stage times are wanted,
        wait for it, so that the time lands on the stage that queued it and not on the next one that synchronises."""
        if self.on or STAGE_LOG is not None:
            if ctx is not None:
                ctx.synchronize()
            now = time.perf_counter()
            if STAGE_LOG is not None:
                STAGE_LOG.append((self.run, name, now - self.last))
            if self.on:
                import sys
                print("[tezip timing] %-34s %7.3f s  (at %.3f s)" % (name, now - self.last, now - self.t0), file=sys.stderr)
            self.last = now


def _stream_outputs(ctx, out_dir, nt, H, W, key, table, warm_up, shuffled, pool, stages=None):
    """key_frame.dat and entropy.dat (compress.py:271-278, 375-400) from the context-resident frames
    and payload, piece by piece: nothing of size nt*H*W lives on the host."""
    n = nt * H * W * 3
    key_idx = [int(i) for i in np.nonzero(key)[0]]
    zero = np.zeros((H, W, 3), np.uint8)
    if len(key_idx) * H * W * 3 <= KEY_PREFETCH_BYTES:
        # the usual case, a few key frames: fetched up front, key_frame.dat is compressed by a worker while the
        # payload is fetched and compressed here (the context is not thread-safe: the worker never touches it)
        key_frames = {i: ctx.frames_get(i, 1)[0] for i in key_idx}

        def key_file():
            t0 = time.perf_counter()
            with open(os.path.join(out_dir, "key_frame.dat"), mode='wb') as f:
                sc = zstd.StreamCompressor(f, n, 9, max(1, zstd.default_threads() // 4))
                for i in range(nt):
                    sc.write(key_frames.get(i, zero))
                size = sc.close()
            if stages:
                stages.add("zstd-9 key_frame.dat (worker)", time.perf_counter() - t0)
            return size

        kf = pool.submit(key_file)
    else:
        # many key frames (-w 1, DWP with a tiny threshold: up to the whole stack): one at a time through
        # frames_get, written before entropy.dat -- host memory stays independent of the number of frames
        is_key = set(key_idx)
        with open(os.path.join(out_dir, "key_frame.dat"), mode='wb') as f:
            sc = zstd.StreamCompressor(f, n, 9, zstd.default_threads())
            for i in range(nt):
                sc.write(ctx.frames_get(i, 1)[0] if i in is_key else zero)
            ksize = sc.close()

        class _Done:
            def result(self):
                return ksize
        kf = _Done()
    if table is not None:
        tail = np.concatenate([table.astype(np.int64), [len(table)]])
    else:
        tail = np.array([-1], dtype=np.int64)
    trailer = np.concatenate([tail, [SHUFFLE_MARK if shuffled else 1, nt, H, W, 3], [warm_up]]).astype(np.int16)
    bufs = [np.empty(min(PAYLOAD_CHUNK, n), np.int16) for _ in range(2)]
    t_e = time.perf_counter()
    with open(os.path.join(out_dir, "entropy.dat"), mode='wb') as f:
        sc = zstd.StreamCompressor(f, n * 2 + trailer.nbytes, 9, zstd.default_threads())
        for k, off in enumerate(range(0, n, PAYLOAD_CHUNK)):
            cnt = min(PAYLOAD_CHUNK, n - off)
            piece = ctx.payload_get(off, cnt, out=bufs[k % 2][:cnt])
            sc.write(piece)
        sc.write(trailer)
        esize = sc.close()
    if stages:
        stages.add("payload fetch + zstd-9 entropy.dat", time.perf_counter() - t_e)
    return kf.result(), esize


def run(WEIGHTS_DIR, DATA_DIR, OUTPUT_DIR, PREPROCESS, WINDOW_SIZE, THRESHOLD, MODE, BOUND_VALUE, GPU_FLAG, VERBOSE,
        ENTROPY_RUN, device=0, SHUFFLE=False):
    """SHUFFLE (--shuffle; NOT in the reference): store the payload as byte planes.  Off by default:
    a shuffled entropy.dat is flagged in its trailer and is not readable by the reference.

    One process: the images stream through a ring of window buffers into HBM while the model loads,
    and key_frame.dat / entropy.dat are written from context-resident data in pieces, so host memory
    does not grow with the number of frames.  Under torch.distributed.run the windows are sharded
    over the ranks (tezip_amd/dist.py) from a stack that every rank loads."""
    if not GPU_FLAG:
        print("ERROR: this build runs the compression path on an AMD MI355X only (no CPU path).")
        exit()
    if tzdist.active() is not None:
        return _run_sharded(WEIGHTS_DIR, DATA_DIR, OUTPUT_DIR, PREPROCESS, WINDOW_SIZE, THRESHOLD, MODE, BOUND_VALUE,
                            VERBOSE, ENTROPY_RUN, device, SHUFFLE)
    if not os.path.exists(OUTPUT_DIR):
        os.mkdir(OUTPUT_DIR)
    stages = _Stages()
    src = FrameSource(DATA_DIR)
    nt, H, W = src.nt, src.H, src.W
    per_chunk = max(1, min(WINDOW_SIZE or 16, 64))
    with ThreadPoolExecutor(max_workers=io_threads() + 1) as pool:
        chunks = src.chunks(per_chunk, pool)   # decoding starts with the first next(); model + HIP start-up overlap it
        stages.mark("list + probe")
        head = next(chunks)
        stages.mark("first window decoded")
        cfg, wts, model_shape = open_model(WEIGHTS_DIR)
        stages.mark("model directory read")
        hp, wp = padding_shape(H, W)
        if model_shape is not None and (model_shape[0] != hp or model_shape[1] != wp):
            print("ERROR:Image size is out of scope for this model.")
            print("Compatible sizes for this model are height", model_shape[0] - 7, "to", model_shape[0], "and width",
                  model_shape[1] - 7, "to", model_shape[1])
            exit()
        if nt < PREPROCESS + 2:
            print("ERROR: need at least warm_up+2 images (%d given, warm_up %d)." % (nt, PREPROCESS))
            exit()
        if SHUFFLE and (nt * H * W * 3) % 8:
            print("ERROR: --shuffle needs nt*H*W*3 to be a multiple of 8 (%d x %d x %d x 3 is not)." % (nt, H, W))
            exit()
        # the job is accepted: only now does the output directory receive its first file (compress.py:133-136
        # writes it after the images are loaded; a rejected job must not leave a partial directory behind)
        with open(os.path.join(OUTPUT_DIR, 'filename.txt'), 'w', encoding='UTF-8') as f:
            f.write(f"{int(src.is_rgb)}\n")
            for file_name in src.files:
                f.write("%s\n" % file_name)
        nwin = 1 if WINDOW_SIZE is None else max(1, (nt - PREPROCESS + WINDOW_SIZE - 1) // WINDOW_SIZE)
        ctx = make_context(cfg, wts, hp, wp, min(nwin, 64), device)
        stages.mark("context + model prepare")
        try:
            ctx.frames_begin(nt, H, W)
            ctx.frames_put(*head)
            for f0, view in chunks:
                ctx.frames_put(f0, view)   # pageable ring buffer: free again when the call returns
            stages.mark("remaining windows decoded + staged", ctx)
            if VERBOSE:
                ctx.prof_enable(True)
            t0 = time.time()
            key, mse = ctx.rollout(None, PREPROCESS, WINDOW_SIZE, THRESHOLD, want_mse=bool(VERBOSE))
            if VERBOSE:
                for i in range(PREPROCESS + 1, nt):
                    print("MSE:", mse[i])
                    if key[i] and i > PREPROCESS:
                        print("move key point")
                print("predict:{0}".format(time.time() - t0) + "[sec]")
            stages.mark("rollout", ctx)
            _, table, _ = ctx.encode(MODE, BOUND_VALUE, ENTROPY_RUN, payload="resident", shuffle=SHUFFLE)
            stages.mark("encode (payload resident)", ctx)
            if VERBOSE:
                prof = ctx.prof_get()
                print("error_bound:{0}".format(prof["quant"][0] / 1e3) + "[sec]")
                print("finding_difference:{0}".format(prof["spatial_delta_hist"][0] / 1e3) + "[sec]")
                if ENTROPY_RUN:
                    # compress.py:351-365 times 1600 - x, bincount and the table sort; the first two are fused into
                    # the spatial-delta kernel here (timed above), what is left is the host-side sort
                    print("table_create:{0}".format(prof["table_create"][0] / 1e3) + "[sec]")
                    print("replacing_based_on_frequency:{0}".format(prof["lut_remap"][0] / 1e3) + "[sec]")
            _stream_outputs(ctx, OUTPUT_DIR, nt, H, W, key, table if ENTROPY_RUN else None, PREPROCESS, SHUFFLE, pool, stages)
            doc = sidecar.write(OUTPUT_DIR, ctx.rollout_contract(), wts, hp, wp, (nt, H, W, PREPROCESS))   # the contract the predictions were made under
            if VERBOSE:
                print("arithmetic contract:", doc["arithmetic_contract"])
            stages.mark("key_frame.dat + entropy.dat")
        finally:
            ctx.close()


class _NotImages(Exception):
    """A rank met a file that is not an image of the stack's size and mode (the reference's message, compress.py:124-131)."""


def _decode_list(src, indices, pool):
    """The frames `indices` of a FrameSource as one uint8 (len, H, W, 3) array, decoded on `pool`."""
    indices = list(indices)
    out = np.empty((len(indices), src.H, src.W, 3), np.uint8)
    futs = [pool.submit(src._decode_into, out[j], src.paths[i]) for j, i in enumerate(indices)]
    try:
        for ft in futs:
            ft.result()
    except src.errors:
        raise _NotImages(src.data_dir)
    return out


def pack_outputs_from_keys(nt, H, W, key, key_frames, payload, table, warm_up, shuffled=False):
    """pack_outputs for a caller that holds only the key frames: `key_frames` maps frame index -> uint8 (H, W, 3)."""
    stack = np.zeros((nt, H, W, 3), np.uint8)
    for i in np.nonzero(key)[0]:
        stack[i] = key_frames[int(i)]
    key_bytes = zstd.compress_array(stack, 9, zstd.default_threads())
    stream = build_stream(payload, table, (SHUFFLE_MARK if shuffled else 1, nt, H, W, 3), warm_up)
    return key_bytes, zstd.compress_array(stream, 9, zstd.default_threads())


def _run_sharded(WEIGHTS_DIR, DATA_DIR, OUTPUT_DIR, PREPROCESS, WINDOW_SIZE, THRESHOLD, MODE, BOUND_VALUE, VERBOSE,
                 ENTROPY_RUN, device, SHUFFLE):
    """Under torch.distributed.run.  SWP: the windows are sharded over the ranks and so is the image I/O -- every rank
    lists the directory (names only) and decodes the files of ITS frame range, nothing else (compress.py:97-122 decodes
    every file in one loop); rank 0 additionally reads the few key frames key_frame.dat needs and writes the files.
    DWP does not shard (its windows are found sequentially): rank 0 runs it alone."""
    job = tzdist.active()
    rank0 = job[0] == 0
    if rank0 and not os.path.exists(OUTPUT_DIR):
        os.mkdir(OUTPUT_DIR)
    src = FrameSource(DATA_DIR)          # lists, probes the first image (mode, size): same messages as load_images
    nt, H, W, files, isRGB = src.nt, src.H, src.W, src.files, src.is_rgb
    cfg, wts, model_shape = open_model(WEIGHTS_DIR)
    hp, wp = padding_shape(H, W)
    if model_shape is not None and (model_shape[0] != hp or model_shape[1] != wp):
        print("ERROR:Image size is out of scope for this model.")
        print("Compatible sizes for this model are height", model_shape[0] - 7, "to", model_shape[0], "and width",
              model_shape[1] - 7, "to", model_shape[1])
        exit()
    if nt < PREPROCESS + 2:
        print("ERROR: need at least warm_up+2 images (%d given, warm_up %d)." % (nt, PREPROCESS))
        exit()
    if SHUFFLE and (nt * H * W * 3) % 8:
        print("ERROR: --shuffle needs nt*H*W*3 to be a multiple of 8 (%d x %d x %d x 3 is not)." % (nt, H, W))
        exit()
    if rank0:
        with open(os.path.join(OUTPUT_DIR, 'filename.txt'), 'w', encoding='UTF-8') as f:
            f.write(f"{int(isRGB)}\n")
            for file_name in files:
                f.write("%s\n" % file_name)
    if WINDOW_SIZE is None:
        if rank0:
            print("NOTE: DWP (-t) finds its windows sequentially and does not shard: running on rank 0 only.")
        else:
            return
        job = None
    if job:
        device = tzdist.init_from_env()
        nt_local = max(b - a for a, b in tzdist.plan_shards(nt, PREPROCESS, WINDOW_SIZE, job[1]))
        nwin = max(1, (nt_local + WINDOW_SIZE - 1) // WINDOW_SIZE)
    else:
        nwin = 1
    ctx = make_context(cfg, wts, hp, wp, min(nwin, 64), device)
    decoded = []          # (f0, f1) ranges this rank decoded: tests read it through TEZIP_IO_LOG
    try:
        with ThreadPoolExecutor(max_workers=io_threads()) as pool:
            own = {}

            def fetch(f0, f1):
                decoded.append((f0, f1))
                own["f0"], own["frames"] = f0, _decode_list(src, range(f0, f1), pool)
                return own["frames"]

            try:
                if job:
                    res = tzdist.compress_sharded(tzdist.HipEngine(ctx, device), fetch, PREPROCESS, WINDOW_SIZE, MODE,
                                                  BOUND_VALUE, ENTROPY_RUN, nt=nt)
                else:
                    frames = fetch(0, nt)
                    key, _ = ctx.rollout(frames, PREPROCESS, None, THRESHOLD)
                    payload, table, _ = ctx.encode(MODE, BOUND_VALUE, ENTROPY_RUN, shuffle=SHUFFLE)
                    res = (payload, table, key)
            except _NotImages:
                src.fail()
            if res is None:
                return
            payload, table, key = res
            if job and SHUFFLE:
                payload = ctx.byte_shuffle(np.ascontiguousarray(payload)).view(np.int16)
            # key_frame.dat (compress.py:271-278): the key frames of rank 0's own range are in memory, the others -- one
            # per window -- are read from their files here
            f0 = own.get("f0", 0)
            have = own.get("frames")
            key_frames = {}
            missing = []
            for i in (int(v) for v in np.nonzero(key)[0]):
                if have is not None and f0 <= i < f0 + len(have):
                    key_frames[i] = have[i - f0]
                else:
                    missing.append(i)
            try:
                for i, arr in zip(missing, _decode_list(src, missing, pool)):
                    key_frames[i] = arr
                decoded.extend((i, i + 1) for i in missing)
            except _NotImages:
                src.fail()
        key_bytes, entropy_bytes = pack_outputs_from_keys(nt, H, W, key, key_frames, payload, table if ENTROPY_RUN else None,
                                                          PREPROCESS, SHUFFLE)
        with open(os.path.join(OUTPUT_DIR, "key_frame.dat"), mode='wb') as f:
            f.write(key_bytes)
        with open(os.path.join(OUTPUT_DIR, "entropy.dat"), mode='wb') as f:
            f.write(entropy_bytes)
        doc = sidecar.write(OUTPUT_DIR, ctx.get_contract(), wts, hp, wp, (nt, H, W, PREPROCESS))   # every rank runs under the same TEZIP_PA / frame size
        if VERBOSE:
            print("arithmetic contract:", doc["arithmetic_contract"])
    finally:
        _log_io("compress", job, decoded)
        ctx.close()


def _log_io(what, job, ranges):
    """TEZIP_IO_LOG=<dir>: every rank leaves `<what>.rank<r>` with the frame ranges it decoded / encoded (tests assert
    that a rank of a sharded job touches the files of its own windows only)."""
    d = os.environ.get("TEZIP_IO_LOG")
    if d:
        rank = job[0] if job else 0
        try:   # (a diagnostic: it must never be what a job fails on, least of all from a `finally`)
            with open(os.path.join(d, "%s.rank%d" % (what, rank)), "w") as f:
                f.write(" ".join("%d:%d" % r for r in ranges) + "\n")
        except OSError:
            pass
