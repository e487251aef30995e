"""Frame-window sharding across the GPUs of one node (one process per GPU).

Windows (a key frame + its predicted frames) are independent through prediction, delta and
quantisation (compress.py:218-220,256-263): each rank runs them on a contiguous range of
windows with NO data-path collective.  Only three things cross shard boundaries
(SURVEY.md §8e), exchanged through torch.distributed (RCCL on GPUs, gloo in CPU tests):
  1. the spatial delta runs over the whole flattened stack (compress.py:339): the FIRST element of a
     shard needs the last delta element of the previous shard (one int16 `carry`; it travels in one
     small all_gather together with the shard's key mask and an error flag);
  2. the rank table is built from the global histogram (compress.py:354-361): all-reduce of
     2111 counters (+ one failure counter), the table is then rebuilt identically on every rank;
  3. rank 0 receives the payload shards point to point, straight into their place in the
     full payload buffer (peers `send`, rank 0 `irecv`s into slices: nothing lands on a rank
     that does not use it).
A rank runs the SAME fused kernels as a single GPU does (round 3): tz_rollout, then tz_encode_begin
(delta, quantiser, spatial delta without a carry, histogram -- the symbols stay in its HBM), the two
small collectives, then tz_encode_finish (patches the shard's first symbol for the carry, remaps
with the global table).  Until round 2 a rank went through the unfused stand-alone operators with a
host round trip for the carry.
The decoder shards the same way; its inverse scan needs the prefix of per-shard sums.

Failure handling: every compute stage that sits in front of a collective reports its outcome THROUGH
that collective (flag in the all_gather, failure counter in the all-reduce, one more one-element
all-reduce in front of the point-to-point gather), so a rank that fails -- out of memory, a TezipError
-- raises its own error while every other rank raises a RuntimeError naming it; nobody is left
waiting in a collective until the watchdog fires.

DWP (-t) discovers window boundaries sequentially and does not shard: replicas only.
"""
import numpy as np

NBINS = 2111


class HipEngine:
    """Per-rank compute on the MI355X through the C ABI (tezip_amd._lib.Context).  Buffers are
    torch CUDA tensors on the context's device; host data only enters as the frame shard and
    leaves as the few scalars the protocol exchanges."""

    def __init__(self, ctx, device=None):
        import torch
        self.ctx, self.torch = ctx, torch
        self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)

    def _new(self, n, dtype):
        return self.torch.empty(int(n), dtype=dtype, device=self.dev)

    def _shared(self):
        """Does the context launch on torch's current stream (then the two are ordered by the stream itself)?"""
        mine = self.ctx.stream_ptr()
        return bool(mine) and mine == self.torch.cuda.current_stream(self.dev).cuda_stream

    def _sync(self):
        # a context with its OWN stream (the CLI, bench.py): torch / RCCL consume the buffers afterwards, so wait
        if not self._shared():
            self.ctx.synchronize()

    def _sync_in(self, x):
        # ... and the other way round: a device tensor torch may still be writing is complete before the library reads it
        if not isinstance(x, np.ndarray) and getattr(x, "is_cuda", False) and not self._shared():
            self.torch.cuda.current_stream(self.dev).synchronize()
        return x

    def _buf(self, x):
        """numpy (host) and torch tensors are both accepted by the C ABI."""
        return np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x.contiguous()

    # encoder: the two phases of tz_encode around the exchange of carry + histogram
    def encode_begin(self, frames, warm_up, window, mode, bound, entropy):
        frames = self._sync_in(self._buf(frames))
        key, _ = self.ctx.rollout(frames, warm_up, window)
        hist, first, last = self.ctx.encode_begin(mode, bound, entropy)
        return key, hist, first, last

    def contract(self):
        """The arithmetic contract (1 = TZ-PA1, 2 = TZ-PA2) this rank's predictions were made under."""
        return int(self.ctx.rollout_contract())

    def encode_finish(self, carry, table):
        nt, h, w = self.ctx._shape
        y = self._new(nt * h * w * 3, self.torch.int16)
        self.ctx.encode_finish(carry, table, out=y)
        self._sync()
        return y

    def build_table(self, hist):
        return self.ctx.build_table(hist)

    def last(self, buf):
        return int(buf[-1].item())

    # decoder
    def decode_prepare(self, key_frames, warm_up):
        return self.ctx.rollout_decode(self._sync_in(self._buf(key_frames)), warm_up)

    def unmap(self, payload, table):
        payload = self._sync_in(self._buf(payload))
        n = payload.size if isinstance(payload, np.ndarray) else payload.numel()
        out = self._new(n, self.torch.int16)
        self.ctx.unmap(payload, table, offset=True, out=out)
        self._sync()
        return out

    def upload(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)

    def undelta(self, sd, carry):
        out = self._new(sd.numel(), self.torch.int16)
        self.ctx.spatial_undelta(self._sync_in(sd), carry=carry, out=out)
        self._sync()
        return out

    def reconstruct(self, delta):
        nt, h, w = self.ctx._shape
        out = self._new(nt * h * w * 3, self.torch.uint8)
        self.ctx.decode_delta(self._sync_in(delta), out=out)
        self._sync()
        return out

    # communication views
    def comm_tensor(self, buf, dev):
        t = buf.reshape(-1).view(self.torch.uint8)
        return t if dev.type == "cuda" else t.cpu()

    def host(self, buf):
        return buf.cpu().numpy() if hasattr(buf, "cpu") else np.asarray(buf)


def plan_shards(nt, warm_up, window, world):
    """Contiguous frame ranges [f0, f1) per rank, cut at SWP window starts warm_up + k*window
    (frames [0, warm_up) stay with rank 0).  Every non-empty shard has >= 2 frames (rank 0:
    >= warm_up + 2, what tz_rollout needs); ranks beyond the number of usable shards get empty
    ones (f0 == f1)."""
    if window is None or window < 1:
        raise ValueError("only SWP (-w) shards; DWP (-t) finds its windows sequentially")
    if nt < warm_up + 2:
        raise ValueError("need at least warm_up+2 frames (nt=%d, warm_up=%d)" % (nt, warm_up))
    # candidate cut points: window starts that leave >= 2 frames on both sides of the cut
    # (window == 1 makes one-frame windows: they are grouped)
    starts = [s for s in range(warm_up + window, nt, window) if s >= warm_up + 2 and nt - s >= 2]
    kept, prev = [], 0
    for s in starts:
        if s - prev >= (warm_up + 2 if prev == 0 else 2):
            kept.append(s)
            prev = s
    g = len(kept) + 1
    used = max(1, min(world, g))
    bounds = [0] + kept
    cuts = [bounds[(i * g) // used] for i in range(used)] + [nt]
    shards = [(cuts[i], cuts[i + 1]) for i in range(used)]
    shards += [(nt, nt)] * (world - used)
    return shards


def _dist():
    import torch.distributed as dist
    return dist


def active():
    """(rank, world) when this process is part of an initialised torch.distributed job with
    more than one rank, else None."""
    try:
        import torch.distributed as dist
    except Exception:
        return None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return None


def init_from_env():
    """Join the job described by torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_*), one process per GPU.  Backend: RCCL ("nccl"), or TEZIP_DIST_BACKEND (e.g. "gloo"
    to rehearse several ranks on one GPU).  Returns the local device index."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TEZIP_SINGLE_DEVICE"):
        local = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            backend = os.environ.get("TEZIP_DIST_BACKEND", "nccl")
            if backend == "nccl":
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
    return local


def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_gather_i64(vals, dist):
    import torch
    dev = _device(dist)
    t = torch.tensor(vals, dtype=torch.int64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.cpu().numpy() for o in out]


def _comm_tensor(engine, buf, dev):
    """uint8 view of an engine buffer on the communication device (neither NCCL nor gloo has
    int16 tensors)."""
    import torch
    if hasattr(engine, "comm_tensor"):
        return engine.comm_tensor(buf, dev)
    return torch.from_numpy(np.ascontiguousarray(buf).reshape(-1).view(np.uint8))


class PendingGather:
    """A point-to-point gather in flight: the requests plus every buffer they touch.  wait() returns
    the assembled uint8 tensor on `dst` (None elsewhere)."""

    def __init__(self, reqs, full, keep):
        self.reqs, self.full, self.keep = reqs, full, keep

    def wait(self):
        for q in self.reqs:
            q.wait()
        self.reqs, self.keep = [], None
        return self.full


def _gather_shards_begin(engine, buf, nbytes, dist, dst=0, self_p2p=False):
    """Point-to-point gather of the per-rank byte shards (sizes `nbytes`, known to everyone from
    the shard plan) into one buffer on `dst`: peers send, `dst` receives each shard straight into
    its slice.  Returns a PendingGather: the transfers run behind whatever the caller does next.
    self_p2p (diagnostic, bench.py's one-GPU `sharded_path` leg): `dst` moves its OWN shard through the same
    grouped isend / irecv instead of a device copy, so that the point-to-point machinery runs on a box with one GPU."""
    import torch
    rank, dev = dist.get_rank(), _device(dist)
    mine = _comm_tensor(engine, buf, dev) if nbytes[rank] else None
    if rank != dst:
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, dst)]) if nbytes[rank] else []
        return PendingGather(reqs, None, mine)
    offs = np.concatenate([[0], np.cumsum(nbytes)]).astype(np.int64)
    full = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
    # ONE group of receives: the shards arrive concurrently over their own xGMI links
    own_by_p2p = bool(self_p2p and nbytes[rank])
    ops = [dist.P2POp(dist.irecv, full[int(offs[r]): int(offs[r + 1])], r)
           for r in range(dist.get_world_size()) if (r != dst or own_by_p2p) and nbytes[r]]
    if own_by_p2p:
        ops.append(dist.P2POp(dist.isend, mine, dst))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if nbytes[rank] and not own_by_p2p:
        full[int(offs[rank]): int(offs[rank + 1])].copy_(mine)
    return PendingGather(reqs, full, mine)


def _gather_shards(engine, buf, nbytes, dist, dst=0):
    return _gather_shards_begin(engine, buf, nbytes, dist, dst).wait()


def _typed(full_u8, dtype, to_host):
    """uint8 gather result -> typed buffer: numpy when it lives on the host (or to_host), else a
    torch device tensor."""
    import torch
    if full_u8.device.type == "cpu":
        return full_u8.numpy().view(dtype)
    t = full_u8.view({np.dtype(np.int16): torch.int16, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)])
    return t.cpu().numpy() if to_host else t


def _raise_if_any_failed(oks, what):
    bad = [r for r, ok in enumerate(oks) if not ok]
    if bad:
        raise RuntimeError("%s failed on rank(s) %s (see that rank's log)" % (what, bad))


class PendingCompress:
    """compress_sharded(..., wait=False): everything but the arrival of the payload shards on rank 0 is
    done; wait() returns what compress_sharded returns."""

    def __init__(self, gather, rank, table, keys, to_host):
        self.gather, self.rank, self.table, self.keys, self.to_host = gather, rank, table, keys, to_host

    def wait(self):
        full = self.gather.wait()
        if self.rank != 0:
            return None
        return _typed(full, np.int16, self.to_host), self.table, self.keys


def _i16(v):
    v &= 0xFFFF
    return v - 65536 if v >= 32768 else v


def _all_ok(ok, dist, what):
    """One-element all-reduce of a failure count: every rank learns whether all of them got here."""
    import torch
    t = torch.tensor([0 if ok else 1], dtype=torch.int64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if int(t.item()):
        raise RuntimeError("%s failed on %d rank(s) (see their logs)" % (what, int(t.item())))


def compress_sharded(engine, frames, warm_up, window, mode, bound, entropy=True, nt=None, to_host=True, wait=True,
                     self_p2p=False):
    """Every rank passes the same arguments.  `frames` is the full (nt,H,W,3) stack (anything whose
    [f0:f1] slice yields frames) or, with `nt` given, a callable (f0, f1) -> this rank's frames
    (a rank then never sees the others' frames).  Returns (payload, table|None, key_mask) on rank
    0, None elsewhere; the payload is a numpy array (to_host) or stays in rank 0's HBM.  The result
    is byte-identical to a single-GPU tz_rollout + tz_encode.  wait=False returns a PendingCompress
    instead: the point-to-point gather is still in flight, so a caller that compresses one sequence
    after the other overlaps it with the next rollout (outputs are fresh buffers every call)."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    if nt is None:
        nt = frames.shape[0]
        fetch = lambda a, b: frames[a:b]  # noqa: E731
    else:
        fetch = frames
    shards = plan_shards(nt, warm_up, window, world)
    f0, f1 = shards[rank]
    klen = max(b - a for a, b in shards)
    ok, err, key, hist, first, last, fe, pa = 1, None, np.zeros(0, bool), None, 0, 0, 0, 0
    try:
        if f1 > f0:
            mine = fetch(f0, f1)
            fe = int(np.prod(mine.shape[1:]))
            key, hist, first, last = engine.encode_begin(mine, warm_up if rank == 0 else 0, window, mode, bound, entropy)
            if hasattr(engine, "contract"):
                pa = engine.contract()
    except Exception as e:  # the other ranks are about to enter a collective: tell them
        ok, err = 0, e
    head = [ok, int(f1 > f0 and ok), int(last), fe, pa]
    kpad = np.zeros(klen, np.int64)
    kpad[: len(key)] = np.asarray(key, np.int64)
    infos = _all_gather_i64(head + kpad.tolist(), dist)
    if err is not None:
        raise err
    _raise_if_any_failed([int(i[0]) for i in infos], "window-sharded encode")
    fe = max(int(i[3]) for i in infos)
    # one stream, one arithmetic contract: the decoder replays every window under the contract rank 0 stamps into
    # tezip_amd.json, so a rank whose TEZIP_PA differs would have written windows nobody can regenerate
    pas = sorted({int(i[4]) for i in infos if int(i[1]) and int(i[4])})
    if len(pas) > 1:
        raise RuntimeError("the ranks of this job predicted under different arithmetic contracts %s "
                           "(TEZIP_PA / --pa must agree on every rank)" % ["TZ-PA%d" % v for v in pas])
    carry = None
    for r in range(rank - 1, -1, -1):
        if infos[r][1]:
            carry = int(infos[r][2])
            break
    # stage 2: the global table.  The failure counter rides in the same all-reduce as the histogram.
    table, y = None, None
    try:
        h = np.zeros(NBINS + 1, np.int64)
        if entropy and f1 > f0:
            h[:NBINS] = np.asarray(hist).astype(np.int64)
            if carry is not None:  # the shard's first symbol moves: sd = carry - x[0] instead of x[0] (compress.py:73-77)
                old, new = 1600 - _i16(first), _i16(1600 - _i16(carry - first))
                if 0 <= old < NBINS:
                    h[old] -= 1
                if 0 <= new < NBINS:
                    h[new] += 1
    except Exception as e:
        err = e
        h = np.zeros(NBINS + 1, np.int64)
    h[NBINS] = 0 if err is None else 1
    ht = torch.from_numpy(h).to(_device(dist))
    dist.all_reduce(ht, op=dist.ReduceOp.SUM)
    h = ht.cpu().numpy()
    if err is not None:
        raise err
    if int(h[NBINS]):
        raise RuntimeError("window-sharded encode (histogram stage) failed on %d rank(s) (see their logs)" % int(h[NBINS]))
    # stage 3: remap with the global table; its outcome is agreed on before anyone posts a send / receive
    try:
        if entropy:
            table = engine.build_table(h[:NBINS].astype(np.uint64))
        if f1 > f0:
            y = engine.encode_finish(carry, table)
    except Exception as e:
        err = e
    try:
        _all_ok(err is None, dist, "window-sharded encode (remap stage)")
    except RuntimeError:
        if err is not None:
            raise err
        raise
    gather = _gather_shards_begin(engine, y, [(b - a) * fe * 2 for a, b in shards], dist, self_p2p=self_p2p)
    keys = np.concatenate([np.asarray(i[5: 5 + (b - a)]) for i, (a, b) in zip(infos, shards)]).astype(bool)
    pending = PendingCompress(gather, rank, table, keys, to_host)
    return pending.wait() if wait else pending


def engine_last(engine, buf):
    return engine.last(buf) if hasattr(engine, "last") else int(np.asarray(buf).reshape(-1)[-1])


def plan_decode_shards(keys, nt, warm_up, world):
    """Decoder shards: cut at key frames (decompress.py:123-129), every non-empty shard >= 2 frames."""
    starts = [k for k in keys if k >= warm_up]
    if not starts or starts[0] != warm_up:
        raise ValueError("key frames do not cover the sequence")
    kept, prev = [], 0
    for s in starts[1:]:
        if s - prev >= (warm_up + 2 if prev == 0 else 2) and nt - s >= 2:
            kept.append(s)
            prev = s
    g = len(kept) + 1
    used = max(1, min(world, g))
    bounds = [0] + kept
    cuts = [bounds[(i * g) // used] for i in range(used)] + [nt]
    return [(cuts[i], cuts[i + 1]) for i in range(used)] + [(nt, nt)] * (world - used)


def decompress_sharded(engine, key_frames, payload, table, warm_up, to_host=True, gather=True):
    """Sharded decode: every rank passes the same key-frame stack, payload and table; rank 0
    gets the (nt, H, W, 3) uint8 frames, other ranks None.
    gather=False: nothing is sent to rank 0 -- every rank gets (f0, f1, its own (f1-f0, H, W, 3) uint8 frames on the
    host), so that each rank can write the image files of its own windows (decompress.py:266-279 is one loop over all
    frames; PNG files are independent)."""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    nt, H, W, C = key_frames.shape
    fe = H * W * C
    if np.asarray(payload).size != nt * fe:  # decompress.py:240: the reference's reshape raises
        raise ValueError("payload holds %d elements, the key-frame stack implies %d" % (np.asarray(payload).size, nt * fe))
    # decompress.py:123-129: key frames are the frames with a non-zero sample
    keys = [i for i in range(nt) if np.asarray(key_frames[i]).any()]
    shards = plan_decode_shards(keys, nt, warm_up, world)
    f0, f1 = shards[rank]
    payload = np.asarray(payload, np.int16).reshape(-1)
    info, sd, delta, err = [1, 0, 0, 0], None, None, None
    try:
        if f1 > f0:
            engine.decode_prepare(np.asarray(key_frames[f0:f1]), warm_up if rank == 0 else 0)
            part = payload[f0 * fe: f1 * fe]
            if table is not None:
                sd = engine.unmap(part, table)
            else:
                sd = engine.upload(part) if hasattr(engine, "upload") else np.ascontiguousarray(part)
            if rank == 0:
                delta = engine.undelta(sd, None)
                info = [1, 1, engine_last(engine, delta), 0]   # decoded value at the end of shard 0
            else:
                probe = engine.undelta(sd, 0)                   # x = 0 - prefix sums  =>  last = -(shard sum)
                info = [1, 1, 0, (-engine_last(engine, probe)) & 0xFFFF]
    except Exception as e:
        info, err = [0, 0, 0, 0], e
    infos = _all_gather_i64(info, dist)
    if err is not None:
        raise err
    _raise_if_any_failed([int(i[0]) for i in infos], "window-sharded decode")
    frames = None
    try:
        if f1 > f0:
            if rank > 0:
                x_end = int(infos[0][2])
                for r in range(1, rank):
                    if infos[r][1]:
                        x_end = (x_end - int(infos[r][3])) & 0xFFFF
                carry = x_end - 65536 if x_end >= 32768 else x_end
                delta = engine.undelta(sd, carry)
            if isinstance(delta, np.ndarray):
                delta = delta.reshape(f1 - f0, H, W, C)
            frames = engine.reconstruct(delta)
    except Exception as e:
        err = e
    try:  # agreed on before anyone posts a send / receive
        _all_ok(err is None, dist, "window-sharded decode (reconstruct stage)")
    except RuntimeError:
        if err is not None:
            raise err
        raise
    if not gather:
        if f1 == f0:
            return f0, f1, np.zeros((0, H, W, C), np.uint8)
        local = engine.host(frames) if hasattr(engine, "host") else np.asarray(frames)
        return f0, f1, np.asarray(local, np.uint8).reshape(f1 - f0, H, W, C)
    full = _gather_shards(engine, frames, [(b - a) * fe for a, b in shards], dist)
    if rank != 0:
        return None
    out = _typed(full, np.uint8, to_host)
    return out.reshape(nt, H, W, C)
