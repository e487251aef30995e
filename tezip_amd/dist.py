"""Frame-window sharding across the GPUs of one node (one process per GPU).

Windows (a key frame + its predicted frames) are independent through prediction, delta and
quantisation (compress.py:218-220,256-263): each rank runs them on a contiguous range of
windows with NO data-path collective.  Only three tiny things cross shard boundaries
(SURVEY.md §8e), exchanged through torch.distributed (RCCL on GPUs, gloo in CPU tests):
  1. the spatial delta runs over the whole flattened stack (compress.py:339): a shard needs the
     last delta element of the previous shard (one int16 `carry`);
  2. the rank table is built from the global histogram (compress.py:354-361): all-reduce of
     2111 counters, the table is then rebuilt identically on every rank;
  3. rank 0 collects the payload shards, key masks (and decoded frames).
The decoder shards the same way; its inverse scan needs the prefix of per-shard sums.

DWP (-t) discovers window boundaries sequentially and does not shard: replicas only.

The per-rank compute is behind a small `engine` interface so that the protocol can be
exercised on CPU ranks (tests plug the oracle in); production uses HipEngine.
"""
import numpy as np

NBINS = 2111


class HipEngine:
    """Per-rank compute on the MI355X through the C ABI (tezip_amd._lib.Context)."""

    def __init__(self, ctx):
        self.ctx = ctx

    # encoder
    def encode_delta(self, frames, warm_up, window, mode, bound):
        key, _ = self.ctx.rollout(np.ascontiguousarray(frames), warm_up, window)
        return key, self.ctx.encode_delta(mode, bound).reshape(-1)

    def spatial_delta(self, d, carry, offset):
        hist = np.zeros(NBINS, np.uint64) if offset else None
        y = self.ctx.spatial_delta(np.ascontiguousarray(d), offset, carry=carry, hist=hist)
        return y, hist

    def build_table(self, hist):
        return self.ctx.build_table(hist)

    def remap(self, y, table):
        return self.ctx.remap(y, table)

    # decoder
    def decode_prepare(self, key_frames, warm_up):
        return self.ctx.rollout_decode(np.ascontiguousarray(key_frames), warm_up)

    def unmap(self, payload, table):
        return self.ctx.unmap(np.ascontiguousarray(payload), table, offset=True)

    def undelta(self, sd, carry):
        return self.ctx.spatial_undelta(np.ascontiguousarray(sd), carry=carry)

    def reconstruct(self, delta):
        return self.ctx.decode_delta(np.ascontiguousarray(delta))


def plan_shards(nt, warm_up, window, world):
    """Contiguous frame ranges [f0, f1) per rank, cut at SWP window starts warm_up + k*window
    (frames [0, warm_up) stay with rank 0).  Every non-empty shard has >= 2 frames; ranks beyond
    the number of usable windows get empty shards (f0 == f1)."""
    if window is None or window < 1:
        raise ValueError("only SWP (-w) shards; DWP (-t) finds its windows sequentially")
    starts = list(range(warm_up, nt, window))
    if starts and nt - starts[-1] < 2 and len(starts) > 1:
        starts.pop()  # a trailing single-frame group joins the previous shard
    g = len(starts)
    used = max(1, min(world, g))
    cuts = [starts[(i * g) // used] for i in range(used)] + [nt]
    cuts[0] = 0
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


def _gather_var(arr, dist, dst=0):
    """Gather variable-length 1-D arrays on `dst` (padded all_gather of raw bytes: gloo has no
    int16 tensors; fine for shard-sized data)."""
    import torch
    dev = _device(dist)
    arr = np.ascontiguousarray(arr).reshape(-1)
    raw = arr.view(np.uint8)
    sizes = [int(v[0]) for v in _all_gather_i64([raw.size], dist)]
    m = max(max(sizes), 1)
    buf = np.zeros(m, dtype=np.uint8)
    buf[: raw.size] = raw
    t = torch.from_numpy(buf).to(dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    if dist.get_rank() != dst:
        return None
    return np.concatenate([o.cpu().numpy()[:s] for o, s in zip(out, sizes)]).view(arr.dtype)


def compress_sharded(engine, frames, warm_up, window, mode, bound, entropy=True):
    """Every rank passes the same arguments (`frames` may be the full stack or any object whose
    [f0:f1] slice yields this rank's frames).  Returns (payload, table|None, key_mask) on rank 0,
    None elsewhere.  The result is byte-identical to a single-GPU tz_rollout + tz_encode."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    nt = frames.shape[0]
    f0, f1 = plan_shards(nt, warm_up, window, world)[rank]
    if f1 > f0:
        key, d = engine.encode_delta(np.asarray(frames[f0:f1]), warm_up if rank == 0 else 0, window, mode, bound)
        d = np.ascontiguousarray(d, np.int16)
        last = [1, int(d[-1])]
    else:
        key, d, last = np.zeros(0, bool), np.zeros(0, np.int16), [0, 0]
    lasts = _all_gather_i64(last, dist)
    carry = None
    for r in range(rank - 1, -1, -1):
        if lasts[r][0]:
            carry = int(lasts[r][1])
            break
    if f1 > f0:
        y, hist = engine.spatial_delta(d, carry, 1 if entropy else 0)
    else:
        y, hist = np.zeros(0, np.int16), (np.zeros(NBINS, np.uint64) if entropy else None)
    table = None
    if entropy:
        dev = _device(dist)
        h = torch.from_numpy(hist.astype(np.int64)).to(dev)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        table = engine.build_table(h.cpu().numpy().astype(np.uint64))
        payload = engine.remap(y, table) if f1 > f0 else y
    else:
        payload = y
    full = _gather_var(np.asarray(payload, np.int16), dist)
    keys = _gather_var(np.asarray(key, np.uint8), dist)
    if rank != 0:
        return None
    return full, table, keys.astype(bool)


def decompress_sharded(engine, key_frames, payload, table, warm_up):
    """Sharded decode: every rank passes the same key-frame stack, payload and table; rank 0
    gets the (nt, H, W, 3) uint8 frames, other ranks None."""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    nt, H, W, C = key_frames.shape
    fe = H * W * C
    # decompress.py:123-129: key frames are the frames with a non-zero sample
    keys = [i for i in range(nt) if np.asarray(key_frames[i]).any()]
    starts = [k for k in keys if k >= warm_up]
    if not starts or starts[0] != warm_up:
        raise ValueError("key frames do not cover the sequence")
    g = len(starts)
    used = max(1, min(world, g))
    cuts = [starts[(i * g) // used] for i in range(used)] + [nt]
    cuts[0] = 0
    shards = [(cuts[i], cuts[i + 1]) for i in range(used)] + [(nt, nt)] * (world - used)
    f0, f1 = shards[rank]
    payload = np.asarray(payload, np.int16).reshape(-1)
    info = [0, 0, 0]
    sd = None
    if f1 > f0:
        engine.decode_prepare(np.asarray(key_frames[f0:f1]), warm_up if rank == 0 else 0)
        part = payload[f0 * fe: f1 * fe]
        sd = engine.unmap(part, table) if table is not None else np.ascontiguousarray(part)
        if rank == 0:
            delta = engine.undelta(sd, None)
            info = [1, int(delta[-1]), 0]            # decoded value at the end of shard 0
        else:
            probe = engine.undelta(sd, 0)            # x = 0 - prefix sums  =>  last = -(shard sum)
            info = [1, 0, (-int(probe[-1])) & 0xFFFF]
    infos = _all_gather_i64(info, dist)
    frames = np.zeros((0, H, W, C), np.uint8)
    if f1 > f0:
        if rank > 0:
            x_end = int(infos[0][1])
            for r in range(1, rank):
                if infos[r][0]:
                    x_end = (x_end - int(infos[r][2])) & 0xFFFF
            carry = x_end - 65536 if x_end >= 32768 else x_end
            delta = engine.undelta(sd, carry)
        frames = engine.reconstruct(np.asarray(delta, np.int16).reshape(f1 - f0, H, W, C))
    out = _gather_var(np.asarray(frames, np.uint8), dist)
    if rank != 0:
        return None
    return out.reshape(nt, H, W, C)
