"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the window-sharding
protocol of tezip_amd/dist.py with the oracle plugged in as the per-rank engine, and must
produce exactly the bytes of the single-process run."""
import os
import socket

import numpy as np
import pytest

from tezip_amd import dist as tzdist


def test_plan_shards_cuts_at_window_starts():
    assert tzdist.plan_shards(80, 0, 20, 4) == [(0, 20), (20, 40), (40, 60), (60, 80)]
    assert tzdist.plan_shards(80, 0, 20, 8) == [(0, 20), (20, 40), (40, 60), (60, 80)] + [(80, 80)] * 4
    assert tzdist.plan_shards(14, 2, 4, 2) == [(0, 6), (6, 14)]
    assert tzdist.plan_shards(9, 0, 4, 3) == [(0, 4), (4, 9), (9, 9)]  # trailing 1-frame group is merged
    for nt, p, w, n in [(37, 3, 5, 4), (12, 0, 5, 2), (10, 1, 1, 3)]:
        sh = tzdist.plan_shards(nt, p, w, n)
        assert sh[0][0] == 0 and max(b for _, b in sh) == nt
        for (a, b), (c, _) in zip(sh, sh[1:]):
            assert b == c or (a == b == nt)
        for a, b in sh:
            assert b == a or (b - a >= 2 and (a == 0 or (a - p) % w == 0))
    with pytest.raises(ValueError):
        tzdist.plan_shards(10, 0, None, 2)


def test_plan_shards_never_cuts_a_shard_tz_rollout_would_reject():
    # window = 1 makes one-frame windows: they must be grouped (tz_rollout needs warm_up + 2 frames)
    assert tzdist.plan_shards(4, 0, 1, 2) == [(0, 2), (2, 4)]
    for nt, p, w, n in [(4, 0, 1, 2), (12, 0, 1, 8), (5, 1, 1, 4), (7, 2, 1, 3), (3, 1, 1, 2), (2, 0, 1, 2), (9, 3, 2, 4)]:
        sh = tzdist.plan_shards(nt, p, w, n)
        assert len(sh) == n and sh[0][0] == 0 and max(b for _, b in sh) == nt
        for r, (a, b) in enumerate(sh):
            assert b == a == nt or b - a >= (p + 2 if r == 0 else 2), (nt, p, w, n, sh)
            assert a == 0 or a == nt or (a - p) % w == 0
        used = [x for x in sh if x[1] > x[0]]
        assert all(x[1] == y[0] for x, y in zip(used, used[1:]))
    with pytest.raises(ValueError):
        tzdist.plan_shards(3, 2, 1, 2)  # nt < warm_up + 2: the reference breaks there too


def test_plan_decode_shards_cut_at_key_frames():
    assert tzdist.plan_decode_shards([0, 4, 8, 12], 14, 0, 2) == [(0, 8), (8, 14)]
    assert tzdist.plan_decode_shards([0, 1, 2, 6, 10], 14, 2, 3) == [(0, 6), (6, 10), (10, 14)]
    assert tzdist.plan_decode_shards([0, 1, 2, 3], 4, 0, 2) == [(0, 2), (2, 4)]
    with pytest.raises(ValueError):
        tzdist.plan_decode_shards([1, 5], 9, 0, 2)


class OracleEngine:
    """Per-rank compute done by the CPU oracle (test double for dist.HipEngine)."""

    def __init__(self, pred):
        from oracle import coracle, oracle
        self.O, self.C, self.pred = oracle, coracle, pred

    def encode_begin(self, frames, warm_up, window, mode, bound, entropy):
        """What tz_rollout + tz_encode_begin do: symbols of the shard taken WITHOUT a carry, their
        histogram, first / last element of the quantised delta stack."""
        ro = self.O.rollout(frames, warm_up, window, None, self.pred)
        d = self.O.encode_stream(frames, ro, warm_up, mode, bound, entropy=False)["delta"].reshape(-1)
        self.entropy, self.first = entropy, int(d[0])
        self.y = self.C.spatial_delta(d, 1 if entropy else 0)
        hist = self.C.histogram(self.y).astype(np.uint64) if entropy else None
        return ro["key"], hist, int(d[0]), int(d[-1])

    def encode_finish(self, carry, table):
        """tz_encode_finish: the first symbol is re-made with the carry, then the remap."""
        y = self.y.copy()
        if carry is not None:
            sd = np.int16(np.int64(carry) - self.first)
            y[0] = np.int16(1600 - np.int64(sd)) if self.entropy else sd
        return self.O.remap_enc(y, table) if table is not None else y

    def build_table(self, hist):
        syms = [int(s) for s in np.nonzero(hist)[0]]
        syms.sort(key=lambda s: int(hist[s]), reverse=True)
        return np.array(syms, np.int16)

    def decode_prepare(self, key_frames, warm_up):
        self.key_frames = key_frames
        self.x_hat, _ = self.O.decoder_rollout(key_frames, warm_up, self.pred)

    def unmap(self, payload, table):
        return (1600 - self.O.remap_dec(payload, table).astype(np.int64)).astype(np.int16)

    def undelta(self, sd, carry):
        if carry is None:
            return self.O.finding_difference_dec(sd)
        return self.O.finding_difference_dec(np.concatenate([[carry], sd]).astype(np.int16))[1:]

    def reconstruct(self, delta):
        h, w = delta.shape[1:3]
        return self.O.reconstruct(self.x_hat[:, :h, :w], delta)


class OracleContext:
    """Test double for tezip_amd._lib.Context as sweep.sweep drives it (prepare / rollout / encode), computed by
    the CPU oracle."""

    def __init__(self, pred, fail_on_window=None):
        from oracle import oracle
        self.O, self.pred, self.fail_on_window = oracle, pred, fail_on_window

    def prepare(self, hp, wp, max_batch):
        self.prepared = (hp, wp, max_batch)

    def rollout(self, frames, warm_up, window):
        if window == self.fail_on_window:
            raise MemoryError("injected failure at window %d" % window)
        self.job = (frames, warm_up, window)
        self.ro = self.O.rollout(frames, warm_up, window, None, self.pred)
        return self.ro["key"], None

    def encode(self, mode, bound, entropy):
        frames, warm_up, window = self.job
        st = self.O.encode_stream(frames, self.ro, warm_up, mode, bound, entropy=entropy)
        payload, table, _, _ = self.O.parse_stream(st["stream"])
        return payload, table, None


def _case(nt=14, h=13, w=19):
    import fake_predictor
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    frames = np.stack([np.clip(np.stack([120 + 60 * np.sin((xx + 2 * t) / 5.0), 90 + 40 * np.cos((yy - t) / 4.0),
                                         128 + 0.0 * xx], -1) + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
                       for t in range(nt)])
    pred = O.FnPredictor(fake_predictor.c0_image, fake_predictor.g_next)
    return frames, pred


def _worker(rank, world, port, p, window, mode, bound, entropy):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(here, "golden")]
    import torch.distributed as dist
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        frames, pred = _case()
        eng = OracleEngine(pred)
        res = tzdist.compress_sharded(eng, frames, p, window, mode, bound, entropy)
        ref = O.compress_oracle(frames, p, window, None, mode, bound, pred, entropy)
        ref_payload, ref_table, _, _ = O.parse_stream(ref["stream"])
        if rank == 0:
            payload, table, key = res
            assert (key == ref["key"]).all()
            assert payload.shape == ref_payload.shape and (payload == ref_payload).all()
            assert (table is None) == (ref_table is None) and (table is None or (table == ref_table).all())
        else:
            assert res is None
        # pipelined form: two sequences in flight, the gather of the first behind the compute of the second
        p1 = tzdist.compress_sharded(eng, frames, p, window, mode, bound, entropy, wait=False)
        p2 = tzdist.compress_sharded(eng, frames[::-1].copy() if window > 1 else frames, p, window, mode, bound, entropy, wait=False)
        r1, r2 = p1.wait(), p2.wait()
        if rank == 0:
            assert (r1[0] == ref_payload).all() and (r1[2] == ref["key"]).all()
            if window > 1:
                ref2 = O.compress_oracle(frames[::-1].copy(), p, window, None, mode, bound, pred, entropy)
                assert (r2[0] == O.parse_stream(ref2["stream"])[0]).all()
        else:
            assert r1 is None and r2 is None
        key_stack = ref["key_frame"].reshape(frames.shape)
        dec = tzdist.decompress_sharded(eng, key_stack, ref_payload, ref_table, p)
        if rank == 0:
            assert (dec == O.decode_stream(ref["stream"], ref["key_frame"], pred)).all()
            if bound[0] == 0:
                assert (dec == frames).all()
        else:
            assert dec is None
    finally:
        dist.destroy_process_group()


def _failing_worker(rank, world, port, stage):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(here, "golden")]
    import torch.distributed as dist
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        frames, pred = _case()
        eng = OracleEngine(pred)

        def boom(*a, **k):
            raise MemoryError("injected failure on rank 1 (%s)" % stage)
        decode = stage in ("decode_prepare", "undelta", "reconstruct")
        if rank == 1:
            if stage == "undelta":      # the first undelta is the probe of stage 1: fail the SECOND one
                real, calls = eng.undelta, []

                def second(*a, **k):
                    calls.append(1)
                    return real(*a, **k) if len(calls) == 1 else boom()
                eng.undelta = second
            else:
                setattr(eng, stage, boom)
        try:
            if decode:
                ref = O.compress_oracle(frames, 0, 4, None, "abs", [0.0], pred, True)
                pl, tb, _, _ = O.parse_stream(ref["stream"])
                tzdist.decompress_sharded(eng, ref["key_frame"].reshape(frames.shape), pl, tb, 0)
            else:
                tzdist.compress_sharded(eng, frames, 0, 4, "abs", [0.0], True)
        except MemoryError:
            assert rank == 1
        except RuntimeError as e:  # the healthy rank learns of it instead of hanging in a collective
            assert rank == 0 and "failed on" in str(e), str(e)
        else:
            raise AssertionError("a failed rank went unnoticed")
    finally:
        dist.destroy_process_group()


def _contract_worker(rank, world, port, contracts):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(here, "golden")]
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        frames, pred = _case()
        eng = OracleEngine(pred)
        eng.contract = lambda: contracts[rank]
        agree = len({c for c in contracts if c}) <= 1
        try:
            res = tzdist.compress_sharded(eng, frames, 0, 4, "abs", [0.0], True)
        except RuntimeError as e:   # EVERY rank raises: nobody goes on into the histogram all-reduce alone
            assert not agree and "different arithmetic contracts" in str(e) and "TZ-PA1" in str(e) and "TZ-PA2" in str(e), str(e)
        else:
            assert agree and (res is None) == (rank != 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("contracts", [(2, 1), (2, 2), (0, 2)])
def test_ranks_must_agree_on_the_arithmetic_contract(contracts):
    """One stream is decoded under ONE contract (the one rank 0 stamps into tezip_amd.json): ranks whose TEZIP_PA differ
    are found out in the first all_gather of compress_sharded; an engine without a contract (0: the CPU
    engines of these tests) takes no part in the comparison."""
    import torch.multiprocessing as mp
    mp.spawn(_contract_worker, args=(2, _free_port(), contracts), nprocs=2, join=True)


@pytest.mark.parametrize("stage", ["encode_begin", "build_table", "encode_finish", "decode_prepare", "undelta", "reconstruct"])
def test_a_failing_rank_stops_every_rank(stage):
    """A failure in ANY compute stage between two collectives (not only the first one) reaches the
    other ranks through the next collective: nobody waits in an all-reduce or a receive."""
    import torch.multiprocessing as mp
    mp.spawn(_failing_worker, args=(2, _free_port(), stage), nprocs=2, join=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,p,window,mode,bound,entropy", [
    (2, 0, 4, "abs", [0.0], True),
    (2, 2, 3, "abs", [3.0], True),
    (3, 0, 5, "rel", [0.02], False),
    (3, 1, 2, "pwrel", [0.05], True),
    (3, 0, 1, "abs", [1.0], True),     # one-frame windows: shards are groups of them
])
def test_sharded_equals_single_process(world, p, window, mode, bound, entropy):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), p, window, mode, bound, entropy), nprocs=world, join=True)


def _sweep_worker(rank, world, port, windows, p, mode, bound, fail_on_window):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(here, "golden")]
    import torch.distributed as dist
    from tezip_amd import sweep
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        frames, pred = _case()
        # what one process finds (compress.py:249 run once per candidate -w by hand)
        ref_rows, (ref_w, ref_kb, ref_eb) = sweep.sweep(OracleContext(pred), frames, p, windows, mode, bound)
        if fail_on_window is None:
            rows, bw, blobs = sweep.sweep_sharded(OracleContext(pred), frames, p, windows, mode, bound)
            assert rows == sorted(ref_rows, key=lambda r: r["window"])
            assert bw == ref_w
            owner = list(windows).index(bw) % world
            if rank == owner:      # the rank that compressed the best candidate holds its two files
                assert blobs is not None and blobs[0] == ref_kb and blobs[1] == ref_eb
            else:
                assert blobs is None
            return
        owner = list(windows).index(fail_on_window) % world
        try:
            sweep.sweep_sharded(OracleContext(pred, fail_on_window), frames, p, windows, mode, bound)
        except MemoryError:
            assert rank == owner
        except RuntimeError as e:
            assert rank != owner and "window sweep failed on rank(s) [%d]" % owner in str(e), str(e)
        else:
            raise AssertionError("a failed rank went unnoticed")
        # a rank that could not even make its context reports through the same collective
        try:
            sweep.sweep_sharded(None if rank == 1 else OracleContext(pred), frames, p, windows, mode, bound,
                                ctx_error=OSError("no device") if rank == 1 else None)
        except OSError:
            assert rank == 1
        except RuntimeError as e:
            assert rank != 1 and "[1]" in str(e)
        else:
            raise AssertionError("a rank without a context went unnoticed")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,windows,p,mode,bound", [
    (2, (2, 3, 4, 5, 7), 0, "abs", [0.0]),
    (3, (1, 2, 3, 4, 5, 6, 13, 20), 0, "abs", [0.0]),     # 8 candidates on 3 ranks (3 + 3 + 2), a window longer than nt
    (3, (3, 4), 2, "abs", [2.0]),                          # fewer candidates than ranks: rank 2 has nothing to do
])
def test_sweep_sharded_equals_single_process(world, windows, p, mode, bound):
    """BASELINE configs[4] on N ranks: one candidate window size per rank (sweep.sweep_sharded), sizes all-gathered --
    rows, best window and the owner's two files equal the one-process sweep."""
    import torch.multiprocessing as mp
    mp.spawn(_sweep_worker, args=(world, _free_port(), windows, p, mode, bound, None), nprocs=world, join=True)


@pytest.mark.parametrize("world,fail_on_window", [(2, 3), (3, 5)])
def test_sweep_failure_on_one_rank_stops_every_rank(world, fail_on_window):
    import torch.multiprocessing as mp
    mp.spawn(_sweep_worker, args=(world, _free_port(), (2, 3, 4, 5, 7), 0, "abs", [0.0], fail_on_window), nprocs=world, join=True)


def _rank_local_worker(rank, world, port, nt, window, mode, bound):
    """A rank is handed a FETCH callable instead of the stack (what the sharded CLI does with the image files): it must
    ask for its own frame range and nothing else, and with gather=False the decoder gives every rank its own frames."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), os.path.join(here, "golden")]
    import torch.distributed as dist
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        frames, pred = _case(nt, 16, 16)
        frames = np.repeat(frames[..., :1], 3, axis=-1)          # cfg4's detector frames are 'L' expanded to RGB
        shards = tzdist.plan_shards(nt, 0, window, world)
        asked = []

        def fetch(a, b):
            asked.append((a, b))
            return frames[a:b]

        eng = OracleEngine(pred)
        res = tzdist.compress_sharded(eng, fetch, 0, window, mode, bound, True, nt=nt)
        f0, f1 = shards[rank]
        assert asked == ([(f0, f1)] if f1 > f0 else []), (rank, asked, shards)
        ref = O.compress_oracle(frames, 0, window, None, mode, bound, pred, True)
        ref_payload, ref_table, _, _ = O.parse_stream(ref["stream"])
        if rank == 0:
            payload, table, key = res
            assert (key == ref["key"]).all() and (payload == ref_payload).all() and (table == ref_table).all()
        else:
            assert res is None
        a, b, mine = tzdist.decompress_sharded(eng, ref["key_frame"].reshape(frames.shape), ref_payload, ref_table, 0,
                                               gather=False)
        whole = O.decode_stream(ref["stream"], ref["key_frame"], pred)
        assert mine.shape == (b - a,) + frames.shape[1:] and (mine == whole[a:b]).all()
        # the ranks' ranges tile the sequence: together they hold every frame once
        spans = [None] * world
        dist.all_gather_object(spans, (a, b))
        used = [s for s in spans if s[1] > s[0]]
        assert used[0][0] == 0 and used[-1][1] == nt and all(x[1] == y[0] for x, y in zip(used, used[1:]))
        if nt // window >= world:
            assert len(used) == world
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nt,window,mode,bound", [
    (32, 4, "abs", [2.0]),     # 8 windows on 8 ranks: one window per rank, the split BASELINE.json configs[3] names
    (20, 4, "abs", [0.0]),     # 5 windows on 8 ranks: three ranks hold nothing and still take part in every collective
])
def test_world_8_one_window_per_rank(nt, window, mode, bound):
    """World size 8 (north_star: 'one window per GPU'; /root/reference has no counterpart) on gloo: byte
    identical to one process, every rank touching its own frames only."""
    import torch.multiprocessing as mp
    mp.spawn(_rank_local_worker, args=(8, _free_port(), nt, window, mode, bound), nprocs=8, join=True)
