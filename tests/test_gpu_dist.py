"""Window sharding with the real HIP engine: two processes (both on GPU 0, gloo transport so no
second GPU is needed) must reproduce the single-context result byte for byte."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup():
    from tezip_amd import synth
    from tezip_amd.prednet import PredNetConfig
    cfg = PredNetConfig(stack_sizes=(3, 16, 32))
    frames = synth.translating_scene(17, 24, 40, seed=9)
    return cfg, cfg.init_weights(seed=6, bias_scale=0.1), frames


def _worker(rank, world, port, p, window, mode, bound, entropy, outdir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from tezip_amd import _lib
    from tezip_amd import dist as tzdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        cfg, wts, frames = _setup()
        ctx = _lib.Context(0)
        ctx.load_model(cfg, wts)
        ctx.prepare(24, 40, 4)
        eng = tzdist.HipEngine(ctx)
        res = tzdist.compress_sharded(eng, frames, p, window, mode, bound, entropy)
        ref = np.load(os.path.join(outdir, "ref.npz"))
        if rank == 0:
            payload, table, key = res
            assert (key == ref["key"]).all() and (payload == ref["payload"]).all()
            if entropy:
                assert (table == ref["table"]).all()
        key_stack = np.zeros_like(frames)
        key_stack[ref["key"]] = frames[ref["key"]]
        dec = tzdist.decompress_sharded(eng, key_stack, ref["payload"], ref["table"] if entropy else None, p)
        if rank == 0:
            assert (dec == ref["decoded"]).all()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("p,window,mode,bound,entropy", [(0, 4, "abs", [0.0], True), (1, 5, "abs", [2.0], True),
                                                         (0, 6, "rel", [0.01], False)])
def test_two_rank_sharding_matches_single_context(tmp_path, p, window, mode, bound, entropy):
    import torch.multiprocessing as mp
    from tezip_amd import _lib
    cfg, wts, frames = _setup()
    ctx = _lib.Context(0)
    ctx.load_model(cfg, wts)
    ctx.prepare(24, 40, 8)
    key, _ = ctx.rollout(frames, p, window)
    payload, table, _ = ctx.encode(mode, bound, entropy)
    key_stack = np.zeros_like(frames)
    key_stack[key] = frames[key]
    ctx.rollout_decode(key_stack, p)
    decoded = ctx.decode(payload, table)
    ctx.close()
    if bound[0] == 0:
        assert (decoded == frames).all()
    np.savez(tmp_path / "ref.npz", key=key, payload=payload, table=table if entropy else np.zeros(0, np.int16),
             decoded=decoded)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, p, window, mode, bound, entropy, str(tmp_path)), nprocs=2, join=True)
