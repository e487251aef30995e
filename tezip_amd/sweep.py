"""SWP window-size sweep (BASELINE.json configs[4] "dynamic window-size search (window=5..40)",
SURVEY.md §8d.5).

The reference has no search: `-w` fixes the window (compress.py:249) and `-t` lets the window MSE
cut it (DWP).  "Search" is therefore run as what a user of the reference would do by hand:
compress the same stack once per candidate `-w`, keep the smallest output.  Candidates are
independent jobs, so they shard over the GPUs of a node with nothing exchanged but the resulting
sizes (one value per GPU for the 8 values 5,10,...,40 on 8 GPUs); every candidate's output is
exactly what `tezip.py -c ... -w <value>` writes.
"""
import os

import numpy as np

from . import _lib
from . import dist as tzdist
from .compress import pack_outputs

DEFAULT_WINDOWS = (5, 10, 15, 20, 25, 30, 35, 40)


def sweep(ctx, frames, warm_up, windows, mode, bound, entropy=True, keep_best=True):
    """Compress `frames` (nt,H,W,3 uint8, host) once per window size on this context.
    Returns (rows, best) with rows = [{window, key_frames, key_bytes, entropy_bytes, total_bytes}]
    and best = (window, key_frame.dat bytes, entropy.dat bytes) of the smallest total (ties: the
    smaller window)."""
    nt, h, w = frames.shape[:3]
    hp, wp = _lib.pad8(h), _lib.pad8(w)
    rows, best = [], None
    for win in windows:
        if win < 1:
            raise ValueError("window sizes must be >= 1")
        nwin = max(1, (nt - warm_up + win - 1) // win)
        ctx.prepare(hp, wp, min(nwin, 64))
        key, _ = ctx.rollout(frames, warm_up, win)
        payload, table, _ = ctx.encode(mode, bound, entropy)
        kb, eb = pack_outputs(frames, key, payload, table if entropy else None, warm_up)
        row = dict(window=int(win), key_frames=int(key.sum()), key_bytes=len(kb), entropy_bytes=len(eb),
                   total_bytes=len(kb) + len(eb))
        rows.append(row)
        if best is None or row["total_bytes"] < best[0]["total_bytes"]:
            best = (row, kb if keep_best else None, eb if keep_best else None)
    return rows, (best[0]["window"], best[1], best[2])


def sweep_sharded(ctx, frames, warm_up, windows, mode, bound, entropy=True, ctx_error=None):
    """The same over the ranks of a torch.distributed job: rank r takes windows[r::world]; the
    sizes are all-gathered; returns (rows of every candidate sorted by window, best_window,
    (key bytes, entropy bytes) on the rank that holds the best candidate else None).
    `ctx_error`: the exception this rank met while making its context (ctx is then None); it is
    reported to the other ranks through the all_gather like a failure of the sweep itself."""
    job = tzdist.active()
    if job is None:
        if ctx_error is not None:
            raise ctx_error
        rows, (bw, kb, eb) = sweep(ctx, frames, warm_up, windows, mode, bound, entropy)
        return rows, bw, (kb, eb)
    import torch.distributed as dist
    rank, world = job
    mine = list(windows)[rank::world]
    # a rank whose candidates fail (out of memory, a TezipError, a bad window size) still enters the all_gather and
    # says so in its first element: the others raise a RuntimeError instead of waiting for the watchdog (dist.py)
    rows, best, err = [], (None, None, None), ctx_error
    try:
        if mine and err is None:
            rows, best = sweep(ctx, frames, warm_up, mine, mode, bound, entropy)
    except Exception as e:
        rows, err = [], e
    per = max(1, (len(windows) + world - 1) // world)
    flat = np.full(1 + per * 4, -1, np.int64)
    flat[0] = 1 if err is None else 0
    for i, r in enumerate(rows):
        flat[1 + 4 * i: 5 + 4 * i] = [r["window"], r["key_frames"], r["key_bytes"], r["entropy_bytes"]]
    gathered = tzdist._all_gather_i64(flat.tolist(), dist)
    if err is not None:
        raise err
    tzdist._raise_if_any_failed([int(v[0]) for v in gathered], "window sweep")
    allrows = []
    for v in gathered:
        v = v[1:]
        for i in range(per):
            wv, kf, kbytes, ebytes = (int(x) for x in v[4 * i: 4 * i + 4])
            if wv > 0:
                allrows.append(dict(window=wv, key_frames=kf, key_bytes=kbytes, entropy_bytes=ebytes,
                                    total_bytes=kbytes + ebytes))
    allrows.sort(key=lambda r: r["window"])
    bw = min(allrows, key=lambda r: (r["total_bytes"], r["window"]))["window"]
    return allrows, bw, ((best[1], best[2]) if best[0] == bw else None)


def run(WEIGHTS_DIR, DATA_DIR, OUTPUT_DIR, PREPROCESS, WINDOWS, MODE, BOUND_VALUE, VERBOSE, ENTROPY_RUN, device=0):
    """`tezip.py -c model dir out -p P --sweep 5 10 ... -m MODE -b ...`: writes the output of the
    best window (same three files as compress.run) plus sweep.txt with every candidate's size."""
    from .compress import load_images, make_context, open_model
    from .data_utils import padding_shape
    frames, files, is_rgb = load_images(DATA_DIR)
    nt, H, W = frames.shape[:3]
    cfg, wts, model_shape = open_model(WEIGHTS_DIR)
    hp, wp = padding_shape(H, W)
    # the same checks as compress.run (compress.py:178-181 and the nt >= warm_up + 2 guard)
    if model_shape is not None and (model_shape[0] != hp or model_shape[1] != wp):
        print("ERROR:Image size is out of scope for this model.")
        print("Compatible sizes for this model are height", model_shape[0] - 7, "to", model_shape[0], "and width",
              model_shape[1] - 7, "to", model_shape[1])
        exit()
    if nt < PREPROCESS + 2:
        print("ERROR: need at least warm_up+2 images (%d given, warm_up %d)." % (nt, PREPROCESS))
        exit()
    job = tzdist.active()
    if job:
        device = tzdist.init_from_env()
    ctx, ctx_error, contract = None, None, None
    try:
        ctx = make_context(cfg, wts, hp, wp, 1, device)
        contract = ctx.get_contract()   # TEZIP_PA / the frame size: the same on every rank and for every candidate
    except Exception as e:  # the other ranks are on their way into a collective: tell them there
        ctx_error = e
    try:
        rows, bw, blobs = sweep_sharded(ctx, frames, PREPROCESS, list(WINDOWS or DEFAULT_WINDOWS), MODE, BOUND_VALUE,
                                        ENTROPY_RUN, ctx_error=ctx_error)
    finally:
        if ctx is not None:
            ctx.close()
    rank0 = job is None or job[0] == 0
    mk_error = None
    if rank0 and not os.path.exists(OUTPUT_DIR):
        try:
            os.mkdir(OUTPUT_DIR)
        except OSError as e:
            mk_error = e
    if job:
        import torch.distributed as dist
        # the directory exists before the owner of the best candidate writes into it (and if it could not be made
        # every rank stops here)
        try:
            tzdist._all_ok(mk_error is None, dist, "creating the output directory")
        except RuntimeError:
            if mk_error is not None:
                raise mk_error
            raise
    elif mk_error is not None:
        raise mk_error
    if blobs is not None:
        with open(os.path.join(OUTPUT_DIR, "key_frame.dat"), "wb") as f:
            f.write(blobs[0])
        with open(os.path.join(OUTPUT_DIR, "entropy.dat"), "wb") as f:
            f.write(blobs[1])
    if rank0:
        with open(os.path.join(OUTPUT_DIR, 'filename.txt'), 'w', encoding='UTF-8') as f:
            f.write(f"{int(is_rgb)}\n")
            for name in files:
                f.write("%s\n" % name)
        from . import sidecar
        sidecar.write(OUTPUT_DIR, contract, wts, hp, wp, (nt, H, W, PREPROCESS))
        raw = nt * H * W * (3 if is_rgb else 1)
        with open(os.path.join(OUTPUT_DIR, "sweep.txt"), "w") as f:
            for r in rows:
                line = "window %d: key frames %d, key_frame.dat %d B, entropy.dat %d B, ratio %.4f%s" % (
                    r["window"], r["key_frames"], r["key_bytes"], r["entropy_bytes"], raw / float(r["total_bytes"]),
                    "  <- best" if r["window"] == bw else "")
                f.write(line + "\n")
                if VERBOSE:
                    print(line)
    return rows, bw
