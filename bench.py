#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

metric : frames/sec (predict + delta-encode, whole job over all ranks), 512x512x3 sequences
step   : one pass of the hot path over one synthetic sequence that is already resident in HBM:
         PredNet rollout (SWP) + delta, error-bound quantise, spatial delta, histogram, rank
         table, remap -> int16 payload in (rank 0's) HBM + rank table on the host.
workload (BASELINE.json configs[2]): 512x512x3 synthetic turbulence stack, 80 frames per GPU,
         20-frame windows, lossy `rel 1e-3`, synthetic glorot weights (seed 123) of the reference
         model.
N>1    : one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (a
         torch.distributed.run child, before anything touches the GPU); under torchrun
         (WORLD_SIZE set) it is one of the ranks.  The job is ONE sequence of 80*N frames whose
         windows are sharded over the ranks (tezip_amd/dist.py: no data-path collective; one small
         all_gather for the shard-boundary carries, an all-reduce of 2111 histogram counters, and
         the payload shards sent point to point to rank 0, in flight while the next step rolls out; the
         clock stops only after the last shards have landed) => weak scaling, 4 windows per GPU.
         The sharded result is checked byte-identical to a single-GPU run of the same sequence
         (untimed).  `--mode replicas` (every rank its own 80-frame sequence, nothing exchanged)
         is kept and is reported as an extra key of the same line.

Extra objects on the JSON line: "roofline" (dominant kernel = MFMA convolution),
"roofline_delta" (the HBM-bound delta kernel named by the north star), "cpu_baseline"
(the C oracle = a port of the same path, timed on this box's host cores, rank 0, N=1 only),
"host_to_host" (SURVEY.md §8d's wall-clock definition: uint8 stack in host memory -> int16
payload + table in host memory, PCIe included, pinned and pageable buffers), "lossy_abs2" (the
same step with `abs 2`, real merges in the quantiser), "cfg4_sharded" (BASELINE configs[3]:
ONE 320-frame 1024x1024 sequence, 40-frame windows, `abs 2`, windows sharded over the ranks:
strong scaling), "decode" + "roofline_undelta" (the inverse path of decompress.py on the same job, device
resident: rollout replay + inverse remap fused into the single-pass inverse scan + reconstruct),
"sharded_path" (N = 1 only: the SAME job driven through dist.compress_sharded on a one-rank RCCL
process group, i.e. what every rank of an N > 1 run executes, beside the fused rate) and
"compression_ratio_trained" (untimed: a PredNet trained here for ~12 s with tezip_amd/train.py on
held-out synthetic turbulence, ratios of the cfg3 job lossless and at `abs 2`; the random-weights
ratio of the timed job is meaningless and only reported), "host_pipeline" (SURVEY.md 8d "report them
separately": the same job as 80 PNG files through compress.run / decompress.run, wall seconds and
stage times incl. PNG decode / encode and zstd-9) and "roofline_encode_tail" (the HBM roofline of the
elementwise tail the step itself launches).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H = W = 512
NT = 80            # frames per GPU
WINDOW = 20
WARM_UP = 0
MODE, BOUND = "rel", [1e-3]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0


# ------------------------------------------------------------------------------ synthetic data
def _turb_terms(h, w, seed, device):
    """Same construction as tezip_amd.synth.turbulence, evaluated on the GPU."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    terms = []
    for c in range(3):
        for o in range(6):
            r = torch.rand(4, generator=g).tolist()
            f = (2.0 ** o) * 2 * np.pi / max(h, w)
            th, ph = r[0] * 2 * np.pi, r[1] * 2 * np.pi
            vx, vy = r[2] * 3 - 1.5, r[3] * 3 - 1.5
            kx, ky = f * np.cos(th), f * np.sin(th)
            terms.append((c, 1.0 / (1.5 ** o), kx * xx + ky * yy + ph, kx * vx + ky * vy))
    return terms


def _turb_raw(terms, t0, t1, h, w, device):
    ts = torch.arange(t0, t1, device=device, dtype=torch.float32)
    out = torch.zeros((t1 - t0, h, w, 3), device=device, dtype=torch.float32)
    for c, amp, base, om in terms:
        out[..., c] += amp * torch.sin(base[None] - (om * ts)[:, None, None])
    return out


def turbulence_cuda(nt_total, t0, t1, h, w, seed, device):
    """Frames [t0, t1) of the nt_total-frame turbulence sequence (normalised over the WHOLE
    sequence, so a shard is exactly a slice of the full stack)."""
    terms = _turb_terms(h, w, seed, device)
    lo, hi = float("inf"), float("-inf")
    for a in range(0, nt_total, 40):
        raw = _turb_raw(terms, a, min(a + 40, nt_total), h, w, device)
        lo, hi = min(lo, float(raw.min())), max(hi, float(raw.max()))
    out = torch.empty((t1 - t0, h, w, 3), dtype=torch.uint8, device=device)
    for a in range(t0, t1, 40):
        b = min(a + 40, t1)
        raw = (_turb_raw(terms, a, b, h, w, device) - lo) / (hi - lo) * 255
        out[a - t0: b - t0] = raw.round().clamp(0, 255).to(torch.uint8)
    return out.contiguous()


def detector_cuda(t0, t1, h, w, seed, device):
    """Frames [t0, t1) of a cfg4-style XFEL detector sequence (tezip_amd.synth.detector's
    construction on the GPU): Poisson background (lambda 3) + 50 slowly drifting Gaussian peaks,
    grayscale expanded to 3 channels as the reference does (compress.py:114).  Every frame is
    seeded by its index, so a shard is exactly a slice of the full stack."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    npk = 50
    pos = torch.rand(npk, 2, generator=g) * torch.tensor([h, w], dtype=torch.float32)
    vel = torch.randn(npk, 2, generator=g) * 0.3
    amp = (40 + 180 * torch.rand(npk, generator=g)).to(device)
    sig = (1.5 + 2.5 * torch.rand(npk, generator=g)).to(device)
    pos, vel = pos.to(device), vel.to(device)
    yy, xx = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    out = torch.empty((t1 - t0, h, w, 3), dtype=torch.uint8, device=device)
    gd = torch.Generator(device=device)
    for t in range(t0, t1):
        gd.manual_seed(seed * 1000003 + t)
        img = torch.poisson(torch.full((h, w), 3.0, device=device), generator=gd)
        p = pos + vel * t
        for k0 in range(0, npk, 10):
            k1 = min(k0 + 10, npk)
            d2 = (yy[None] - p[k0:k1, 0, None, None]) ** 2 + (xx[None] - p[k0:k1, 1, None, None]) ** 2
            img += (amp[k0:k1, None, None] * torch.exp(-d2 / (2 * sig[k0:k1, None, None] ** 2))).sum(0)
        out[t - t0] = img.clamp(0, 255).to(torch.uint8)[..., None]
    return out.contiguous()


# ------------------------------------------------------------------------------ accounting
def measured_traffic(prefix):
    """Per-launch HBM bytes of the kernels whose name starts with `prefix`, from the committed
    rocprofv3 PMC summary of this same bench command (profiles/CURRENT -> traffic.json, made by
    profiles/collect.sh + profiles/summarize.py; FETCH_SIZE/WRITE_SIZE corrected per
    MI355X_MICROARCH.md).  None when no profile is committed."""
    try:
        cur = open(os.path.join(ROOT, "profiles", "CURRENT")).read().strip()
        t = json.load(open(os.path.join(ROOT, "profiles", cur, "traffic.json")))
        num = den = 0.0
        for k, v in t.items():
            if k.startswith(prefix) and "hbm_bytes_per_launch" in v:
                num += v["hbm_bytes_per_launch"] * v.get("launches", 1)
                den += v.get("launches", 1)
        return num / den if den else None
    except Exception:
        return None


def live_flops_per_px0(cfg):
    """MACs*2 per level-0 pixel that the per-frame path executes (SURVEY.md §8d 'live work',
    minus the r_{t-1} part of the gate convolutions, which is constant per model and folded
    into G0 at prepare time, and with the upsampled source up(r_{l+1}) evaluated through 4
    collapsed taps instead of 9: DESIGN.md §3)."""
    st, rs, L = cfg.stack_sizes, cfg.R_stack_sizes, cfg.nb_layers
    mac = 0.0
    for l in range(L):
        taps_ch = 9 * 2 * st[l] + (4 * rs[l + 1] if l < L - 1 else 0)
        mac += taps_ch * 4 * rs[l] / 4 ** l          # t1 gates over [e_l (9 taps), up(r_{l+1}) (4 taps)]
        if l < L - 1:
            mac += 9 * 2 * st[l] * st[l + 1] / 4 ** l  # t0 A conv
    mac += 9 * rs[0] * st[0]                          # Ahat_0 at t1
    return 2 * mac


def wino_executed_flops_per_px0(cfg):
    """FLOPs the TZ-PA2 kernel k_wino EXECUTES for the convolutions conv16_flops_per_px0 counts: the same-resolution source
    of every one of them through F(2x2, 3x3) -- 16 multiplies per 2x2 outputs and input channel instead of 36, i.e. 1/2.25 of
    the direct form's --, the upsampled source through its 4 collapsed taps as before."""
    st, rs, L = cfg.stack_sizes, cfg.R_stack_sizes, cfg.nb_layers
    mac = 0.0
    for l in range(L):
        gate_srcs_ok = (2 * st[l]) % 16 == 0 and (l == L - 1 or rs[l + 1] % 16 == 0) and rs[l] % 16 == 0
        if gate_srcs_ok:
            mac += (4 * 2 * st[l] + (4 * rs[l + 1] if l < L - 1 else 0)) * 4 * rs[l] / 4 ** l
        if l < L - 1 and (2 * st[l]) % 16 == 0 and (st[l + 1] % 64 == 0 or st[l + 1] % 48 == 0):   # tz_prednet.hip plain_nt
            mac += 4 * 2 * st[l] * st[l + 1] / 4 ** l
    return 2 * mac


def conv16_flops_per_px0(cfg):
    """The part of live_flops_per_px0 that k_conv16 executes: gate convolutions and A convolutions
    whose sources all have a multiple of 16 channels (levels >= 1 of the default model)."""
    st, rs, L = cfg.stack_sizes, cfg.R_stack_sizes, cfg.nb_layers
    mac = 0.0
    for l in range(L):
        gate_srcs_ok = (2 * st[l]) % 16 == 0 and (l == L - 1 or rs[l + 1] % 16 == 0) and rs[l] % 16 == 0
        if gate_srcs_ok:
            mac += (9 * 2 * st[l] + (4 * rs[l + 1] if l < L - 1 else 0)) * 4 * rs[l] / 4 ** l
        if l < L - 1 and (2 * st[l]) % 16 == 0 and (st[l + 1] % 64 == 0 or st[l + 1] % 48 == 0):   # tz_prednet.hip plain_nt
            mac += 9 * 2 * st[l] * st[l + 1] / 4 ** l
    return 2 * mac


# ------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without torchrun: start the N ranks as a torch.distributed.run
    child.  Nothing in THIS process has touched the GPU (device_count() does not initialise it),
    and the child is a fresh process, not an exec of this one."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (a rehearsal asked for explicitly never counts devices here: on a one-GPU box the pool allows six processes with the
    # GPU open, and hipGetDeviceCount in this launcher would be one of them)
    have = 0 if env.get("TEZIP_BENCH_SINGLE_DEVICE") else torch.cuda.device_count()
    if have < n:  # rehearsal on a smaller box: all ranks share GPU 0, gloo carries the exchange
        print("bench.py: %d GPU(s) visible for --gpus %d: rehearsing with all ranks on GPU 0 over gloo" % (have, n),
              file=sys.stderr)
        env["TEZIP_BENCH_SINGLE_DEVICE"] = "1"
        env.setdefault("TEZIP_BENCH_BACKEND", "gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    # the ranks' stdout passes through a filter: ONE JSON line is the contract, and libraries (gloo: "[Gloo] Rank 0
    # is connected to ...") write to stdout too -- everything that is not the bench line goes to stderr
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in p.stdout:
        is_line = line.startswith('{"metric"')
        (sys.stdout if is_line else sys.stderr).write(line)
        (sys.stdout if is_line else sys.stderr).flush()
    return p.wait()


# ------------------------------------------------------------------------------ timing helper
class Job:
    def __init__(self, rank, world, dist, dev):
        self.rank, self.world, self.dist, self.dev = rank, world, dist, dev

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, step, steps, warmup, drain=None):
        """W untimed + K timed steps between barrier + synchronize; MAX over ranks (seconds).
        drain: completes whatever the last step left in flight, inside the timed region."""
        for _ in range(warmup):
            step()
        if drain:
            drain()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if drain:
            drain()
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.dist:
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device=self.dev if self.dist.get_backend() == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed


def sharded_step_fn(job, ctx, engine, fetch, nt_total, warm_up, window, mode, bound, state):
    """One whole-job step: windows sharded over the ranks (rank 0 ends with the payload in HBM)."""
    from tezip_amd import dist as tzdist
    if job.world == 1:
        def step():
            frames = fetch(0, nt_total)
            key, _ = ctx.rollout(frames, warm_up, window)
            _, table, _ = ctx.encode(mode, bound, True, payload=state["payload"])
            state["key"], state["table"] = key, table
    else:
        # the payload shards of step k travel to rank 0 while step k + 1 rolls out (fresh output buffers
        # every step); drain() lands the last ones before the clock stops
        def drain():
            p = state.pop("pending", None)
            if p is not None:
                res = p.wait()
                if res is not None:
                    state["payload"], state["table"], state["key"] = res

        def step():
            nxt = tzdist.compress_sharded(engine, fetch, warm_up, window, mode, bound, True, nt=nt_total, to_host=False,
                                          wait=False)
            drain()
            state["pending"] = nxt
        step.drain = drain
    return step


def verify_against_one_gpu(ctx, frames_full, warm_up, window, mode, bound, state, max_batch, hp, wp):
    """Rank 0, untimed: the sharded payload / table / key mask must equal a single-GPU run."""
    ctx.prepare(hp, wp, max_batch)
    ref = torch.empty(frames_full.numel(), dtype=torch.int16, device=frames_full.device)
    key, _ = ctx.rollout(frames_full, warm_up, window)
    _, table, _ = ctx.encode(mode, bound, True, payload=ref)
    ctx.synchronize()
    pl = state["payload"]
    if isinstance(pl, np.ndarray):  # gloo rehearsal: the gathered payload lives on the host
        pl = torch.from_numpy(pl).to(ref.device)
    state = dict(state, payload=pl)
    ok = bool((key == state["key"]).all()) and len(table) == len(state["table"]) and bool((table == state["table"]).all()) \
        and bool(torch.equal(ref, state["payload"]))
    return "byte-identical to the 1-GPU result" if ok else "MISMATCH vs the 1-GPU result"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded",
                    help="N>1: shard the windows of ONE 80*N-frame sequence (default) or one 80-frame sequence per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip host_to_host / lossy_abs2 / cfg4_sharded / the other mode")
    ap.add_argument("--no-host-pipeline", action="store_true", help="skip the PNG -> compress.run -> decompress.run -> PNG leg")
    ap.add_argument("--no-trained-ratio", action="store_true", help="skip training a model for compression_ratio_trained")
    ap.add_argument("--profile-legs", action="store_true",
                    help="with --no-extras: still run ONE untimed `abs 2` encode and ONE decode of the job, so that a rocprofv3 "
                         "trace of this command also holds the lossy-tail and the decoder kernels (profiles/collect.sh)")
    ap.add_argument("--frames", type=int, default=NT,
                    help="frames per GPU (default 80 = the BASELINE.json configuration; other values are "
                         "exploration only and are labelled as such in config.workload)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    explore = args.frames != NT
    nt_rank = args.frames
    # ONE JSON line on stdout is the contract, and native libraries write there too (RCCL prints a version
    # banner when its first communicator comes up): from here on file descriptor 1 IS stderr; the bench
    # line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback)")
    single_dev = bool(os.environ.get("TEZIP_BENCH_SINGLE_DEVICE"))  # rehearsal of N>1 on a one-GPU box
    if single_dev:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("TEZIP_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from tezip_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    from tezip_amd import dist as tzdist
    from tezip_amd.prednet import PredNetConfig

    dev = torch.device("cuda", local)
    job = Job(rank, world, dist, dev)
    ctx = _lib.Context(local)   # a stream of its own: torch.cuda.synchronize() orders it with the torch side where they meet
    engine = tzdist.HipEngine(ctx, local)
    cfg = PredNetConfig()
    ctx.load_model(cfg, cfg.init_weights(seed=123))
    nwin = (nt_rank - WARM_UP + WINDOW - 1) // WINDOW
    ctx.prepare(H, W, max_batch=nwin)

    # ---------------------------------------------------------------- primary line (cfg3)
    sharded = args.mode == "sharded"
    nt_total = nt_rank * world if sharded else nt_rank
    if sharded:
        f0, f1 = tzdist.plan_shards(nt_total, WARM_UP, WINDOW, world)[rank] if world > 1 else (0, nt_total)
        frames = turbulence_cuda(nt_total, f0, f1, H, W, 3, dev)
    else:
        f0, f1 = 0, nt_rank
        frames = turbulence_cuda(nt_rank, 0, nt_rank, H, W, 3 + rank, dev)
    state = {"payload": torch.empty(nt_total * H * W * 3, dtype=torch.int16, device=dev) if world == 1 or not sharded else None}

    def fetch(a, b):
        assert (a, b) == (f0, f1), "a rank only holds its own shard"
        return frames

    if sharded:
        step = sharded_step_fn(job, ctx, engine, fetch, nt_total, WARM_UP, WINDOW, MODE, BOUND, state)
    else:
        def step():
            key, _ = ctx.rollout(frames, WARM_UP, WINDOW)
            _, table, _ = ctx.encode(MODE, BOUND, True, payload=state["payload"])
            state["key"], state["table"] = key, table
    elapsed = job.timed(step, args.steps, args.warmup, drain=getattr(step, "drain", None))
    total_frames = (nt_total if sharded else nt_rank * world) * args.steps
    value = total_frames / elapsed

    check = None
    if sharded and world > 1 and rank == 0:
        full = turbulence_cuda(nt_total, 0, nt_total, H, W, 3, dev)
        check = verify_against_one_gpu(ctx, full, WARM_UP, WINDOW, MODE, BOUND, state, nwin * world, H, W)
        del full
        ctx.prepare(H, W, max_batch=nwin)
    if dist:
        dist.barrier()

    # ---------------------------------------------------------------- per-kernel device time (rank 0's shard)
    own_state = {"payload": torch.empty(frames.shape[0] * H * W * 3, dtype=torch.int16, device=dev)}

    def own_step(mode=MODE, bound=BOUND):
        key, _ = ctx.rollout(frames, WARM_UP if rank == 0 or not sharded else 0, WINDOW)
        _, table, _ = ctx.encode(mode, bound, True, payload=own_state["payload"])
        own_state["key"], own_state["table"] = key, table

    ctx.prof_enable(True)
    ctx.prof_reset()
    own_step()
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    n_pred = frames.shape[0] - int(own_state["key"].sum())
    conv_ms, conv_n = prof["conv3x3_mfma"]
    flops_step = live_flops_per_px0(cfg) * H * W * n_pred
    conv_tflops = flops_step / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    SUB = ("conv16_lds_dma", "conv16b_level0", "conv_small_valu", "conv3x3_general", "convlat_small_grid", "wino_pa2")  # sub-classes of conv3x3_mfma
    step_dev_ms = sum(v[0] for k, v in prof.items() if k not in SUB)
    c16_ms, c16_n = prof["conv16_lds_dma"]               # the dominant kernel on its own
    wino_ms, wino_n = prof.get("wino_pa2", (0.0, 0))      # ... under TZ-PA2 (the contract in force at this frame size)
    contract = ctx.get_contract()
    lat_ms, lat_n = prof.get("convlat_small_grid", (0.0, 0))
    # (with fewer than 4 windows per GPU -- `--frames 40` -- the upper levels run on k_convlat: same convolutions,
    # same FLOPs; counted with the dominant kernel so that the fraction stays FLOPs of these launches / their time)
    c16_ms, c16_n = c16_ms + lat_ms, c16_n + lat_n
    c16_flops = conv16_flops_per_px0(cfg) * H * W * n_pred
    c16_tflops = c16_flops / (c16_ms * 1e-3) / 1e12 if c16_ms > 0 else 0.0
    wino_flops = wino_executed_flops_per_px0(cfg) * H * W * n_pred
    wino_tflops = wino_flops / (wino_ms * 1e-3) / 1e12 if wino_ms > 0 else 0.0
    wino_direct_tflops = c16_flops / (wino_ms * 1e-3) / 1e12 if wino_ms > 0 else 0.0
    # the elementwise delta kernel the north star names (compress.py:292-314).  Since round 3 the STEP forms its deltas inside
    # the fused kernels (k_q_tiles reads pred / orig directly; lossless: k_delta_sd_fused), so the stand-alone kernel is
    # measured through its own C-ABI entry point (tz_delta_encode) on the step's own prediction stack, device resident
    pred_dev = torch.empty(frames.shape[0] * H * W * 3, dtype=torch.float32, device=dev)
    delta_dev = torch.empty(frames.shape[0] * H * W * 3, dtype=torch.int16, device=dev)
    ctx.get_predictions(out=pred_dev)
    ctx.synchronize()
    ctx.delta_encode(pred_dev, frames, own_state["key"].astype(np.uint8), out=delta_dev)   # warm
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(5):
        ctx.delta_encode(pred_dev, frames, own_state["key"].astype(np.uint8), out=delta_dev)
    delta_ms, delta_n = ctx.prof_get()["delta"]
    ctx.prof_enable(False)
    del pred_dev, delta_dev
    delta_bytes = 7.0 * frames.shape[0] * H * W * 3  # f32 pred + u8 orig in, i16 out
    delta_gbs = delta_bytes * delta_n / (delta_ms * 1e-3) / 1e9 if delta_ms > 0 else 0.0

    # the elementwise tail the STEP launches: north_star's ">= 60 % of HBM for the elementwise delta
    # kernel" stated for what the timed step runs, not only for the stand-alone tz_delta_encode.  Algorithmic bytes as
    # DESIGN section 5 counts them: through the quantiser 5 in (pred f32 + orig u8) + 2 out (run values) / 2 in + 2 out (fill,
    # spatial delta, offset, histogram -> symbols) / 2 in + 2 out (rank remap) = 15 B per element; where the tolerance
    # cannot merge two different deltas (E <= 0.499: error_bound is then the identity, tz_quant_is_identity, round 6) or the
    # job is lossless, ONE fused pass 5 in + 2 out, then the remap 2 in + 2 out = 11 B per element.
    tail_keys = ("delta", "quant", "spatial_delta_hist", "lut_remap")
    tail_ms = sum(prof[k][0] for k in tail_keys if k in prof)
    tail_through_quantiser = prof.get("spatial_delta_hist", (0.0, 0))[1] > 0   # k_q_fill_sym / k_sdelta ran: not the one-pass form
    tail_bpe = 15.0 if tail_through_quantiser else (12.0 if prof.get("quant", (0.0, 0))[1] > 0 else 11.0)
    tail_bytes = tail_bpe * frames.shape[0] * H * W * 3
    tail_gbs = tail_bytes / (tail_ms * 1e-3) / 1e9 if tail_ms > 0 else 0.0
    # counter traffic of the same tail, per step, from the committed PMC passes (the kernels of the form that ran)
    tail_kernels = (("k_q_minmax", "k_q_tiles<", "k_q_bstitch<", "k_q_chain", "k_q_last", "k_q_fill_sym<", "k_lut") if tail_through_quantiser
                    else (("k_q_minmax",) if tail_bpe == 12.0 else ()) + ("k_delta_sd_fused<", "k_lut"))
    tail_parts = [measured_traffic(k) for k in tail_kernels]
    tail_traffic = sum(tail_parts) if tail_parts and all(v is not None for v in tail_parts) else None

    extras = {}
    ratio = None
    if args.no_extras and args.profile_legs and world == 1:
        own_step("abs", [2.0])
        own_step()
        decode_leg(job, ctx, frames, own_state, argparse.Namespace(steps=1))
    if not args.no_extras:
        # ------------------------------------------------------------ host -> host (SURVEY.md §8d wall-clock definition)
        if world == 1:
            hsteps = max(3, min(args.steps, 10))
            h2h = {}
            for kind in ("pinned", "pageable"):
                if kind == "pinned":
                    hf = _lib.pinned_copy(frames.cpu().numpy())
                    hp_ = _lib.pinned_empty(frames.shape[0] * H * W * 3, np.int16)
                else:
                    hf = frames.cpu().numpy()
                    hp_ = np.empty(frames.shape[0] * H * W * 3, np.int16)

                def hstep():
                    ctx.rollout(hf, WARM_UP, WINDOW)
                    ctx.encode(MODE, BOUND, True, payload=hp_)
                el = job.timed(hstep, hsteps, 1)
                h2h[kind + "_frames_per_s"] = frames.shape[0] * hsteps / el
                h2h[kind + "_ms_per_step"] = el / hsteps * 1e3
                same = bool((hp_ == state["payload"].cpu().numpy()).all())
                h2h[kind + "_payload_equals_device_resident"] = same
                del hf, hp_
            # the same pinned job with the payload hand-over DEFERRED (tz_set_payload_deferred): a caller that compresses
            # one sequence after the other alternates between two host buffers; the device -> host transfer of a payload
            # then runs under the next sequence's rollout.  Every payload is checked once it has been waited for.
            try:
                hf = _lib.pinned_copy(frames.cpu().numpy())
                hp2 = [_lib.pinned_empty(frames.shape[0] * H * W * 3, np.int16) for _ in range(2)]
                ref_payload = state["payload"].cpu().numpy()
                seq = {"k": 0, "ok": True}
                ctx.set_payload_deferred(True)

                dbg = [] if os.environ.get("TEZIP_BENCH_DEBUG") else None

                def pstep():
                    t0 = time.perf_counter()
                    ctx.rollout(hf, WARM_UP, WINDOW)
                    t1 = time.perf_counter()
                    ctx.payload_wait()                      # (free: the transfer had a whole rollout to finish)
                    t2 = time.perf_counter()
                    ctx.encode(MODE, BOUND, True, payload=hp2[seq["k"] & 1])
                    seq["k"] += 1
                    if dbg is not None:
                        dbg.append((t1 - t0, t2 - t1, time.perf_counter() - t2))

                el = job.timed(pstep, hsteps, 1, drain=ctx.payload_wait)
                ctx.payload_wait()
                if dbg:
                    print("pipelined step (rollout, payload_wait, encode) ms:", [tuple(round(v * 1e3, 2) for v in d) for d in dbg], file=sys.stderr)
                seq["ok"] = bool((hp2[0] == ref_payload).all() and (hp2[1] == ref_payload).all())
                h2h["pinned_pipelined_frames_per_s"] = frames.shape[0] * hsteps / el
                h2h["pinned_pipelined_ms_per_step"] = el / hsteps * 1e3
                h2h["pinned_pipelined_payloads_equal_device_resident"] = seq["ok"]
                h2h["pinned_pipelined_note"] = ("%d sequences back to back, payload hand-over deferred: the clock runs from the first "
                                                "stack in host memory to the LAST payload complete in host memory" % hsteps)
            except Exception as e:
                h2h["pinned_pipelined_error"] = repr(e)
            finally:
                ctx.set_payload_deferred(False)
                hf = hp2 = None
            h2h["steps"] = hsteps
            h2h["note"] = ("uint8 stack in host memory -> int16 payload + table in host memory, PCIe both ways inside the "
                           "timed region; pinned = tz_host_alloc buffers (key frames go first, the rest of the stack and the "
                           "payload chunks cross PCIe on a copy stream under the kernels), pageable = plain numpy arrays "
                           "(pipelined through the context's pinned staging buffers)")
            h2h["fraction_of_device_resident"] = h2h["pinned_frames_per_s"] / value
            extras["host_to_host"] = h2h

            # -------------------------------------------------------- the lossy path with real merges
            el = job.timed(lambda: own_step("abs", [2.0]), max(3, min(args.steps, 10)), 1)
            ctx.prof_enable(True)
            ctx.prof_reset()
            own_step("abs", [2.0])
            pq = ctx.prof_get()
            q = pq["quant"]
            ctx.prof_enable(False)
            extras["lossy_abs2"] = {"frames_per_s": frames.shape[0] * max(3, min(args.steps, 10)) / el,
                                    "quantiser_ms_per_step": q[0], "table_symbols": len(own_state["table"]),
                                    "encode_tail_ms_per_step": {k: pq[k][0] for k in ("delta", "quant", "spatial_delta_hist", "lut_remap")},
                                    "encode_tail_ms_total": sum(pq[k][0] for k in ("delta", "quant", "spatial_delta_hist", "lut_remap")),
                                    # the same roofline statement as roofline_encode_tail, for a tolerance that DOES merge: 15 B per
                                    # element through the general quantiser (speculative walks, issue-bound: DESIGN.md section 5)
                                    "encode_tail_bytes_per_element": 15.0,
                                    "encode_tail_frac_of_hbm": 15.0 * frames.shape[0] * H * W * 3 / max(1e-9, 1e-3 * sum(
                                        pq[k][0] for k in ("delta", "quant", "spatial_delta_hist", "lut_remap"))) / 1e9 / PEAK_HBM_GBS}
            own_step()

        # ------------------------------------------------------------ compression ratio (untimed; libzstd level 9 as the reference)
        if rank == 0:
            try:
                from tezip_amd import zstd
                host_payload = own_state["payload"].cpu().numpy()
                tb = own_state["table"]
                n_own = frames.shape[0]
                trailer = np.concatenate([tb.astype(np.int64), [len(tb)], [1, n_own, H, W, 3], [WARM_UP]]).astype(np.int16)
                ent = zstd.compress_array(np.concatenate([host_payload, trailer]), 9, zstd.default_threads())
                fr = frames.cpu().numpy()
                kf = np.zeros_like(fr)
                kf[own_state["key"]] = fr[own_state["key"]]
                keyb = zstd.compress_array(kf, 9, zstd.default_threads())
                ratio = fr.nbytes / float(len(ent) + len(keyb) + 8 * n_own)
            except Exception as e:  # ratio is informational
                print("ratio unavailable:", e, file=sys.stderr)

        # ------------------------------------------------------------ decode of the same job (decompress.py:87-256), device resident
        if world == 1:
            try:
                extras.update(decode_leg(job, ctx, frames, own_state, args))
            except Exception as e:
                extras["decode"] = {"error": repr(e)}

        # ------------------------------------------------------------ what a rank of an N > 1 job runs, on this one GPU
        if world == 1:
            try:
                extras["sharded_path"] = sharded_path_n1(job, ctx, engine, frames, own_state, value, args, dev)
            except Exception as e:
                extras["sharded_path"] = {"error": repr(e)}

        # ------------------------------------------------------------ the same step under the other arithmetic contract
        if world == 1 and contract == 2:
            try:
                ctx.set_contract(1)
                el = job.timed(own_step, max(3, min(args.steps, 10)), 1)
                ctx.prof_enable(True)
                ctx.prof_reset()
                own_step()
                p1 = ctx.prof_get()
                ctx.prof_enable(False)
                k16 = p1["conv16_lds_dma"][0]
                rate1 = frames.shape[0] * max(3, min(args.steps, 10)) / el
                extras["tz_pa1"] = {"frames_per_s": rate1, "ms_per_step": el / max(3, min(args.steps, 10)) * 1e3,
                                    "k_conv16_ms_per_step": k16, "k_conv16_frac_of_mfma_peak": c16_flops / (k16 * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                    "tz_pa2_over_tz_pa1": value / rate1,
                                    "note": "the step of rounds 1-3 (every convolution a direct fmaf chain, k_conv16), same job, same box"}
            except Exception as e:
                extras["tz_pa1"] = {"error": repr(e)}
            finally:
                ctx.set_contract(0)
                own_step()

        # ------------------------------------------------------------ the other BASELINE.json configurations (seconds each)
        if world == 1:
            try:
                extras["configs"] = configs_leg(job, ctx, cfg, frames, dev)
            except Exception as e:
                extras["configs"] = {"error": repr(e)}
            ctx.prepare(H, W, max_batch=nwin)
        try:   # configs[4] as a search, every N: one candidate window size per rank and turn
            f3 = turbulence_cuda(NT, 0, NT, H, W, 3, dev).cpu().numpy()
            extras["cfg5_sweep"] = cfg5_sweep(job, ctx, f3, world)
            del f3
        except Exception as e:
            extras["cfg5_sweep"] = {"error": repr(e)}
            if dist:
                raise
        ctx.prepare(H, W, max_batch=nwin)

        # ------------------------------------------------------------ the compression-ratio half of the metric, with a TRAINED model
        if world == 1 and not args.no_trained_ratio:
            try:
                extras["compression_ratio_trained"] = trained_ratio(ctx, cfg, frames, nwin)
            except Exception as e:
                extras["compression_ratio_trained"] = {"error": repr(e)}
            ctx.load_model(cfg, cfg.init_weights(seed=123))
            ctx.prepare(H, W, max_batch=nwin)

        # ------------------------------------------------------------ the other N>1 mode
        if world > 1:
            if sharded:
                rf = turbulence_cuda(nt_rank, 0, nt_rank, H, W, 3 + rank, dev)
                rp = torch.empty(nt_rank * H * W * 3, dtype=torch.int16, device=dev)

                def rstep():
                    ctx.rollout(rf, WARM_UP, WINDOW)
                    ctx.encode(MODE, BOUND, True, payload=rp)
                el = job.timed(rstep, args.steps, 1)
                extras["replicas"] = {"frames_per_s": nt_rank * world * args.steps / el,
                                      "note": "one 80-frame sequence per GPU, nothing exchanged (round 1's N>1 mode)"}
                del rf, rp

        # ------------------------------------------------------------ cfg4: ONE 1024x1024 sequence, windows sharded (strong scaling)
        try:
            extras["cfg4_sharded"] = cfg4_sharded(job, ctx, engine, cfg, rank, world, dev)
        except Exception as e:
            extras["cfg4_sharded"] = {"error": repr(e)}
            if dist:
                raise

    if rank == 0 and world == 1 and not args.no_extras and not args.no_host_pipeline:
        try:
            extras["host_pipeline"] = host_pipeline_leg(cfg, frames.cpu().numpy())
        except (Exception, SystemExit) as e:   # (the CLI functions end a refused job with exit())
            extras["host_pipeline"] = {"error": repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, frames[:20].cpu().numpy())
        tr = extras.get("compression_ratio_trained")
        if isinstance(tr, dict) and "_weights" in tr:   # checker leg: small slice, HIP stream vs oracle stream
            cpu["trained_ratio_check"] = ratio_check_vs_oracle(ctx, cfg, tr.pop("_weights"), frames)
            ctx.load_model(cfg, cfg.init_weights(seed=123))
            ctx.prepare(H, W, max_batch=nwin)
    if isinstance(extras.get("compression_ratio_trained"), dict):
        extras["compression_ratio_trained"].pop("_weights", None)

    if rank == 0:
        line = {
            "metric": "frames/sec (predict+delta-encode), 512x512 seq",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "value_definition": "device-resident: frames already in HBM when the clock starts, payload left in HBM, table on the "
                                "host.  The bench contract fixes this definition (inputs resident in HBM when the timed region "
                                "starts; a PCIe-inclusive rate is never `value`).  SURVEY.md 8d / BASELINE.md 4 define the "
                                "BASELINE metric host to host: THAT number is value_host_to_host (pinned buffers, PCIe both ways "
                                "inside the clock), measured in this same run -- quote it when comparing with the 50 frames/s target",
            "value_host_to_host": (extras.get("host_to_host") or {}).get("pinned_frames_per_s"),
            "value_host_to_host_pipelined": (extras.get("host_to_host") or {}).get("pinned_pipelined_frames_per_s"),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("512x512x3 synthetic turbulence stack, %d frames per GPU (ONE %d-frame sequence), SWP 20-frame "
                                    "windows, warm_up 0, lossy rel 1e-3, entropy remap on; PredNet (3,48,96,192) glorot seed 123"
                                    % (nt_rank, nt_total))
                       + (" [EXPLORATION: not the BASELINE.json sequence length]" if explore else ""),
                       "frames_per_step": nt_total if sharded else nt_rank * world,
                       "predicted_frames_per_step_rank0": n_pred,
                       "sharding": ("windows of one sequence over the ranks (dist.compress_sharded), payload gathered on rank 0"
                                    if sharded else "replicas: one sequence per GPU"),
                       "backend": backend, "ranks_share_one_gpu": single_dev if world > 1 else False},
            "sharded_check": check,
            "compression_ratio": ratio,
            "arithmetic_contract": "TZ-PA%d" % contract,
            "roofline": ({"kernel": "k_wino (TZ-PA2: every convolution of levels >= 1, the same-resolution source as Winograd F(2x2,3x3) "
                                    "chains on fp32 MFMA, staged by LDS-DMA; %.0f %% of the step's device time)"
                                    % (100.0 * wino_ms / max(step_dev_ms, 1e-9)),
                          "bound": "mfma", "achieved": wino_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                          "frac": wino_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": measured_traffic("k_wino<"),
                          "launches_per_step": wino_n, "ms_per_launch": wino_ms / max(wino_n, 1), "ms_per_step": wino_ms,
                          "algorithmic_flops_per_step": wino_flops, "algorithmic_flops_per_launch": wino_flops / max(wino_n, 1),
                          "flop_accounting": "EXECUTED: 16 multiplies per 2x2 outputs and input channel of a same-resolution source "
                                             "(the direct form has 36), 4 collapsed taps on an upsampled source",
                          "direct_equivalent": {"achieved": wino_direct_tflops, "frac": wino_direct_tflops / PEAK_FP32_MFMA_TFLOPS,
                                                "flops_per_step": c16_flops,
                                                "note": "the same launches priced with the FLOPs of the direct form (what k_conv16 "
                                                        "executes under TZ-PA1): a speed-up figure, not a utilisation"}}
                         if wino_n else
                         {"kernel": "k_conv16 (fp32 MFMA implicit GEMM staged by LDS-DMA: every convolution of levels >= 1, "
                                    "%.0f %% of the step's device time)" % (100.0 * c16_ms / max(step_dev_ms, 1e-9)),
                          "bound": "mfma", "achieved": c16_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                          "frac": c16_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": measured_traffic("k_conv16<"),
                          "launches_per_step": c16_n, "ms_per_launch": c16_ms / max(c16_n, 1), "ms_per_step": c16_ms,
                          "algorithmic_flops_per_step": c16_flops,
                          "algorithmic_flops_per_launch": c16_flops / max(c16_n, 1)}),
            "roofline_all_convolutions": {"kernel": "all PredNet convolutions of a step: k_wino / k_conv16 + k_conv16b + k_conv_small",
                                          "bound": "mfma",
                                          "achieved": (flops_step - c16_flops + wino_flops) / (conv_ms * 1e-3) / 1e12 if wino_n else conv_tflops,
                                          "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                          "frac": ((flops_step - c16_flops + wino_flops) / (conv_ms * 1e-3) / 1e12 if wino_n else conv_tflops) / PEAK_FP32_MFMA_TFLOPS,
                                          "direct_equivalent_tflops": conv_tflops,
                                          "traffic": measured_traffic("k_wino<" if wino_n else "k_conv"), "launches_per_step": conv_n,
                                          "ms_per_step": conv_ms,
                                          "algorithmic_flops_per_step": flops_step - c16_flops + wino_flops if wino_n else flops_step},
            "roofline_delta": {"kernel": "k_delta_flat (stand-alone tz_delta_encode on the step's prediction stack; the step itself forms "
                                         "its deltas inside the fused quantiser / spatial-delta kernels)", "bound": "hbm", "achieved": delta_gbs, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": delta_gbs / PEAK_HBM_GBS, "traffic": measured_traffic("k_delta_flat"),
                               "bytes_per_launch": delta_bytes, "ms_per_launch": delta_ms / max(delta_n, 1)},
            "roofline_encode_tail": {"kernel": "the elementwise tail the timed step launches behind the rollout: "
                                               + ("k_q_tiles + stitch kernels + k_q_fill_sym + k_lut (deltas formed inside the quantiser)"
                                                  if tail_through_quantiser else
                                                  "k_delta_sd_fused (delta + spatial delta + offset + histogram in one pass: the job's tolerance "
                                                  "cannot merge different deltas, so error_bound is the identity) + k_lut"),
                                     "bound": "hbm", "achieved": tail_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": tail_gbs / PEAK_HBM_GBS,
                                     "traffic": tail_traffic, "traffic_kernels": list(tail_kernels),
                                     "bytes_per_element": tail_bpe, "bytes_per_step": tail_bytes,
                                     "ms_per_step": tail_ms, "ms_per_step_by_stage": {k: prof[k][0] for k in tail_keys if k in prof},
                                     "share_of_step_device_time": tail_ms / max(step_dev_ms, 1e-9)},
            "kernel_ms_per_step": {k: v[0] for k, v in prof.items()},
            "cpu_baseline": cpu,
        }
        line.update(extras)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    ctx.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def decode_leg(job, ctx, frames, own_state, args):
    """decompress.run's device work on the job just encoded: key discovery + rollout replay
    (decompress.py:123-186), inverse remap + inverse spatial delta (203-245, one fused single-pass scan),
    reconstruct (252-256); key stack, payload and decoded frames resident in HBM."""
    dev = frames.device
    key = own_state["key"]
    kidx = torch.from_numpy(key).to(dev)
    keys = torch.zeros_like(frames)
    keys[kidx] = frames[kidx]
    out = torch.empty_like(frames)
    payload, table = own_state["payload"], own_state["table"]
    torch.cuda.synchronize()

    def dstep():
        ctx.rollout_decode(keys, WARM_UP)
        ctx.decode(payload, table, out=out)
    steps = max(3, min(args.steps, 10))
    el = job.timed(dstep, steps, 1)
    ctx.prof_enable(True)
    ctx.prof_reset()
    dstep()
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    torch.cuda.synchronize()
    n = frames.numel()
    scan_ms, scan_n = prof["undelta_scan"]
    rec_ms, rec_n = prof["reconstruct"]
    exact = bool(torch.equal(out, frames))
    # the decoder's tail is one launch (k_scan2p<LUT, RECON>: inverse remap + inverse spatial delta + reconstruct) when the
    # frames are unpadded multiples of 16 elements -- then 2 (payload) + 4 (prediction) + 1 (frame out) algorithmic bytes
    # per element; with separate launches the scan alone is 2 + 2
    fused = rec_n == 0
    per_el = 7.0 if fused else 4.0
    scan_gbs = per_el * n / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    dec = {"frames_per_s": frames.shape[0] * steps / el, "ms_per_step": el / steps * 1e3, "steps": steps,
           "round_trip": "bit-exact (rel 1e-3 merges nothing at these amplitudes)" if exact else "within the bound",
           "kernel_ms_per_step": {k: v[0] for k, v in prof.items() if v[1]}}
    if not fused:
        dec["reconstruct_GBps"] = 7.0 * n / (rec_ms * 1e-3) / 1e9 if rec_ms > 0 else 0.0
    return {
        "decode": dec,
        "roofline_undelta": {"kernel": ("k_scan2p<LUT, RECON> (the decoder's tail as ONE launch of resident blocks: inverse rank remap, chunk "
                                        "sums, prefix scan mod 2^16, reconstruct + clip; decompress.py:22-36,203-256)") if fused else
                                       ("k_scan2p<LUT> (inverse rank remap + inverse spatial delta as ONE launch of resident blocks: chunk "
                                        "sums, then a prefix scan mod 2^16; decompress.py:22-36,203-245)"),
                             "bound": "hbm", "achieved": scan_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": scan_gbs / PEAK_HBM_GBS, "traffic": measured_traffic("k_scan2p"),
                             "bytes_per_launch": per_el * n, "ms_per_launch": scan_ms / max(scan_n, 1)},
    }


def sharded_path_n1(job, ctx, engine, frames, own_state, fused_value, args, dev):
    """The per-rank code path of an N > 1 run (dist.compress_sharded: tz_rollout, tz_encode_begin, the
    all_gather of carries / key masks, the histogram all-reduce, tz_encode_finish, the point-to-point
    gather), driven on THIS one GPU through a one-rank RCCL process group, so that its cost beside the
    fused tz_rollout + tz_encode step is a measured number before the first multi-GPU run."""
    import torch.distributed as dist
    from tezip_amd import dist as tzdist
    if dist.is_initialized():
        return {"skipped": "a process group already exists"}
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("TEZIP_BENCH_BACKEND", "nccl")
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1, **kw)
    try:
        nt = frames.shape[0]
        state = {}

        def drain():
            p = state.pop("pending", None)
            if p is not None:
                state["payload"], state["table"], state["key"] = p.wait()

        def step():
            nxt = tzdist.compress_sharded(engine, lambda a, b: frames, WARM_UP, WINDOW, MODE, BOUND, True, nt=nt,
                                          to_host=False, wait=False)
            drain()
            state["pending"] = nxt
        steps = max(3, min(args.steps, 10))
        el = job.timed(step, steps, 1, drain=drain)
        pl = state["payload"]
        if isinstance(pl, np.ndarray):
            pl = torch.from_numpy(pl).to(dev)
        same = bool(torch.equal(pl, own_state["payload"])) and bool((state["table"] == own_state["table"]).all()) \
            and bool((state["key"] == own_state["key"]).all())
        rate = nt * steps / el
        res = {"frames_per_s": rate, "ms_per_step": el / steps * 1e3, "steps": steps, "backend": backend,
               "fraction_of_fused": rate / fused_value,
               "check": "byte-identical to tz_rollout + tz_encode" if same else "MISMATCH vs tz_rollout + tz_encode",
               "note": "one-rank process group on this GPU: every collective of the N > 1 protocol is issued"}
        # the point-to-point gather itself: with one rank its op list is empty, so the payload shard (126 MB, int16 seen
        # as uint8 device views) is sent by rank 0 to ITSELF through the same grouped isend / irecv that peers use,
        # wait=False overlap included -- the only way the RCCL send/recv path can run on a one-GPU box
        try:
            state.clear()

            def step_p2p():
                nxt = tzdist.compress_sharded(engine, lambda a, b: frames, WARM_UP, WINDOW, MODE, BOUND, True, nt=nt,
                                              to_host=False, wait=False, self_p2p=True)
                drain()
                state["pending"] = nxt
            el2 = job.timed(step_p2p, steps, 1, drain=drain)
            pl = state["payload"]
            if isinstance(pl, np.ndarray):
                pl = torch.from_numpy(pl).to(dev)
            same2 = bool(torch.equal(pl, own_state["payload"])) and bool((state["table"] == own_state["table"]).all())
            # the transfer alone: one grouped self send/recv of the payload bytes
            src = own_state["payload"].view(torch.uint8)
            dst = torch.empty_like(src)
            torch.cuda.synchronize()
            for _ in range(2):
                for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, 0), dist.P2POp(dist.isend, src, 0)]):
                    q.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst, 0), dist.P2POp(dist.isend, src, 0)]):
                    q.wait()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            res["self_p2p"] = {"frames_per_s": nt * steps / el2, "fraction_of_fused": nt * steps / el2 / fused_value,
                               "check": "byte-identical to tz_rollout + tz_encode" if same2 else "MISMATCH",
                               "transfer_GBps": src.numel() / dt / 1e9, "transfer_ms": dt * 1e3, "bytes": int(src.numel()),
                               "transfer_check": "byte-identical" if bool(torch.equal(src, dst)) else "MISMATCH",
                               "note": "rank 0's own shard through dist.batch_isend_irecv to itself (grouped RCCL send/recv, "
                                       "device uint8 views of the int16 payload) instead of the device copy"}
        except Exception as e:   # RCCL may refuse a self send/recv: recorded, not fatal
            res["self_p2p"] = {"error": repr(e)[:500]}
        return res
    finally:
        dist.destroy_process_group()


def configs_leg(job, ctx, cfg, frames3, dev):
    """Driver-run numbers for the BASELINE.json configurations the headline does not cover, each a few seconds, device
    resident (frames in HBM, payload into HBM, table on the host), every result decoded again and checked:
      cfg1  64x64 moving blobs, 40 frames, -w 20, lossless          (configs[0])
      cfg2  128x160 KITTI-like, 40 frames, -w 10, lossless          (configs[1])
      cfg5_dwp  512x512 turbulence, 80 frames, DWP (-t inside the observed window-MSE range), lossless   (configs[4])
      cfg4_dwp  1024x1024 detector frames, 80 frames, DWP (one window at a time, B = 1 steps), lossless  (configs[3]'s data under -t)"""
    from tezip_amd import synth
    out = {}

    def run(name, frames, window, thr, max_batch, steps):
        nt, h, w = frames.shape[:3]
        hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
        ctx.prepare(hp, wp, max_batch)
        payload = torch.empty(nt * h * w * 3, dtype=torch.int16, device=dev)
        st = {}

        def step():
            st["key"], _ = ctx.rollout(frames, 0, window, thr)
            _, st["table"], _ = ctx.encode("abs", [0.0], True, payload=payload)
        el = job.timed(step, steps, 2)
        key = st["key"]
        kidx = torch.from_numpy(key).to(dev)
        keys = torch.zeros_like(frames)
        keys[kidx] = frames[kidx]
        dec = torch.empty_like(frames)
        torch.cuda.synchronize()

        def dstep():
            ctx.rollout_decode(keys, 0)
            ctx.decode(payload, st["table"], out=dec)
        eld = job.timed(dstep, steps, 1)
        lens = np.diff(np.concatenate([key.nonzero()[0], [nt]])).tolist()
        out[name] = {"frames": nt, "size": "%dx%d" % (h, w), "key_frames": int(key.sum()), "window_lengths": lens,
                     "frames_per_s": nt * steps / el, "ms_per_step": el / steps * 1e3,
                     "decode_frames_per_s": nt * steps / eld, "steps": steps,
                     "round_trip": "bit-exact" if bool(torch.equal(dec, frames)) else "MISMATCH"}

    run("cfg1", torch.from_numpy(synth.moving_blobs(40, 64, 64)).to(dev), 20, None, 2, 20)
    run("cfg2", torch.from_numpy(synth.translating_scene(40, 128, 160)).to(dev), 10, None, 4, 20)
    ctx.prepare(H, W, 1)
    _, mse = ctx.rollout(frames3[:41], 0, None, 1e9, want_mse=True)
    thr = float(mse[16])           # the window MSE of compress.py:246 reaches it after about 16 frames
    run("cfg5_dwp", frames3, None, thr, 1, 3)
    out["cfg5_dwp"]["threshold"] = thr
    # cfg4's frames under -t: B = 1 steps at 1024x1024 (the "E-part ahead" decision outside 512x512)
    f4 = detector_cuda(0, 80, 1024, 1024, 4, dev)
    ctx.prepare(1024, 1024, 1)
    _, mse4 = ctx.rollout(f4[:41], 0, None, 1e9, want_mse=True)
    thr4 = float(mse4[16])
    run("cfg4_dwp", f4, None, thr4, 1, 2)
    out["cfg4_dwp"]["threshold"] = thr4
    del f4
    return out


def cfg5_sweep(job, ctx, frames3_host, world):
    """BASELINE.json configs[4] read as a search: the SWP sweep -w in {5, 10, ..., 40} of tezip_amd/sweep.py (`tezip.py
    --sweep`), one candidate per rank and turn under N > 1 (sweep.sweep_sharded: nothing exchanged but the sizes), all
    eight on this GPU at N = 1; lossless; each candidate is rollout + encode + the two zstd-9 frames, as `-w <value>`
    would write them (the sizes ARE the result of the search, so zstd is inside the timed region here)."""
    from tezip_amd import sweep
    st = {}

    def step():
        st["rows"], st["best"], _ = sweep.sweep_sharded(ctx, frames3_host, 0, sweep.DEFAULT_WINDOWS, "abs", [0.0])
    el = job.timed(step, 1, 0)
    nt = frames3_host.shape[0]
    raw = float(frames3_host.nbytes)
    return {"workload": "512x512x3 turbulence, %d frames, lossless, SWP candidates -w %s, %d rank(s)"
                        % (nt, list(sweep.DEFAULT_WINDOWS), world),
            "seconds": el, "candidates_per_s": len(sweep.DEFAULT_WINDOWS) / el,
            "frames_per_s": nt * len(sweep.DEFAULT_WINDOWS) / el, "best_window": int(st["best"]),
            "ratio_by_window": {str(r["window"]): raw / r["total_bytes"] for r in st["rows"]},
            "note": "random glorot weights: the ratios only order the candidates"}


def _ratio_of(ctx, frames, mode, bound):
    """compress.run's three files for the resident job: raw bytes / (entropy.dat + key_frame.dat + filename.txt)."""
    from tezip_amd import compress, zstd
    key, _ = ctx.rollout(frames, WARM_UP, WINDOW)
    payload, table, _ = ctx.encode(mode, bound, True)
    n_own = frames.shape[0]
    stream = compress.build_stream(np.asarray(payload), table, (1, n_own, H, W, 3), WARM_UP)
    ent = zstd.compress_array(stream, 9, zstd.default_threads())
    fr = frames.cpu().numpy()
    kf = np.zeros_like(fr)
    kf[key] = fr[key]
    keyb = zstd.compress_array(kf, 9, zstd.default_threads())
    names = 2 + n_own * len("frame_0000.png\n")
    return fr.nbytes / float(len(ent) + len(keyb) + names), len(table)


def trained_ratio(ctx, cfg, frames, nwin):
    """BASELINE.json's metric names a compression ratio; the reference ships no weights, and glorot
    weights predict a constant.  Train the reference's model with the reference's schedule
    (tezip_amd/train.py = train.py:43-112 on PyTorch autograd, ~12 s on this GPU) on HELD-OUT synthetic
    turbulence (other seeds, 128x128), then compress the cfg3 job with it: lossless and `abs 2`."""
    import tempfile
    from tezip_amd import synth, train, weights
    t0 = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="tz_bench_model_")
    data = os.path.join(tmp, "set")
    os.makedirs(data)
    seqs = [synth.turbulence(12, 128, 128, seed=100 + k) for k in range(10)]
    np.save(os.path.join(data, "X_train.npy"), np.concatenate(seqs[:9]))
    np.save(os.path.join(data, "sources_train.npy"), np.repeat(["train-%d" % k for k in range(9)], 12))
    np.save(os.path.join(data, "X_val.npy"), seqs[9])
    np.save(os.path.join(data, "sources_val.npy"), np.repeat(["val-9"], 12))
    train.run(os.path.join(tmp, "model"), data, False)
    wts = weights.load_model(os.path.join(tmp, "model"))[1]
    train_s = time.perf_counter() - t0
    ctx.load_model(cfg, wts)
    ctx.prepare(H, W, max_batch=nwin)
    out = {"model": "PredNet (3,48,96,192) trained here: 100 epochs x 5 samples, Adam 1e-3 -> 1e-4 (train.py:43-112), "
                    "10 held-out 128x128 turbulence sequences (seeds 100-109; the job is seed 3 at 512x512)",
           "train_seconds": train_s, "_weights": wts}
    for name, mode, bound in (("lossless", "abs", [0.0]), ("abs2", "abs", [2.0]), ("rel_1e-3", MODE, BOUND)):
        r, t = _ratio_of(ctx, frames, mode, bound)
        out[name] = {"ratio": r, "table_symbols": t}
    return out


def ratio_check_vs_oracle(ctx, cfg, wts, frames):
    """Checker (part of the cpu_baseline leg, the only place bench.py may use oracle/): on a small slice of
    the job -- 6 frames, 64x64 crop, 3-frame windows -- the HIP path with the TRAINED model must produce
    the oracle's pre-zstd streams byte for byte for lossless and `abs 2`; identical streams through the
    same libzstd are identical files, hence equal ratios."""
    from oracle import coracle
    from oracle import oracle as O
    from tezip_amd import compress
    sl = np.ascontiguousarray(frames[:6, 100:164, 200:264].cpu().numpy())
    ctx.load_model(cfg, wts)
    ctx.prepare(64, 64, max_batch=2)

    class P:
        net = coracle.CPredNet(wts, cfg.stack_sizes, cfg.R_stack_sizes, 64, 64)

        def c0(self, a, b):
            return self.net.c0()

        def next(self, f):
            return self.net.next(np.asarray(f, np.float32))
    res = {}
    for name, bound in (("lossless", [0.0]), ("abs2", [2.0])):
        ref = O.compress_oracle(sl, 0, 3, None, "abs", bound, P(), True)
        key, _ = ctx.rollout(sl, 0, 3)
        payload, table, _ = ctx.encode("abs", bound, True)
        stream = compress.build_stream(np.asarray(payload), table, (1, 6, 64, 64, 3), 0)
        res[name] = bool(stream.tobytes() == ref["stream"].tobytes() and (key == ref["key"]).all())
    return {"streams_byte_identical_to_oracle": res, "slice": "6 frames, 64x64 crop, 3-frame windows, trained model"}


def cfg4_sharded(job, ctx, engine, cfg, rank, world, dev):
    """BASELINE.json configs[3]: ONE 320-frame 1024x1024 detector sequence, 40-frame windows, lossy
    `abs 2`; the 8 windows are sharded over the ranks (strong scaling: total work fixed)."""
    from tezip_amd import dist as tzdist
    h = w = 1024
    nt, window, mode, bound = 320, 40, "abs", [2.0]
    shards = tzdist.plan_shards(nt, 0, window, world) if world > 1 else [(0, nt)]
    f0, f1 = shards[rank]
    frames = detector_cuda(f0, f1, h, w, 4, dev) if f1 > f0 else None
    ctx.prepare(h, w, max_batch=max(1, (f1 - f0 + window - 1) // window))
    state = {"payload": torch.empty(nt * h * w * 3, dtype=torch.int16, device=dev) if world == 1 else None}

    def fetch(a, b):
        return frames

    step = sharded_step_fn(job, ctx, engine, fetch, nt, 0, window, mode, bound, state)
    steps = 2
    el = job.timed(step, steps, 1, drain=getattr(step, "drain", None))
    out = {"workload": "ONE 1024x1024x1(->3) detector sequence, 320 frames, SWP 40-frame windows, lossy abs 2, "
                       "windows sharded over %d rank(s)" % world,
           "frames_per_s": nt * steps / el, "ms_per_step": el / steps * 1e3, "steps": steps, "scaling": "strong"}
    if world > 1 and rank == 0:
        full = detector_cuda(0, nt, h, w, 4, dev)
        out["check"] = verify_against_one_gpu(ctx, full, 0, window, mode, bound, state, nt // window, h, w)
    if job.dist:
        job.dist.barrier()
    return out


def host_pipeline_leg(cfg, frames_host):
    """"Exclude PNG decode and zstd from the GPU number, report them separately."  The host side of the
    CLI on the cfg3 job, end to end and warm: the stack written as 80 PNG files, `tezip_amd.compress.run` (replaces
    compress.py:97-122 image load, 183-373, 375-400 trailer + zstd-9) and `tezip_amd.decompress.run` (decompress.py:87-103,
    119-256, 266-279 PNG save) each run three times (median wall seconds), then once more with the stage log on
    (a device synchronise at every stage mark, so the stages add up to a little more than the untimed-stage wall).  Untimed
    for `value`; rank 0, N = 1 only.  The contexts these runs make live beside the bench's own on the same GPU."""
    import contextlib
    import io
    import shutil
    import tempfile
    from PIL import Image
    from tezip_amd import compress, decompress, weights, zstd
    tmp = tempfile.mkdtemp(prefix="tzbench_host_")
    try:
        mdir, ddir, cdir, udir = (os.path.join(tmp, d) for d in ("model", "png", "tz", "out"))
        weights.save_model(mdir, cfg, cfg.init_weights(seed=123), H, W)
        os.mkdir(ddir)
        t0 = time.perf_counter()
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=compress.io_threads()) as pool:   # (PIL's encoder releases the GIL)
            list(pool.map(lambda t: Image.fromarray(frames_host[t]).save(os.path.join(ddir, "f%04d.png" % t)), range(frames_host.shape[0])))
        png_write_s = time.perf_counter() - t0
        png_bytes = sum(os.path.getsize(os.path.join(ddir, f)) for f in os.listdir(ddir))
        nt = frames_host.shape[0]
        quiet = io.StringIO()

        def crun():
            shutil.rmtree(cdir, ignore_errors=True)
            with contextlib.redirect_stdout(quiet):
                compress.run(mdir, ddir, cdir, WARM_UP, WINDOW, None, MODE, BOUND, True, False, True)

        def urun():
            shutil.rmtree(udir, ignore_errors=True)
            with contextlib.redirect_stdout(quiet):
                decompress.run(mdir, cdir, udir, True, False)

        def wall(fn, reps=3):
            fn()                                   # page cache, PIL codecs, libzstd
            ts = []
            for _ in range(reps):
                t1 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t1)
            return sorted(ts)

        def staged(fn, which):
            compress.STAGE_LOG = []
            try:
                t1 = time.perf_counter()
                fn()
                total = time.perf_counter() - t1
                log = [(name, sec) for run, name, sec in compress.STAGE_LOG if run == which]
            finally:
                compress.STAGE_LOG = None
            return total, log

        cw = wall(crun)
        ctot, clog = staged(crun, "compress")
        uw = wall(urun)
        utot, ulog = staged(urun, "decompress")
        files = {f: os.path.getsize(os.path.join(cdir, f)) for f in sorted(os.listdir(cdir))}
        back = np.stack([np.asarray(Image.open(os.path.join(udir, "f%04d.png" % t)).convert("RGB")) for t in range(nt)])
        err = int(np.abs(back.astype(np.int16) - frames_host.astype(np.int16)).max())
        cs = dict(clog)
        z_ent = cs.get("payload fetch + zstd-9 entropy.dat", 0.0)
        z_stage = cs.get("key_frame.dat + entropy.dat", 0.0)
        out = {
            "what": "tezip.py -c / -u on the cfg3 job through the CLI's own functions: %d PNG files of %dx%dx3 (%.1f MB) -> "
                    "filename.txt + key_frame.dat + entropy.dat (+ tezip_amd.json) -> %d PNG files; warm, median of 3; "
                    "NOT part of `value`" % (nt, H, W, png_bytes / 1e6, nt),
            "replaces": "compress.py:97-122 (image load), 375-400 (trailer + zstd-9); decompress.py:87-103 (zstd-d), 266-279 (PNG save)",
            "compress_run_s": cw[len(cw) // 2], "compress_run_s_min_max": [cw[0], cw[-1]],
            "compress_frames_per_s": nt / cw[len(cw) // 2],
            "decompress_run_s": uw[len(uw) // 2], "decompress_run_s_min_max": [uw[0], uw[-1]],
            "decompress_frames_per_s": nt / uw[len(uw) // 2],
            "compress_stages_s": {name: round(sec, 4) for name, sec in clog},
            "compress_stages_total_s": ctot,
            "decompress_stages_s": {name: round(sec, 4) for name, sec in ulog},
            "decompress_stages_total_s": utot,
            "stage_note": "stage runs synchronise the device at every mark; '(worker)' / 'payload fetch + zstd-9 entropy.dat' "
                          "run INSIDE the 'key_frame.dat + entropy.dat' stage, side by side",
            "zstd_level": 9, "zstd_threads": zstd.default_threads(), "png_threads": compress.io_threads(), "host_cores": host_cores(),
            "zstd9_share_of_compress_run": z_stage / ctot if ctot else None,
            "zstd9_entropy_share_of_compress_run": z_ent / ctot if ctot else None,
            "output_bytes": files,
            "round_trip_max_abs_error": err,
            "png_write_of_the_input_s": png_write_s,
        }
        if ctot and z_stage / ctot > 0.6:
            out["bound_by"] = ("zstd level 9 (%.0f %% of compress.run): the level is the reference's (compress.py:377,398) and is "
                               "kept for ratio parity" % (100.0 * z_stage / ctot))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, frames):
    """The oracle (a C/OpenMP port of the same path; the reference's Keras predictor cannot
    run here) on a bounded sample: one 512x512 window of 1 key + k predicted frames,
    predict + delta, sized to ~20-25 s of CPU work; every frame is timed on its own so that the
    line carries the spread (a shared box's host cores are noisy: rounds 3-4 saw 0.59 .. 0.79)."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library loads
    from oracle import coracle
    coracle.build()
    net = coracle.CPredNet(cfg.init_weights(seed=123), cfg.stack_sizes, cfg.R_stack_sizes, H, W)
    net.c0()  # t0 constants, untimed like tz_model_prepare
    per = []
    t0 = time.perf_counter()
    cur = coracle.u8_to_f32_frame(frames[0], H, W)
    budget, i = 22.0, 0
    while i < len(frames) - 1 and i < 19 and (i < 2 or (time.perf_counter() - t0) + per[-1] < budget):
        t1 = time.perf_counter()
        cur = net.next(cur)
        coracle.delta_frame(cur, frames[1 + i])
        per.append(time.perf_counter() - t1)
        i += 1
    total = time.perf_counter() - t0
    n = len(per)
    rates = sorted(1.0 / t for t in per)
    return {"value": n / total, "unit": "frames/s", "cores": cores, "kind": "port",
            "per_frame_frames_per_s": {"min": rates[0], "median": rates[n // 2], "max": rates[-1]},
            "sample": "%d predicted 512x512x3 frames of the same stack (one window: PredNet live work + delta each), "
                      "C oracle with OpenMP on the %d host cores of this job; %.1f s of CPU work" % (n, cores, total)}


if __name__ == "__main__":
    main()
