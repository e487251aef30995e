#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

metric : frames/sec (predict + delta-encode, whole job over all ranks), 512x512x3 sequences
step   : one pass of the hot path over one synthetic sequence that is already resident in HBM:
         tz_rollout (PredNet rollout, SWP) + tz_encode (delta, error-bound quantise, spatial
         delta, histogram, rank table, remap) -> int16 payload in HBM + rank table on host.
workload (BASELINE.json configs[2]): 512x512x3 synthetic turbulence stack, nt=80, 20-frame
         windows, lossy `rel 1e-3`, synthetic glorot weights (seed 123) of the reference model.
N>1    : one process per GPU (torch.distributed/RCCL only for the barrier and the max over
         ranks); every rank compresses its own sequence (windows shard with no data-path
         collective) => weak scaling.

Extra objects on the JSON line: "roofline" (dominant kernel = MFMA convolution),
"roofline_delta" (the HBM-bound delta kernel named by the north star), "cpu_baseline"
(the C oracle = a port of the same path, timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H = W = 512
NT = 80
WINDOW = 20
WARM_UP = 0
MODE, BOUND = "rel", [1e-3]
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0


def turbulence_cuda(nt, h, w, seed, device):
    """Same construction as tezip_amd.synth.turbulence, evaluated on the GPU."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                            torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    ts = torch.arange(nt, device=device, dtype=torch.float32)
    out = torch.zeros((nt, h, w, 3), device=device, dtype=torch.float32)
    for c in range(3):
        for o in range(6):
            r = torch.rand(4, generator=g).tolist()
            f = (2.0 ** o) * 2 * np.pi / max(h, w)
            th, ph = r[0] * 2 * np.pi, r[1] * 2 * np.pi
            vx, vy = r[2] * 3 - 1.5, r[3] * 3 - 1.5
            kx, ky = f * np.cos(th), f * np.sin(th)
            amp = 1.0 / (1.5 ** o)
            base = kx * xx + ky * yy + ph
            om = (kx * vx + ky * vy) * ts
            out[..., c] += amp * torch.sin(base[None] - om[:, None, None])
    out = (out - out.min()) / (out.max() - out.min()) * 255
    return out.round().clamp(0, 255).to(torch.uint8).contiguous()


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


def conv16_flops_per_px0(cfg):
    """The part of live_flops_per_px0 that k_conv16 executes: gate convolutions and A convolutions
    whose sources all have a multiple of 16 channels (levels >= 1 of the default model)."""
    st, rs, L = cfg.stack_sizes, cfg.R_stack_sizes, cfg.nb_layers
    mac = 0.0
    for l in range(L):
        gate_srcs_ok = (2 * st[l]) % 16 == 0 and (l == L - 1 or rs[l + 1] % 16 == 0) and rs[l] % 16 == 0
        if gate_srcs_ok:
            mac += (9 * 2 * st[l] + (4 * rs[l + 1] if l < L - 1 else 0)) * 4 * rs[l] / 4 ** l
        if l < L - 1 and (2 * st[l]) % 16 == 0 and st[l + 1] % 48 == 0:
            mac += 9 * 2 * st[l] * st[l + 1] / 4 ** l
    return 2 * mac


def main():
    global NT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames", type=int, default=NT,
                    help="frames per sequence (default 80 = the BASELINE.json configuration; other values are "
                         "exploration only and are labelled as such in config.workload)")
    args = ap.parse_args()
    explore = args.frames != NT
    NT = args.frames

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback)")
    if os.environ.get("TEZIP_BENCH_SINGLE_DEVICE"):  # rehearsal of N>1 on a one-GPU box
        local = 0
    torch.cuda.set_device(local)
    dist = None
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
    from tezip_amd.prednet import PredNetConfig

    dev = torch.device("cuda", local)
    ctx = _lib.Context(local, stream=torch.cuda.current_stream().cuda_stream)
    cfg = PredNetConfig()
    ctx.load_model(cfg, cfg.init_weights(seed=123))
    nwin = (NT - WARM_UP + WINDOW - 1) // WINDOW
    ctx.prepare(H, W, max_batch=nwin)
    frames = turbulence_cuda(NT, H, W, 3 + rank, dev)
    payload = torch.empty(NT * H * W * 3, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()

    state = {}

    def step():
        key, _ = ctx.rollout(frames, WARM_UP, WINDOW)
        _, table, _ = ctx.encode(MODE, BOUND, True, payload=payload)
        state["key"], state["table"] = key, table

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel device time (HIP events on the launch stream), one extra untimed step
    ctx.prof_enable(True)
    ctx.prof_reset()
    step()
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    n_pred = NT - int(state["key"].sum())
    conv_ms, conv_n = prof["conv3x3_mfma"]
    flops_step = live_flops_per_px0(cfg) * H * W * n_pred
    conv_tflops = flops_step / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    SUB = ("conv16_lds_dma", "conv16b_level0", "conv_small_valu", "conv3x3_general")  # sub-classes of conv3x3_mfma
    step_dev_ms = sum(v[0] for k, v in prof.items() if k not in SUB)
    c16_ms, c16_n = prof["conv16_lds_dma"]               # the dominant kernel on its own
    c16_flops = conv16_flops_per_px0(cfg) * H * W * n_pred
    c16_tflops = c16_flops / (c16_ms * 1e-3) / 1e12 if c16_ms > 0 else 0.0
    delta_ms, delta_n = prof["delta"]
    delta_bytes = 7.0 * NT * H * W * 3  # f32 pred + u8 orig in, i16 out
    delta_gbs = delta_bytes / (delta_ms * 1e-3) / 1e9 if delta_ms > 0 else 0.0

    # ---- host-buffer (PCIe-inclusive) rate of the same step: informational, never `value`
    host_frames = frames.cpu().numpy()
    host_payload = np.empty(NT * H * W * 3, np.int16)
    t1 = time.perf_counter()
    ctx.rollout(host_frames, WARM_UP, WINDOW)
    ctx.encode(MODE, BOUND, True, payload=host_payload)
    pcie_fps = NT / (time.perf_counter() - t1)

    # ---- compression ratio (untimed; same libzstd for both files, level 9 as the reference)
    ratio = None
    if rank == 0:
        try:
            from tezip_amd import zstd
            host_payload = payload.cpu().numpy()
            tb = state["table"]
            trailer = np.concatenate([tb.astype(np.int64), [len(tb)], [1, NT, H, W, 3], [WARM_UP]]).astype(np.int16)
            ent = zstd.compress_array(np.concatenate([host_payload, trailer]), 9)
            fr = frames.cpu().numpy()
            kf = np.zeros_like(fr)
            kf[state["key"]] = fr[state["key"]]
            keyb = zstd.compress_array(kf, 9)
            ratio = fr.nbytes / float(len(ent) + len(keyb) + 8 * NT)
        except Exception as e:  # ratio is informational
            ratio = None
            print("ratio unavailable:", e, file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, frames[:8].cpu().numpy())

    if rank == 0:
        total_frames = NT * args.steps * world
        line = {
            "metric": "frames/sec (predict+delta-encode), 512x512 seq",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("512x512x3 synthetic turbulence stack, nt=%d, SWP 20-frame windows, warm_up 0, "
                                    "lossy rel 1e-3, entropy remap on; PredNet (3,48,96,192) glorot seed 123" % NT)
                       + (" [EXPLORATION: not the BASELINE.json sequence length]" if explore else ""),
                       "frames_per_step": NT, "predicted_frames_per_step": n_pred, "sharding": "one sequence per GPU"},
            "compression_ratio": ratio,
            "pcie_inclusive_frames_per_s_rank0": pcie_fps,
            "roofline": {"kernel": "k_conv16 (fp32 MFMA implicit GEMM staged by LDS-DMA: every convolution of levels >= 1, "
                                   "%.0f %% of the step's device time)" % (100.0 * c16_ms / max(step_dev_ms, 1e-9)),
                         "bound": "mfma", "achieved": c16_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": c16_tflops / PEAK_FP32_MFMA_TFLOPS, "traffic": measured_traffic("k_conv16<"),
                         "launches_per_step": c16_n, "ms_per_launch": c16_ms / max(c16_n, 1), "ms_per_step": c16_ms,
                         "algorithmic_flops_per_step": c16_flops,
                         "algorithmic_flops_per_launch": c16_flops / max(c16_n, 1)},
            "roofline_all_convolutions": {"kernel": "k_conv16 + k_conv16b + k_conv_small (+ k_conv3x3): all PredNet convolutions",
                                          "bound": "mfma", "achieved": conv_tflops, "peak": PEAK_FP32_MFMA_TFLOPS,
                                          "unit": "TFLOP/s", "frac": conv_tflops / PEAK_FP32_MFMA_TFLOPS,
                                          "traffic": measured_traffic("k_conv"), "launches_per_step": conv_n,
                                          "ms_per_step": conv_ms, "algorithmic_flops_per_step": flops_step},
            "roofline_delta": {"kernel": "k_delta_flat", "bound": "hbm", "achieved": delta_gbs, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": delta_gbs / PEAK_HBM_GBS, "traffic": measured_traffic("k_delta_flat"),
                               "bytes_per_launch": delta_bytes, "ms_per_launch": delta_ms / max(delta_n, 1)},
            "kernel_ms_per_step": {k: v[0] for k, v in prof.items()},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    ctx.close()
    if dist:
        dist.destroy_process_group()


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


def cpu_baseline(cfg, frames8):
    """The oracle (a C/OpenMP port of the same path; the reference's Keras predictor cannot
    run here) on a bounded sample: one 512x512 window of 1 key + k predicted frames,
    predict + delta, sized to ~15-25 s of CPU work."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle library loads
    from oracle import coracle
    coracle.build()
    net = coracle.CPredNet(cfg.init_weights(seed=123), cfg.stack_sizes, cfg.R_stack_sizes, H, W)
    net.c0()  # t0 constants, untimed like tz_model_prepare
    t0 = time.perf_counter()
    cur = coracle.u8_to_f32_frame(frames8[0], H, W)
    cur = net.next(cur)
    coracle.delta_frame(cur, frames8[1])
    one = time.perf_counter() - t0
    k = int(max(1, min(6, round(18.0 / max(one, 1e-3)) - 1)))
    for i in range(k):
        cur = net.next(cur)
        coracle.delta_frame(cur, frames8[2 + i])
    total = time.perf_counter() - t0
    n = 1 + k
    return {"value": n / total, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d predicted 512x512x3 frames of the same stack (PredNet live work + delta), "
                      "C oracle with OpenMP on the %d host cores of this job" % (n, cores)}


if __name__ == "__main__":
    main()
